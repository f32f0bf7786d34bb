"""Empty stand-in so that the reference's SAR map modules import; unused by the hot path."""
