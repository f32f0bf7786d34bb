def label(*a, **k):
    raise NotImplementedError("skimage stand-in")
