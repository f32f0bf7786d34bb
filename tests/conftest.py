import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gym-minigrid_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases():
    """Trace cases whose state can be injected with set_state (DynObs-* carry an RNG stream: tests/test_dynobs.py)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and not f.startswith(("DynObs-", "Bonus-", "Dac-"))
                  and f not in ("levels.npz", "level_streams.npz", "onehot.npz", "flat.npz", "levels_obstructed.npz"))


def bonus_cases():
    """Traces recorded through the reference's ActionBonus / StateBonus wrappers (meta['bonus'] = the stacking order, innermost first)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("Bonus-") and f.endswith(".npz"))


def dac_cases():
    """Traces recorded through the fork's DACWrapper (meta['bonus'] = the wrapper stack, innermost first, "dac" among them)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("Dac-") and f.endswith(".npz"))


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return meta, z


@pytest.fixture(scope="session")
def levels():
    return np.load(os.path.join(GOLDEN, "levels.npz"))
