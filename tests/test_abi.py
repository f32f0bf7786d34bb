"""The C-ABI library loads without a GPU and exports every symbol include/mgx.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from gym_minigrid_amd import _lib


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "mgx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mgx_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    syms = declared_symbols()
    assert len(syms) >= 20 and "mgx_step" in syms and "mgx_create" in syms
    L = ctypes.CDLL(_lib.SO_PATH)
    for s in syms:
        assert hasattr(L, s), "libmgx.so does not export %s" % s
        assert s in _lib.SIGNATURES, "python binding lacks %s" % s
    assert sorted(_lib.SIGNATURES) == syms


def test_no_gpu_calls_fail_loudly_not_silently():
    """Without a HIP device mgx_create must fail with an error (there is no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = _lib.env_config("MiniGrid-Empty-8x8-v0")
    h = ctypes.c_void_p()
    rc = _lib.lib().mgx_create(ctypes.byref(cfg), 64, 0, ctypes.byref(h))
    assert rc != 0 and not h.value
    assert b"HIP" in _lib.lib().mgx_last_error() or b"device" in _lib.lib().mgx_last_error()


def test_config_validation_messages():
    c = _lib.Config()
    rc = _lib.lib().mgx_env_config(b"nope", ctypes.byref(c))
    assert rc == -5 and b"unknown env id" in _lib.lib().mgx_last_error()
    assert _lib.lib().mgx_version().startswith(b"mgx")
