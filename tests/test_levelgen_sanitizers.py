"""The level generators (levelgen_core.h: the same source k_levelgen runs on the GPU, where an out-of-bounds access can
take the node down) built for the host with AddressSanitizer + UndefinedBehaviorSanitizer and run over every built-in
id: batches of seeds and level streams.  (GPU sanitizers are not available on the pool; this is the CPU build.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gym-minigrid_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_level_generators_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "asan_levelgen")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-D__host__=", "-D__device__=",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, os.path.join(ROOT, "tests", "asan_levelgen.cpp"), os.path.join(CSRC, "levelgen.cpp"),
           "-o", exe, "-lpthread"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe, "400"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "asan levelgen ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
