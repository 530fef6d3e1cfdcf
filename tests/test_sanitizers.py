"""CPU sanitizer run of the host-side native code ("run CPU restatement under
ASAN/UBSAN in tests"): `make -C oracle asan` builds oracle/asan_driver.cpp with the oracle's C
restatement and the product's host-only translation units (transition-matrix estimators, Gibbs
parameter samplers, generator, chunk / segment planners) under -fsanitize=address,undefined and runs
it on ragged lengths, T = 1, 1e5 short trajectories, re-plan paths and degenerate count matrices.
GPU code cannot run under a sanitizer on this pool."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_and_ubsan():
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no host compiler")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode("utf-8", "replace")
    if r.returncode != 0 and ("cannot find -lasan" in out or "libasan" in out and "No such file" in out):
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, out[-4000:]
    assert "asan_driver: ok" in out
    assert "runtime error" not in out and "AddressSanitizer" not in out, out[-4000:]
