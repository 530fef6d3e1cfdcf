"""The C-ABI library loads on a GPU-less host and exports every symbol include/bhmm_amd.h
declares; argument validation works without a device (no compute calls here)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "bhmm_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bhmm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from bhmm_amd import _lib
    L = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(L, name), name
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_error_channel():
    from bhmm_amd import _lib
    L = _lib.load()
    assert b"gfx950" in L.bhmm_version()
    h = ctypes.c_void_p()
    if L.bhmm_device_count() == 0:
        rc = L.bhmm_ctx_create(ctypes.byref(h), 0, None)
        assert rc == _lib.ERR_NO_DEVICE
        assert b"device" in L.bhmm_last_error()
        with pytest.raises(_lib.BhmmAmdError):
            _lib.require_device()


def test_product_fails_loudly_without_gpu():
    """No silent CPU fallback anywhere in the product path."""
    from bhmm_amd import _lib
    if _lib.load().bhmm_device_count() > 0:
        pytest.skip("GPU present")
    import bhmm_amd
    A = np.array([[0.9, 0.1], [0.1, 0.9]])
    pobs = np.full((5, 2), 0.5)
    with pytest.raises(_lib.BhmmAmdError):
        bhmm_amd.hidden.forward(A, pobs, np.array([0.5, 0.5]))
    with pytest.raises(_lib.BhmmAmdError):
        bhmm_amd.hidden.viterbi(A, pobs, np.array([0.5, 0.5]))
    with pytest.raises(_lib.BhmmAmdError):
        bhmm_amd.GaussianOutputModel(2, [0, 1], [1, 1]).p_obs(np.zeros(4))
    model = bhmm_amd.gaussian_hmm([0.5, 0.5], A, [0.0, 1.0], [1.0, 1.0])
    with pytest.raises(_lib.BhmmAmdError):
        bhmm_amd.MaximumLikelihoodEstimator([np.zeros(10)], 2, initial_model=model)


def test_libc_uniform_helper_matches_reference_stream():
    from bhmm_amd import _lib
    from oracle import oracle as orc
    L = _lib.load()
    u = np.empty(1000)
    assert L.bhmm_libc_uniforms(_lib.dp(u), 1000, 123) == 0
    assert np.array_equal(u, orc.libc_uniforms(1000, 123))


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bhmm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "liboracle" in src \
                        or "bhmm_oracle" in src:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_tools_do_not_use_the_oracle_either():
    """tools/ holds measurement scripts for the GPU box; whatever needs the oracle lives under tests/."""
    bad = []
    for f in os.listdir(os.path.join(ROOT, "tools")):
        path = os.path.join(ROOT, "tools", f)
        if os.path.isfile(path) and f.endswith((".py", ".sh")):
            src = open(path).read()
            if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "liboracle" in src:
                bad.append(f)
    assert not bad, bad
