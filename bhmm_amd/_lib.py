"""ctypes binding of the C ABI declared in include/bhmm_amd.h.

The shared library is built in-tree (bhmm_amd/lib/libbhmm_amd.so, see bhmm_amd/csrc/Makefile
and __graft_entry__.build()).  There is NO fallback: if the library is missing, or no HIP
device is visible when a kernel is requested, the call raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BHMM_AMD_LIB selects another build of the same library (kernel experiments, tools/build_variant.sh)
LIB_PATH = os.environ.get("BHMM_AMD_LIB") or os.path.join(_HERE, "lib", "libbhmm_amd.so")

OK = 0
ERR_NO_MEM = 2
ERR_INVALID = 3
ERR_HIP = 4
ERR_NONFINITE = 5
ERR_CHOICE = 6
ERR_NO_DEVICE = 7
ERR_SIGMA = 8
ERR_DISCONNECTED = 9

EMIT_GAUSSIAN = 0
EMIT_DISCRETE = 1
EMIT_EXPLICIT = 2

FLAG_STORE_GAMMA = 1

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_int64_p = ctypes.POINTER(ctypes.c_int64)
c_void_p = ctypes.c_void_p

# every symbol include/bhmm_amd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "bhmm_last_error": (ctypes.c_char_p, []),
    "bhmm_device_count": (ctypes.c_int, []),
    "bhmm_version": (ctypes.c_char_p, []),
    "bhmm_forward": (ctypes.c_int, [c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                    ctypes.c_int, ctypes.c_int64]),
    "bhmm_backward": (ctypes.c_int, [c_double_p, c_double_p, c_double_p, ctypes.c_int,
                                     ctypes.c_int64]),
    "bhmm_state_probabilities": (ctypes.c_int, [c_double_p, c_double_p, c_double_p, ctypes.c_int,
                                                ctypes.c_int64]),
    "bhmm_transition_counts": (ctypes.c_int, [c_double_p, c_double_p, c_double_p, c_double_p,
                                              c_double_p, ctypes.c_int, ctypes.c_int64]),
    "bhmm_viterbi": (ctypes.c_int, [c_int32_p, c_double_p, c_double_p, c_double_p, ctypes.c_int,
                                    ctypes.c_int64]),
    "bhmm_sample_path": (ctypes.c_int, [c_int32_p, c_double_p, c_double_p, c_double_p,
                                        ctypes.c_int, ctypes.c_int64]),
    "bhmm_libc_uniforms": (ctypes.c_int, [c_double_p, ctypes.c_int64, ctypes.c_int]),
    "bhmm_pobs_gaussian": (ctypes.c_int, [c_double_p, c_double_p, c_double_p, c_double_p,
                                          ctypes.c_int, ctypes.c_int64, ctypes.c_int]),
    "bhmm_update_pout": (ctypes.c_int, [c_double_p, c_int32_p, c_double_p, ctypes.c_int64,
                                        ctypes.c_int, ctypes.c_int]),
    "bhmm_ctx_create": (ctypes.c_int, [ctypes.POINTER(c_void_p), ctypes.c_int, c_void_p]),
    "bhmm_ctx_destroy": (ctypes.c_int, [c_void_p]),
    "bhmm_ctx_set_observations": (ctypes.c_int, [c_void_p, ctypes.c_int, c_void_p, c_int64_p,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int]),
    "bhmm_ctx_set_observations_lagged": (ctypes.c_int, [c_void_p, ctypes.c_int, c_void_p, c_int64_p,
                                                        ctypes.c_int, ctypes.c_int, c_int32_p,
                                                        c_int32_p, ctypes.c_int, ctypes.c_int,
                                                        ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "bhmm_ctx_stats_size": (ctypes.c_int, [c_void_p]),
    "bhmm_estep": (ctypes.c_int, [c_void_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                  c_void_p, ctypes.c_int]),
    "bhmm_estep_fetch": (ctypes.c_int, [c_void_p, c_double_p, c_double_p]),
    "bhmm_get_gamma": (ctypes.c_int, [c_void_p, ctypes.c_int, c_double_p]),
    "bhmm_viterbi_batch": (ctypes.c_int, [c_void_p, c_double_p, c_double_p, c_double_p,
                                          c_double_p, c_int32_p]),
    "bhmm_viterbi_batch_u8": (ctypes.c_int, [c_void_p, c_double_p, c_double_p, c_double_p,
                                             c_double_p, c_void_p, ctypes.c_int]),
    "bhmm_ctx_path_stats_size": (ctypes.c_int, [c_void_p]),
    "bhmm_ctx_set_stream_offsets": (ctypes.c_int, [c_void_p, c_int64_p]),
    "bhmm_sample_paths_dev": (ctypes.c_int, [c_void_p, c_double_p, c_double_p, c_double_p,
                                             c_double_p, c_double_p, ctypes.c_uint64, c_int32_p,
                                             c_void_p]),
    "bhmm_sample_paths": (ctypes.c_int, [c_void_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                         c_double_p, ctypes.c_uint64, c_int32_p, c_int64_p,
                                         c_int64_p, c_double_p]),
    "bhmm_ctx_set_option": (ctypes.c_int, [c_void_p, ctypes.c_char_p, ctypes.c_double]),
    "bhmm_ctx_get_option": (ctypes.c_int, [c_void_p, ctypes.c_char_p, c_double_p]),
    "bhmm_ctx_total_steps": (ctypes.c_int64, [c_void_p]),
    "bhmm_ctx_num_chunks": (ctypes.c_int, [c_void_p]),
    "bhmm_ctx_chunk_len": (ctypes.c_int, [c_void_p]),
    "bhmm_ctx_last_kernel_ms": (ctypes.c_double, [c_void_p, ctypes.c_int]),
    "bhmm_ctx_last_kernel_ms_all": (ctypes.c_int, [c_void_p, c_double_p]),
    "bhmm_ctx_stream": (c_void_p, [c_void_p]),
    "bhmm_ctx_sync": (ctypes.c_int, [c_void_p]),
    "bhmm_comm_unique_id": (ctypes.c_int, [c_void_p]),
    "bhmm_comm_init_rank": (ctypes.c_int, [ctypes.POINTER(c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           c_void_p]),
    "bhmm_comm_destroy": (ctypes.c_int, [c_void_p]),
    "bhmm_comm_size": (ctypes.c_int, [c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "bhmm_ctx_allreduce_stats": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_int64]),
    "bhmm_synth_observations": (ctypes.c_int, [c_void_p, c_void_p, ctypes.c_int, c_void_p,
                                               ctypes.c_int, c_double_p, c_double_p, c_double_p,
                                               c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int64, ctypes.c_uint64]),
    "bhmm_synth_observations_at": (ctypes.c_int, [c_void_p, c_void_p, ctypes.c_int, c_void_p,
                                                  ctypes.c_int, c_double_p, c_double_p, c_double_p,
                                                  c_double_p, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int64, ctypes.c_uint64,
                                                  ctypes.c_int64]),
    "bhmm_mle_reversible": (ctypes.c_int, [c_double_p, c_int64_p, c_double_p, ctypes.c_int,
                                           ctypes.c_int64, ctypes.c_double]),
    "bhmm_mstep": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p,
                                  c_double_p, c_double_p, ctypes.c_int, ctypes.c_int, c_double_p,
                                  ctypes.c_int64, ctypes.c_double, ctypes.c_double, c_double_p,
                                  c_double_p, c_double_p, c_double_p, c_int32_p, c_double_p]),
    "bhmm_gibbs_parameters": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p,
                                             c_double_p, c_double_p, c_double_p, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int64, ctypes.c_uint64,
                                             ctypes.c_uint64, c_double_p, c_double_p, c_double_p,
                                             c_double_p, c_int32_p]),
    "bhmm_host_connected_sets": (ctypes.c_int, [c_int32_p, c_double_p, ctypes.c_int,
                                                ctypes.c_double, ctypes.c_int]),
    "bhmm_host_stationary_vector": (ctypes.c_int, [c_double_p, c_double_p, ctypes.c_int]),
    "bhmm_host_estimate_tmatrix": (ctypes.c_int, [c_double_p, c_double_p, ctypes.c_int, ctypes.c_int,
                                            c_double_p, ctypes.c_int64, ctypes.c_double,
                                            ctypes.c_double, c_int64_p]),
    "bhmm_host_is_reversible": (ctypes.c_int, [c_double_p, ctypes.c_int]),
    "bhmm_host_sample_reversible": (ctypes.c_int, [c_double_p, c_double_p, ctypes.c_int, ctypes.c_int64,
                                                   ctypes.c_uint64, ctypes.c_int]),
    "bhmm_host_partial_rev": (ctypes.c_int, [c_double_p, c_double_p, ctypes.c_int, c_int32_p,
                                             ctypes.c_int64, ctypes.c_double]),
    "bhmm_host_rng_draws": (ctypes.c_int, [c_double_p, ctypes.c_int64, ctypes.c_int,
                                           ctypes.c_double, ctypes.c_uint64, ctypes.c_uint64]),
    "bhmm_diag_exp_nonpos": (ctypes.c_int, [c_double_p, c_double_p, ctypes.c_int64]),
    "bhmm_diag_gauss_pdf": (ctypes.c_int, [c_double_p, c_double_p, ctypes.c_int64, ctypes.c_double,
                                           ctypes.c_double, ctypes.c_int]),
}

_lib = None


class BhmmAmdError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "bhmm_amd error %d: %s" % (code, msg))
        self.code = code


def load():
    """Load libbhmm_amd.so and attach prototypes.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "bhmm_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C bhmm_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the library does not export the symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    """Map a status code to the exception the reference would raise
    (hidden.pyx:150-151 MemoryError; maximum_likelihood.py:385 AssertionError)."""
    if rc == OK:
        return
    msg = load().bhmm_last_error().decode("utf-8", "replace")
    if rc == ERR_NO_MEM:
        raise MemoryError(msg)
    if rc == ERR_INVALID:
        raise ValueError(msg)
    if rc == ERR_NONFINITE:
        raise AssertionError(msg)
    if rc == ERR_SIGMA:
        raise RuntimeError(msg)          # gaussian.py:271-272
    if rc == ERR_DISCONNECTED:
        raise NotImplementedError(msg)   # bayesian_sampling.py:347-350
    raise BhmmAmdError(rc, msg)


def dp(a):
    return a.ctypes.data_as(c_double_p) if a is not None else None


def ip(a):
    return a.ctypes.data_as(c_int32_p) if a is not None else None


def lp(a):
    return a.ctypes.data_as(c_int64_p) if a is not None else None


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def require_device():
    n = load().bhmm_device_count()
    if n <= 0:
        raise BhmmAmdError(ERR_NO_DEVICE, "no HIP device visible: the 'hip' implementation "
                           "needs an AMD GPU and has no CPU fallback")
    return n
