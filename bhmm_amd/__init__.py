"""bhmm_amd -- MI355X-native forward-backward / Baum-Welch engine with the bhmm hot-path API.

Only the hot path of bhmm (bhmm/hidden + its drivers in bhmm/estimators) lives here; see
DESIGN.md.  The compute path is the HIP library bhmm_amd/lib/libbhmm_amd.so (C ABI in
include/bhmm_amd.h); there is no CPU fallback.
"""
__version__ = "0.1"
