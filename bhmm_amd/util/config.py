"""Global configuration, mirroring bhmm/util/config.py:24-37.

kernel : implementation of the hidden-variable kernels and output-model hot loops.  The
         reference offers 'python' and 'c'; this package offers 'hip' only (MI355X kernels
         behind the C ABI of include/bhmm_amd.h -- there is no CPU implementation).
dtype  : float64 only (the reference's native path rejects anything else too,
         bhmm/hidden/impl_c/hidden.pyx:59-68).
"""
import numpy as np

kernel = 'hip'
dtype = np.float64
verbose = False
