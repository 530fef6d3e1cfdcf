"""bhmm_amd -- MI355X-native forward-backward / Baum-Welch engine with the bhmm hot-path API.

Only the hot path of bhmm (bhmm/hidden + its drivers in bhmm/estimators) lives here; see
DESIGN.md.  The compute path is the HIP library bhmm_amd/lib/libbhmm_amd.so (C ABI in
include/bhmm_amd.h); there is no CPU fallback.

Names follow the reference (bhmm/__init__.py:23-43); the two identifiers BASELINE.json uses
(PyEMMA spellings) are provided as aliases.
"""
from .util import config  # noqa: F401
from .util import testsystems  # noqa: F401
from . import hidden  # noqa: F401
from .hmm import (HMM, SampledHMM, GaussianHMM, DiscreteHMM, SampledGaussianHMM,  # noqa: F401
                  SampledDiscreteHMM)
from .output_models import OutputModel, GaussianOutputModel, DiscreteOutputModel  # noqa: F401
from .estimators.maximum_likelihood import MaximumLikelihoodEstimator  # noqa: F401
from .estimators.bayesian_sampling import BayesianHMMSampler  # noqa: F401
from .api import (estimate_hmm, bayesian_hmm, lag_observations, gaussian_hmm,  # noqa: F401
                  discrete_hmm, init_hmm, init_gaussian_hmm, init_discrete_hmm)

MLHMM = MaximumLikelihoodEstimator          # bhmm/__init__.py:36
BHMM = BayesianHMMSampler                   # bhmm/__init__.py:35
MaximumLikelihoodHMM = MaximumLikelihoodEstimator   # name used by BASELINE.json
BayesianHMM = BayesianHMMSampler                    # name used by BASELINE.json

__version__ = "0.1"
version = __version__          # (bhmm/__init__.py exposes `version` as well)
