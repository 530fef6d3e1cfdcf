"""Initial-model heuristics: what `estimate_hmm` needs to
start from raw data.  Host-side numpy; the E-step that follows is the accelerated path."""
from .gaussian import init_model_gaussian1d, fit_gmm1d  # noqa: F401
from .discrete import init_discrete_hmm_spectral, pcca_memberships, count_matrix  # noqa: F401
