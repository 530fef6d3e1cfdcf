from .outputmodel import OutputModel  # noqa: F401
from .gaussian import GaussianOutputModel  # noqa: F401
from .discrete import DiscreteOutputModel  # noqa: F401
