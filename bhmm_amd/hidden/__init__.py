from .api import *  # noqa: F401,F403
