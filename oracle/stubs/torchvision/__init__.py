"""Import-time stand-in for torchvision (absent in this image). TEST INFRASTRUCTURE ONLY."""
from . import models, utils, ops  # noqa: F401
__version__ = "0.0-stub"
