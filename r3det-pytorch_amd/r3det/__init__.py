"""r3det (MI355X-native): drop-in operator package for the r3det custom-op hot path.

Import as ``r3det`` after putting ``<repo>/r3det-pytorch_amd`` on ``sys.path``; the public
names of ``r3det.ops`` (and its per-op subpackages), ``r3det.core`` (``core.anchor``,
``core.bbox``, ``core.post_processing``) are those of the reference package
(r3det/__init__.py:4-7 star-imports core and ops the same way).
"""
from .core import *  # noqa: F401, F403
from .ops import *  # noqa: F401, F403

__version__ = '0.2.0'
short_version = __version__
