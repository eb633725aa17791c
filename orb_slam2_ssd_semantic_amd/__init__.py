"""MI355X-native ORB front-end (ORBextractor + ORBmatcher Hamming core) behind a C-ABI.

Product path: liborbfe.so (hand-written HIP for gfx950), bound through ctypes.  Nothing in this
package imports the CPU oracle under oracle/ -- that is test infrastructure.
"""
from .extractor import ORBextractor  # noqa: F401
from .matcher import FrameGrid, ORBVocabulary, ORBmatcher, feature_vector_to_csr  # noqa: F401
from ._ffi import KP_DTYPE, OrbfeError  # noqa: F401
from . import mapio  # noqa: F401
from .mapio import VocabularyFile  # noqa: F401
from .pipeline import FramePipeline  # noqa: F401

__all__ = ["ORBextractor", "ORBmatcher", "FramePipeline", "FrameGrid", "ORBVocabulary", "VocabularyFile", "mapio", "feature_vector_to_csr",
           "KP_DTYPE", "OrbfeError"]
