"""orbslamm_amd -- MI355X-native ORB front-end (extract + Hamming match) for ORBSLAMM.

Only what the hot path needs: csrc/ (HIP kernels + C ABI), and thin ctypes mirrors
of the reference's ORBextractor / ORBmatcher interfaces."""
from .extractor import ORBextractor, unpack_candidates  # noqa: F401
from .matcher import ORBmatcher, make_grid  # noqa: F401
from .vocabulary import ORBVocabulary  # noqa: F401
from ._lib import KP_DTYPE, OrbError  # noqa: F401
