"""dvslam_amd — thin ctypes driver over libdvslam_hip.so for tests and bench.py.

The product is the C-ABI library (include/dvslam_hip.h) and the C++ adapters in include/dvslam/;
this package only mirrors the reference's three call signatures in Python so the parity tests read
like calls into the reference (ORBextractor.hpp:50-60, cv::BFMatcher::match, SlidingWindowBA).
There is NO CPU fallback: every class raises if the library or a gfx950 device is missing.
"""
from ._lib import lib, DvsError, KP_DTYPE, device_count, build_library, stream_create, stream_destroy  # noqa: F401
from .orb import ORBextractor  # noqa: F401
from .matcher import BFMatcher  # noqa: F401
from .ba import BAProblem, SlidingWindowBA  # noqa: F401
from .glue import FrontendGlue  # noqa: F401
from .cvorb import CvORB  # noqa: F401
