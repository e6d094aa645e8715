"""founderblockgraphs_amd -- MI355X-native segmentation engine for founder block graphs.

Product path: libfbg_hip.so (hand-written HIP for gfx950, C ABI in include/fbg_hip.h) and the C++
host program built from csrc/host/.  This package is the thin ctypes mirror used by the tests and
bench.py; it contains no compute and no fallback.
"""
from .api import (Engine, FbgError, Group, NoSegmentation, as_msa, segment, segment2elasticValid,  # noqa: F401
                  segment_elastic_minmaxlength)
