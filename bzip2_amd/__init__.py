"""bzip2_amd -- MI355X-native bzip2 block compressor (drop-in for bastie/bzip2's encoder path).

Host-side mirror of the reference interface lives in bzip2_amd.streams; the compute path is
libbz2hip.so (HIP, gfx950) behind the C ABI of include/bz2hip.h.
"""
from ._lib import Context, BZ2HipError, PinnedBuffer, load, SO_PATH, SYMBOLS  # noqa: F401
