"""pecaller_amd -- MI355X-native hot path of PEMapper / PECaller behind a C-ABI.

The product is pecaller_amd/libpemap_hip.so (HIP kernels for gfx950 + the C entry points of include/pemap_hip.h)
and the C host programs under pecaller_amd/csrc/.  This Python package is a thin ctypes mirror of that C-ABI for
tests and bench.py; it never computes anything itself and raises if the library or a GPU is missing.
"""
from .pemap import PemapDev, PemapError, load_library, LIB_PATH  # noqa: F401
