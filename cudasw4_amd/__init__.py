"""cudasw4_amd — MI355X-native Smith-Waterman protein database search (hot path behind a C ABI).

Layout:
  csrc/      HIP kernels (gfx950) + the C ABI (include/cudasw4_amd.h) + the C++ host driver
  lib/       built shared library / binaries (git-ignored, built by __graft_entry__.build())
  capi.py    ctypes binding of the C ABI
  search.py  host-side mirror of the reference's CudaSW4 driver (setDatabase / scan) on top of the C ABI
"""
__version__ = "0.1.0"
