"""cudasw4_amd — MI355X-native Smith-Waterman protein database search (hot path behind a C ABI).

Layout:
  csrc/       HIP kernels (gfx950) + the C ABI (include/cudasw4_amd.h) + the C++ host driver, makedb, align
  lib/        built shared libraries / binaries (git-ignored, built by __graft_entry__.build())
  capi.py     ctypes binding of the kernel library's C ABI (libcudasw4_amd.so)
  driver.py   ctypes binding of the C++ host driver (libcudasw4_host.so): what bench.py measures
  search.py   Python mirror of the reference's CudaSW4 driver on top of capi, for fine-grained tests
  synthdb.py  seeded synthetic databases (Swiss-Prot-like stand-in) for benchmarks and tests
"""
__version__ = "0.2.0"
