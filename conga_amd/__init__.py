"""conga_amd -- MI355X-native drop-in for CONGA's read-depth / likelihood hot path.

Layout:
  csrc/        HIP kernels (gfx950) and the C-ABI of include/conga_hip.h -> libconga_hip.so
  host/        C++ host side mirroring the reference's per-chromosome driver and CLI
  capi.py      ctypes binding of the C-ABI (tests, bench, multi-GPU driver)
  synth.py     seeded synthetic inputs (GC track, read tuples, SV BEDs, mappability)
  shard.py     chromosome -> GPU partition and the result gather over torch.distributed

The product path never imports anything under oracle/.
"""
__version__ = "0.1.0"
