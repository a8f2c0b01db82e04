"""cugraph_amd -- MI355X-native PageRank behind the cugraph / pylibcugraph API of this path.

Layout:
  csrc/            hand-written gfx950 HIP kernels + the C ABI (libcugraph_amd_c.so)
  _capi.py         ctypes binding of include/cugraph_amd_c.h
  pylibcugraph/    mirror of the reference's Cython layer (ResourceHandle, SGGraph, pagerank, ...)
  link_analysis.py cugraph.pagerank-shaped wrapper (Graph + DataFrame in/out)
  generators.py    R-MAT edge lists on the device (benchmark inputs)
  mg.py            1D destination-partitioned multi-GPU driver over torch.distributed (RCCL)
"""
__version__ = "0.1.0"
