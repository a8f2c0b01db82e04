"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's PageRank hot path (see pagerank_oracle.c for the
reference file:line each function follows) plus the NetworkX oracle named by
BASELINE.json's north_star.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; cugraph_amd/ never does.

Pinning (tests/test_oracle.py): the C restatement and the NetworkX/SciPy oracle are
checked against every golden vector the reference's own tests hold for this path
(tests/golden/reference_vectors.json cites them) and against each other.

The reference itself (libcugraph: CUDA + RAFT/RMM/CCCL fetched by CMake, gtest) is
unbuildable in this image, so there is no oracle/_ref build; see DESIGN.md.
"""
from .c_oracle import (  # noqa: F401
    build,
    lib,
    compress,
    pagerank,
    rmat_edges,
    rmat_weights,
    rmat_thresholds,
    num_threads,
)
from .np_oracle import csc_lexsort, nx_pagerank, scipy_pagerank  # noqa: F401
