"""cugraph-level API for the PageRank path (reference: python/cugraph/cugraph/):
`Graph`, `MultiGraph`, `pagerank`, `exceptions.FailedToConvergeError`, plus the NetworkX-backend
entry `nx_pagerank` (reference: python/nx-cugraph/nx_cugraph/algorithms/link_analysis/pagerank_alg.py).

cuDF does not exist on ROCm: (vertex, value) frames and edge lists are pandas DataFrames here and
the result frame is a pandas DataFrame; the device work is the same pylibcugraph-level calls."""
from . import exceptions
from .exceptions import FailedToConvergeError
from .graph import Graph, MultiGraph
from .centrality import eigenvector_centrality, hits, katz_centrality
from .link_analysis import nx_pagerank, pagerank

__all__ = ["Graph", "MultiGraph", "pagerank", "nx_pagerank", "katz_centrality", "eigenvector_centrality", "hits",
           "FailedToConvergeError", "exceptions"]
