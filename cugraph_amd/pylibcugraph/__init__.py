"""Host-side mirror of the pylibcugraph objects on the PageRank path.

Same names, argument order, return values and exceptions as the reference's Cython layer
(python/pylibcugraph/pylibcugraph/{resource_handle,graph_properties,graphs,pagerank,
personalized_pagerank,utils}.pyx), bound with ctypes to libcugraph_amd_c.so instead of Cython to
libcugraph_c.  Device arrays are anything exposing ``__cuda_array_interface__`` (PyTorch-ROCm
tensors do); results come back as torch tensors on the current device (the reference returns
cupy arrays, which do not exist on this platform).
"""
from .exceptions import FailedToConvergeError  # noqa: F401
from .resource_handle import ResourceHandle  # noqa: F401
from .graph_properties import GraphProperties  # noqa: F401
from .graphs import SGGraph, MGGraph  # noqa: F401
from .pagerank import pagerank  # noqa: F401
from .personalized_pagerank import personalized_pagerank  # noqa: F401
from .centrality import eigenvector_centrality, hits, katz_centrality  # noqa: F401

__all__ = [
    "FailedToConvergeError",
    "ResourceHandle",
    "GraphProperties",
    "SGGraph",
    "MGGraph",
    "pagerank",
    "personalized_pagerank",
    "katz_centrality",
    "eigenvector_centrality",
    "hits",
]
