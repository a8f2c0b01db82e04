"""ResourceHandle (reference: python/pylibcugraph/pylibcugraph/resource_handle.pyx:25-60)."""
from .. import _capi


class ResourceHandle:
    """RAII owner of a cugraph_resource_handle_t.

    ``ResourceHandle()`` binds the current HIP device and the NULL stream (torch's default
    stream).  ``stream`` / ``rank`` / ``comm_size`` are extensions used by the 1D multi-GPU
    driver, replacing the raft comms handle the reference passes in
    (python/cugraph/cugraph/dask/link_analysis/pagerank.py:118).
    """

    def __init__(self, handle=None, *, stream=None, rank=0, comm_size=1):
        lib = _capi.load()
        self.c_resource_handle_ptr = None
        if handle is not None:
            raise ValueError("a raft handle cannot be passed on this platform; use rank/comm_size")
        if stream is None and rank == 0 and comm_size == 1:
            ptr = lib.cugraph_create_resource_handle(None)
        else:
            ptr = lib.cugraph_amd_create_resource_handle(stream, int(rank), int(comm_size))
        if not ptr:
            raise RuntimeError("cugraph_create_resource_handle failed: no usable HIP device")
        self.c_resource_handle_ptr = ptr
        self._lib = lib

    @property
    def comm_size(self):
        return self._lib.cugraph_resource_handle_get_comm_size(self.c_resource_handle_ptr)

    @property
    def rank(self):
        return self._lib.cugraph_resource_handle_get_rank(self.c_resource_handle_ptr)

    def __del__(self):
        ptr = getattr(self, "c_resource_handle_ptr", None)
        if ptr:
            self._lib.cugraph_free_resource_handle(ptr)
            self.c_resource_handle_ptr = None
