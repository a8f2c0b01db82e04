"""SGGraph / MGGraph (reference: python/pylibcugraph/pylibcugraph/graphs.pyx:52-262, 265-420)."""
import ctypes

from .. import _capi
from .graph_properties import GraphProperties
from .resource_handle import ResourceHandle
from .utils import assert_CAI_type, create_view_from_py_obj, free_view


class _GPUGraph:
    c_graph_ptr = None

    def __del__(self):
        ptr = getattr(self, "c_graph_ptr", None)
        if ptr:
            try:
                _capi.load().cugraph_sg_graph_free(ptr)
            except TypeError:  # interpreter shutdown: the module globals are already gone
                pass
            self.c_graph_ptr = None

    def info(self):
        """(extension) dict with the graph's shape as held by the library."""
        lib = _capi.load()
        arr = (ctypes.c_int64 * 8)()
        err = ctypes.c_void_p()
        code = lib.cugraph_amd_graph_info(self.c_graph_ptr, arr, ctypes.byref(err))
        _capi.check(code, err, "cugraph_amd_graph_info")
        keys = ["num_vertices", "num_edges", "num_global_vertices", "vertex_first",
                "store_transposed", "offsets_int64", "weighted", "renumbered"]
        return dict(zip(keys, [int(x) for x in arr]))


class SGGraph(_GPUGraph):
    """RAII owner of a single-GPU cugraph_graph_t; arguments as graphs.pyx:144-158."""

    def __init__(self,
                 resource_handle,
                 graph_properties,
                 src_or_offset_array,
                 dst_or_index_array,
                 weight_array=None,
                 store_transposed=False,
                 renumber=False,
                 do_expensive_check=False,
                 edge_id_array=None,
                 edge_type_array=None,
                 input_array_format="COO",
                 vertices_array=None,
                 drop_self_loops=False,
                 drop_multi_edges=False):
        if not isinstance(resource_handle, ResourceHandle):
            raise TypeError("resource_handle must be a ResourceHandle")
        if not isinstance(graph_properties, GraphProperties):
            raise TypeError("graph_properties must be a GraphProperties")
        if not isinstance(store_transposed, (int, bool)):
            raise TypeError("expected int or bool for store_transposed, got "
                            f"{type(store_transposed)}")
        if not isinstance(renumber, (int, bool)):
            raise TypeError(f"expected int or bool for renumber, got {type(renumber)}")
        if not isinstance(do_expensive_check, (int, bool)):
            raise TypeError("expected int or bool for do_expensive_check, got "
                            f"{type(do_expensive_check)}")
        assert_CAI_type(src_or_offset_array, "src_or_offset_array")
        assert_CAI_type(dst_or_index_array, "dst_or_index_array")
        assert_CAI_type(vertices_array, "vertices_array", True)
        assert_CAI_type(weight_array, "weight_array", True)
        assert_CAI_type(edge_id_array, "edge_id_array", True)
        assert_CAI_type(edge_type_array, "edge_type_array", True)
        if input_array_format not in ("COO", "CSR"):
            raise ValueError("invalid 'input_array_format'. Only "
                             "'COO' and 'CSR' format are supported.")

        lib = _capi.load()
        views = [create_view_from_py_obj(a) for a in (src_or_offset_array, dst_or_index_array,
                                                      vertices_array, weight_array, edge_id_array,
                                                      edge_type_array)]
        srcs_v, dsts_v, verts_v, wgt_v, eid_v, etype_v = views
        graph = ctypes.c_void_p()
        err = ctypes.c_void_p()
        try:
            if input_array_format == "COO":
                code = lib.cugraph_graph_create_sg(
                    resource_handle.c_resource_handle_ptr,
                    ctypes.byref(graph_properties.c_graph_properties), verts_v, srcs_v, dsts_v,
                    wgt_v, eid_v, etype_v, int(bool(store_transposed)), int(bool(renumber)),
                    int(bool(drop_self_loops)), int(bool(drop_multi_edges)),
                    int(bool(do_expensive_check)), ctypes.byref(graph), ctypes.byref(err))
                _capi.check(code, err, "cugraph_graph_create_sg()")
            else:
                code = lib.cugraph_sg_graph_create_from_csr(
                    resource_handle.c_resource_handle_ptr,
                    ctypes.byref(graph_properties.c_graph_properties), srcs_v, dsts_v, wgt_v,
                    eid_v, etype_v, int(bool(store_transposed)), int(bool(renumber)),
                    int(bool(do_expensive_check)), ctypes.byref(graph), ctypes.byref(err))
                _capi.check(code, err, "cugraph_sg_graph_create_from_csr()")
        finally:
            for v in views:
                free_view(v)
        self.c_graph_ptr = graph
        self._resource_handle = resource_handle  # the library keeps no reference; we do


class MGGraph(_GPUGraph):
    """Local part of a 1D destination-partitioned graph (one rank = one GPU = one process).

    The reference's MGGraph (graphs.pyx:265-420) shuffles edges over a 2D GPU grid through raft
    comms; here the multi-GPU driver (cugraph_amd.mg) routes every edge to the rank that owns its
    destination and hands this class exactly those edges.
    """

    def __init__(self, resource_handle, graph_properties, src_array, dst_array, weight_array=None,
                 *, vertex_first, vertex_last, num_global_vertices, num_source_slots=None,
                 x_pos=None, hot_slots=0):
        """num_source_slots / x_pos: compact source space (src_array then holds slots of the
        gathered vector, see cugraph_amd_graph_create_local_compact).  hot_slots: the slots below
        it are the first of two regions of that vector (cugraph_amd_graph_set_hot_slots)."""
        assert_CAI_type(src_array, "src_array")
        assert_CAI_type(dst_array, "dst_array")
        assert_CAI_type(weight_array, "weight_array", True)
        assert_CAI_type(x_pos, "x_pos", True)
        lib = _capi.load()
        views = [create_view_from_py_obj(a) for a in (src_array, dst_array, weight_array, x_pos)]
        graph = ctypes.c_void_p()
        err = ctypes.c_void_p()
        try:
            if num_source_slots is None:
                code = lib.cugraph_amd_graph_create_local(
                    resource_handle.c_resource_handle_ptr,
                    ctypes.byref(graph_properties.c_graph_properties), views[0], views[1], views[2],
                    int(vertex_first), int(vertex_last), int(num_global_vertices),
                    ctypes.byref(graph), ctypes.byref(err))
            else:
                code = lib.cugraph_amd_graph_create_local_compact(
                    resource_handle.c_resource_handle_ptr,
                    ctypes.byref(graph_properties.c_graph_properties), views[0], views[1], views[2],
                    int(vertex_first), int(vertex_last), int(num_global_vertices),
                    int(num_source_slots), views[3], ctypes.byref(graph), ctypes.byref(err))
            _capi.check(code, err, "cugraph_amd_graph_create_local()")
        finally:
            for v in views:
                free_view(v)
        self.c_graph_ptr = graph
        self._resource_handle = resource_handle
        if hot_slots:
            err = ctypes.c_void_p()
            code = lib.cugraph_amd_graph_set_hot_slots(graph, int(hot_slots), ctypes.byref(err))
            _capi.check(code, err, "cugraph_amd_graph_set_hot_slots()")
