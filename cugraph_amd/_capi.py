"""ctypes binding of libcugraph_amd_c.so (include/cugraph_amd_c.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C cugraph_amd/csrc``.
There is no fallback: if the shared object is missing or a symbol cannot be bound, importing
the compute API raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CUGRAPH_AMD_LIBRARY: another build of the same library (e.g. the timing-aid build, `make AIDS=1`)
LIB_PATH = os.environ.get("CUGRAPH_AMD_LIBRARY") or os.path.join(_HERE, "libcugraph_amd_c.so")

# cugraph_data_type_id_t (reference: cpp/include/cugraph_c/resource_handle.h:35)
INT32, INT64, FLOAT32, FLOAT64, SIZE_T = 0, 1, 2, 3, 4
# cugraph_error_code_t (reference: cpp/include/cugraph_c/error.h:25-33)
(CUGRAPH_SUCCESS, CUGRAPH_UNKNOWN_ERROR, CUGRAPH_INVALID_HANDLE, CUGRAPH_ALLOC_ERROR,
 CUGRAPH_INVALID_INPUT, CUGRAPH_NOT_IMPLEMENTED, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION) = range(7)

_vp = ctypes.c_void_p
_vpp = ctypes.POINTER(ctypes.c_void_p)
_sz = ctypes.c_size_t
_int = ctypes.c_int
_dbl = ctypes.c_double
_i64 = ctypes.c_int64
_u64 = ctypes.c_uint64


class GraphPropertiesStruct(ctypes.Structure):
    _fields_ = [("is_symmetric", ctypes.c_int), ("is_multigraph", ctypes.c_int)]


# name -> (restype, argtypes).  Every symbol include/cugraph_amd_c.h declares is listed here;
# tests/test_capi_symbols.py checks the two against each other.
SIGNATURES = {
    "cugraph_error_message": (ctypes.c_char_p, [_vp]),
    "cugraph_error_free": (None, [_vp]),
    "cugraph_create_resource_handle": (_vp, [_vp]),
    "cugraph_resource_handle_get_comm_size": (_int, [_vp]),
    "cugraph_resource_handle_get_rank": (_int, [_vp]),
    "cugraph_free_resource_handle": (None, [_vp]),
    "cugraph_type_erased_device_array_create": (_int, [_vp, _sz, _int, _vpp, _vpp]),
    "cugraph_type_erased_device_array_create_from_view": (_int, [_vp, _vp, _vpp, _vpp]),
    "cugraph_type_erased_device_array_free": (None, [_vp]),
    "cugraph_type_erased_device_array_view": (_vp, [_vp]),
    "cugraph_type_erased_device_array_view_as_type": (_int, [_vp, _int, _vpp, _vpp]),
    "cugraph_type_erased_device_array_view_create": (_vp, [_vp, _sz, _int]),
    "cugraph_type_erased_device_array_view_free": (None, [_vp]),
    "cugraph_type_erased_device_array_view_size": (_sz, [_vp]),
    "cugraph_type_erased_device_array_view_type": (_int, [_vp]),
    "cugraph_type_erased_device_array_view_pointer": (_vp, [_vp]),
    "cugraph_type_erased_host_array_create": (_int, [_vp, _sz, _int, _vpp, _vpp]),
    "cugraph_type_erased_host_array_free": (None, [_vp]),
    "cugraph_type_erased_host_array_view": (_vp, [_vp]),
    "cugraph_type_erased_host_array_view_create": (_vp, [_vp, _sz, _int]),
    "cugraph_type_erased_host_array_view_free": (None, [_vp]),
    "cugraph_type_erased_host_array_size": (_sz, [_vp]),
    "cugraph_type_erased_host_array_type": (_int, [_vp]),
    "cugraph_type_erased_host_array_pointer": (_vp, [_vp]),
    "cugraph_type_erased_host_array_view_copy": (_int, [_vp, _vp, _vp, _vpp]),
    "cugraph_type_erased_device_array_view_copy_from_host": (_int, [_vp, _vp, _vp, _vpp]),
    "cugraph_type_erased_device_array_view_copy_to_host": (_int, [_vp, _vp, _vp, _vpp]),
    "cugraph_type_erased_device_array_view_copy": (_int, [_vp, _vp, _vp, _vpp]),
    "cugraph_graph_create_sg": (_int, [_vp, ctypes.POINTER(GraphPropertiesStruct), _vp, _vp, _vp,
                                       _vp, _vp, _vp, _int, _int, _int, _int, _int, _vpp, _vpp]),
    "cugraph_sg_graph_create": (_int, [_vp, ctypes.POINTER(GraphPropertiesStruct), _vp, _vp, _vp,
                                       _vp, _vp, _int, _int, _int, _vpp, _vpp]),
    "cugraph_graph_create_sg_from_csr": (_int, [_vp, ctypes.POINTER(GraphPropertiesStruct), _vp, _vp,
                                                _vp, _vp, _vp, _int, _int, _int, _vpp, _vpp]),
    "cugraph_sg_graph_create_from_csr": (_int, [_vp, ctypes.POINTER(GraphPropertiesStruct), _vp, _vp,
                                                _vp, _vp, _vp, _int, _int, _int, _vpp, _vpp]),
    "cugraph_graph_free": (None, [_vp]),
    "cugraph_sg_graph_free": (None, [_vp]),
    "cugraph_mg_graph_free": (None, [_vp]),
    "cugraph_mg_graph_create": (_int, [_vp, ctypes.POINTER(GraphPropertiesStruct), _vp, _vp, _vp, _vp, _vp, _int,
                                       _sz, _int, _vpp, _vpp]),
    "cugraph_graph_create_mg": (_int, [_vp, ctypes.POINTER(GraphPropertiesStruct), _vp, _vp, _vp, _vp, _vp, _vp,
                                       _int, _sz, _int, _int, _int, _vpp, _vpp]),
    "cugraph_centrality_result_get_vertices": (_vp, [_vp]),
    "cugraph_centrality_result_get_values": (_vp, [_vp]),
    "cugraph_centrality_result_get_num_iterations": (_sz, [_vp]),
    "cugraph_centrality_result_converged": (_int, [_vp]),
    "cugraph_centrality_result_free": (None, [_vp]),
    "cugraph_eigenvector_centrality": (_int, [_vp, _vp, _dbl, _sz, _int, _vpp, _vpp]),
    "cugraph_katz_centrality": (_int, [_vp, _vp, _vp, _dbl, _dbl, _dbl, _sz, _int, _vpp, _vpp]),
    "cugraph_hits": (_int, [_vp, _vp, _dbl, _sz, _vp, _vp, _int, _int, _vpp, _vpp]),
    "cugraph_hits_result_get_vertices": (_vp, [_vp]),
    "cugraph_hits_result_get_hubs": (_vp, [_vp]),
    "cugraph_hits_result_get_authorities": (_vp, [_vp]),
    "cugraph_hits_result_get_hub_score_differences": (_dbl, [_vp]),
    "cugraph_hits_result_get_number_of_iterations": (_sz, [_vp]),
    "cugraph_hits_result_free": (None, [_vp]),
    "cugraph_pagerank": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _sz, _int, _vpp, _vpp]),
    "cugraph_pagerank_allow_nonconvergence": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _sz,
                                                     _int, _vpp, _vpp]),
    "cugraph_personalized_pagerank": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl,
                                             _sz, _int, _vpp, _vpp]),
    "cugraph_personalized_pagerank_allow_nonconvergence": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                                  _vp, _dbl, _dbl, _sz, _int, _vpp,
                                                                  _vpp]),
    # ---- extensions
    "cugraph_amd_create_resource_handle": (_vp, [_vp, _int, _int]),
    "cugraph_amd_resource_handle_set_comm": (_int, [_vp, _vp, _vpp]),
    "cugraph_amd_graph_create_local": (_int, [_vp, ctypes.POINTER(GraphPropertiesStruct), _vp, _vp,
                                              _vp, _i64, _i64, _i64, _vpp, _vpp]),
    "cugraph_amd_graph_create_local_compact": (_int, [_vp, ctypes.POINTER(GraphPropertiesStruct), _vp, _vp,
                                                        _vp, _i64, _i64, _i64, _i64, _vp, _vpp, _vpp]),
    "cugraph_amd_edgelist_source_weight_sums": (_int, [_vp, _vp, _vp, _vp, _vpp]),
    "cugraph_amd_edgelist_source_weight_sums_noalloc": (_int, [_vp, _vp, _vp, _vp, _vp, _vpp]),
    "cugraph_amd_partition_degree_order": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vpp]),
    "cugraph_amd_partition_group_edges": (_int, [_vp, _vp, _vp, _vp, ctypes.POINTER(_i64), _sz, _vp, _vp, _vp,
                                                 ctypes.POINTER(_i64), _vpp]),
    "cugraph_amd_graph_info": (_int, [_vp, ctypes.POINTER(_i64), _vpp]),
    "cugraph_amd_graph_get_compressed": (_int, [_vp, _vpp, _vpp, _vpp, _vpp, _vpp]),
    "cugraph_amd_graph_source_weight_sums": (_int, [_vp, _vp, _vp, _vpp]),
    "cugraph_amd_generate_rmat_edgelist": (_int, [_vp, _sz, _u64, _u64, _dbl, _dbl, _dbl, _int, _vp,
                                                  _vp, _vpp]),
    "cugraph_amd_generate_rmat_weights": (_int, [_vp, _u64, _u64, _vp, _vpp]),
    "cugraph_amd_pagerank_session_create": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl,
                                                   _int, _vpp, _vpp]),
    "cugraph_amd_pagerank_session_iterate": (_int, [_vp, _sz, _vpp]),
    "cugraph_amd_pagerank_session_iterate_phase": (_int, [_vp, _int, _vpp]),
    "cugraph_amd_graph_set_hot_slots": (_int, [_vp, ctypes.c_int64, _vpp]),
    "cugraph_amd_xs_window_slots": (ctypes.c_int64, [ctypes.c_int64]),
    "cugraph_amd_pagerank_session_timed_iterate": (_int, [_vp, _sz, ctypes.POINTER(ctypes.c_float),
                                                          ctypes.POINTER(ctypes.c_float), _vpp]),
    "cugraph_amd_pagerank_session_exchange_buffers": (_int, [_vp, _vpp, _vpp, _vpp]),
    "cugraph_amd_pagerank_session_status": (_int, [_vp, ctypes.POINTER(_sz), ctypes.POINTER(_int),
                                                   ctypes.POINTER(_dbl), ctypes.POINTER(_dbl), _vpp]),
    "cugraph_amd_pagerank_session_result": (_int, [_vp, _sz, _vpp, _vpp]),
    "cugraph_amd_pagerank_session_info": (_int, [_vp, ctypes.POINTER(_i64), _vpp]),
    "cugraph_amd_pagerank_session_free": (None, [_vp]),
    "cugraph_amd_version": (ctypes.c_char_p, []),
}

_lib = None


def load():
    """Load the HIP library; raises (no CPU fallback exists) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C cugraph_amd/csrc` (there is no CPU fallback)")
        # PyTorch ships its own HIP runtime and is this package's device-memory plumbing: import
        # it first, so that the library binds to the runtime torch has already loaded.  (Loaded
        # the other way round, the process ends up with /opt/rocm's runtime initialised and
        # torch's own initialisation then finds "No HIP GPUs".)
        import torch  # noqa: F401

        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def check(code, err_ptr, api_name):
    """Error-code -> exception mapping of pylibcugraph's assert_success
    (reference: python/pylibcugraph/pylibcugraph/utils.pyx:37-83)."""
    if code == CUGRAPH_SUCCESS:
        return
    lib = load()
    msg = ""
    if err_ptr is not None and err_ptr.value:
        raw = lib.cugraph_error_message(err_ptr)
        msg = raw.decode() if isinstance(raw, bytes) else str(raw)
        lib.cugraph_error_free(err_ptr)
    names = {
        CUGRAPH_UNKNOWN_ERROR: ("CUGRAPH_UNKNOWN_ERROR", RuntimeError),
        CUGRAPH_INVALID_HANDLE: ("CUGRAPH_INVALID_HANDLE", ValueError),
        CUGRAPH_ALLOC_ERROR: ("CUGRAPH_ALLOC_ERROR", MemoryError),
        CUGRAPH_INVALID_INPUT: ("CUGRAPH_INVALID_INPUT", ValueError),
        CUGRAPH_NOT_IMPLEMENTED: ("CUGRAPH_NOT_IMPLEMENTED", NotImplementedError),
        CUGRAPH_UNSUPPORTED_TYPE_COMBINATION: ("CUGRAPH_UNSUPPORTED_TYPE_COMBINATION", ValueError),
    }
    code_str, exc = names.get(code, ("unknown error code", RuntimeError))
    raise exc(f"non-success value returned from {api_name}: {code_str} {msg}")
