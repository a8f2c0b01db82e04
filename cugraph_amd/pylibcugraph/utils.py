"""Array plumbing between Python device arrays and type-erased views
(reference: python/pylibcugraph/pylibcugraph/utils.pyx:86-237)."""
import ctypes

import numpy as np

from .. import _capi

_NUMPY_TO_C = {
    np.dtype(np.int32): _capi.INT32,
    np.dtype(np.int64): _capi.INT64,
    np.dtype(np.float32): _capi.FLOAT32,
    np.dtype(np.float64): _capi.FLOAT64,
}


def assert_CAI_type(obj, var_name, allow_None=False):
    if allow_None:
        if obj is None:
            return
        msg = f"{var_name} must be None or support __cuda_array_interface__"
    else:
        msg = f"{var_name} does not support __cuda_array_interface__"
    if not hasattr(obj, "__cuda_array_interface__"):
        raise TypeError(msg)


def _numpy_dtype_of(obj):
    dt = obj.dtype
    try:
        return np.dtype(dt)
    except TypeError:
        # torch.dtype: go through its name ("torch.float32" -> "float32")
        return np.dtype(str(dt).split(".")[-1])


def get_c_type_from_numpy_type(numpy_type):
    try:
        return _NUMPY_TO_C[np.dtype(numpy_type)]
    except KeyError:
        raise RuntimeError("Internal error: got invalid data type enum value "
                           f"from Numpy: {numpy_type}")


def get_numpy_type_from_c_type(c_type):
    for k, v in _NUMPY_TO_C.items():
        if v == c_type:
            return k
    if c_type == _capi.SIZE_T:
        return np.dtype(np.int64)
    raise RuntimeError(f"Internal error: got invalid data type enum value from C: {c_type}")


def device_pointer(obj):
    # torch tensors refuse __cuda_array_interface__ when they require grad; data_ptr() is the same
    # address and never synchronises
    if hasattr(obj, "data_ptr") and getattr(obj, "is_cuda", False):
        return int(obj.data_ptr())
    return int(obj.__cuda_array_interface__["data"][0])


def create_view_from_py_obj(python_obj):
    """utils.pyx:226-237: NULL for None, else a new non-owning view the caller frees."""
    if python_obj is None:
        return None
    lib = _capi.load()
    n = len(python_obj)
    ptr = device_pointer(python_obj) if n > 0 else 0
    return lib.cugraph_type_erased_device_array_view_create(
        ctypes.c_void_p(ptr), n, get_c_type_from_numpy_type(_numpy_dtype_of(python_obj)))


def free_view(view_ptr):
    if view_ptr:
        _capi.load().cugraph_type_erased_device_array_view_free(view_ptr)


def copy_to_device_array(c_resource_handle_ptr, device_array_view_ptr):
    """utils.pyx:153-187 (copy_to_cupy_array): allocate a device array of the view's type and
    size, copy into it through the C ABI, free the view.  Returns a torch tensor."""
    import torch

    lib = _capi.load()
    c_type = lib.cugraph_type_erased_device_array_view_type(device_array_view_ptr)
    size = lib.cugraph_type_erased_device_array_view_size(device_array_view_ptr)
    np_dtype = get_numpy_type_from_c_type(c_type)
    out = torch.zeros(size, dtype=getattr(torch, np_dtype.name), device="cuda")
    out_view = lib.cugraph_type_erased_device_array_view_create(
        ctypes.c_void_p(out.data_ptr() if size else 0), size, c_type)
    err = ctypes.c_void_p()
    code = lib.cugraph_type_erased_device_array_view_copy(c_resource_handle_ptr, out_view,
                                                          device_array_view_ptr, ctypes.byref(err))
    lib.cugraph_type_erased_device_array_view_free(out_view)
    _capi.check(code, err, "cugraph_type_erased_device_array_view_copy")
    lib.cugraph_type_erased_device_array_view_free(device_array_view_ptr)
    return out
