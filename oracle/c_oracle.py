"""ctypes wrapper over oracle/pagerank_oracle.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_SRC = os.path.join(_HERE, "pagerank_oracle.c")
_lib = None


def build(force=False):
    """gcc -O3 -fopenmp the C restatement into oracle/liboracle.so (git-ignored, travels to the GPU box)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(
            ["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", "-std=c11", "-o", _SO, _SRC, "-lm"]
        )
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        i64, i32, u64, u32 = ctypes.c_int64, ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint32
        p = ctypes.c_void_p
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_rmat.argtypes = [ctypes.c_int, u64, u64, u64, u32, u32, u32, ctypes.c_int, p, p]
        L.oracle_rmat.restype = None
        L.oracle_rmat_weights.argtypes = [u64, u64, u64, p]
        L.oracle_rmat_weights.restype = None
        L.oracle_compress.argtypes = [i64, i64, p, p, p, p, p]
        L.oracle_compress.restype = ctypes.c_int
        for name in ("oracle_pagerank_f32", "oracle_pagerank_f64"):
            f = getattr(L, name)
            f.argtypes = [i64, p, p, p, p, i64, p, p, p, ctypes.c_int, ctypes.c_double,
                          ctypes.c_double, i64, p]
            f.restype = ctypes.c_int
        L.oracle_philox.argtypes = [u32] * 6 + [p]
        L.oracle_philox.restype = None
        _lib = L
    return _lib


def num_threads():
    return int(lib().oracle_num_threads())


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def rmat_thresholds(a=0.57, b=0.19, c=0.19):
    """24-bit integer thresholds shared by the CPU and HIP generators."""
    ab = a + b
    t_ab = int(np.floor(ab * (1 << 24)))
    t_a = int(np.floor((a / ab if ab > 0 else 0.0) * (1 << 24)))
    t_c = int(np.floor((c / (1.0 - ab) if (1.0 - ab) > 0 else 0.0) * (1 << 24)))
    return t_ab, t_a, t_c


def rmat_edges(scale, num_edges, seed=42, a=0.57, b=0.19, c=0.19, scramble=False, first=0):
    src = np.empty(num_edges, dtype=np.int32)
    dst = np.empty(num_edges, dtype=np.int32)
    t_ab, t_a, t_c = rmat_thresholds(a, b, c)
    lib().oracle_rmat(scale, first, num_edges, seed, t_ab, t_a, t_c, int(bool(scramble)),
                      _ptr(src), _ptr(dst))
    return src, dst


def rmat_weights(num_edges, seed=43, first=0):
    w = np.empty(num_edges, dtype=np.float64)
    lib().oracle_rmat_weights(first, num_edges, seed, _ptr(w))
    return w


def compress(num_vertices, major, minor):
    """Stable (major, minor)-lexicographic compression.  Returns offsets[int64], minors[int32], perm[int64]."""
    major = np.ascontiguousarray(major, dtype=np.int32)
    minor = np.ascontiguousarray(minor, dtype=np.int32)
    ne = major.shape[0]
    offsets = np.empty(num_vertices + 1, dtype=np.int64)
    minors = np.empty(ne, dtype=np.int32)
    perm = np.empty(ne, dtype=np.int64)
    rc = lib().oracle_compress(num_vertices, ne, _ptr(major), _ptr(minor), _ptr(offsets),
                               _ptr(minors), _ptr(perm))
    if rc != 0:
        raise ValueError(f"oracle_compress failed rc={rc}")
    return offsets, minors, perm


def pagerank(num_vertices, offsets, indices, weights=None, *, dtype=np.float64, alpha=0.85,
             epsilon=1e-6, max_iterations=500, nstart=None, out_weight_sums=None,
             pers_vertices=None, pers_values=None):
    """C-API-semantics PageRank on a CSC (offsets over dst, indices = src).

    Returns (pageranks, iterations, converged)."""
    dtype = np.dtype(dtype)
    fn = lib().oracle_pagerank_f32 if dtype == np.float32 else lib().oracle_pagerank_f64
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=dtype)
    ow = None if out_weight_sums is None else np.ascontiguousarray(out_weight_sums, dtype=dtype)
    if nstart is None:
        pr = np.zeros(num_vertices, dtype=dtype)
    else:
        pr = np.array(nstart, dtype=dtype, copy=True)
        assert pr.shape[0] == num_vertices
    n_pers = 0
    pv = pvals = None
    if pers_vertices is not None:
        pv = np.ascontiguousarray(pers_vertices, dtype=np.int32)
        pvals = np.ascontiguousarray(pers_values, dtype=dtype)
        n_pers = pv.shape[0]
    info = np.zeros(2, dtype=np.int64)
    rc = fn(num_vertices, _ptr(offsets), _ptr(indices), _ptr(w), _ptr(ow), n_pers, _ptr(pv),
            _ptr(pvals), _ptr(pr), int(nstart is not None), float(alpha), float(epsilon),
            int(max_iterations), _ptr(info))
    if rc != 0:
        raise ValueError(f"oracle_pagerank failed rc={rc}")
    return pr, int(info[0]), bool(info[1])
