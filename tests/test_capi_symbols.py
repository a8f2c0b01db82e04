"""CPU: the C-ABI library loads and exports exactly what include/cugraph_amd_c.h declares, and the
host-side mirror has the reference's signatures.  No compute calls (no GPU here)."""
import inspect
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cugraph_amd_c.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cugraph_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from cugraph_amd import _capi

    lib = _capi.load()
    declared = _declared_functions()
    assert len(declared) > 50
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert sorted(_capi.SIGNATURES) == declared, "ctypes table and header disagree"
    out = subprocess.check_output(["nm", "-D", "--defined-only", _capi.LIB_PATH], text=True)
    exported = set(re.findall(r" T (cugraph_[a-z0-9_]+)", out))
    assert exported == set(declared)


def test_header_compiles_as_c():
    subprocess.check_call(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", HEADER])


def test_no_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cugraph_amd import pylibcugraph as plc

    with pytest.raises(RuntimeError):
        plc.ResourceHandle()


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cugraph_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f


def test_mirror_signatures_match_pylibcugraph():
    """Argument names and order of python/pylibcugraph/pylibcugraph/pagerank.pyx:54-64,
    personalized_pagerank.pyx:54-66 and graphs.pyx:144-158."""
    from cugraph_amd import pylibcugraph as plc

    assert list(inspect.signature(plc.pagerank).parameters) == [
        "resource_handle", "graph", "precomputed_vertex_out_weight_vertices",
        "precomputed_vertex_out_weight_sums", "initial_guess_vertices", "initial_guess_values",
        "alpha", "epsilon", "max_iterations", "do_expensive_check", "fail_on_nonconvergence"]
    assert list(inspect.signature(plc.personalized_pagerank).parameters) == [
        "resource_handle", "graph", "precomputed_vertex_out_weight_vertices",
        "precomputed_vertex_out_weight_sums", "initial_guess_vertices", "initial_guess_values",
        "personalization_vertices", "personalization_values", "alpha", "epsilon", "max_iterations",
        "do_expensive_check", "fail_on_nonconvergence"]
    assert list(inspect.signature(plc.SGGraph.__init__).parameters)[1:] == [
        "resource_handle", "graph_properties", "src_or_offset_array", "dst_or_index_array",
        "weight_array", "store_transposed", "renumber", "do_expensive_check", "edge_id_array",
        "edge_type_array", "input_array_format", "vertices_array", "drop_self_loops",
        "drop_multi_edges"]
    assert plc.pagerank.__defaults__ == (True,)
    assert issubclass(plc.FailedToConvergeError, Exception)
    p = plc.GraphProperties(is_symmetric=True)
    assert p.is_symmetric and not p.is_multigraph


def test_error_code_mapping():
    # python/pylibcugraph/pylibcugraph/utils.pyx:37-83
    from cugraph_amd import _capi

    for code, exc in ((1, RuntimeError), (2, ValueError), (3, MemoryError), (4, ValueError),
                      (5, NotImplementedError), (6, ValueError), (99, RuntimeError)):
        with pytest.raises(exc, match="non-success value returned from api_x"):
            _capi.check(code, None, "api_x")
    _capi.check(0, None, "api_x")
