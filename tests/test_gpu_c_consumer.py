"""The boundary from the language the reference's FFI binds: tests/c_consumer/pagerank_consumer.c, a plain C
program compiled with gcc against include/cugraph_amd_c.h and linked to libcugraph_amd_c.so (built by
__graft_entry__.build()), run on every golden case of the reference's own C boundary test
(cpp/tests/c_api/pagerank_test.c:396-555) and checked with that test's tolerance (nearlyEqual, 0.001)."""
import json
import os
import subprocess

import pytest

from helpers import case_arrays

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
BINARY = os.path.join(HERE, "c_consumer", "pagerank_consumer")

with open(os.path.join(HERE, "golden", "reference_vectors.json")) as _f:
    CASES = json.load(_f)["c_api"]["cases"]


@pytest.mark.parametrize("entry", ["sg", "mg"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_c_program_reproduces_the_reference_goldens(name, entry, tmp_path):
    """entry "mg": the same cases through cugraph_mg_graph_create and an MG handle of one rank, the way
    cpp/tests/c_api/mg_pagerank_test.c runs the same graphs (its goldens are pagerank_test.c's)."""
    assert os.path.exists(BINARY), "tests/c_consumer/pagerank_consumer is built by __graft_entry__.build()"
    c = CASES[name]
    src, dst, wgt = case_arrays(c)
    pers = list(zip(c.get("pers_vertices", []), c.get("pers_values", [])))
    lines = [f"{c['num_vertices']} {len(src)} {int(c['store_transposed'])} {c['alpha']!r} {c['epsilon']!r} "
             f"{c['max_iterations']} {int(c['allow_nonconvergence'])} {len(pers)}"]
    lines += [f"{s} {d} {float(w)!r}" for s, d, w in zip(src, dst, wgt)]
    lines += [f"{v} {x!r}" for v, x in pers]
    case_file = tmp_path / "case.txt"
    case_file.write_text("\n".join(lines) + "\n")
    r = subprocess.run([BINARY, str(case_file)] + (["mg"] if entry == "mg" else []), capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr
    out = r.stdout.split("\n")
    got = {}
    for line in out:
        parts = line.split()
        if len(parts) == 2 and parts[0] != "converged":
            got[int(parts[0])] = float(parts[1])
    want = c["expected"]
    assert sorted(got) == list(range(len(want)))
    for v, x in enumerate(want):
        assert abs(got[v] - x) <= 1e-3 * max(abs(got[v]), abs(x)), (name, v, got[v], x)
    # the non-convergence cases run out of iterations on purpose (pagerank_test.c:474-491, 525-555)
    assert f"converged {0 if c['allow_nonconvergence'] else 1}" in out
