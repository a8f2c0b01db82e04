"""cugraph.pagerank (reference: python/cugraph/cugraph/link_analysis/pagerank.py:33-308) and the
NetworkX-backend entry (reference: python/nx-cugraph/nx_cugraph/algorithms/link_analysis/
pagerank_alg.py:29-112).  Frames are pandas DataFrames (the ROCm stand-in for cuDF)."""
import warnings

import numpy as np

from .. import pylibcugraph as plc
from ..pylibcugraph import exceptions as plc_exceptions
from .exceptions import FailedToConvergeError
from .graph import Graph


# The two warning texts are contract: the reference's tests assert them word for word
# (python/cugraph/cugraph/tests/link_analysis/test_pagerank.py:391-410).  {0}: frame name, {1}: the
# graph's dtype, {2}: the frame's.
_DTYPE_WARNINGS = {
    "values": ("PageRank requires '{0}' values to match the graph's 'edge_attr' type. "
               "edge_attr type is: {1} and got '{0}' values of type: {2}."),
    "vertex": ("PageRank requires '{0}' vertex to match the graph's 'vertex' type. "
               "input graph's vertex type is: {1} and got '{0}' vertex of type: {2}."),
}


def _graph_value_dtype(G):
    """The type PageRank computes in: the weights' type, float32 for an unweighted graph (graph_sg.cpp:247)."""
    if G.edgelist.weights is False:
        return np.dtype(np.float32)
    return G.edgelist.edgelist_df["weights"].dtype


def _match_graph_dtypes(G, frame, name):
    """A (vertex, values) frame with its columns cast to what the C layer insists on -- value
    arrays of the graph's weight type, vertex arrays of its vertex type (pagerank.cpp:268-299) --
    warning once per column that had to be cast (the behaviour of pagerank.py:45-80).  Vertices of
    a graph with a host-side number map are translated later (_vertex_ids): nothing to match."""
    wanted = {"values": _graph_value_dtype(G)}
    if G.renumber_map is None:
        wanted["vertex"] = G.edgelist.edgelist_df.dtypes.iloc[0]
    casts = {}
    for column, dtype in wanted.items():
        if column in frame.columns and frame[column].dtype != dtype:
            warnings.warn(_DTYPE_WARNINGS[column].format(name, dtype, frame[column].dtype), UserWarning)
            casts[column] = dtype
    return frame.astype(casts) if casts else frame


def _graph_from_nx(nxG, weight, store_transposed):
    """utilities/nx_factory.py:95-160: DiGraph -> directed Graph, Graph -> undirected; integer
    node ids int32, named weight float32"""
    import networkx as nx
    import pandas as pd

    if isinstance(nxG, (nx.MultiGraph, nx.MultiDiGraph)) or not isinstance(nxG, nx.Graph):
        raise TypeError(f"nxG must be either a NetworkX Graph or DiGraph, got {type(nxG)}")
    G = Graph(directed=isinstance(nxG, nx.DiGraph))
    weighted = nx.is_weighted(nxG, weight=weight) if weight is not None else False
    if weighted:
        edges = list(nxG.edges(data=weight))
        pdf = pd.DataFrame({"src": [s for s, _, _ in edges], "dst": [d for _, d, _ in edges],
                            "weight": np.array([w for _, _, w in edges], dtype=np.float32)})
    else:
        edges = list(nxG.edges(data=False))
        pdf = pd.DataFrame({"src": [s for s, _ in edges], "dst": [d for _, d in edges]})
    if pdf["src"].dtype.kind in "iu":
        pdf = pdf.astype({"src": np.int32, "dst": np.int32})
    nodes = np.array(list(nxG.nodes()), dtype=pdf["src"].dtype) if len(pdf) else None
    G.from_pandas_edgelist(pdf, "src", "dst", "weight" if weighted else None, renumber=True,
                           store_transposed=store_transposed, vertices=nodes)
    return G


def _dev(series):
    import torch

    return torch.from_numpy(np.ascontiguousarray(series.to_numpy())).cuda()


def _vertex_ids(G, input_df):
    """device ids of a (vertex, value) frame's vertices.  Through the number map when the graph's
    vertices are strings or several columns: then every column but the last names the vertex
    (pagerank.py:33-40 renumber_vertices)."""
    import torch

    if G.renumber_map is None:
        return _dev(input_df["vertex"])
    cols = list(input_df.columns[:-1]) if len(G.renumber_map.col_names) > 1 else ["vertex"]
    return torch.from_numpy(G.renumber_map.to_internal(input_df[cols])).cuda()


def pagerank(G, alpha=0.85, personalization=None, precomputed_vertex_out_weight=None, max_iter=100,
             tol=1.0e-5, nstart=None, weight=None, dangling=None, fail_on_nonconvergence=True):
    """Same arguments, defaults, warnings, exceptions and return shape as cugraph.pagerank;
    `weight` only matters for NetworkX input, `dangling` is accepted and ignored
    (pagerank.py:162-168).  Returns a DataFrame with columns `vertex`, `pagerank` (a dict for
    NetworkX input), or `(that, converged)` when fail_on_nonconvergence is False."""
    import pandas as pd

    isNx = False
    if not isinstance(G, Graph):
        try:
            import networkx as nx
        except ImportError:  # pragma: no cover
            nx = None
        if nx is not None and isinstance(G, nx.Graph):
            G, isNx = _graph_from_nx(G, weight, True), True
        else:
            raise TypeError("input must be either a cuGraph or NetworkX graph "
                            f"type, got {type(G)}")
    if G.store_transposed is False:
        warning_msg = (
            "Pagerank expects the 'store_transposed' flag "
            "to be set to 'True' for optimal performance during "
            "the graph creation"
        )
        warnings.warn(warning_msg, UserWarning)

    initial_guess_vertices = initial_guess_values = None
    pre_vtx_o_wgt_vertices = pre_vtx_o_wgt_sums = None
    # the library takes external ids directly (number_map lives below the C ABI), so the
    # reference's renumber_vertices / unrenumber round trips (pagerank.py:33-40, 299-300) vanish
    if nstart is not None:
        nstart = _match_graph_dtypes(G, nstart, "nstart")
        initial_guess_vertices, initial_guess_values = _vertex_ids(G, nstart), _dev(nstart["values"])
    if precomputed_vertex_out_weight is not None:
        precomputed_vertex_out_weight = _match_graph_dtypes(G, precomputed_vertex_out_weight,
                                                            "precomputed_vertex_out_weight")
        sums = precomputed_vertex_out_weight["sums"].astype(_graph_value_dtype(G), copy=False)
        pre_vtx_o_wgt_vertices = _vertex_ids(G, precomputed_vertex_out_weight)
        pre_vtx_o_wgt_sums = _dev(sums)
    try:
        if personalization is not None:
            if not isinstance(personalization, pd.DataFrame):
                raise NotImplementedError(
                    "personalization other than a cudf dataframe currently not "
                    "supported"
                )
            personalization = _match_graph_dtypes(G, personalization, "personalization")
            result_tuple = plc.personalized_pagerank(
                resource_handle=G._resource_handle,
                graph=G._plc_graph,
                precomputed_vertex_out_weight_vertices=pre_vtx_o_wgt_vertices,
                precomputed_vertex_out_weight_sums=pre_vtx_o_wgt_sums,
                personalization_vertices=_vertex_ids(G, personalization),
                personalization_values=_dev(personalization["values"]),
                initial_guess_vertices=initial_guess_vertices,
                initial_guess_values=initial_guess_values,
                alpha=alpha,
                epsilon=tol,
                max_iterations=max_iter,
                do_expensive_check=False,
                fail_on_nonconvergence=fail_on_nonconvergence,
            )
        else:
            result_tuple = plc.pagerank(
                resource_handle=G._resource_handle,
                graph=G._plc_graph,
                precomputed_vertex_out_weight_vertices=pre_vtx_o_wgt_vertices,
                precomputed_vertex_out_weight_sums=pre_vtx_o_wgt_sums,
                initial_guess_vertices=initial_guess_vertices,
                initial_guess_values=initial_guess_values,
                alpha=alpha,
                epsilon=tol,
                max_iterations=max_iter,
                do_expensive_check=False,
                fail_on_nonconvergence=fail_on_nonconvergence,
            )
    # re-raised as this package's exception, as the reference does (pagerank.py:290-293)
    except plc_exceptions.FailedToConvergeError as exc:
        raise FailedToConvergeError from exc
    # PageRank may have transposed the graph in place (pagerank.cpp:105-110)
    df = pd.DataFrame({"vertex": result_tuple[0].cpu().numpy(),
                       "pagerank": result_tuple[1].cpu().numpy()})
    if G.renumber_map is not None:  # G.unrenumber(df, "vertex"), number_map.py:659-668 for the names
        ext = G.renumber_map.from_internal(df["vertex"].to_numpy(), "vertex")
        df = pd.concat([ext, df[["pagerank"]]], axis=1)
    out = dict(zip(df["vertex"].tolist(), df["pagerank"].tolist())) if isNx else df
    if fail_on_nonconvergence:
        return out
    return (out, bool(result_tuple[2]))


def nx_pagerank(G, alpha=0.85, personalization=None, max_iter=100, tol=1.0e-6, nstart=None,
                weight="weight", dangling=None, *, dtype=None):
    """What `nx.pagerank(G, backend="cugraph")` runs (pagerank_alg.py:35-96): NetworkX
    argument conventions (dicts, tol scaled by N, PowerIterationFailedConvergence)."""
    import networkx as nx
    import torch

    N = len(G)
    if N == 0:
        return {}
    nodes = list(G.nodes())
    index = {n: i for i, n in enumerate(nodes)}
    directed = G.is_directed()
    wname = weight if (weight is not None and nx.is_weighted(G, weight=weight)) else None
    if dtype is None:
        dtype = np.float32  # _to_graph(G, weight, 1, np.float32), pagerank_alg.py:48-51
    dtype = np.dtype(dtype).type
    tdtype = torch.float32 if dtype == np.float32 else torch.float64

    def nodearray(d):
        a = np.zeros(N, dtype=dtype)
        for k, v in d.items():
            a[index[k]] = v  # KeyError for unknown nodes, as NetworkX
        return a

    if nstart is not None:
        nstart = nodearray(nstart)
        if (total := nstart.sum()) == 0:
            raise ZeroDivisionError
        nstart /= total
    if personalization is not None:
        personalization = nodearray(personalization)
        if (total := personalization.sum()) == 0:
            raise ZeroDivisionError
        personalization /= total
    edges = list(G.edges(data=wname)) if wname else [(s, d, 1) for s, d in G.edges()]
    src = np.fromiter((index[s] for s, _, _ in edges), dtype=np.int32, count=len(edges))
    dst = np.fromiter((index[d] for _, d, _ in edges), dtype=np.int32, count=len(edges))
    wgt = np.fromiter((1 if w is None else w for _, _, w in edges), dtype=dtype, count=len(edges))
    if not directed:
        keep = src != dst  # an undirected self-loop is one edge, not two
        src, dst, wgt = (np.concatenate([src, dst[keep]]), np.concatenate([dst, src[keep]]),
                         np.concatenate([wgt, wgt[keep]]))
    if dangling is not None:
        dang = nodearray(dangling)
        if dang.sum() == 0:
            raise ZeroDivisionError
        if (np.bincount(src, minlength=N) == 0).any():
            raise NotImplementedError("custom dangling weights is not supported")
    if max_iter <= 0:
        raise nx.PowerIterationFailedConvergence(max_iter)
    h = plc.ResourceHandle()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    graph = plc.SGGraph(h, plc.GraphProperties(is_symmetric=not directed, is_multigraph=G.is_multigraph()),
                        dev(src), dev(dst), weight_array=dev(wgt), store_transposed=True,
                        renumber=False, vertices_array=torch.arange(N, dtype=torch.int32, device="cuda"))
    ids = torch.arange(N, dtype=torch.int32, device="cuda")
    kwargs = dict(resource_handle=h, graph=graph, precomputed_vertex_out_weight_vertices=None,
                  precomputed_vertex_out_weight_sums=None,
                  initial_guess_vertices=None if nstart is None else ids,
                  initial_guess_values=None if nstart is None else dev(nstart).to(tdtype),
                  alpha=alpha, epsilon=N * tol, max_iterations=max_iter, do_expensive_check=False,
                  fail_on_nonconvergence=False)
    if personalization is None:
        node_ids, values, is_converged = plc.pagerank(**kwargs)
    else:
        node_ids, values, is_converged = plc.personalized_pagerank(
            personalization_vertices=ids, personalization_values=dev(personalization).to(tdtype),
            **kwargs)
    if not is_converged:
        raise nx.PowerIterationFailedConvergence(max_iter)
    vals = values.cpu().numpy().tolist()
    return {nodes[i]: vals[k] for k, i in enumerate(node_ids.cpu().numpy().tolist())}
