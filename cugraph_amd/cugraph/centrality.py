"""cugraph.katz_centrality / eigenvector_centrality / hits at the DataFrame level (reference:
python/cugraph/cugraph/centrality/katz_centrality.py:30-205, centrality/eigenvector_centrality.py:27-140,
link_analysis/hits.py:28-140): same arguments, defaults, argument checks, warnings and result
columns; frames are pandas DataFrames."""
import warnings

import numpy as np

from .. import pylibcugraph as plc
from .graph import Graph
from .link_analysis import _dev, _graph_from_nx, _vertex_ids


def _as_graph(G, name):
    isNx = False
    if not isinstance(G, Graph):
        try:
            import networkx as nx
        except ImportError:  # pragma: no cover
            nx = None
        if nx is not None and isinstance(G, nx.Graph):
            G, isNx = _graph_from_nx(G, "weight", True), True
        else:
            raise TypeError(f"input must be either a cuGraph or NetworkX graph type, got {type(G)}")
    if G.store_transposed is False:
        warnings.warn(f"{name} expects the 'store_transposed' flag to be set to 'True' for optimal "
                      "performance during the graph creation", UserWarning)
    return G, isNx


def _frame(G, vertices, **columns):
    import pandas as pd

    ids = vertices.cpu().numpy()
    df = pd.DataFrame({"vertex": ids}) if G.renumber_map is None else G.renumber_map.from_internal(ids, "vertex")
    for name, values in columns.items():
        df[name] = values.cpu().numpy()
    return df


def _score_dict(df, column):
    return dict(zip(df["vertex"].tolist(), df[column].tolist()))


def katz_centrality(G, alpha=None, beta=1.0, max_iter=100, tol=1.0e-6, nstart=None, normalized=True):
    """alpha defaults to 1 / (largest degree) (katz_centrality.py:140-142); `normalized` is accepted
    and, as in the reference, not looked at.  Returns `vertex`, `katz_centrality` (a dict for
    NetworkX input)."""
    G, isNx = _as_graph(G, "Katz centrality")
    if alpha is None:
        alpha = 1 / G.degree()["degree"].max()
    if alpha <= 0.0:
        raise ValueError(f"'alpha' must be a positive float or None, got: {alpha}")
    if (not isinstance(beta, float)) or (beta <= 0.0):
        raise ValueError(f"'beta' must be a positive float or None, got: {beta}")
    if (not isinstance(max_iter, int)) or (max_iter <= 0):
        raise ValueError(f"'max_iter' must be a positive integer, got: {max_iter}")
    if (not isinstance(tol, float)) or (tol <= 0.0):
        raise ValueError(f"'tol' must be a positive float, got: {tol}")
    if nstart is not None:
        # the reference hands the first column of the renumbered frame over as `betas`
        # (katz_centrality.py:158-166), i.e. vertex ids as values; nothing sensible to mirror
        raise NotImplementedError("nstart is not supported for Katz centrality")
    vertices, values = plc.katz_centrality(G._resource_handle, G._plc_graph, None, float(alpha), beta, tol,
                                           max_iter, False)
    df = _frame(G, vertices, katz_centrality=values)
    return _score_dict(df, "katz_centrality") if isNx else df


def eigenvector_centrality(G, max_iter=100, tol=1.0e-6):
    """Returns `vertex`, `eigenvector_centrality` (a dict for NetworkX input)."""
    if (not isinstance(max_iter, int)) or max_iter <= 0:
        raise ValueError(f"'max_iter' must be a positive integer, got: {max_iter}")
    if (not isinstance(tol, float)) or (tol <= 0.0):
        raise ValueError(f"'tol' must be a positive float, got: {tol}")
    G, isNx = _as_graph(G, "Eigenvector centrality")
    vertices, values = plc.eigenvector_centrality(G._resource_handle, G._plc_graph, tol, max_iter, False)
    df = _frame(G, vertices, eigenvector_centrality=values)
    return _score_dict(df, "eigenvector_centrality") if isNx else df


def hits(G, max_iter=100, tol=1.0e-5, nstart=None, normalized=True):
    """nstart: optional frame `vertex`, `values` with the initial hub scores.  Returns `vertex`,
    `hubs`, `authorities` (a pair of dicts for NetworkX input)."""
    G, isNx = _as_graph(G, "HITS")
    guess_v = guess_x = None
    if nstart is not None:
        values = nstart["values"]
        want = np.float32 if G.edgelist.weights is False else G.edgelist.edgelist_df["weights"].dtype
        if values.dtype != want:
            values = values.astype(want)
        guess_v, guess_x = _vertex_ids(G, nstart), _dev(values)
        if G.renumber_map is None and guess_v.dtype != _dev(G.edgelist.edgelist_df["src"].iloc[:0]).dtype:
            guess_v = guess_v.to(_dev(G.edgelist.edgelist_df["src"].iloc[:0]).dtype)
    vertices, hubs, authorities = plc.hits(G._resource_handle, G._plc_graph, tol, max_iter, guess_v, guess_x,
                                           normalized, False)
    df = _frame(G, vertices, hubs=hubs, authorities=authorities)
    if isNx:
        return _score_dict(df, "hubs"), _score_dict(df, "authorities")
    return df
