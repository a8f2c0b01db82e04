"""cugraph.Graph for the PageRank path (reference: python/cugraph/cugraph/structure/graph_classes.py:
107-170 from_cudf_edgelist, :360-430 from_pandas_edgelist; graph_implementation/simpleGraph.py:
130-330 edge-list intake, :1120-1200 _make_plc_graph; structure/symmetrize.py:24-120).

Only what `cugraph.pagerank` touches is here: edge-list intake with optional weights, the
undirected => symmetrised convention, renumber / store_transposed, and the pylibcugraph graph."""
import numpy as np


class _EdgeList:
    """what pagerank.py reads: `.weights` (bool) and `.edgelist_df` (src, dst[, weights])"""

    def __init__(self, df, weighted):
        self.edgelist_df = df
        self.weights = weighted


class _NumberMap:
    """External vertices that are not one integer column -- strings, or several columns per vertex
    -- against dense integer ids (reference: structure/number_map.py:28-110 NumberMap.SingleGPU:
    to_internal_vertex_id / from_internal_vertex_id; :603-722 unrenumber and its column naming).
    Host-side, with pandas: the distinct vertices in sorted order are the ids."""

    def __init__(self, frames):
        import pandas as pd

        k = frames[0].shape[1]
        self.col_names = [str(i) for i in range(k)]
        parts = [f.set_axis(self.col_names, axis=1) for f in frames]
        self.table = (pd.concat(parts, ignore_index=True).drop_duplicates()
                      .sort_values(self.col_names, kind="stable").reset_index(drop=True))
        self._index = self._key(self.table)
        if len(self.table) >= 2**31 - 1:
            raise NotImplementedError("2^31 or more vertices are not supported")

    def _key(self, frame):
        import pandas as pd

        if frame.shape[1] != len(self.col_names):
            raise ValueError(f"a vertex of this graph has {len(self.col_names)} column(s), got {frame.shape[1]}")
        if len(self.col_names) == 1:
            return pd.Index(frame.iloc[:, 0])
        return pd.MultiIndex.from_frame(frame.set_axis(self.col_names, axis=1))

    def to_internal(self, frame):
        ids = self._index.get_indexer(self._key(frame))
        if (ids < 0).any():
            raise ValueError("vertex not found in the graph")
        return ids.astype(np.int32)

    def from_internal(self, ids, column_name):
        """frame of the external vertices of `ids`: `column_name`, or `{i}_{column_name}` per column"""
        out = self.table.iloc[np.asarray(ids, dtype=np.int64)].reset_index(drop=True)
        if len(self.col_names) == 1:
            return out.set_axis([column_name], axis=1)
        return out.set_axis([f"{c}_{column_name}" for c in self.col_names], axis=1)


class Graph:
    def __init__(self, m_graph=None, directed=False):
        if m_graph is not None:
            raise NotImplementedError("construction from another graph is outside the PageRank path")
        self.graph_properties_directed = bool(directed)
        self.multi = False
        self.edgelist = None
        self.store_transposed = False
        self.renumbered = False
        self._plc_graph = None
        self._resource_handle = None
        self._num_vertices = 0
        self.renumber_map = None  # _NumberMap: string or multi-column vertices

    # ------------------------------------------------------------------ queries
    def is_directed(self):
        return self.graph_properties_directed

    def is_multigraph(self):
        return self.multi

    def is_weighted(self):
        return bool(self.edgelist is not None and self.edgelist.weights)

    def number_of_vertices(self):
        return self._num_vertices

    number_of_nodes = number_of_vertices

    def number_of_edges(self, directed_edges=False):
        n = len(self.edgelist.edgelist_df)
        if self.is_directed() or directed_edges:
            return n
        df = self.edgelist.edgelist_df
        loops = int((df["src"] == df["dst"]).sum())
        return (n - loops) // 2 + loops

    def _external(self, ids, column_name):
        """frame of the caller's vertex names for internal ids (one or several columns)"""
        import pandas as pd

        if self.renumber_map is None:
            return pd.DataFrame({column_name: np.asarray(ids)})
        return self.renumber_map.from_internal(ids, column_name)

    def view_edge_list(self):
        """simpleGraph.py:630-700: the edge list in the caller's vertex names -- `src`, `dst`
        (`0_src`, ... for multi-column vertices) and `weights`; one row per undirected edge"""
        import pandas as pd

        df = self.edgelist.edgelist_df
        if not self.is_directed():
            df = df[df["src"] <= df["dst"]].reset_index(drop=True)
        out = pd.concat([self._external(df["src"].to_numpy(), "src"), self._external(df["dst"].to_numpy(), "dst")],
                        axis=1)
        if self.edgelist.weights:
            out["weights"] = df["weights"].to_numpy()
        return out

    def nodes(self):
        """simpleGraph.py:1020-1050: the vertices that occur in the edge list (plus the vertex list, if given)"""
        df = self.edgelist.edgelist_df
        ids = np.union1d(df["src"].to_numpy(), df["dst"].to_numpy())
        if self._vertices is not None:
            ids = np.union1d(ids, self._vertices)
        out = self._external(ids, "vertex")
        return out["vertex"] if out.shape[1] == 1 else out

    def _degrees(self):
        df = self.edgelist.edgelist_df
        ids = self.nodes() if self.renumber_map is None else None
        verts = np.asarray(ids) if ids is not None else np.arange(len(self.renumber_map.table))
        pos = {int(v): i for i, v in enumerate(verts.tolist())}
        ind, outd = np.zeros(len(verts), np.int64), np.zeros(len(verts), np.int64)
        np.add.at(outd, [pos[int(v)] for v in df["src"].tolist()], 1)
        np.add.at(ind, [pos[int(v)] for v in df["dst"].tolist()], 1)
        return verts, ind, outd

    def in_degree(self):
        """frame `vertex`, `degree` (simpleGraph.py:780-830); an undirected edge counts once per end"""
        verts, ind, _ = self._degrees()
        out = self._external(verts, "vertex")
        out["degree"] = ind
        return out

    def out_degree(self):
        verts, _, outd = self._degrees()
        out = self._external(verts, "vertex")
        out["degree"] = outd
        return out

    def degree(self):
        """in + out for a directed graph, the number of incident edges for an undirected one"""
        verts, ind, outd = self._degrees()
        out = self._external(verts, "vertex")
        out["degree"] = ind + outd if self.is_directed() else outd
        return out

    def has_isolated_vertices(self):
        """graph_classes.py:669-686: vertices (of the vertex list) that no edge touches"""
        _, ind, outd = self._degrees()
        return bool(np.any((ind + outd) == 0))

    # ------------------------------------------------------------------ intake
    def from_pandas_edgelist(self, pdf, source="source", destination="destination", edge_attr=None,
                             weight=None, renumber=True, store_transposed=False,
                             legacy_renum_only=False, vertices=None, _as_given=False):
        """_as_given (internal, from_cudf_adjlist): the rows are the stored edges, both directions of
        an undirected edge included; nothing is symmetrised or dropped"""
        import pandas as pd

        if not isinstance(pdf, pd.DataFrame):
            raise TypeError("pdf input is not a Pandas DataFrame")
        if self.edgelist is not None:
            raise RuntimeError("Graph already has values")
        s_cols = list(source) if isinstance(source, (list, tuple)) else [source]
        d_cols = list(destination) if isinstance(destination, (list, tuple)) else [destination]
        if len(s_cols) != len(d_cols) or not s_cols:
            raise ValueError("source and destination must name the same number of columns")
        for col in s_cols + d_cols:
            if col not in pdf.columns:
                raise ValueError("source column names and/or destination column names not found in "
                                 "input. Recheck the source and destination parameters")
        wcol = edge_attr if edge_attr is not None else weight
        if isinstance(wcol, (list, tuple)):
            if len(wcol) != 1:
                raise NotImplementedError("edge ids / edge types are outside the PageRank path")
            wcol = wcol[0]
        if wcol is not None and wcol not in pdf.columns:
            raise ValueError("edge_attr column name not found in input. Recheck the edge_attr parameter")
        if len(s_cols) > 1 or pdf[s_cols[0]].dtype.kind not in "iu" or pdf[d_cols[0]].dtype.kind not in "iu":
            # number_map.py: vertices named by strings or by several columns get integer ids here;
            # results and (vertex, value) frames are translated in cugraph.pagerank
            frames = [pdf[s_cols], pdf[d_cols]]
            if vertices is not None:
                vf = vertices if isinstance(vertices, pd.DataFrame) else pd.DataFrame({"0": np.asarray(vertices)})
                frames.append(vf)
                vertices = None
            self.renumber_map = _NumberMap(frames)
            df = pd.DataFrame({"src": self.renumber_map.to_internal(pdf[s_cols]),
                               "dst": self.renumber_map.to_internal(pdf[d_cols])})
            vertices = np.arange(len(self.renumber_map.table), dtype=np.int32)
        else:
            df = pd.DataFrame({"src": pdf[s_cols[0]].to_numpy(), "dst": pdf[d_cols[0]].to_numpy()})
        if df["src"].dtype != df["dst"].dtype:
            raise ValueError("source and destination columns must have the same dtype")
        if df["src"].dtype not in (np.int32, np.int64):
            df = df.astype({"src": np.int64, "dst": np.int64})
        if wcol is not None:
            w = pdf[wcol].to_numpy()
            if w.dtype not in (np.float32, np.float64):
                w = w.astype(np.float64)
            df["weights"] = w
        if _as_given:
            pass
        elif not self.is_directed():
            df = _symmetrize(df, self.multi)
        elif not self.multi:
            # simpleGraph.py:306-312: a Graph (not MultiGraph) drops parallel edges
            df = _drop_multi(df)
        self.edgelist = _EdgeList(df, wcol is not None)
        self.store_transposed = bool(store_transposed)
        self.renumbered = bool(renumber)
        self._vertices = None if vertices is None else np.asarray(vertices, dtype=df["src"].dtype)
        self._make_plc_graph()
        return self

    # cuDF is CUDA-only; the method name is kept so reference call sites read the same
    from_cudf_edgelist = from_pandas_edgelist

    def from_cudf_adjlist(self, offset_col, index_col, value_col=None, renumber=True, store_transposed=False):
        """graph_classes.py:206-271: a graph from its adjacency list (CSR: offsets over sources of
        size V + 1, destination indices of size E, optional weights).  Undirected edges must be
        stored in both directions.  Vertex ids are the row numbers, so a vertex without edges
        still counts."""
        import pandas as pd

        if self.edgelist is not None:
            raise RuntimeError("Graph already has values")
        off = np.asarray(offset_col).astype(np.int64)
        idx = np.asarray(index_col)
        if off.ndim != 1 or len(off) < 1 or off[0] != 0 or off[-1] != len(idx) or np.any(np.diff(off) < 0):
            raise ValueError("offset_col must hold V + 1 non-decreasing offsets from 0 to the number of edges")
        nv = len(off) - 1
        if len(idx) and (idx.min() < 0 or idx.max() >= nv):
            raise ValueError("index_col must hold vertex ids in [0, V)")
        id_dtype = idx.dtype if idx.dtype in (np.int32, np.int64) else np.int32
        pdf = pd.DataFrame({"src": np.repeat(np.arange(nv, dtype=id_dtype), np.diff(off)), "dst": idx.astype(id_dtype)})
        if value_col is not None:
            pdf["weight"] = np.asarray(value_col)
        return self.from_pandas_edgelist(pdf, "src", "dst", None if value_col is None else "weight",
                                         renumber=renumber, store_transposed=store_transposed,
                                         vertices=np.arange(nv, dtype=id_dtype), _as_given=True)

    def from_pandas_adjacency(self, pdf):
        """graph_classes.py:443-457: adjacency matrix as a DataFrame, column labels name the nodes"""
        import pandas as pd

        if not isinstance(pdf, pd.DataFrame):
            raise TypeError("pdf input is not a Pandas DataFrame")
        return self.from_numpy_array(pdf.to_numpy(), pdf.columns)

    def from_numpy_array(self, np_array, nodes=None):
        """graph_classes.py:459-485: adjacency matrix; non-zero entries are the weighted edges"""
        import pandas as pd

        a = np.asarray(np_array)
        if len(a.shape) != 2:
            raise ValueError("np_array is not a 2D matrix")
        src, dst = a.nonzero()
        weight = a[src, dst]
        if nodes is not None:
            nodes = np.asarray(nodes)
            pdf = pd.DataFrame({"src": nodes[src], "dst": nodes[dst]})
        else:
            pdf = pd.DataFrame({"src": src.astype(np.int32), "dst": dst.astype(np.int32)})
        pdf["weight"] = weight if weight.dtype in (np.float32, np.float64) else weight.astype(np.float64)
        return self.from_pandas_edgelist(pdf, "src", "dst", edge_attr="weight")

    def from_numpy_matrix(self, np_matrix):
        """graph_classes.py:487-500"""
        if not isinstance(np_matrix, np.matrix):
            raise TypeError("np_matrix input is not a Numpy matrix")
        return self.from_numpy_array(np.asarray(np_matrix))

    def _make_plc_graph(self):
        import torch

        from .. import pylibcugraph as plc

        df = self.edgelist.edgelist_df
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        self._resource_handle = plc.ResourceHandle()
        props = plc.GraphProperties(is_symmetric=not self.is_directed(), is_multigraph=self.multi)
        src, dst = dev(df["src"].to_numpy()), dev(df["dst"].to_numpy())
        wgt = dev(df["weights"].to_numpy()) if self.edgelist.weights else None
        verts = None if self._vertices is None else dev(self._vertices)
        self._plc_graph = plc.SGGraph(self._resource_handle, props, src, dst, weight_array=wgt,
                                      store_transposed=self.store_transposed,
                                      renumber=self.renumbered, do_expensive_check=False,
                                      vertices_array=verts)
        self._num_vertices = self._plc_graph.info()["num_vertices"]


class MultiGraph(Graph):
    def __init__(self, directed=False):
        super().__init__(directed=directed)
        self.multi = True


def _drop_multi(df):
    """one edge per (src, dst); the minimum weight survives (symmetrize.py:100-108 groupby.min)"""
    if "weights" in df.columns:
        return df.groupby(["src", "dst"], as_index=False, sort=False).min()
    return df.drop_duplicates(["src", "dst"], ignore_index=True)


def _symmetrize(df, multi):
    """undirected graph: both directions of every edge (symmetrize.py:24-120)"""
    import pandas as pd

    rev = df.rename(columns={"src": "dst", "dst": "src"})[list(df.columns)]
    out = pd.concat([df, rev], ignore_index=True)
    return out if multi else _drop_multi(out)
