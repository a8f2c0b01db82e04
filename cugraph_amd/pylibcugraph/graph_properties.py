"""GraphProperties (reference: python/pylibcugraph/pylibcugraph/graph_properties.pyx)."""
from .. import _capi


class GraphProperties:
    def __init__(self, is_symmetric=False, is_multigraph=False):
        self.c_graph_properties = _capi.GraphPropertiesStruct(int(bool(is_symmetric)),
                                                              int(bool(is_multigraph)))

    @property
    def is_symmetric(self):
        return bool(self.c_graph_properties.is_symmetric)

    @is_symmetric.setter
    def is_symmetric(self, value):
        self.c_graph_properties.is_symmetric = int(bool(value))

    @property
    def is_multigraph(self):
        return bool(self.c_graph_properties.is_multigraph)

    @is_multigraph.setter
    def is_multigraph(self, value):
        self.c_graph_properties.is_multigraph = int(bool(value))
