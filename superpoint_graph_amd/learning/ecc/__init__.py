"""Mirror of the reference's `learning/ecc` package (learning/ecc/__init__.py) for the SPG hot path."""
from .GraphConvInfo import GraphConvInfo
from .GraphConvModule import GraphConvFunction, GraphConvModule
from .utils import get_edge_shards
