"""Module mirrors of the reference's `generator/` package (same constructors, state_dict keys and
predict signatures); compute runs in libtsg.so."""
from .gat import GATLayerImp3
from .distance_gat_fc import DistanceGatFC
from .function_g_fc import FunctionGFC, IntraAttention
from .generator_v4 import GeneratorV4

__all__ = ['GATLayerImp3', 'DistanceGatFC', 'FunctionGFC', 'IntraAttention', 'GeneratorV4']
