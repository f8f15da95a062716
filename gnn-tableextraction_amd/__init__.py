"""MI355X-native hot path of GNN-TableExtraction: the GcnSAGE node classifier and its train step.

Import name: ``gnn_tableextraction_amd`` (the directory is ``gnn-tableextraction_amd``; the shim
``gnn_tableextraction_amd.py`` at the repository root maps one onto the other).
"""
from . import _lib, function, graph, ops                                   # noqa: F401
from .graph import PageGraph, batch, from_edge_index                       # noqa: F401
from .components.graphs.models import GcnSAGE, GcnSAGELayer, MeanSAGE, WeightedMeanSAGELayer  # noqa: F401
from .components.graphs.gat import GAT, GATLayer                          # noqa: F401

__version__ = "0.1.0"
