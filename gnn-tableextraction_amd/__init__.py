"""MI355X-native hot path of GNN-TableExtraction: the GcnSAGE node classifier and its train step.

Import name: ``gnn_tableextraction_amd`` (the directory is ``gnn-tableextraction_amd``; the shim
``gnn_tableextraction_amd.py`` at the repository root maps one onto the other).
"""
import os as _os

# HIP deals a process's streams round-robin onto GPU_MAX_HW_QUEUES hardware queues (4 by default); two streams on one queue run
# IN ORDER with each other.  This package keeps several streams busy at once -- the step's launch stream, the batch-assembly side
# stream (models/loop.py), the window-upload stream of a host-resident training set (models/residency.py) -- and an upload that
# lands on the queue of a stream full of kernels crawls (measured: 21 instead of 38 GB/s).  More queues, fewer collisions; read
# by the runtime when HIP initialises, so it is set here, at import, unless the user chose a value.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


from . import _lib, function, graph, ops                                   # noqa: F401
from .graph import PageGraph, batch, from_edge_index                       # noqa: F401
from .components.graphs.models import GcnSAGE, GcnSAGELayer, MeanSAGE, WeightedMeanSAGELayer  # noqa: F401
from .components.graphs.gat import GAT, GATLayer                          # noqa: F401

__version__ = "0.1.0"
