"""sgs_gnn_amd -- MI355X (gfx950) native hot path of SGS-GNN behind the reference's own
Python interface.  The numeric work lives in libsgs_hip.so (csrc/*.hip, C ABI in
include/sgs_hip.h); this package is the host-side mirror of the reference's call sites
(model.py, sampling.py, training*.py, utils.py).  No CPU fallback exists."""
from . import _lib, ops  # noqa: F401
from . import torch_ops  # noqa: F401  (registers torch.ops.sgs.*)
from .model import GCNConv, GNNModel, GATConv, GAT, GATModel, GINConv, GIN, GINModel, ChebConv, ChebModel, set_dropout_seed  # noqa: F401
from .scorer import EdgeProbGCN, EdgeProbMLP, EdgeProbSAGE, SAGEConv, get_edge_mlp  # noqa: F401
from .sampling import gumbel_softmax_sampling, random_edge_sampling, manual_seed  # noqa: F401
from .training import train, train_hybrid, train_straight_through, train_two_pass, prepare_step_graphs  # noqa: F401
from .evaluate import evaluate, ensemble_evaluate  # noqa: F401
from .utils import calculate_f1, consistency_loss, fix_seeds, GpuMemoryProfiler  # noqa: F401
from .optim import FusedAdam  # noqa: F401
from .data import Batch, ResidentPartitions, degree_prior, synthetic_graph, reddit_partition_stream, reddit_partition_sizes  # noqa: F401
