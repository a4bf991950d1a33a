"""Mirror of the hot-path helpers of the reference's utils.py."""
from __future__ import annotations

import random

import numpy as np
import torch

from . import ops


def calculate_f1(logits, labels, mask) -> float:
    """utils.py:163-169: sklearn micro-F1 of the argmax on masked rows == accuracy.  Computed on
    the device (sgs_masked_correct); only two ints cross to the host."""
    c = ops.masked_correct(logits, labels, mask).tolist()
    return c[0] / c[1] if c[1] else 0.0


def consistency_loss(edge_probs, edge_indices, node_embeddings):
    """utils.py:187-211: MSE(edge_probs, cos(emb[src], emb[dst]))."""
    N = node_embeddings.shape[0]
    dev = node_embeddings.device
    y = torch.zeros(N, dtype=torch.int64, device=dev)
    tm = torch.zeros(N, dtype=torch.bool, device=dev)
    total, _ = ops.edge_regularizers(edge_probs, node_embeddings, edge_indices, y, tm, 0.0, 1.0)
    return total


def fix_seeds(seed=42):
    """utils.py:82-89, plus the counter-based noise / dropout streams of this package."""
    from .sampling import manual_seed
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    manual_seed(seed)
