"""sgs_gnn_amd -- MI355X (gfx950) native hot path of SGS-GNN behind the reference's own
Python interface.  The numeric work lives in libsgs_hip.so (csrc/*.hip, C ABI in
include/sgs_hip.h); this package is the host-side mirror of the reference's call sites."""
from . import _lib, ops  # noqa: F401

__all__ = ["_lib", "ops"]
