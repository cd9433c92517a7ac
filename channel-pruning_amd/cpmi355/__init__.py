"""cpmi355 -- host-side plumbing over libcpmi355.so (ctypes), MI355X / gfx950 only."""
from .capi import (CP_CD_DELTA, CP_CD_RECIPROCAL, CP_F32, CP_F64, Context, CpError, DevBuf, default_context,  # noqa: F401
                   device_count, load)
from .pruner import LayerProblem, prune_layer, prune_layers_batched  # noqa: F401
