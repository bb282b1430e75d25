"""megreader_amd -- MI355X (gfx950) native training hot path for MegReader's recognition models.

Host-side mirror of the reference's plugin interface (``backbones``, ``decoders``, ``ops``, ``apex.parallel``)
on top of a C-ABI HIP library (``include/megreader_hip.h``).  PyTorch supplies device memory, streams,
autograd bookkeeping and ``torch.distributed`` (RCCL); every arithmetic kernel on the path is hand-written HIP.
"""
import os

import torch

_compute_dtype = torch.bfloat16 if os.environ.get("MEGREADER_DTYPE", "bf16").lower() in ("bf16", "bfloat16") \
    else torch.float32


def set_compute_dtype(dtype):
    """Storage type of activations / MFMA operands: torch.bfloat16 (default) or torch.float32 (parity mode)."""
    global _compute_dtype
    if dtype not in (torch.bfloat16, torch.float32):
        raise TypeError("compute dtype must be torch.bfloat16 or torch.float32")
    _compute_dtype = dtype


def get_compute_dtype():
    return _compute_dtype


__all__ = ["set_compute_dtype", "get_compute_dtype"]
