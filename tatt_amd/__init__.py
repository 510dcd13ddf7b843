"""tatt_amd -- MI355X (gfx950) native implementation of the TATT super-resolution hot path.

    from tatt_amd import TSRN, TSRN_TL_TRANS      # drop-ins for the reference's model/tsrn.py classes
    from tatt_amd import TBSRN                    # drop-in for the reference's model/tbsrn.py class
    from tatt_amd import CRNN                     # drop-in for the reference's model/crnn/crnn.py text-prior generator

Host code is Python on PyTorch-ROCm (device memory, streams, autograd tape, torch.distributed/RCCL);
all arithmetic of the path runs in hand-written HIP kernels (tatt_amd/csrc -> lib/libtatt_hip.so, C ABI in
include/tatt_hip.h).  No CPU fallback: see oracle/ for the CPU restatement used by the tests.
"""
from .tsrn import TSRN, TSRN_TL_TRANS  # noqa: F401
from .tbsrn import TBSRN  # noqa: F401
from .crnn import CRNN  # noqa: F401
from . import torch_ops  # noqa: F401  (registers torch.ops.tatt_hip.*: the operator-registry view of the kernels)

__all__ = ["TSRN", "TSRN_TL_TRANS", "TBSRN", "CRNN", "set_arithmetic", "get_arithmetic", "sync_check"]


def set_arithmetic(mode: str) -> None:
    """Process-wide choice of the arithmetic inside the GEMM-shaped kernels of the path (storage, accumulation and results are fp32
    either way; INTEGRATION.md "Arithmetic"):

      "split_bf16" (default) -- the 64-channel 3x3 convolutions (forward, data and weight gradients), the 9x9 convolutions at the image end (forward, data and weight gradients), the GruBlock projections, the
          GruBlock weight gradients, the backward of the TP-interpreter layers, the recurrent products and recurrent weight gradient of the query GRU and the TBSRN self-attention run on the bf16 matrix cores with every fp32 operand split a = hi + lo and
          a b ~ hi hi + hi lo + lo hi: 2^-16 relative per product, ~16-17 mantissa bits instead of 24
          (profiles/r03_split_bf16_probe.txt: eval SR moves 1.1e-6, gradients ~1e-5 relative);
      "fp32" -- the same operators on v_mfma_f32_* (exact fp32 products), about 2 ms per training step slower at B = 48 (`exact_fp32` in the bench line).

    Call it before building a Trainer / capturing a hipGraph: captured graphs keep the kernels they were captured with."""
    from . import functional as _F, ops as _ops
    if mode not in ("split_bf16", "fp32"):
        raise ValueError("arithmetic must be 'split_bf16' or 'fp32', got %r" % (mode,))
    on = mode == "split_bf16"
    _ops.CONV3_SB = _ops.CONV3_WGRAD_SB = _ops.TPLAYER_BWD2 = _ops.CONV9_SB = on
    _F.TOKGEMM_SB = _F.GRU_WGRAD_SB = _F.QGRU_CHAIN_SB = _F.QGRU_WGRAD_SB = _F.SATTN_SB = _F.TOK_WGRAD_SB = on


def get_arithmetic() -> str:
    from . import functional as _F, ops as _ops
    flags = (_ops.CONV3_SB, _ops.CONV3_WGRAD_SB, _ops.TPLAYER_BWD2, _ops.CONV9_SB, _F.TOKGEMM_SB, _F.GRU_WGRAD_SB, _F.QGRU_CHAIN_SB, _F.QGRU_WGRAD_SB, _F.SATTN_SB, _F.TOK_WGRAD_SB)
    return "split_bf16" if all(flags) else ("fp32" if not any(flags) else "mixed")


def sync_check() -> None:
    """Some launches of the path synchronise their work-groups in flight (the persistent query-GRU recurrences, the STN head's
    BatchNorm launches): every such wait is bounded by the wall clock (2 s) and raises an error word instead of hanging when it
    expires (work-groups that never became co-resident).  This reads those words and raises RuntimeError if one is set -- results of
    that launch are then invalid.  It synchronises the device: call it where the training loop synchronises anyway (logging, eval)."""
    from . import functional as _F
    _F.sync_check()
