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

__all__ = ["TSRN", "TSRN_TL_TRANS", "TBSRN", "CRNN"]
