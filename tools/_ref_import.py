"""Import the read-only reference (``/root/reference``) in THIS container only.

Used by ``tools/gen_golden.py`` to validate the oracle and to emit golden
vectors.  The reference never travels to the GPU box; nothing under
``tests/ -m gpu``, ``bench.py`` or ``__graft_entry__.smoke()`` imports this.

The reference's hot-path modules import three packages that are absent here and
unused on the path: ``IPython`` (model/tsrn.py:9), ``cv2``
(model/transformer_v2.py:20) and ``torchvision`` (model/model_transformer.py:2,11).
They are replaced by empty module stubs.
"""
import sys
import types
import warnings

REF = "/root/reference"


def import_reference():
    warnings.filterwarnings("ignore")
    for name in ("IPython", "cv2", "torchvision", "torchvision.models"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["IPython"].embed = lambda *a, **k: None
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from model import tsrn as ref_tsrn  # noqa
    return ref_tsrn
