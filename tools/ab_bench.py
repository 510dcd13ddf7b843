#!/usr/bin/env python3
"""A/B measurements: run bench.py with module-level test hooks of the product flipped, e.g.

    python tools/ab_bench.py tatt_amd.tsrn.TP_FUSED=0 -- --steps 30 --warmup 10 --no-cpu-baseline

(the operator-by-operator TP interpreter instead of the one-kernel layers).  Same box, same process layout as bench.py: gpurun
boxes differ by up to 15 %, so only runs inside ONE gpurun call compare."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sep = sys.argv.index("--") if "--" in sys.argv else len(sys.argv)
for kv in sys.argv[1:sep]:
    path, val = kv.split("=", 1)
    mod, attr = path.rsplit(".", 1)
    setattr(importlib.import_module(mod), attr, type(getattr(importlib.import_module(mod), attr))(int(val)) if val.lstrip("-").isdigit() else val)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[sep + 1:]
import bench  # noqa: E402
bench.main()
