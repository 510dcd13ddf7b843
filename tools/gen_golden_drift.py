#!/usr/bin/env python3
"""fp64 yardstick of the multi-step drift test (tests/test_model_gpu.py::test_split_bf16_vs_fp32_training_drift): 12 training steps of
the TATT generator (B = 8, STN on, dropout off, fresh data every step, Adam lr 1e-5, clip 0.25) evaluated by the CPU oracle in float64
(oracle.tatt_oracle.train_step_fp64) for three seeds.  Writes tests/golden/drift_fp64.npz: losses [3][12] (float64) and the l2 norm of
the weights after step 12.  The oracle itself is pinned to the imported reference by tests/golden/*.npz (tools/gen_golden.py); this
script needs no /root/reference.  CPU only, ~10 minutes on 8 cores.

    python tools/gen_golden_drift.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tatt_amd  # noqa: E402
from oracle import tatt_oracle as O  # noqa: E402
from oracle.fixtures import make_inputs, randomize_state_dict  # noqa: E402

SEEDS = (1234, 2345, 3456)
NSTEP, B, LR = 12, 8, 1e-5
STD = dict(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32)


def data_seed(model_seed, step):
    return 40 + step + (0 if model_seed == 1234 else model_seed)      # (seed 1234 = the drift test's historical inputs 40, 41, ...)


def main():
    torch.set_num_threads(os.cpu_count())
    losses = np.zeros((len(SEEDS), NSTEP))
    wnorm = np.zeros(len(SEEDS))
    for si, seed in enumerate(SEEDS):
        torch.manual_seed(seed)
        m = tatt_amd.TSRN_TL_TRANS(**STD)
        sd = randomize_state_dict(m.state_dict())
        sd = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        opt = {}
        for i in range(NSTEP):
            x, tp, hr = make_inputs(B, seed=data_seed(seed, i))
            loss, _, sd, opt, _, total = O.train_step(sd, x.double(), tp.double(), hr.double(), tatt=True, stn=True, drop_on=False,
                                                      opt_state=opt, step=i + 1, lr=LR)
            losses[si, i] = float(loss)
            print("seed %d step %2d loss %.12f  |g| %.6f" % (seed, i + 1, losses[si, i], float(total)), flush=True)
        wnorm[si] = float(torch.sqrt(sum((v.double() ** 2).sum() for k, v in sd.items() if O.is_param(k))))
    out = os.path.join(ROOT, "tests", "golden", "drift_fp64.npz")
    np.savez(out, seeds=np.array(SEEDS), losses=losses, weight_norm=wnorm, nstep=NSTEP, batch=B, lr=LR)
    print("wrote", out)


if __name__ == "__main__":
    main()
