#!/usr/bin/env python3
"""Print the parity error (max|y_gpu - y_ref| / max|y_ref|) of each conv arithmetic against the golden
vectors captured from the real reference.  Run on the GPU box: python tools/accuracy.py"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from articulatory_amd.models import HiFiGANGenerator  # noqa: E402
from articulatory_amd.utils.synth import synth_state_dict  # noqa: E402
from tests.conftest import E2W_PARAMS, GOLDEN, rel_err  # noqa: E402


def main():
    sd = synth_state_dict(E2W_PARAMS, seed=1234)
    for prec in ("f32", "bf16x3"):
        g = HiFiGANGenerator(**E2W_PARAMS, precision=prec)
        g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        g.remove_weight_norm()
        g = g.eval().cuda()
        with torch.no_grad():
            gold = np.load(os.path.join(GOLDEN, "gold_fwd_full.npz"))
            y = g(torch.from_numpy(gold["c"]).cuda(), ar=torch.from_numpy(gold["ar"]).cuda())
            e1 = rel_err(y.cpu().numpy(), gold["out"])
            gold = np.load(os.path.join(GOLDEN, "gold_arloop.npz"))
            x = torch.from_numpy(gold["x"]).cuda().t().unsqueeze(0)
            e2 = rel_err(g.ar_synthesis(x, 25)[0].cpu().numpy(), gold["out_bms2000"])
            e3 = rel_err(g.ar_synthesis(x, 100)[0].cpu().numpy(), gold["out_bms8000"])
            gold = np.load(os.path.join(GOLDEN, "gold_predict_wav.npz"))
            x = torch.from_numpy(gold["x"]).cuda().t().unsqueeze(0)
            e4 = rel_err(g.ar_synthesis(x, 100)[0].cpu().numpy(), gold["out"])
        print(f"{prec:7s} forward(B2,T25) {e1:.3e} | ar_loop 260f chunk25 {e2:.3e} chunk100 {e3:.3e} | predict_wav 700f {e4:.3e}")


if __name__ == "__main__":
    main()
