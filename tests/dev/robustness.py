#!/usr/bin/env python3
"""Parity of both arithmetics under unfriendly weight / input scales (vs the CPU oracle on the same box)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np, torch
from articulatory_amd.models import HiFiGANGenerator
from articulatory_amd.utils.synth import synth_features, synth_state_dict
from oracle import hificar_oracle as O
from tests.conftest import E2W_PARAMS, rel_err

torch.set_num_threads(16)
for gain, in_scale in ((1.0, 1.0), (2.0, 1.0), (0.3, 1.0), (1.0, 20.0), (1.5, 5.0)):
    sd = synth_state_dict(E2W_PARAMS, seed=77, gain=gain)
    w = O.fold_weight_norm(sd)
    x = torch.from_numpy(synth_features(2, 50, 13, seed=3) * in_scale)
    with torch.no_grad():
        ref = O.ar_loop_batched(w, E2W_PARAMS, x, 2000, 80)
        # pre-tanh magnitude: how saturated is the output?
    res = []
    for prec in ("f32", "bf16x3"):
        g = HiFiGANGenerator(**E2W_PARAMS, precision=prec)
        g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        g.remove_weight_norm(); g = g.eval().cuda()
        with torch.no_grad():
            y = g.ar_synthesis(x.permute(0, 2, 1).contiguous().cuda(), 25).cpu()
        res.append(f"{prec} {rel_err(y.numpy(), ref.numpy()):.2e}")
    print(f"gain {gain} input x{in_scale}: AR loop max|y|={float(ref.abs().max()):.3f}  " + "  ".join(res), flush=True)
    # one chunk (no feedback amplification): same weights, random AR context
    ar = torch.from_numpy(np.random.default_rng(5).uniform(-0.5, 0.5, (2, 1, 512)).astype(np.float32))
    c1 = x[:, :25].permute(0, 2, 1).contiguous()
    with torch.no_grad():
        ref1 = O.generator_forward(w, E2W_PARAMS, c1, ar=ar)
    res = []
    for prec in ("f32", "bf16x3"):
        g = HiFiGANGenerator(**E2W_PARAMS, precision=prec)
        g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        g.remove_weight_norm(); g = g.eval().cuda()
        with torch.no_grad():
            y = g(c1.cuda(), ar=ar.cuda()).cpu()
        res.append(f"{prec} {rel_err(y.numpy(), ref1.numpy()):.2e}")
    print(f"    one chunk: max|y|={float(ref1.abs().max()):.3f}  " + "  ".join(res), flush=True)
