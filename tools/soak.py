#!/usr/bin/env python3
"""Race / determinism soak: repeat the batch-64 AR synthesis and require bit-identical waveforms every time.
   python tools/soak.py [--reps 100] [--frames 500]"""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from articulatory_amd.models import HiFiGANGenerator
from articulatory_amd.utils.synth import synth_features, synth_state_dict
from bench import CAR_PARAMS

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=100)
ap.add_argument("--frames", type=int, default=500)
a = ap.parse_args()
sd = synth_state_dict(CAR_PARAMS, seed=1234)
for prec in ("bf16x3", "f32"):
    g = HiFiGANGenerator(**CAR_PARAMS, precision=prec)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    g.remove_weight_norm(); g = g.eval().cuda()
    for B, chunk in ((64, 25), (24, 25), (8, 100), (3, 25), (1, 25)):  # (24: the two-stream form of the AR loop)
        x = torch.from_numpy(synth_features(B, a.frames, 13, seed=B)).permute(0, 2, 1).contiguous().cuda()
        with torch.no_grad():
            ref = g.ar_synthesis(x, chunk).clone()
            bad = 0
            for _ in range(a.reps):
                bad += int(not torch.equal(g.ar_synthesis(x, chunk), ref))
        print(f"{prec} B={B} chunk={chunk}: {a.reps} repeats, {bad} differing", flush=True)
        assert bad == 0
    # continuous batching over a mixed-length list (tile skipping, step table): repeated calls must agree bit for bit
    import numpy as np
    lens = [int(v) for v in np.random.default_rng(3).integers(20, 400, size=96)]
    lens.sort(reverse=True)
    x = torch.from_numpy(synth_features(len(lens), max(lens), 13, seed=9)).permute(0, 2, 1).contiguous().cuda()
    with torch.no_grad():
        ref = g.ar_synthesis_packed(x, 25, lens, batch=32).clone()
        bad = sum(int(not torch.equal(g.ar_synthesis_packed(x, 25, lens, batch=32), ref)) for _ in range(a.reps // 2))
        one = g.ar_synthesis(x[5:6, :, :lens[5]].contiguous(), 25)
    # utterance 5 alone is a different launch shape (split-K form on small launches): equal to rounding, bit-equal with HIFICAR_KSPLIT=0
    close = float((ref[5, :80 * lens[5]] - one[0]).abs().max() / one[0].abs().max())
    print(f"{prec} packed 96 utterances, 32 in flight: {a.reps // 2} repeats, {bad} differing; utterance 5 vs alone: {close:.1e}", flush=True)
    assert bad == 0 and close < (5e-6 if prec == "f32" else 2e-4)
print("soak ok")
