#!/usr/bin/env python3
"""Per-kernel time of one HiFi-CAR AR chunk step at a given batch (library event profile):
   python tools/batch_profile.py --batch 8 [--precision f32] [--frames 25]"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import time  # noqa: E402

import torch  # noqa: E402

from articulatory_amd.models import HiFiGANGenerator  # noqa: E402
from articulatory_amd.utils.synth import synth_features, synth_state_dict  # noqa: E402
from bench import CAR_PARAMS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--frames", type=int, default=25)
ap.add_argument("--precision", default="f32")
ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
params = dict(CAR_PARAMS)
sd = synth_state_dict(params, seed=1234)
g = HiFiGANGenerator(**params, precision=a.precision)
g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
g = g.eval().cuda()
c = torch.from_numpy(synth_features(a.batch, a.frames, 13, seed=1)).permute(0, 2, 1).contiguous().cuda()
ar = torch.zeros(a.batch, 1, 512, device="cuda")
with torch.no_grad():
    for _ in range(20):
        g(c, ar=ar)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        g(c, ar=ar)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"batch {a.batch} x {a.frames} frames {a.precision}: {dt * 1e6:.1f} us per chunk step, {a.batch * 80 * a.frames / dt / 1e6:.2f} M samples/s")
    g.profile_begin()
    for _ in range(10):
        g(c, ar=ar)
    torch.cuda.synchronize()
    st = g.profile_end()
tot = sum(s["total_ms"] for s in st) / 10
print(f"kernel time {tot * 1e3:.1f} us per step")
for s in st:
    print(f"  {s['name']:44s} {s['launches'] // 10:4d} launches {s['total_ms'] * 100:8.1f} us  {s['flops'] / max(s['total_ms'], 1e-9) / 1e9:7.1f} TF-alg")
