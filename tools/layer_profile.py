#!/usr/bin/env python3
"""Per-layer kernel times of one AR chunk step (HIP-event profile hooks, HIFICAR_PROFILE_DETAIL=1).
   python tools/layer_profile.py [--precision f32|bf16x3] [--batch 64] [--frames 25] [--steps 20]"""
import argparse
import os
import sys

os.environ["HIFICAR_PROFILE_DETAIL"] = "1"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from articulatory_amd.models import HiFiGANGenerator  # noqa: E402
from articulatory_amd.utils.synth import synth_features, synth_state_dict  # noqa: E402
from bench import CAR_PARAMS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="f32")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--frames", type=int, default=25)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
params = dict(CAR_PARAMS)
sd = synth_state_dict(params, seed=1234)
g = HiFiGANGenerator(**params, precision=a.precision)
g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
g.remove_weight_norm()
g = g.eval().cuda()
x = torch.from_numpy(synth_features(a.batch, a.frames * a.steps, 13, seed=1)).permute(0, 2, 1).contiguous().cuda()
with torch.no_grad():
    g.ar_synthesis(x, a.frames)
    g.profile_begin()
    g.ar_synthesis(x, a.frames)
    st = g.profile_end()
tot = sum(s["total_ms"] for s in st)
print(f"total {tot / a.steps * 1e3:.1f} us per chunk step ({a.batch} x {a.frames} frames)")
for s in sorted(st, key=lambda s: s["name"].split("|")[-1]):
    us = s["total_ms"] * 1e3 / s["launches"]
    print(f"{s['name']:75s} {s['launches'] // a.steps:2d}/step  {us:7.1f} us  {s['flops'] / s['total_ms'] / 1e9:7.1f} TF-alg  {100 * s['total_ms'] / tot:5.1f}%")
