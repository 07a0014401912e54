#!/usr/bin/env python3
"""Throughput sweep over BASELINE.json's configurations: HiFi-CAR (chunk 25 / 100) and non-AR 12-dim, batch 1/8/64,
10-s clips, for both arithmetics.  python tools/sweep.py [--precisions bf16x3 f32]"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from articulatory_amd.models import GBlockGenerator, HiFiGANGenerator  # noqa: E402
from articulatory_amd.utils.synth import synth_features, synth_gblock_state_dict, synth_state_dict  # noqa: E402
from bench import CAR_PARAMS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precisions", nargs="+", default=["bf16x3", "f32"])
ap.add_argument("--batches", nargs="+", type=int, default=[1, 8, 64])
a = ap.parse_args()


def make(params, prec):
    sd = synth_state_dict(params, seed=1234)
    g = HiFiGANGenerator(**params, precision=prec)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    g.remove_weight_norm()
    return g.eval().cuda()


GBLOCK_PARAMS = dict(in_channels=141, out_channels=1, channels=512, kernel_size=7, g_scales=[5, 1, 4, 1, 1, 2, 1, 2, 1, 1], g_kernel_sizes=[3] * 10,
                     use_weight_norm=True, use_ar=True, ar_input=512, ar_hidden=256, ar_output=128, use_tanh=True)


def make_gblock():
    """The reference's other a2w generator (gblock_gen.py:14-132) at its runnable shape: ten GBlocks, x80, channels 512 (exact fp32 only)."""
    g = GBlockGenerator(**GBLOCK_PARAMS)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in synth_gblock_state_dict(GBLOCK_PARAMS, seed=1234).items()})
    g.remove_weight_norm()
    return g.eval().cuda()


def timeit(fn, n_samples, reps=3):
    with torch.no_grad():
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return n_samples / dt, dt


PEAK = {"bf16x3": 2500.0, "f32": 157.3}  # dense MFMA peak of the arithmetic's instruction, TFLOP/s
print("| model | arithmetic | batch | samples/s | x real time | ms per 10 s batch | algorithmic TFLOP/s | of MFMA peak |")
print("|---|---|---:|---:|---:|---:|---:|---:|")
T = 2000
for prec in a.precisions:
    car = make(dict(CAR_PARAMS), prec)
    nonar = make(dict(CAR_PARAMS, in_channels=12, use_ar=False), prec)
    gblock = make_gblock() if prec == "f32" else None
    for B in a.batches:
        x13 = torch.from_numpy(synth_features(B, T, 13, seed=5)).permute(0, 2, 1).contiguous().cuda()
        x12 = torch.from_numpy(synth_features(B, T, 12, seed=6)).permute(0, 2, 1).contiguous().cuda()
        for name, fn, macs in (("HiFi-CAR chunk 25", lambda: car.ar_synthesis(x13, 25), 80 * car.macs(B, 25)),
                               ("HiFi-CAR chunk 100", lambda: car.ar_synthesis(x13, 100), 20 * car.macs(B, 100)),
                               ("HiFi-GAN non-AR 12-dim", lambda: nonar(x12), nonar.macs(B, T))) + \
                (() if gblock is None else (("GBlockGenerator chunk 25", lambda: gblock.ar_synthesis(x13, 25), 80 * gblock.macs(B, 25)),)):
            sps, dt = timeit(fn, B * T * 80)
            tf = 2.0 * macs / dt / 1e12
            print(f"| {name} | {prec} | {B} | {sps / 1e6:.2f} M | {sps / 16000:.0f} | {dt * 1e3:.1f} | {tf:.1f} | {tf / PEAK[prec]:.3f} |",
                  flush=True)
