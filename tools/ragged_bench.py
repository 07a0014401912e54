#!/usr/bin/env python3
"""Throughput on a mixed-length dataset (ragged batches) vs one utterance at a time.
   python tools/ragged_bench.py [--n 256] [--batch 64]"""
import argparse, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
from articulatory_amd.models import HiFiGANGenerator
from articulatory_amd.bin.decode import ar_loop, ar_loop_ragged, length_batches
from articulatory_amd.utils.synth import synth_features, synth_state_dict
from bench import CAR_PARAMS

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--precision", default="f32")
a = ap.parse_args()
sd = synth_state_dict(CAR_PARAMS, seed=1234)
g = HiFiGANGenerator(**CAR_PARAMS, precision=a.precision)
g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
g.remove_weight_norm(); g = g.eval().cuda()
rng = np.random.default_rng(1)
lens = rng.integers(300, 2001, size=a.n)  # 1.5 - 10 s utterances
utts = [(f"u{i}", torch.from_numpy(synth_features(1, int(T), 13, seed=i)[0]).cuda()) for i, T in enumerate(lens)]
config = dict(generator_params=dict(CAR_PARAMS), hop_size=80, batch_max_steps=2000, sampling_rate=16000)
total = int(lens.sum()) * 80
with torch.no_grad():
    for mode in ("ragged", "ragged", "one-at-a-time"):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode == "ragged":
            pad = int(lens.sum())
            ys = ar_loop_ragged(g, [c for _, c in utts], config, batch=a.batch)  # one continuously batched device call
        else:
            for _, c in utts[:32]:
                ar_loop(g, c, config)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        n = total if mode == "ragged" else int(lens[:32].sum()) * 80
        extra = f" ({a.batch} utterances in flight)" if mode == "ragged" else " (first 32 utterances)"
        print(f"{mode}: {n / dt / 1e6:.2f} M samples/s ({n / dt / 16000:.0f} x real time){extra}", flush=True)
