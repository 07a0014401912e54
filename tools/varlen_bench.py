#!/usr/bin/env python3
"""Non-AR decode of a dataset of DISTINCT utterance lengths, one utterance per call (the shape of the reference's loop,
egs/ema/voc1/local/predict_wav.py:124-137), against the same amount of audio at one fixed length.
   python tools/varlen_bench.py [--n 200] [--precision f32]
Every new length is a new launch shape: the library builds its tile schedules on the host and uploads them asynchronously from
a pinned arena (no allocation, no blocking copy), and rounds launch geometry to 32-frame buckets so that keys repeat.
Run under `rocprofv3 --hip-trace --stats` to count hipMalloc / hipMemcpy in the steady state."""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from articulatory_amd.models import HiFiGANGenerator  # noqa: E402
from articulatory_amd.utils.synth import synth_features, synth_state_dict  # noqa: E402
from bench import CAR_PARAMS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=200)
ap.add_argument("--precision", default="f32")
ap.add_argument("--passes", type=int, default=2)
a = ap.parse_args()
params = dict(CAR_PARAMS, in_channels=12, use_ar=False)
sd = synth_state_dict(params, seed=1234)
g = HiFiGANGenerator(**params, precision=a.precision)
g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
g.remove_weight_norm()
g = g.eval().cuda()
rng = np.random.default_rng(0)
lens = rng.permutation(np.arange(300, 300 + 8 * a.n, 8) + rng.integers(0, 8, a.n))[: a.n]  # a.n distinct lengths, 1.5 .. 9.5 s
assert len(set(lens.tolist())) == a.n
mean_len = int(round(lens.mean()))
x_var = [torch.from_numpy(synth_features(1, int(T), 12, seed=int(T))).permute(0, 2, 1).contiguous().cuda() for T in lens]
x_fix = [torch.from_numpy(synth_features(1, mean_len, 12, seed=i)).permute(0, 2, 1).contiguous().cuda() for i in range(a.n)]


def run(xs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    with torch.no_grad():
        for x in xs:
            n += g(x).numel()
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


with torch.no_grad():
    g(x_fix[0])
first_var = run(x_var)  # every length seen for the first time: schedules built + uploaded on the fly
fixed = max(run(x_fix) for _ in range(a.passes))
steady_var = max(run(x_var) for _ in range(a.passes))
print(f"{a.n} utterances, lengths {lens.min()}..{lens.max()} frames (all distinct), mean {mean_len}, {a.precision}")
print(f"equal-length      : {fixed / 1e6:8.2f} M samples/s")
print(f"distinct, 1st pass: {first_var / 1e6:8.2f} M samples/s  ({first_var / fixed:.3f} of equal-length; schedules built on first use)")
print(f"distinct, steady  : {steady_var / 1e6:8.2f} M samples/s  ({steady_var / fixed:.3f} of equal-length)")
