#!/usr/bin/env python3
"""Discriminator forward / backward of the HiFi-CAR recipe (e2w_hifigan_car.yaml: batch 64, batch_max_steps 2000 + 512 AR context =
2512 samples, train.py:340-346) on one MI355X:   python tools/disc_bench.py [--batch 64] [--samples 2512] [--profile]
Timed: the discriminator step's three passes (D(real), D(fake), backward of both with parameter gradients) and the generator step's
(D(fake) with the gradient back to the waveform, D(real) without)."""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from articulatory_amd.models import HiFiGANMultiScaleMultiPeriodDiscriminator  # noqa: E402
from articulatory_amd.utils.synth import disc_params, synth_disc_state_dict  # noqa: E402

CAR_DISC = dict(scale_discriminator_params=dict(disc_params()["scale_discriminator_params"], downsample_scales=[4, 4, 4, 4, 1]))

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--samples", type=int, default=2512)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--profile", action="store_true")
a = ap.parse_args()
sd = synth_disc_state_dict(CAR_DISC, seed=4321)
d = HiFiGANMultiScaleMultiPeriodDiscriminator(**CAR_DISC)
d.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
d = d.cuda()
real = torch.rand(a.batch, 1, a.samples, device="cuda") - 0.5
fake = torch.rand(a.batch, 1, a.samples, device="cuda") - 0.5


def d_step():
    d.zero_grad(set_to_none=True)
    total, _, _ = d.discriminator_loss(fake, real, average_by_discriminators=False)
    total.backward()


def g_step():
    x = fake.clone().requires_grad_(True)
    total, _, _ = d.generator_loss(x, real, average_by_discriminators=False, lambda_adv=1.0, lambda_feat_match=2.0, fm_average_by_layers=False,
                                   fm_average_by_discriminators=False)
    (gx,) = torch.autograd.grad(total, x)
    return gx


def fwd():
    with torch.no_grad():
        d(real, native=True)


for name, fn in (("forward", fwd), ("discriminator step (2 fwd + bwd)", d_step), ("generator-side (2 fwd + signal gradient)", g_step)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / a.steps * 1e3:.2f} ms  (batch {a.batch} x {a.samples} samples)")
if a.profile:
    d.profile_begin()
    d_step()
    torch.cuda.synchronize()
    st = d.profile_end()
    tot = sum(s["total_ms"] for s in st)
    print(f"discriminator step kernel time {tot:.2f} ms")
    for s in st[: int(os.environ.get("DISC_BENCH_ROWS", "16"))]:
        print(f"  {s['name']:44s} {s['launches']:4d} launches {s['total_ms']:8.3f} ms  {s['flops'] / max(s['total_ms'], 1e-9) / 1e9:7.1f} TF-alg")
