#!/usr/bin/env python3
"""Generator forward + backward (the generator half of BASELINE config 5's train step) on one MI355X:
   python tools/train_bench.py [--batch 16] [--frames 25] [--steps 10]
batch_size 64 x batch_max_steps 2000 samples (25 frames) is the reference recipe's shape (e2w_hifigan_car.yaml:134-135).
Loss here is a plain L1 against a random target (the reference's mel / adversarial losses and discriminators are not built):
the timed region is generator forward (tape kept) + backward (all parameter gradients, weight norm in the graph) + Adam step."""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from articulatory_amd.models import HiFiGANGenerator  # noqa: E402
from articulatory_amd.utils.synth import synth_features, synth_state_dict  # noqa: E402
from bench import CAR_PARAMS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--frames", type=int, default=25)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--profile", action="store_true")
ap.add_argument("--fused-adam", action="store_true", help="torch.optim.Adam(fused=True) instead of the default foreach implementation")
ap.add_argument("--torch-profile", action="store_true", help="torch.profiler table of one step (torch-side kernels around the native path)")
a = ap.parse_args()
params = dict(CAR_PARAMS)
sd = synth_state_dict(params, seed=1234)
g = HiFiGANGenerator(**params, precision="f32")
g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
g = g.train().cuda()
opt = torch.optim.Adam(g.parameters(), lr=1e-4, betas=(0.5, 0.9), fused=True if a.fused_adam else None)
c = torch.from_numpy(synth_features(a.batch, a.frames, 13, seed=1)).permute(0, 2, 1).contiguous().cuda()
ar = torch.zeros(a.batch, 1, 512, device="cuda")
target = torch.rand(a.batch, 1, 80 * a.frames, device="cuda") - 0.5


def step():
    opt.zero_grad(set_to_none=True)
    y = g(c, ar=ar)
    loss = (y - target).abs().mean()
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
macs = g.macs(a.batch, a.frames)
print(f"generator train step (fwd + bwd + Adam), batch {a.batch} x {a.frames} frames: {dt * 1e3:.2f} ms/step, "
      f"{6 * macs / dt / 1e12:.1f} TFLOP/s algorithmic (3 x forward FLOPs), {a.batch * 80 * a.frames / dt / 1e6:.2f} M samples/s, loss {float(loss):.4f}")
if a.profile:
    g.profile_begin()
    step()
    torch.cuda.synchronize()
    st = g.profile_end()
    tot = sum(s["total_ms"] for s in st)
    print(f"kernel time {tot:.2f} ms per step")
    for s in st[: int(os.environ.get("TRAIN_BENCH_ROWS", "14"))]:
        print(f"  {s['name']:44s} {s['launches']:4d} launches {s['total_ms']:8.3f} ms  {s['flops'] / max(s['total_ms'], 1e-9) / 1e9:7.1f} TF-alg")
if a.torch_profile:
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
    ev = {}
    for name, fn in (("zero_grad", lambda: opt.zero_grad(set_to_none=True)),):
        pass
    # wall-clock split of one step
    def timed(fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, (time.perf_counter() - t) * 1e3

    opt.zero_grad(set_to_none=True)
    y, t_f = timed(lambda: g(c, ar=ar))
    loss, t_l = timed(lambda: (y - target).abs().mean())
    _, t_b = timed(loss.backward)
    _, t_o = timed(opt.step)
    print(f"serialised: forward {t_f:.2f} ms, loss {t_l:.2f} ms, backward {t_b:.2f} ms, optimizer {t_o:.2f} ms")
