"""Host-side cost of ENQUEUEING the native passes (no synchronisation inside the timed region): is the training iteration launch-bound on
the host?   python tests/dev/host_time_probe.py"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from articulatory_amd.models import HiFiGANGenerator, HiFiGANMultiScaleMultiPeriodDiscriminator  # noqa: E402
from articulatory_amd.utils.recipes import recipe_train_config  # noqa: E402

cfg = recipe_train_config("car")
d = HiFiGANMultiScaleMultiPeriodDiscriminator(**cfg["discriminator_params"]).cuda()
g = HiFiGANGenerator(**cfg["generator_params"], precision="f32").cuda().train()
real = torch.rand(64, 1, 2512, device="cuda") - 0.5
fake = torch.rand(64, 1, 2512, device="cuda") - 0.5
x = torch.randn(64, 13, 25, device="cuda")
ar = torch.zeros(64, 1, 512, device="cuda")


def timed(name, fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    th = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    tw = (time.perf_counter() - t0) / n
    print(f"{name:44s} host {th * 1e3:7.2f} ms   wall {tw * 1e3:7.2f} ms")


def d_fwd():
    with torch.no_grad():
        d(real, native=True)


def d_step():
    d.zero_grad(set_to_none=True)
    t, _, _ = d.discriminator_loss(fake, real, average_by_discriminators=False)
    t.backward()


def g_side():
    xx = fake.clone().requires_grad_(True)
    t, _, _ = d.generator_loss(xx, real, average_by_discriminators=False, lambda_adv=1.0, lambda_feat_match=2.0, fm_average_by_layers=False,
                               fm_average_by_discriminators=False)
    torch.autograd.grad(t, xx)


def g_fwd():
    with torch.no_grad():
        g(x, ar=ar)


def g_fb():
    g.zero_grad(set_to_none=True)
    g(x, ar=ar).abs().mean().backward()


timed("discriminator forward", d_fwd)
timed("discriminator step (2 fwd + 2 bwd)", d_step)
timed("generator-side D (2 fwd + data bwd)", g_side)
timed("generator forward (no grad)", g_fwd)
timed("generator forward + backward", g_fb)
