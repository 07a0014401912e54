#!/usr/bin/env python3
"""Full GAN training iteration of the HiFi-CAR recipe: libhificar (Trainer.train_step) vs the same iteration as plain PyTorch-ROCm ops
(the oracles' functional restatements of the reference's generator / discriminators / losses on the GPU through MIOpen, rocBLAS and
rocFFT under torch autograd, fp32, default MIOpen heuristics — `cudnn.benchmark` searches take minutes per new shape).  Development aid:
    python tests/dev/gan_step_vs_torch.py [--batch 64] [--steps 5]"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import disc_oracle as DO  # noqa: E402
from oracle import hificar_oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
sys.argv = [sys.argv[0], "--batch", str(a.batch), "--steps", str(a.steps)]
import gan_bench as GB  # noqa: E402  (runs the native bench and leaves trainer / batch / config behind)

cfg, batch = GB.config, GB.batch
dev = torch.device("cuda")
gp, dp = cfg["generator_params"], cfg["discriminator_params"]
G = {k: v.detach().clone().requires_grad_(True) for k, v in GB.trainer.G.state_dict().items()}
D = {k: v.detach().clone().requires_grad_(True) for k, v in GB.trainer.D.state_dict().items()}
og = torch.optim.Adam(list(G.values()), **cfg["generator_optimizer_params"])
od = torch.optim.Adam(list(D.values()), **cfg["discriminator_optimizer_params"])
x, y, ar = batch["x"].to(dev), batch["y"].to(dev), batch["ar"].to(dev)
melmat = torch.from_numpy(DO.mel_filterbank(16000, 1024, 80, 0, 11025).T.copy()).to(dev)
win = torch.hann_window(1024, device=dev)


def fold(sd):
    w = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            continue
        if k.endswith(".weight_v"):
            g = sd[k[:-1] + "g"]
            w[k[:-2]] = v * (g / v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape))
        else:
            w[k] = v
    return w


def logmel(s):
    spec = torch.view_as_real(torch.stft(s.squeeze(1), 1024, 256, 1024, win, center=True, return_complex=True)).transpose(1, 2)
    amp = torch.sqrt(torch.clamp(spec[..., 0] ** 2 + spec[..., 1] ** 2, min=1e-10))
    return torch.log(torch.clamp(amp @ melmat, min=1e-10))


def step():
    y_ = O.generator_forward(fold(G), gp, x, ar)
    gen = 45.0 * torch.nn.functional.l1_loss(logmel(y_), logmel(y))
    dw = fold(D)
    p_ = DO.disc_forward(dw, dp, torch.cat([ar, y_], 2))
    with torch.no_grad():
        p = DO.disc_forward(dw, dp, torch.cat([ar, y], 2))
    gen = gen + DO.gen_adv_loss(p_, False) + 2.0 * DO.feat_match_loss(p_, p, False, False, False)
    og.zero_grad(set_to_none=True)
    od.zero_grad(set_to_none=True)
    gen.backward()
    og.step()
    with torch.no_grad():
        y_ = O.generator_forward(fold(G), gp, x, ar)
    dw = fold(D)
    r, f = DO.dis_adv_loss(DO.disc_forward(dw, dp, torch.cat([ar, y_], 2)), DO.disc_forward(dw, dp, torch.cat([ar, y], 2)), False)
    od.zero_grad(set_to_none=True)
    (r + f).backward()
    od.step()
    return gen


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
print(f"the same iteration as PyTorch-ROCm ops: {(time.perf_counter() - t0) / a.steps * 1e3:.2f} ms")
