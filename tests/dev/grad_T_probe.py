#!/usr/bin/env python3
"""Development aid: gradient error of one fuzz configuration (tests/test_gpu_train_fuzz.py) against the float64 oracle as a function of the
number of frames — is an error tied to launch lengths that are not a whole bucket of 32 frames?   python tests/dev/grad_T_probe.py 326 64 96 107 128"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_gpu_train_fuzz as TF  # noqa: E402
from articulatory_amd.models import HiFiGANGenerator  # noqa: E402
from articulatory_amd.utils.synth import synth_features, synth_state_dict, uniform  # noqa: E402
from conftest import rel_err  # noqa: E402
from oracle import hificar_oracle as O  # noqa: E402

case = int(sys.argv[1])
rng = np.random.default_rng(31000 + case)
params, cf = TF.draw(rng)
if case % 2 == 0:
    params["nonlinear_activation_params"] = {"negative_slope": 1.0}
hop = int(np.prod(params["upsample_scales"]))
sd = synth_state_dict(params, seed=700 + case)
B = int(rng.integers(1, 5)) if rng.integers(0, 4) else int(rng.integers(5, 24))
print({k: params[k] for k in ("channels", "kernel_size", "upsample_scales", "resblock_kernel_sizes", "resblock_dilations", "use_ar", "bias", "use_weight_norm", "use_tanh")}, "B", B)
torch.set_num_threads(min(16, os.cpu_count() or 1))
for T in [int(a) for a in sys.argv[2:]]:
    for attempt in range(3):
        c_np = synth_features(B, T, cf, seed=case + 1000 * attempt).transpose(0, 2, 1).copy()
        ar_np = ((synth_features(B, 512, 1, seed=case + 1 + 1000 * attempt)[:, :, 0] * 0.4).reshape(B, 1, 512).astype(np.float32) if params["use_ar"] else None)
        margin = TF.kink_margin(sd, params, c_np, ar_np)
        g = HiFiGANGenerator(**params, precision="f32")
        g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        g = g.train().cuda()
        cot = uniform(case, "cot", (B, 1, hop * T), -1.0, 1.0)
        c = torch.from_numpy(c_np).cuda().requires_grad_(True)
        ar = torch.from_numpy(ar_np).cuda().requires_grad_(True) if ar_np is not None else None
        y = g(c, ar=ar)
        (y * torch.from_numpy(cot).cuda()).sum().backward()
        out64, ref64 = O.gradients(sd, params, c_np, ar_np, cot, dtype=torch.float64)
        got = {k: p.grad for k, p in g.named_parameters()}
        got["c"] = c.grad
        if ar is not None:
            got["ar"] = ar.grad
        errs = {k: rel_err(got[k].cpu().numpy(), ref64[k].numpy()) for k in ref64}
        v = np.array(list(errs.values()))
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
        print(f"T {T:4d} attempt {attempt}: kink margin {margin:.1e}  out err {rel_err(y.detach().cpu().numpy(), out64.numpy()):.1e}  "
              f"gradient err median {np.median(v):.1e} max {v.max():.1e}  {[(k, float('%.1e' % e)) for k, e in worst]}")
