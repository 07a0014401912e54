#!/usr/bin/env python3
"""Development aid: signal gradient of sub-discriminator subsets, native vs oracle."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from articulatory_amd import losses as NL  # noqa: E402
from articulatory_amd.models import HiFiGANMultiScaleMultiPeriodDiscriminator  # noqa: E402
from articulatory_amd.utils.synth import synth_disc_state_dict, uniform  # noqa: E402
from oracle import disc_oracle as DO  # noqa: E402
from oracle.make_golden_disc import SMALL  # noqa: E402

B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 700
for tag, over in [("scale0", dict(scales=1, periods=[])), ("scales", dict(scales=2, periods=[])), ("p2", dict(scales=0, periods=[2])),
                  ("p3", dict(scales=0, periods=[3])), ("p5", dict(scales=0, periods=[5]))]:
    params = dict(SMALL, **over)
    sd = synth_disc_state_dict(params, seed=77)
    d = HiFiGANMultiScaleMultiPeriodDiscriminator(**params)
    d.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    d = d.cuda()
    x_np = uniform(5, "real", (B, 1, T), -0.5, 0.5)
    xh_np = uniform(5, "fake", (B, 1, T), -0.5, 0.5)
    for mode in ("adv", "fm"):
        xh = torch.from_numpy(xh_np).cuda().requires_grad_(True)
        fake = d(xh, native=True)
        with torch.no_grad():
            real = d(torch.from_numpy(x_np).cuda(), native=True)
        loss = NL.generator_adversarial_loss(fake, False) if mode == "adv" else NL.feature_match_loss(fake, real, False, False, False)
        loss.backward()
        w = DO.fold_disc_weight_norm(sd)
        xr = torch.from_numpy(xh_np).requires_grad_(True)
        f_ref = DO.disc_forward(w, params, xr)
        with torch.no_grad():
            r_ref = DO.disc_forward(w, params, torch.from_numpy(x_np))
        lr = DO.gen_adv_loss(f_ref, False) if mode == "adv" else DO.feat_match_loss(f_ref, r_ref, False, False, False)
        lr.backward()
        g, gr = xh.grad.cpu().numpy(), xr.grad.numpy()
        err = np.abs(g - gr)
        where = np.argwhere(err > 1e-3 * np.abs(gr).max())
        print(f"{tag:7s} {mode}: loss {float(loss):.6f} / {float(lr):.6f}  max err {err.max() / np.abs(gr).max():.2e}  bad at {where[:6].tolist()}")
