#!/usr/bin/env python3
"""Golden vectors of the REAL reference's period discriminators with SPECTRAL norm (HiFiGANMultiPeriodDiscriminator with
discriminator_params use_spectral_norm=True, use_weight_norm=False — articulatory/models/hifigan.py:390-399, 440-448: torch.nn.utils.spectral_norm,
one power iteration per forward in training mode): layer outputs of TWO consecutive training-mode forwards (the power iteration advances between
them), the gradients of the second (weight_orig, bias, input) and the weight_u / weight_v buffers after each.  Same rules as oracle/make_golden.py.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_disc_sn.py
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

from make_golden import import_reference  # noqa: E402

SN_PERIOD = {"in_channels": 1, "out_channels": 1, "kernel_sizes": [5, 3], "channels": 8, "downsample_scales": [3, 3, 1], "max_downsample_channels": 64,
             "bias": True, "nonlinear_activation": "LeakyReLU", "nonlinear_activation_params": {"negative_slope": 0.1}, "use_weight_norm": False,
             "use_spectral_norm": True}
PERIODS = [2, 3, 5]


def main():
    import torch

    from articulatory_amd.utils.synth import synth_disc_state_dict, uniform

    torch.manual_seed(0)
    ref_models, _, _ = import_reference()
    params = dict(scales=0, periods=PERIODS, period_discriminator_params=SN_PERIOD)
    B, T = 3, 1031
    for seed in range(950, 990):
        sd = {k[4:]: v for k, v in synth_disc_state_dict(params, seed=seed).items()}  # "mpd." prefix off: the stand-alone class's keys
        x_np = uniform(seed, "x", (B, 1, T), -0.6, 0.6)
        res = {}
        for dtype in (torch.float32, torch.float64):
            D = ref_models.HiFiGANMultiPeriodDiscriminator(periods=PERIODS, discriminator_params=SN_PERIOD).to(dtype)
            assert list(D.state_dict().keys()) == list(sd.keys())
            D.load_state_dict({k: torch.from_numpy(v).to(dtype) for k, v in sd.items()})
            D.train()
            x = torch.from_numpy(x_np).to(dtype).requires_grad_(True)
            outs1 = [[t.detach().clone() for t in o] for o in D(x)]
            state1 = {k: v.detach().clone() for k, v in D.state_dict().items() if k.endswith(("weight_u", "weight_v"))}
            outs = D(x)
            loss, cots = 0.0, []
            for i, o in enumerate(outs):
                cots.append([])
                for l, t in enumerate(o):
                    c = uniform(seed, f"cot.{i}.{l}", tuple(t.shape), -1.0, 1.0) / np.sqrt(t[0].numel())
                    cots[-1].append(c)
                    loss = loss + (t * torch.from_numpy(c).to(dtype)).sum()
            loss.backward()
            g = {k: p.grad.detach().double().numpy() for k, p in D.named_parameters()}
            g["x"] = x.grad.detach().double().numpy()
            state2 = {k: v.detach().clone() for k, v in D.state_dict().items() if k.endswith(("weight_u", "weight_v"))}
            res[dtype] = (outs1, outs, g, state1, state2)
        worst = max(np.abs(res[torch.float32][2][k] - res[torch.float64][2][k]).max() / max(np.abs(res[torch.float64][2][k]).max(), 1e-30)
                    for k in res[torch.float64][2])
        print(f"seed {seed}: fp32-vs-fp64 reference gradients differ by {worst:.2e}")
        if worst < 1e-4:
            break
    else:
        raise SystemExit("no kink-free seed found")
    outs1, outs2, grads, state1, state2 = res[torch.float32]
    out = {"seed": np.array(seed), "B": np.array(B), "T": np.array(T), "keys": np.array(list(sd.keys()))}
    for tag, outs in (("out1", outs1), ("out2", outs2)):
        for i, o in enumerate(outs):
            for l, t in enumerate(o):
                out[f"{tag}::{i}.{l}"] = t.detach().numpy()
    for k, v in grads.items():
        out["grad::" + k] = v.astype(np.float32)
    for tag, st in (("state1", state1), ("state2", state2)):
        for k, v in st.items():
            out[f"{tag}::{k}"] = v.numpy()
    path = os.path.join(REPO, "tests", "golden", "gold_disc_sn.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
