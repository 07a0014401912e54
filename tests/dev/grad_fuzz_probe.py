import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
import test_gpu_train_fuzz as F
from conftest import rel_err
from oracle import hificar_oracle as O
from articulatory_amd.models import HiFiGANGenerator
from articulatory_amd.utils.synth import synth_features, synth_state_dict, uniform
for case in [int(a) for a in sys.argv[1:]] or (13, 17):
    rng = np.random.default_rng(31000 + case)
    params, cf = F.draw(rng)
    print(case, params["channels"], params["upsample_scales"], params["resblock_kernel_sizes"], params["resblock_dilations"], params["nonlinear_activation_params"], params["use_weight_norm"], params["bias"], params["use_ar"])
    hop = int(np.prod(params["upsample_scales"]))
    sd = synth_state_dict(params, seed=700 + case)
    B = int(rng.integers(1, 5)) if rng.integers(0, 4) else int(rng.integers(5, 24))
    T = int(rng.integers(2, 30)) if rng.integers(0, 4) else int(rng.integers(30, 120))
    print("B,T", B, T)
    for slope in (params["nonlinear_activation_params"]["negative_slope"], 1.0):
        p2 = dict(params, nonlinear_activation_params={"negative_slope": slope})
        g = HiFiGANGenerator(**p2, precision="f32")
        g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        g = g.train().cuda()
        c_np = synth_features(B, T, cf, seed=case).transpose(0, 2, 1).copy()
        ar_np = (synth_features(B, 512, 1, seed=case + 1)[:, :, 0] * 0.4).reshape(B, 1, 512).astype(np.float32) if params["use_ar"] else None
        cot = uniform(case, "cot", (B, 1, hop * T), -1.0, 1.0)
        c = torch.from_numpy(c_np).cuda().requires_grad_(True)
        ar = torch.from_numpy(ar_np).cuda().requires_grad_(True) if ar_np is not None else None
        y = g(c, ar=ar)
        (y * torch.from_numpy(cot).cuda()).sum().backward()
        out64, ref64 = O.gradients(sd, p2, c_np, ar_np, cot, dtype=torch.float64)
        got = {k: p.grad for k, p in g.named_parameters()}; got["c"] = c.grad
        if ar is not None: got["ar"] = ar.grad
        errs = {k: rel_err(got[k].cpu().numpy(), ref64[k].numpy()) for k in ref64}
        v = np.array(sorted(errs.values()))
        print(" slope", slope, "fwd err", rel_err(y.detach().cpu().numpy(), out64.numpy()), "grad err percentiles", [f"{np.percentile(v,q):.1e}" for q in (10,50,90,100)])
        for k, e in sorted(errs.items(), key=lambda kv: -kv[1])[:6]:
            a = got[k].cpu().double().numpy().reshape(-1); b = ref64[k].numpy().reshape(-1)
            d = np.abs(a-b)/np.abs(b).max()
            print(f"    {k:40s} {e:.1e}  frac>1e-4: {(d>1e-4).mean():.3f} n={d.size}")
