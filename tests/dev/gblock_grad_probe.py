"""Dev probe: per-tensor gradient error of a GBlockGenerator fuzz case against the float64 oracle.  python tests/dev/gblock_grad_probe.py <case> [k=..] [s=..]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from test_gpu_gblock import _draw, build  # noqa: E402
from articulatory_amd.utils.synth import synth_features, uniform  # noqa: E402
from oracle import gblock_oracle as G  # noqa: E402

case = int(sys.argv[1])
rng = np.random.default_rng(5000 + case)
p, cf = _draw(rng)
p["channels"] = int(rng.choice([24, 64, 100]))
for a in [x for x in sys.argv[2:] if "=" in x]:
    k, v = a.split("=")
    if k == "k":
        p["g_kernel_sizes"] = [int(v)] * len(p["g_scales"])
    elif k == "s":
        p["g_scales"] = [int(x) for x in v.split(",")]
    elif k == "ch":
        p["channels"] = int(v)
    elif k == "ks":
        p["g_kernel_sizes"] = [int(x) for x in v.split(",")]
hop = int(np.prod(p["g_scales"]))
print(p, "hop", hop)
model, sd = build(p, seed=600 + case, train=True)
B, T = int(rng.integers(1, 3)), (int(rng.integers(1, 4)) if hop > 40 else int(rng.integers(2, 12)))
for a in [x for x in sys.argv[2:] if x.startswith(("T=", "B="))]:
    if a[0] == "T":
        T = int(a[2:])
    else:
        B = int(a[2:])
spk = rng.integers(0, 5, size=B) if p["use_spk_id"] else None
c_np = synth_features(B, T, cf, seed=9000 + 13 * case).transpose(0, 2, 1).copy()
ar_np = (synth_features(B, 512, 1, seed=9500 + 13 * case)[:, :, 0] * 0.3).reshape(B, 1, 512).astype(np.float32) if p["use_ar"] else None
cot = uniform(700 + case, "cotangent", (B, 1, hop * T), -1.0, 1.0)
c = torch.from_numpy(c_np).cuda().requires_grad_(True)
ar = torch.from_numpy(ar_np).cuda().requires_grad_(True) if ar_np is not None else None
y = model(c, ar=ar, spk_id=torch.from_numpy(spk).cuda() if spk is not None else None)
(y * torch.from_numpy(cot).cuda()).sum().backward()
out64, ref = G.gradients(sd, p, c_np, ar_np, cot, dtype=torch.float64, spk_id=spk)
bad = [k for k in ref if float(np.abs(got[k].cpu().numpy().astype(np.float64) - ref[k].numpy()).max() / max(np.abs(ref[k].numpy()).max(), 1e-30)) > 2e-4] if False else None
print("B", B, "T", T, "forward err", float((y.detach().cpu().double() - out64).abs().max() / out64.abs().max()))
got = {k: q.grad for k, q in model.named_parameters()}
got["c"] = c.grad
def _e(k):
    r = ref[k].numpy()
    return float(np.abs(got[k].cpu().numpy().astype(np.float64) - r).max() / max(np.abs(r).max(), 1e-30))
if ar is not None:
    got["ar"] = ar.grad
for k in ref:
    r = ref[k].numpy()
    e = float(np.abs(got[k].cpu().numpy().astype(np.float64) - r).max() / max(np.abs(r).max(), 1e-30))
    if (e > 1e-4 and "-q" not in sys.argv) or "-v" in sys.argv:
        print(f"  {k:40s} {e:.2e}  shape {tuple(r.shape)}")

worst = sorted(((k, _e(k)) for k in ref), key=lambda kv: -kv[1])
print("worst:", worst[:3], " first bad from the output side:", next((k for k in reversed(list(ref)) if _e(k) > 2e-4), None))
