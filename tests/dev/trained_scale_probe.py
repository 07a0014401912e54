#!/usr/bin/env python3
"""Dev probe (GPU box): which synthetic-weight scalings put the waveform near full scale (as trained HiFi-GAN checkpoints do),
and how far each arithmetic is from the CPU oracle there, through the full 80-step AR loop.
   python tests/dev/trained_scale_probe.py [B]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from articulatory_amd.models import HiFiGANGenerator  # noqa: E402
from articulatory_amd.utils.synth import synth_features, synth_state_dict  # noqa: E402
from oracle import hificar_oracle as O  # noqa: E402
from tests.conftest import E2W_PARAMS  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.set_num_threads(16)
x = synth_features(B, 2000, 13, seed=20260929 + 3)
for gain, out_gain in ((1.0, 1.0), (1.0, 4.0), (1.0, 6.0), (1.0, 8.0), (1.0, 12.0), (1.15, 1.0), (1.3, 1.0), (1.15, 3.0)):
    sd = synth_state_dict(E2W_PARAMS, seed=1234, gain=gain)
    sd["output_conv.1.weight_g"] = sd["output_conv.1.weight_g"] * np.float32(out_gain)
    w = O.fold_weight_norm(sd)
    with torch.no_grad():
        ref = O.ar_loop_batched(w, E2W_PARAMS, torch.from_numpy(x), 2000, 80)
    res = []
    for prec in ("f32", "bf16x3"):
        g = HiFiGANGenerator(**E2W_PARAMS, precision=prec)
        g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        g.remove_weight_norm()
        g = g.eval().cuda()
        with torch.no_grad():
            y = g.ar_synthesis(torch.from_numpy(x).permute(0, 2, 1).contiguous().cuda(), 25).cpu()
        e = (y - ref).abs()
        per_utt = (e.amax(dim=1) / ref.abs().amax(dim=1)).max()
        # error growth along the AR loop: first chunk vs last chunk
        res.append(f"{prec}: all {float(e.max() / ref.abs().max()):.2e} per-utt {float(per_utt):.2e} "
                   f"chunk0 {float(e[:, :2000].max()):.2e} chunk79 {float(e[:, -2000:].max()):.2e}")
    print(f"gain {gain} out_gain {out_gain}: peak {float(ref.abs().max()):.3f} rms {float(ref.pow(2).mean().sqrt()):.3f} "
          f"|y|>0.5: {float((ref.abs() > 0.5).float().mean()):.3f}   " + "   ".join(res), flush=True)
