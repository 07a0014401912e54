#!/usr/bin/env python3
"""Development aid for tests/test_gpu_train_fuzz.py: does a failing EVEN case (generator slope 1: the only kinks left are the output conv's
LeakyReLU(0.01) and the AR encoder's) fail because the fp32 device and the float64 oracle sit on different sides of a kink?  Reproduces the
case's inputs, taps the last stage's ResBlock outputs on the device, forms the MRF mean in the kernel's order and compares its signs with
the oracle's output-conv input.   python tests/dev/kink_flip_probe.py 326 [136 ...]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_gpu_train_fuzz as TF  # noqa: E402
from articulatory_amd.models import HiFiGANGenerator  # noqa: E402
from articulatory_amd.utils.synth import synth_features, synth_state_dict  # noqa: E402
from oracle import hificar_oracle as O  # noqa: E402

for case in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(31000 + case)
    params, cf = TF.draw(rng)
    assert case % 2 == 0, "even cases only"
    params["nonlinear_activation_params"] = {"negative_slope": 1.0}
    sd = synth_state_dict(params, seed=700 + case)
    B = int(rng.integers(1, 5)) if rng.integers(0, 4) else int(rng.integers(5, 24))
    T = int(rng.integers(2, 30)) if rng.integers(0, 4) else int(rng.integers(30, 120))
    for attempt in range(6):
        c_np = synth_features(B, T, cf, seed=case + 1000 * attempt).transpose(0, 2, 1).copy()
        ar_np = ((synth_features(B, 512, 1, seed=case + 1 + 1000 * attempt)[:, :, 0] * 0.4).reshape(B, 1, 512).astype(np.float32)
                 if params["use_ar"] else None)
        if TF.kink_margin(sd, params, c_np, ar_np) > 2e-6:
            break
    g = HiFiGANGenerator(**params, precision="f32")
    g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    g = g.eval().cuda()
    n_stages, n_blocks = len(params["upsample_scales"]), len(params["resblock_kernel_sizes"])
    names = [f"blocks.{(n_stages - 1) * n_blocks + j}" for j in range(n_blocks)]
    with torch.no_grad():
        _, taps = g.debug_taps(names, torch.from_numpy(c_np).cuda(), ar=torch.from_numpy(ar_np).cuda() if ar_np is not None else None)
    bs = [taps[n].cpu().numpy().astype(np.float32) for n in names]
    m32 = bs[0]
    if n_blocks == 2:
        m32 = (bs[0] + bs[1]) / np.float32(2.0)
    elif n_blocks == 3:
        m32 = ((bs[0] + bs[1]) + bs[2]) / np.float32(3.0)
    seen = []
    real = O.F.leaky_relu

    def spy(x, negative_slope=0.01, *a, **kw):
        if negative_slope == 0.01:
            seen.append(x.detach().clone())
        return real(x, negative_slope, *a, **kw)

    O.F.leaky_relu = spy
    try:
        with torch.no_grad():
            O.generator_forward(O.fold_weight_norm(sd, dtype=torch.float64), params, torch.from_numpy(c_np).double(),
                                torch.from_numpy(ar_np).double() if ar_np is not None else None)
    finally:
        O.F.leaky_relu = real
    m64 = seen[-1].numpy()
    assert m64.shape == m32.shape, (m64.shape, m32.shape)
    flips = np.argwhere(np.sign(m32) != np.sign(m64))
    print(f"case {case}: B {B} T {T} attempt {attempt}, output-conv input {m64.shape}: min |m64| / max = {np.abs(m64).min() / np.abs(m64).max():.2e}, "
          f"{len(flips)} elements on the other side of the kink in fp32")
    for f in flips[:5]:
        print("   ", tuple(f), float(m64[tuple(f)]), float(m32[tuple(f)]))
