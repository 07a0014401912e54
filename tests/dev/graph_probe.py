#!/usr/bin/env python3
"""Development aid: does replaying the AR loop's launches from a captured hipGraph shorten the small-batch step?  Eager enqueue vs
torch.cuda.CUDAGraph replay of ``ar_synthesis`` (same launches, same buffers).  python tests/dev/graph_probe.py [batches]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from articulatory_amd.models import HiFiGANGenerator  # noqa: E402
from articulatory_amd.utils.recipes import recipe_train_config  # noqa: E402
from articulatory_amd.utils.synth import synth_features, synth_state_dict  # noqa: E402

params = recipe_train_config("car")["generator_params"]
g = HiFiGANGenerator(**params)
g.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(params, seed=1234).items()})
g.remove_weight_norm()
g = g.eval().cuda()
T, chunk = 2000, 25
for B in [int(b) for b in (sys.argv[1:] or ["1", "8", "64"])]:
    c = torch.from_numpy(synth_features(B, T, 13, seed=5).transpose(0, 2, 1).copy()).cuda()
    with torch.no_grad():
        for _ in range(2):
            ref = g.ar_synthesis(c, chunk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            ref = g.ar_synthesis(c, chunk)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n
        graph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            g.ar_synthesis(c, chunk)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.graph(graph):
            out = g.ar_synthesis(c, chunk)
        torch.cuda.synchronize()
        t_cap = time.perf_counter() - t0
        graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            graph.replay()
        torch.cuda.synchronize()
        rep = (time.perf_counter() - t0) / n
    same = bool(torch.equal(out, ref))
    print(f"batch {B}: eager {eager * 1e3:.2f} ms ({B * T * 80 / eager / 1e6:.2f} M samples/s), graph replay {rep * 1e3:.2f} ms "
          f"({B * T * 80 / rep / 1e6:.2f} M samples/s), capture {t_cap * 1e3:.0f} ms, bit-identical {same}", flush=True)
