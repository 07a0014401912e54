import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from articulatory_amd.models import HiFiGANGenerator
from articulatory_amd.utils.synth import synth_features, synth_state_dict
from bench import CAR_PARAMS
params = dict(CAR_PARAMS)
sd = synth_state_dict(params, seed=1234)
g = HiFiGANGenerator(**params, precision="bf16x3")
g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
g.remove_weight_norm(); g = g.eval().cuda()
for B in (1, 8, 64):
    x = torch.from_numpy(synth_features(B, 2000, 13, seed=1)).permute(0, 2, 1).contiguous().cuda()
    with torch.no_grad():
        g.ar_synthesis(x, 25); torch.cuda.synchronize()
        t0 = time.perf_counter(); g.ar_synthesis(x, 25); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B}: enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
