import sys, time, os, threading
sys.path.insert(0, os.getcwd())
import torch
from articulatory_amd.models import HiFiGANGenerator
from articulatory_amd.utils.synth import synth_features, synth_state_dict
from bench import CAR_PARAMS
params = dict(CAR_PARAMS)
sd = synth_state_dict(params, seed=1234)
def mk():
    g = HiFiGANGenerator(**params, precision="bf16x3")
    g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    g.remove_weight_norm()
    return g.eval().cuda()
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
gs = [mk() for _ in range(nsplit)]
B, T = 64, 2000
x = torch.from_numpy(synth_features(B, T, 13, seed=1)).permute(0, 2, 1).contiguous().cuda()
parts = [p.contiguous() for p in x.chunk(nsplit, dim=0)]
streams = [torch.cuda.Stream() for _ in range(nsplit)]
def work(i, reps):
    with torch.no_grad(), torch.cuda.stream(streams[i]):
        for _ in range(reps):
            gs[i].ar_synthesis(parts[i], 25)
def run(reps):
    th = [threading.Thread(target=work, args=(i, reps)) for i in range(nsplit)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
run(1)
t0 = time.perf_counter(); run(3); dt = (time.perf_counter() - t0) / 3
print(f"nsplit={nsplit} SMALL_LDS={os.environ.get('HIFICAR_SMALL_LDS','0')}: {dt*1e3:.1f} ms/step  {B*T*80/dt/1e6:.1f} M samples/s")
