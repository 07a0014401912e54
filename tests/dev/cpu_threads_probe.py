import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from articulatory_amd.utils.synth import synth_features, synth_state_dict
from oracle import hificar_oracle as O
from bench import CAR_PARAMS
params = dict(CAR_PARAMS); sd = synth_state_dict(params, seed=1234); w = O.fold_weight_norm(sd)
for B, T in ((8, 250), (64, 50)):
    x = torch.from_numpy(synth_features(B, T, 13, seed=3))
    for nt in (128, 64, 32, 16, 8, 1):
        torch.set_num_threads(nt)
        with torch.no_grad():
            O.ar_loop_batched(w, params, x[:, :25], 2000, 80)
            t0 = time.perf_counter(); y = O.ar_loop_batched(w, params, x, 2000, 80); dt = time.perf_counter() - t0
        print(f"B={B} T={T} threads={nt}: {y.numel()/dt/1e3:.1f} k samples/s ({dt:.1f} s)", flush=True)
