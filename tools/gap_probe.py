#!/usr/bin/env python3
"""Kernel durations vs inter-kernel gaps of the AR loop at small batch.
  rocprofv3 --kernel-trace -f csv -d gpurun_out/gap -- python tools/gap_probe.py run 1
  python tools/gap_probe.py report gpurun_out/gap"""
import csv, glob, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

if sys.argv[1] == "run":
    import torch
    from articulatory_amd.models import HiFiGANGenerator
    from articulatory_amd.utils.synth import synth_features, synth_state_dict
    from bench import CAR_PARAMS
    B = int(sys.argv[2])
    sd = synth_state_dict(CAR_PARAMS, seed=1234)
    g = HiFiGANGenerator(**CAR_PARAMS, precision="bf16x3")
    g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    g.remove_weight_norm(); g = g.eval().cuda()
    x = torch.from_numpy(synth_features(B, 500, 13, seed=1)).permute(0, 2, 1).contiguous().cuda()
    with torch.no_grad():
        for _ in range(3):
            g.ar_synthesis(x, 25)
        torch.cuda.synchronize()
else:
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) * 2 // 3:]  # last repetition
    dur, gap = {}, []
    for a, b in zip(rows, rows[1:]):
        gap.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
    for r in rows:
        n = r["Kernel_Name"].split("(")[0][:60]
        d = dur.setdefault(n, [0, 0]); d[0] += 1; d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
    busy = sum(v[1] for v in dur.values())
    print(f"{len(rows)} launches, span {span/1e3:.0f} us, busy {busy/1e3:.0f} us, mean gap {sum(gap)/len(gap)/1e3:.2f} us")
    for n, (c, t) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:60s} x{c:5d}  avg {t/c/1e3:7.2f} us")
