#!/usr/bin/env python3
"""Copy rocprofv3 outputs from gpurun_out/ into profiles/ as small committed summaries.
   python tools/summarize_profiles.py <tag> <stats_csv> <pmc_fetch_dir> <pmc_write_dir> <bench_json> [precision]"""
import collections
import csv
import glob
import json
import os
import sys


def find(path, pattern):
    """path itself if it is a file, else the single rocprofv3 output matching pattern below it"""
    if os.path.isfile(path):
        return path
    hits = sorted(glob.glob(os.path.join(path, "**", pattern), recursive=True))
    assert hits, (path, pattern)
    return hits[0]


tag, stats_csv, fdir, wdir, bench_json = sys.argv[1:6]
prec = sys.argv[6] if len(sys.argv) > 6 else "bf16x3"
rows = list(csv.DictReader(open(find(stats_csv, '*kernel_stats.csv'))))
with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline  ({tag}); durations in ns\n")
    w = csv.writer(f)
    w.writerow(rows[0].keys())
    for r in rows:
        if float(r.get("Percentage", 0) or 0) >= 0.01 or "hificar" in r["Name"]:
            w.writerow(r.values())
res = {}
for c, d in (("FETCH_SIZE", fdir), ("WRITE_SIZE", wdir)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(find(d, "*counter_collection.csv"))):
        if row["Counter_Name"] == c:
            agg[row["Kernel_Name"]][0] += 1
            agg[row["Kernel_Name"]][1] += float(row["Counter_Value"])
    res[c] = agg
traffic = {}
with open(f"profiles/{tag}_pmc_hbm.csv", "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 1 --warmup 0 "
            "--no-cpu-baseline --no-roofline\n# counter units KiB; gfx950 correction (MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of wide "
            "coalesced reads) applied in the last column = (2*FETCH + WRITE)*1024\n")
    f.write("Kernel,Launches,FETCH_SIZE_KiB_per_launch_raw,WRITE_SIZE_KiB_per_launch,HBM_bytes_per_launch_corrected\n")
    for k in res["FETCH_SIZE"]:
        n, fe = res["FETCH_SIZE"][k]
        n2, wr = res["WRITE_SIZE"].get(k, [1, 0.0])
        b = (2 * fe / n + wr / max(n2, 1)) * 1024
        if "hificar" in k or "elementwise" in k:
            f.write(f"\"{k}\",{n},{fe / n:.1f},{wr / max(n2, 1):.1f},{b:.0f}\n")
        if "hificar" in k:
            traffic[k.replace("void hificar::", "").replace("hificar::", "").split("(")[0].replace(", ", ",")] = round(b)
try:
    allt = json.load(open("profiles/hbm_traffic.json"))
except Exception:
    allt = {}
allt[prec] = traffic
allt["_comment"] = ("HBM bytes per launch from rocprofv3 PMC, (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction), "
                    "see profiles/*_pmc_hbm.csv; keyed by precision then kernel name as bench.py reports it")
json.dump(allt, open("profiles/hbm_traffic.json", "w"), indent=1, sort_keys=True)
open(f"profiles/{tag}_bench.json", "w").write(open(bench_json).read())
print(open(f"profiles/{tag}_kernel_stats.csv").read()[:1800])
print(traffic)
