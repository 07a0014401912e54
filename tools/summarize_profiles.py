#!/usr/bin/env python3
"""Copy rocprofv3 outputs of tools/collect_profiles.sh from gpurun_out/ into profiles/ as small committed summaries.
   python tools/summarize_profiles.py <tag>          (reads gpurun_out/prof_<tag>_<precision>_{stats,FETCH_SIZE,WRITE_SIZE,mfma})
Writes per precision: profiles/<tag>_<prec>_kernel_stats.csv, _pmc_hbm.csv, _pmc_mfma.csv, _bench.json, and updates
profiles/hbm_traffic.json (HBM bytes per launch that bench.py reports as roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import sys


def find(path, pattern):
    hits = sorted(glob.glob(os.path.join(path, "**", pattern), recursive=True), key=os.path.getmtime)
    assert hits, (path, pattern)
    return hits[-1]  # gpurun merges a new run's files into the old directories: the newest is this run's


def short(k):
    return k.replace("void hificar::", "").replace("hificar::", "").split("(")[0].replace(", ", ",")


def counters(path):
    """{kernel: {counter: [n, sum]}} and {kernel: [n, total duration ns]} of one PMC pass"""
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    seen = set()
    for row in csv.DictReader(open(find(path, "*counter_collection.csv"))):
        k = row["Kernel_Name"]
        a = agg[k][row["Counter_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
        if row["Dispatch_Id"] not in seen:
            seen.add(row["Dispatch_Id"])
            dur[k][0] += 1
            dur[k][1] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    return agg, dur


tag = sys.argv[1]
try:
    allt = json.load(open("profiles/hbm_traffic.json"))
except Exception:
    allt = {}
import datetime
import subprocess


def stamp(key):
    """which library the PMC passes of `key` ran on: the commit checked out when the profiles are summarised (+ "-dirty": uncommitted changes)"""
    try:
        head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
        dirty = subprocess.run(["git", "status", "--porcelain", "--", "articulatory_amd", "bench.py", "tools"], capture_output=True, text=True).stdout.strip()
        head += "-dirty" if dirty else ""
    except Exception:
        head = "unknown"
    allt.setdefault("_collected", {})[key] = {"commit": head, "date": datetime.date.today().isoformat(), "tag": tag}


# (file prefix / directory key, key in hbm_traffic.json, the profiled command): the two arithmetics of the headline leg, then the secondary legs
LEGS = [("f32", "f32", "python bench.py --precision f32 --steps 3 --warmup 1 --no-cpu-baseline --no-fast-leg --no-batch-sweep --no-training --no-nonar --no-gblock"),
        ("bf16x3", "bf16x3", "python bench.py --precision bf16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-fast-leg --no-batch-sweep --no-training --no-nonar --no-gblock"),
        ("nonar", "nonar_f32", "python tools/leg_bench.py --leg nonar --steps 3 --warmup 1  (BASELINE config 2: non-AR 12-dim, batch 8)"),
        ("gblock", "gblock_f32", "python tools/leg_bench.py --leg gblock --steps 3 --warmup 1  (GBlockGenerator, batch 64, chunk 25)")]
for prec, tkey, cmd in LEGS:
    base = f"gpurun_out/prof_{tag}_{prec}"
    if not os.path.isdir(base + "_stats"):
        continue
    rows = list(csv.DictReader(open(find(base + "_stats", "*kernel_stats.csv"))))
    with open(f"profiles/{tag}_{prec}_kernel_stats.csv", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -f csv -- {cmd}  ({tag}, {prec}); durations in ns\n")
        w = csv.writer(f)
        w.writerow(rows[0].keys())
        for r in rows:
            if float(r.get("Percentage", 0) or 0) >= 0.01 or "hificar" in r["Name"]:
                w.writerow(r.values())
    fe, _ = counters(base + "_FETCH_SIZE")
    wr, _ = counters(base + "_WRITE_SIZE")
    traffic = {}
    with open(f"profiles/{tag}_{prec}_pmc_hbm.csv", "w") as f:
        f.write(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- {cmd.replace('--steps 3 --warmup 1', '--steps 1 --warmup 0')}"
                "\n# counter units KiB; gfx950 correction "
                "(MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of wide coalesced reads) applied in the last column = (2*FETCH + WRITE)*1024\n")
        f.write("Kernel,Launches,FETCH_SIZE_KiB_per_launch_raw,WRITE_SIZE_KiB_per_launch,HBM_bytes_per_launch_corrected\n")
        for k in fe:
            n, v = fe[k]["FETCH_SIZE"]
            n2, v2 = wr.get(k, {}).get("WRITE_SIZE", [1, 0.0])
            b = (2 * v / n + v2 / max(n2, 1)) * 1024
            if "hificar" in k:
                f.write(f"\"{k}\",{n},{v / n:.1f},{v2 / max(n2, 1):.1f},{b:.0f}\n")
                traffic[short(k)] = round(b)
    allt[tkey] = traffic
    stamp(tkey)
    mf, dur = counters(base + "_mfma")
    with open(f"profiles/{tag}_{prec}_pmc_mfma.csv", "w") as f:
        f.write(f"# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace "
                f"-- {cmd.replace('--steps 3 --warmup 1', '--steps 1 --warmup 0')} ({tag}); per-launch averages.\n"
                "# Counter values are sums over the 8 XCDs.  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 1024 SIMDs): the share of\n"
                "# SIMD-cycles in which the matrix pipe is busy (rocprofv3's MfmaUtil expression; gfx950 has no derived-counter section,\n"
                "# MI355X_MICROARCH.md).  SQ_VALU_MFMA_BUSY_CYCLES / 1024 equals 64 x (fp32 MFMAs per SIMD) resp. 32 x (bf16 MFMAs) exactly.\n"
                "# gui_cycles_per_us = GRBM_GUI_ACTIVE / 8 / duration (the graphics clock the counter ticks at, in MHz).\n")
        f.write("Kernel,Launches,avg_duration_us,GRBM_GUI_ACTIVE,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,mfma_util,gui_cycles_per_us\n")
        for k in sorted(mf, key=lambda k: -dur[k][1]):
            if "hificar" not in k:
                continue
            c = {name: v[1] / max(v[0], 1) for name, v in mf[k].items()}
            d_us = dur[k][1] / dur[k][0] / 1e3
            gui = c.get("GRBM_GUI_ACTIVE", 0.0)
            util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024) if gui else 0.0
            f.write(f"\"{k}\",{dur[k][0]},{d_us:.2f},{gui:.0f},{c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.0f},{c.get('SQ_BUSY_CYCLES', 0):.0f},"
                    f"{c.get('SQ_WAVE_CYCLES', 0):.0f},{c.get('SQ_WAIT_INST_ANY', 0):.0f},{c.get('SQ_ACTIVE_INST_ANY', 0):.0f},{util:.4f},{gui / 8 / d_us if d_us else 0:.0f}\n")
    bj = f"{base}_bench_under_rocprof.json"
    if os.path.exists(bj):
        open(f"profiles/{tag}_{prec}_bench_under_rocprof.json", "w").write(open(bj).read())
    print(open(f"profiles/{tag}_{prec}_kernel_stats.csv").read()[:1500])
    print(open(f"profiles/{tag}_{prec}_pmc_mfma.csv").read()[:2500])
    print(traffic)
for what, cmd in (("train", "python tools/train_bench.py --steps 5"), ("gan", "python tools/gan_bench.py --steps 5")):
    base = f"gpurun_out/prof_{tag}_{what}"
    if not os.path.isdir(base + "_stats"):
        continue
    rows = list(csv.DictReader(open(find(base + "_stats", "*kernel_stats.csv"))))
    with open(f"profiles/{tag}_{what}_kernel_stats.csv", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -f csv -- {cmd}  ({tag}; 3 warm-up + 5 timed iterations, torch kernels included); durations in ns\n")
        w = csv.writer(f)
        w.writerow(rows[0].keys())
        for r in rows:
            if float(r.get("Percentage", 0) or 0) >= 0.05:
                w.writerow(r.values())
    if os.path.isdir(base + "_mfma"):
        mf, dur = counters(base + "_mfma")
        with open(f"profiles/{tag}_{what}_pmc_mfma.csv", "w") as f:
            f.write(f"# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace "
                    f"-- {cmd.replace('--steps 5', '--steps 1')} ({tag}; the GAN pass with HIFICAR_DISC_STREAMS=0 so that launches do not overlap); "
                    "per-launch averages, columns as in the forward-path files\n")
            f.write("Kernel,Launches,avg_duration_us,GRBM_GUI_ACTIVE,SQ_VALU_MFMA_BUSY_CYCLES,mfma_util\n")
            for k in sorted(mf, key=lambda k: -dur[k][1]):
                if "hificar" not in k:
                    continue
                c = {name: v[1] / max(v[0], 1) for name, v in mf[k].items()}
                gui = c.get("GRBM_GUI_ACTIVE", 0.0)
                util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024) if gui else 0.0
                f.write(f"\"{k}\",{dur[k][0]},{dur[k][1] / dur[k][0] / 1e3:.2f},{gui:.0f},{c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.0f},{util:.4f}\n")
    if os.path.isdir(base + "_FETCH_SIZE"):  # round 4: HBM bytes per launch of the training kernels (serial GAN iteration, 3 warm-up + 1 timed)
        fe, _ = counters(base + "_FETCH_SIZE")
        wr, _ = counters(base + "_WRITE_SIZE")
        train = {}
        by_base = collections.defaultdict(lambda: [0, 0.0])
        with open(f"profiles/{tag}_{what}_pmc_hbm.csv", "w") as f:
            f.write(f"# HIFICAR_DISC_STREAMS=0 rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- {cmd.replace('--steps 5', '--steps 1')}\n"
                    "# counter units KiB; gfx950 correction (FETCH_SIZE reports 1/2 of wide coalesced reads) in the last column = (2*FETCH + WRITE)*1024; "
                    "averages over ALL launches of a kernel name in the run (many different layers share a name)\n")
            f.write("Kernel,Launches,FETCH_SIZE_KiB_per_launch_raw,WRITE_SIZE_KiB_per_launch,HBM_bytes_per_launch_corrected\n")
            for k in sorted(fe, key=lambda k: -fe[k]["FETCH_SIZE"][1]):
                if "hificar" not in k:
                    continue
                n, v = fe[k]["FETCH_SIZE"]
                n2, v2 = wr.get(k, {}).get("WRITE_SIZE", [1, 0.0])
                b = (2 * v / n + v2 / max(n2, 1)) * 1024
                f.write(f"\"{k}\",{n},{v / n:.1f},{v2 / max(n2, 1):.1f},{b:.0f}\n")
                train[short(k)] = round(b)
                import re

                # the library's event profile names kernels by FAMILY: no template arguments, and col2im_mask4_kernel / im2col4_kernel under
                # their scalar siblings' names — launch-weighted averages over the family
                for fam in {short(k).split("<")[0], re.sub(r"\d+_kernel$", "_kernel", short(k).split("<")[0])}:
                    a = by_base[fam]
                    a[0] += n
                    a[1] += b * n
        for k, (n, tot) in by_base.items():
            if "<" not in k:
                train[k] = round(tot / n)
        allt["train_" + what] = train
        stamp("train_" + what)
    txt = f"{base}_bench.txt"
    if os.path.exists(txt):
        lines = [ln for ln in open(txt).read().splitlines() if "ms" in ln and ("step" in ln or "iteration" in ln)]
        open(f"profiles/{tag}_{what}_bench_under_rocprof.txt", "w").write("\n".join(lines) + "\n")
allt["_comment"] = ("HBM bytes per launch from rocprofv3 PMC, (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction), "
                    "see profiles/*_pmc_hbm.csv; keyed by precision then kernel name as bench.py reports it")
json.dump(allt, open("profiles/hbm_traffic.json", "w"), indent=1, sort_keys=True)
