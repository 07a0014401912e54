#!/usr/bin/env python3
"""HBM traffic PER LAUNCH SHAPE (layer), not per kernel name.

rocprofv3's counter rows carry a kernel name only, and one name (conv_f32do_kernel<4,1,4,4>, wgrad_gemm_kernel ...) covers dozens of different
layers of a training iteration.  libhificar writes every launch it enqueues to $HIFICAR_LAUNCH_LOG — in enqueue order, as
"kernel|layer xN<TAB>flops<TAB>algorithmic bytes" — and rocprofv3 numbers dispatches in the same order (one host thread), so the n-th logged launch
of the process is the n-th dispatch of a hificar kernel.  This tool joins the two for a FETCH_SIZE pass and a WRITE_SIZE pass of the same command:

    HIFICAR_LAUNCH_LOG=$out/fetch.log rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $out/fetch -- python tools/gan_bench.py --steps 1
    HIFICAR_LAUNCH_LOG=$out/write.log rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $out/write -- python tools/gan_bench.py --steps 1
    python tools/pmc_by_layer.py --fetch $out/fetch --fetch-log $out/fetch.log --write $out/write --write-log $out/write.log \
           --last-iterations 1 --out profiles/r06_gan_pmc_hbm_by_layer.csv

Counter units are KiB; the corrected column applies the gfx950 rule of MI355X_MICROARCH.md (FETCH_SIZE reports half of wide coalesced reads):
HBM bytes = (2 * FETCH + WRITE) * 1024.  Dispatches of hificar kernels that the library does not log (none today) are listed in the footer.
"""
import argparse
import collections
import csv
import glob
import os
import re
import sys


def find(path, pattern):
    hits = sorted(glob.glob(os.path.join(path, "**", pattern), recursive=True), key=os.path.getmtime)
    assert hits, (path, pattern)
    return hits[-1]


def short(k):
    return k.replace("void hificar::", "").replace("hificar::", "").split("(")[0].replace(", ", ",")


def family(name):
    """kernel family of a rocprof name or a log label: no template arguments, the vectorised / GEMM-form siblings under one name"""
    base = re.split(r"[<| ]", name, 1)[0]
    base = re.sub(r"\d+_kernel$", "_kernel", base)
    return base.replace("wreduce_gemm_kernel", "wreduce_kernel")


def group_of(kernel, layer):
    """the engine / role a launch belongs to (bench.py's training leg aggregates its live event pass the same way)"""
    base = family(kernel)
    if layer.startswith(("blocks.", "upsamples.", "input_conv")):
        return "generator convs (forward x2, data gradients)"
    if "mpd." in layer:
        return "period discriminators (convs, im2col / col2im)"
    if "msd." in layer:
        return "scale discriminators (convs, im2col / col2im)"
    if base.startswith("wgrad"):
        return "weight gradients"
    return "other (reductions, packs, losses, element-wise)"


def dispatches(path, counter):
    """[(dispatch id, kernel, counter value, duration ns)] of the hificar kernels, in dispatch order"""
    rows = {}
    for r in csv.DictReader(open(find(path, "*counter_collection.csv"))):
        if r["Counter_Name"] != counter or "hificar" not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        v = rows.setdefault(d, [short(r["Kernel_Name"]), 0.0, float(r["End_Timestamp"]) - float(r["Start_Timestamp"])])
        v[1] += float(r["Counter_Value"])
    return [(d, *rows[d]) for d in sorted(rows)]


def read_log(path):
    out = []
    for ln in open(path):
        parts = ln.rstrip("\n").split("\t")
        if len(parts) == 3:
            out.append((parts[0], float(parts[1]), float(parts[2])))
    return out


def join(disp, log):
    """greedy alignment: a dispatch whose family differs from the next log entry's is an unlogged launch"""
    joined, unlogged, li = [], collections.Counter(), 0
    for d, k, v, dur in disp:
        if li < len(log) and family(log[li][0]) == family(k):
            joined.append((log[li], k, v, dur))
            li += 1
        else:
            unlogged[k] += 1
    return joined, unlogged, len(log) - li


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--fetch-log", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--write-log", required=True)
    ap.add_argument("--last-iterations", type=int, default=0,
                    help="keep only the last N iterations of the run (an iteration starts at --iteration-marker, --markers-per-iteration times)")
    ap.add_argument("--iteration-marker", default="front_kernel", help="a kernel every iteration launches first (training: the generator's front_kernel, twice per iteration)")
    ap.add_argument("--markers-per-iteration", type=int, default=2)
    ap.add_argument("--out", required=True)
    ap.add_argument("--command", default="")
    ap.add_argument("--groups-json", default=None, help="also write the per-group and per-(kernel, group) totals as JSON (bench.py's training leg reads it)")
    a = ap.parse_args()

    per = {}
    notes = []
    for what, path, logp, ctr in (("fetch", a.fetch, a.fetch_log, "FETCH_SIZE"), ("write", a.write, a.write_log, "WRITE_SIZE")):
        joined, unlogged, left = join(dispatches(path, ctr), read_log(logp))
        notes.append(f"{what}: {len(joined)} launches joined, {sum(unlogged.values())} unlogged dispatches {dict(unlogged)}, {left} log entries without a dispatch")
        if a.last_iterations > 0:
            marks = [i for i, (lg, _, _, _) in enumerate(joined) if family(lg[0]) == a.iteration_marker]
            need = a.last_iterations * a.markers_per_iteration
            if len(marks) >= need:
                joined = joined[marks[-need]:]
        agg = collections.OrderedDict()
        for (label, flops, abytes), k, v, dur in joined:
            key = (k, label)
            e = agg.setdefault(key, [0, 0.0, 0.0, 0.0, 0.0])
            e[0] += 1
            e[1] += v
            e[2] += dur
            e[3] += abytes  # (launches that share a label may differ in size: the two engines' pack_all_kernel launches)
            e[4] += flops
        per[what] = agg
    keys = list(per["fetch"].keys())
    with open(a.out, "w") as f:
        f.write(f"# per launch shape: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- {a.command}, joined with HIFICAR_LAUNCH_LOG by dispatch order\n")
        f.write("# counter units KiB per launch; HBM_bytes_corrected = (2*FETCH + WRITE)*1024 (gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads); "
                "algorithmic bytes as the library counts them for the launch (inputs + residual + outputs + weights, each once)\n")
        for n in notes:
            f.write("# " + n + "\n")
        f.write("Kernel,Layer,Launches,avg_duration_us,GFLOP_per_launch,algorithmic_MB,FETCH_KiB_raw,WRITE_KiB,HBM_MB_corrected,traffic_over_algorithmic,TFLOP_s\n")
        tot = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
        grp = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])  # launches, HBM bytes, algorithmic bytes, ns, flops
        for key in keys:
            k, label = key
            n, fv, dur, ab, fl = per["fetch"][key]
            ab, fl = ab / n, fl / n  # per launch
            wn, wv = per["write"].get(key, [1, 0.0])[:2]
            fe, wr = fv / n, wv / max(wn, 1)
            hbm = (2 * fe + wr) * 1024
            layer = label.split("|", 1)[1] if "|" in label else (label.split(" ", 1)[1] if " " in label else "")
            d_us = dur / n / 1e3
            f.write(f"\"{k}\",\"{layer}\",{n},{d_us:.2f},{fl / 1e9:.3f},{ab / 1e6:.2f},{fe:.1f},{wr:.1f},{hbm / 1e6:.2f},"
                    f"{hbm / ab if ab else 0:.2f},{fl / (d_us * 1e-6) / 1e12 if d_us else 0:.1f}\n")
            t = tot[k]
            t[0] += n
            t[1] += hbm * n
            t[2] += ab * n
            t[3] += dur
            for gk in (group_of(k, layer), family(k) + " @ " + group_of(k, layer)):
                g = grp[gk]
                g[0] += n
                g[1] += hbm * n
                g[2] += ab * n
                g[3] += dur
                g[4] += fl * n
        f.write("# per kernel name (launch-weighted): kernel, launches, HBM MB, algorithmic MB, ratio, total ms\n")
        for k, (n, hb, ab, dur) in sorted(tot.items(), key=lambda kv: -kv[1][3]):
            f.write(f"# {k},{n},{hb / n / 1e6:.2f},{ab / n / 1e6:.2f},{hb / ab if ab else 0:.2f},{dur / 1e6:.3f}\n")
    if a.groups_json:
        import json

        out = {"_comment": "HBM bytes (2*FETCH + WRITE, rocprofv3 PMC) against the library's algorithmic bytes, summed per group of launches of ONE serial GAN "
                           "iteration (tools/pmc_by_layer.sh); keys 'family @ group' restrict a kernel family to one group", "_source": os.path.basename(a.out)}
        for gk, (n, hb, ab, dur, fl) in grp.items():
            out[gk] = {"launches": n, "hbm_MB": round(hb / 1e6, 1), "algorithmic_MB": round(ab / 1e6, 1), "traffic_over_algorithmic": round(hb / ab, 3) if ab else None,
                       "ms": round(dur / 1e6, 3), "tflops": round(fl / (dur * 1e-9) / 1e12, 1) if dur else 0.0}
        json.dump(out, open(a.groups_json, "w"), indent=1, sort_keys=True)
    print(open(a.out).read()[:6000])
    for n in notes:
        print(n, file=sys.stderr)


if __name__ == "__main__":
    main()
