#!/usr/bin/env python3
"""GPU idle time from a rocprofv3 --kernel-trace CSV: union of the kernel intervals (streams overlap) vs the span, and the biggest gaps with
the kernels on either side.   python tools/trace_gaps.py <kernel_trace.csv> [skip_fraction] [context]
context > 0: the `context` kernels in front of and behind each of the ten largest gaps, by name (who was the host waiting for?)."""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void hificar::", "").replace("void at::native::", "at::").replace("(anonymous namespace)::", "").split("(")[0][:110]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]  # the steady state: the later part of the run
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
last_name = rows[0][2]
where = {}
for idx, (s, e, n) in enumerate(rows[1:], 1):
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, n))
        where[(s - cur_e, last_name, n)] = idx
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    last_name = n if e >= cur_e else last_name
busy += cur_e - cur_s
print(f"{len(rows)} kernels, span {span / 1e6:.2f} ms, GPU busy (union) {busy / 1e6:.2f} ms = {busy / span:.3f}, idle {(span - busy) / 1e6:.2f} ms in {len(gaps)} gaps")
big = sorted(gaps, reverse=True)[:25]
for g, a, b in big:
    print(f"  {g / 1e3:8.1f} us  after {a:45s} before {b}")
ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if ctx:
    for gap in big[:10]:
        i = where[gap]
        print(f"--- gap of {gap[0] / 1e3:.0f} us; before it:")
        for s_, e_, n_ in rows[max(0, i - ctx):i]:
            print(f"      {(e_ - s_) / 1e3:8.1f} us  {n_}")
        print("    after it:")
        for s_, e_, n_ in rows[i:i + ctx]:
            print(f"      {(e_ - s_) / 1e3:8.1f} us  {n_}")
hist = {}
for g, a, b in gaps:
    k = "<5us" if g < 5000 else "<20us" if g < 20000 else "<100us" if g < 100000 else ">=100us"
    hist[k] = hist.get(k, 0) + g
print({k: round(v / 1e6, 2) for k, v in hist.items()}, "ms")
