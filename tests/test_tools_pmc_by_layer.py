"""tools/pmc_by_layer.py on synthetic inputs (CPU): the library's launch log and rocprofv3's per-dispatch counter rows are joined by dispatch order,
dispatches the library does not log are skipped (and counted), launches that share a label but differ in size are averaged, and the per-engine
groups that bench.py's training leg reads come out of the same join."""
import csv
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_pass(d, counter, values):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "1_counter_collection.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
        for i, (kernel, v, dur) in enumerate(values, 1):
            for xcd_share in (0.5, 0.5):  # rocprofv3 writes one row per XCD group: the tool sums them per dispatch
                w.writerow([i, kernel, counter, v * xcd_share, 1000 * i, 1000 * i + dur])
            w.writerow([i, kernel, "SOMETHING_ELSE", 7, 1000 * i, 1000 * i + dur])


def test_join_by_dispatch_order_and_groups(tmp_path):
    disp = [("void hificar::front_kernel(hificar::FrontParams)", 10.0, 100),
            ("void hificar::conv_f32do_kernel<4, 1, 4, 4>(hificar::MultiConvParams)", 1000.0, 2000),
            ("void hificar::loss_reduce_kernel(hificar::LossEntry const*)", 1.0, 10),           # not in the log: skipped, counted
            ("void hificar::conv_f32do_kernel<4, 1, 4, 4>(hificar::MultiConvParams)", 3000.0, 4000),
            ("void hificar::im2col4_kernel(hificar::Im2colParams)", 50.0, 100),                   # logged under its scalar sibling's family name
            ("void hificar::pack_all_kernel(hificar::PackParams const*, int const*, int)", 100.0, 300),
            ("void hificar::pack_all_kernel(hificar::PackParams const*, int const*, int)", 300.0, 500),
            ("at::native::something_else()", 5.0, 5)]
    log = ["front_kernel\t0\t1000", "conv_f32do_kernel<4,1,4,4>|blocks.2.convs1.0.1 x3\t2e9\t1024000",
           "conv_f32do_kernel<4,1,4,4>|mpd.discriminators.1.convs.4.0#g0 x1\t4e9\t1024000", "im2col_kernel|mpd.discriminators.1.convs.4.0\t0\t51200",
           "pack_all_kernel\t0\t102400", "pack_all_kernel\t0\t307200"]
    for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        _write_pass(str(tmp_path / name), ctr, disp)
        (tmp_path / f"{name}.log").write_text("\n".join(log) + "\n")
    out, gj = tmp_path / "t.csv", tmp_path / "g.json"
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "pmc_by_layer.py"), "--fetch", str(tmp_path / "fetch"), "--fetch-log", str(tmp_path / "fetch.log"),
                        "--write", str(tmp_path / "write"), "--write-log", str(tmp_path / "write.log"), "--out", str(out), "--groups-json", str(gj)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "6 launches joined, 1 unlogged dispatches {'loss_reduce_kernel': 1}, 0 log entries without a dispatch" in r.stderr
    rows = {(x["Kernel"], x["Layer"]): x for x in csv.DictReader(ln for ln in open(out) if not ln.startswith("#"))}
    gen = rows[("conv_f32do_kernel<4,1,4,4>", "blocks.2.convs1.0.1 x3")]
    # FETCH 1000 KiB raw, WRITE 1000 KiB -> (2 * 1000 + 1000) * 1024 bytes against 1 024 000 algorithmic
    assert float(gen["FETCH_KiB_raw"]) == 1000.0 and float(gen["HBM_MB_corrected"]) == 3.07 and float(gen["traffic_over_algorithmic"]) == 3.0
    assert float(gen["avg_duration_us"]) == 2.0 and float(gen["TFLOP_s"]) == 1000.0
    assert float(rows[("conv_f32do_kernel<4,1,4,4>", "mpd.discriminators.1.convs.4.0#g0 x1")]["traffic_over_algorithmic"]) == 9.0
    assert ("im2col4_kernel", "mpd.discriminators.1.convs.4.0") in rows
    pk = rows[("pack_all_kernel", "")]  # two launches of different size under one label: averaged per launch
    assert int(pk["Launches"]) == 2 and float(pk["algorithmic_MB"]) == 0.2 and float(pk["FETCH_KiB_raw"]) == 200.0
    g = json.load(open(gj))
    assert g["generator convs (forward x2, data gradients)"]["launches"] == 1
    assert g["period discriminators (convs, im2col / col2im)"]["launches"] == 2
    assert g["conv_f32do_kernel @ period discriminators (convs, im2col / col2im)"]["traffic_over_algorithmic"] == 9.0
    assert g["other (reductions, packs, losses, element-wise)"]["launches"] == 3  # front_kernel + the two packs
