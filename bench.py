#!/usr/bin/env python3
"""bench.py — HiFi-CAR EMA->waveform synthesis throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one complete pass of the hot path over one batch of synthetic input: the batched
autoregressive synthesis of `--batch` (64) utterances of `--seconds` (10 s = 2000 frames of 13-dim
pitch+EMA at 200 Hz) into 16-kHz audio — 80 sequential 25-frame chunks (e2w_hifigan_car.yaml:135),
PastFCEncoder and the sample feedback included.  Inputs and weights are resident in HBM before the timed
region.  With N > 1 every rank synthesises its own 64 utterances (weak scaling, no data-path collective)
and the step ends with the one RCCL all-gather that collects the waveforms.

The headline (`value`, `dtype: "fp32"`, `roofline`) is the reference's arithmetic: exact IEEE fp32 products on
v_mfma_f32_32x32x2_f32.  The split-bf16 fast mode (`--precision bf16x3`, 16-bit-significand products, 1.5e-5 of
max|y| from fp32) is opt-in and reported as the labelled secondary leg `fast_bf16x3` with the same steps / warm-up
and its own roofline block.  `batch_sweep` repeats the fp32 measurement at batch 1 and 8 (north_star: 1/8/64).

A plain `python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) launches its own N ranks through
torch.distributed.run on 127.0.0.1 and forwards rank 0's line; under an outer torchrun it is one of the ranks.

The N = 1 line also carries `nonar` (BASELINE config 2: HiFi-GAN non-AR 12-dim EMA, batch 1 / 8 / 64 — north_star's literal sweep; roofline on
config 2's batch 8) and `gblock` (the reference's other a2w generator, GBlockGenerator, batch 64 AR synthesis), each with its own roofline block.

Prints ONE JSON line on rank 0.  `roofline` is measured in a second pass of the same K steps with every
kernel launch bracketed by HIP events on the launch stream (libhificar's profile hooks) so that the
event traffic does not perturb `value`; `cpu_baseline` times the CPU oracle on a bounded sample.
"""

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# generator_params of the reference's egs/ema/voc1/conf/e2w_hifigan_car.yaml:34-58 (values only)
CAR_PARAMS = dict(
    in_channels=141, out_channels=1, channels=512, kernel_size=7,
    upsample_scales=[5, 4, 2, 2], upsample_kernel_sizes=[10, 8, 4, 4], final_scale=80,
    resblock_kernel_sizes=[3, 7, 11], resblock_dilations=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    use_additional_convs=True, bias=True, nonlinear_activation="LeakyReLU",
    nonlinear_activation_params={"negative_slope": 0.1}, use_weight_norm=True, extra_art=False,
    use_ar=True, ar_input=512, ar_hidden=256, ar_output=128,
)
HOP = 80
SAMPLING_RATE = 16000
PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0}  # /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA peaks
PEAK_HBM_GBS = 8000.0                              # same guide: HBM3E 8 TB/s spec
DTYPE_NAME = {"f32": "fp32", "bf16x3": "bf16x3"}
MFMA_PER_MAC = {"f32": 1, "bf16x3": 3}             # bf16x3 issues 3 bf16 MFMAs per algorithmic MAC


def cpu_baseline(params, sd, chunk_frames, feats8):
    """CPU oracle (torch fp32 restatement, pinned to the reference's golden vectors) next to `value`, as SURVEY.md §8(d) specifies it:
    on the FIRST 8 UTTERANCES of the very batch `value` was measured on (8 x 2000 frames = 10-s clips, the same features), timed
      threads_1      torch.set_num_threads(1) — the recipe default (egs/ema/voc1/path.sh:13, OMP_NUM_THREADS=1), batched (B = 8)
      threads_best   the best of a few thread counts (chosen on a small probe: every core of a big host is ~10x SLOWER than 16), batched
      per_utterance  the B = 1 loop of egs/ema/voc1/local/predict_wav.py:124-137 — one utterance after the other — with 1 thread (first two
                     utterances) and with the best thread count (all eight)
    `value` / `cores` are the threads_best figure.  About 30 s of CPU work on the GPU box's host."""
    import torch

    from oracle import hificar_oracle as O

    all_cores = torch.get_num_threads()
    w = O.fold_weight_norm(sd)
    x8 = feats8.permute(0, 2, 1).contiguous()  # (8, T, 13): the oracle's (B, T, C) layout
    T = x8.shape[1]

    def run(x, threads, per_utterance=False):
        torch.set_num_threads(threads)
        with torch.no_grad():
            O.ar_loop_batched(w, params, x[:1, :chunk_frames], chunk_frames * HOP, HOP)  # warm-up: one chunk
            t0 = time.perf_counter()
            n = 0
            if per_utterance:
                for i in range(x.shape[0]):
                    n += O.ar_loop_batched(w, params, x[i:i + 1], chunk_frames * HOP, HOP).numel()
            else:
                n = O.ar_loop_batched(w, params, x, chunk_frames * HOP, HOP).numel()
            dt = time.perf_counter() - t0
        return {"samples_per_s": round(n / dt, 1), "seconds": round(dt, 2), "utterances": int(x.shape[0]), "threads": threads}

    probe = {}
    for nt in sorted({1, 8, 16, 32, all_cores}):
        if nt <= all_cores:
            probe[nt] = run(x8[:, :5 * chunk_frames], nt)["samples_per_s"]
    best = max(probe, key=probe.get)
    t_best = run(x8, best)
    t_one = run(x8, 1)
    pu_one = run(x8[:2], 1, per_utterance=True)
    pu_best = run(x8, best, per_utterance=True)
    torch.set_num_threads(all_cores)
    return {
        "value": t_best["samples_per_s"], "unit": "samples/s", "cores": best, "kind": "port",
        "sample": f"oracle.ar_loop_batched on the first 8 utterances of the batch `value` ran on: 8 x {T} frames ({T // chunk_frames} chunks of "
                  f"{chunk_frames}), torch {torch.__version__} CPU fp32, {all_cores} host threads available",
        "threads_1": t_one, "threads_best": t_best,
        "per_utterance": {"note": "B = 1, one utterance after the other (predict_wav.py:124-137)", "threads_1": pu_one, "threads_best": pu_best},
        # (a smaller sample: only there to choose `cores`)
        "by_threads_probe": {"sample": f"batch 8 x {5 * chunk_frames} frames", "samples_per_s": {str(k): v for k, v in probe.items()}},
    }


def make_step(synth, feats, use_dist, world, gather="f32", pcm16=None):
    """One bench step on this rank: synthesise this rank's batch, then (N > 1) the one collective that collects the
    waveforms ("waveform collection only").  ``gather="pcm16"`` converts to PCM_16 on the device first
    and gathers bytes (half the traffic on the links; RCCL / gloo have no int16 type).  Returns (step, gathered)."""
    import torch
    import torch.distributed as dist

    state = {"gathered": None, "gather_events": [], "gather_s": []}

    def step():
        y = synth(feats)
        if use_dist:
            send = pcm16(y) if gather == "pcm16" else y
            if state["gathered"] is None:
                state["gathered"] = torch.empty((world * send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
            out = state["gathered"]
            # the collective's own time, so that a scaling run separates synthesis from collection: device events on the GPU (the all-gather is
            # enqueued on the current stream's RCCL work queue), the host clock under gloo
            on_gpu = send.is_cuda
            if on_gpu:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            else:
                t0 = time.perf_counter()
            if send.dtype == torch.int16:
                dist.all_gather_into_tensor(out.view(torch.uint8), send.contiguous().view(torch.uint8))
            else:
                dist.all_gather_into_tensor(out, send.contiguous())
            if on_gpu:
                e1.record()
                state["gather_events"].append((e0, e1))
            else:
                state["gather_s"].append(time.perf_counter() - t0)
        return y

    return step, state


def timed_steps(step, fence, steps, warmup, use_dist=False, device=None):
    """W untimed warm-up steps, then exactly K steps between two fences (barrier + device sync); MAX over ranks."""
    import torch
    import torch.distributed as dist

    y = None
    for _ in range(warmup):
        y = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        y = step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, y


def collected_at(traffic_table, table_key):
    """where and when the PMC table's entry was collected: tools/summarize_profiles.py stamps every leg it rewrites"""
    c = ((traffic_table or {}).get("_collected") or {}).get(table_key)
    return f"collected at commit {c['commit']} on {c['date']}, profile tag {c['tag']}" if c else "collection commit not recorded"


def roofline_block(stats, precision, wall_s, steps, traffic_table, strict=False, table_key=None):
    """`roofline` object for the kernel with the largest total time of an event-bracketed pass.

    achieved = algorithmic flops (or bytes) of the launches / their summed event time; `bound` is derived from the
    kernel's own arithmetic intensity against the machine balance of its MFMA instruction (not hard-coded)."""
    dom = stats[0]
    total_ms = sum(s["total_ms"] for s in stats)
    sec = dom["total_ms"] * 1e-3
    tflops = dom["flops"] / sec / 1e12
    gbs = dom["bytes"] / sec / 1e9
    peak_tf = PEAK_TFLOPS[precision]
    intensity = dom["flops"] / max(dom["bytes"], 1.0)
    # machine balance of THIS arithmetic: bf16x3 spends 3 MFMAs per algorithmic MAC, so its matrix roof is peak / 3
    balance = peak_tf / MFMA_PER_MAC[precision] * 1e12 / (PEAK_HBM_GBS * 1e9)
    bound = "mfma" if intensity >= balance else "hbm"
    table_key = table_key or precision  # (the secondary legs have their own PMC passes: "nonar_f32", "gblock_f32")
    have_table = traffic_table is not None and table_key in traffic_table
    traffic = (traffic_table or {}).get(table_key, {}).get(dom["name"])
    if (have_table or (strict and traffic_table is not None)) and traffic is None:
        # the table is keyed by kernel name incl. template arguments: a tile picker that changes one silently turned `traffic` into null in
        # round 3's review.  A table that exists but does not know the dominant kernel is a STALE table: the headline leg refuses to print a
        # line on it, the labelled secondary leg shouts on stderr and says so in its block.
        msg = (f"bench.py: profiles/hbm_traffic.json has no '{table_key}' entry for the dominant kernel {dom['name']!r} (it knows "
               f"{sorted((traffic_table or {}).get(table_key, {}))}): re-run tools/collect_profiles.sh and commit the new PMC passes")
        if strict:
            raise SystemExit(msg + ", or pass --no-roofline")
        print("WARNING: " + msg, file=sys.stderr, flush=True)
    alg_bytes = dom["bytes"] / dom["launches"]
    mm = MFMA_PER_MAC[precision]
    blk = {
        "bound": bound, "kernel": dom["name"],
        "achieved": round(tflops if bound == "mfma" else gbs, 2),
        "peak": peak_tf if bound == "mfma" else PEAK_HBM_GBS,
        "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
        "frac": round((tflops / peak_tf) if bound == "mfma" else (gbs / PEAK_HBM_GBS), 4),
        "traffic": traffic,
        "traffic_source": (f"profiles/hbm_traffic.json[{table_key}] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench, per launch; "
                           f"{collected_at(traffic_table, table_key)})") if traffic else ("STALE TABLE: no entry for this kernel" if have_table else None),
        "algorithmic_bytes": round(alg_bytes),
        "traffic_over_algorithmic": round(traffic / alg_bytes, 3) if traffic else None,
        "flops_per_byte": round(intensity, 1), "machine_balance_flops_per_byte": round(balance, 1),
        "achieved_tflops": round(tflops, 2), "achieved_hbm_gbs_algorithmic": round(gbs, 1),
        "mfma_issue_tflops": round(tflops * mm, 2), "mfma_issue_frac": round(tflops * mm / peak_tf, 4),
        "launches": dom["launches"], "avg_launch_us": round(dom["total_ms"] * 1e3 / dom["launches"], 2),
        "flops_per_launch": round(dom["flops"] / dom["launches"], 1),
        "kernel_time_share": round(dom["total_ms"] / total_ms, 4),
        "all_kernels": [{"name": s["name"], "launches": s["launches"], "total_ms": round(s["total_ms"], 3),
                         "avg_us": round(s["total_ms"] * 1e3 / s["launches"], 2),
                         "tflops": round(s["flops"] / (s["total_ms"] * 1e-3) / 1e12, 2),
                         "algorithmic_bytes": round(s["bytes"] / s["launches"]),
                         "traffic": (traffic_table or {}).get(table_key, {}).get(s["name"])} for s in stats],
        "events_pass_ms_per_step": round(wall_s / steps * 1e3, 3),
    }
    return blk


def training_leg(steps=10, traffic_table=None):
    """Secondary leg: BASELINE config 5's train step on this GPU — the whole GAN iteration of
    articulatory_amd/bin/train.py::Trainer on the shipped recipe e2w_hifigan_car.yaml (full generator + multi-scale / multi-period
    discriminators, mel + adversarial + feature-matching losses, both Adam updates) at the recipe's batch (64 windows of 2000 samples +
    512 AR samples).  Exact fp32 (the reference has no bf16 path).  Not part of `value`.

    parity_gate   the fixture iteration first: the same Trainer at batch 8 on the seeds of tests/golden/gold_train_step.npz — every
                  logged loss against the values the REAL reference's Trainer._train_step produced (oracle/make_golden_train.py).
    flops         algorithmic FLOPs of one iteration AS THE STEP RUNS IT: generator forward x2 (with tape; again without, train.py:389)
                  + backward (data + weight gradients = 2 forwards); discriminator forward x3 (fake, real; fake again — D(real) of the
                  generator part is re-used), data-only backward x1 (generator part), full backward x2 (fake, real: 2 forwards each).
                  Loss GEMMs (DFT / mel: < 0.5 %) and element-wise work are not counted.
    kernels       an event-bracketed pass over the same iterations (libhificar's profile hooks on both engines)."""
    import numpy as np
    import torch

    from articulatory_amd.bin.train import Trainer
    from articulatory_amd.utils.recipes import recipe_train_config
    from articulatory_amd.utils.synth import synth_disc_state_dict, synth_state_dict, synth_train_batch

    dev = torch.device("cuda")

    def build(batch, seed_g, seed_d):
        cfg = recipe_train_config("car", aux="mel", batch=batch)
        t = Trainer(cfg, dev)
        t.G.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg["generator_params"], seed=seed_g).items()})
        t.D.load_state_dict({k: torch.from_numpy(v) for k, v in synth_disc_state_dict(cfg["discriminator_params"], seed=seed_d).items()})
        return t, cfg

    # ---- parity gate on the reference's own numbers
    gate = {"fixture": "tests/golden/gold_train_step.npz (articulatory.bin.train.Trainer._train_step on e2w_hifigan_car.yaml, batch 8)"}
    gpath = os.path.join(REPO, "tests", "golden", "gold_train_step.npz")
    gold = np.load(gpath)
    seed_g, seed_d, seed_x = (int(v) for v in gold["seeds"])
    t, cfg = build(int(gold["B"]), seed_g, seed_d)
    t.steps = 2
    log = {k: float(v) for k, v in t.train_step({k: torch.from_numpy(v) for k, v in synth_train_batch(cfg, seed_x, int(gold["B"])).items()}).items()}
    worst = 0.0
    for k, v in sorted(log.items()):
        ref = float(gold[f"mel::log::{k}"])
        err = abs(v - ref) / max(abs(ref), 1e-3)
        worst = max(worst, err)
        gate[k.split("/")[1]] = {"value": v, "reference": ref, "rel_err": float(f"{err:.2e}")}
    gate["tolerance"], gate["ok"] = 1e-4, bool(worst < 1e-4)
    assert gate["ok"], f"training parity gate failed: {gate}"
    del t

    # ---- the recipe's batch
    B = 64
    t, cfg = build(B, 1234, 4321)
    # (pinned, as the recipe's DataLoader delivers it — pin_memory: true — so that the copies to the device are asynchronous; a pageable batch drains
    #  the stream at every .to(device): about 1 ms per iteration)
    batch = {k: torch.from_numpy(v).pin_memory() for k, v in synth_train_batch(cfg, 20260929, B).items()}
    frames = cfg["batch_max_steps"] // HOP
    T_disc = cfg["generator_params"]["ar_input"] + cfg["batch_max_steps"]

    def iteration():
        t.steps = 2
        return t.train_step(batch)

    for _ in range(3):
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        log = iteration()
    t_enq = time.perf_counter()  # every launch of the K iterations is enqueued (train_step returns device scalars: nothing waits inside)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    host_enqueue = (t_enq - t0) / steps
    assert all(np.isfinite(float(v)) for v in log.values())
    g_fwd, d_fwd = 2.0 * t.G.macs(B, frames), 2.0 * t.D.macs(B, T_disc)
    breakdown = {"generator_forward_x2": 2 * g_fwd, "generator_backward": 2 * g_fwd, "discriminator_forward_x3": 3 * d_fwd,
                 "discriminator_backward_data_only_x1": d_fwd, "discriminator_backward_full_x2": 4 * d_fwd}
    flops = sum(breakdown.values())
    # per-kernel table from a SERIAL pass: the timed iterations above overlap the eight sub-discriminators (and the auxiliary loss, and the real
    # pass's backward) on side streams, where an event-bracketed launch's duration is an occupancy artefact; a second Trainer with every engine on
    # the caller's stream (HIFICAR_DISC_STREAMS=0, no side-stream overlap in the step) gives launch times that ARE a fraction of something
    del t
    # (the engine keeps the tile shapes of the overlapped run in that mode, so this table describes the kernels the timed iterations ran)
    # (HIFICAR_PROFILE_DETAIL: the serial pass's profile rows carry the layer, so the table below can be split by engine as well as by kernel name)
    serial_env = {"HIFICAR_DISC_STREAMS": "0", "HIFICAR_PROFILE_DETAIL": "1"}
    saved_env = {k: os.environ.get(k) for k in serial_env}
    os.environ.update(serial_env)  # (read when the discriminators' native handle is created: at the first forward)
    try:
        ts, _ = build(B, 1234, 4321)
        ts.config["overlap_aux_loss"] = ts.config["early_real_gradient"] = False

        def serial_iteration():
            ts.steps = 2
            return ts.train_step(batch)

        for _ in range(2):
            serial_iteration()
        torch.cuda.synchronize()
    finally:
        for k, v in saved_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    ts.G.profile_begin()
    ts.D.profile_begin()
    t1 = time.perf_counter()
    for _ in range(steps):
        serial_iteration()
    torch.cuda.synchronize()
    serial_dt = (time.perf_counter() - t1) / steps
    import re

    def group_of(base, layer):  # (the grouping of tools/pmc_by_layer.py, whose PMC totals sit in profiles/gan_traffic_by_group.json)
        if layer.startswith(("blocks.", "upsamples.", "input_conv")):
            return "generator convs (forward x2, data gradients)"
        if "mpd." in layer:
            return "period discriminators (convs, im2col / col2im)"
        if "msd." in layer:
            return "scale discriminators (convs, im2col / col2im)"
        if base.startswith("wgrad"):
            return "weight gradients"
        return "other (reductions, packs, losses, element-wise)"

    stats, groups = {}, {}
    for s in ts.G.profile_end() + ts.D.profile_end():
        full = s["name"]
        name = re.split(r"[| ]", full, maxsplit=1)[0]  # rows carry "kernel|layer xN" / "kernel shape ...": the per-name table strips that
        layer = full.split("|", 1)[1] if "|" in full else ""
        fam = re.sub(r"\d+_kernel$", "_kernel", name.split("<")[0]).replace("wreduce_gemm_kernel", "wreduce_kernel")
        g = group_of(fam, layer)
        for table, key in ((stats, name), (groups, g), (groups, fam + " @ " + g)):
            a = table.setdefault(key, dict(name=key, launches=0, total_ms=0.0, flops=0.0, bytes=0.0))
            a["launches"] += s["launches"]
            a["total_ms"] += s["total_ms"]
            a["flops"] += s["flops"]
            a["bytes"] += s["bytes"]
    stats = sorted(stats.values(), key=lambda s: -s["total_ms"])
    total_ms = sum(s["total_ms"] for s in stats)
    dom = stats[0]
    tf = flops / dt / 1e12
    table = (traffic_table or {}).get("train_gan", {})

    def row(s):
        alg = s["bytes"] / s["launches"]
        tr = table.get(s["name"])
        tfk = s["flops"] / (s["total_ms"] * 1e-3) / 1e12
        return {"name": s["name"], "launches_per_iteration": round(s["launches"] / steps, 1), "avg_launch_us": round(s["total_ms"] * 1e3 / s["launches"], 2),
                "ms_per_iteration": round(s["total_ms"] / steps, 3), "tflops": round(tfk, 2), "frac_of_fp32_mfma_peak": round(tfk / PEAK_TFLOPS["f32"], 4),
                "kernel_time_share": round(s["total_ms"] / total_ms, 4), "algorithmic_bytes": round(alg), "traffic": tr,
                "traffic_over_algorithmic": round(tr / alg, 3) if (tr and alg > 0) else None}

    gtable = {}
    gpath = os.path.join(REPO, "profiles", "gan_traffic_by_group.json")
    if os.path.exists(gpath):
        with open(gpath) as f:
            gtable = json.load(f)

    def group_row(gk):
        s = groups[gk]
        tfk = s["flops"] / (s["total_ms"] * 1e-3) / 1e12 if s["total_ms"] else 0.0
        pm = gtable.get(gk) or {}
        return {"group": gk, "launches_per_iteration": round(s["launches"] / steps, 1), "ms_per_iteration": round(s["total_ms"] / steps, 3),
                "tflops": round(tfk, 2), "frac_of_fp32_mfma_peak": round(tfk / PEAK_TFLOPS["f32"], 4), "kernel_time_share": round(s["total_ms"] / total_ms, 4),
                "algorithmic_MB_per_iteration": round(s["bytes"] / steps / 1e6, 1), "traffic_over_algorithmic": pm.get("traffic_over_algorithmic")}

    dom_gen = dom["name"].split("<")[0] + " @ generator convs (forward x2, data gradients)"
    by_group = [group_row(k) for k in sorted((k for k in groups if " @ " not in k), key=lambda k: -groups[k]["total_ms"])]
    dom_split = [group_row(k) for k in sorted((k for k in groups if k.startswith(dom["name"].split("<")[0] + " @ ")), key=lambda k: -groups[k]["total_ms"])]

    return {"note": "BASELINE config 5's recipe on ONE GPU, exact fp32 (the reference has no bf16 path); losses gated against the reference's "
                    "own _train_step fixture before timing",
            "workload": f"e2w_hifigan_car.yaml: batch {B} x ({cfg['batch_max_steps']} + {cfg['generator_params']['ar_input']} AR) samples, mel + adversarial + "
                        "feature-matching losses, Adam x2",
            "parity_gate": gate, "steps": steps, "gan_iteration_ms": round(dt * 1e3, 2), "gan_windows_per_s": round(B / dt, 1),
            # host time to ENQUEUE one iteration (Python + ctypes + launches, no profiler attached) next to its wall time: at >= 0.85 the
            # iteration is host-bound and a slower host — or 8 ranks sharing one — becomes the bottleneck
            "host_enqueue_ms": round(host_enqueue * 1e3, 2), "host_enqueue_over_wall": round(host_enqueue / dt, 3),
            "gan_training_samples_per_s": round(B * cfg["batch_max_steps"] / dt, 1),
            "flops_per_iteration": flops, "flops_breakdown": breakdown,
            "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_TFLOPS["f32"], "unit": "TFLOP/s", "frac": round(tf / PEAK_TFLOPS["f32"], 4),
                         "traffic": table.get(dom["name"]),
                         "traffic_source": ("profiles/hbm_traffic.json[train_gan] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of tools/gan_bench.py with "
                                            f"HIFICAR_DISC_STREAMS=0, average per launch of the kernel name; {collected_at(traffic_table, 'train_gan')})") if table else None,
                         "dominant_kernel": row(dom),
                         "note": "achieved / frac: the whole iteration (overlapped, as timed); dominant_kernel and `kernels`: a serial pass "
                                 "(every engine on one stream), so each row's tflops is that kernel's own rate"},
            "serial_iteration_ms": round(serial_dt * 1e3, 2),
            "kernel_ms_per_iteration_sum": round(total_ms / steps, 2),
            "kernels": [row(s) for s in stats[:14]],
            # the same serial pass split by ENGINE (layer names from HIFICAR_PROFILE_DETAIL): one kernel name covers the generator's ResBlock launches and
            # the discriminators' GEMM-form launches, whose rates and traffic differ by 2-3 x; traffic_over_algorithmic per group from the per-launch-shape
            # PMC join (tools/pmc_by_layer.py -> profiles/gan_traffic_by_group.json, profiles/r06_gan_pmc_hbm_by_layer.csv)
            "by_group": by_group,
            "dominant_kernel_family_by_group": dom_split,
            "dominant_kernel_on_generator_layers": group_row(dom_gen) if dom_gen in groups else None,
            "first_losses": {k.split("/")[1]: float(v) for k, v in sorted(log.items())}}


GBLOCK_PARAMS = dict(in_channels=141, out_channels=1, channels=512, kernel_size=7, g_scales=[5, 1, 4, 1, 1, 2, 1, 2, 1, 1], g_kernel_sizes=[3] * 10,
                     use_weight_norm=True, use_ar=True, ar_input=512, ar_hidden=256, ar_output=128, use_tanh=True)


def self_launch(gpus, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: this process is the launcher.  It starts N ranks of this very file through
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1 — the reference's only rendezvous is its training launcher,
    articulatory/bin/train.py:1459,1610-1615), forwards their output (rank 0 prints the one JSON line) and returns their exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or gpus) // gpus)))
    return subprocess.run(cmd, env=env).returncode


def _standin_factory():
    """Test hook: HIFICAR_BENCH_STANDIN="path/to/file.py:function" names a stand-in synthesis factory, so that a CPU / gloo test can call
    `python bench.py --gpus 2` exactly as the driver does (tests/test_distributed_gloo.py).  REFUSED when a GPU is visible, and a stand-in
    line says so (`"standin": true`, `dtype: "none"`, its own `metric` string): it can never pass for a measurement."""
    spec = os.environ.get("HIFICAR_BENCH_STANDIN")
    if not spec:
        return None
    import torch

    if torch.cuda.is_available():
        raise SystemExit("bench.py: HIFICAR_BENCH_STANDIN is a CPU test hook and a GPU is visible: unset it (a stand-in line is not a measurement)")
    import importlib.util

    path, fn = spec.rsplit(":", 1)
    ms = importlib.util.spec_from_file_location("_bench_standin", path)
    mod = importlib.util.module_from_spec(ms)
    ms.loader.exec_module(mod)
    return getattr(mod, fn)


def main(argv=None, synth_factory=None):
    """``synth_factory`` is a test hook (tests/test_distributed_gloo.py runs this very function under a 2-process gloo
    torchrun on CPU with a stand-in synthesis function); the product path never passes it."""
    if synth_factory is None:
        synth_factory = _standin_factory()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--seconds", type=float, default=10.0, help="clip length")
    ap.add_argument("--chunk-frames", type=int, default=25, help="batch_max_steps // hop_size (e2w_hifigan_car.yaml: 2000/80)")
    ap.add_argument("--precision", default=os.environ.get("HIFICAR_PRECISION", "f32"), choices=["f32", "bf16x3"],
                    help="conv arithmetic of the headline: f32 = the reference's IEEE fp32 products (default)")
    ap.add_argument("--gather", default="f32", choices=["f32", "pcm16"],
                    help="N > 1: collect float waveforms (default) or PCM_16 converted on the device (half the bytes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast-leg", action="store_true", help="skip the secondary bf16x3 leg (N=1 only)")
    ap.add_argument("--no-batch-sweep", action="store_true", help="skip the batch 1 / 8 legs (N=1 only)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-training", action="store_true", help="skip the training leg (N=1 only): generator train step and GAN iteration")
    ap.add_argument("--no-nonar", action="store_true", help="skip the non-AR leg (N=1 only): BASELINE config 2, batch 1 / 8 / 64")
    ap.add_argument("--no-gblock", action="store_true", help="skip the GBlockGenerator leg (N=1 only)")
    args = ap.parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's command is a plain `python3 bench.py --gpus N ...`: become the launcher of N ranks (one process per GPU)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:] if argv is None else argv))

    from articulatory_amd.utils.affinity import pin_rank

    # N ranks must not share one OpenMP pool / core set; before torch starts any thread (RCCL's watchdog, OpenMP workers inherit the mask)
    affinity = pin_rank(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    from articulatory_amd.utils.synth import synth_features, synth_state_dict

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    on_gpu = synth_factory is None
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a MI355X: no GPU visible (the generator has no CPU path)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")
    use_dist = world > 1 or "RANK" in os.environ  # under torchrun the RCCL path is exercised even at world size 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if on_gpu:
            dist.init_process_group(backend="nccl", device_id=dev)  # RCCL on ROCm
        else:
            dist.init_process_group(backend="gloo")

    ranks = 1
    if use_dist:
        ranks = dist.get_world_size()  # what the backend (RCCL under "nccl") actually formed, not what the environment promised
        if ranks != args.gpus and not (args.gpus == 1 and ranks == 1):
            raise SystemExit(f"--gpus {args.gpus} but the process group has {ranks} ranks")
    params = dict(CAR_PARAMS)
    sd = synth_state_dict(params, seed=1234)
    g = None
    if on_gpu:
        from articulatory_amd.models import HiFiGANGenerator
        from articulatory_amd.utils import pcm16

        g = HiFiGANGenerator(**params, precision=args.precision)
        g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        g.remove_weight_norm()
        g = g.eval().to(dev)

        def synth(x):
            return g.ar_synthesis(x, args.chunk_frames)
    else:
        synth, pcm16 = synth_factory(params, sd, args)

    B = args.batch
    T = int(round(args.seconds * SAMPLING_RATE / HOP))
    n_samples = B * T * HOP

    def features(batch):
        # synthetic 13-dim pitch+EMA, seed 20260929 + config index 3 + rank
        return torch.from_numpy(synth_features(batch, T, 13, seed=20260929 + 3 + 1000 * rank)).permute(0, 2, 1).contiguous().to(dev)

    feats = features(B)
    step, gstate = make_step(synth, feats, use_dist, world, args.gather, pcm16)

    def fence():
        if use_dist:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    with torch.no_grad():
        dt, y = timed_steps(step, fence, args.steps, args.warmup, use_dist, dev)
    gather_ms = None
    if use_dist:  # the timed steps' collectives (the warm-up steps' come first in the lists)
        if gstate["gather_events"]:
            gather_ms = sum(e0.elapsed_time(e1) for e0, e1 in gstate["gather_events"][-args.steps:]) / args.steps
        elif gstate["gather_s"]:
            gather_ms = sum(gstate["gather_s"][-args.steps:]) / args.steps * 1e3
    if use_dist:
        # the gathered block of this rank must be its own waveform (collection only, no arithmetic)
        mine = gstate["gathered"][rank * B:(rank + 1) * B]
        assert torch.equal(mine, pcm16(y) if args.gather == "pcm16" else y), "all-gather returned a different waveform"
    assert bool(torch.isfinite(y).all()), "non-finite output"

    ms_per_step = dt / args.steps * 1e3
    value = world * n_samples * args.steps / dt

    def macs_per_step(batch):
        if g is None:
            return 0.0
        return g.macs(batch, args.chunk_frames) * (T // args.chunk_frames) + (g.macs(batch, T % args.chunk_frames) if T % args.chunk_frames else 0.0)

    macs_step = macs_per_step(B)
    out = {
        "metric": "audio samples/sec (16 kHz) EMA->wav HiFi-CAR, batch 64",
        "value": round(value, 1),
        "unit": "samples/s",
        "n_gpus": world,
        "ranks": ranks,  # torch.distributed's own count (RCCL ranks at N > 1)
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPE_NAME[args.precision],
        "data": "synthetic",
        "config": {
            "workload": "configs[2]: HiFi-CAR (e2w_hifigan_car.yaml generator) 13-dim pitch+EMA -> 16 kHz, "
                        f"batch {B}/GPU, {args.seconds:g} s clips, {T // args.chunk_frames} sequential chunks of {args.chunk_frames} frames",
            "batch_per_gpu": B, "frames": T, "chunk_frames": args.chunk_frames, "samples_per_step_per_gpu": n_samples,
            "weights": "synthetic seed 1234 (articulatory_amd.utils.synth)", "parallelism": f"utterance-sharded x{world}",
            "arithmetic": "exact fp32 products (v_mfma_f32_32x32x2_f32), fp32 accumulate" if args.precision == "f32"
                          else "split-bf16 products hi*hi + hi*lo + lo*hi (3 x v_mfma_f32_32x32x16_bf16), fp32 accumulate",
            "gather": (args.gather if use_dist else "none"),
            "host_affinity": affinity,
        },
        "x_realtime": round(value / SAMPLING_RATE, 1),
        "algorithmic_tflops": round(2.0 * macs_step * world * args.steps / dt / 1e12, 2),
    }
    if use_dist:
        # rank 0's average time inside the waveform all-gather per step (device events under RCCL): ms_per_step - gather_ms is synthesis
        out["gather_ms"] = None if gather_ms is None else round(gather_ms, 3)
        out["gather_bytes_per_rank"] = int(n_samples * (2 if args.gather == "pcm16" else 4))
    if not on_gpu:
        # a stand-in synthesis ran (HIFICAR_BENCH_STANDIN / the synth_factory test hook): plumbing only, and the line says so
        out["standin"] = True
        out["metric"] = "STAND-IN (no synthesis ran): launcher / rendezvous / gather plumbing of bench.py on CPU"
        out["dtype"] = "none"
        out["config"]["arithmetic"] = "none (stand-in synthesis function)"
        out["config"]["weights"] = "none"
        if rank == 0:
            print(json.dumps(out), flush=True)
        if use_dist:
            dist.destroy_process_group()
        return out

    traffic_table = None
    tpath = os.path.join(REPO, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            traffic_table = json.load(f)

    def event_pass(x, steps):
        g.profile_begin()
        te0 = time.perf_counter()
        for _ in range(steps):
            g.ar_synthesis(x, args.chunk_frames)
        stats = g.profile_end()
        return stats, time.perf_counter() - te0

    if rank == 0 and not args.no_roofline:
        with torch.no_grad():
            stats, te = event_pass(feats, args.steps)
        out["roofline"] = roofline_block(stats, args.precision, te, args.steps, traffic_table, strict=(world == 1))  # (never strand the other ranks at the barrier)
    if use_dist:
        dist.barrier()

    solo = rank == 0 and world == 1
    if solo and not args.no_batch_sweep:
        # north_star: throughput at batch 1 / 8 / 64 (the headline above is batch 64), same clips, same arithmetic
        sweep = []
        for b in (1, 8):
            xb = feats[:b].contiguous() if b <= B else features(b)
            with torch.no_grad():
                k = max(3, min(args.steps, 10))
                dtb, _ = timed_steps(lambda: g.ar_synthesis(xb, args.chunk_frames), torch.cuda.synchronize, k, 2)
            vb = b * T * HOP * k / dtb
            tf = 2.0 * macs_per_step(b) * k / dtb / 1e12
            sweep.append({"batch": b, "value": round(vb, 1), "unit": "samples/s", "x_realtime": round(vb / SAMPLING_RATE, 1),
                          "ms_per_step": round(dtb / k * 1e3, 3), "steps": k, "algorithmic_tflops": round(tf, 2),
                          "frac_of_mfma_peak": round(tf / PEAK_TFLOPS[args.precision], 4)})
        sweep.append({"batch": B, "value": out["value"], "unit": "samples/s", "x_realtime": out["x_realtime"],
                      "ms_per_step": out["ms_per_step"], "steps": args.steps, "algorithmic_tflops": out["algorithmic_tflops"],
                      "frac_of_mfma_peak": round(out["algorithmic_tflops"] / PEAK_TFLOPS[args.precision], 4)})
        out["batch_sweep"] = sweep

    if solo and args.precision == "f32" and not args.no_fast_leg:
        # Secondary leg: the same batch through the opt-in split-bf16 arithmetic of the same library, same steps and warm-up,
        # its own roofline block, and how far its waveform is from the fp32 one on this very input.
        with torch.no_grad():
            y_exact = g.ar_synthesis(feats, args.chunk_frames)
            g.set_precision("bf16x3")
            dtf, y_fast = timed_steps(lambda: g.ar_synthesis(feats, args.chunk_frames), torch.cuda.synchronize, args.steps, args.warmup)
            vf = n_samples * args.steps / dtf
            leg = {
                "dtype": "bf16x3", "note": "opt-in fast mode (precision='bf16x3'): 16-bit-significand products, NOT the reference's arithmetic",
                "value": round(vf, 1), "unit": "samples/s", "x_realtime": round(vf / SAMPLING_RATE, 1),
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dtf / args.steps * 1e3, 3),
                "algorithmic_tflops": round(2.0 * macs_step * args.steps / dtf / 1e12, 2),
                "max_rel_diff_vs_fp32": float((y_fast - y_exact).abs().max() / y_exact.abs().max()), "tolerance": 1e-3,
            }
            if not args.no_roofline:
                stats, te = event_pass(feats, args.steps)
                leg["roofline"] = roofline_block(stats, "bf16x3", te, args.steps, traffic_table)
            g.set_precision(args.precision)
        out["fast_bf16x3"] = leg

    def secondary_leg(model, call, batch, macs, table_key, note, steps_leg, with_roofline=True):
        """value + roofline of one more model / configuration on this GPU: 2 warm-up + `steps_leg` timed passes between device syncs, then an
        event-bracketed pass of the same steps for the per-kernel table.  `call()` runs one pass over `batch` 10-s clips; `macs` = its MACs."""
        with torch.no_grad():
            dtl, _ = timed_steps(call, torch.cuda.synchronize, steps_leg, 2)
            v = batch * T * HOP * steps_leg / dtl
            tf = 2.0 * macs * steps_leg / dtl / 1e12
            leg = {"batch": batch, "value": round(v, 1), "unit": "samples/s", "x_realtime": round(v / SAMPLING_RATE, 1), "steps": steps_leg, "warmup": 2,
                   "ms_per_step": round(dtl / steps_leg * 1e3, 3), "algorithmic_tflops": round(tf, 2), "frac_of_mfma_peak": round(tf / PEAK_TFLOPS["f32"], 4)}
            if note:
                leg["note"] = note
            if with_roofline and not args.no_roofline:
                model.profile_begin()
                te0 = time.perf_counter()
                for _ in range(steps_leg):
                    call()
                stats = model.profile_end()
                leg["roofline"] = roofline_block(stats, "f32", time.perf_counter() - te0, steps_leg, traffic_table, table_key=table_key)
        return leg

    if solo and args.precision == "f32" and not args.no_nonar:
        # BASELINE config 2 / north_star's literal sweep: HiFi-GAN (non-AR) 12-dim EMA -> 16 kHz, one forward over whole 10-s clips
        # (articulatory/models/hifigan.py:298-314 is the reference's per-utterance entry; batched here), batch 1 / 8 / 64, exact fp32
        from articulatory_amd.models import HiFiGANGenerator as _G

        np_params = dict(CAR_PARAMS, in_channels=12, use_ar=False)
        gn = _G(**np_params, precision="f32")
        gn.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(np_params, seed=1234).items()})
        gn.remove_weight_norm()
        gn = gn.eval().to(dev)
        legs = []
        for b in (1, 8, 64):
            x12 = torch.from_numpy(synth_features(b, T, 12, seed=20260929 + 2)).permute(0, 2, 1).contiguous().to(dev)
            legs.append(secondary_leg(gn, lambda: gn(x12), b, gn.macs(b, T), "nonar_f32", None, max(3, min(args.steps, 10 if b < 64 else 5)),
                                      with_roofline=(b == 8)))
        out["nonar"] = {"workload": "configs[1]: HiFi-GAN (non-AR) 12-dim EMA -> 16 kHz, 10 s clips, one forward per batch; `roofline` on batch 8 "
                                    "(the configuration BASELINE.json names)", "dtype": "fp32", "batches": legs}
        del gn

    if solo and args.precision == "f32" and not args.no_gblock:
        # the reference's other a2w generator (articulatory/models/gblock_gen.py:111-132) at its runnable shape: ten GBlocks, channels 512, x80,
        # HiFi-CAR-shaped AR synthesis (13-dim features, chunk 25), exact fp32 only
        from articulatory_amd.models import GBlockGenerator
        from articulatory_amd.utils.synth import synth_gblock_state_dict

        gb = GBlockGenerator(**GBLOCK_PARAMS)
        gb.load_state_dict({k: torch.from_numpy(v) for k, v in synth_gblock_state_dict(GBLOCK_PARAMS, seed=1234).items()})
        gb.remove_weight_norm()
        gb = gb.eval().to(dev)
        nchunk = T // args.chunk_frames
        gmacs = gb.macs(B, args.chunk_frames) * nchunk + (gb.macs(B, T % args.chunk_frames) if T % args.chunk_frames else 0.0)
        leg = secondary_leg(gb, lambda: gb.ar_synthesis(feats, args.chunk_frames), B, gmacs, "gblock_f32",
                            "GBlockGenerator (g_scales 5,1,4,1,1,2,1,2,1,1, kernel 3, channels 512), AR synthesis at chunk 25", max(3, min(args.steps, 5)))
        leg["workload"] = f"GBlockGenerator 13-dim pitch+EMA -> 16 kHz, batch {B}, {args.seconds:g} s clips, {nchunk} sequential chunks of {args.chunk_frames} frames"
        leg["dtype"] = "fp32"
        out["gblock"] = leg
        del gb

    if solo and args.precision == "f32" and not args.no_training:
        out["training"] = training_leg(traffic_table=traffic_table)

    if solo and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(params, sd, args.chunk_frames, feats[:8].cpu())

    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
