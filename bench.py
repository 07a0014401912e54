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


def cpu_baseline(params, sd, chunk_frames, seed):
    """CPU oracle (torch fp32 restatement, pinned to the reference's golden vectors) on a bounded sample.

    The reference recipe runs single-threaded (egs/ema/voc1/path.sh:13, OMP_NUM_THREADS=1); torch's intra-op
    parallelism helps these small convolutions only up to a point (using every core of a big host is ~10x SLOWER
    than 16 threads), so a few thread counts are timed on a small sample and the best one is re-timed on a larger one.
    """
    import torch

    from articulatory_amd.utils.synth import synth_features
    from oracle import hificar_oracle as O

    all_cores = torch.get_num_threads()
    w = O.fold_weight_norm(sd)

    def run(B, T, threads):
        torch.set_num_threads(threads)
        x = torch.from_numpy(synth_features(B, T, 13, seed=seed))
        with torch.no_grad():
            O.ar_loop_batched(w, params, x[:, :chunk_frames], chunk_frames * HOP, HOP)  # warm-up: one chunk
            t0 = time.perf_counter()
            y = O.ar_loop_batched(w, params, x, chunk_frames * HOP, HOP)
            dt = time.perf_counter() - t0
        return y.numel() / dt, dt

    by_threads = {}
    for nt in sorted({1, 8, 16, 32, all_cores}):
        if nt <= all_cores:
            by_threads[nt] = run(8, 5 * chunk_frames, nt)[0]
    best = max(by_threads, key=by_threads.get)
    B, T = 64, 20 * chunk_frames  # 64 utterances x 2.5 s = 20 chunks of 25 frames each (~5 s of CPU work)
    value, dt = run(B, T, best)
    torch.set_num_threads(all_cores)
    return {
        "value": round(value, 1), "unit": "samples/s", "cores": best, "kind": "port",
        "sample": f"oracle.ar_loop_batched, batch {B} x {T} frames ({T // chunk_frames} chunks of {chunk_frames}), "
                  f"torch {torch.__version__} CPU fp32, best of the thread counts tried = {best} of {all_cores} host threads, {dt:.1f} s",
        "by_threads": {str(k): round(v, 1) for k, v in by_threads.items()},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--seconds", type=float, default=10.0, help="clip length")
    ap.add_argument("--chunk-frames", type=int, default=25, help="batch_max_steps // hop_size (e2w_hifigan_car.yaml: 2000/80)")
    ap.add_argument("--precision", default=os.environ.get("HIFICAR_PRECISION", "bf16x3"), choices=["f32", "bf16x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-check", action="store_true",
                    help="skip the extra leg that re-runs the batch with the exact-fp32 arithmetic (N=1 only)")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from articulatory_amd.models import HiFiGANGenerator
    from articulatory_amd.utils.synth import synth_features, synth_state_dict

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs one process per GPU: launch with "
                             f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: no GPU visible (the generator has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # under torchrun the RCCL path is exercised even at world size 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group(backend="nccl", device_id=dev)  # RCCL on ROCm

    params = dict(CAR_PARAMS)
    sd = synth_state_dict(params, seed=1234)
    g = HiFiGANGenerator(**params, precision=args.precision)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    g.remove_weight_norm()
    g = g.eval().to(dev)

    B = args.batch
    T = int(round(args.seconds * SAMPLING_RATE / HOP))
    n_samples = B * T * HOP
    # synthetic 13-dim pitch+EMA, seed 20260929 + config index 3 + rank
    feats = torch.from_numpy(synth_features(B, T, 13, seed=20260929 + 3 + 1000 * rank)).permute(0, 2, 1).contiguous().to(dev)
    gathered = torch.empty((world * B, T * HOP), dtype=torch.float32, device=dev) if use_dist else None

    def step():
        y = g.ar_synthesis(feats, args.chunk_frames)
        if use_dist:
            dist.all_gather_into_tensor(gathered, y)  # waveform collection only
        return y

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = step()
        fence()
        dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the gathered block of this rank must be its own waveform (collection only, no arithmetic)
        assert torch.equal(gathered[rank * B:(rank + 1) * B], y), "all-gather returned a different waveform"
    assert bool(torch.isfinite(y).all()), "non-finite output"

    ms_per_step = dt / args.steps * 1e3
    value = world * n_samples * args.steps / dt
    macs_step = g.macs(B, args.chunk_frames) * (T // args.chunk_frames) + (g.macs(B, T % args.chunk_frames) if T % args.chunk_frames else 0.0)

    out = {
        "metric": "audio samples/sec (16 kHz) EMA->wav HiFi-CAR, batch 64",
        "value": round(value, 1),
        "unit": "samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic",
        "config": {
            "workload": "configs[2]: HiFi-CAR (e2w_hifigan_car.yaml generator) 13-dim pitch+EMA -> 16 kHz, "
                        f"batch {B}/GPU, {args.seconds:g} s clips, {T // args.chunk_frames} sequential chunks of {args.chunk_frames} frames",
            "batch_per_gpu": B, "frames": T, "chunk_frames": args.chunk_frames, "samples_per_step_per_gpu": n_samples,
            "weights": "synthetic seed 1234 (articulatory_amd.utils.synth)", "parallelism": f"utterance-sharded x{world}",
        },
        "x_realtime": round(value / SAMPLING_RATE, 1),
        "algorithmic_tflops": round(2.0 * macs_step * world * args.steps / dt / 1e12, 2),
    }

    if rank == 0 and not args.no_roofline:
        with torch.no_grad():
            g.profile_begin()
            te0 = time.perf_counter()
            for _ in range(args.steps):
                g.ar_synthesis(feats, args.chunk_frames)
            stats = g.profile_end()
            te = time.perf_counter() - te0
        dom = stats[0]
        total_ms = sum(s["total_ms"] for s in stats)
        achieved = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.precision]
        traffic = None
        tpath = os.path.join(REPO, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get(args.precision, {}).get(dom["name"])
        out["roofline"] = {
            "bound": "mfma", "kernel": dom["name"], "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic,
            # bf16x3 issues 3 bf16 MFMAs per algorithmic MAC: the matrix pipe itself runs at 3x `achieved`
            "mfma_issue_tflops": round(achieved * (3 if args.precision == "bf16x3" else 1), 2),
            "mfma_issue_frac": round(achieved * (3 if args.precision == "bf16x3" else 1) / peak, 4),
            "launches": dom["launches"], "avg_launch_us": round(dom["total_ms"] * 1e3 / dom["launches"], 2),
            "flops_per_launch": round(dom["flops"] / dom["launches"], 1),
            "kernel_time_share": round(dom["total_ms"] / total_ms, 4),
            "all_kernels": [{"name": s["name"], "launches": s["launches"], "total_ms": round(s["total_ms"], 3),
                             "tflops": round(s["flops"] / (s["total_ms"] * 1e-3) / 1e12, 2)} for s in stats],
            "events_pass_ms_per_step": round(te / args.steps * 1e3, 3),
        }
    if use_dist:
        dist.barrier()

    if rank == 0 and world == 1 and args.precision != "f32" and not args.no_exact_check:
        # Same batch through the exact-fp32 MFMA arithmetic of the same library: its throughput, and how far the
        # default (split-bf16) arithmetic is from it on this very input — the parity figure that goes with `value`.
        with torch.no_grad():
            y_fast = g.ar_synthesis(feats, args.chunk_frames)
            g.set_precision("f32")
            y_exact = g.ar_synthesis(feats, args.chunk_frames)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                g.ar_synthesis(feats, args.chunk_frames)
            torch.cuda.synchronize()
            dte = (time.perf_counter() - t0) / 2
            g.set_precision(args.precision)
        out["exact_f32"] = {
            "value": round(n_samples / dte, 1), "unit": "samples/s", "x_realtime": round(n_samples / dte / SAMPLING_RATE, 1),
            "algorithmic_tflops": round(2.0 * macs_step / dte / 1e12, 2),
            "max_rel_diff_of_default_arithmetic": float((y_fast - y_exact).abs().max() / y_exact.abs().max()),
            "tolerance": 1e-3,
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(params, sd, args.chunk_frames, seed=20260929)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
