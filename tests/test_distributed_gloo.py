"""Utterance-batch sharding + waveform all-gather with a 2-process gloo group on CPU (the GPU box runs the
same code over RCCL).  The synthesis function is the CPU oracle here — test-only stand-in for the HIP path."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import E2W_PARAMS
from articulatory_amd.bin.shard import shard_items, shard_range, synthesize_sharded


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from articulatory_amd.utils.synth import synth_features, synth_state_dict
        from oracle import hificar_oracle as O
        params = dict(E2W_PARAMS)
        w = O.fold_weight_norm(synth_state_dict(params, seed=1234))
        feats = torch.from_numpy(synth_features(4, 30, 13, seed=77))  # 4 utterances, 25 + 5 ragged frames
        calls = []

        def synth(x):
            calls.append(tuple(x.shape))
            with torch.no_grad():
                return O.ar_loop_batched(w, params, x, 2000, 80)

        y = synthesize_sharded(synth, feats)
        assert calls == [(2, 30, 13)]
        if rank == 0:
            with torch.no_grad():
                full = O.ar_loop_batched(w, params, feats, 2000, 80)
            q.put(("ok", float((y - full).abs().max()), tuple(y.shape)))
        # PCM_16 collection (half the bytes on the wire): the gather is dtype-agnostic; on the GPU the conversion is
        # articulatory_amd.utils.pcm16 (hificar_pcm16), here its arithmetic restated in torch
        def synth_pcm(x):
            return torch.clamp(torch.round(synth(x).double() * 32767.0), -32768, 32767).to(torch.int16)

        y16 = synthesize_sharded(synth_pcm, feats)
        assert y16.dtype == torch.int16 and tuple(y16.shape) == tuple(y.shape)
        assert int((y16.int() - torch.round(y.double() * 32767.0).int()).abs().max()) == 0
        with pytest.raises(ValueError):
            synthesize_sharded(synth, feats[:3])
    except Exception as e:  # pragma: no cover
        if rank == 0:
            q.put(("err", repr(e), None))
        raise
    finally:
        dist.destroy_process_group()


def test_shard_range():
    assert shard_range(512, 8, 3) == (192, 256)
    with pytest.raises(ValueError):
        shard_range(10, 4, 0)


def test_sharded_synthesis_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    status, err, shape = q.get(timeout=10)
    assert status == "ok", err
    assert shape == (4, 2400)
    assert err < 1e-6  # each rank's slice equals the unsharded result (utterances are independent)


@pytest.mark.parametrize("gather", ["f32", "pcm16"])
def test_bench_main_world2_gloo(gather):
    """bench.py's own main() — step function, barrier + MAX-over-ranks timing, the waveform all-gather (float, or PCM_16
    as bytes) and its self-check — under a 2-process torchrun on CPU/gloo with a stand-in synthesis function."""
    import json
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(repo, "tests", "dev", "bench_gloo_worker.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--seconds", "0.25", "--gather", gather]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=repo)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["gather"] == gather and d["config"]["parallelism"] == "utterance-sharded x2"
    # whole-job aggregate: both ranks' samples over the max-over-ranks time
    assert abs(d["value"] - 2 * 2 * 50 * 80 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-3


def test_bench_self_launches_its_ranks_like_the_driver_calls_it():
    """The driver's bench command is a plain ``python3 bench.py --gpus N --steps K --warmup W`` (BENCH_rNN.json's `cmd`), no torchrun: with
    N > 1 bench.py must start its own N ranks (torch.distributed.run on 127.0.0.1) and forward rank 0's single JSON line.  CPU / gloo stand-in
    for the synthesis through the HIFICAR_BENCH_STANDIN test hook; everything else — argument parsing, the launcher, rendezvous, barriers,
    MAX-over-ranks timing, the gather and its self-check — is the product code path."""
    import json
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="2", HIFICAR_BENCH_STANDIN=os.path.join(repo, "tests", "dev", "bench_gloo_worker.py") + ":factory")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--seconds", "0.25"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=repo)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "utterance-sharded x2" and d["config"]["gather"] == "f32"
    assert abs(d["value"] - 2 * 2 * 50 * 80 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-3
    # a stand-in line can never pass for a measurement: it is marked, names no arithmetic, and carries its own metric string
    assert d["standin"] is True and d["dtype"] == "none" and d["metric"].startswith("STAND-IN") and "HiFi-CAR" not in d["metric"]
    assert d["config"]["arithmetic"].startswith("none") and "roofline" not in d and "cpu_baseline" not in d
    # the collective's own time is in the N > 1 line (the first real SCALE run separates synthesis from collection with it)
    assert 0.0 <= d["gather_ms"] <= d["ms_per_step"] and d["gather_bytes_per_rank"] == 2 * 50 * 80 * 4


def _bench_env(repo, factory):
    env = dict(os.environ, HIFICAR_BENCH_STANDIN=os.path.join(repo, "tests", "dev", "bench_gloo_worker.py") + ":" + factory)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS"):
        env.pop(k, None)
    return env


def test_bench_gpus_8_as_the_driver_launches_it():
    """`python bench.py --gpus 8` exactly as the driver's SCALE run calls it, on CPU / gloo with a stand-in synthesis: the launcher picks a free
    port and bounds OMP_NUM_THREADS per rank, eight ranks rendezvous on 127.0.0.1, each pins itself to its own eighth of the cores, ONE JSON line
    comes back with the whole-job aggregate over the max-over-ranks time, the eight-way all-gather checks itself, `gather_ms` is there."""
    import json
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "2", "--seconds", "0.25"],
                       capture_output=True, text=True, timeout=900, env=_bench_env(repo, "factory_light"), cwd=repo)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks"] == 8 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["standin"] is True
    assert d["config"]["parallelism"] == "utterance-sharded x8"
    assert abs(d["value"] - 8 * 2 * 50 * 80 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-3
    assert d["gather_ms"] is not None and 0.0 <= d["gather_ms"] <= d["ms_per_step"]
    aff = d["config"]["host_affinity"]
    if aff is not None and hasattr(os, "sched_getaffinity") and len(os.sched_getaffinity(0)) >= 8:
        n = len(os.sched_getaffinity(0))
        assert f"({n // 8} of {n})" in aff  # rank 0's own eighth of the cores


def test_bench_gpus_8_propagates_a_failing_rank():
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "1", "--warmup", "0", "--batch", "1", "--seconds", "0.25"],
                       capture_output=True, text=True, timeout=900, env=_bench_env(repo, "factory_rank5_dies"), cwd=repo)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_refuses_a_stand_in_when_a_gpu_is_visible(monkeypatch):
    """HIFICAR_BENCH_STANDIN on a box with a GPU must stop bench.py, not produce a line."""
    import bench

    monkeypatch.setenv("HIFICAR_BENCH_STANDIN", "/nonexistent.py:f")
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    with pytest.raises(SystemExit, match="HIFICAR_BENCH_STANDIN"):
        bench._standin_factory()


def test_decode_cli_shards_over_8_ranks_under_torchrun(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 8 articulatory_amd/bin/decode.py ... --dry-run`: every rank takes its own share of the
    utterance list (lengths from the .npy headers, longest first to the least-loaded rank), the shares are disjoint, cover the list, and are
    balanced in frames — the file-to-file decode needs no collective at all."""
    import json
    import socket
    import subprocess
    import sys

    import numpy as np
    import yaml

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(0)
    lens = [int(v) for v in rng.integers(260, 900, size=37)]
    scp = tmp_path / "feats.scp"
    with open(scp, "w") as f:
        for i, n in enumerate(lens):
            np.save(tmp_path / f"u{i:02d}.npy", np.zeros((n, 13)))
            f.write(f"u{i:02d} {tmp_path / f'u{i:02d}.npy'}\n")
    cfg = tmp_path / "config.yml"
    cfg.write_text(yaml.safe_dump({"format": "npy", "generator_type": "HiFiGANGenerator", "generator_params": {}}))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1", PYTHONPATH=repo + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port",
                        str(port), os.path.join(repo, "articulatory_amd", "bin", "decode.py"), "--feats-scp", str(scp), "--outdir", str(tmp_path / "out"),
                        "--checkpoint", str(tmp_path / "ckpt.pkl"), "--config", str(cfg), "--verbose", "0", "--dry-run"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=repo)
    assert r.returncode == 0, r.stderr[-3000:]
    shares = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert sorted(s["rank"] for s in shares) == list(range(8)) and all(s["world_size"] == 8 and s["local_rank"] == s["rank"] for s in shares)
    got = sorted(u for s in shares for u in s["utterances"])
    assert got == [f"u{i:02d}" for i in range(37)]  # disjoint and complete
    frames = [s["frames"] for s in shares]
    assert sum(frames) == sum(lens) and max(frames) - min(frames) <= max(lens)


def test_bench_self_launch_propagates_a_failing_rank():
    """A rank that dies must make the launcher exit non-zero (no JSON line, no hang)."""
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="2", HIFICAR_BENCH_STANDIN=os.path.join(repo, "tests", "dev", "bench_gloo_worker.py") + ":no_such_factory")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1", "--seconds", "0.25"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=repo)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_shard_items_deals_by_length():
    items = [("a", 5), ("b", 50), ("c", 7), ("d", 40), ("e", 6), ("f", 45)]
    shares = [shard_items(items, 2, r, length_of=lambda kv: kv[1]) for r in range(2)]
    assert sorted(shares[0] + shares[1]) == sorted(items) and not set(shares[0]) & set(shares[1])
    assert abs(sum(v for _, v in shares[0]) - sum(v for _, v in shares[1])) <= 20  # longest first to the least-loaded rank: 50+7+6+5 vs 45+40
    assert shard_items(items, 1, 0) == items
    assert shard_items(items, 3, 1) == [items[1], items[4]]


def test_unsharded_passthrough():
    x = torch.arange(6.0).reshape(3, 2)
    assert torch.equal(synthesize_sharded(lambda t: t * 2, x), x * 2)


# ---------------------------------------------------------------- bucketed, overlapped gradient all-reduce (articulatory_amd/utils/buckets.py)
def test_bucket_ranges_merge_adjacent_slots():
    from articulatory_amd.utils.buckets import bucket_ranges

    # raw order of a tiny generator: input conv (g, v, bias) | upsampler | block of stage 0 | block of stage 1 | output conv | MLP
    ids = [2, 2, 2, 2, 1, 1, 0, 0, 0, 2]
    numels = [5, 40, 5, 13, 8, 3, 16, 2, 7, 9]
    ranges, total = bucket_ranges(ids, numels, 3)
    assert total == 8 + 40 + 8 + 16 + 8 + 4 + 16 + 4 + 8 + 12
    assert ranges[2] == [(0, 72), (112, 12)]          # front: the head of the buffer and the MLP at its end
    assert ranges[1] == [(72, 12)] and ranges[0] == [(84, 28)]


def _bucket_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from articulatory_amd.utils.buckets import BucketReducer, bucket_ranges

        rng = np.random.default_rng(5)
        n_buckets = 5
        numels = [int(v) for v in rng.integers(1, 3000, size=40)]
        ids = sorted(int(v) for v in rng.integers(0, n_buckets, size=40))[::-1]        # completion order differs from memory order
        ids[3], ids[-2] = ids[-2], ids[3]                                             # ... and one bucket is split into two ranges
        ranges, total = bucket_ranges(ids, numels, n_buckets)
        grads = torch.from_numpy(np.random.default_rng(100 + rank).standard_normal(total).astype(np.float32))  # this rank's gradients
        want = grads.clone()
        dist.all_reduce(want)                                                          # the single-collective result
        want /= world
        raw = torch.zeros(total)
        red = BucketReducer(raw, ranges, None, True)
        for b in (3, 0, 4, 1, 2):                                                      # buckets complete in "backward" order
            for off, n in ranges[b]:
                raw[off:off + n] = grads[off:off + n]                                  # the bucket's gradients land ...
            red.reduce(b)                                                              # ... and its collective starts at once
        with pytest.raises(RuntimeError):
            red.reduce(1)
        out = red.finish()
        assert out is raw and torch.equal(raw, want), float((raw - want).abs().max())
        red2 = BucketReducer(torch.zeros(total), ranges, None, True)
        red2.reduce(0)
        with pytest.raises(RuntimeError, match="never reduced"):
            red2.finish()
        for w in red2.pending:
            w.wait()
        if rank == 0:
            q.put(("ok", None, None))
    except Exception as e:  # pragma: no cover
        if rank == 0:
            q.put(("err", repr(e), None))
        raise
    finally:
        dist.destroy_process_group()


def test_bucketed_all_reduce_equals_single_collective_world2():
    """BucketReducer at world size 2 over gloo: gradients injected bucket by bucket in backward order, every bucket's all-reduce started
    as it completes — bit-identical to ONE all-reduce of the whole buffer followed by the average (what sync_gradients did before)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    status, msg, _ = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
    assert status == "ok", msg
    assert all(p.exitcode == 0 for p in procs)


def _hook_worker(rank, world, port, q):
    """The bucket wiring both networks' backward passes use (articulatory_amd/utils/buckets.py::BucketHook + BucketReducer) driven by a stand-in
    native backward: callbacks in the native (non-ascending) order, some of them from side threads as the discriminators' sub-network streams
    would, a chain rule between "folded" and "raw" gradients, an error inside a callback."""
    import threading

    import torch.distributed as dist

    from articulatory_amd.utils.buckets import BucketHook, BucketReducer, bucket_ranges

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_buckets = 6
        ids = [5, 0, 0, 3, 3, 3, 1, 2, 2, 4, 4, 5]
        numels = [9, 130, 64, 7, 1, 300, 32, 17, 5, 1024, 3, 11]
        ranges, total = bucket_ranges(ids, numels, n_buckets)
        folded = torch.from_numpy(np.random.default_rng(7 + rank).standard_normal(total).astype(np.float32))  # this rank's folded gradients
        want = folded * 0.5
        dist.all_reduce(want)
        want /= world
        raw = torch.zeros(total)
        seen = []

        def chain_rule(bucket, bstream):  # stand-in for hificar[_disc]_weight_norm_backward_bucket: raw slices of ONE bucket from the folded ones
            seen.append((bucket, bstream, threading.get_ident()))
            for off, n in ranges[bucket]:
                raw[off:off + n] = folded[off:off + n] * 0.5

        entered = []

        class Ctx:  # stand-in for torch.cuda.stream(ExternalStream(bstream))
            def __init__(self, s):
                self.s = s

            def __enter__(self):
                entered.append(self.s)

            def __exit__(self, *a):
                return False

        hook = BucketHook(BucketReducer(raw, ranges, None, True), chain_rule, Ctx)
        order = (4, 1, 5, 0, 3, 2)  # the order the (stand-in) native backward completes its buckets in: identical on every rank
        for i, b in enumerate(order):
            if i % 2:  # every other callback arrives on a side thread (one at a time: the collectives' order must match across ranks)
                t = threading.Thread(target=hook, args=(b, 1000 + b, None))
                t.start()
                t.join()
            else:
                hook(b, 1000 + b, None)
        out = hook.finish()
        assert out is raw and torch.equal(raw, want), float((raw - want).abs().max())
        assert [s[0] for s in seen] == list(order) and entered == [1000 + b for b in order]
        assert len({s[2] for s in seen}) >= 2  # callbacks really came from more than one thread

        # a failing chain rule inside a callback: nothing crosses the (C) caller's frames, finish() raises it — on every rank alike
        def bad_rule(bucket, bstream):
            if bucket == 1:
                raise ValueError("chain rule failed")
            chain_rule(bucket, bstream)

        hook2 = BucketHook(BucketReducer(torch.zeros(total), ranges, None, True), bad_rule)
        for b in order:
            hook2(b, None, None)  # must not raise here
        try:
            hook2.finish()
            raise AssertionError("finish() swallowed the callback's error")
        except ValueError as e:
            assert "chain rule failed" in str(e)
        if rank == 0:
            q.put(("ok", None, None))
    except Exception as e:  # pragma: no cover
        if rank == 0:
            q.put(("err", repr(e), None))
        raise
    finally:
        dist.destroy_process_group()


def test_bucket_hook_wiring_with_a_stand_in_backward_world2():
    """What sync_gradients() installs in both networks' backward passes, end to end at world size 2 over gloo, with a stand-in for the native
    backward: buckets reported out of order and from side threads, the per-bucket chain rule in front of each collective, errors inside the
    callback.  (The CUDA-side plumbing around it — ctypes callback, ExternalStream — runs in the world-1 RCCL tests on the GPU box.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hook_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    status, msg, _ = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
    assert status == "ok", msg
    assert all(p.exitcode == 0 for p in procs)


def test_pin_rank_partitions_the_allowed_cores():
    """articulatory_amd/utils/affinity.py: rank r of W gets the r-th W-th of the cores this process may use, and a bounded intra-op pool."""
    if not hasattr(os, "sched_getaffinity"):
        pytest.skip("no sched_getaffinity")
    import subprocess
    import sys

    code = ("import os, sys; sys.path.insert(0, %r); from articulatory_amd.utils.affinity import pin_rank; import torch;"
            "before = sorted(os.sched_getaffinity(0)); d = pin_rank(int(sys.argv[1]), 2); after = sorted(os.sched_getaffinity(0));"
            "print(len(before), after[0], after[-1], len(after), torch.get_num_threads(), d is not None)") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = [subprocess.run([sys.executable, "-c", code, str(r)], capture_output=True, text=True, check=True).stdout.split() for r in (0, 1)]
    n = int(outs[0][0])
    if n < 2:
        pytest.skip("one core")
    a, b = outs
    assert int(a[3]) == int(b[3]) == n // 2 and int(a[2]) < int(b[1])      # disjoint contiguous halves
    assert int(a[4]) == min(n // 2, 16) and a[5] == "True"
    env = dict(os.environ, HIFICAR_NO_AFFINITY="1")
    off = subprocess.run([sys.executable, "-c", code, "0"], capture_output=True, text=True, check=True, env=env).stdout.split()
    assert int(off[3]) == n and off[5] == "False"
