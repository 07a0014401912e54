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
