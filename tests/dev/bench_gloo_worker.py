#!/usr/bin/env python3
"""Test-only: runs bench.py's own ``main`` (its step function, fences, MAX-over-ranks timing and the waveform
all-gather, float or PCM_16 bytes) under a CPU / gloo torchrun with a stand-in synthesis function (the CPU oracle).
Launched by tests/test_distributed_gloo.py::test_bench_main_world2_gloo; never used by the product path."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)

import torch  # noqa: E402

import bench  # noqa: E402
from oracle import hificar_oracle as O  # noqa: E402


def factory(params, sd, args):
    torch.set_num_threads(2)
    w = O.fold_weight_norm(sd)

    def synth(x):  # (B, C, T) -> (B, hop*T), as HiFiGANGenerator.ar_synthesis
        with torch.no_grad():
            return O.ar_loop_batched(w, params, x.permute(0, 2, 1), args.chunk_frames * bench.HOP, bench.HOP)

    def pcm16(y):  # arithmetic of hificar_pcm16 restated in torch
        return torch.clamp(torch.round(y.double() * 32767.0), -32768, 32767).to(torch.int16)

    return synth, pcm16


def factory_light(params, sd, args):
    """plumbing only (the 8-rank CPU tests): a cheap deterministic function of the features instead of the oracle"""
    torch.set_num_threads(1)

    def synth(x):  # (B, C, T) -> (B, hop*T)
        return torch.tanh(x.mean(1)).repeat_interleave(bench.HOP, dim=1).contiguous()

    def pcm16(y):
        return torch.clamp(torch.round(y.double() * 32767.0), -32768, 32767).to(torch.int16)

    return synth, pcm16


def factory_rank5_dies(params, sd, args):
    """one of eight ranks fails before its first step: the launcher must exit non-zero, no JSON line, no hang"""
    if int(os.environ.get("RANK", "0")) == 5:
        raise RuntimeError("rank 5 dies (test)")
    return factory_light(params, sd, args)


if __name__ == "__main__":
    bench.main(synth_factory=factory)
