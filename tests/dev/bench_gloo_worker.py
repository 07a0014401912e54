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


if __name__ == "__main__":
    bench.main(synth_factory=factory)
