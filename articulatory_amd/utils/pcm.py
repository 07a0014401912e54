"""Device-side float -> PCM_16 conversion (C ABI: hificar_pcm16)."""

import ctypes

import torch

from .. import _native


def pcm16(y: torch.Tensor) -> torch.Tensor:
    """float waveform on the GPU -> int16 tensor of the same shape, y = clip(round(x * 32767)): the sample format
    the reference writes with ``sf.write(..., "PCM_16")`` (articulatory/bin/decode.py:319-324), produced before the
    device->host copy (or the multi-GPU gather) so that half the bytes move."""
    if not y.is_cuda:
        raise RuntimeError("pcm16 needs a CUDA/HIP tensor; there is no CPU fallback")
    lib = _native.load_library()
    x = y.to(torch.float32).contiguous()
    out = torch.empty(x.shape, dtype=torch.int16, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.hificar_pcm16(x.data_ptr(), out.data_ptr(), x.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _native.check(rc, "hificar_pcm16")
    return out
