"""GAN losses of the reference's train step on the native discriminators' outputs (``HiFiGANMultiScaleMultiPeriodDiscriminator(x,
native=True)``): the same values as articulatory/losses/adversarial_loss.py:12-123 and feat_match_loss.py:12-54 on the reference-shaped
tensors — every term is a mean over the elements of a layer output, which does not depend on how the engine lays them out.  Plain torch
element-wise reductions on views of the engine's buffers (differentiable: the gradients land in the buffers' own layout)."""
import torch


def _mean(out, fn):
    """mean over the valid elements of one DiscOutput of fn(view)."""
    return sum(fn(v).sum() for v in out.valid()) / out.numel()


def _last(o):
    return o[-1] if isinstance(o, (list, tuple)) else o


def generator_adversarial_loss(outputs, average_by_discriminators=True, loss_type="mse"):
    """adversarial_loss.py:12-62."""
    assert loss_type in ("mse", "hinge")
    loss = 0.0
    for i, o in enumerate(outputs):
        o = _last(o)
        loss = loss + (_mean(o, lambda v: (v - 1.0) ** 2) if loss_type == "mse" else -_mean(o, lambda v: v))
    return loss / (i + 1) if average_by_discriminators else loss


def discriminator_adversarial_loss(outputs_hat, outputs, average_by_discriminators=True, loss_type="mse"):
    """adversarial_loss.py:65-123 -> (real_loss, fake_loss)."""
    assert loss_type in ("mse", "hinge")
    real, fake = 0.0, 0.0
    for i, (oh, o) in enumerate(zip(outputs_hat, outputs)):
        oh, o = _last(oh), _last(o)
        if loss_type == "mse":
            real = real + _mean(o, lambda v: (v - 1.0) ** 2)
            fake = fake + _mean(oh, lambda v: v ** 2)
        else:
            real = real - _mean(o, lambda v: torch.clamp(v - 1.0, max=0.0))
            fake = fake - _mean(oh, lambda v: torch.clamp(-v - 1.0, max=0.0))
    if average_by_discriminators:
        real, fake = real / (i + 1), fake / (i + 1)
    return real, fake


def feature_match_loss(feats_hat, feats, average_by_layers=True, average_by_discriminators=True, include_final_outputs=False):
    """feat_match_loss.py:12-54 (the ground-truth features are constants)."""
    total = 0.0
    for i, (fh, f) in enumerate(zip(feats_hat, feats)):
        if not include_final_outputs:
            fh, f = fh[:-1], f[:-1]
        part = 0.0
        for j, (a, b) in enumerate(zip(fh, f)):
            part = part + sum((va - vb.detach()).abs().sum() for va, vb in zip(a.valid(), b.valid())) / a.numel()
        if average_by_layers:
            part = part / (j + 1)
        total = total + part
    return total / (i + 1) if average_by_discriminators else total


# ------------------------------------------------------------------------------------------------
# Mel-spectrogram loss (mel_loss.py:114-166) on libhificar: value and gradient in one native call
# ------------------------------------------------------------------------------------------------
import ctypes  # noqa: E402

import numpy as np  # noqa: E402

from . import _native  # noqa: E402
from .utils.mel import mel_filterbank  # noqa: E402


class _MelLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, y_hat, y):
        lib, handle = module._lib, module._handle
        B, T = y_hat.shape[0] * y_hat.shape[1], y_hat.shape[-1]
        dev = y_hat.device
        yh = y_hat.detach().to(torch.float32).contiguous()
        yr = y.detach().to(torch.float32).contiguous()
        need = y_hat.requires_grad
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            wsb = int(lib.hificar_mel_workspace_bytes(handle, B, T))
            ws = torch.empty(wsb // 4 + 64, dtype=torch.float32, device=dev)
            woff = ((-ws.data_ptr()) % 256) // 4
            value = torch.empty(1, dtype=torch.float32, device=dev)
            dy = torch.empty_like(yh) if need else None
            rc = lib.hificar_mel_loss(handle, yh.data_ptr(), yr.data_ptr(), B, T, value.data_ptr(), dy.data_ptr() if need else None,
                                      ws.data_ptr() + 4 * woff, wsb, stream)
        _native.check(rc, "hificar_mel_loss")
        ctx.dy = dy
        ctx.shape = y_hat.shape
        return value[0]

    @staticmethod
    def backward(ctx, g):
        dy = ctx.dy
        ctx.dy = None
        return None, (dy * g).view(ctx.shape) if dy is not None else None, None


class MelSpectrogramLoss(torch.nn.Module):
    """Mel-spectrogram loss, constructor arguments as the reference's (mel_loss.py:117-132).  MI355X-native (no CPU fallback)."""

    def __init__(self, fs=22050, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80, fmin=80, fmax=7600, center=True,
                 normalized=False, onesided=True, eps=1e-10, log_base=10.0, melmat=None):
        """``melmat`` (not in the reference): a (num_mels, fft_size / 2 + 1) filterbank to use instead of this package's restatement of
        ``librosa.filters.mel`` — e.g. librosa's own matrix where librosa is installed."""
        super().__init__()
        if window != "hann" or not center or normalized or not onesided:
            raise NotImplementedError("MelSpectrogramLoss: only window='hann', center=True, normalized=False, onesided=True are built")
        if log_base not in (None, 2.0, 10.0):
            raise ValueError(f"log_base: {log_base} is not supported.")
        fmin = 0 if fmin is None else fmin
        fmax = fs / 2 if fmax is None else fmax
        if melmat is None:
            melmat = mel_filterbank(fs, fft_size, num_mels, fmin, fmax)
        self.melmat = np.ascontiguousarray(np.asarray(melmat, dtype=np.float32))
        if self.melmat.shape != (num_mels, fft_size // 2 + 1):
            raise ValueError(f"melmat must be ({num_mels}, {fft_size // 2 + 1}), got {self.melmat.shape}")
        self._cfg = _native.HificarMelConfig(fft_size, hop_size, fft_size if win_length is None else win_length, num_mels, eps,
                                             0 if log_base is None else int(log_base), 0)
        self._lib = self._handle = None

    def _native_handle(self, dev):
        if self._handle is None:
            self._lib = _native.load_library()
            handle = ctypes.c_void_p()
            with torch.cuda.device(dev):
                _native.check(self._lib.hificar_mel_create(ctypes.byref(self._cfg), self.melmat.ctypes.data_as(ctypes.c_void_p), ctypes.byref(handle)),
                              "hificar_mel_create")
            self._handle = handle
        return self._handle

    def __del__(self):
        handle, lib = self.__dict__.get("_handle"), self.__dict__.get("_lib")
        if handle is not None and lib is not None:
            self.__dict__["_handle"] = None
            try:
                lib.hificar_mel_destroy(handle)
            except Exception:
                pass

    def forward(self, y_hat, y):
        """y_hat, y: (B, 1, T) -> scalar (mel_loss.py:152-166)."""
        if not y_hat.is_cuda:
            raise RuntimeError("MelSpectrogramLoss needs CUDA/HIP tensors; there is no CPU fallback")
        if y_hat.shape != y.shape or y_hat.dim() != 3:
            raise RuntimeError(f"Expected two (B, C, T) tensors of one shape, got {tuple(y_hat.shape)} and {tuple(y.shape)}")
        self._native_handle(y_hat.device)
        return _MelLossFunction.apply(self, y_hat, y)


# ------------------------------------------------------------------------------------------------
# Multi-resolution STFT loss (stft_loss.py:128-170)
# ------------------------------------------------------------------------------------------------
class _StftLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, y_hat, y):
        lib, handle = module._lib, module._handle
        B, T = y_hat.shape
        dev = y_hat.device
        yh = y_hat.detach().to(torch.float32).contiguous()
        yr = y.detach().to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            wsb = int(lib.hificar_mel_workspace_bytes(handle, B, T))
            ws = torch.empty(wsb // 4 + 64, dtype=torch.float32, device=dev)
            woff = ((-ws.data_ptr()) % 256) // 4
            values = torch.empty(2, dtype=torch.float32, device=dev)
            rc = lib.hificar_stft_loss_forward(handle, yh.data_ptr(), yr.data_ptr(), B, T, values.data_ptr(), ws.data_ptr() + 4 * woff, wsb, stream)
        _native.check(rc, "hificar_stft_loss_forward")
        ctx.module, ctx.ws, ctx.woff, ctx.wsb, ctx.BT = module, ws, woff, wsb, (B, T)
        return values[0], values[1]

    @staticmethod
    def backward(ctx, g_sc, g_mag):
        module = ctx.module
        B, T = ctx.BT
        dev = ctx.ws.device
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            gw = torch.stack([g_sc, g_mag]).to(torch.float32).contiguous()
            dy = torch.empty((B, T), dtype=torch.float32, device=dev)
            rc = module._lib.hificar_stft_loss_backward(module._handle, B, T, gw.data_ptr(), dy.data_ptr(), ctx.ws.data_ptr() + 4 * ctx.woff, ctx.wsb, stream)
        _native.check(rc, "hificar_stft_loss_backward")
        ctx.ws = None
        return None, dy, None


class STFTLoss(torch.nn.Module):
    """One resolution (stft_loss.py:87-125): (spectral convergence, log STFT magnitude).  MI355X-native."""

    def __init__(self, fft_size=1024, shift_size=120, win_length=600, window="hann_window"):
        super().__init__()
        if window != "hann_window":
            raise NotImplementedError("STFTLoss: only window='hann_window' is built")
        self._cfg = _native.HificarMelConfig(fft_size, shift_size, win_length, 0, 1e-7, 0, 1)  # clamp(re^2 + im^2, 1e-7), stft_loss.py:40
        self._lib = self._handle = None

    _native_handle = MelSpectrogramLoss._native_handle
    __del__ = MelSpectrogramLoss.__del__
    melmat = np.zeros((1, 1), np.float32)  # (unused in this mode)

    def forward(self, x, y):
        """x (predicted), y (ground truth): (B, T) -> (sc_loss, mag_loss)."""
        if not x.is_cuda:
            raise RuntimeError("STFTLoss needs CUDA/HIP tensors; there is no CPU fallback")
        self._native_handle(x.device)
        return _StftLossFunction.apply(self, x, y)


class MultiResolutionSTFTLoss(torch.nn.Module):
    """Constructor arguments and return values as the reference's (stft_loss.py:128-170)."""

    def __init__(self, fft_sizes=[1024, 2048, 512], hop_sizes=[120, 240, 50], win_lengths=[600, 1200, 240], window="hann_window"):
        super().__init__()
        assert len(fft_sizes) == len(hop_sizes) == len(win_lengths)
        self.stft_losses = torch.nn.ModuleList([STFTLoss(fs, ss, wl, window) for fs, ss, wl in zip(fft_sizes, hop_sizes, win_lengths)])

    def forward(self, x, y):
        if x.dim() == 3:
            x = x.reshape(-1, x.size(2))  # (B, C, T) -> (B x C, T)
            y = y.reshape(-1, y.size(2))
        sc_loss, mag_loss = 0.0, 0.0
        for f in self.stft_losses:
            sc, mag = f(x, y)
            sc_loss = sc_loss + sc
            mag_loss = mag_loss + mag
        n = len(self.stft_losses)
        return sc_loss / n, mag_loss / n
