"""GAN-TTS style ``GBlockGenerator`` behind the reference's ``generator_type`` plugin surface.

Drop-in for ``articulatory.models.GBlockGenerator`` (reference articulatory/models/gblock_gen.py:14-213; the GBlock itself
articulatory/layers/pytorch_layers.py:32-91): same class name, constructor keywords, parameter names and shapes — ``resamples.<i>.conv1.<n>``,
``resamples.<i>.res1.<n>``, ``resamples.<i>.conv2.<n>`` with the Sequential indices the reference's modules have (they move by one when a
GBlock upsamples), weight-norm keys included — and the same ``forward(c, spk_id=None, ar=None)`` / ``inference`` / ``remove_weight_norm`` /
``apply_weight_norm`` / ``register_stats``.  The arithmetic runs in ``libhificar.so`` (C ABI ``hificar_gblock_create`` + the shared entry points of
include/hificar.h): no PyTorch-operator implementation, no CPU fallback.

Which configurations exist.  The reference class only RUNS with ten (or nine) GBlocks and odd ``g_kernel_sizes`` — its defaults
(four even-sized GBlocks) fail in ``forward`` (oracle/make_golden_gblock.py's header has the details).  This class accepts exactly the
configurations the reference can run and raises ``ValueError`` at construction for the others instead of failing in ``forward`` — and
likewise for what libhificar's engine does not take (``_native.check_gblock_params``: ``g_kernel_sizes`` above 11, ``g_scales`` above 64,
``out_channels`` other than 1, PastFCEncoder widths).  ``g_kernel_sizes`` 9 and 11 run in inference; their dilation-27 conv's weight gradient
does not fit the LDS staging, so training such a model raises at the first backward.  One arithmetic, exact fp32 (``HIFICAR_PRECISION`` is not
consulted).
"""

import ctypes
import os

import numpy as np
import torch

from .. import _native
from .hifigan import _ConvParams, _NativeGenerator, _PastFCParams

GBLOCK_IN = (1, 1, 1, 2, 2, 2, 2, 4, 4, 8)    # gblock_gen.py:63
GBLOCK_OUT = (1, 1, 2, 2, 2, 2, 4, 4, 8, 8)   # gblock_gen.py:64


class _GBlockParams(torch.nn.Module):
    """Parameter holder of one GBlock (pytorch_layers.py:32-83): three Sequentials whose conv slots carry the reference's indices."""

    def __init__(self, input_dim, output_dim, upsample, kernel_size):
        super().__init__()
        up = [torch.nn.Upsample(scale_factor=upsample)] if upsample > 1 else []
        self.conv1 = torch.nn.Sequential(torch.nn.ReLU(), *up, _ConvParams((output_dim, input_dim, kernel_size), output_dim), torch.nn.ReLU(),
                                         _ConvParams((output_dim, output_dim, kernel_size), output_dim))
        self.res1 = torch.nn.Sequential(*up, _ConvParams((output_dim, input_dim, 1), output_dim))
        self.conv2 = torch.nn.Sequential(torch.nn.ReLU(), _ConvParams((output_dim, output_dim, kernel_size), output_dim), torch.nn.ReLU(),
                                         _ConvParams((output_dim, output_dim, kernel_size), output_dim))


class GBlockGenerator(_NativeGenerator):
    """Generator module based on GAN-TTS vocoder (MI355X-native forward and backward)."""

    def __init__(
        self,
        in_channels=80,
        out_channels=1,
        channels=512,
        kernel_size=7,
        g_scales=(8, 8, 2, 2),
        g_kernel_sizes=(16, 16, 4, 4),
        use_weight_norm=True,
        use_ar=False,
        ar_input=512,
        ar_hidden=256,
        ar_output=128,
        use_tanh=True,
        use_spk_id=False,
        num_spk=None,
        spk_emb_size=32,
        precision=None,  # "f32" only: res1 contracts over raw (un-activated) rows, which the bf16x3 row format does not carry
    ):
        super().__init__()
        # the reference's own checks (gblock_gen.py:50-52)
        assert kernel_size % 2 == 1, "Kernel size must be odd number."
        assert len(g_scales) == len(g_kernel_sizes)
        n = len(g_kernel_sizes)
        # ... and what it leaves to a shape error inside forward
        if n > len(GBLOCK_IN):
            raise ValueError(f"{n} GBlocks: the reference's channel plan has {len(GBLOCK_IN)} entries (gblock_gen.py:63-64)")
        if n < 1 or channels // GBLOCK_OUT[n - 1] != channels // 8:
            raise ValueError(f"{n} GBlocks end at channels // {GBLOCK_OUT[n - 1] if n else 1} channels but the output conv takes channels // 8 "
                             "(gblock_gen.py:63-64, 75-76): the reference class only runs with 9 or 10 GBlocks")
        if any(k % 2 != 1 for k in g_kernel_sizes):
            raise ValueError("g_kernel_sizes must be odd: with an even kernel a GBlock's main path loses samples against its residual path in the "
                             "reference (pytorch_layers.py:24-29, 85-91)")
        if precision is None:
            precision = "f32"  # (HIFICAR_PRECISION, the HiFi-GAN generators' process-wide default, does not apply: this class has one arithmetic)
        if precision != "f32":
            raise ValueError("GBlockGenerator runs in the exact-fp32 arithmetic only (precision='f32'; the environment variable HIFICAR_PRECISION "
                             "is not consulted by this class)")

        self.use_ar = use_ar
        self.use_spk_id = use_spk_id
        self.use_ph = False
        self.use_ph_loss = False
        self.num_resamples = n
        self._params = dict(
            in_channels=in_channels, out_channels=out_channels, channels=channels, kernel_size=kernel_size, g_scales=list(g_scales),
            g_kernel_sizes=list(g_kernel_sizes), use_tanh=use_tanh, use_ar=use_ar, ar_input=ar_input, ar_hidden=ar_hidden, ar_output=ar_output,
            use_spk_id=use_spk_id, num_spk=num_spk, spk_emb_size=spk_emb_size, ph_emb_size=0, num_ph=None,
        )
        _native.check_gblock_params(self._params)  # libhificar's own limits: fail here, not at the first forward on the device
        self.hop = int(np.prod(g_scales))
        self.precision = precision

        self.input_conv = _ConvParams((channels, in_channels, kernel_size), channels)
        self.resamples = torch.nn.ModuleList(
            [_GBlockParams(channels // GBLOCK_IN[i], channels // GBLOCK_OUT[i], g_scales[i], g_kernel_sizes[i]) for i in range(n)])
        out_mods = [torch.nn.LeakyReLU(), _ConvParams((out_channels, channels // 8, kernel_size), out_channels)]
        if use_tanh:
            out_mods.append(torch.nn.Tanh())
        self.output_conv = torch.nn.Sequential(*out_mods)
        if use_ar:
            self.ar_model = _PastFCParams(ar_input, ar_hidden, ar_output)
        if use_spk_id:
            assert num_spk is not None
            self.spk_emb_mat = torch.nn.Embedding(num_spk, spk_emb_size)
            self.spk_fc = torch.nn.Linear(spk_emb_size, in_channels)
        if use_weight_norm:
            self.apply_weight_norm()
        else:
            self.reset_parameters()
        self._handle = None
        self._workspaces = {}
        self._lib = None
        self._grad_sync = None
        from ..utils.optim_hook import watch

        watch(self)

    def _create_handle(self, lib, handle):
        cfg = _native.make_gblock_config(self._params, _native.PRECISIONS[self.precision])
        _native.check(lib.hificar_gblock_create(ctypes.byref(cfg), ctypes.byref(handle)), "hificar_gblock_create")

    def set_precision(self, precision):
        if precision != "f32":
            raise ValueError("GBlockGenerator runs in the exact-fp32 arithmetic only (precision='f32')")

    def _tap_shape(self, name, B, T, Tb):
        """Taps (hificar_debug_tap): "ar_feats", "input_conv", "resamples.<i>", "resamples.<i>.conv1a" / ".res1" / ".mid"."""
        p = self._params
        if name == "ar_feats":
            return (B, p["ar_output"]), None
        if name == "input_conv":
            return (B, p["channels"], Tb), T
        parts = name.split(".")
        if parts[0] != "resamples":
            raise ValueError(f"unknown tap {name!r}")
        i = int(parts[1])
        up = int(np.prod(p["g_scales"][:i + 1]))
        return (B, p["channels"] // GBLOCK_OUT[i], Tb * up), T * up

    def forward(self, c, spk_id=None, ar=None, ph=None, lengths=None):
        """c: (B, in_channels[-ar_output], T) -> (B, out_channels, T * prod(g_scales))  (gblock_gen.py:111-132).
        ``ph`` is accepted when None only: the reference's trainer passes ph= to every generator (train.py:276), which its own
        GBlockGenerator.forward(c, spk_id, ar) rejects with a TypeError."""
        if ph is not None:
            raise TypeError("GBlockGenerator.forward() has no phoneme conditioning (gblock_gen.py:111)")
        return super().forward(c, spk_id=spk_id, ar=ar, lengths=lengths)

    def inference(self, c, normalize_before=False):
        """gblock_gen.py:172-190, statement for statement: the reference unsqueezes BEFORE it transposes, so only a 1-D input
        (T,) of a one-channel model comes out 3-D: (T,) -> (T * prod(g_scales), out_channels)."""
        c = c.unsqueeze(1)
        if not isinstance(c, torch.Tensor):
            c = torch.tensor(c, dtype=torch.float).to(self._device())
        if normalize_before:
            c = (c - self.mean) / self.scale
        c = self.forward(c.transpose(1, 0).unsqueeze(0))
        return c.squeeze(0).transpose(1, 0)
