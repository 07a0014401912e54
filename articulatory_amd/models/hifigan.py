"""HiFi-GAN / HiFi-CAR generator behind the reference's ``generator_type`` plugin surface.

Drop-in for ``articulatory.models.HiFiGANGenerator`` (reference articulatory/models/hifigan.py:21-314)
on the *inference* path: same class name, same constructor keywords (so the shipped YAMLs'
``generator_params`` construct it unchanged; ``final_scale`` / ``extra_art``, which make
``e2w_hifigan_car.yaml`` fail on the reference class, are accepted and ignored), same parameter
names and shapes (so reference checkpoints ``load_state_dict`` unchanged, weight-norm keys
included), same ``forward(c, spk_id=None, ar=None, ph=None)`` / ``inference`` / ``remove_weight_norm`` /
``apply_weight_norm`` / ``register_stats`` methods.

What differs is where the arithmetic runs: this module owns the parameters only.  ``forward`` hands
device pointers to ``libhificar.so`` (hand-written HIP kernels for gfx950, C ABI in
include/hificar.h).  There is no PyTorch-operator implementation of the network in this package and
no CPU fallback: calling ``forward`` on a CPU tensor, without a GPU, or without the built library
raises.  Under autograd (training) ``forward`` is an autograd node whose forward AND backward run in
libhificar (hificar_forward_train / hificar_backward); the weight-norm re-parametrisation stays in PyTorch's graph.
"""

import ctypes
import logging
import math
import os

import numpy as np
import torch

from .. import _native


def _fold(v: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """w = v * g / ||v||, norm over every dim but 0 (torch.nn.utils.weight_norm(dim=0) semantics,
    which the reference applies at hifigan.py:268-278 and bakes at :256-266)."""
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape)
    return v * (g / norm)


_PARAM_EPOCH = [0]  # bumped whenever a Parameter OBJECT is (re)assigned on any _ConvParams: cached parameter lists are rebuilt when it moves


class _ConvParams(torch.nn.Module):
    """Parameter holder for one Conv1d / ConvTranspose1d of the reference, with or without weight norm.

    Registration order matches torch's weight-normed modules (bias, weight_g, weight_v) so that
    ``state_dict()`` key order equals the reference's.
    """

    def __setattr__(self, name, value):
        if isinstance(value, torch.nn.Parameter) or (value is None and name in self.__dict__.get("_parameters", ())):
            _PARAM_EPOCH[0] += 1  # (a direct ``m.weight = Parameter(...)`` must not leave a stale list in the owning generator)
        super().__setattr__(name, value)

    def __init__(self, weight_shape, n_bias, bias=True, fan_in=None):
        super().__init__()
        self.weight_shape = tuple(weight_shape)
        fan_in = fan_in or int(np.prod(weight_shape[1:]))
        bound = 1.0 / math.sqrt(fan_in)
        if bias:
            self.bias = torch.nn.Parameter(torch.empty(n_bias).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        # torch's default Conv init: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
        self.weight = torch.nn.Parameter(torch.empty(self.weight_shape).uniform_(-bound, bound))

    @property
    def has_weight_norm(self):
        return "weight_g" in self._parameters

    def apply_weight_norm(self):
        if self.has_weight_norm:
            return
        w = self._parameters.pop("weight").detach()
        g = w.reshape(w.shape[0], -1).norm(dim=1).reshape((w.shape[0],) + (1,) * (w.dim() - 1))
        self.weight_g = torch.nn.Parameter(g)
        self.weight_v = torch.nn.Parameter(w.clone())

    def remove_weight_norm(self):
        if not self.has_weight_norm:
            raise ValueError("weight_norm not found")
        w = self.folded_weight()
        del self._parameters["weight_g"]
        del self._parameters["weight_v"]
        self.weight = torch.nn.Parameter(w)

    def folded_weight(self) -> torch.Tensor:
        if self.has_weight_norm:
            return _fold(self.weight_v.detach(), self.weight_g.detach())
        return self.weight.detach()


class _ResBlockParams(torch.nn.Module):
    """Parameters of one HiFiGANResidualBlock (reference articulatory/layers/residual_block.py:141-205)."""

    def __init__(self, kernel_size, channels, dilations, bias, use_additional_convs, slope):
        super().__init__()
        assert kernel_size % 2 == 1, "Kernel size must be odd number."

        def slot():
            return torch.nn.Sequential(torch.nn.LeakyReLU(slope), _ConvParams((channels, channels, kernel_size), channels, bias))

        self.convs1 = torch.nn.ModuleList([slot() for _ in dilations])
        if use_additional_convs:
            self.convs2 = torch.nn.ModuleList([slot() for _ in dilations])


class _PastFCParams(torch.nn.Module):
    """Parameters of PastFCEncoder (reference articulatory/layers/pytorch_layers.py:426-449)."""

    def __init__(self, input_len, hidden_dim, output_dim):
        super().__init__()
        dims = [input_len] + [hidden_dim] * 4 + [output_dim]
        mods = []
        for i in range(5):
            mods.append(torch.nn.Linear(dims[i], dims[i + 1]))
            if i < 4:
                mods.append(torch.nn.LeakyReLU(0.1))
        self.model = torch.nn.Sequential(*mods)


def _send_parameters(module, names, tensors, stream):
    """Hand every RAW parameter over in one call (hificar_set_parameters_device: weight norm folded and every pack refreshed on the
    device, two launches).  Returns the tensors actually read (kept alive by the caller until the stream has consumed them)."""
    lib, handle = module._lib, module._handle
    # Nothing changed since the last hand-over to THIS handle (same tensors at the same addresses with the same version counters, and no
    # invalidate_parameters() in between — which is what a fused optimizer's step triggers): skip the fold + 166 packs.  The graph forward of an
    # iteration sees the weights the previous iteration's second (no-graph) forward already sent (train.py:389: "re-compute y_").
    direct = all(t.dtype == torch.float32 and t.is_contiguous() for t in tensors)
    sig = (id(handle), names, tuple(t.data_ptr() for t in tensors), tuple(t._version for t in tensors))
    if direct and module.__dict__.get("_sent_sig") == sig:
        return module.__dict__["_sent_held"]
    # (the parameters themselves are read when they are fp32 and contiguous — the normal case; a converted copy otherwise)
    held = [t if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().to(torch.float32).contiguous() for t in tensors]
    cnames = getattr(module, "_raw_cnames", None)
    if cnames is None or cnames[0] != names:
        arr = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
        cnames = module._raw_cnames = (names, arr)
    ptrs = (ctypes.c_void_p * len(held))(*[t.data_ptr() for t in held])
    _native.check(lib.hificar_set_parameters_device(handle, cnames[1], ptrs, len(held), stream), "hificar_set_parameters_device")
    if direct:
        module.__dict__["_sent_sig"], module.__dict__["_sent_held"] = sig, held
    else:
        module.__dict__.pop("_sent_sig", None)
    return held


class _GeneratorFunction(torch.autograd.Function):
    """Autograd node of the native generator: forward = hificar_forward_train_cond (keeps a tape), backward = hificar_backward_cond.

    Inputs after (module, c, ar, spk_id, ph, names) are the module's RAW parameters (weight_g / weight_v of the weight-normed convs,
    hifigan.py:268-278; plain weights; biases; the speaker / phoneme embeddings and Linear layers of the conditioned variants,
    hifigan.py:176-189): the fold w = g v / ||v||, every convolution, activation, the PastFCEncoder and the conditioning branches —
    forward and backward, including the weight norm's chain rule — run in libhificar.  Returns the waveform, or (waveform, ph_out)
    for a use_ph_loss model (hifigan.py:232-237)."""

    @staticmethod
    def forward(ctx, module, c, ar, spk_id, ph, names, *params):
        lib, handle = module._lib, module._handle
        B, _, T = c.shape
        dev = c.device
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            held = _send_parameters(module, names, params, stream)
            tape = torch.empty(lib.hificar_tape_bytes(handle, B, T) + 256, dtype=torch.uint8, device=dev)
            toff = (-tape.data_ptr()) % 256
            out = torch.empty((B, 1, T * module.hop), dtype=torch.float32, device=dev)
            ph_out = torch.empty((B, module._params["num_ph"], T), dtype=torch.float32, device=dev) if module.use_ph_loss else None
            ws_ptr, ws_bytes = module._workspace(B, T)
            rc = lib.hificar_forward_train_cond(handle, c.data_ptr(), ar.data_ptr() if ar is not None else None,
                                                spk_id.data_ptr() if spk_id is not None else None, ph.data_ptr() if ph is not None else None,
                                                out.data_ptr(), ph_out.data_ptr() if ph_out is not None else None, B, T,
                                                ws_ptr, ws_bytes, tape.data_ptr() + toff, tape.numel() - toff, stream)
        _native.check(rc, "hificar_forward_train")
        ctx.module, ctx.names, ctx.tape, ctx.toff, ctx.BT = module, names, tape, toff, (B, T)
        ctx.shapes = [tuple(w.shape) for w in params]
        ctx.held = held  # the weight norm's backward reads weight_g / weight_v again
        ctx.versions = [p._version for p in params]
        ctx.params = params
        ctx.has_ar = ar is not None
        ctx.cond = (spk_id, ph)  # the backward reads the same indices again (integer tensors: nothing to differentiate)
        ctx.save_for_backward(out)
        return out if ph_out is None else (out, ph_out)

    @staticmethod
    def backward(ctx, dout, dph_out=None):
        module = ctx.module
        lib, handle = module._lib, module._handle
        (out,) = ctx.saved_tensors
        B, T = ctx.BT
        dev = out.device
        p = module._params
        if any(t._version != v for t, v in zip(ctx.params, ctx.versions)):
            raise RuntimeError("a generator parameter was modified in place between forward and backward (the weight norm's "
                               "gradient reads weight_g / weight_v)")
        dout = (torch.zeros_like(out) if dout is None else dout).to(torch.float32).contiguous()
        dph_out = dph_out.to(torch.float32).contiguous() if dph_out is not None else None
        spk_id, ph = ctx.cond
        need_c, need_ar = ctx.needs_input_grad[1], ctx.has_ar and ctx.needs_input_grad[2]
        cf = p["in_channels"] - (p["ar_output"] if module.use_ar else 0) - (p["ph_emb_size"] if module.use_ph else 0)
        dc = torch.empty((B, cf, T), dtype=torch.float32, device=dev) if need_c else None
        dar = torch.empty((B, 1, p["ar_input"]), dtype=torch.float32, device=dev) if need_ar else None
        with torch.cuda.device(dev):
            grads = torch.zeros(int(lib.hificar_grad_floats(handle)), dtype=torch.float32, device=dev)
            ws = torch.empty(lib.hificar_backward_workspace_bytes(handle, B, T) + 256, dtype=torch.uint8, device=dev)
            woff = (-ws.data_ptr()) % 256
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            raw = torch.zeros(int(lib.hificar_raw_grad_floats(handle)), dtype=torch.float32, device=dev)
            reducer = cb = None
            if module._grad_sync is not None:
                # data-parallel training: the gradients are all-reduced bucket by bucket (RCCL over xGMI under the "nccl" backend) WHILE
                # the backward pass still runs — libhificar calls back when a bucket's gradient kernels are enqueued (last stage first),
                # the bucket's weight-norm chain rule runs right behind them and its collective starts on the communication stream
                from ..utils.buckets import BucketHook, BucketReducer, bucket_ranges

                group, average = module._grad_sync
                nb = int(lib.hificar_grad_bucket_count(handle))
                ids = [int(lib.hificar_raw_param_bucket(handle, i)) for i in range(len(ctx.shapes))]
                ranges, total = bucket_ranges(ids, [int(np.prod(sh)) for sh in ctx.shapes], nb)
                assert total == raw.numel()

                def chain_rule(bucket, bstream):  # (bstream is the current stream here: the generator's backward runs on one stream)
                    _native.check(lib.hificar_weight_norm_backward_bucket(handle, grads.data_ptr(), raw.data_ptr(), bucket, ctypes.c_void_p(bstream)),
                                  "hificar_weight_norm_backward_bucket")

                reducer = BucketHook(BucketReducer(raw, ranges, group, average), chain_rule)
                cb = _native.BUCKET_FN(reducer)
                _native.check(lib.hificar_set_bucket_callback(handle, cb, None), "hificar_set_bucket_callback")
            try:
                rc = lib.hificar_backward_cond(handle, dout.data_ptr(), dph_out.data_ptr() if dph_out is not None else None, out.data_ptr(),
                                               spk_id.data_ptr() if spk_id is not None else None, ph.data_ptr() if ph is not None else None,
                                               B, T, ctx.tape.data_ptr() + ctx.toff, ctx.tape.numel() - ctx.toff, grads.data_ptr(),
                                               dc.data_ptr() if dc is not None else None, dar.data_ptr() if dar is not None else None,
                                               ws.data_ptr() + woff, ws.numel() - woff, stream)
            finally:
                if cb is not None:
                    lib.hificar_set_bucket_callback(handle, _native.BUCKET_FN(), None)
            _native.check(rc, "hificar_backward")
            if reducer is not None:
                reducer.finish()  # (re-raises what a bucket callback caught)
            else:
                _native.check(lib.hificar_weight_norm_backward(handle, grads.data_ptr(), raw.data_ptr(), stream), "hificar_weight_norm_backward")
        gw, off = [], 0
        for shape in ctx.shapes:
            n = int(np.prod(shape))
            gw.append(raw[off:off + n].view(shape))
            off += (n + 3) & ~3
        ctx.tape = ctx.held = ctx.params = ctx.cond = None
        return (None, dc, dar, None, None, None, *gw)


class _NoGraph(torch.autograd.Function):
    """Marks the output of an eval-mode forward that ran on the inference kernels (no tape): backward raises a message that says why."""

    @staticmethod
    def forward(ctx, out, name, _witness):  # _witness: one trainable parameter, so that the output is part of a graph at all
        ctx.name = name
        return out.view_as(out)

    @staticmethod
    def backward(ctx, grad):
        raise RuntimeError(f"{ctx.name}: this forward ran on the inference kernels because the module is in eval() mode (no activations were "
                           "kept for a backward pass).  Call .train(), or set `module.eval_autograd = True` to get torch's semantics "
                           "(a graph whenever gradients are enabled), before the forward whose gradients you need.")


class _NativeGenerator(torch.nn.Module):
    """What the generator classes of this package share: parameter plumbing (weight norm, folded state, the raw-parameter hand-over), the
    libhificar handle, workspaces, the inference / AR-synthesis / autograd entry points.  A subclass builds the parameter-holder modules with
    the reference's names in ``__init__``, fills ``self._params`` and implements ``_create_handle``."""

    # ------------------------------------------------------------------ parameter plumbing
    def _conv_params(self):
        return [m for m in self.modules() if isinstance(m, _ConvParams)]

    def reset_parameters(self):
        """N(0, 0.01) on conv weights (hifigan.py:241-254).  With weight norm on this has no lasting
        effect in the reference (the hook recomputes ``weight`` from g/v), so it is only applied to
        un-normalised weights here."""
        for m in self._conv_params():
            if not m.has_weight_norm:
                m.weight.data.normal_(0.0, 0.01)
        self._invalidate()

    def remove_weight_norm(self):
        """Bake w = v*g/||v|| into ``weight`` for every conv layer (hifigan.py:256-266)."""
        for m in self._conv_params():
            if m.has_weight_norm:
                logging.debug(f"Weight norm is removed from {m}.")
                m.remove_weight_norm()
        self._invalidate()

    def apply_weight_norm(self):
        """Re-parametrise every conv weight as (weight_g, weight_v) (hifigan.py:268-278)."""
        for m in self._conv_params():
            m.apply_weight_norm()
        self._invalidate()

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._invalidate()
        return out

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._invalidate()
        return out

    def register_stats(self, stats):
        """Register mean/scale buffers for input normalisation (hifigan.py:280-296)."""
        assert stats.endswith(".h5") or stats.endswith(".npy")
        if stats.endswith(".h5"):
            from ..utils.hdf5 import read_hdf5  # h5py when importable, else the built-in reader of plain HDF5 files

            mean = read_hdf5(stats, "mean").reshape(-1)
            scale = read_hdf5(stats, "scale").reshape(-1)
        else:
            arr = np.load(stats)
            mean = arr[0].reshape(-1)
            scale = arr[1].reshape(-1)
        self.register_buffer("mean", torch.from_numpy(np.asarray(mean)).float())
        self.register_buffer("scale", torch.from_numpy(np.asarray(scale)).float())
        logging.info("Successfully registered stats as buffer.")

    def folded_state(self):
        """{reference post-remove_weight_norm key: fp32 CPU tensor} — what the C ABI consumes."""
        out = {}
        for name, m in self.named_modules():
            if isinstance(m, _ConvParams):
                out[name + ".weight"] = m.folded_weight().float().cpu().contiguous()
                if m.bias is not None:
                    out[name + ".bias"] = m.bias.detach().float().cpu().contiguous()
            elif isinstance(m, torch.nn.Linear):
                out[name + ".weight"] = m.weight.detach().float().cpu().contiguous()
                out[name + ".bias"] = m.bias.detach().float().cpu().contiguous()
            elif isinstance(m, torch.nn.Embedding):
                out[name + ".weight"] = m.weight.detach().float().cpu().contiguous()
        return out

    # ------------------------------------------------------------------ copies and pickles
    # copy.deepcopy / pickle build a module WITHOUT running __init__: the native handle (a raw pointer: two modules must never share one),
    # the workspaces, the cached parameter lists (they would point at the ORIGINAL's tensors) and the gradient-sync wiring stay behind, and
    # the copy registers with the optimizer post-step hook itself — a deep-copied generator trained with a fused optimizer (no
    # Parameter._version bump) would otherwise keep running on the weights it was copied with.
    _EPHEMERAL = ("_handle", "_lib", "_workspaces", "_grad_slots", "_grad_sync", "_param_sig", "_plist_cache", "_plist_epoch", "_raw_cache", "_sent_sig",
                  "_sent_held", "_raw_cnames")

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in self._EPHEMERAL:
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        d = self.__dict__
        d["_handle"] = d["_lib"] = d["_grad_slots"] = d["_grad_sync"] = d["_param_sig"] = d["_plist_cache"] = d["_raw_cache"] = None
        d["_workspaces"] = {}
        from ..utils.optim_hook import watch

        watch(self)

    # ------------------------------------------------------------------ native handle
    def _invalidate(self):
        h = getattr(self, "_handle", None)
        if h is not None and self._lib is not None:
            self._lib.hificar_destroy(h)
        self._handle = None
        self._workspaces = {}
        self._grad_slots = None
        self.__dict__["_plist_cache"] = None
        self.__dict__.pop("_sent_sig", None)
        self.__dict__.pop("_sent_held", None)

    def __del__(self):
        try:
            self._invalidate()
        except Exception:
            pass

    def refresh_native(self):
        """Re-upload weights after an in-place parameter edit (e.g. ``p.data.copy_``)."""
        self._invalidate()

    def invalidate_parameters(self):
        """The parameters were updated in a way their version counters do not show — ``p.data`` writes, and torch's FUSED optimizers
        (``torch.optim.Adam(fused=True)`` updates in place without bumping ``_version``): the next forward hands them over again.
        The module registers itself with ``articulatory_amd.utils.optim_hook.watch`` at construction, so every ``optimizer.step()`` of an
        optimizer that holds one of its parameters calls this — no wiring by the training loop."""
        self._param_sig = None
        self.__dict__.pop("_sent_sig", None)

    def set_precision(self, precision):
        if precision not in _native.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_native.PRECISIONS)}")
        self.precision = precision
        if self._handle is not None:
            _native.check(self._lib.hificar_set_precision(self._handle, _native.PRECISIONS[precision]), "hificar_set_precision")

    def _device(self):
        return next(self.parameters()).device

    def _plist(self):
        """The parameters as a flat list, cached: walking the module tree (self.parameters()) costs ~1 ms per call on this model, and the
        training forward needs the list three times.  Rebuilt whenever a parameter OBJECT may have changed: _invalidate (.to(), weight
        norm applied / removed, load_state_dict) and any Parameter assignment on a conv holder (``_PARAM_EPOCH``).  Replacing the Parameter
        object of one of torch's own sub-modules (the PastFCEncoder's Linear layers, the embeddings) by assignment is not seen: call
        ``refresh_native()`` after such surgery."""
        lst = self.__dict__.get("_plist_cache")
        if lst is None or self.__dict__.get("_plist_epoch") != _PARAM_EPOCH[0]:
            lst = self.__dict__["_plist_cache"] = list(self.parameters())
            self.__dict__["_plist_epoch"] = _PARAM_EPOCH[0]
            self.__dict__["_raw_cache"] = None
        return lst

    def _param_signature(self):
        return tuple(p._version for p in self._plist())

    def _native_handle(self):
        if self._handle is not None:
            if getattr(self, "_param_sig", None) != self._param_signature():
                # parameters were updated in place since the weights were handed over (optimizer.step(), p.data.copy_): re-send them
                if self.precision == "f32":
                    names, tensors = self._raw_parameters()
                    dev = self._device()
                    with torch.no_grad(), torch.cuda.device(dev):
                        _send_parameters(self, names, tensors, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                    self._param_sig = self._param_signature()
                    return self._handle
                self._invalidate()
            else:
                return self._handle
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError(
                "%s: parameters are on %s; the generator forward only exists as HIP kernels "
                "(move the model to a MI355X with .to('cuda')). There is no CPU fallback." % (type(self).__name__, dev))
        lib = _native.load_library()
        self._lib = lib
        handle = ctypes.c_void_p()
        with torch.cuda.device(dev):
            self._create_handle(lib, handle)
            try:
                for name, t in self.folded_state().items():
                    shape = (ctypes.c_int64 * t.dim())(*t.shape)
                    _native.check(lib.hificar_set_weight(handle, name.encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim()),
                                  "hificar_set_weight")
                _native.check(lib.hificar_finalize(handle), "hificar_finalize")
            except Exception:
                lib.hificar_destroy(handle)
                raise
        self._handle = handle
        self._param_sig = self._param_signature()
        return handle

    def _workspace(self, B, T):
        """One grow-only scratch buffer per model (the library plans its layout per call): a dataset of many distinct
        utterance lengths re-uses it instead of allocating per shape."""
        n = self._lib.hificar_workspace_bytes(self._handle, B, T) + 256
        ws = self._workspaces.get("buf")
        if ws is None or ws.numel() < n:
            # work already enqueued on the old buffer keeps it alive through the caching allocator's stream ordering
            ws = torch.empty(int(n * 1.25) if ws is not None else n, dtype=torch.uint8, device=self._device())
            self._workspaces["buf"] = ws
        off = (-ws.data_ptr()) % 256
        return ws.data_ptr() + off, ws.numel() - off

    def macs(self, B, T):
        """Algorithmic multiply-accumulates of one forward of B x T frames (SURVEY.md §8d constant)."""
        self._native_handle()
        return float(self._lib.hificar_macs(self._handle, B, T))

    def profile_begin(self):
        """Start bracketing every kernel launch with HIP events (bench.py roofline leg)."""
        self._native_handle()
        _native.check(self._lib.hificar_profile_begin(self._handle), "hificar_profile_begin")

    def profile_end(self):
        """Stop profiling; returns [{name, launches, total_ms, flops, bytes}], slowest kernel first."""
        stats = (_native.HificarKernelStat * 96)()
        n = ctypes.c_int(0)
        _native.check(self._lib.hificar_profile_end(self._handle, stats, 96, ctypes.byref(n)), "hificar_profile_end")
        return [dict(name=stats[i].name.decode(), launches=int(stats[i].launches), total_ms=float(stats[i].total_ms),
                     flops=float(stats[i].flops), bytes=float(stats[i].bytes)) for i in range(min(n.value, 96))]

    def debug_taps(self, names, c, ar=None, spk_id=None):
        """Parity aid (C ABI: hificar_debug_tap): one forward that also returns the named per-layer intermediates in the
        reference's (B, C, L) layout — what forward hooks on the reference's modules see.  Names: "ar_feats", "input_conv",
        "upsamples.<i>", "blocks.<n>.convs1.<d>", "blocks.<n>.x.<d>" (residual stream after dilation d), "blocks.<n>".
        Returns (out, {name: tensor})."""
        handle = self._native_handle()
        B, _, T = c.shape
        Tb = T if T <= 32 else -(-T // 32) * 32  # the launch covers a bucket of frames (hificar.hip: bucket_frames); taps hold that many rows
        taps, keep = {}, {}
        for name in names:
            shape, rows = self._tap_shape(name, B, T, Tb)
            if rows is not None:
                keep[name] = rows
            taps[name] = torch.full(shape, float("nan"), dtype=torch.float32, device=c.device)
        try:
            for name, t in taps.items():
                _native.check(self._lib.hificar_debug_tap(handle, name.encode(), t.data_ptr(), t.numel()), "hificar_debug_tap")
            out = self.forward(c, spk_id=spk_id, ar=ar)
            torch.cuda.synchronize(c.device)
        finally:
            self._lib.hificar_debug_tap(handle, None, None, 0)
        return out, {name: (t[..., :keep[name]] if name in keep else t) for name, t in taps.items()}

    # ------------------------------------------------------------------ forward paths
    def sync_gradients(self, group=None, average=True, enabled=True):
        """Data-parallel training (the reference's DDP wrap is disabled, articulatory/bin/train.py:1790-1801): all-reduce the
        generator's gradients across the ranks of ``group`` inside backward — one collective on the flat gradient buffer (the
        whole generator as a single bucket), averaged like DistributedDataParallel does.  Needs an initialised process group."""
        self._grad_sync = (group, bool(average)) if enabled else None
        return self

    def _grad_layout(self):
        """{folded parameter name: (offset, numel)} inside the flat gradient buffer hificar_backward fills."""
        if getattr(self, "_grad_slots", None) is None:
            n = self._lib.hificar_grad_count(self._handle)
            if n < 0:
                _native.check(-1, "hificar_grad_count")
            slots = {}
            name = ctypes.create_string_buffer(96)
            off, cnt = ctypes.c_int64(), ctypes.c_int64()
            for i in range(n):
                _native.check(self._lib.hificar_grad_info(self._handle, i, name, ctypes.byref(off), ctypes.byref(cnt)), "hificar_grad_info")
                slots[name.value.decode()] = (off.value, cnt.value)
            self._grad_slots = slots
        return self._grad_slots

    def _raw_parameters(self):
        """(names, tensors): every parameter as it sits in this module (state_dict keys: weight_g / weight_v of weight-normed convs,
        plain weights, biases) — the form hificar_set_parameters_device consumes."""
        self._plist()  # (drops the cache below when a Parameter object was re-assigned)
        cache = self.__dict__.get("_raw_cache")
        if cache is not None:
            cur = list(cache[2]())
            if len(cur) == len(cache[1]) and all(t is p for t, p in zip(cache[1], cur)):
                return cache[0], cache[1]

        def current():
            for m in mods:
                if isinstance(m, _ConvParams):
                    if m.has_weight_norm:
                        yield m._parameters.get("weight_g")
                        yield m._parameters.get("weight_v")
                    else:
                        yield m._parameters.get("weight")
                    if m.bias is not None:
                        yield m.bias
                elif isinstance(m, torch.nn.Embedding):
                    yield m.weight
                else:
                    yield m.weight
                    yield m.bias

        names, mods = [], []
        for name, m in self.named_modules():
            if isinstance(m, _ConvParams):
                mods.append(m)
                names += [name + ".weight_g", name + ".weight_v"] if m.has_weight_norm else [name + ".weight"]
                if m.bias is not None:
                    names.append(name + ".bias")
            elif isinstance(m, torch.nn.Linear):
                mods.append(m)
                names += [name + ".weight", name + ".bias"]
            elif isinstance(m, torch.nn.Embedding):
                mods.append(m)
                names.append(name + ".weight")
        tensors = list(current())
        self.__dict__["_raw_cache"] = (tuple(names), tensors, current)
        return self.__dict__["_raw_cache"][0], tensors

    def _forward_autograd(self, c, ar, spk_id=None, ph=None):
        """Training-mode forward (train.py:276,398: y_ = generator(x, spk_id=spk_id, ar=ar, ph=ph) under autograd)."""
        if self.precision != "f32":
            raise RuntimeError("training runs in the exact-fp32 arithmetic: construct with precision='f32'")
        if self._handle is None:
            self._native_handle()
        names, tensors = self._raw_parameters()
        out = _GeneratorFunction.apply(self, c, ar, spk_id, ph, names, *tensors)
        self._param_sig = self._param_signature()  # the forward above handed the current weights over
        return out

    def _check_input(self, c):
        if not c.is_cuda:
            raise RuntimeError(f"{type(self).__name__}.forward needs a CUDA/HIP tensor; there is no CPU fallback")
        cf = (self._params["in_channels"] - (self._params["ar_output"] if self.use_ar else 0)
              - (self._params["ph_emb_size"] if self.use_ph else 0))
        if c.dim() != 3 or c.shape[1] != cf:
            raise RuntimeError(f"Expected input of shape (B, {cf}, T), got {tuple(c.shape)}")

    def _lengths_arg(self, lengths, B, T, device):
        """lengths (sequence / tensor of B frame counts) -> int32 device tensor for the ragged entry points."""
        if isinstance(lengths, torch.Tensor):
            host = lengths.detach().to("cpu", torch.int32).reshape(-1).contiguous()  # the C ABI reads the host copy on the host
        else:
            host = torch.as_tensor(lengths, dtype=torch.int32).reshape(-1).contiguous()
        if host.numel() != B:
            raise RuntimeError(f"lengths has {host.numel()} entries for a batch of {B}")
        if B and (int(host.min()) < 0 or int(host.max()) > T):
            raise RuntimeError(f"lengths must lie in [0, {T}]")
        return host, host.to(device).contiguous()  # (host copy, device copy)

    def forward(self, c, spk_id=None, ar=None, ph=None, lengths=None):
        """c: (B, in_channels[-ar_output][-ph_emb_size], T) -> (B, out_channels, T * prod(upsample_scales))  (hifigan.py:198-239);
        with use_ph_loss the reference's pair (out, ph_out), ph_out: (B, num_ph, T).  spk_id: (B,) speaker indices (use_spk_id);
        ph: (B, T) phoneme indices (use_ph).

        ``lengths`` (not in the reference, which is batch-1 at inference): frame counts of a padded batch of utterances of
        different lengths; utterance b is computed exactly as if it were alone and out[b, :, hop*lengths[b]:] is zero."""
        self._check_input(c)
        if self.use_ar:
            if ar is None:
                raise RuntimeError("use_ar=True: forward() needs ar=(B, out_channels, ar_input/out_channels) past samples")
            if ar.numel() != c.shape[0] * self._params["ar_input"]:
                raise RuntimeError(f"ar has {ar.numel()} elements, expected {c.shape[0]}x{self._params['ar_input']}")
            ar = ar.to(device=c.device, dtype=torch.float32).contiguous()
        c = c.to(torch.float32).contiguous()
        B, _, T = c.shape
        if self.use_spk_id:
            if spk_id is None or spk_id.numel() != B:
                raise RuntimeError("use_spk_id=True: forward() needs spk_id=(B,) speaker indices")
            if int(spk_id.min()) < 0 or int(spk_id.max()) >= self._params["num_spk"]:
                raise IndexError("index out of range in self")  # torch.nn.Embedding's message
            spk_id = spk_id.to(device=c.device, dtype=torch.int32).contiguous()
        if self.use_ph:
            if ph is None or tuple(ph.shape) != (B, T):
                raise RuntimeError(f"use_ph=True: forward() needs ph=(B, T)=({B}, {T}) phoneme indices")
            if int(ph.min()) < 0 or int(ph.max()) >= self._params["num_ph"]:
                raise IndexError("index out of range in self")
            ph = ph.to(device=c.device, dtype=torch.int32).contiguous()
        # the autograd node (and its full-utterance tape) when a gradient can be asked for: an input that requires grad, or trainable
        # parameters of a module in training mode (or with ``eval_autograd = True``).  model.eval() inference without torch.no_grad() stays
        # on the inference kernels; its output then carries a node whose backward explains that instead of torch's bare "element 0 of
        # tensors does not require grad" (plain torch modules build a graph in eval mode too: set ``eval_autograd`` for that)
        trainable = torch.is_grad_enabled() and any(p.requires_grad for p in self._plist())
        if torch.is_grad_enabled() and (c.requires_grad or (ar is not None and ar.requires_grad)
                                        or (trainable and (self.training or getattr(self, "eval_autograd", False)))):
            if lengths is not None:
                raise NotImplementedError("autograd with ragged lengths is not built")
            return self._forward_autograd(c, ar, spk_id if self.use_spk_id else None, ph if self.use_ph else None)
        handle = self._native_handle()
        if lengths is None:
            out = torch.empty((B, 1, T * self.hop), dtype=torch.float32, device=c.device)
        else:
            _, lengths = self._lengths_arg(lengths, B, T, c.device)
            out = torch.zeros((B, 1, T * self.hop), dtype=torch.float32, device=c.device)
        ph_out = torch.zeros((B, self._params["num_ph"], T), dtype=torch.float32, device=c.device) if self.use_ph_loss else None
        with torch.cuda.device(c.device):
            ws_ptr, ws_bytes = self._workspace(B, T)
            stream = torch.cuda.current_stream().cuda_stream
            rc = self._lib.hificar_forward_cond(handle, c.data_ptr(), ar.data_ptr() if self.use_ar else None,
                                                spk_id.data_ptr() if self.use_spk_id else None, ph.data_ptr() if self.use_ph else None,
                                                lengths.data_ptr() if lengths is not None else None, out.data_ptr(),
                                                ph_out.data_ptr() if ph_out is not None else None,
                                                B, T, ws_ptr, ws_bytes, ctypes.c_void_p(stream))
        _native.check(rc, "hificar_forward")
        if trainable:  # eval mode with trainable parameters: no tape was kept — say so if somebody calls backward on this
            out = _NoGraph.apply(out, type(self).__name__, next(p for p in self._plist() if p.requires_grad))
        return (out, ph_out) if self.use_ph_loss else out

    def ar_synthesis(self, c, chunk_frames, lengths=None):
        """Batched autoregressive synthesis on device.

        c: (B, C, T_total) features; returns (B, hop*T_total).  Per utterance this equals the
        reference's ``ar_loop`` (articulatory/bin/decode.py:54-83) with
        ``chunk_frames = batch_max_steps // hop_size``; the reference driver is batch-1 only.
        ``lengths``: frame counts of a padded batch of utterances of different lengths (see ``forward``); each utterance's
        last chunk is then its own shorter tail chunk, as in the reference loop.
        """
        if not self.use_ar:
            raise RuntimeError("ar_synthesis needs a use_ar=True generator")
        self._check_input(c)
        c = c.to(torch.float32).contiguous()
        B, _, T = c.shape
        handle = self._native_handle()
        if lengths is None:
            out = torch.empty((B, T * self.hop), dtype=torch.float32, device=c.device)
        else:
            lengths_host, lengths = self._lengths_arg(lengths, B, T, c.device)
            out = torch.zeros((B, T * self.hop), dtype=torch.float32, device=c.device)
        with torch.cuda.device(c.device):
            ws_ptr, ws_bytes = self._workspace(B, min(int(chunk_frames), T))
            stream = torch.cuda.current_stream().cuda_stream
            rc = self._lib.hificar_ar_loop_ragged(handle, c.data_ptr(), lengths.data_ptr() if lengths is not None else None,
                                                  lengths_host.data_ptr() if lengths is not None else None,
                                                  out.data_ptr(), B, T, int(chunk_frames), ws_ptr, ws_bytes, ctypes.c_void_p(stream))
        _native.check(rc, "hificar_ar_loop")
        return out

    def ar_synthesis_packed(self, c, chunk_frames, lengths, batch=64):
        """Continuously batched autoregressive synthesis of a list of utterances (C ABI: hificar_ar_loop_packed).

        c: (N, C, T_max) zero-padded features, ``lengths`` their N frame counts; at most ``batch`` utterances are in flight
        and a finished one is replaced by the next of the list.  Returns (N, hop*T_max) with zeros past each utterance's
        end; per utterance the reference's ``ar_loop`` result (articulatory/bin/decode.py:54-83)."""
        if not self.use_ar:
            raise RuntimeError("ar_synthesis needs a use_ar=True generator")
        self._check_input(c)
        c = c.to(torch.float32).contiguous()
        N, _, T = c.shape
        lengths_host, _ = self._lengths_arg(lengths, N, T, c.device)
        handle = self._native_handle()
        out = torch.zeros((N, T * self.hop), dtype=torch.float32, device=c.device)
        batch = max(1, min(int(batch), N))
        with torch.cuda.device(c.device):
            ws_ptr, ws_bytes = self._workspace(batch, min(int(chunk_frames), T))
            stream = torch.cuda.current_stream().cuda_stream
            rc = self._lib.hificar_ar_loop_packed(handle, c.data_ptr(), lengths_host.data_ptr(), out.data_ptr(), N, T,
                                                  int(chunk_frames), batch, ws_ptr, ws_bytes, ctypes.c_void_p(stream))
        _native.check(rc, "hificar_ar_loop_packed")
        return out

    def inference(self, c, normalize_before=False):
        """(T, in_channels) -> (T * prod(upsample_scales), out_channels)  (hifigan.py:298-314)."""
        if not isinstance(c, torch.Tensor):
            c = torch.tensor(c, dtype=torch.float).to(self._device())
        if normalize_before:
            c = (c - self.mean) / self.scale
        c = self.forward(c.transpose(1, 0).unsqueeze(0))
        return c.squeeze(0).transpose(1, 0)


class HiFiGANGenerator(_NativeGenerator):
    """HiFiGAN generator module (MI355X-native forward and backward)."""

    def __init__(
        self,
        in_channels=80,
        out_channels=1,
        channels=512,
        kernel_size=7,
        upsample_scales=(8, 8, 2, 2),
        upsample_kernel_sizes=(16, 16, 4, 4),
        paddings=None,
        output_paddings=None,
        resblock_kernel_sizes=(3, 7, 11),
        resblock_dilations=[(1, 3, 5), (1, 3, 5), (1, 3, 5)],
        use_additional_convs=True,
        bias=True,
        nonlinear_activation="LeakyReLU",
        nonlinear_activation_params={"negative_slope": 0.1},
        use_weight_norm=True,
        use_ar=False,
        ar_input=512,
        ar_hidden=256,
        ar_output=128,
        use_tanh=True,
        use_spk_id=False,
        num_spk=None,
        spk_emb_size=32,
        use_ph=False,
        num_ph=None,
        ph_emb_size=8,
        use_ph_loss=False,
        final_scale=None,  # present in e2w_hifigan_car.yaml:42; unused by the network
        extra_art=None,  # present in e2w_hifigan_car.yaml:54; only read by the WSOLA driver
        precision=None,  # "f32" (default: the reference's IEEE fp32 products; or $HIFICAR_PRECISION) | "bf16x3" (opt-in fast mode,
                         # 16-bit-significand products): conv arithmetic, see DESIGN.md §3
    ):
        super().__init__()
        # same validity checks as the reference (hifigan.py:78-80)
        assert kernel_size % 2 == 1, "Kernel size must be odd number."
        assert len(upsample_scales) == len(upsample_kernel_sizes)
        assert len(resblock_dilations) == len(resblock_kernel_sizes)
        if use_spk_id and use_ph:
            # spk_fc maps to in_channels values but is added before the phoneme channels are appended (hifigan.py:212-220):
            # the reference fails with a shape mismatch in forward; fail at construction here
            raise ValueError("use_spk_id together with use_ph is ill-formed in the reference (shape mismatch at hifigan.py:216)")
        # the reference builds getattr(torch.nn, nonlinear_activation)(**nonlinear_activation_params) (hifigan.py:121-123, 142-143).  The kernels'
        # activation is max(x, slope * x) with 0 <= slope <= 1: LeakyReLU, and with it ReLU (slope 0) and Identity (slope 1); other modules are not built
        nonlinear_activation_params = dict(nonlinear_activation_params or {})
        if nonlinear_activation_params.get("inplace"):
            # in the reference an in-place activation CHANGES the result: convs1[idx](x) activates x itself, so the residual `xt + x` adds the
            # activated x and the first block's activation overwrites the `c` every block of a stage shares (residual_block.py:217-221,
            # hifigan.py:226-230).  That arithmetic is not built: refuse instead of silently computing the out-of-place network.
            raise NotImplementedError("nonlinear_activation_params['inplace']=True changes the reference's result (the residual adds the activated "
                                      "input); only inplace=False (or no such key) is built")
        if nonlinear_activation == "ReLU":
            if set(nonlinear_activation_params) - {"inplace"}:
                raise ValueError(f"ReLU takes no parameters besides inplace: {nonlinear_activation_params}")
            nonlinear_activation_params = {"negative_slope": 0.0}
        elif nonlinear_activation == "Identity":
            nonlinear_activation_params = {"negative_slope": 1.0}
        elif nonlinear_activation != "LeakyReLU":
            raise NotImplementedError(f"nonlinear_activation={nonlinear_activation!r}: LeakyReLU, ReLU and Identity are built")
        elif set(nonlinear_activation_params) - {"negative_slope", "inplace"}:
            raise ValueError(f"LeakyReLU parameters: {nonlinear_activation_params}")
        for name, val in (("paddings", paddings), ("output_paddings", output_paddings)):
            if val is not None and any(v != "default" for v in val):
                raise NotImplementedError(f"{name}: only None / 'default' entries are supported (as in the reference)")
        if precision is None:
            precision = os.environ.get("HIFICAR_PRECISION", "f32")
        if precision not in _native.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_native.PRECISIONS)}")

        self.use_ar = use_ar
        self.use_spk_id = use_spk_id
        self.use_ph = use_ph
        self.use_ph_loss = use_ph_loss
        self.num_upsamples = len(upsample_kernel_sizes)
        self.num_blocks = len(resblock_kernel_sizes)
        slope = float(nonlinear_activation_params.get("negative_slope", 0.01))
        self._params = dict(
            in_channels=in_channels, out_channels=out_channels, channels=channels, kernel_size=kernel_size,
            upsample_scales=list(upsample_scales), upsample_kernel_sizes=list(upsample_kernel_sizes),
            resblock_kernel_sizes=list(resblock_kernel_sizes), resblock_dilations=[list(d) for d in resblock_dilations],
            use_additional_convs=use_additional_convs, bias=bias,
            nonlinear_activation_params={"negative_slope": slope}, use_tanh=use_tanh,
            use_ar=use_ar, ar_input=ar_input, ar_hidden=ar_hidden, ar_output=ar_output,
            use_spk_id=use_spk_id, num_spk=num_spk, spk_emb_size=spk_emb_size, use_ph=use_ph, num_ph=num_ph,
            ph_emb_size=ph_emb_size, use_ph_loss=use_ph_loss,
        )
        self.hop = int(np.prod(upsample_scales))
        self.precision = precision

        self.input_conv = _ConvParams((channels, in_channels, kernel_size), channels)
        self.upsamples = torch.nn.ModuleList()
        self.blocks = torch.nn.ModuleList()
        for i in range(self.num_upsamples):
            cin, cout = channels // (2 ** i), channels // (2 ** (i + 1))
            k = upsample_kernel_sizes[i]
            # ConvTranspose1d weight is (Cin, Cout, K); torch computes its fan_in from dim 1
            self.upsamples.append(torch.nn.Sequential(torch.nn.LeakyReLU(slope), _ConvParams((cin, cout, k), cout, fan_in=cout * k)))
            for j in range(self.num_blocks):
                self.blocks.append(_ResBlockParams(resblock_kernel_sizes[j], cout, resblock_dilations[j], bias,
                                                   use_additional_convs, slope))
        c_last = channels // (2 ** self.num_upsamples)
        out_mods = [torch.nn.LeakyReLU(), _ConvParams((out_channels, c_last, kernel_size), out_channels)]
        if use_tanh:
            out_mods.append(torch.nn.Tanh())
        self.output_conv = torch.nn.Sequential(*out_mods)
        if use_ar:
            self.ar_model = _PastFCParams(ar_input, ar_hidden, ar_output)
        # speaker / phoneme conditioning parameters (hifigan.py:176-189), same names and registration order as the reference
        if use_spk_id:
            assert num_spk is not None
            self.spk_emb_mat = torch.nn.Embedding(num_spk, spk_emb_size)
            self.spk_fc = torch.nn.Linear(spk_emb_size, in_channels)
        if use_ph:
            assert num_ph is not None
            self.ph_emb_mat = torch.nn.Embedding(num_ph, ph_emb_size)
        if use_ph_loss:
            assert num_ph is not None
            assert self.hop % 2 == 0
            self.ph_fc = torch.nn.Linear(c_last, num_ph)

        if use_weight_norm:
            self.apply_weight_norm()
        else:
            self.reset_parameters()

        self._handle = None
        self._workspaces = {}
        self._lib = None
        self._grad_sync = None
        from ..utils.optim_hook import watch

        watch(self)  # fused optimizers do not bump Parameter._version: every optimizer.step() over these parameters invalidates the hand-over

    def _tap_shape(self, name, B, T, Tb):
        """(buffer shape, valid rows or None) of a debug tap; Tb = the launch's bucket of frames."""
        p = self._params
        parts = name.split(".")
        if name == "ar_feats":
            return (B, p["ar_output"]), None
        if name == "input_conv":
            return (B, p["channels"], Tb), T
        if parts[0] == "upsamples":
            stage = int(parts[1])
        elif parts[0] == "blocks":
            stage = int(parts[1]) // self.num_blocks
        else:
            raise ValueError(f"unknown tap {name!r}")
        up = int(np.prod(p["upsample_scales"][:stage + 1]))
        return (B, p["channels"] // (2 ** (stage + 1)), Tb * up), T * up

    def _create_handle(self, lib, handle):
        cfg = _native.make_config(self._params, _native.PRECISIONS[self.precision])
        _native.check(lib.hificar_create(ctypes.byref(cfg), ctypes.byref(handle)), "hificar_create")
