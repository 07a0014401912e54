"""``HiFiGANMultiScaleMultiPeriodDiscriminator`` — the reference's ``discriminator_type`` plugin for the HiFi-GAN / HiFi-CAR recipes
(articulatory/models/hifigan.py:741-825; scale discriminators :503-738, period discriminators :317-500), with the reference's
constructor arguments and ``state_dict`` layout, running on libhificar's discriminator engine (csrc/hificar_disc.hip.inc): forward AND
backward, weight norm folded and differentiated on the device.  There is no CPU / PyTorch-ops fallback.

``forward(x)`` returns the reference's list (scales first, then periods) of lists of layer outputs in the reference's shapes.
``forward(x, native=True)`` returns the same nesting with every layer output as a ``DiscOutput``: the engine's own buffers (one per
conv group, ``[nseq][rows][pitch]`` with ``channels`` valid columns) without any re-layout — what the loss functions in
``articulatory_amd.losses`` consume (means over elements do not depend on the order of the elements)."""
import ctypes

import numpy as np
import torch

from .. import _native
from ..utils.synth import disc_params, period_disc_layers, scale_disc_layers
from .hifigan import _ConvParams


class DiscOutput:
    """One layer output of one sub-discriminator in the engine's layout."""

    __slots__ = ("groups", "channels", "period", "B")

    def __init__(self, groups, channels, period, B):
        self.groups, self.channels, self.period, self.B = groups, channels, period, B  # groups: [(nseq, rows, pitch) tensors]

    def valid(self):
        """The valid part of every group buffer: [(nseq, rows, channels) views]."""
        return [g[:, :, : self.channels] for g in self.groups]

    def numel(self):
        return sum(g.shape[0] * g.shape[1] for g in self.groups) * self.channels

    def reference_layout(self):
        """(B, C, L) for a scale discriminator, (B, C, H, P) for a period discriminator (a copy)."""
        x = torch.cat(self.valid(), dim=2)  # (nseq, rows, C)
        if self.period == 0:
            return x.permute(0, 2, 1).contiguous()
        nseq, rows, C = x.shape
        return x.view(self.B, self.period, rows, C).permute(0, 3, 2, 1).contiguous()


def _send(module, names, tensors, stream):
    """Hand the raw parameters over (weight norm folded, every pack refreshed: ~0.9 ms for the 70 M-parameter discriminator) — skipped
    when nothing changed since the last hand-over: the generator part and the discriminator part of one iteration see the same weights."""
    sig = (names, tuple(t.data_ptr() for t in tensors), tuple(t._version for t in tensors))
    if module.__dict__.get("_sent_sig") == sig and all(t.dtype == torch.float32 and t.is_contiguous() for t in tensors):
        return module.__dict__["_sent_held"]
    held = [t.detach() if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().to(torch.float32).contiguous() for t in tensors]
    cn = getattr(module, "_raw_cnames", None)
    if cn is None or cn[0] != names:
        cn = module._raw_cnames = (names, (ctypes.c_char_p * len(names))(*[n.encode() for n in names]))
    ptrs = (ctypes.c_void_p * len(held))(*[t.data_ptr() for t in held])
    _native.check(module._lib.hificar_disc_set_parameters_device(module._handle, cn[1], ptrs, len(held), stream),
                  "hificar_disc_set_parameters_device")
    module.__dict__["_sent_sig"], module.__dict__["_sent_held"] = sig, held
    return held


class _DiscFunction(torch.autograd.Function):
    """Autograd node of the native discriminators: forward = hificar_disc_forward (every layer output, tape kept), backward =
    hificar_disc_backward + hificar_disc_weight_norm_backward.  Inputs after (module, x, names): the module's RAW parameters."""

    @staticmethod
    def forward(ctx, module, x, names, *params):
        lib, handle = module._lib, module._handle
        B, _, T = x.shape
        dev = x.device
        x = x.detach().to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            held = _send(module, names, params, stream)
            nbytes = int(lib.hificar_disc_tape_bytes(handle, B, T))
            tape = torch.empty(nbytes // 4 + 64, dtype=torch.float32, device=dev)
            toff = ((-tape.data_ptr()) % 256) // 4
            rc = lib.hificar_disc_forward(handle, x.data_ptr(), B, T, tape.data_ptr() + 4 * toff, nbytes, stream)
        _native.check(rc, "hificar_disc_forward")
        outs = []
        for info in module._output_infos(B, T):
            o = toff + info.offset_bytes // 4
            outs.append(tape[o:o + info.nseq * info.rows * info.pitch].view(info.nseq, info.rows, info.pitch))
        ctx.module, ctx.tape, ctx.toff, ctx.nbytes, ctx.BT = module, tape, toff, nbytes, (B, T)
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.held, ctx.params, ctx.versions = held, params, [p._version for p in params]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        module = ctx.module
        lib, handle = module._lib, module._handle
        B, T = ctx.BT
        dev = ctx.tape.device
        if any(t._version != v for t, v in zip(ctx.params, ctx.versions)):
            raise RuntimeError("a discriminator parameter was modified in place between forward and backward")
        need_x = ctx.needs_input_grad[1]
        need_p = any(ctx.needs_input_grad[3:])
        keep = [None if g is None else g.to(torch.float32).contiguous() for g in douts]
        ptrs = (ctypes.c_void_p * len(keep))(*[None if g is None else g.data_ptr() for g in keep])
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            grads = torch.zeros(int(lib.hificar_disc_grad_floats(handle)), dtype=torch.float32, device=dev) if need_p else None
            dx = torch.empty((B, 1, T), dtype=torch.float32, device=dev) if need_x else None
            wsb = int(lib.hificar_disc_backward_workspace_bytes(handle, B, T))
            ws = torch.empty(wsb // 4 + 64, dtype=torch.float32, device=dev)
            woff = ((-ws.data_ptr()) % 256) // 4
            rc = lib.hificar_disc_backward(handle, ptrs, B, T, ctx.tape.data_ptr() + 4 * ctx.toff, ctx.nbytes,
                                           grads.data_ptr() if need_p else None, dx.data_ptr() if need_x else None,
                                           ws.data_ptr() + 4 * woff, wsb, stream)
            _native.check(rc, "hificar_disc_backward")
            gw = [None] * len(ctx.shapes)
            if need_p:
                raw = torch.zeros(int(lib.hificar_disc_raw_grad_floats(handle)), dtype=torch.float32, device=dev)
                _native.check(lib.hificar_disc_weight_norm_backward(handle, grads.data_ptr(), raw.data_ptr(), stream),
                              "hificar_disc_weight_norm_backward")
                if module._grad_sync is not None:
                    import torch.distributed as dist

                    group, average = module._grad_sync
                    dist.all_reduce(raw, group=group)  # one bucket: every discriminator gradient (RCCL under "nccl")
                    if average:
                        raw.div_(dist.get_world_size(group))
                off = 0
                for i, shape in enumerate(ctx.shapes):
                    n = int(np.prod(shape))
                    gw[i] = raw[off:off + n].view(shape)
                    off += (n + 3) & ~3
        ctx.tape = ctx.held = ctx.params = None
        return (None, dx, None, *gw)


def _aligned(nfloats, dev):
    """(tensor, float offset of its first 256-byte aligned element)."""
    t = torch.empty(nfloats + 64, dtype=torch.float32, device=dev)
    return t, ((-t.data_ptr()) % 256) // 4


class _Pass:
    """One native forward pass kept for the loss / backward kernels."""

    def __init__(self, module, x, stream):
        lib, handle = module._lib, module._handle
        B, _, T = x.shape
        self.nbytes = int(lib.hificar_disc_tape_bytes(handle, B, T))
        self.tape, self.off = _aligned(self.nbytes // 4, x.device)
        self.ptr = self.tape.data_ptr() + 4 * self.off
        x = x.detach().to(torch.float32).contiguous()
        _native.check(lib.hificar_disc_forward(handle, x.data_ptr(), B, T, self.ptr, self.nbytes, stream), "hificar_disc_forward")


def _loss_cfg(loss_type, average_by_discriminators, fm_average_by_layers, fm_average_by_discriminators, fm_include_final_outputs,
              lambda_adv, lambda_feat_match):
    assert loss_type in ("mse", "hinge"), f"{loss_type} is not supported."
    c = _native.HificarGanLossConfig()
    c.loss_type = 0 if loss_type == "mse" else 1
    c.average_by_discriminators = int(bool(average_by_discriminators))
    c.fm_average_by_layers = int(bool(fm_average_by_layers))
    c.fm_average_by_discriminators = int(bool(fm_average_by_discriminators))
    c.fm_include_final_outputs = int(bool(fm_include_final_outputs))
    c.lambda_adv, c.lambda_feat_match = float(lambda_adv), float(lambda_feat_match)
    return c


class _GeneratorLossFunction(torch.autograd.Function):
    """Generator side of the GAN criterion in one node (train.py:341-362): D(fake), D(real) (no graph), adversarial + feature-matching
    losses and their gradients on the engine's buffers (hificar_disc_loss), backward = the discriminators' data gradient down to the
    waveform.  Returns (lambda_adv * (adv + lambda_feat_match * fm), adv, fm); only the first is differentiable (with respect to y_fake;
    the discriminator's parameters get no gradient here — the reference computes and then discards them, train.py:364-372,429)."""

    @staticmethod
    def forward(ctx, module, y_fake, y_real, cfg, names, *params):
        lib, handle = module._lib, module._handle
        B, _, T = y_fake.shape
        dev = y_fake.device
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _send(module, names, params, stream)
            fake = _Pass(module, y_fake, stream)
            real = module._real_pass(y_real, params, stream) if y_real is not None else None
            douts, doff = _aligned(int(lib.hificar_disc_dout_floats(handle, B, T)), dev)
            values = torch.empty(3, dtype=torch.float32, device=dev)
            rc = lib.hificar_disc_loss(handle, ctypes.byref(cfg), 0, fake.ptr, real.ptr if real is not None else None, B, T, values.data_ptr(),
                                       douts.data_ptr() + 4 * doff, stream)
        _native.check(rc, "hificar_disc_loss")
        ctx.module, ctx.fake, ctx.douts, ctx.doff, ctx.BT, ctx.cfg, ctx.with_fm = module, fake, douts, doff, (B, T), cfg, real is not None
        ctx.real_pass = real  # (identity only: which cached D(y_real) pass this loss was computed against)
        total, adv, fm = values[2], values[0], values[1]
        ctx.mark_non_differentiable(adv, fm)
        return total, adv, fm

    @staticmethod
    def backward(ctx, g_total, _g_adv, _g_fm):
        module = ctx.module
        lib, handle = module._lib, module._handle
        B, T = ctx.BT
        dev = ctx.douts.device
        if not ctx.needs_input_grad[1]:
            return (None,) * (5 + len(ctx.needs_input_grad) - 5)
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            dx = torch.empty((B, 1, T), dtype=torch.float32, device=dev)
            wsb = int(lib.hificar_disc_backward_workspace_bytes(handle, B, T))
            ws, woff = _aligned(wsb // 4, dev)
            rc = lib.hificar_disc_backward_flat(handle, ctx.douts.data_ptr() + 4 * ctx.doff, 0, int(ctx.with_fm), ctx.cfg.fm_include_final_outputs, B, T,
                                                ctx.fake.ptr, ctx.fake.nbytes, None, dx.data_ptr(), ws.data_ptr() + 4 * woff, wsb, stream)
        _native.check(rc, "hificar_disc_backward_flat")
        ctx.fake = ctx.douts = None
        done = torch.cuda.Event()
        done.record()
        # start_real_gradient's side stream may start behind this point instead of behind everything enqueued since — but only for the real
        # pass THIS loss used: the event is stored with that pass's identity
        module.__dict__["_gside_done"] = (ctx.real_pass, done)
        ctx.real_pass = None
        return (None, dx * g_total, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 5)


class _DiscriminatorLossFunction(torch.autograd.Function):
    """Discriminator side (train.py:421-424): D(real), D(fake), real + fake adversarial losses, backward = both passes' parameter
    gradients (weight norm included).  Returns (real_loss + fake_loss, real_loss, fake_loss); the first is differentiable with
    respect to the discriminator's parameters."""

    @staticmethod
    def forward(ctx, module, y_fake, y_real, cfg, names, *params):
        lib, handle = module._lib, module._handle
        B, _, T = y_fake.shape
        dev = y_fake.device
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            held = _send(module, names, params, stream)
            passes, douts, vals = [], [], []
            for mode, y in ((1, y_fake), (2, y_real)):
                ps = _Pass(module, y, stream) if mode == 1 else module._real_pass(y, params, stream)
                d, doff = _aligned(int(lib.hificar_disc_dout_floats(handle, B, T)), dev)
                v = torch.empty(3, dtype=torch.float32, device=dev)
                _native.check(lib.hificar_disc_loss(handle, ctypes.byref(cfg), mode, ps.ptr, None, B, T, v.data_ptr(), d.data_ptr() + 4 * doff, stream),
                              "hificar_disc_loss")
                passes.append(ps)
                douts.append((d, doff))
                vals.append(v)
        ctx.module, ctx.passes, ctx.douts, ctx.BT, ctx.cfg = module, passes, douts, (B, T), cfg
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.held, ctx.params, ctx.versions = held, params, [p._version for p in params]
        fake_loss, real_loss = vals[0][0], vals[1][0]
        total = real_loss + fake_loss
        ctx.mark_non_differentiable(real_loss, fake_loss)
        return total, real_loss, fake_loss

    @staticmethod
    def backward(ctx, g_total, _g_real, _g_fake):
        module = ctx.module
        lib, handle = module._lib, module._handle
        B, T = ctx.BT
        dev = ctx.douts[0][0].device
        if any(t._version != v for t, v in zip(ctx.params, ctx.versions)):
            raise RuntimeError("a discriminator parameter was modified in place between forward and backward")
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            wsb = int(lib.hificar_disc_backward_workspace_bytes(handle, B, T))
            ws, woff = _aligned(wsb // 4, dev)
            nfold = int(lib.hificar_disc_grad_floats(handle))
            raw = torch.zeros(int(lib.hificar_disc_raw_grad_floats(handle)), dtype=torch.float32, device=dev)
            reducer, cb = None, None
            # ONE folded-gradient buffer: the second pass adds to what the first left (hificar_disc_set_grad_accumulate), and the weight-norm
            # chain rule multiplies by the upstream gradient while it writes (hificar_disc_set_grad_scale): no add pass, no scaling pass
            g_total = g_total.to(torch.float32).contiguous()
            plan = list(zip((1, 2), ctx.passes, ctx.douts))
            early = module.__dict__.pop("_early_real", None)
            if early is not None and early["pass"] is ctx.passes[1] and early["cfg"] == (ctx.cfg.loss_type, ctx.cfg.average_by_discriminators):
                # the real pass's parameter gradients were computed next to the generator's backward (start_real_gradient): only the fake
                # pass is left, and it adds to them
                g = early["g"]
                torch.cuda.current_stream().wait_event(early["done"])
                g.record_stream(torch.cuda.current_stream())
                plan, first_accumulates = plan[:1], True
            else:
                g = torch.zeros(nfold, dtype=torch.float32, device=dev)
                first_accumulates = False
            early = None
            for step, (mode, ps, (d, doff)) in enumerate(plan):
                if step == len(plan) - 1 and module._grad_sync is not None:
                    # data-parallel training: one gradient bucket per sub-discriminator.  During the LAST pass libhificar calls back as
                    # each sub-network's gradients (both passes' sum by then) are enqueued on its side stream: there the weight-norm chain
                    # rule runs and the bucket's all-reduce (RCCL over xGMI) starts while the other sub-networks still compute.
                    from ..utils.buckets import BucketHook, BucketReducer, bucket_ranges

                    group, average = module._grad_sync
                    nb = int(lib.hificar_disc_grad_bucket_count(handle))
                    ids = [int(lib.hificar_disc_raw_param_bucket(handle, i)) for i in range(len(ctx.shapes))]
                    ranges, total = bucket_ranges(ids, [int(np.prod(sh)) for sh in ctx.shapes], nb)
                    assert total == raw.numel()

                    def chain_rule(bucket, bstream):
                        _native.check(lib.hificar_disc_weight_norm_backward_bucket(handle, g.data_ptr(), raw.data_ptr(), bucket, ctypes.c_void_p(bstream)),
                                      "hificar_disc_weight_norm_backward_bucket")

                    # (the sub-network's side stream becomes torch's current stream inside the callback: the collective orders itself behind it)
                    reducer = BucketHook(BucketReducer(raw, ranges, group, average), chain_rule,
                                         lambda bstream: torch.cuda.stream(torch.cuda.ExternalStream(bstream, device=dev)))
                    cb = _native.BUCKET_FN(reducer)
                    _native.check(lib.hificar_disc_set_bucket_callback(handle, cb, None), "hificar_disc_set_bucket_callback")
                try:
                    _native.check(lib.hificar_disc_set_grad_accumulate(handle, 1 if (step > 0 or first_accumulates) else 0), "hificar_disc_set_grad_accumulate")
                    if cb is not None:  # (the bucket callback runs the weight-norm chain rule during this pass)
                        _native.check(lib.hificar_disc_set_grad_scale(handle, g_total.data_ptr()), "hificar_disc_set_grad_scale")
                    rc = lib.hificar_disc_backward_flat(handle, d.data_ptr() + 4 * doff, mode, 0, 0, B, T, ps.ptr, ps.nbytes, g.data_ptr(), None,
                                                        ws.data_ptr() + 4 * woff, wsb, stream)
                finally:
                    lib.hificar_disc_set_grad_accumulate(handle, 0)
                    if cb is not None:
                        lib.hificar_disc_set_bucket_callback(handle, _native.BUCKET_FN(), None)
                        lib.hificar_disc_set_grad_scale(handle, None)
                _native.check(rc, "hificar_disc_backward_flat")
            if reducer is not None:
                reducer.finish()  # (re-raises what a bucket callback caught)
            else:
                try:
                    _native.check(lib.hificar_disc_set_grad_scale(handle, g_total.data_ptr()), "hificar_disc_set_grad_scale")
                    _native.check(lib.hificar_disc_weight_norm_backward(handle, g.data_ptr(), raw.data_ptr(), stream), "hificar_disc_weight_norm_backward")
                finally:
                    lib.hificar_disc_set_grad_scale(handle, None)
        gw, off = [], 0
        for shape in ctx.shapes:
            n = int(np.prod(shape))
            gw.append(raw[off:off + n].view(shape))
            off += (n + 3) & ~3
        ctx.passes = ctx.douts = ctx.held = ctx.params = None
        return (None, None, None, None, None, *gw)


class _SubDisc(torch.nn.Module):
    pass


def _apply_spectral_norm(conv, eps=1e-12):
    """torch.nn.utils.spectral_norm on a parameter holder (hifigan.py:440-448): ``weight`` becomes the parameter ``weight_orig`` plus the
    power-iteration buffers ``weight_u`` (cout) / ``weight_v`` (cin * k), registered in torch's order (bias, weight_orig | weight_u, weight_v)."""
    w = conv._parameters.pop("weight").detach()
    conv.weight_orig = torch.nn.Parameter(w)
    mat = w.reshape(w.shape[0], -1)
    u = torch.nn.functional.normalize(torch.randn(mat.shape[0]), dim=0, eps=eps)
    v = torch.nn.functional.normalize(torch.randn(mat.shape[1]), dim=0, eps=eps)
    conv.register_buffer("weight_u", u)
    conv.register_buffer("weight_v", v)
    conv.sn_eps = eps


def _spectral_weight(conv, training):
    """The weight a spectrally normalised conv uses in THIS forward (torch.nn.utils.spectral_norm.SpectralNorm.compute_weight, one power
    iteration): in training mode u, v are advanced in place first (under no_grad), then sigma = u . (W v) and weight = weight_orig / sigma,
    differentiable with respect to weight_orig with u, v held constant.  A handful of tiny torch ops per layer; the result is handed to the
    engine as a plain weight."""
    w = conv.weight_orig
    mat = w.reshape(w.shape[0], -1)
    u, v = conv.weight_u, conv.weight_v
    if training:
        with torch.no_grad():
            v.copy_(torch.nn.functional.normalize(torch.mv(mat.t(), u), dim=0, eps=conv.sn_eps))
            u.copy_(torch.nn.functional.normalize(torch.mv(mat, v), dim=0, eps=conv.sn_eps))
        u, v = u.clone(), v.clone()
    sigma = torch.dot(u, torch.mv(mat, v))
    return w / sigma


def _slope(sub_params):
    """LeakyReLU slope of a scale / period discriminator's parameter dict.  A dict WITHOUT the key gets the sub-discriminator class's own
    default {"negative_slope": 0.1} (hifigan.py:331,517: the user's sub-dict replaces the multi-discriminator's default dict as a whole);
    an explicit dict without "negative_slope" (e.g. {}) gets torch.nn.LeakyReLU's 0.01."""
    return float(sub_params.get("nonlinear_activation_params", {"negative_slope": 0.1}).get("negative_slope", 0.01))


class HiFiGANMultiScaleMultiPeriodDiscriminator(torch.nn.Module):
    """HiFi-GAN multi-scale + multi-period discriminator (MI355X-native forward and backward).  Constructor arguments as the
    reference's (hifigan.py:744-781)."""

    def __init__(self, **kw):
        super().__init__()
        layout = kw.pop("_layout", "msmpd")  # "msd" / "mpd": the stand-alone multi-scale / multi-period classes below
        p = self._params = disc_params(**kw)
        if p["scale_downsample_pooling"] != "AvgPool1d":
            raise NotImplementedError("scale_downsample_pooling: only AvgPool1d is built")
        sp, pp = p["scale_discriminator_params"], p["period_discriminator_params"]
        for q in (sp, pp):
            if q.get("nonlinear_activation", "LeakyReLU") != "LeakyReLU":
                raise NotImplementedError("discriminator activations: only LeakyReLU is built")
            if q.get("in_channels", 1) != 1 or q.get("out_channels", 1) != 1:
                raise NotImplementedError("discriminators with in_channels / out_channels other than 1 are not built")
        self._spectral = bool(pp.get("use_spectral_norm", False))
        if self._spectral and pp.get("use_weight_norm", True):
            raise ValueError("Either use use_weight_norm or use_spectral_norm.")  # hifigan.py:390-391
        self._s_layers = scale_disc_layers(**sp)
        self._p_layers = period_disc_layers(**pp)
        if max(len(self._s_layers), len(self._p_layers)) > _native.DISC_MAX_LAYERS:
            raise ValueError("too many discriminator layers")
        # ---- parameters under the reference's names
        self.msd = _SubDisc()
        self.msd.discriminators = torch.nn.ModuleList()
        for _ in range(p["scales"]):
            d = _SubDisc()
            d.layers = torch.nn.ModuleList()
            for L in self._s_layers:
                conv = _ConvParams((L["cout"], L["cin"] // L["groups"], L["k"]), L["cout"], bias=L["bias"])
                conv._parameters["bias"] = conv._parameters.pop("bias")  # a plain Conv1d lists weight before bias
                # (follow_official_norm asks for spectral / weight norm, but the reference's ScaleDiscriminator.apply_* test for Conv2d
                # on this Conv1d stack: no norm is ever applied, hifigan.py:645-663 — plain weights, as its checkpoints hold)
                slope = _slope(sp)
                d.layers.append(torch.nn.Sequential(conv, torch.nn.LeakyReLU(slope)) if L["act"] else conv)
            self.msd.discriminators.append(d)
        self.mpd = _SubDisc()
        self.mpd.discriminators = torch.nn.ModuleList()
        for _period in p["periods"]:
            d = _SubDisc()
            d.convs = torch.nn.ModuleList()
            slope = _slope(pp)
            for L in self._p_layers:
                conv = _ConvParams((L["cout"], L["cin"], L["k"], 1), L["cout"], bias=True)
                if self._spectral:
                    _apply_spectral_norm(conv)
                elif pp.get("use_weight_norm", True):
                    conv.apply_weight_norm()
                else:
                    conv._parameters["bias"] = conv._parameters.pop("bias")
                if L["act"]:
                    d.convs.append(torch.nn.Sequential(conv, torch.nn.LeakyReLU(slope)))
                else:
                    d.output_conv = conv
            self.mpd.discriminators.append(d)
        self._name_prefix = ""
        self._single = layout in ("scale", "period")
        if self._single:  # ONE scale / period discriminator: its layers sit at the top ("layers.0.0.weight", "convs.0.0.bias", "output_conv.*")
            sub = (self.msd if layout == "scale" else self.mpd).discriminators[0]
            self._name_prefix = ("msd" if layout == "scale" else "mpd") + ".discriminators.0."
            del self.msd, self.mpd
            for name, child in list(sub.named_children()):
                setattr(self, name, child)
        elif layout != "msmpd":  # stand-alone multi class: its sub-discriminators sit at the top ("discriminators.0...."), hifigan.py:451-500,666-738
            self._name_prefix = layout + "."
            subs = getattr(self, layout).discriminators
            del self.msd, self.mpd
            self.discriminators = subs
        self._lib = self._handle = None
        self._grad_sync = None
        self._info_cache = {}
        from ..utils.optim_hook import watch

        watch(self)  # fused optimizers do not bump Parameter._version: every optimizer.step() over these parameters calls invalidate_parameters()

    # ------------------------------------------------------------------ native handle
    def _config(self):
        p = self._params
        c = _native.HificarDiscConfig()
        c.n_scales = p["scales"]
        pool = p["scale_downsample_pooling_params"]
        c.pool_kernel, c.pool_stride, c.pool_pad = pool["kernel_size"], pool.get("stride", pool["kernel_size"]), pool.get("padding", 0)
        c.s_n_layers = len(self._s_layers)
        for l, L in enumerate(self._s_layers):
            c.s_cin[l], c.s_cout[l], c.s_k[l], c.s_stride[l], c.s_pad[l], c.s_groups[l] = L["cin"], L["cout"], L["k"], L["stride"], L["pad"], L["groups"]
        c.s_bias = int(bool(self._s_layers[0]["bias"]))
        c.s_slope = _slope(p["scale_discriminator_params"])
        c.n_periods = len(p["periods"])
        for i, per in enumerate(p["periods"]):
            c.periods[i] = per
        c.p_n_layers = len(self._p_layers)
        for l, L in enumerate(self._p_layers):
            c.p_cin[l], c.p_cout[l], c.p_k[l], c.p_stride[l], c.p_pad[l] = L["cin"], L["cout"], L["k"], L["stride"], L["pad"]
        c.p_slope = _slope(p["period_discriminator_params"])
        return c

    # copy.deepcopy / pickle build a module without running __init__ (see _NativeGenerator): the engine handle, the cached tapes / parameter
    # lists / side stream and the gradient-sync wiring stay behind; the copy registers with the optimizer post-step hook itself
    # per-process state a copy must not carry: native handles, ctypes arrays (``_raw_cnames``: c_char_p arrays cannot be pickled), the held copy of the
    # parameters last sent to the device (``_sent_held``: a second 70 M floats), cached passes / streams / events
    _EPHEMERAL = ("_handle", "_lib", "_grad_sync", "_info_cache", "_sent_sig", "_sent_held", "_raw_cnames", "_real_cache", "_early_real", "_gside_done",
                  "_raw_cache", "_early_stream")

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in self._EPHEMERAL:
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        d = self.__dict__
        d["_handle"] = d["_lib"] = d["_grad_sync"] = None
        d["_info_cache"] = {}
        from ..utils.optim_hook import watch

        watch(self)

    def _native_handle(self):
        if self._handle is None:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("HiFiGANMultiScaleMultiPeriodDiscriminator: parameters are on %s; the discriminators only exist as HIP "
                                   "kernels (move the model to a MI355X with .to('cuda')). There is no CPU fallback." % dev)
            self._lib = _native.load_library()
            cfg = self._config()
            handle = ctypes.c_void_p()
            with torch.cuda.device(dev):
                _native.check(self._lib.hificar_disc_create(ctypes.byref(cfg), ctypes.byref(handle)), "hificar_disc_create")
            self._handle = handle
        return self._handle

    def __del__(self):
        handle, lib = self.__dict__.get("_handle"), self.__dict__.get("_lib")
        if handle is not None and lib is not None:
            self.__dict__["_handle"] = None  # (not setattr: torch's Module.__setattr__ may be half torn down at interpreter exit)
            try:
                lib.hificar_disc_destroy(handle)
            except Exception:
                pass

    def _output_infos(self, B, T):
        key = (B, T)
        if key not in self._info_cache:
            n = self._lib.hificar_disc_output_count(self._handle)
            infos = []
            for i in range(n):
                o = _native.HificarDiscOutput()
                _native.check(self._lib.hificar_disc_output_info(self._handle, B, T, i, ctypes.byref(o)), "hificar_disc_output_info")
                infos.append(o)
            if len(self._info_cache) > 64:
                self._info_cache.clear()
            self._info_cache[key] = infos
        return self._info_cache[key]

    def invalidate_parameters(self):
        """Force the next call to hand the parameters over again.  In-place updates through autograd-visible ops (foreach optimizer.step(),
        load_state_dict, p.copy_ under no_grad) are noticed by themselves through the tensors' version counters; writes through
        ``p.data`` are not.  torch's FUSED optimizers (``Adam(fused=True)`` does not bump ``_version``) are covered by the process-wide
        post-step hook the constructor registers (articulatory_amd/utils/optim_hook.py)."""
        self.__dict__.pop("_sent_sig", None)
        self.__dict__["_real_cache"] = None
        self.__dict__.pop("_early_real", None)
        self.__dict__.pop("_gside_done", None)

    def _raw_parameters(self):
        cached = self.__dict__.get("_raw_cache")  # (walking the module tree costs ~1 ms per call; the criterion needs it every pass)
        if cached is None:
            names, tensors = [], []
            for name, p in self.named_parameters():
                names.append(self._name_prefix + name)  # the engine's names are the combined discriminator's
                tensors.append(p)
            cached = self.__dict__["_raw_cache"] = (tuple(names), tensors)
        return cached

    def _effective_parameters(self):
        """(names, tensors) handed to the engine.  Without spectral norm: the raw parameters.  With it (period discriminators,
        use_spectral_norm): every normalised conv contributes "<conv>.weight" = weight_orig / sigma, recomputed — and, in training mode, its
        power iteration advanced — at EVERY call, as torch's forward pre-hook does for every ``D(x)`` of the reference."""
        if not self._spectral:
            return self._raw_parameters()
        names, tensors = [], []
        for mname, m in self.named_modules():
            if not isinstance(m, _ConvParams):
                continue
            base = self._name_prefix + mname
            if "weight_orig" in m._parameters:
                names += [base + ".bias", base + ".weight"]
                tensors += [m.bias, _spectral_weight(m, self.training)]
            else:
                for k in m._parameters:
                    if m._parameters[k] is not None:
                        names.append(base + "." + k)
                        tensors.append(m._parameters[k])
        return tuple(names), tensors

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self.__dict__["_raw_cache"] = None
        self.invalidate_parameters()
        return out

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.__dict__["_raw_cache"] = None
        return out

    def sync_gradients(self, group=None, average=True, enabled=True):
        """Data-parallel training: all-reduce the discriminator's gradients inside backward (one flat bucket), see
        HiFiGANGenerator.sync_gradients."""
        self._grad_sync = (group, bool(average)) if enabled else None
        return self

    def macs(self, B, T):
        """Algorithmic multiply-accumulates of one forward over (B, 1, T) (hificar_disc_macs)."""
        self._native_handle()
        return float(self._lib.hificar_disc_macs(self._handle, B, T))

    def profile_begin(self):
        self._native_handle()
        _native.check(self._lib.hificar_profile_begin(self._lib.hificar_disc_engine(self._handle)), "hificar_profile_begin")

    def profile_end(self):
        stats = (_native.HificarKernelStat * 96)()
        n = ctypes.c_int(0)
        _native.check(self._lib.hificar_profile_end(self._lib.hificar_disc_engine(self._handle), stats, 96, ctypes.byref(n)), "hificar_profile_end")
        return [dict(name=stats[i].name.decode(), launches=int(stats[i].launches), total_ms=float(stats[i].total_ms),
                     flops=float(stats[i].flops), bytes=float(stats[i].bytes)) for i in range(min(n.value, 96))]

    # ------------------------------------------------------------------ fused GAN criterion
    def generator_loss(self, y_fake, y_real=None, loss_type="mse", average_by_discriminators=True, lambda_adv=1.0, lambda_feat_match=0.0,
                       fm_average_by_layers=True, fm_average_by_discriminators=True, fm_include_final_outputs=False):
        """The generator step's adversarial part in one autograd node (train.py:341-362): GeneratorAdversarialLoss(D(y_fake)) and, when
        y_real is given, FeatureMatchLoss(D(y_fake), D(y_real)) -> (lambda_adv * (adv + lambda_feat_match * fm), adv, fm).  Defaults as
        the reference's loss modules (adversarial_loss.py:15-19, feat_match_loss.py:15-20)."""
        self._check(y_fake)
        self._no_fused_spectral()
        self._native_handle()
        cfg = _loss_cfg(loss_type, average_by_discriminators, fm_average_by_layers, fm_average_by_discriminators, fm_include_final_outputs,
                        lambda_adv, lambda_feat_match)
        names, tensors = self._raw_parameters()
        return _GeneratorLossFunction.apply(self, y_fake, y_real, cfg, names, *tensors)

    def discriminator_loss(self, y_fake, y_real, loss_type="mse", average_by_discriminators=True):
        """DiscriminatorAdversarialLoss(D(y_fake), D(y_real)) in one autograd node (train.py:421-424) -> (real + fake, real, fake)."""
        self._check(y_fake)
        self._check(y_real)
        self._no_fused_spectral()
        self._native_handle()
        cfg = _loss_cfg(loss_type, average_by_discriminators, True, True, False, 1.0, 0.0)
        names, tensors = self._raw_parameters()
        return _DiscriminatorLossFunction.apply(self, y_fake.detach(), y_real.detach(), cfg, names, *tensors)

    def start_real_gradient(self, loss_type="mse", average_by_discriminators=True):
        """The real half of the NEXT discriminator_loss(y_fake, y_real) backward, started now on a side stream.

        In a GAN iteration (train.py:341-437) D(y_real) is computed in the generator part (feature matching) and its parameter gradients
        depend neither on the generator update nor on the re-computed fake batch that the discriminator part waits for: called right after
        the generator loss's backward, the real pass's backward (one of the two backward passes of the discriminator update) runs next to
        the generator's own backward / optimizer step / re-computed forward — another engine, single-stream — instead of behind them.
        discriminator_loss's backward then finds the gradients, runs the fake pass only and adds (same sum: bit-identical gradients).
        Returns False (and does nothing) when there is no cached D(y_real) pass to start from."""
        cached = self.__dict__.get("_real_cache")
        if cached is None or self._spectral or self._handle is None:
            return False
        _, ps, y_real = cached
        lib, handle = self._lib, self._handle
        B, _, T = y_real.shape
        dev = y_real.device
        cfg = _loss_cfg(loss_type, average_by_discriminators, True, True, False, 1.0, 0.0)
        with torch.cuda.device(dev):
            side = self.__dict__.get("_early_stream")
            if side is None:
                side = self.__dict__["_early_stream"] = torch.cuda.Stream(device=dev)
            after = self.__dict__.pop("_gside_done", None)
            if after is not None and after[0] is ps:  # recorded by the backward of the generator loss that used THIS cached real pass
                side.wait_event(after[1])
            else:  # no such backward ran (early return, retain_graph, a newer cached pass): order behind everything enqueued so far
                side.wait_stream(torch.cuda.current_stream())
            ps.tape.record_stream(side)
            with torch.cuda.stream(side):
                stream = ctypes.c_void_p(side.cuda_stream)
                d, doff = _aligned(int(lib.hificar_disc_dout_floats(handle, B, T)), dev)
                v = torch.empty(3, dtype=torch.float32, device=dev)
                _native.check(lib.hificar_disc_loss(handle, ctypes.byref(cfg), 2, ps.ptr, None, B, T, v.data_ptr(), d.data_ptr() + 4 * doff, stream),
                              "hificar_disc_loss")
                g = torch.zeros(int(lib.hificar_disc_grad_floats(handle)), dtype=torch.float32, device=dev)
                wsb = int(lib.hificar_disc_backward_workspace_bytes(handle, B, T))
                ws, woff = _aligned(wsb // 4, dev)
                _native.check(lib.hificar_disc_set_grad_accumulate(handle, 0), "hificar_disc_set_grad_accumulate")
                _native.check(lib.hificar_disc_backward_flat(handle, d.data_ptr() + 4 * doff, 2, 0, 0, B, T, ps.ptr, ps.nbytes, g.data_ptr(), None,
                                                             ws.data_ptr() + 4 * woff, wsb, stream), "hificar_disc_backward_flat")
                done = torch.cuda.Event()
                done.record(side)
        self.__dict__["_early_real"] = {"pass": ps, "cfg": (cfg.loss_type, cfg.average_by_discriminators), "g": g, "done": done}
        return True

    def _no_fused_spectral(self):
        if self._spectral:
            raise NotImplementedError("spectral norm advances its power iteration at every D(x) of the reference (two different weight sets inside "
                                      "one criterion): use forward(x, native=True) with articulatory_amd.losses (the Trainer does)")

    def _real_pass(self, y_real, params, stream):
        """D(y_real) of the generator step and of the discriminator step that follows it are the same computation (same batch, the
        discriminator is only updated afterwards — train.py:347,421): the forward tape is kept and handed out again while neither the
        tensor nor any parameter has changed."""
        key = (y_real.data_ptr(), y_real._version, tuple(y_real.shape), tuple(p._version for p in params), tuple(p.data_ptr() for p in params[:4]))
        cached = self.__dict__.get("_real_cache")
        if cached is not None and cached[0] == key:  # (cached[2] keeps the storage alive: same address + version = same data)
            self.__dict__["_real_cache"] = None  # one re-use: the discriminator update that follows invalidates it anyway
            return cached[1]
        ps = _Pass(self, y_real, stream)
        self.__dict__["_real_cache"] = (key, ps, y_real)
        return ps

    def _check(self, x):
        if not x.is_cuda:
            raise RuntimeError("HiFiGANMultiScaleMultiPeriodDiscriminator needs CUDA/HIP tensors; there is no CPU fallback")
        if x.dim() != 3 or x.shape[1] != 1:
            raise RuntimeError(f"Expected input of shape (B, 1, T), got {tuple(x.shape)}")

    # ------------------------------------------------------------------ forward
    def forward(self, x, native=False):
        """x: (B, 1, T) -> list (msd scales, then mpd periods) of lists of layer outputs (hifigan.py:806-825)."""
        if not x.is_cuda:
            raise RuntimeError("HiFiGANMultiScaleMultiPeriodDiscriminator.forward needs a CUDA/HIP tensor; there is no CPU fallback")
        if x.dim() != 3 or x.shape[1] != 1:
            raise RuntimeError(f"Expected input of shape (B, 1, T), got {tuple(x.shape)}")
        self._native_handle()
        B, _, T = x.shape
        names, tensors = self._effective_parameters()
        bufs = _DiscFunction.apply(self, x, names, *tensors)
        infos = self._output_infos(B, T)
        outs, i = [], 0
        while i < len(infos):
            sub = infos[i].sub
            layers = []
            while i < len(infos) and infos[i].sub == sub:
                n = infos[i].n_groups
                layers.append(DiscOutput(list(bufs[i:i + n]), infos[i].channels, infos[i].period, B))
                i += n
            outs.append(layers)
        if native:
            return outs[0] if self._single else outs
        ref = []
        for layers in outs:
            r = [o.reference_layout() for o in layers]
            if layers[-1].period:
                r[-1] = torch.flatten(r[-1], 1, -1)  # (B, 1, H, P) -> (B, H * P)   (hifigan.py:414)
            ref.append(r)
        return ref[0] if self._single else ref


class HiFiGANMultiScaleDiscriminator(HiFiGANMultiScaleMultiPeriodDiscriminator):
    """HiFi-GAN multi-scale discriminator alone (hifigan.py:666-738), same engine."""

    def __init__(self, scales=3, downsample_pooling="AvgPool1d", downsample_pooling_params={"kernel_size": 4, "stride": 2, "padding": 2},
                 discriminator_params=None, follow_official_norm=False):
        kw = {} if discriminator_params is None else {"scale_discriminator_params": discriminator_params}
        super().__init__(scales=scales, scale_downsample_pooling=downsample_pooling, scale_downsample_pooling_params=downsample_pooling_params,
                         follow_official_norm=follow_official_norm, periods=[], _layout="msd", **kw)
        if scales < 1:
            raise ValueError("scales must be at least 1")


class HiFiGANMultiPeriodDiscriminator(HiFiGANMultiScaleMultiPeriodDiscriminator):
    """HiFi-GAN multi-period discriminator alone (hifigan.py:451-500), same engine."""

    def __init__(self, periods=[2, 3, 5, 7, 11], discriminator_params=None):
        kw = {} if discriminator_params is None else {"period_discriminator_params": discriminator_params}
        super().__init__(scales=0, periods=list(periods), _layout="mpd", **kw)
        if not periods:
            raise ValueError("periods must not be empty")


class HiFiGANScaleDiscriminator(HiFiGANMultiScaleMultiPeriodDiscriminator):
    """ONE HiFi-GAN scale discriminator (hifigan.py:503-663): forward returns the list of its layer outputs."""

    def __init__(self, **discriminator_params):
        super().__init__(scales=1, periods=[], scale_discriminator_params=dict(disc_params()["scale_discriminator_params"], **discriminator_params),
                         _layout="scale")


class HiFiGANPeriodDiscriminator(HiFiGANMultiScaleMultiPeriodDiscriminator):
    """ONE HiFi-GAN period discriminator (hifigan.py:317-448)."""

    def __init__(self, period=3, **discriminator_params):
        super().__init__(scales=0, periods=[period], period_discriminator_params=dict(disc_params()["period_discriminator_params"], **discriminator_params),
                         _layout="period")
