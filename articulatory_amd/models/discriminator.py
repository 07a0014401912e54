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
    held = [t.detach() if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().to(torch.float32).contiguous() for t in tensors]
    cn = getattr(module, "_raw_cnames", None)
    if cn is None or cn[0] != names:
        cn = module._raw_cnames = (names, (ctypes.c_char_p * len(names))(*[n.encode() for n in names]))
    ptrs = (ctypes.c_void_p * len(held))(*[t.data_ptr() for t in held])
    _native.check(module._lib.hificar_disc_set_parameters_device(module._handle, cn[1], ptrs, len(held), stream),
                  "hificar_disc_set_parameters_device")
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


class _SubDisc(torch.nn.Module):
    pass


class HiFiGANMultiScaleMultiPeriodDiscriminator(torch.nn.Module):
    """HiFi-GAN multi-scale + multi-period discriminator (MI355X-native forward and backward).  Constructor arguments as the
    reference's (hifigan.py:744-781)."""

    def __init__(self, **kw):
        super().__init__()
        p = self._params = disc_params(**kw)
        if p["scale_downsample_pooling"] != "AvgPool1d":
            raise NotImplementedError("scale_downsample_pooling: only AvgPool1d is built")
        sp, pp = p["scale_discriminator_params"], p["period_discriminator_params"]
        for q in (sp, pp):
            if q.get("nonlinear_activation", "LeakyReLU") != "LeakyReLU":
                raise NotImplementedError("discriminator activations: only LeakyReLU is built")
            if q.get("in_channels", 1) != 1 or q.get("out_channels", 1) != 1:
                raise NotImplementedError("discriminators with in_channels / out_channels other than 1 are not built")
        if pp.get("use_spectral_norm", False):
            raise NotImplementedError("spectral norm on the period discriminators is not built")
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
                slope = sp.get("nonlinear_activation_params", {}).get("negative_slope", 0.01)
                d.layers.append(torch.nn.Sequential(conv, torch.nn.LeakyReLU(slope)) if L["act"] else conv)
            self.msd.discriminators.append(d)
        self.mpd = _SubDisc()
        self.mpd.discriminators = torch.nn.ModuleList()
        for _period in p["periods"]:
            d = _SubDisc()
            d.convs = torch.nn.ModuleList()
            slope = pp.get("nonlinear_activation_params", {}).get("negative_slope", 0.01)
            for L in self._p_layers:
                conv = _ConvParams((L["cout"], L["cin"], L["k"], 1), L["cout"], bias=True)
                if pp.get("use_weight_norm", True):
                    conv.apply_weight_norm()
                else:
                    conv._parameters["bias"] = conv._parameters.pop("bias")
                if L["act"]:
                    d.convs.append(torch.nn.Sequential(conv, torch.nn.LeakyReLU(slope)))
                else:
                    d.output_conv = conv
            self.mpd.discriminators.append(d)
        self._lib = self._handle = None
        self._grad_sync = None
        self._info_cache = {}

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
        c.s_slope = p["scale_discriminator_params"].get("nonlinear_activation_params", {}).get("negative_slope", 0.01)
        c.n_periods = len(p["periods"])
        for i, per in enumerate(p["periods"]):
            c.periods[i] = per
        c.p_n_layers = len(self._p_layers)
        for l, L in enumerate(self._p_layers):
            c.p_cin[l], c.p_cout[l], c.p_k[l], c.p_stride[l], c.p_pad[l] = L["cin"], L["cout"], L["k"], L["stride"], L["pad"]
        c.p_slope = p["period_discriminator_params"].get("nonlinear_activation_params", {}).get("negative_slope", 0.01)
        return c

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

    def _raw_parameters(self):
        names, tensors = [], []
        for name, p in self.named_parameters():
            names.append(name)
            tensors.append(p)
        return tuple(names), tensors

    def sync_gradients(self, group=None, average=True, enabled=True):
        """Data-parallel training: all-reduce the discriminator's gradients inside backward (one flat bucket), see
        HiFiGANGenerator.sync_gradients."""
        self._grad_sync = (group, bool(average)) if enabled else None
        return self

    def profile_begin(self):
        self._native_handle()
        _native.check(self._lib.hificar_profile_begin(self._lib.hificar_disc_engine(self._handle)), "hificar_profile_begin")

    def profile_end(self):
        stats = (_native.HificarKernelStat * 96)()
        n = ctypes.c_int(0)
        _native.check(self._lib.hificar_profile_end(self._lib.hificar_disc_engine(self._handle), stats, 96, ctypes.byref(n)), "hificar_profile_end")
        return [dict(name=stats[i].name.decode(), launches=int(stats[i].launches), total_ms=float(stats[i].total_ms),
                     flops=float(stats[i].flops), bytes=float(stats[i].bytes)) for i in range(min(n.value, 96))]

    # ------------------------------------------------------------------ forward
    def forward(self, x, native=False):
        """x: (B, 1, T) -> list (msd scales, then mpd periods) of lists of layer outputs (hifigan.py:806-825)."""
        if not x.is_cuda:
            raise RuntimeError("HiFiGANMultiScaleMultiPeriodDiscriminator.forward needs a CUDA/HIP tensor; there is no CPU fallback")
        if x.dim() != 3 or x.shape[1] != 1:
            raise RuntimeError(f"Expected input of shape (B, 1, T), got {tuple(x.shape)}")
        self._native_handle()
        B, _, T = x.shape
        names, tensors = self._raw_parameters()
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
            return outs
        ref = []
        for layers in outs:
            r = [o.reference_layout() for o in layers]
            if layers[-1].period:
                r[-1] = torch.flatten(r[-1], 1, -1)  # (B, 1, H, P) -> (B, H * P)   (hifigan.py:414)
            ref.append(r)
        return ref
