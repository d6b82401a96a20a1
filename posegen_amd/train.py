"""The A-NeRF training step on the HIP path: `render` in training mode with a gradient.

In the reference, `Trainer.train_batch` (core/trainer.py:232-275) renders a ray batch through
`render_kwargs_train['ray_caster']`, forms the loss from rgb_map / acc_map / rgb0 / acc0 (trainer.py:321-383) and calls
`loss.backward()` (trainer.py:463): autograd walks back through compositing, both MLPs and the embedding's inputs.
`TrainableRayCaster` gives the HIP caster the same property: its parameters are ordinary torch Parameters on the
device (an optimiser owns them), its call returns tensors that carry a grad_fn, and backward runs in the library
(`pg_train_forward` / `pg_train_backward`, csrc/pg_train.hip) -- exact fp32, activations kept on a tape inside the
handle.  disp_map and the alpha tensors are returned without a gradient (the reference's losses do not read them).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _ffi
from .raycaster import NET_TENSOR_ORDER, HipRayCaster, _dev_f32, _ptr, make_training_draws


class _RenderRaysFn(torch.autograd.Function):
    """outputs (rgb_map, acc_map, rgb0, acc0, disp_map, disp0) of one training-mode render_rays call; differentiable
    with respect to the 24 (+ frame codes) tensors of each net."""

    @staticmethod
    def forward(ctx, caster, call, *params):
        r = caster.renderer
        lib, dev = r.lib, r.device
        rb, sk, ps, cy, cs, cam, S, N, flags, draws = call
        n = rb.shape[0]
        nper = 24 + (1 if caster.cfg.framecode_ch > 0 else 0)
        nets = [params[:nper], params[nper:2 * nper]] if N > 0 else [params[:nper]]
        keep = [rb, sk, cy, cam]
        structs = []
        for tens in nets:
            st = _ffi.PgNetParams()
            for i in range(24):
                t = tens[i].detach()
                if not (t.is_contiguous() and t.dtype == torch.float32 and t.device == dev):
                    raise ValueError("training parameters must be contiguous float32 tensors on the caster's device")
                st.w[i] = t.data_ptr()
            if nper == 25:
                codes = tens[24].detach()
                ext = torch.cat([codes, codes.mean(0, keepdim=True)], 0).contiguous()      # embedding.py:25-26
                keep.append(ext)
                st.codes, st.n_codes = ext.data_ptr(), codes.shape[0]
            structs.append(st)
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        out = {"rgb_map": new(n, 3), "disp_map": new(n), "acc_map": new(n)}
        if N > 0:
            out.update({"rgb0": new(n, 3), "disp0": new(n), "acc0": new(n)})
        po = _ffi.PgOutputs()
        for k, v in out.items():
            setattr(po, k, v.data_ptr())
        pd = None
        if draws:
            pd = _ffi.PgTrainDraws()
            for k, t in draws.items():
                t = _dev_f32(t, dev)
                keep.append(t)
                setattr(pd, k, t.data_ptr())
        tape = C.c_int64(0)
        r._check(lib.pg_train_forward(r.handle, r._stream(), n, _ptr(rb), _ptr(sk), ps, _ptr(cy), cs, _ptr(cam), S, N, flags,
                                      None if pd is None else C.byref(pd), C.byref(structs[0]),
                                      C.byref(structs[1]) if N > 0 else None, C.byref(po), C.byref(tape)))
        ctx.caster, ctx.n_nets, ctx.nper, ctx.keep, ctx.tape_id = caster, len(nets), nper, keep, tape.value
        ctx.shapes = [tuple(p.shape) for p in params]
        zero = lambda k: out[k] if k in out else torch.zeros(0, device=dev)
        outs = (out["rgb_map"], out["acc_map"], zero("rgb0"), zero("acc0"), out["disp_map"], zero("disp0"))
        ctx.mark_non_differentiable(outs[4], outs[5])
        return outs

    @staticmethod
    def backward(ctx, g_rgb, g_acc, g_rgb0, g_acc0, _g_disp, _g_disp0):
        caster = ctx.caster
        r = caster.renderer
        dev = r.device
        grads = [torch.empty(s, device=dev, dtype=torch.float32) for s in ctx.shapes]
        structs = []
        for k in range(ctx.n_nets):
            st = _ffi.PgNetGrads()
            for i in range(24):
                st.w[i] = grads[k * ctx.nper + i].data_ptr()
            if ctx.nper == 25:
                st.codes = grads[k * ctx.nper + 24].data_ptr()
            structs.append(st)
        for g in grads[ctx.n_nets * ctx.nper:]:         # (parameters of a fine net that this call did not use)
            g.zero_()
        gp = lambda g: None if g is None or g.numel() == 0 else _dev_f32(g, dev)
        a, b, c, d = gp(g_rgb), gp(g_acc), gp(g_rgb0), gp(g_acc0)
        # (a forward pass in between has overwritten the handle's one tape: the library refuses the stale id)
        r._check(r.lib.pg_train_backward(r.handle, r._stream(), ctx.tape_id, _ptr(a), _ptr(b), _ptr(c), _ptr(d), C.byref(structs[0]),
                                         C.byref(structs[1]) if ctx.n_nets > 1 else None))
        caster._stale = True                            # (an optimiser step follows: the inference kernels' packed weights lag from here)
        return (None, None) + tuple(grads)


class _NetParams(torch.nn.Module):
    """The parameters of one NeRF net under the reference's names (core/networks/nerf.py:57-88): pts_linears.{0..7},
    alpha_linear, feature_linear, views_linears.0, rgb_linear (+ framecodes.codes, core/networks/embedding.py).  The
    nn.Linear / nn.Embedding modules are containers only -- the arithmetic runs in the library -- but they make
    `state_dict()` the reference checkpoint's `network_fn_state_dict`, `named_parameters()` its parameter list, and
    `pts_linears[i].parameters()` what get_grad_vars' freeze_weights walks (raycasters.py:194-203)."""

    def __init__(self, sd: Dict[str, torch.Tensor], framecodes: bool, device):
        super().__init__()
        lin = lambda name: self._linear(sd[f"{name}.weight"], sd[f"{name}.bias"], device)
        self.pts_linears = torch.nn.ModuleList([lin(f"pts_linears.{i}") for i in range(8)])
        self.alpha_linear = lin("alpha_linear")
        self.feature_linear = lin("feature_linear")
        self.views_linears = torch.nn.ModuleList([lin("views_linears.0")])
        self.rgb_linear = lin("rgb_linear")
        if framecodes:
            codes = sd["framecodes.codes.weight"]
            self.framecodes = torch.nn.Module()
            self.framecodes.codes = torch.nn.Embedding(codes.shape[0], codes.shape[1], device=device)
            with torch.no_grad():
                self.framecodes.codes.weight.copy_(codes)

    @staticmethod
    def _linear(w, b, device):
        m = torch.nn.Linear(w.shape[1], w.shape[0], device=device)
        with torch.no_grad():
            m.weight.copy_(w)
            m.bias.copy_(b)
        return m

    def tensors(self, names):
        p = dict(self.named_parameters())
        return [p[k] for k in names]


class _EmbedState(torch.nn.Module):
    """State of one CutoffEmbedder as the reference checkpoints it (core/cutoff_embedder.py:89-94): `cutoff_dist`
    Parameter[24] (requires_grad=False, opt_cutoff off) and the `tau` buffer, with the tau schedule of
    update_threshold (cutoff_embedder.py:176-183; the frequency schedule is off in every shipped config)."""

    def __init__(self, renderer, which: int, sd: Dict[str, torch.Tensor]):
        super().__init__()
        self._renderer, self._which = [renderer], which          # (a list: not a sub-module)
        self.cutoff_dist = torch.nn.Parameter(sd["cutoff_dist"].clone().float(), requires_grad=False)
        self.register_buffer("tau", sd["tau"].clone().float().reshape(()))
        self.init_tau = 20.0                                      # CutoffEmbedder(init_tau=20): cutoff_embedder.py:66

    def get_tau(self):
        return float(self.tau)

    def set(self, tau, cutoff_dist=None):
        if cutoff_dist is not None:
            with torch.no_grad():
                self.cutoff_dist.copy_(torch.as_tensor(cutoff_dist).float())
        self.tau.fill_(float(tau))
        self._renderer[0].set_embedder(self._which, float(tau), self.cutoff_dist.detach().cpu().numpy())

    def update_threshold(self, global_step, tau_step, tau_rate, alpha_step=None, alpha_target=None):
        self.set(min(self.init_tau * tau_rate ** (global_step / float(tau_step * 1000)), 2000.))


class TrainableRayCaster(torch.nn.Module):
    """`HipRayCaster` with a gradient: the object to put under `render_kwargs_train['ray_caster']`
    (core/raycasters.py:156-165).  It has the surface the reference's trainer touches: `get_networks()` /
    `get_embed_fns()` (so `get_grad_vars` + Adam, raycasters.py:186-228, work unchanged), `update_embed_fns`
    (trainer.py:265-266), `.module`, and `state_dict()` / `load_state_dict()` in the reference checkpoint layout
    (raycasters.py:752-788: `network_fn_state_dict` with `pts_linears.0.weight`, ..., `embed_state_dict`, ...), which
    `HipRayCaster.load_state_dict`, `load_raycaster` and the reference itself read back.
    `sync_inference_weights()` hands the current values to the fused inference kernels (validation renders); a render
    in eval mode does it by itself when a backward pass or a checkpoint load has changed the parameters since.

    `train_precision`: "fp32" (default: the reference trains in fp32, trainer.py:232-275; gradients within 1e-4 of its
    autograd) or "bf16" (opt-in: the tape and the large GEMMs' operands in bf16, fp32 accumulate).  It is the TRAINING
    step's arithmetic only and independent of the caster's rendering precision (`set_precision`)."""

    def __init__(self, caster: HipRayCaster, train_precision: str = "fp32"):
        super().__init__()
        if train_precision not in ("fp32", "bf16"):
            raise ValueError(f"train_precision must be 'fp32' or 'bf16', not {train_precision!r}")
        self.caster = caster
        self.train_precision = train_precision
        caster.renderer.set_train_precision(train_precision)
        self._stale = False                              # the inference kernels' packed weights lag the parameters
        self.cfg = caster.cfg
        dev = caster.renderer.device
        caster.renderer._refresh_state()                 # (a device-side weight load leaves the host copies to be fetched on demand)
        st = caster.renderer._state
        fc = self.cfg.framecode_ch > 0
        self._names = list(NET_TENSOR_ORDER) + (["framecodes.codes.weight"] if fc else [])
        self.network = _NetParams(st["network_fn_state_dict"], fc, dev)
        self.network_fine = _NetParams(st["network_fine_state_dict"], fc, dev) if "network_fine_state_dict" in st else None
        self.embed_fn = _EmbedState(caster.renderer, 0, st["embed_state_dict"])
        self.embedbones_fn = None                       # multires_bones = 0: a parameter-free identity Embedder
        self.embeddirs_fn = _EmbedState(caster.renderer, 1, st["embeddirs_state_dict"])

    @property
    def module(self):
        return self

    @property
    def renderer(self):
        return self.caster.renderer

    # ---- the reference RayCaster's accessors (core/raycasters.py:726-794) -----------------------------------------
    def get_networks(self):
        return self.network, self.network_fine

    def get_embed_fns(self):
        return self.embed_fn, self.embedbones_fn, self.embeddirs_fn

    def update_embed_fns(self, global_step, args):
        for fn in (self.embed_fn, self.embeddirs_fn):
            fn.update_threshold(global_step, args.cutoff_step, args.cutoff_rate, getattr(args, "freq_schedule_step", None),
                                self.cfg.multires - 1)

    def state_dict(self, *args, **kwargs):
        """The reference's checkpoint entries (raycasters.py:752-766): one state dict per sub-module."""
        cpu = lambda m: {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        sd = {"network_fn_state_dict": cpu(self.network), "embed_state_dict": cpu(self.embed_fn),
              "embedbones_state_dict": {}, "embeddirs_state_dict": cpu(self.embeddirs_fn)}
        if self.network_fine is not None:
            sd["network_fine_state_dict"] = cpu(self.network_fine)
        return sd

    def load_state_dict(self, ckpt, strict=True):
        """Resume from a checkpoint in that layout (written by this class, by HipRayCaster.state_dict or by the
        reference's trainer, trainer.py:496-507): parameters, embedder state and the inference kernels' weights."""
        self.network.load_state_dict(ckpt["network_fn_state_dict"], strict=strict)
        if self.network_fine is not None and ckpt.get("network_fine_state_dict") is not None:
            self.network_fine.load_state_dict(ckpt["network_fine_state_dict"], strict=strict)
        for fn, key in ((self.embed_fn, "embed_state_dict"), (self.embeddirs_fn, "embeddirs_state_dict")):
            e = ckpt.get(key)
            if e is not None and "tau" in e:
                fn.set(float(e["tau"]), e.get("cutoff_dist"))
            elif strict:
                raise KeyError(key)
        self.sync_inference_weights()

    def _flat(self):
        out = self.network.tensors(self._names)
        if self.network_fine is not None:
            out += self.network_fine.tensors(self._names)
        return out

    def net_state_dict(self, which: int) -> Dict[str, torch.Tensor]:
        net = self.network if which == 0 else self.network_fine
        return {k: v.detach().cpu() for k, v in net.state_dict().items()}

    def _state_provider(self, net):
        """What the inner caster's state_dict() / parameters() fetch after a device-side sync: the net's tensors -- after another
        sync if the parameters have moved since, so that the state handed out is the state the kernels render with."""
        def provide():
            if self._stale:
                self.sync_inference_weights()
            return net.state_dict()
        return provide

    def sync_inference_weights(self, on_device: Optional[bool] = None):
        """Hand the current parameter values to the fused inference kernels (after optimiser steps).  on_device (default:
        whenever the renderer has one device): the packed weight images are re-formed on the GPU from the parameter tensors
        themselves (pg_load_weights_device: no host copy, bitwise the host packing); False: through the host (pg_load_weights)."""
        r = self.caster.renderer
        if on_device is None:
            on_device = len(getattr(r, "devices", [0])) <= 1
        nets = [(0, self.network)] + ([(1, self.network_fine)] if self.network_fine is not None else [])
        for which, net in nets:
            if on_device:
                p = dict(net.named_parameters())
                codes = p["framecodes.codes.weight"] if self.cfg.framecode_ch > 0 else None
                r.load_network_device(which, [p[k] for k in NET_TENSOR_ORDER], codes, state_provider=self._state_provider(net))
            else:
                r.load_network(which, self.net_state_dict(which))
        self._stale = False

    def _check_unused(self, unused):
        """The reference's keywords the kernels do not honour are refused, in training and in eval mode alike."""
        if not unused:
            return
        unused = dict(unused)
        self.caster._check_preproc_kwargs(unused.pop("preproc_kwargs", None))
        if unused.pop("nerf_type", "nerf") != "nerf" or not unused.pop("use_viewdirs", True):
            raise NotImplementedError("only nerf_type='nerf' with view directions is on the HIP path")
        if unused.get("network_fine") is not None and unused["network_fine"] is not self.network_fine:
            raise NotImplementedError("a network_fine argument other than this caster's own fine network")
        for k in ("retraw", "verbose", "ext_scale", "network_fine"):
            unused.pop(k, None)
        if unused:
            raise TypeError(f"TrainableRayCaster.forward: unexpected keyword arguments {sorted(unused)}")

    def forward(self, ray_batch, N_samples=None, kp_batch=None, skts=None, cyls=None, bones=None, cams=None,
                subject_idxs=None, lindisp=False, perturb=0., N_importance=0, raw_noise_std=0., ray_noise_std=0.,
                pytest=False, draws: Optional[Dict[str, torch.Tensor]] = None, **unused):
        self._check_unused(unused)
        if not (self.training and torch.is_grad_enabled()):
            if self._stale:                              # a backward pass has run since the last packing: render what was trained
                self.sync_inference_weights()
            return self.caster(ray_batch, N_samples=N_samples, kp_batch=kp_batch, skts=skts, cyls=cyls, bones=bones, cams=cams,
                               subject_idxs=subject_idxs, lindisp=lindisp, perturb=perturb, N_importance=N_importance,
                               raw_noise_std=raw_noise_std, ray_noise_std=ray_noise_std, pytest=pytest, draws=draws)
        if subject_idxs is not None:
            raise NotImplementedError("subject_idxs (multi-subject nets) are not supported")
        if skts is None or cyls is None:
            raise ValueError("skts and cyls are required (A-NeRF bone-relative rendering)")
        # The backward pass differentiates with respect to the networks' tensors only.  The reference's pose
        # optimisation (popt_layer, trainer.py:496-515) backpropagates into skts / kp through the embedding: refused
        # here rather than left without a gradient (out of scope).
        for name, t in (("ray_batch", ray_batch), ("skts", skts), ("kp_batch", kp_batch), ("cyls", cyls), ("bones", bones)):
            if torch.is_tensor(t) and t.requires_grad:
                raise NotImplementedError(f"{name} requires a gradient: the HIP training step has no gradient for poses / rays "
                                          "(pose optimisation is not on the HIP path); pass a detached tensor")
        r = self.caster.renderer
        cfg = self.cfg
        S = cfg.n_samples if N_samples is None else int(N_samples)
        N = int(N_importance or 0)
        if N > 0 and self.network_fine is None:
            raise ValueError("N_importance > 0 needs the fine network")
        rb = _dev_f32(ray_batch, r.device)
        n = rb.shape[0]
        if rb.shape[1] != 11:
            pad = torch.zeros(n, 11, device=r.device)
            pad[:, :min(11, rb.shape[1])] = rb[:, :11]
            rb = pad
        sk, ps = r._pose_args(skts, n)
        cy, cs = r._cyl_args(cyls, n)
        cam = None
        if cams is not None:
            cam = _dev_f32(torch.as_tensor(cams).reshape(-1), r.device)
            if cam.shape[0] == 1 and n > 1:
                cam = cam.expand(n).contiguous()
        if draws is None and (perturb or raw_noise_std or ray_noise_std):
            draws = make_training_draws(n, S, N, perturb, raw_noise_std, ray_noise_std, pytest=pytest,
                                        density_scale=cfg.density_scale, device=r.device)
        if draws and N == 0:
            draws = {k: v for k, v in draws.items() if k not in ("u_rand", "noise1")}
        keep = r._chunk
        r.set_chunk(max(n, 1))                          # one call = one nanmean group (ray_utils.py:292-344)
        try:
            flags = _ffi.PG_FLAG_LINDISP if lindisp else 0
            rgb, acc, rgb0, acc0, disp, disp0 = _RenderRaysFn.apply(self, (rb, sk, ps, cy, cs, cam, S, N, flags, draws or None), *self._flat())
        finally:
            r.set_chunk(keep)
        out = {"rgb_map": rgb, "disp_map": disp, "acc_map": acc}
        if N > 0:
            out.update({"rgb0": rgb0, "disp0": disp0, "acc0": acc0})
        return out
