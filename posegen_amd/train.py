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
        r._check(lib.pg_train_forward(r.handle, r._stream(), n, _ptr(rb), _ptr(sk), ps, _ptr(cy), cs, _ptr(cam), S, N, flags,
                                      None if pd is None else C.byref(pd), C.byref(structs[0]),
                                      C.byref(structs[1]) if N > 0 else None, C.byref(po)))
        ctx.caster, ctx.n_nets, ctx.nper, ctx.keep = caster, len(nets), nper, keep
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
        r._check(r.lib.pg_train_backward(r.handle, r._stream(), _ptr(a), _ptr(b), _ptr(c), _ptr(d), C.byref(structs[0]),
                                         C.byref(structs[1]) if ctx.n_nets > 1 else None))
        return (None, None) + tuple(grads)


class TrainableRayCaster(torch.nn.Module):
    """`HipRayCaster` with a gradient: the object to put under `render_kwargs_train['ray_caster']`
    (core/raycasters.py:156-165).  `parameters()` are the reference's tensors (same names under `network.` /
    `network_fine.`, nerf.py:57-88), so `get_grad_vars` + Adam (raycasters.py:186-228) work unchanged;
    `sync_inference_weights()` hands the current values to the fused inference kernels (validation renders)."""

    def __init__(self, caster: HipRayCaster):
        super().__init__()
        self.caster = caster
        self.cfg = caster.cfg
        dev = caster.renderer.device
        st = caster.renderer._state
        names = list(NET_TENSOR_ORDER) + (["framecodes.codes.weight"] if self.cfg.framecode_ch > 0 else [])
        self._names = names

        def net(sd):
            return torch.nn.ParameterDict({k.replace(".", "__"): torch.nn.Parameter(sd[k].to(dev).float().contiguous()) for k in names})
        self.network = net(st["network_fn_state_dict"])
        self.network_fine = net(st["network_fine_state_dict"]) if "network_fine_state_dict" in st else None

    @property
    def module(self):
        return self

    @property
    def renderer(self):
        return self.caster.renderer

    def _flat(self):
        out = [self.network[k.replace(".", "__")] for k in self._names]
        if self.network_fine is not None:
            out += [self.network_fine[k.replace(".", "__")] for k in self._names]
        return out

    def net_state_dict(self, which: int) -> Dict[str, torch.Tensor]:
        net = self.network if which == 0 else self.network_fine
        return {k: net[k.replace(".", "__")].detach().cpu() for k in self._names}

    def sync_inference_weights(self):
        """Re-pack the current parameter values for the fused inference kernels (after optimiser steps)."""
        self.caster.renderer.load_network(0, self.net_state_dict(0))
        if self.network_fine is not None:
            self.caster.renderer.load_network(1, self.net_state_dict(1))

    def forward(self, ray_batch, N_samples=None, kp_batch=None, skts=None, cyls=None, bones=None, cams=None,
                subject_idxs=None, lindisp=False, perturb=0., N_importance=0, raw_noise_std=0., ray_noise_std=0.,
                pytest=False, draws: Optional[Dict[str, torch.Tensor]] = None, **unused):
        if not (self.training and torch.is_grad_enabled()):
            return self.caster(ray_batch, N_samples=N_samples, kp_batch=kp_batch, skts=skts, cyls=cyls, bones=bones, cams=cams,
                               subject_idxs=subject_idxs, lindisp=lindisp, perturb=perturb, N_importance=N_importance,
                               raw_noise_std=raw_noise_std, ray_noise_std=ray_noise_std, pytest=pytest, draws=draws)
        if subject_idxs is not None:
            raise NotImplementedError("subject_idxs (multi-subject nets) are not supported")
        if skts is None or cyls is None:
            raise ValueError("skts and cyls are required (A-NeRF bone-relative rendering)")
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
