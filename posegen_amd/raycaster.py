"""HIP-backed drop-in for the reference's `RayCaster` (core/raycasters.py:326-794).

`HipRenderer` is a typed wrapper around one `pg_handle` (device memory and streams
come from PyTorch-ROCm: plumbing only).  `HipRayCaster` is the duck-typed object the
reference stores under ``render_kwargs['ray_caster']`` and calls at
core/trainer.py:74: same call signature, same returned dict keys
(`_collect_outputs`, raycasters.py:711-724), same checkpoint key scheme
(`state_dict` / `load_state_dict`, raycasters.py:752-788).

Forward values only (no autograd graph).  Eval mode is the measured path; the training-mode
arguments (perturb, raw_noise_std, ray_noise_std, pytest) are honoured with the random numbers
drawn on the host side of the ABI (`training_draws`, pg_train_draws).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _ffi
from .config import PREC_BF16, PREC_BY_NAME, PREC_NAMES, RenderConfig

NET_TENSOR_ORDER = ([f"pts_linears.{l}.{k}" for l in range(8) for k in ("weight", "bias")]
                    + [f"{n}.{k}" for n in ("alpha_linear", "feature_linear", "views_linears.0", "rgb_linear")
                       for k in ("weight", "bias")])


def _density_act(density_type: str) -> int:
    """--density_type -> PG_ACT_* (get_density_fn, core/raycasters.py:230-238 raises on anything else)."""
    try:
        return {"relu": _ffi.PG_ACT_RELU, "softplus": _ffi.PG_ACT_SOFTPLUS}[density_type]
    except KeyError:
        raise NotImplementedError(f"density activation {density_type} is undefined") from None


def _np32(x) -> np.ndarray:
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x), dtype=np.float32)


def _dev_f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class HipRenderer:
    """Owns a pg_handle on one HIP device."""

    def __init__(self, cfg: RenderConfig, device="cuda:0", precision=PREC_BF16, devices=None):
        """`devices`: HIP device indices for in-process multi-GPU frame rendering (render_frames /
        render_path); the first one is `device`, where every ray-level call runs.  An index may
        repeat (two workers on one GPU)."""
        if isinstance(precision, str):
            precision = PREC_BY_NAME[precision]
        self.cfg = cfg
        if devices is not None:
            devices = [int(torch.device(d).index) if not isinstance(d, int) else int(d) for d in devices]
            device = f"cuda:{devices[0]}"
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _ffi.HipLibraryError("HipRenderer needs a HIP device (torch device 'cuda:N'); "
                                       "the render path has no CPU fallback")
        self.lib = _ffi.load_library()
        self.precision = int(precision)
        pc = _ffi.PgConfig(n_joints=cfg.n_joints, multires=cfg.multires, multires_views=cfg.multires_views,
                           multires_bones=cfg.multires_bones, net_depth=cfg.net_depth, net_width=cfg.net_width,
                           skip_layer=cfg.skips[0], view_width=cfg.net_width // 2,
                           framecode_ch=cfg.framecode_ch, n_framecodes=cfg.n_framecodes, chunk=cfg.chunk,
                           precision=self.precision, cutoff_dist=cfg.cutoff_dist,
                           density_scale=cfg.density_scale, rgb_eps=cfg.rgb_eps,
                           softplus_shift=float(cfg.softplus_shift), density_act=_density_act(cfg.density_type), reserved0=0)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        dev_ids = [idx] if devices is None else devices
        ids = (C.c_int * len(dev_ids))(*dev_ids)
        self.devices = list(dev_ids)
        h = C.c_void_p()
        rc = self.lib.pg_create(C.byref(pc), len(dev_ids), ids, C.byref(h))
        _ffi.check(self.lib, None, rc)
        self.handle = h
        self._state: Dict[str, dict] = {}
        self._state_lazy: Dict[str, object] = {}      # state dicts to fetch from their owner on demand (load_network_device)
        self._chunk = cfg.chunk

    # -- lifetime ---------------------------------------------------------------------
    def close(self):
        if getattr(self, "handle", None):
            self.lib.pg_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        _ffi.check(self.lib, self.handle, rc)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- state ------------------------------------------------------------------------
    def load_network(self, which: int, sd: Dict[str, np.ndarray]):
        """which 0 = coarse ('network_fn_state_dict'), 1 = fine ('network_fine_state_dict')."""
        arrs = [_np32(sd[k]) for k in NET_TENSOR_ORDER]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        shp = (C.c_int64 * (2 * len(arrs)))()
        for i, a in enumerate(arrs):
            shp[2 * i] = a.shape[0]
            shp[2 * i + 1] = a.shape[1] if a.ndim == 2 else 1
        self._check(self.lib.pg_load_weights(self.handle, which, ptrs, shp, len(arrs)))
        if self.cfg.framecode_ch > 0:
            codes = _np32(sd["framecodes.codes.weight"])
            self._check(self.lib.pg_set_framecodes(self.handle, which, codes.ctypes.data, codes.shape[0]))
        self._state["network_fn_state_dict" if which == 0 else "network_fine_state_dict"] = {
            k: torch.from_numpy(_np32(v).copy()) for k, v in sd.items()}

    def load_network_device(self, which: int, tensors, codes=None, state_provider=None):
        """New values of an already loaded net from DEVICE tensors (pg_load_weights_device): `tensors` = the 24 parameters in
        NET_TENSOR_ORDER (contiguous fp32 on this device), `codes` the frame codes [n,16] when the config has them.  The packed
        images of the fast paths are re-formed on the device; nothing is copied to the host.  `state_provider()` must return the
        net's state dict (CPU tensors) when somebody asks this renderer for it (state_dict / parameters): until then the host
        copy kept for checkpoints is stale."""
        ts = [t.detach() for t in tensors]
        for t in ts:
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError("load_network_device: contiguous float32 device tensors expected")
        ptrs = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        cptr, ncodes = None, 0
        if self.cfg.framecode_ch > 0:
            codes = codes.detach()
            if not (codes.is_cuda and codes.dtype == torch.float32 and codes.is_contiguous()):
                raise ValueError("load_network_device: contiguous float32 device frame codes expected")
            cptr, ncodes = codes.data_ptr(), codes.shape[0]
        self._check(self.lib.pg_load_weights_device(self.handle, self._stream(), which, ptrs, len(ts), cptr, ncodes))
        self._state_lazy["network_fn_state_dict" if which == 0 else "network_fine_state_dict"] = state_provider

    def _refresh_state(self):
        """the host-side state dicts brought up to date after device-side weight loads"""
        for key, provider in list(self._state_lazy.items()):
            if provider is not None:
                self._state[key] = {k: v.detach().cpu().clone() for k, v in provider().items()}
            del self._state_lazy[key]

    def set_embedder(self, which: int, tau: float, cutoff_dist=None):
        """which 0 = embed_fn, 1 = embeddirs_fn (cutoff_embedder.py:89-94)."""
        cd = _np32(np.full(24, self.cfg.cutoff_dist) if cutoff_dist is None else cutoff_dist)
        self._check(self.lib.pg_set_embedder(self.handle, which, cd.ctypes.data, float(tau)))
        self._state["embed_state_dict" if which == 0 else "embeddirs_state_dict"] = {
            "cutoff_dist": torch.from_numpy(cd.copy()), "tau": torch.tensor(float(tau))}

    def set_precision(self, precision):
        if isinstance(precision, str):
            precision = PREC_BY_NAME[precision]
        self._check(self.lib.pg_set_precision(self.handle, int(precision)))
        self.precision = int(precision)

    def set_chunk(self, chunk: int):
        if int(chunk) != self._chunk:
            self._check(self.lib.pg_set_chunk(self.handle, int(chunk)))
            self._chunk = int(chunk)

    def set_onchip(self, mode="auto"):
        """Which form of the 16x16x32 kernel calls with >= 64 samples per ray take (pg_set_onchip): "records" (per-ray records in
        HBM), "auto" (on chip up to 112 samples per ray: the faster of the two, the default) or "always" (on chip whatever the
        sample count: no record workspace)."""
        modes = {"records": 0, "auto": 1, "always": 2, 0: 0, 1: 1, 2: 2}
        if mode not in modes:
            raise ValueError(f"set_onchip: mode must be 'records', 'auto' or 'always', not {mode!r}")
        self._check(self.lib.pg_set_onchip(self.handle, modes[mode]))

    def set_far_skip(self, on=True):
        """Test / measurement aid (pg_set_far_skip): off = the fused kernels compute every limb for every point."""
        self._check(self.lib.pg_set_far_skip(self.handle, 1 if on else 0))

    def set_train_precision(self, precision="fp32"):
        """Arithmetic of the training step (pg_set_train_precision): "fp32" (the reference's, default) or "bf16" (bf16 tape
        and GEMM operands); independent of the rendering precision."""
        p = PREC_BY_NAME[precision] if isinstance(precision, str) else int(precision)
        self._check(self.lib.pg_set_train_precision(self.handle, p))

    def profile_enable(self, on=True):
        self._check(self.lib.pg_profile_enable(self.handle, 1 if on else 0))

    def profile_read(self):
        """(launches, summed device ms, points) of the fused embed+MLP kernel since the last read."""
        n, ms, pts = C.c_int64(), C.c_double(), C.c_int64()
        self._check(self.lib.pg_profile_read(self.handle, C.byref(n), C.byref(ms), C.byref(pts)))
        return n.value, ms.value, pts.value

    def profile_read_aux(self):
        """(launches, summed device ms) of the per-ray record kernel in front of the factorised 16-bit launches."""
        n, ms = C.c_int64(), C.c_double()
        self._check(self.lib.pg_profile_read_aux(self.handle, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def device_info(self):
        n, k = C.c_int32(), C.c_int32()
        self._check(self.lib.pg_device_info(self.handle, C.byref(n), C.byref(k)))
        return {"n_cu": n.value, "clock_khz": k.value}

    def calibrate_mfma(self, f16=False, lds_fed=False, min_ms=20.0):
        """TFLOP/s this device sustains on bare 32x32x16 MFMAs, operands in registers or (lds_fed) the A
        operand read from LDS per MFMA (pg_calibrate_mfma); synchronous."""
        tf, ms = C.c_double(), C.c_double()
        self._check(self.lib.pg_calibrate_mfma(self.handle, 1 if f16 else 0, int(lds_fed), float(min_ms),
                                               C.byref(tf), C.byref(ms)))
        return {"tflops": tf.value, "ms": ms.value}

    def query(self, precision=None):
        sb, mf = C.c_int64(), C.c_int64()
        self._check(self.lib.pg_query(self.handle, self.precision if precision is None else int(precision),
                                      C.byref(sb), C.byref(mf)))
        return {"stream_bytes": sb.value, "mfma_per_group": mf.value}

    # -- the hot path -----------------------------------------------------------------
    def _pose_args(self, skts: torch.Tensor, n: int):
        """[1|n,24,4,4] -> (contiguous device tensor, stride in floats)."""
        if skts.dim() == 3:
            skts = skts[None]
        if skts.shape[0] == 1 or skts.stride(0) == 0:
            return _dev_f32(skts[:1], self.device), 0
        if skts.shape[0] != n:
            raise ValueError(f"skts has {skts.shape[0]} poses for {n} rays")
        return _dev_f32(skts, self.device), 24 * 16

    def _cyl_args(self, cyls: torch.Tensor, n: int):
        if cyls.dim() == 1:
            cyls = cyls[None]
        if cyls.shape[0] == 1 or cyls.stride(0) == 0:
            return _dev_f32(cyls[:1], self.device), 0
        if cyls.shape[0] != n:
            raise ValueError(f"cyls has {cyls.shape[0]} rows for {n} rays")
        return _dev_f32(cyls, self.device), 5

    def render_rays(self, ray_batch: torch.Tensor, skts: torch.Tensor, cyls: torch.Tensor,
                    cams: Optional[torch.Tensor] = None, n_samples: Optional[int] = None,
                    n_importance: Optional[int] = None, lindisp: bool = False,
                    want_alpha: bool = True, extras: bool = False,
                    draws: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """One `RayCaster.render_rays` call (core/raycasters.py:361-474).  `draws` = None: eval mode.
        Otherwise the random numbers of a training-mode call (pg_train_draws, posegen_hip.h): any of
        t_rand [n,S], u_rand [n,N], noise0 [n,S], noise1 [n,S+N], ray_noise [n,S+N,3]."""
        cfg = self.cfg
        S = cfg.n_samples if n_samples is None else int(n_samples)
        N = cfg.n_importance if n_importance is None else int(n_importance)
        rb = _dev_f32(ray_batch, self.device)
        n = rb.shape[0]
        if rb.dim() != 2 or rb.shape[1] < 8:
            raise ValueError("ray_batch must be [n, >=8] (o, d, near, far [, viewdir])")
        if rb.shape[1] != 11:
            pad = torch.zeros(n, 11, device=self.device)
            pad[:, :min(11, rb.shape[1])] = rb[:, :11]
            rb = pad
        sk, ps = self._pose_args(skts, n)
        cy, cs = self._cyl_args(cyls, n)
        cam = None
        if cams is not None:
            cam = _dev_f32(cams.reshape(-1), self.device)
            if cam.shape[0] == 1 and n > 1:
                cam = cam.expand(n).contiguous()
        dev = self.device
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        SF = S + N
        out = {"rgb_map": new(n, 3), "disp_map": new(n), "acc_map": new(n)}
        if want_alpha:
            out["alpha"] = new(n, SF)
        if N > 0:
            out.update({"rgb0": new(n, 3), "disp0": new(n), "acc0": new(n)})
            if want_alpha:
                out["alpha0"] = new(n, S)
        ex = {}
        if extras:
            ex = {"near_far": new(n, 2), "z_coarse": new(n, S), "raw_coarse": new(n, S, 4), "weights0": new(n, S)}
            if N > 0:
                ex.update({"z_fine": new(n, SF), "raw_fine": new(n, SF, 4)})
        po = _ffi.PgOutputs()
        for k in ("rgb_map", "disp_map", "acc_map", "alpha", "rgb0", "disp0", "acc0", "alpha0"):
            setattr(po, k, out[k].data_ptr() if k in out else None)
        for k in ("near_far", "z_coarse", "z_fine", "raw_coarse", "raw_fine", "weights0"):
            setattr(po, k, ex[k].data_ptr() if k in ex else None)
        flags = _ffi.PG_FLAG_LINDISP if lindisp else 0
        if draws:
            shapes = {"t_rand": (n, S), "u_rand": (n, N), "noise0": (n, S), "noise1": (n, SF), "ray_noise": (n, SF, 3)}
            unknown = set(draws) - set(shapes)
            if unknown:
                raise ValueError(f"unknown draws {sorted(unknown)}; expected a subset of {sorted(shapes)}")
            pd, keep = _ffi.PgTrainDraws(), []
            for k, shp in shapes.items():
                t = draws.get(k)
                if t is None or (N == 0 and k in ("u_rand", "noise1")):
                    continue
                t = _dev_f32(t, dev)
                if tuple(t.shape) != shp:
                    raise ValueError(f"draws[{k!r}] must be {shp}, got {tuple(t.shape)}")
                keep.append(t)
                setattr(pd, k, t.data_ptr())
            if n > 0:
                self._check(self.lib.pg_render_rays_train(self.handle, self._stream(), n, _ptr(rb), _ptr(sk), ps, _ptr(cy),
                                                          cs, _ptr(cam), S, N, flags, C.byref(pd), C.byref(po)))
        elif n > 0:
            self._check(self.lib.pg_render_rays(self.handle, self._stream(), n, _ptr(rb), _ptr(sk), ps, _ptr(cy), cs,
                                                _ptr(cam), S, N, flags, C.byref(po)))
        if extras:
            out["extras"] = ex
        return out

    def render_frame(self, H: int, W: int, focal, c2w, box, skts: torch.Tensor, cyl: torch.Tensor,
                     center=None, cam: Optional[float] = None, near: float = 0., far: float = 1.,
                     n_samples: Optional[int] = None, n_importance: Optional[int] = None, lindisp: bool = False,
                     bg: Optional[torch.Tensor] = None, base_bg: float = 0., want_uint8: bool = False):
        """One frame entirely on the device (pg_render_frame): rays of the pixels in `box` =
        ((tl_x, tl_y), (br_x, br_y)), render, scatter over the background.  Returns device
        tensors rgb [H,W,3], disp [H,W,1], acc [H,W,1] (+ rgb8 uint8 [H,W,3])."""
        cfg = self.cfg
        S = cfg.n_samples if n_samples is None else int(n_samples)
        N = cfg.n_importance if n_importance is None else int(n_importance)
        f = np.asarray(focal.detach().cpu() if isinstance(focal, torch.Tensor) else focal, dtype=np.float64).reshape(-1)
        fx, fy = (float(f[0]), float(f[0])) if f.size < 2 else (float(f[0]), float(f[1]))
        cx, cy = (W * 0.5, H * 0.5) if center is None else (float(center[0]), float(center[1]))
        c2w_h = np.ascontiguousarray(np.asarray(c2w.detach().cpu() if isinstance(c2w, torch.Tensor) else c2w,
                                                dtype=np.float32)[:3, :4])
        (tlx, tly), (brx, bry) = box
        sk, _ = self._pose_args(skts, 1)
        cy_t, _ = self._cyl_args(cyl, 1)
        dev = self.device
        rgb = torch.empty(H, W, 3, device=dev)
        disp = torch.empty(H, W, 1, device=dev)
        acc = torch.empty(H, W, 1, device=dev)
        rgb8 = torch.empty(H, W, 3, device=dev, dtype=torch.uint8) if want_uint8 else None
        bgt = None if bg is None else _dev_f32(bg.reshape(H * W, 3), dev)
        self._check(self.lib.pg_render_frame(
            self.handle, self._stream(), int(H), int(W), c2w_h.ctypes.data_as(C.POINTER(C.c_float)),
            (C.c_float * 4)(fx, fy, cx, cy), (C.c_int * 4)(int(tlx), int(tly), int(brx), int(bry)),
            float(near), float(far), _ptr(sk), _ptr(cy_t), -1.0 if cam is None else float(cam), S, N,
            _ffi.PG_FLAG_LINDISP if lindisp else 0, _ptr(bgt), float(base_bg), _ptr(rgb), _ptr(disp), _ptr(acc),
            _ptr(rgb8)))
        return (rgb, disp, acc, rgb8) if want_uint8 else (rgb, disp, acc)

    def _frame_args(self, H, W, focal, c2w, box, center):
        f = np.asarray(focal.detach().cpu() if isinstance(focal, torch.Tensor) else focal, dtype=np.float64).reshape(-1)
        fx, fy = (float(f[0]), float(f[0])) if f.size < 2 else (float(f[0]), float(f[1]))
        cx, cy = (W * 0.5, H * 0.5) if center is None else (float(center[0]), float(center[1]))
        c2w_h = np.ascontiguousarray(np.asarray(c2w.detach().cpu() if isinstance(c2w, torch.Tensor) else c2w,
                                                dtype=np.float32)[:3, :4])
        (tlx, tly), (brx, bry) = box
        return (c2w_h, (C.c_float * 4)(fx, fy, cx, cy), (C.c_int * 4)(int(tlx), int(tly), int(brx), int(bry)))

    def render_frame_range(self, H: int, W: int, focal, c2w, box, skts: torch.Tensor, cyl: torch.Tensor,
                           ray_begin: int, ray_end: int, center=None, cam: Optional[float] = None, near: float = 0.,
                           far: float = 1., n_samples: Optional[int] = None, n_importance: Optional[int] = None,
                           lindisp: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Rays [ray_begin, ray_end) of the box's row-major ray list -> their maps, packed as one device tensor
        [5, n]: rows 0-2 hold rgb_map [n,3] (flat), row 3 disp_map, row 4 acc_map (pg_render_frame_range; the
        unit of work of the multi-process partition, dist.plan_tasks).  `ray_begin` must be 0 or a multiple
        of the nanmean group size, so the values are those of the whole frame."""
        cfg = self.cfg
        S = cfg.n_samples if n_samples is None else int(n_samples)
        N = cfg.n_importance if n_importance is None else int(n_importance)
        n = int(ray_end) - int(ray_begin)
        buf = torch.empty(5 * n, device=self.device) if out is None else out
        if buf.numel() != 5 * n or buf.dtype != torch.float32 or not buf.is_contiguous():
            raise ValueError("render_frame_range: `out` must be a contiguous float32 tensor of 5 * n elements")
        if n > 0:
            c2w_h, intr, bx = self._frame_args(H, W, focal, c2w, box, center)
            sk, _ = self._pose_args(skts, 1)
            cy_t, _ = self._cyl_args(cyl, 1)
            flat = buf.view(-1)
            self._check(self.lib.pg_render_frame_range(
                self.handle, self._stream(), int(H), int(W), c2w_h.ctypes.data_as(C.POINTER(C.c_float)), intr, bx,
                float(near), float(far), _ptr(sk), _ptr(cy_t), -1.0 if cam is None else float(cam), S, N,
                _ffi.PG_FLAG_LINDISP if lindisp else 0, int(ray_begin), int(ray_end),
                _ptr(flat[:3 * n]), _ptr(flat[3 * n:4 * n]), _ptr(flat[4 * n:])))
        return buf

    def compose_frame(self, H: int, W: int, box, rgb_map: torch.Tensor, disp_map: torch.Tensor, acc_map: torch.Tensor,
                      bg: Optional[torch.Tensor] = None, base_bg: float = 0., want_uint8: bool = False):
        """The maps of a whole box over the background (pg_compose_frame): device tensors rgb [H,W,3],
        disp [H,W,1], acc [H,W,1] (+ rgb8)."""
        dev = self.device
        (tlx, tly), (brx, bry) = box
        rgb = torch.empty(H, W, 3, device=dev)
        disp = torch.empty(H, W, 1, device=dev)
        acc = torch.empty(H, W, 1, device=dev)
        rgb8 = torch.empty(H, W, 3, device=dev, dtype=torch.uint8) if want_uint8 else None
        bgt = None if bg is None else _dev_f32(bg.reshape(H * W, 3), dev)
        rm, dm, am = (_dev_f32(rgb_map, dev), _dev_f32(disp_map, dev), _dev_f32(acc_map, dev))
        self._check(self.lib.pg_compose_frame(
            self.handle, self._stream(), int(H), int(W), (C.c_int * 4)(int(tlx), int(tly), int(brx), int(bry)),
            _ptr(rm), _ptr(dm), _ptr(am), _ptr(bgt), float(base_bg), _ptr(rgb), _ptr(disp), _ptr(acc), _ptr(rgb8)))
        return (rgb, disp, acc, rgb8) if want_uint8 else (rgb, disp, acc)

    @property
    def n_devices(self) -> int:
        return len(self.devices)

    def render_frames(self, H: int, W: int, focals, c2ws, boxes, skts, cyls, centers=None, cams=None,
                      near: float = 0., far: float = 1., n_samples: Optional[int] = None,
                      n_importance: Optional[int] = None, lindisp: bool = False, bg=None, base_bg: float = 0.,
                      want_uint8: bool = False):
        """Frames on ALL devices of the handle (pg_render_frames), host in / host out: numpy arrays
        rgbs [F,H,W,3], disps [F,H,W,1], accs [F,H,W,1] (+ rgb8 uint8 [F,H,W,3]).  `boxes` = list of
        ((tl_x, tl_y), (br_x, br_y)); skts [F,24,4,4], cyls [F,5] (one pose per frame)."""
        cfg = self.cfg
        S = cfg.n_samples if n_samples is None else int(n_samples)
        N = cfg.n_importance if n_importance is None else int(n_importance)
        F = len(boxes)
        c2w_h = np.ascontiguousarray(np.stack([np.asarray(torch.as_tensor(c).detach().cpu(), dtype=np.float32)[:3, :4]
                                               for c in c2ws]))
        intr = np.zeros((F, 4), dtype=np.float32)
        for i in range(F):
            f = np.asarray(torch.as_tensor(focals[i] if np.ndim(focals) > 0 else focals).detach().cpu(), dtype=np.float64).reshape(-1)
            intr[i, 0], intr[i, 1] = (f[0], f[0]) if f.size < 2 else (f[0], f[1])
            intr[i, 2], intr[i, 3] = (W * 0.5, H * 0.5) if centers is None else (float(centers[i][0]), float(centers[i][1]))
        bx = np.ascontiguousarray(np.array([[b[0][0], b[0][1], b[1][0], b[1][1]] for b in boxes], dtype=np.int32))
        sk = _np32(skts).reshape(-1, 24, 4, 4)
        cy = _np32(cyls).reshape(-1, 5)
        if sk.shape[0] == 1 and F > 1:
            sk = np.ascontiguousarray(np.repeat(sk, F, 0))
        if cy.shape[0] == 1 and F > 1:
            cy = np.ascontiguousarray(np.repeat(cy, F, 0))
        if sk.shape[0] != F or cy.shape[0] != F:
            raise ValueError(f"need one pose per frame: {sk.shape[0]} skts / {cy.shape[0]} cyls for {F} frames")
        cm = None if cams is None else _np32(cams).reshape(-1)
        bgh = None if bg is None else _np32(bg).reshape(H * W, 3)
        # results in page-locked memory when they are small enough to pin as a whole: the library then copies every
        # frame straight into them on a copy stream while the next one renders; pageable arrays (long paths) are
        # filled through the library's own pinned staging
        pin = F * H * W * (23 if want_uint8 else 20) <= (256 << 20) and torch.cuda.is_available()
        new = lambda c, dt: torch.empty((F, H, W, c), dtype=dt, pin_memory=pin).numpy()
        rgbs, disps, accs = new(3, torch.float32), new(1, torch.float32), new(1, torch.float32)
        rgb8 = new(3, torch.uint8) if want_uint8 else None
        hp = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
        self._check(self.lib.pg_render_frames(
            self.handle, F, int(H), int(W), hp(c2w_h), hp(intr), hp(bx), float(near), float(far), hp(sk), hp(cy), hp(cm),
            S, N, _ffi.PG_FLAG_LINDISP if lindisp else 0, hp(bgh), float(base_bg), hp(rgbs), hp(disps), hp(accs), hp(rgb8)))
        return (rgbs, disps, accs, rgb8) if want_uint8 else (rgbs, disps, accs)

    def query_density(self, pts: torch.Tensor, skts: torch.Tensor, which: Optional[int] = None) -> torch.Tensor:
        """Raw density (alpha_linear output, no activation) of net `which` (default: the fine net if
        loaded, like the reference) at explicit points [...,3] for one pose: render_pts_density
        (core/raycasters.py:598-646).  Returns a device tensor [..., 1]."""
        if which is None:
            which = 1 if "network_fine_state_dict" in self._state else 0
        p = _dev_f32(torch.as_tensor(pts).reshape(-1, 3), self.device)
        n = p.shape[0]
        sk, _ = self._pose_args(skts, 1)
        raw = torch.empty(n, 4, device=self.device)
        if n > 0:
            self._check(self.lib.pg_query_density(self.handle, self._stream(), int(which), n, _ptr(p), _ptr(sk), _ptr(raw)))
        return raw[:, 3:4].reshape(*torch.as_tensor(pts).shape[:-1], 1)

    def mesh_density(self, kps: torch.Tensor, skts: torch.Tensor, radius: float = 1.0, res: int = 64,
                     which: Optional[int] = None) -> torch.Tensor:
        """Raw density on the (res+1)^3 grid of half-width `radius` around the root joint, laid out
        like RayCaster.render_mesh_density (core/raycasters.py:579-596)."""
        t = np.linspace(-radius, radius, res + 1)
        grid = np.stack(np.meshgrid(t, t, t), axis=-1).astype(np.float32)
        sh = grid.shape
        pts = torch.tensor(grid.reshape(-1, 3)) + torch.as_tensor(kps, dtype=torch.float32).reshape(-1, 24, 3)[0, 0]
        d = self.query_density(pts, skts, which)
        return d.reshape(*sh[:-1]).transpose(1, 0)

    def pose_kinematics(self, bones: torch.Tensor, rest_pose, parents=None, want_l2ws: bool = False):
        """bones [F,24,3] axis-angle (any device/dtype) -> device kps [F,24,3] f32, skts [F,24,4,4] f32
        (+ l2ws f64), computed on the device in float64 (pg_pose_kinematics; the host equivalent is
        skeleton.bones_to_pose)."""
        from .skeleton import SMPLSkeleton
        b = torch.as_tensor(bones).to(device=self.device, dtype=torch.float64).contiguous()
        F = b.shape[0]
        par = np.ascontiguousarray(np.asarray(SMPLSkeleton.joint_trees if parents is None else parents, dtype=np.int32))
        rp = np.asarray(rest_pose).reshape(24, 3)             # offsets in the rest pose's own dtype, like
        offs = rp.copy()                                      # get_smpl_l2ws (float32 for smpl_rest_pose)
        offs[1:] = rp[1:] - rp[par[1:]]
        rest = np.ascontiguousarray(offs.astype(np.float64))
        kps = torch.empty(F, 24, 3, device=self.device)
        skts = torch.empty(F, 24, 4, 4, device=self.device)
        l2ws = torch.empty(F, 24, 4, 4, device=self.device, dtype=torch.float64) if want_l2ws else None
        self._check(self.lib.pg_pose_kinematics(self.handle, self._stream(), F, _ptr(b),
                                                rest.ctypes.data_as(C.POINTER(C.c_double)),
                                                par.ctypes.data_as(C.POINTER(C.c_int32)), _ptr(kps), _ptr(skts), _ptr(l2ws)))
        return (kps, skts, l2ws) if want_l2ws else (kps, skts)

    def pose_boxes(self, kps: torch.Tensor, c2ws, H: int, W: int, focal, ext_scale: float, center=None,
                   extend_mm: float = 250., top_expand_ratio: float = 1.60, bot_expand_ratio: float = 1.10):
        """Device bounding cylinders [F,5] f32 and integer boxes [F,4] i32 (tl_x, tl_y, br_x, br_y) of device
        key points [F,24,3] (pg_pose_boxes: kp_to_valid_rays' cull without the host round trip).  `c2ws`:
        one camera [4,4] or one per pose [F,4,4] (host); the extrinsic is inverted on the host in float32
        exactly as the reference does (nerf_c2w_to_extrinsic)."""
        from .skeleton import nerf_c2w_to_extrinsic
        kp = _dev_f32(kps.reshape(-1, 24, 3), self.device)
        F = kp.shape[0]
        c2w = np.asarray(torch.as_tensor(c2ws).detach().cpu(), dtype=np.float32).reshape(-1, 4, 4)
        w2c = np.stack([nerf_c2w_to_extrinsic(c) for c in c2w]).astype(np.float64)      # float32 inverse, widened
        if w2c.shape[0] not in (1, F):
            raise ValueError(f"{w2c.shape[0]} cameras for {F} poses")
        phi = np.linspace(0., 2 * np.pi, 50)
        ring = np.ascontiguousarray(np.stack([np.cos(phi), np.sin(phi)], -1))
        f = np.asarray(torch.as_tensor(focal).detach().cpu(), dtype=np.float64).reshape(-1)
        fx, fy = (float(np.float32(f[0])), float(np.float32(f[0]))) if f.size < 2 else (float(np.float32(f[0])), float(np.float32(f[1])))
        offx, offy = (int(W * .5), int(H * .5)) if center is None else (int(center[0]), int(center[1]))
        ext = extend_mm * ext_scale
        d_w2c = torch.tensor(w2c, dtype=torch.float64, device=self.device)
        d_ring = torch.tensor(ring, dtype=torch.float64, device=self.device)
        cyls = torch.empty(F, 5, device=self.device)
        boxes = torch.empty(F, 4, device=self.device, dtype=torch.int32)
        self._check(self.lib.pg_pose_boxes(self.handle, self._stream(), F, _ptr(kp), _ptr(d_w2c), 16 if w2c.shape[0] > 1 else 0,
                                           _ptr(d_ring), float(ext), float(ext * top_expand_ratio), float(ext * bot_expand_ratio),
                                           fx, fy, int(H), int(W), offx, offy, _ptr(cyls), _ptr(boxes)))
        return cyls, boxes

    # -- stage entry points (tests / profiling) ---------------------------------------
    def stage_sample_coarse(self, ray_batch, cyls, n_samples, lindisp=False):
        rb = _dev_f32(ray_batch, self.device)
        n = rb.shape[0]
        cy, cs = self._cyl_args(cyls, n)
        nf = torch.empty(n, 2, device=self.device)
        z = torch.empty(n, n_samples, device=self.device)
        self._check(self.lib.pg_stage_sample_coarse(self.handle, self._stream(), n, _ptr(rb), _ptr(cy), cs,
                                                    int(n_samples), _ffi.PG_FLAG_LINDISP if lindisp else 0,
                                                    _ptr(nf), _ptr(z)))
        return nf, z

    def stage_eval(self, which, ray_batch, z, skts, cams=None, want_dbg=False, dbg_stage=0):
        rb = _dev_f32(ray_batch, self.device)
        zz = _dev_f32(z, self.device)
        n, S = zz.shape
        sk, ps = self._pose_args(skts, n)
        cam = None if cams is None else _dev_f32(cams.reshape(-1), self.device)
        raw = torch.empty(n, S, 4, device=self.device)
        dbg = torch.zeros(n * S, 256, device=self.device) if want_dbg else None
        self._check(self.lib.pg_stage_eval(self.handle, self._stream(), int(which), n, S, _ptr(rb), _ptr(zz),
                                           _ptr(sk), ps, _ptr(cam), _ptr(raw), _ptr(dbg), int(dbg_stage)))
        return (raw, dbg) if want_dbg else raw

    def limb_skip_stats(self, which, ray_batch, z, skts, cams=None):
        """Measurement aid (pg_stage_eval, dbg_stage 97): what the limb masks of the fused kernel leave out on this launch --
        the kernel itself counts.  Returns the fraction of (pass, limb) pairs left out of whole passes and the fraction of
        (wave or column tile, limb) pairs left out at the finer level (which includes the former)."""
        rb = _dev_f32(ray_batch, self.device)
        zz = _dev_f32(z, self.device)
        n, S = zz.shape
        sk, ps = self._pose_args(skts, n)
        cam = None if cams is None else _dev_f32(cams.reshape(-1), self.device)
        raw = torch.empty(n, S, 4, device=self.device)
        cnt = torch.zeros(64, device=self.device, dtype=torch.int32)
        self._check(self.lib.pg_stage_eval(self.handle, self._stream(), int(which), n, S, _ptr(rb), _ptr(zz),
                                           _ptr(sk), ps, _ptr(cam), _ptr(raw), cnt.data_ptr(), 97))
        passes, per_pass, fine = (int(v) for v in cnt[:3].cpu())
        per_fine = 48                                   # 8 waves / column tiles x 6 limbs per pass in both kernels
        return {"passes": passes, "limbs_left_out_of_whole_passes_frac": per_pass / max(6 * passes, 1),
                "limbs_left_out_frac": fine / max(per_fine * passes, 1)}

    def stage_composite(self, ray_batch, z, raw, n_importance=0):
        rb = _dev_f32(ray_batch, self.device)
        zz = _dev_f32(z, self.device)
        rw = _dev_f32(raw, self.device)
        n, S = zz.shape
        new = lambda *s: torch.empty(*s, device=self.device, dtype=torch.float32)
        o = {"rgb_map": new(n, 3), "disp_map": new(n), "acc_map": new(n), "alpha": new(n, S),
             "weights": new(n, S)}
        zf = new(n, S + n_importance) if n_importance > 0 else None
        self._check(self.lib.pg_stage_composite(self.handle, self._stream(), n, S, _ptr(rb), _ptr(zz), _ptr(rw),
                                                _ptr(o["rgb_map"]), _ptr(o["disp_map"]), _ptr(o["acc_map"]),
                                                _ptr(o["alpha"]), _ptr(o["weights"]), int(n_importance), _ptr(zf)))
        if zf is not None:
            o["z_fine"] = zf
        return o


class HipRayCaster:
    """Call-compatible stand-in for `RayCaster` / `nn.DataParallel(RayCaster)`.

    Reference call site: ``ray_caster(rays_flat[i:i+chunk].to('cuda'), **batch_kwargs)``
    (core/trainer.py:74) with the kwargs of `render_kwargs_test`
    (core/raycasters.py:156-178) plus the per-ray pose tensors.
    """

    def __init__(self, cfg: RenderConfig, device="cuda:0", precision=PREC_BF16, devices=None):
        """`devices=[0, 1, ...]`: all GPUs of this process behind one caster, the replacement of
        `nn.DataParallel(RayCaster)` (core/raycasters.py:157): `render_path` then spreads the frames
        (or, with fewer frames than GPUs, the frames' ray chunks) over them inside one call."""
        self.cfg = cfg
        self.renderer = HipRenderer(cfg, device, precision, devices=devices)
        self.training = False

    # ---- construction helpers -------------------------------------------------------
    @classmethod
    def from_weights(cls, cfg, w_coarse, w_fine, tau_v, tau_d, device="cuda:0", precision=PREC_BF16, devices=None):
        rc = cls(cfg, device, precision, devices=devices)
        rc.renderer.load_network(0, w_coarse)
        if w_fine is not None:
            rc.renderer.load_network(1, w_fine)
        rc.renderer.set_embedder(0, tau_v)
        rc.renderer.set_embedder(1, tau_d)
        return rc

    # ---- density queries (core/raycasters.py:579-646) ----------------------------------------
    def render_pts_density(self, pts, kps, skts, bones=None, render_kwargs=None, subject_idxs=None,
                           netchunk=1024 * 64, network=None, color=False, v=None):
        """Raw density [..., 1] at points `pts` [n,1,3] (or [n,3]) for one pose; fine net unless
        `network` is 0/1.  `kps`, `bones`, `netchunk` are accepted for call compatibility."""
        if color or v is not None or subject_idxs is not None:
            raise NotImplementedError("render_pts_density: color / precomputed v / subject_idxs are not supported")
        which = network if network in (0, 1) else None
        return self.renderer.query_density(torch.as_tensor(pts), skts, which)

    def render_mesh_density(self, kps, skts, bones=None, subject_idxs=None, radius=1.0, res=64,
                            render_kwargs=None, netchunk=1024 * 64, v=None):
        """Raw density on the (res+1)^3 grid around the root joint, as the reference lays it out."""
        if v is not None or subject_idxs is not None:
            raise NotImplementedError("render_mesh_density: precomputed v / subject_idxs are not supported")
        return self.renderer.mesh_density(kps, skts, radius=radius, res=res)

    # ---- nn.Module-like surface the reference touches -------------------------------
    @property
    def module(self):            # trainer.py:267,272,506 reach through DataParallel
        return self

    def to(self, *a, **k):       # trainer.py:73 `ray_caster.to('cuda')`
        return self

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        self.training = bool(mode)
        return self

    def parameters(self):
        self.renderer._refresh_state()
        for sd in self.renderer._state.values():
            for v in sd.values():
                yield v

    def state_dict(self):
        self.renderer._refresh_state()
        sd = {k: dict(v) for k, v in self.renderer._state.items()}
        sd.setdefault("embedbones_state_dict", {})
        return sd

    def load_state_dict(self, ckpt, strict=True):
        r = self.renderer
        if "network_fn_state_dict" in ckpt:
            r.load_network(0, ckpt["network_fn_state_dict"])
        elif strict:
            raise KeyError("network_fn_state_dict")
        if ckpt.get("network_fine_state_dict") is not None:
            r.load_network(1, ckpt["network_fine_state_dict"])
        for which, key in ((0, "embed_state_dict"), (1, "embeddirs_state_dict")):
            e = ckpt.get(key)
            if e is not None and "tau" in e:
                r.set_embedder(which, float(e["tau"]), e.get("cutoff_dist"))
            elif strict:
                raise KeyError(key)

    # ---- the call -------------------------------------------------------------------
    def __call__(self, *args, fwd_type="", **kwargs):
        # the reference dispatches on fwd_type before anything else (core/raycasters.py:349-359)
        if fwd_type == "density":
            return self.render_pts_density(*args, **kwargs)
        if fwd_type == "mesh":
            return self.render_mesh_density(*args, **kwargs)
        return self.forward(*args, fwd_type=fwd_type, **kwargs)

    def forward(self, ray_batch, N_samples=None, kp_batch=None, skts=None, cyls=None, bones=None,
                cams=None, subject_idxs=None, retraw=False, lindisp=False, perturb=0., N_importance=0,
                network_fine=None, raw_noise_std=0., ray_noise_std=0., verbose=False, ext_scale=0.001,
                pytest=False, preproc_kwargs=None, nerf_type="nerf", fwd_type="", use_viewdirs=True,
                want_alpha=True, extras=False, draws=None, **unused):
        """`RayCaster.forward` (core/raycasters.py:349-474), forward values only (no autograd graph:
        SURVEY.md 8 scopes the renderer, not NeRF training).  perturb / raw_noise_std /
        ray_noise_std behave as in render_kwargs_train (raycasters.py:156-165): the random numbers
        come from torch's generator on the caster's device, from numpy after np.random.seed(0) when
        pytest=True (the reference's deterministic test mode; it has no override for the position
        noise, which stays a torch draw), or from `draws` when the caller supplies them."""
        if fwd_type:
            raise NotImplementedError(f"fwd_type={fwd_type!r} is not on the HIP path ('density' and 'mesh' are)")
        if subject_idxs is not None:
            raise NotImplementedError("subject_idxs (multi-subject nets) are not supported")
        if skts is None or cyls is None:
            raise ValueError("skts and cyls are required (A-NeRF bone-relative rendering)")
        # refuse, don't render differently: a caller configured for anything the fused kernels do not compute
        if nerf_type != "nerf":
            raise NotImplementedError(f"nerf_type={nerf_type!r}: only 'nerf' is on the HIP path")
        if not use_viewdirs:
            raise NotImplementedError("use_viewdirs=False: the HIP kernels always evaluate the view branch (nerf.py:112-121)")
        if network_fine is not None:
            raise NotImplementedError("network_fine: the caster renders with the nets it was loaded with (load_state_dict)")
        if unused:
            raise TypeError(f"HipRayCaster.forward: unexpected keyword arguments {sorted(unused)}")
        self._check_preproc_kwargs(preproc_kwargs)
        # One call = one nanmean group, like get_near_far_in_cylinder on the reference's ray_batch
        # (ray_utils.py:292-344): only batchify_rays / render_path split a frame into `chunk` groups.
        # The group size is a property of THIS call: the renderer's own setting (what later direct
        # render_rays / render_frame calls see) is put back afterwards.
        keep = self.renderer._chunk
        if not getattr(self, "_grouped_call", False):
            self.renderer.set_chunk(max(int(ray_batch.shape[0]), 1))
        try:
            if draws is None and (perturb or raw_noise_std or ray_noise_std):
                S = self.cfg.n_samples if N_samples is None else int(N_samples)
                draws = self.training_draws(int(ray_batch.shape[0]), S, int(N_importance or 0), perturb, raw_noise_std,
                                            ray_noise_std, pytest=pytest)
            return self.renderer.render_rays(ray_batch, skts, cyls, cams=cams, n_samples=N_samples,
                                             n_importance=N_importance, lindisp=bool(lindisp),
                                             want_alpha=want_alpha, extras=extras, draws=draws)
        finally:
            self.renderer.set_chunk(keep)

    # the reference's preproc_kwargs (core/raycasters.py:140-152): the encoder objects and the density function
    # `create_raycaster` picked.  The kernels implement exactly one choice of each.
    _ENCODERS = {"pts_tr_fn": "WorldToLocalEncoder", "kp_input_fn": "RelDistEncoder",
                 "view_input_fn": "VecNormEncoder", "bone_input_fn": "VecNormEncoder"}

    def _check_preproc_kwargs(self, pk):
        if not pk:
            return
        cfg = self.cfg
        unknown = set(pk) - set(self._ENCODERS) - {"density_scale", "density_fn"}
        if unknown:
            raise NotImplementedError(f"preproc_kwargs {sorted(unknown)} are not supported by the HIP renderer")
        for key, want in self._ENCODERS.items():
            fn = pk.get(key)
            if fn is not None and type(fn).__name__ != want:
                raise NotImplementedError(f"preproc_kwargs[{key!r}] is a {type(fn).__name__}; the HIP kernels compute {want} only")
        ds = pk.get("density_scale")
        if ds is not None and float(ds) != float(cfg.density_scale):
            raise ValueError(f"preproc_kwargs['density_scale']={float(ds)} but the caster was created with "
                             f"density_scale={cfg.density_scale} (RenderConfig)")
        dfn = pk.get("density_fn")
        if dfn is not None:
            # any callable may arrive here (get_density_fn returns F.relu or a lambda): compare it with the
            # configured activation on probe values instead of guessing from its identity
            x = torch.tensor([-30., -2., -0.25, 0., 0.5, 1., 3., 25.])
            want = torch.relu(x) if cfg.density_type == "relu" else torch.nn.functional.softplus(x - cfg.softplus_shift, beta=1)
            got = torch.as_tensor(dfn(x)).detach().float().cpu()
            if got.shape != want.shape or not torch.allclose(got, want, rtol=1e-6, atol=1e-7):
                raise NotImplementedError(
                    f"preproc_kwargs['density_fn'] is not the configured density activation (density_type={cfg.density_type!r}"
                    + (f", softplus_shift={cfg.softplus_shift}" if cfg.density_type != "relu" else "")
                    + "): create the caster with the matching RenderConfig")

    def training_draws(self, n, S, N, perturb=0., raw_noise_std=0., ray_noise_std=0., pytest=False):
        return make_training_draws(n, S, N, perturb, raw_noise_std, ray_noise_std, pytest=pytest,
                                   density_scale=self.cfg.density_scale, device=self.renderer.device)


def make_training_draws(n, S, N, perturb=0., raw_noise_std=0., ray_noise_std=0., pytest=False,
                        density_scale=1., device="cpu"):
    """The random numbers one training-mode render_rays call consumes, in the reference's
    places: t_rand (ray_utils.py:238-244), u_rand (ray_utils.py:166-180), the density noise
    randn * raw_noise_std * B of both passes (nerf.py:174-182; pytest: rand * raw_noise_std,
    numpy, no B) and the position noise randn * ray_noise_std (raycasters.py:660-661, 673-674)."""
    dev = device
    d = {}
    f32 = lambda a: torch.Tensor(a).to(dev)
    if perturb and perturb > 0.:
        if pytest:
            np.random.seed(0); d["t_rand"] = f32(np.random.rand(n, S))
            if N > 0:
                np.random.seed(0); d["u_rand"] = f32(np.random.rand(n, N))
        else:
            d["t_rand"] = torch.rand(n, S, device=dev)
            if N > 0:
                d["u_rand"] = torch.rand(n, N, device=dev)
    if raw_noise_std and raw_noise_std > 0.:
        B = float(density_scale)
        for key, m in (("noise0", S),) + ((("noise1", S + N),) if N > 0 else ()):
            if pytest:
                np.random.seed(0); d[key] = f32(np.random.rand(n, m) * raw_noise_std)
            else:
                d[key] = torch.randn(n, m, device=dev) * (raw_noise_std * B)
    if ray_noise_std and ray_noise_std > 0.:
        d["ray_noise"] = torch.randn(n, S + N, 3, device=dev) * ray_noise_std
    return d


def create_raycaster(cfg: RenderConfig, ckpt=None, device="cuda:0", precision=PREC_BF16, devices=None):
    """Counterpart of `create_raycaster` (core/raycasters.py:17-184) for rendering:
    returns `render_kwargs_test` with the HIP caster under 'ray_caster'.  `devices` = the GPUs the
    reference would hand to nn.DataParallel (raycasters.py:157)."""
    caster = HipRayCaster(cfg, device, precision, devices=devices)
    if ckpt is not None:
        caster.load_state_dict(ckpt)
    caster.eval()
    return {"ray_caster": caster, "perturb": False, "N_importance": cfg.n_importance,
            "N_samples": cfg.n_samples, "use_viewdirs": True, "raw_noise_std": 0., "ray_noise_std": 0.,
            "ext_scale": cfg.ext_scale, "preproc_kwargs": {"density_scale": cfg.density_scale}, "lindisp": cfg.lindisp,
            "nerf_type": "nerf"}


def find_checkpoint(basedir: str, expname: str, ft_path: Optional[str] = None, no_reload: bool = False) -> Optional[str]:
    """The checkpoint `create_raycaster` would reload (core/raycasters.py:124-141): `ft_path` if given (and not
    the string 'None'), otherwise the LAST entry, in sorted name order, of basedir/expname whose name contains
    'tar' and not 'pose'; None when there is none or `no_reload` is set."""
    import os
    if ft_path is not None and ft_path != "None":
        ckpts = [ft_path]
    else:
        d = os.path.join(basedir, expname)
        ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if "tar" in f and "pose" not in f]
    return ckpts[-1] if ckpts and not no_reload else None


_RAYCASTER_CACHE: Dict[tuple, dict] = {}


def load_raycaster(ckpt_path: str, cfg: RenderConfig, device="cuda:0", precision=PREC_BF16, devices=None):
    """`create_raycaster` on an A-NeRF checkpoint file (`.tar`, the reference's five state dicts,
    core/raycasters.py:752-766), memoised on (path, mtime, size, config, device, precision).

    The reference's `run_render` reloads and re-wraps the checkpoint on every call of the GAN
    loop (run_gan.py:135-165, 2290-2330); here a repeated call returns the caster whose packed
    weights are already resident on the device."""
    import os
    if isinstance(precision, str):
        precision = PREC_BY_NAME[precision]
    st = os.stat(ckpt_path)
    key = (os.path.abspath(ckpt_path), st.st_mtime_ns, st.st_size, repr(cfg), str(device), int(precision),
           None if devices is None else tuple(devices))
    kw = _RAYCASTER_CACHE.get(key)
    if kw is None:
        ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
        kw = create_raycaster(cfg, ckpt, device=device, precision=precision, devices=devices)
        _RAYCASTER_CACHE[key] = kw
    return dict(kw)
