#!/usr/bin/env python3
"""Capture golden vectors from the REAL reference implementation.

Runs only where the read-only reference checkout exists (the build container);
the GPU box never sees the reference.  The reference is imported in-process
with stub modules for its unused third-party imports and a 'cuda'->'cpu'
device shim (the reference hard-codes .to('cuda')); nothing of the reference is
written into this repository -- only inputs (seeds / small tensors) and the
numeric outputs, as .npz fixtures under tests/golden/.

Usage:  python tools/gen_golden.py [--ref /root/reference] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile
import types
from argparse import Namespace

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True


class _StubModule(types.ModuleType):
    """Placeholder for third-party modules the reference imports but the render
    path never calls (cv2, pytorch3d, h5py, ...): any attribute is a dummy."""

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _StubModule(f"{self.__name__}.{item}")

    def __call__(self, *a, **k):
        raise RuntimeError(f"stubbed module attribute {self.__name__} was called")


def _install_shims(ref_root: str):
    import torch
    import torch.nn as nn

    if not os.path.isdir(os.path.join(ref_root, "core")):
        sys.exit(f"gen_golden: reference checkout not found at {ref_root}; "
                 "fixtures can only be regenerated in the build container")
    for p in (ref_root, os.path.join(ref_root, "smplx"), os.path.join(ref_root, "pytorch-msssim")):
        sys.path.insert(0, p)
    for name in ("cv2", "pytorch3d", "pytorch3d.transforms",
                 "pytorch3d.transforms.rotation_conversions", "h5py", "imageio",
                 "configargparse", "tensorboard", "tensorboard.backend",
                 "tensorboard.backend.event_processing",
                 "tensorboard.backend.event_processing.event_accumulator",
                 "torch.utils.tensorboard", "deepdish", "skimage", "skimage.transform",
                 "skimage.metrics", "lpips"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = _StubModule(name)
                m.__path__ = []          # behave like a package
                sys.modules[name] = m
    sys.modules["torch.utils.tensorboard"].SummaryWriter = object
    sys.modules["tensorboard.backend.event_processing.event_accumulator"].EventAccumulator = object

    def _cpu_dev(a):
        if isinstance(a, str) and a.startswith("cuda"):
            return "cpu"
        if isinstance(a, torch.device) and a.type == "cuda":
            return torch.device("cpu")
        return a

    t_to = torch.Tensor.to
    m_to = nn.Module.to

    def tensor_to(self, *args, **kw):
        args = tuple(_cpu_dev(a) for a in args)
        kw = {k: _cpu_dev(v) for k, v in kw.items()}
        return t_to(self, *args, **kw)

    def module_to(self, *args, **kw):
        args = tuple(_cpu_dev(a) for a in args)
        kw = {k: _cpu_dev(v) for k, v in kw.items()}
        return m_to(self, *args, **kw)

    torch.Tensor.to = tensor_to
    nn.Module.to = module_to


def _nerf_args(cfg, workdir) -> Namespace:
    """argparse.Namespace with the renderer-relevant flag values."""
    return Namespace(
        n_framecodes=cfg.n_framecodes if cfg.framecode_ch else None,
        use_cutoff=True, normalize_cutoff=False, cutoff_mm=cfg.cutoff_mm,
        ext_scale=cfg.ext_scale, cutoff_inputs=True, opt_cutoff=False, freq_schedule=False,
        init_freq=0., cut_to_dist=False, cutoff_shift=False, multires=cfg.multires, i_embed=0,
        cutoff_bones=False, multires_bones=cfg.multires_bones, use_viewdirs=True,
        cutoff_viewdir=True, multires_views=cfg.multires_views, N_importance=cfg.n_importance,
        netdepth=cfg.net_depth, netwidth=cfg.net_width, opt_framecode=cfg.framecode_ch > 0,
        framecode_size=cfg.framecode_ch if cfg.framecode_ch else 16, density_scale=cfg.density_scale,
        single_net=False, lrate=5e-4, basedir=workdir, expname="golden", ft_path=None,
        no_reload=True, finetune=False, perturb=0., N_samples=cfg.n_samples, raw_noise_std=0.,
        ray_noise_std=0., lindisp=False, nerf_type="nerf", debug=False, density_type=cfg.density_type,
        softplus_shift=cfg.softplus_shift, pts_tr_type="local", kp_dist_type="reldist", view_type="relray",
        bone_type="reldir", fix_layer=0, weight_decay=None, chunk=cfg.chunk)


def _build_reference_caster(cfg, seed, workdir):
    import torch
    from core.raycasters import create_raycaster
    from core.utils.skeleton_utils import SMPLSkeleton, get_per_joint_coords
    from posegen_amd import synthetic as syn
    from posegen_amd.skeleton import smpl_rest_pose, SURREAL_REST_SCALE

    os.makedirs(os.path.join(workdir, "golden"), exist_ok=True)
    rest = smpl_rest_pose * SURREAL_REST_SCALE
    attrs = {"skel_type": SMPLSkeleton, "near": 0., "far": 1.,
             "n_views": max(cfg.n_framecodes, 1),
             "joint_coords": get_per_joint_coords(rest)[None]}
    _, kw_test, *_ = create_raycaster(_nerf_args(cfg, workdir), attrs)
    caster = kw_test["ray_caster"]
    wc, wf, tau_v, tau_d = syn.make_model(cfg, seed)
    sd = {"network_fn_state_dict": {k: torch.tensor(v) for k, v in wc.items()},
          "network_fine_state_dict": {k: torch.tensor(v) for k, v in wf.items()}}
    caster.network.load_state_dict(sd["network_fn_state_dict"])
    if caster.network_fine is not None:
        caster.network_fine.load_state_dict(sd["network_fine_state_dict"])
    caster.embed_fn.tau = torch.tensor(tau_v)
    caster.embeddirs_fn.tau = torch.tensor(tau_d)
    caster.eval()
    return caster, kw_test, (wc, wf, tau_v, tau_d)


def _weights_digest(w):
    import hashlib
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k]).tobytes())
    return h.hexdigest()


def gen_kinematics(out):
    """a-18: bones -> l2ws / kps / skts via the reference's get_smpl_l2ws."""
    from core.utils.skeleton_utils import get_smpl_l2ws
    from posegen_amd import synthetic as syn
    from posegen_amd.skeleton import smpl_rest_pose, SURREAL_REST_SCALE
    rest = smpl_rest_pose * SURREAL_REST_SCALE
    bones = syn.make_bones(4, seed=7)
    bones[3, 5] = 0.0                               # exercise the small-angle branch
    bones[3, 6] = 1e-5
    l2ws = np.array([get_smpl_l2ws(b, rest, 1.0) for b in bones])
    np.savez_compressed(os.path.join(out, "kinematics.npz"), bones=bones, rest_pose=rest,
                        l2ws=l2ws, kps=l2ws[..., :3, -1], skts=np.linalg.inv(l2ws))


def gen_valid_rays(out):
    """a-2: cylinder, box, pixel ids and rays of the bbox cull."""
    import torch
    from core.utils.ray_utils import kp_to_valid_rays
    from posegen_amd import synthetic as syn
    H = W = 96
    _, kps, _ = syn.make_pose(3, seed=11)
    c2ws, focals = syn.make_camera(3, H, W)
    rays, vids, cyls, boxes = kp_to_valid_rays(torch.tensor(c2ws), H, W, focals,
                                               kps=torch.tensor(kps), ext_scale=0.001)
    d = {"H": H, "W": W, "kps": kps, "c2ws": c2ws, "focals": focals,
         "cyls": cyls.numpy(), "boxes": np.array([[b[0], b[1]] for b in boxes])}
    for i, (r, v) in enumerate(zip(rays, vids)):
        d[f"n_valid_{i}"] = len(v)
        d[f"vid_head_{i}"] = v[:8].numpy()
        d[f"vid_tail_{i}"] = v[-8:].numpy()
        d[f"rays_o_head_{i}"] = r[0][:8].numpy()
        d[f"rays_d_head_{i}"] = r[1][:8].numpy()
        d[f"rays_d_tail_{i}"] = r[1][-8:].numpy()
    np.savez_compressed(os.path.join(out, "valid_rays.npz"), **d)


def _stagewise(caster, kw, batch, kp_b, skt_b, cyl_b, bones_b, cams_b, n_samples, n_importance):
    """Run the reference's own stage functions in render_rays order and keep
    every intermediate (raycasters.py:413-474)."""
    import torch
    from core.utils.ray_utils import get_near_far_in_cylinder
    pk = kw["preproc_kwargs"]
    n = batch.shape[0]
    o, d = batch[:, 0:3], batch[:, 3:6]
    bounds = torch.reshape(batch[..., 6:8], [-1, 1, 2])
    near, far = bounds[..., 0], bounds[..., 1]
    near, far = get_near_far_in_cylinder(o, d, cyl_b, near=near, far=far)
    pts, z = caster.sample_pts(o, d, near, far, n, n_samples, 0., False)
    jc = caster.get_subject_joint_coords(None, pts.device)
    enc = caster.encode_inputs(pts, [o[:, None, :], d[:, None, :]], kp_b, skt_b, bones_b,
                               cam_idxs=cams_b, subject_idxs=None, joint_coords=jc,
                               network=caster.network, **pk)
    x = torch.cat([enc["v"], enc["r"], enc["d"]], -1).clone()
    raw = caster.run_network(enc, caster.network)
    rd = caster.network.raw2outputs(raw, z, d, 0., B=pk["density_scale"], act_fn=pk["density_fn"])
    res = {"near": near, "far": far, "z_coarse": z, "x_coarse": x, "raw_coarse": raw,
           "weights_coarse": rd["weights"], "alpha0": rd["alpha"], "rgb0": rd["rgb_map"],
           "disp0": rd["disp_map"], "acc0": rd["acc_map"]}
    if n_importance > 0:
        pts_is, z_all, z_new, order = caster.sample_pts_is(o, d, z, rd["weights"], n_importance,
                                                           det=True, is_only=False)
        enc_is = caster.encode_inputs(pts_is, [o[:, None, :], d[:, None, :]], kp_b, skt_b, bones_b,
                                      cam_idxs=cams_b, subject_idxs=None, joint_coords=jc,
                                      network=caster.network_fine, **pk)
        merged = caster._merge_encodings(enc, enc_is, order, n, n_samples + n_importance)
        raw_f = caster.run_network(merged, caster.network_fine)
        rf = caster.network_fine.raw2outputs(raw_f, z_all, d, 0., B=pk["density_scale"],
                                             act_fn=pk["density_fn"])
        res.update({"z_fine": z_all, "z_new": z_new, "order": order, "raw_fine": raw_f,
                    "alpha": rf["alpha"], "rgb_map": rf["rgb_map"], "disp_map": rf["disp_map"],
                    "acc_map": rf["acc_map"], "weights_fine": rf["weights"]})
    else:
        res.update({"alpha": rd["alpha"], "rgb_map": rd["rgb_map"], "disp_map": rd["disp_map"],
                    "acc_map": rd["acc_map"]})
    return res


def gen_render_rays(out, name, cfg, *, n_rays, H, all_hit, seed_model=0, seed_pose=1,
                    use_cams=False, keep_x=16):
    """a-5..a-16: one `RayCaster.__call__` on a strided subset of a culled frame."""
    import torch
    from core.utils.ray_utils import kp_to_valid_rays
    from posegen_amd import synthetic as syn
    with tempfile.TemporaryDirectory() as wd:
        caster, kw, (wc, wf, tau_v, tau_d) = _build_reference_caster(cfg, seed_model, wd)
    W = H
    bones, kps, skts = syn.make_pose(1, seed_pose)
    c2ws, focals = syn.make_camera(1, H, W)
    rays, vids, cyls, boxes = kp_to_valid_rays(torch.tensor(c2ws), H, W, focals,
                                               kps=torch.tensor(kps), ext_scale=cfg.ext_scale)
    ro, rd = rays[0]
    sel = np.unique(np.linspace(0, ro.shape[0] - 1, n_rays).round().astype(np.int64))
    ro, rd = ro[sel].float(), rd[sel].float()
    n = ro.shape[0]
    if all_hit:
        cyls = cyls.clone()
        cyls[:, 2] = 2.5
    vd = rd / torch.norm(rd, dim=-1, keepdim=True)
    ones = torch.ones(n, 1)
    batch = torch.cat([ro, rd, 0. * ones, 1. * ones, vd], -1)
    kp_b = torch.tensor(kps).expand(n, -1, -1)
    skt_b = torch.tensor(skts).expand(n, -1, -1, -1)
    cyl_b = cyls.expand(n, -1)
    bones_b = torch.tensor(bones).expand(n, -1, -1)
    cams_b = None
    cams_np = None
    if use_cams:
        cams_np = (np.arange(n) % cfg.n_framecodes).astype(np.float32)
        cams_np[: n // 4] = 3.0
        cams_b = torch.tensor(cams_np)
    with torch.no_grad():
        st = _stagewise(caster, kw, batch, kp_b, skt_b, cyl_b, bones_b, cams_b,
                        cfg.n_samples, cfg.n_importance)
        call_kw = {k: v for k, v in kw.items() if k != "ray_caster"}
        call_kw.pop("use_viewdirs", None)
        full = caster(batch, kp_batch=kp_b, skts=skt_b, cyls=cyl_b, bones=bones_b, cams=cams_b,
                      subject_idxs=None, **call_kw)
    for k in ("rgb_map", "disp_map", "acc_map", "alpha"):
        assert torch.equal(full[k], st[k]) or torch.allclose(full[k], st[k], atol=0, rtol=0, equal_nan=True), k
    # the 1080(+1)-vector is kept for a few points only (fixture size)
    pick_r = np.unique(np.linspace(0, n - 1, keep_x).round().astype(np.int64))
    pick_s = np.array([0, cfg.n_samples // 3, cfg.n_samples // 2, cfg.n_samples - 1])
    x_pick = st["x_coarse"][pick_r][:, pick_s].numpy()
    d = {"ray_batch": batch.numpy(), "kps": kps, "skts": skts, "bones": bones, "cyl": cyls.numpy(),
         "tau_v": tau_v, "tau_d": tau_d, "seed_model": seed_model, "n_samples": cfg.n_samples,
         "n_importance": cfg.n_importance, "framecode_ch": cfg.framecode_ch,
         "n_framecodes": cfg.n_framecodes, "digest_coarse": _weights_digest(wc),
         "digest_fine": _weights_digest(wf), "x_pick": x_pick, "x_pick_rays": pick_r,
         "x_pick_samples": pick_s}
    if cfg.density_type != "relu":
        d["density_softplus"] = int(cfg.density_type == "softplus")
        d["softplus_shift"] = float(cfg.softplus_shift)
    if cams_np is not None:
        d["cams"] = cams_np
    for k, v in st.items():
        if k == "x_coarse":
            continue
        d[k] = v.numpy()
    d["n_rays"] = n
    np.savez_compressed(os.path.join(out, f"{name}.npz"), **d)
    acc = st["acc_map"].numpy()
    print(f"[{name}] rays={n} acc in [{acc.min():.3f},{acc.max():.3f}] "
          f"mid-fraction={(np.logical_and(acc > 0.05, acc < 0.95)).mean():.2f}")


def gen_render_rays_train(out, name, cfg, *, n_rays, H, seed_model=0, seed_pose=1, perturb=1.,
                          raw_noise_std=1., ray_noise_std=0.005):
    """Training-mode `RayCaster.__call__` (render_kwargs_train: perturb, raw_noise_std,
    ray_noise_std) in the reference's own deterministic test mode, pytest=True: the stratified
    jitter, the inverse-cdf positions and the density noise are numpy draws after
    np.random.seed(0) (ray_utils.py:171-180, 241-244; nerf.py:179-182).  The position noise has
    no such override (raycasters.py:660-661, 673-674): the two torch.randn_like results are
    recorded as the reference makes them.  The fixture holds the draws (inputs) and the
    outputs of the call."""
    import torch
    from core.utils.ray_utils import kp_to_valid_rays
    from posegen_amd import synthetic as syn
    with tempfile.TemporaryDirectory() as wd:
        caster, kw, (wc, wf, tau_v, tau_d) = _build_reference_caster(cfg, seed_model, wd)
    W = H
    bones, kps, skts = syn.make_pose(1, seed_pose)
    c2ws, focals = syn.make_camera(1, H, W)
    rays, vids, cyls, boxes = kp_to_valid_rays(torch.tensor(c2ws), H, W, focals,
                                               kps=torch.tensor(kps), ext_scale=cfg.ext_scale)
    ro, rd = rays[0]
    sel = np.unique(np.linspace(0, ro.shape[0] - 1, n_rays).round().astype(np.int64))
    ro, rd = ro[sel].float(), rd[sel].float()
    n = ro.shape[0]
    vd = rd / torch.norm(rd, dim=-1, keepdim=True)
    ones = torch.ones(n, 1)
    batch = torch.cat([ro, rd, 0. * ones, 1. * ones, vd], -1)
    kp_b = torch.tensor(kps).expand(n, -1, -1)
    skt_b = torch.tensor(skts).expand(n, -1, -1, -1)
    cyl_b = cyls.expand(n, -1)
    bones_b = torch.tensor(bones).expand(n, -1, -1)
    S, N = cfg.n_samples, cfg.n_importance
    recorded = []
    randn_like = torch.randn_like

    def recording_randn_like(t, *a, **k):
        r = randn_like(t, *a, **k)
        recorded.append(r.clone())
        return r

    call_kw = {k: v for k, v in kw.items() if k != "ray_caster"}
    call_kw.pop("use_viewdirs", None)
    call_kw.update(perturb=perturb, raw_noise_std=raw_noise_std, ray_noise_std=ray_noise_std, pytest=True)
    caster.train()
    torch.manual_seed(1234)
    torch.randn_like = recording_randn_like
    try:
        with torch.no_grad():
            full = caster(batch, kp_batch=kp_b, skts=skt_b, cyls=cyl_b, bones=bones_b, cams=None,
                          subject_idxs=None, **call_kw)
    finally:
        torch.randn_like = randn_like
        caster.eval()
    d = {"ray_batch": batch.numpy(), "kps": kps, "skts": skts, "bones": bones, "cyl": cyls.numpy(),
         "tau_v": tau_v, "tau_d": tau_d, "seed_model": seed_model, "n_samples": S, "n_importance": N,
         "framecode_ch": cfg.framecode_ch, "n_framecodes": cfg.n_framecodes,
         "digest_coarse": _weights_digest(wc), "digest_fine": _weights_digest(wf),
         "perturb": perturb, "raw_noise_std": raw_noise_std, "ray_noise_std": ray_noise_std, "n_rays": n}
    # the pytest=True draws, exactly as the reference forms them (float64 numpy -> torch.Tensor = float32)
    f32 = lambda a: torch.Tensor(a).numpy()
    if perturb > 0:
        np.random.seed(0); d["t_rand"] = f32(np.random.rand(n, S))
        if N > 0:
            np.random.seed(0); d["u_rand"] = f32(np.random.rand(n, N))
    if raw_noise_std > 0:
        np.random.seed(0); d["noise0"] = f32(np.random.rand(n, S) * raw_noise_std)
        if N > 0:
            np.random.seed(0); d["noise1"] = f32(np.random.rand(n, S + N) * raw_noise_std)
    if ray_noise_std > 0:
        assert len(recorded) == (2 if N > 0 else 1) and recorded[0].shape == (n, S, 3), [r.shape for r in recorded]
        d["ray_noise"] = torch.cat([r * ray_noise_std for r in recorded], 1).numpy()
    for k, v in full.items():
        if torch.is_tensor(v):
            d[k] = v.numpy()
    np.savez_compressed(os.path.join(out, f"{name}.npz"), **d)
    acc = full["acc_map"].numpy()
    print(f"[{name}] rays={n} keys={sorted(k for k in full)} acc in [{acc.min():.3f},{acc.max():.3f}]")


GRAD_SAMPLES = 256


def grad_sample_index(numel):
    """Fixed positions at which a gradient tensor is stored in the train_grads fixtures (plus its L2 norm)."""
    return (np.arange(GRAD_SAMPLES, dtype=np.int64) * 7919) % numel


N_COND = 8


def _sampled_grads(caster):
    out = {}
    for tag, net in (("coarse", caster.network), ("fine", caster.network_fine)):
        if net is None:
            continue
        for pname, p_ in net.named_parameters():
            g = p_.grad.detach().numpy().reshape(-1)
            out[(tag, pname)] = g[grad_sample_index(g.size)].copy()
    return out


def _grad_sensitivity(caster, step):
    """Largest change of a sampled gradient value, relative to its tensor's largest sampled entry, over N_COND reruns
    of `step` with every parameter multiplied by (1 + 1e-7 * normal)."""
    import torch
    base = _sampled_grads(caster)
    params = list(caster.parameters())
    keep = [p_.detach().clone() for p_ in params]
    gen = torch.Generator().manual_seed(5)
    worst = 0.0
    for _ in range(N_COND):
        with torch.no_grad():
            for p_, k_ in zip(params, keep):
                p_.copy_(k_ * (1 + 1e-7 * torch.randn(k_.shape, generator=gen)))
        step()
        got = _sampled_grads(caster)
        for key, b in base.items():
            worst = max(worst, float(np.abs(got[key] - b).max()) / max(float(np.abs(b).max()), 1e-12))
    with torch.no_grad():
        for p_, k_ in zip(params, keep):
            p_.copy_(k_)
    return worst


def gen_train_grads(out, name, cfg, *, n_rays, H, seed_model=0, seed_pose=1, perturb=1., raw_noise_std=1.,
                    use_cams=False, max_seed_tries=16, max_sens=1e-5):
    """One training step of the reference, up to the gradients: `RayCaster.__call__` in training mode with
    pytest=True draws (as rays_train), the loss of Trainer.compute_loss for the shipped surreal config
    (core/trainer.py:321-383: img2mse of rgb + (1 - acc) * bg, use_background=True, base_bg=1, for the fine and the
    coarse maps, coarse_weight 1) against seeded target colours, and `loss.backward()` (trainer.py:463).  Stored: the
    inputs, the draws, the outputs, the loss and every parameter gradient of both nets as (L2 norm, 256 values at
    grad_sample_index) -- the full tensors are 7 MB.

    Conditioning: the gradient of a ReLU net jumps where a pre-activation crosses zero, and a batch has millions of
    them, some within a rounding error of zero.  On such a batch the reference's own gradient moves by up to 4e-3 of
    a tensor's largest entry when its weights move by 1e-7 relative (seen with seed_pose=2 of the h36m case: one
    view-layer unit of one coarse point), so the batch cannot pin another implementation to 1e-4.  The pose seed is
    therefore advanced until the reference's gradients move by <= `max_sens` (1e-5) under N_COND such perturbations; the
    seed used and the measured sensitivity are stored (`seed_pose`, `grad_sensitivity`).  `max_sens=None` takes the
    FIRST batch whatever its conditioning (the un-filtered fixture: its test sets the tolerance from the stored
    sensitivity instead of rejecting the batch)."""
    import torch
    from core.trainer import img2mse
    from core.utils.ray_utils import kp_to_valid_rays
    from posegen_amd import synthetic as syn
    with tempfile.TemporaryDirectory() as wd:
        caster, kw, (wc, wf, tau_v, tau_d) = _build_reference_caster(cfg, seed_model, wd)
    W = H
    for seed_pose in range(seed_pose, seed_pose + max_seed_tries):
        bones, kps, skts = syn.make_pose(1, seed_pose)
        c2ws, focals = syn.make_camera(1, H, W)
        rays, vids, cyls, boxes = kp_to_valid_rays(torch.tensor(c2ws), H, W, focals,
                                                   kps=torch.tensor(kps), ext_scale=cfg.ext_scale)
        ro, rd = rays[0]
        sel = np.unique(np.linspace(0, ro.shape[0] - 1, n_rays).round().astype(np.int64))
        ro, rd = ro[sel].float(), rd[sel].float()
        n = ro.shape[0]
        vd = rd / torch.norm(rd, dim=-1, keepdim=True)
        ones = torch.ones(n, 1)
        batch = torch.cat([ro, rd, 0. * ones, 1. * ones, vd], -1)
        kp_b = torch.tensor(kps).expand(n, -1, -1)
        skt_b = torch.tensor(skts).expand(n, -1, -1, -1)
        cyl_b = cyls.expand(n, -1)
        bones_b = torch.tensor(bones).expand(n, -1, -1)
        S, N = cfg.n_samples, cfg.n_importance
        rng = np.random.RandomState(11)
        target = torch.tensor(rng.uniform(0, 1, size=(n, 3)).astype(np.float32))
        cams = torch.tensor(rng.randint(0, cfg.n_framecodes, size=n).astype(np.float32)) if use_cams else None
        call_kw = {k: v for k, v in kw.items() if k != "ray_caster"}
        call_kw.pop("use_viewdirs", None)
        call_kw.update(perturb=perturb, raw_noise_std=raw_noise_std, ray_noise_std=0., pytest=True)

        def step():
            caster.train()
            for p_ in caster.parameters():
                p_.grad = None
            full = caster(batch, kp_batch=kp_b, skts=skt_b, cyls=cyl_b, bones=bones_b, cams=cams, subject_idxs=None, **call_kw)
            loss = img2mse(full["rgb_map"] + (1. - full["acc_map"])[..., None] * 1.0, target, reduction="mean")
            if "rgb0" in full:
                loss = loss + img2mse(full["rgb0"] + (1. - full["acc0"])[..., None] * 1.0, target, reduction="mean") * 1.0
            loss.backward()
            return full, loss

        full, loss = step()
        sens = _grad_sensitivity(caster, step)
        print(f"[{name}] seed_pose={seed_pose}: gradient sensitivity to 1e-7 weight noise {sens:.2e}")
        if max_sens is None or sens <= max_sens:
            break
    else:
        raise SystemExit(f"[{name}] no well-conditioned batch in {max_seed_tries} pose seeds")
    full, loss = step()             # the gradients of the unperturbed weights back in .grad
    caster.eval()
    d = {"ray_batch": batch.numpy(), "kps": kps, "skts": skts, "bones": bones, "cyl": cyls.numpy(), "target": target.numpy(),
         "tau_v": tau_v, "tau_d": tau_d, "seed_model": seed_model, "n_samples": S, "n_importance": N,
         "framecode_ch": cfg.framecode_ch, "n_framecodes": cfg.n_framecodes,
         "digest_coarse": _weights_digest(wc), "digest_fine": _weights_digest(wf),
         "perturb": perturb, "raw_noise_std": raw_noise_std, "ray_noise_std": 0., "n_rays": n, "loss": float(loss),
         "seed_pose": seed_pose, "grad_sensitivity": sens}
    if cfg.density_type != "relu":
        d["density_softplus"] = int(cfg.density_type == "softplus")
        d["softplus_shift"] = float(cfg.softplus_shift)
    if cams is not None:
        d["cams"] = cams.numpy()
    f32 = lambda a: torch.Tensor(a).numpy()
    if perturb > 0:
        np.random.seed(0); d["t_rand"] = f32(np.random.rand(n, S))
        if N > 0:
            np.random.seed(0); d["u_rand"] = f32(np.random.rand(n, N))
    if raw_noise_std > 0:
        np.random.seed(0); d["noise0"] = f32(np.random.rand(n, S) * raw_noise_std)
        if N > 0:
            np.random.seed(0); d["noise1"] = f32(np.random.rand(n, S + N) * raw_noise_std)
    for k in ("rgb_map", "acc_map", "rgb0", "acc0"):
        if k in full:
            d[k] = full[k].detach().numpy()
    worst = 0.0
    for tag, net in (("coarse", caster.network), ("fine", caster.network_fine)):
        if net is None:
            continue
        for pname, p_ in net.named_parameters():
            g = p_.grad
            assert g is not None, (tag, pname)
            g = g.detach().numpy().reshape(-1)
            d[f"gnorm_{tag}_{pname}"] = np.float64(np.linalg.norm(g.astype(np.float64)))
            d[f"gval_{tag}_{pname}"] = g[grad_sample_index(g.size)]
            worst = max(worst, float(np.abs(g).max()))
    np.savez_compressed(os.path.join(out, f"{name}.npz"), **d)
    print(f"[{name}] rays={n} loss={float(loss):.6f} max |grad| {worst:.3e}, {sum(1 for k in d if k.startswith('gnorm_'))} gradient tensors")


def gen_frame(out, name, cfg, H, chunk, seed_model=0, seed_pose=1, n_frames=2):
    """a-1/a-3: whole frames through the reference's render_path (bbox cull,
    chunk loop with a chunk boundary inside the frame, white background)."""
    import torch
    import run_nerf
    from posegen_amd import synthetic as syn
    with tempfile.TemporaryDirectory() as wd:
        caster, kw, (wc, wf, tau_v, tau_d) = _build_reference_caster(cfg, seed_model, wd)
    W = H
    bones, kps, skts = syn.make_pose(n_frames, seed_pose)
    c2ws, focals = syn.make_camera(n_frames, H, W)
    import tqdm as _tqdm
    run_nerf.tqdm = lambda x, *a, **k: x
    rgbs, disps, accs, vids, boxes = run_nerf.render_path(
        torch.tensor(c2ws), (H, W, focals), chunk, kw, kp=torch.tensor(kps),
        skts=torch.tensor(skts), bones=torch.tensor(bones), cams=None, white_bkgd=True,
        ret_acc=True, ext_scale=cfg.ext_scale)
    np.savez_compressed(
        os.path.join(out, f"{name}.npz"), H=H, W=W, chunk=chunk, c2ws=c2ws, focals=focals,
        bones=bones, kps=kps, skts=skts, tau_v=tau_v, tau_d=tau_d, seed_model=seed_model,
        n_samples=cfg.n_samples, n_importance=cfg.n_importance,
        rgbs=rgbs, disps=disps, accs=accs, n_valid=np.array([len(v) for v in vids]),
        boxes=np.array([[b[0], b[1]] for b in boxes]),
        digest_coarse=_weights_digest(wc), digest_fine=_weights_digest(wf))
    print(f"[{name}] frames={n_frames} valid={[len(v) for v in vids]} acc max={accs.max():.3f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    _install_shims(a.ref)
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    from posegen_amd.config import surreal_config, h36m_config
    os.makedirs(a.out, exist_ok=True)
    only = set(filter(None, a.only.split(",")))
    want = lambda k: not only or k in only
    if want("kinematics"):
        gen_kinematics(a.out)
    if want("valid_rays"):
        gen_valid_rays(a.out)
    if want("rays_surreal"):     # culled frame: contains cylinder misses -> nanmean path
        gen_render_rays(a.out, "rays_surreal", surreal_config(), n_rays=256, H=128, all_hit=False)
    if want("rays_allhit"):      # headline variant: radius 2.5, every ray hits
        gen_render_rays(a.out, "rays_allhit", surreal_config(), n_rays=128, H=128, all_hit=True,
                        seed_pose=2)
    if want("rays_coarse32"):    # N_importance = 0, 32 samples
        gen_render_rays(a.out, "rays_coarse32", surreal_config(n_samples=32, n_importance=0),
                        n_rays=96, H=64, all_hit=False, seed_pose=3)
    if want("rays_cfg1"):        # BASELINE config 1: 32 coarse + 16 importance
        gen_render_rays(a.out, "rays_cfg1", surreal_config(n_samples=32, n_importance=16),
                        n_rays=96, H=128, all_hit=False, seed_pose=4)
    if want("rays_train"):       # training-mode call: perturb + raw noise + ray noise, pytest=True draws
        gen_render_rays_train(a.out, "rays_train", surreal_config(), n_rays=64, H=128, seed_pose=6)
    if want("rays_train_coarse"):  # N_importance = 0, density noise and jitter only
        gen_render_rays_train(a.out, "rays_train_coarse", surreal_config(n_samples=32, n_importance=0),
                              n_rays=48, H=64, seed_pose=7, ray_noise_std=0.)
    if want("train_grads"):      # one training step up to loss.backward(): gradients of all 24 tensors per net
        gen_train_grads(a.out, "train_grads", surreal_config(), n_rays=48, H=128, seed_pose=8)
    if want("train_grads_h36m"): # the same with frame codes (per-ray index): + framecodes.codes.weight
        gen_train_grads(a.out, "train_grads_h36m", h36m_config(n_samples=64, n_importance=16), n_rays=24, H=128,
                        seed_model=5, seed_pose=9, use_cams=True)
    if want("rays_softplus"):    # --density_type softplus --softplus_shift 1.0 (raycasters.py:230-238), hierarchical
        gen_render_rays(a.out, "rays_softplus", surreal_config(density_type="softplus", softplus_shift=1.0),
                        n_rays=96, H=128, all_hit=False, seed_pose=10)
    if want("train_grads_softplus"):   # the training step with the softplus density (its derivative in the backward pass)
        gen_train_grads(a.out, "train_grads_softplus", surreal_config(density_type="softplus", softplus_shift=1.0),
                        n_rays=32, H=128, seed_pose=12)
    if want("train_grads_raw"):  # the FIRST h36m batch, whatever its conditioning (a ReLU kink within rounding of zero)
        gen_train_grads(a.out, "train_grads_raw", h36m_config(n_samples=64, n_importance=16), n_rays=24, H=128,
                        seed_model=5, seed_pose=2, use_cams=True, max_sens=None)
    if want("rays_h36m"):        # BASELINE config 4: frame codes, 128 coarse + 16
        gen_render_rays(a.out, "rays_h36m", h36m_config(), n_rays=64, H=128, all_hit=False,
                        seed_model=5, seed_pose=5, use_cams=True)
    if want("frame64"):
        gen_frame(a.out, "frame64", surreal_config(), H=64, chunk=1024)


if __name__ == "__main__":
    main()
