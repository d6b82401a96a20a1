"""Frame-level drivers with the reference's call surface.

`render_path` mirrors run_nerf.render_path (run_nerf.py:27-147) and `render` /
`batchify_rays` mirror core/trainer.py:64-147, so the GAN / dataset-generation loop can
call them unchanged -- but the chunk loop, the per-chunk host->device copies and the
per-ray replication of the pose tensors are gone: a whole frame's rays go to the HIP
library in one call (the library applies the `chunk`-ray nanmean groups itself) and the
pose of a frame is passed once.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .rays import kp_to_valid_rays


def _caster_device(ray_caster):
    r = getattr(getattr(ray_caster, "module", ray_caster), "renderer", None)
    if r is None:
        raise TypeError("render() needs a posegen_amd.HipRayCaster under 'ray_caster' "
                        "(there is no eager / CPU fallback on this path)")
    return r, r.device


def batchify_rays(rays_flat, chunk=1024 * 32, ray_caster=None, **kwargs):
    """All rays in ONE call; `chunk` only sets the nanmean group size (trainer.py:64-81)."""
    r, dev = _caster_device(ray_caster)
    r.set_chunk(int(chunk))
    return ray_caster(rays_flat.to(dev), **kwargs)


def render(H, W, focal, chunk=1024 * 32, rays=None, c2w=None, near=0., far=1., center=None,
           use_viewdirs=False, c2w_staticcam=None, **kwargs):
    """Pack `ray_batch = [o, d, near, far, viewdir]` and render it (trainer.py:84-147)."""
    if rays is None:
        raise NotImplementedError("render(): pass rays=(rays_o, rays_d); full-frame c2w rendering goes "
                                  "through render_path")
    r, dev = _caster_device(kwargs["ray_caster"])
    rays_o, rays_d = rays
    sh = rays_d.shape
    rays_o = torch.reshape(rays_o, [-1, 3]).float().to(dev)
    rays_d = torch.reshape(rays_d, [-1, 3]).float().to(dev)
    ones = torch.ones_like(rays_d[..., :1])
    parts = [rays_o, rays_d, near * ones, far * ones]
    viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    parts.append(viewdirs)                      # columns 8..10 are carried but unused (SURVEY a-5)
    all_ret = batchify_rays(torch.cat(parts, -1), chunk, **kwargs)
    for k in all_ret:
        if all_ret[k].dim() >= 4:
            continue
        all_ret[k] = torch.reshape(all_ret[k], list(sh[:-1]) + list(all_ret[k].shape[1:]))
    return all_ret


def _pick(x, i):
    """The reference's reuse_input: pose i % n, un-expanded ([1,...])."""
    if x is None:
        return None
    return x[i % x.shape[0]:i % x.shape[0] + 1] if x.shape[0] > 1 else x


@torch.no_grad()
def render_path(render_poses, hwf, chunk, render_kwargs, centers=None, kp=None, skts=None, cyls=None,
                bones=None, gt_imgs=None, bg_imgs=None, bg_indices=None, cams=None, subject_idxs=None,
                render_factor=0, white_bkgd=False, ret_acc=False, ext_scale=0.00035, base_bg=1.0,
                frame_ids: Optional[list] = None):
    """Render frames; returns (rgbs [F,H,W,3], disps [F,H,W,1], accs, valid_idxs, bboxes).

    `frame_ids` (extension) restricts rendering to a subset of frames (multi-GPU
    partition); the returned arrays then hold those frames in the given order.
    """
    H, W, focal = hwf
    if render_factor != 0:
        H, W = H // render_factor, W // render_factor
        focal = focal / render_factor if isinstance(focal, float) else focal.copy() / render_factor
        if centers is not None:
            centers = centers / render_factor if isinstance(focal, float) else centers.copy() / render_factor
    if kp is None and cyls is None:
        raise NotImplementedError("render_path needs kp or cyls (bounding-cylinder cull)")
    r, dev = _caster_device(render_kwargs["ray_caster"])
    rays, valid_idxs, cyls, bboxes = kp_to_valid_rays(render_poses, H, W, focal, kps=kp, cylinder_params=cyls,
                                                      skts=skts, ext_scale=ext_scale, centers=centers)
    ids = list(range(len(render_poses))) if frame_ids is None else list(frame_ids)
    rgbs, disps, accs = [], [], []
    kw = dict(render_kwargs)
    kw["want_alpha"] = False                    # render_path reads rgb/disp/acc only (run_nerf.py:98)
    for i in ids:
        h = H if isinstance(H, int) else H[i]
        w = W if isinstance(W, int) else W[i]
        ro, rd = rays[i]
        if bg_imgs is not None and not white_bkgd:
            import torch.nn.functional as F
            bg = torch.tensor(bg_imgs[bg_indices[i]] if bg_indices is not None else bg_imgs[0])
            rgb_img = F.interpolate(bg.permute(2, 0, 1)[None].float(), size=(h, w), mode="bilinear",
                                    align_corners=False)[0].permute(1, 2, 0).reshape(h * w, 3).to(dev)
        else:
            rgb_img = torch.ones(h * w, 3, device=dev) if white_bkgd else torch.zeros(h * w, 3, device=dev)
        disp_img = torch.zeros(h * w, device=dev)
        acc_img = torch.zeros(h * w, device=dev)
        if len(ro) > 0:
            ret = render(h, w, focal, rays=(ro, rd), chunk=chunk, kp_batch=_pick(kp, i), skts=_pick(skts, i),
                         cyls=_pick(cyls, i), cams=_pick(cams, i), subject_idxs=_pick(subject_idxs, i),
                         bones=_pick(bones, i), **kw)
            vid = valid_idxs[i].to(dev)
            acc = ret["acc_map"]
            rgb_img[vid] = ret["rgb_map"] + (1. - acc[..., None]) * rgb_img[vid]
            disp_img[vid] = ret["disp_map"]
            acc_img[vid] = acc
        rgbs.append(rgb_img.view(h, w, 3))
        disps.append(disp_img.view(h, w, 1))
        accs.append(acc_img.view(h, w, 1))
    rgbs = torch.stack(rgbs).cpu().numpy()
    disps = torch.stack(disps).cpu().numpy()
    disps[np.isnan(disps)] = 0.
    accs = torch.stack(accs).cpu().numpy() if ret_acc else []
    return rgbs, disps, accs, valid_idxs, bboxes
