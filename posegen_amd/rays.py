"""Ray generation and the cylinder -> 2-D box cull in front of the renderer.

Mirrors `get_rays` and `kp_to_valid_rays` of the reference
(core/utils/ray_utils.py:6-28, 83-136).  Only the rays inside the box are generated
(the reference builds the full frame and gathers; the per-pixel arithmetic is the same).
"""
from __future__ import annotations

import numpy as np
import torch

from .skeleton import cylinder_to_box_2d, get_kp_bounding_cylinder, nerf_c2w_to_extrinsic


def _focal_xy(focal):
    f = np.asarray(focal.detach().cpu() if isinstance(focal, torch.Tensor) else focal, dtype=np.float64).reshape(-1)
    return (float(f[0]), float(f[0])) if f.size < 2 else (float(f[0]), float(f[1]))


def pixel_rays(rows: torch.Tensor, cols: torch.Tensor, H, W, focal, c2w: torch.Tensor, center=None):
    """Un-normalised rays through pixels (row, col): dirs = ((i-cx)/fx, -(j-cy)/fy, -1),
    rays_d = dirs . c2w[:3,:3]^T, rays_o = c2w[:3,3]   (ray_utils.py:6-28)."""
    fx, fy = _focal_xy(focal)
    cx, cy = (W * 0.5, H * 0.5) if center is None else (float(center[0]), float(center[1]))
    c2w = torch.as_tensor(c2w, dtype=torch.float32)
    i = cols.to(torch.float32)
    j = rows.to(torch.float32)
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def get_rays(H, W, focal, c2w, center=None):
    rows = torch.arange(H)[:, None].expand(H, W)
    cols = torch.arange(W)[None, :].expand(H, W)
    return pixel_rays(rows, cols, H, W, focal, c2w, center)


def kp_to_boxes(poses, H, W, focal, kps=None, cylinder_params=None, centers=None, ext_scale=0.00035):
    """Per frame: the bounding cylinder (float32 tensor [F,5]), its projected 2-D box (tl, br) and
    the flat ids of the pixels inside it (`br` row/column excluded, like the reference's
    torch.arange(tl, br)) -- kp_to_valid_rays without the rays (ray_utils.py:83-136)."""
    if cylinder_params is None:
        assert kps is not None
        cylinder_params = get_kp_bounding_cylinder(kps.detach().cpu().numpy(), ext_scale=ext_scale,
                                                   extend_mm=250, top_expand_ratio=1.60,
                                                   bot_expand_ratio=1.10, head="-y")
        cylinder_params = torch.tensor(np.asarray(cylinder_params), dtype=torch.float32)
    n_pose = cylinder_params.shape[0] if kps is None else kps.shape[0]
    grids, bboxes = [], []
    for i, c2w in enumerate(poses):
        cyl = cylinder_params[i % n_pose]
        f = focal if isinstance(focal, float) else focal[i]
        center = None if centers is None else centers[i]
        h = int(H) if isinstance(H, (int, np.integer)) else H[i]
        w = int(W) if isinstance(W, (int, np.integer)) else W[i]
        c2w_np = np.asarray(c2w.detach().cpu() if isinstance(c2w, torch.Tensor) else c2w)
        tl, br, _ = cylinder_to_box_2d(cyl.detach().cpu().numpy(), [h, w, f], nerf_c2w_to_extrinsic(c2w_np),
                                       center=center)
        rr = torch.arange(int(tl[1]), int(br[1]))
        cc = torch.arange(int(tl[0]), int(br[0]))
        rows = rr[:, None].expand(len(rr), len(cc)).reshape(-1)
        cols = cc[None, :].expand(len(rr), len(cc)).reshape(-1)
        grids.append((rows, cols, h, w, f, c2w_np, center))
        bboxes.append((tl, br))
    return cylinder_params, bboxes, grids


def kp_to_valid_rays(poses, H, W, focal, kps=None, cylinder_params=None, skts=None, centers=None,
                     ext_scale=0.00035):
    """Per frame: rays of the pixels inside the projected bounding cylinder's box, their
    flat pixel ids, the cylinders (float32 tensor [F,5]) and the (tl, br) boxes."""
    cylinder_params, bboxes, grids = kp_to_boxes(poses, H, W, focal, kps=kps, cylinder_params=cylinder_params,
                                                 centers=centers, ext_scale=ext_scale)
    rays, valid_idxs = [], []
    for rows, cols, h, w, f, c2w_np, center in grids:
        rays.append(pixel_rays(rows, cols, h, w, f, torch.as_tensor(c2w_np), center))
        valid_idxs.append(rows * w + cols)
    return rays, valid_idxs, cylinder_params, bboxes
