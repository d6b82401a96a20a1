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


class BoxPixelIds:
    """`valid_idxs` of kp_to_valid_rays (ray_utils.py:127-133) as a lazy sequence: frame i's flat pixel ids
    (row * W + col of the box, `br` row / column excluded) are built when asked for -- the renderer itself only
    needs the boxes, and 200 k int64 ids per 512 x 512 frame are 10 ms of host work per 20-frame call."""

    def __init__(self, boxes, widths):
        self._boxes = [(int(tl[0]), int(tl[1]), int(br[0]), int(br[1])) for tl, br in boxes]
        self._w = [int(w) for w in widths]
        self._cache = {}

    def __len__(self):
        return len(self._boxes)

    def count(self, i):
        tlx, tly, brx, bry = self._boxes[i]
        return max(brx - tlx, 0) * max(bry - tly, 0)

    def counts(self):
        return [self.count(i) for i in range(len(self))]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        i = range(len(self))[i]
        if i not in self._cache:
            tlx, tly, brx, bry = self._boxes[i]
            rr, cc = torch.arange(tly, bry), torch.arange(tlx, brx)
            self._cache[i] = (rr[:, None] * self._w[i] + cc[None, :]).reshape(-1)
        return self._cache[i]

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def tolist(self):
        """The reference's return type: a plain list of int64 id tensors, one per frame (builds them all)."""
        return list(self)

    def __array__(self, dtype=None, copy=None):
        """numpy view of the sequence: an object array of the per-frame id arrays (frames have different box sizes)."""
        out = np.empty(len(self), dtype=object)
        for i in range(len(self)):
            out[i] = self[i].numpy() if dtype is None else self[i].numpy().astype(dtype)
        return out

    def __add__(self, other):
        return self.tolist() + list(other)

    def __radd__(self, other):
        return list(other) + self.tolist()

    def __reduce__(self):               # pickles as what it stands for: the list of id tensors
        return (list, (self.tolist(),))


def frame_boxes(renderer, poses, H, W, focal, kps=None, cylinder_params=None, centers=None, ext_scale=0.00035):
    """Bounding cylinders and integer boxes of `len(poses)` frames without the pixel grids: (cyls [n_pose,5] float32
    tensor, list of (tl, br) int32 arrays, per-frame (h, w, focal, c2w numpy, center)).

    With key points, one frame size, one focal length and the default principal point the cylinders and boxes come
    from the device (`pg_pose_boxes`: bit for bit the reference's boxes, one 16-byte-per-frame copy back); anything
    else takes the host route of kp_to_boxes, frame by frame in float64 numpy like the reference."""
    F = len(poses)
    scalar_hw = isinstance(H, (int, np.integer)) and isinstance(W, (int, np.integer))
    f_arr = None if isinstance(focal, float) else np.asarray(torch.as_tensor(focal).detach().cpu(), dtype=np.float64)
    one_focal = f_arr is None or f_arr.ndim == 0 or (f_arr.ndim == 1 and f_arr.size > 0 and np.all(f_arr == f_arr[0]))
    meta = []
    for i, c2w in enumerate(poses):
        f = focal if isinstance(focal, float) else focal[i]
        h = int(H) if isinstance(H, (int, np.integer)) else H[i]
        w = int(W) if isinstance(W, (int, np.integer)) else W[i]
        meta.append((h, w, f, np.asarray(c2w.detach().cpu() if isinstance(c2w, torch.Tensor) else c2w),
                     None if centers is None else centers[i]))
    if (hasattr(renderer, "pose_boxes") and kps is not None and cylinder_params is None and centers is None and scalar_hw
            and one_focal and F > 0):
        kp_t = torch.as_tensor(kps).reshape(-1, 24, 3)
        n_pose = kp_t.shape[0]
        c2w_all = np.stack([m[3] for m in meta]).astype(np.float32)
        same_cam = bool(np.all(c2w_all == c2w_all[:1]))
        # one box per FRAME (frame i uses pose i % n_pose, run_nerf.py:63-74) ...
        kp_f = kp_t if (n_pose == F) else kp_t[torch.arange(F) % n_pose]
        f0 = float(focal) if f_arr is None else float(f_arr.reshape(-1)[0])
        cyl_f, boxes = renderer.pose_boxes(kp_f, c2w_all[:1] if same_cam else c2w_all, int(H), int(W), f0, ext_scale)
        bh = boxes.cpu().numpy()                                   # [F,4] int32: the only host round trip
        bboxes = [(bh[i, 0:2].copy(), bh[i, 2:4].copy()) for i in range(F)]
        # ... the cylinders indexed like the reference's cylinder_params[i % n_pose] (frame i of F <= n_pose frames
        # is pose i; with more frames than poses the first n_pose frames are the poses)
        return (cyl_f if F <= n_pose else cyl_f[:n_pose]), bboxes, meta
    if cylinder_params is None:
        assert kps is not None
        cylinder_params = get_kp_bounding_cylinder(torch.as_tensor(kps).detach().cpu().numpy(), ext_scale=ext_scale,
                                                   extend_mm=250, top_expand_ratio=1.60, bot_expand_ratio=1.10, head="-y")
        cylinder_params = torch.tensor(np.asarray(cylinder_params), dtype=torch.float32)
    n_pose = cylinder_params.shape[0] if kps is None else torch.as_tensor(kps).shape[0]
    bboxes = []
    for i, (h, w, f, c2w_np, center) in enumerate(meta):
        cyl = cylinder_params[i % n_pose]
        tl, br, _ = cylinder_to_box_2d(torch.as_tensor(cyl).detach().cpu().numpy(), [h, w, f], nerf_c2w_to_extrinsic(c2w_np), center=center)
        bboxes.append((tl, br))
    return cylinder_params, bboxes, meta


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
