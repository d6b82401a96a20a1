"""The render call of the PoseGen GAN loop, kept on the device (BASELINE config 5).

In the reference every `run_render` call of the training loop (run_gan.py:2041-2091, 2299-2337)
  * re-parses the arguments and re-loads the A-NeRF checkpoint (load_nerf, run_gan.py:135-165),
  * moves the generator's poses to the host, runs the kinematics and the bounding-box cull in numpy,
  * renders rpi = 20 frames through nn.DataParallel(RayCaster),
  * writes them as PNG files, reads them back, crops [100:412]^2, normalises, and resizes to 224 x 224
    with skimage (anti_aliasing=True) for the SPIN / HMR regressor.
`render_for_regressor` is that call with the HIP renderer: poses stay on the GPU (pg_pose_kinematics ->
pg_pose_boxes -> pg_render_frame with a uint8 frame), no files, and one 16-byte-per-frame device->host copy
(the integer boxes size the launches); the crop / normalise / anti-aliased resize run as torch ops on the
device.  The caster comes from `load_raycaster` (memoised: no checkpoint reload per call).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# ImageNet statistics the reference normalises with (run_gan.py:2342-2343)
IMG_NORM_MEAN = (0.485, 0.456, 0.406)
IMG_NORM_STD = (0.229, 0.224, 0.225)


def resize_antialiased(img: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """skimage.transform.resize(img, size, anti_aliasing=True) for [N,C,h,w] -> [N,C,H,W] on the device:
    Gaussian prefilter with sigma = (scale - 1) / 2 per axis (truncated at 4 sigma, mirrored borders),
    then order-1 interpolation at pixel centres (run_gan.py:2064-2065)."""
    n, c, h, w = img.shape
    out = img
    for axis, (src, dst) in ((2, (h, size[0])), (3, (w, size[1]))):
        sigma = max(0.0, (src / dst - 1.0) / 2.0)
        if sigma <= 0:
            continue
        radius = int(4.0 * sigma + 0.5)
        if radius == 0:
            continue
        x = torch.arange(-radius, radius + 1, device=img.device, dtype=torch.float32)
        k = torch.exp(-0.5 * (x / sigma) ** 2)
        k = k / k.sum()
        shape = [1, 1, 1, 1]
        shape[axis] = 2 * radius + 1
        pad = [0, 0, 0, 0]
        pad[(3 - axis) * 2] = pad[(3 - axis) * 2 + 1] = radius
        out = F.pad(out, pad, mode="reflect")              # scipy.ndimage 'mirror' = torch 'reflect'
        out = F.conv2d(out.reshape(n * c, 1, *out.shape[2:]), k.reshape(shape)).reshape(n, c, h, w)
    return F.interpolate(out, size=size, mode="bilinear", align_corners=False)


@torch.no_grad()
def render_for_regressor(caster, bones: torch.Tensor, rest_pose, c2w, H: int, W: int, focal: float,
                         ext_scale: float = 0.001, crop: Tuple[int, int] = (100, 412), out_res: int = 224,
                         white_bkgd: bool = True, chunk: int = 4096, n_samples: Optional[int] = None,
                         n_importance: Optional[int] = None, return_frames: bool = False):
    """bones [F,24,3] axis-angle on the device -> regressor input [F,3,out_res,out_res] on the device
    (+ the uint8 frames [F,H,W,3] if `return_frames`).  One camera `c2w` [4,4] for all frames, like the
    fixed extrinsic of the loop (run_gan.py:2023-2028)."""
    r = caster.renderer
    dev = r.device
    kps, skts = r.pose_kinematics(bones, rest_pose)
    cyls, boxes = r.pose_boxes(kps, c2w, H, W, focal, ext_scale)
    boxes_h = boxes.cpu().numpy()                           # 16 B per frame: the only host round trip
    r.set_chunk(int(chunk))
    c2w_np = np.asarray(torch.as_tensor(c2w).detach().cpu(), dtype=np.float32)
    frames = torch.empty(bones.shape[0], H, W, 3, device=dev, dtype=torch.uint8)
    for i in range(bones.shape[0]):
        b = boxes_h[i]
        _, _, _, rgb8 = r.render_frame(H, W, focal, c2w_np, ((int(b[0]), int(b[1])), (int(b[2]), int(b[3]))), skts[i:i + 1],
                                       cyls[i:i + 1], n_samples=n_samples, n_importance=n_importance,
                                       base_bg=1.0 if white_bkgd else 0.0, want_uint8=True)
        frames[i] = rgb8
    img = frames[:, crop[0]:crop[1], crop[0]:crop[1], :].permute(0, 3, 1, 2).float() / 255.0
    mean = torch.tensor(IMG_NORM_MEAN, device=dev).view(1, 3, 1, 1)
    std = torch.tensor(IMG_NORM_STD, device=dev).view(1, 3, 1, 1)
    img = resize_antialiased((img - mean) / std, (out_res, out_res))
    return (img, frames) if return_frames else img
