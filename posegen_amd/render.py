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

from .rays import BoxPixelIds, frame_boxes, get_rays, kp_to_boxes, kp_to_valid_rays


# render_path: results of up to this many bytes are page-locked as a whole; longer paths go through a ring of
# PINNED_RING page-locked frames (hundreds of megapixel frames must not pin gigabytes)
PINNED_RESULT_BYTES = 256 << 20
PINNED_RING = 3


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
    inner = getattr(ray_caster, "module", ray_caster)
    inner._grouped_call = True          # keep the `chunk` groups: forward() alone is one group per call
    try:
        return ray_caster(rays_flat.to(dev), **kwargs)
    finally:
        inner._grouped_call = False


def render(H, W, focal, chunk=1024 * 32, rays=None, c2w=None, near=0., far=1., center=None,
           use_viewdirs=False, c2w_staticcam=None, **kwargs):
    """Pack `ray_batch = [o, d, near, far, viewdir]` and render it (trainer.py:84-147)."""
    r, dev = _caster_device(kwargs["ray_caster"])
    if rays is None:
        if c2w is None:
            raise ValueError("render(): pass rays=(rays_o, rays_d) or a camera c2w")
        # the full-image special case (trainer.py:109-113): every pixel of the H x W frame
        center = None if center is None else np.asarray(center).ravel()
        rays_o, rays_d = get_rays(H, W, focal, torch.as_tensor(c2w, dtype=torch.float32), center=center)
    else:
        rays_o, rays_d = rays
    if c2w_staticcam is not None:
        # trainer.py:122-124: directions of `c2w` feed the (numerically dead, SURVEY a-5) viewdir columns,
        # the rays themselves come from the static camera
        rays_o, rays_d = get_rays(H, W, focal, torch.as_tensor(c2w_staticcam, dtype=torch.float32))
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
def render_frames_device(render_poses, hwf, chunk, render_kwargs, centers=None, kp=None, skts=None, cyls=None,
                         bg_imgs=None, bg_indices=None, cams=None, render_factor=0, white_bkgd=False,
                         ext_scale=0.00035, frame_ids: Optional[list] = None, boxes=None, frame_sink=None):
    """The frame loop of `render_path` with its results left on the device:
    (rgbs [f,H,W,3], disps [f,H,W,1], accs [f,H,W,1] device tensors, valid_idxs, bboxes).
    `frame_ids` restricts rendering to a subset of the frames (multi-GPU partition; an empty
    list returns empty [0,H,W,C] stacks); `boxes` = a kp_to_boxes result computed by the caller;
    `frame_sink(k, rgb, disp, acc)` is called with the k-th rendered frame's device tensors as soon as its kernels
    are enqueued (render_path starts the device-to-host copy there) -- the stacks are then not built (None)."""
    H, W, focal = hwf
    if render_factor != 0:
        H, W = H // render_factor, W // render_factor
        focal = focal / render_factor if isinstance(focal, float) else focal.copy() / render_factor
        if centers is not None:
            centers = centers / render_factor if isinstance(focal, float) else centers.copy() / render_factor
    if kp is None and cyls is None:
        raise NotImplementedError("render_path needs kp or cyls (bounding-cylinder cull)")
    r, dev = _caster_device(render_kwargs["ray_caster"])
    # Boxes, rays, rendering and the scatter into the background frame on the device (pg_pose_boxes,
    # pg_render_frame): no per-frame meshgrid, no ray copies.
    if boxes is not None:
        cyls, bboxes, grids = boxes
        meta = [g[2:] for g in grids]
    else:                       # (boxes from the device when the call allows it; the pixel ids only if someone reads them)
        cyls, bboxes, meta = frame_boxes(r, render_poses, H, W, focal, kps=kp, cylinder_params=cyls, ext_scale=ext_scale,
                                         centers=centers)
    valid_idxs = BoxPixelIds(bboxes, [m[1] for m in meta])
    ids = list(range(len(render_poses))) if frame_ids is None else list(frame_ids)
    rgbs, disps, accs = [], [], []
    kw = render_kwargs
    r.set_chunk(int(chunk))
    # poses and cylinders go to the device once: a per-frame host-to-device copy of 1.5 KB is a blocking call that
    # keeps the host from enqueueing the next frame while the GPU works on this one
    if torch.is_tensor(skts) and skts.device.type == "cpu":
        skts = skts.to(dev, dtype=torch.float32)
    if cyls is not None:
        cyls = torch.as_tensor(cyls).to(dev, dtype=torch.float32)
    if cams is not None:        # frame-code indices on the host once (a per-frame float() of a device tensor would block)
        cams = torch.as_tensor(cams).detach().float().cpu()
    for k, i in enumerate(ids):
        h, w, f, c2w_np, center = meta[i]
        bg = None
        if bg_imgs is not None and not white_bkgd:
            import torch.nn.functional as F
            bgi = torch.tensor(bg_imgs[bg_indices[i]] if bg_indices is not None else bg_imgs[0])
            bg = F.interpolate(bgi.permute(2, 0, 1)[None].float(), size=(h, w), mode="bilinear",
                               align_corners=False)[0].permute(1, 2, 0).reshape(h * w, 3).to(dev)
        cam = _pick(cams, i)
        rgb_img, disp_img, acc_img = r.render_frame(
            h, w, f, c2w_np, bboxes[i], _pick(skts, i), _pick(cyls, i), center=center,
            cam=None if cam is None else float(cam.reshape(-1)[0]),
            n_samples=kw.get("N_samples"), n_importance=kw.get("N_importance"), lindisp=bool(kw.get("lindisp", False)),
            bg=bg, base_bg=1.0 if white_bkgd else 0.0)
        if frame_sink is not None:
            frame_sink(k, rgb_img, torch.nan_to_num(disp_img, nan=0.0, posinf=float("inf"), neginf=float("-inf")), acc_img)
            continue
        rgbs.append(rgb_img)
        disps.append(disp_img)
        accs.append(acc_img)
    if frame_sink is not None:
        return None, None, None, valid_idxs, bboxes
    if not ids:
        if not (isinstance(H, (int, np.integer)) and isinstance(W, (int, np.integer))):
            raise ValueError("an empty frame share needs scalar H, W to shape its (empty) result")
        e = lambda c: torch.zeros((0, int(H), int(W), c), device=dev)
        return e(3), e(1), e(1), valid_idxs, bboxes
    rgbs, disps, accs = torch.stack(rgbs), torch.stack(disps), torch.stack(accs)
    disps = torch.nan_to_num(disps, nan=0.0, posinf=float("inf"), neginf=float("-inf"))   # run_nerf.py:142-143
    return rgbs, disps, accs, valid_idxs, bboxes


@torch.no_grad()
def render_path(render_poses, hwf, chunk, render_kwargs, centers=None, kp=None, skts=None, cyls=None,
                bones=None, gt_imgs=None, bg_imgs=None, bg_indices=None, cams=None, subject_idxs=None,
                render_factor=0, white_bkgd=False, ret_acc=False, ext_scale=0.00035, base_bg=1.0,
                frame_ids: Optional[list] = None):
    """Render frames; returns (rgbs [F,H,W,3], disps [F,H,W,1], accs, valid_idxs, bboxes) as the
    reference does: numpy arrays on the host (run_nerf.py:27-147).

    `frame_ids` (extension) restricts rendering to a subset of frames (multi-GPU
    partition); the returned arrays then hold those frames in the given order.
    """
    r, _ = _caster_device(render_kwargs["ray_caster"])
    if getattr(r, "n_devices", 1) > 1 and frame_ids is None and (bg_imgs is None or white_bkgd or bg_indices is None):
        return _render_path_multi(r, render_poses, hwf, chunk, render_kwargs, centers, kp, skts, cyls, bg_imgs, cams,
                                  render_factor, white_bkgd, ret_acc, ext_scale)
    H, W = hwf[0], hwf[1]
    n_out = len(render_poses) if frame_ids is None else len(frame_ids)
    _, dev = _caster_device(render_kwargs["ray_caster"])
    if n_out == 0 or torch.device(dev).type != "cuda" or not (isinstance(H, (int, np.integer)) and isinstance(W, (int, np.integer))):
        rgbs, disps, accs, valid_idxs, bboxes = render_frames_device(
            render_poses, hwf, chunk, render_kwargs, centers=centers, kp=kp, skts=skts, cyls=cyls, bg_imgs=bg_imgs,
            bg_indices=bg_indices, cams=cams, render_factor=render_factor, white_bkgd=white_bkgd, ext_scale=ext_scale,
            frame_ids=frame_ids)
        return (rgbs.cpu().numpy(), disps.cpu().numpy(), accs.cpu().numpy() if ret_acc else [], valid_idxs, bboxes)
    if render_factor:
        H, W = int(H) // render_factor, int(W) // render_factor
    dl = FrameDownloader(n_out, int(H), int(W), ret_acc, dev)
    _, _, _, valid_idxs, bboxes = render_frames_device(
        render_poses, hwf, chunk, render_kwargs, centers=centers, kp=kp, skts=skts, cyls=cyls, bg_imgs=bg_imgs,
        bg_indices=bg_indices, cams=cams, render_factor=render_factor, white_bkgd=white_bkgd, ext_scale=ext_scale,
        frame_ids=frame_ids, frame_sink=dl.sink)
    res = dl.finish()
    return (res[0], res[1], res[2] if ret_acc else [], valid_idxs, bboxes)


class FrameDownloader:
    """Frames go to the host while the next ones render: a copy stream, one copy per frame ordered behind that
    frame's kernels by an event.  (Pageable copies of the stacked frames at the end were 3.6 ms per 512 x 512 frame.)
    `sink(k, rgb, disp, acc)` takes the k-th frame's device tensors, `finish()` returns the numpy stacks
    [rgbs [n,H,W,3], disps [n,H,W,1] (, accs [n,H,W,1])]."""

    def __init__(self, n_out: int, H: int, W: int, ret_acc: bool, dev):
        self.H, self.W, self.dev = H, W, dev
        self.chans = (3, 1, 1) if ret_acc else (3, 1)
        self.copy_stream = torch.cuda.Stream(device=dev)
        per_frame = H * W * sum(self.chans) * 4
        self.whole = n_out * per_frame <= PINNED_RESULT_BYTES
        if self.whole:
            # short paths: the results themselves are pinned (torch's caching host allocator hands the blocks back when
            # the returned numpy arrays die), device tensors kept alive until the copies have run
            self.host = [torch.empty((n_out, H, W, c), dtype=torch.float32, pin_memory=True) for c in self.chans]
            self.keep = []
        else:
            # long paths (hundreds of frames, or megapixel frames): page-locking every result would pin gigabytes that
            # the caching host allocator never returns to the OS.  A ring of PINNED_RING staging frames instead: a
            # frame is copied device -> slot on the copy stream, and moved slot -> pageable result when the slot comes
            # round again (its copy event has long completed), at which point its device tensors are dropped too.
            self.out = [np.empty((n_out, H, W, c), dtype=np.float32) for c in self.chans]
            self.ring = [[torch.empty((H, W, c), dtype=torch.float32, pin_memory=True) for c in self.chans] for _ in range(PINNED_RING)]
            self.pending = [None] * PINNED_RING          # (frame index, copy event, device tensors)

    def _views(self, rgb, disp, acc):
        H, W = self.H, self.W
        return (rgb.view(H, W, 3), disp.view(H, W, 1), acc.view(H, W, 1))

    def _drain(self, slot):
        if self.pending[slot] is None:
            return
        k, ev, _dev_tensors = self.pending[slot]
        ev.synchronize()
        for o, st in zip(self.out, self.ring[slot]):
            o[k] = st.numpy()
        self.pending[slot] = None

    def sink(self, k, rgb, disp, acc):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        if self.whole:
            self.keep.append((rgb, disp, acc))
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(ev)
                for dst, src in zip(self.host, self._views(rgb, disp, acc)):
                    dst[k].copy_(src, non_blocking=True)
            return
        slot = k % PINNED_RING
        self._drain(slot)
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(ev)
            for dst, src in zip(self.ring[slot], self._views(rgb, disp, acc)):
                dst.copy_(src, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        self.pending[slot] = (k, done, (rgb, disp, acc))

    def finish(self):
        if self.whole:
            self.copy_stream.synchronize()
            self.keep.clear()
            return [h.numpy() for h in self.host]
        for slot in range(PINNED_RING):
            self._drain(slot)
        return self.out


def _render_path_multi(r, render_poses, hwf, chunk, render_kwargs, centers, kp, skts, cyls, bg_imgs, cams,
                       render_factor, white_bkgd, ret_acc, ext_scale):
    """render_path on a caster that owns several GPUs (HipRayCaster(devices=[...])): one pg_render_frames
    call, frames or ray chunks spread over the devices inside the library (no torch.distributed)."""
    H, W, focal = hwf
    if render_factor != 0:
        H, W = H // render_factor, W // render_factor
        focal = focal / render_factor if isinstance(focal, float) else focal.copy() / render_factor
        if centers is not None:
            centers = centers / render_factor if isinstance(focal, float) else centers.copy() / render_factor
    if kp is None and cyls is None:
        raise NotImplementedError("render_path needs kp or cyls (bounding-cylinder cull)")
    if not (isinstance(H, (int, np.integer)) and isinstance(W, (int, np.integer))):
        raise ValueError("multi-device render_path needs one frame size (scalar H, W)")
    cyls, bboxes, meta = frame_boxes(r, render_poses, H, W, focal, kps=kp, cylinder_params=cyls, ext_scale=ext_scale,
                                     centers=centers)
    cyls = torch.as_tensor(cyls).detach().float().cpu()
    valid_idxs = BoxPixelIds(bboxes, [m[1] for m in meta])
    F = len(render_poses)
    n_pose = cyls.shape[0]
    sk = torch.as_tensor(skts).reshape(-1, 24, 4, 4)
    sk = torch.stack([sk[i % sk.shape[0]] for i in range(F)])
    cy = torch.stack([cyls[i % n_pose] for i in range(F)])
    cm = None
    if cams is not None:
        ct = torch.as_tensor(cams).reshape(-1).float()
        cm = torch.stack([ct[i % ct.shape[0]] for i in range(F)])
    bg = None
    if bg_imgs is not None and not white_bkgd:
        import torch.nn.functional as Fn
        bgi = torch.tensor(bg_imgs[0])
        bg = Fn.interpolate(bgi.permute(2, 0, 1)[None].float(), size=(int(H), int(W)), mode="bilinear",
                            align_corners=False)[0].permute(1, 2, 0).reshape(int(H) * int(W), 3)
    r.set_chunk(int(chunk))
    kw = render_kwargs
    focals = [m[2] for m in meta]
    rgbs, disps, accs = r.render_frames(int(H), int(W), focals, [m[3] for m in meta], bboxes, sk, cy,
                                        centers=None if centers is None else [m[4] for m in meta], cams=cm,
                                        n_samples=kw.get("N_samples"), n_importance=kw.get("N_importance"),
                                        lindisp=bool(kw.get("lindisp", False)), bg=bg, base_bg=1.0 if white_bkgd else 0.0)
    return rgbs, disps, accs if ret_acc else [], valid_idxs, bboxes
