"""Multi-GPU rendering: one process per GPU, the frames' nanmean groups partitioned over the
ranks, one all-gather to reassemble (RCCL over xGMI via torch.distributed backend "nccl"; the
same code runs on gloo for CPU tests).

Replaces the reference's only parallelism, `nn.DataParallel(RayCaster)`
(core/raycasters.py:157, run_gan.py:162): no per-forward weight broadcast (every rank
loads the 7 MB of weights once), no scatter of replicated pose tensors, and 20 B/ray of
result traffic instead of 596 B/ray.  There is no collective on the data path: the one
all-gather carries finished (rgb, disp, acc) maps.

Work plan (`plan_tasks`, the same algorithm as the library's pg_plan_frames): the unit is a
nanmean group -- `chunk` consecutive rays of a frame's box.  Frames go to ranks
whole while they fit under the per-rank target load; the frames that do not -- the tail of a
batch whose size is not a multiple of the world size (the GAN loop renders 20 frames per call,
run_gan.py:2042-2047: 3:2 loads on 8 GPUs otherwise), or every frame when there are fewer frames
than ranks -- are cut on group boundaries, so every value equals the single-device render's.
"""
from __future__ import annotations

from typing import List, NamedTuple, Sequence

import numpy as np
import torch


class Task(NamedTuple):
    frame: int
    r0: int          # first ray of the box's row-major ray list
    r1: int          # one past the last
    worker: int      # rank that renders it
    owner: int       # worker of the frame's first run (composes the frame in the in-process path)


def plan_tasks(n_rays_per_frame: Sequence[int], world: int, chunk: int) -> List[Task]:
    """Tasks covering every ray of every frame exactly once, cuts on multiples of `chunk`, loads
    within about one group of total / world.  Integer arithmetic only, deterministic, identical to
    pg_plan_frames (csrc/pg_api.hip; tests compare the two)."""
    n = [max(int(x), 0) for x in n_rays_per_frame]
    F = len(n)
    tasks: List[Task] = []
    if F == 0:
        return tasks
    order = sorted(range(F), key=lambda f: (-n[f], f))
    target = (sum(n) + world - 1) // world
    load = [0] * world
    least = lambda: min(range(world), key=lambda k: (load[k], k))
    tail = []
    for f in order:
        w = least()
        if n[f] <= chunk or load[w] + n[f] <= target + target // 50:
            load[w] += n[f]
            tasks.append(Task(f, 0, n[f], w, w))
        else:
            tail.append(f)
    for f in tail:
        groups = (n[f] + chunk - 1) // chunk
        g, owner = 0, -1
        while g < groups:
            w = least()
            cap = target - load[w]
            take = (cap + chunk // 2) // chunk if cap > 0 else 0
            take = min(max(take, 1), groups - g)
            rest = groups - g - take
            if 0 < rest and rest * chunk <= max(chunk, target // 32):
                take += rest                      # no sliver of a run for yet another worker
            r0, r1 = g * chunk, min((g + take) * chunk, n[f])
            if owner < 0:
                owner = w
            if tasks and tasks[-1].frame == f and tasks[-1].worker == w and tasks[-1].r1 == r0:
                tasks[-1] = tasks[-1]._replace(r1=r1)
            else:
                tasks.append(Task(f, r0, r1, w, owner))
            load[w] += r1 - r0
            g += take
    return tasks


def partition_frames(n_rays_per_frame: Sequence[int], world: int) -> List[List[int]]:
    """Whole frames to ranks by longest-processing-time on the ray count (no cuts).  Kept for callers
    that must not split a frame; `render_path_distributed` uses `plan_tasks`."""
    order = sorted(range(len(n_rays_per_frame)), key=lambda f: (-int(n_rays_per_frame[f]), f))
    load = [0] * world
    parts: List[List[int]] = [[] for _ in range(world)]
    for f in order:
        r = min(range(world), key=lambda k: (load[k], k))
        parts[r].append(f)
        load[r] += int(n_rays_per_frame[f])
    return [sorted(p) for p in parts]


def gather_frames(local: torch.Tensor, frame_ids: Sequence[int], parts: List[List[int]], n_frames: int,
                  group=None) -> torch.Tensor:
    """All-gather the per-rank frame stacks [f_local, H, W, C] into [n_frames, H, W, C].
    Ranks own different numbers of frames: stacks are padded to the largest share."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    fmax = max(len(p) for p in parts)
    shape = local.shape[1:]
    pad = torch.zeros((fmax,) + tuple(shape), dtype=local.dtype, device=local.device)
    if len(frame_ids) > 0:
        pad[:len(frame_ids)] = local
    out = torch.empty((world * fmax,) + tuple(shape), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    out = out.view(world, fmax, *shape)
    full = torch.zeros((n_frames,) + tuple(shape), dtype=local.dtype, device=local.device)
    for r, p in enumerate(parts):
        if p:
            full[torch.as_tensor(p, device=local.device)] = out[r, :len(p)]
    return full


def _world(group):
    """(world, rank, dist or None): a process that never initialised torch.distributed is world 1."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group), dist
    return 1, 0, None


@torch.no_grad()
def render_frames_distributed(render_poses, hwf, chunk, render_kwargs, group=None, centers=None, kp=None, skts=None,
                              cyls=None, bg_imgs=None, bg_indices=None, cams=None, render_factor=0, white_bkgd=False,
                              ext_scale=0.00035, frame_sink=None, stats=None):
    """The frames of `render_path` rendered by all ranks of the group and assembled on EVERY rank, left on
    the device: (rgbs [F,H,W,3], disps [F,H,W,1], accs [F,H,W,1], valid_idxs, bboxes).

    Every rank renders the ray ranges `plan_tasks` gives it (pg_render_frame_range) straight into its slice of one
    flat buffer, ONE all-gather (`all_gather_into_tensor`, fed from device memory: 20 B per ray of a box, not per
    pixel of a frame) makes every rank's pieces known to all, and each rank composes the frames over the background
    (pg_compose_frame).  A rank without work still takes part in the collective.

    Host work that every rank repeats is kept off the clock of the GPUs: the boxes come from the device
    (`rays.frame_boxes`: one 16-byte-per-frame copy back), the pixel ids of `valid_idxs` are built only if the
    caller reads them, poses and cylinders are uploaded once, and nothing between the first and the last launch
    waits for the device.  `frame_sink(i, rgb, disp, acc)`: called with each composed frame instead of stacking
    them (render_path_distributed starts the device-to-host copy there; the three stacks are then None).
    `stats` (dict): receives `host_pre_launch_ms` (call entry -> first render launch: the part no GPU overlaps)
    and `host_ms` (whole call, host side)."""
    import time
    from .rays import BoxPixelIds, frame_boxes
    from .render import _caster_device, _pick
    t_enter = time.perf_counter()
    world, rank, dist = _world(group)
    H, W, focal = hwf
    if not (isinstance(H, (int, np.integer)) and isinstance(W, (int, np.integer))):
        raise ValueError("render_path_distributed needs one frame size (scalar H, W) for the gather")
    H, W = int(H), int(W)
    if render_factor:
        H, W = H // render_factor, W // render_factor
        focal = focal / render_factor if isinstance(focal, float) else focal.copy() / render_factor
        if centers is not None:
            centers = centers / render_factor if isinstance(focal, float) else centers.copy() / render_factor
    if kp is None and cyls is None:
        raise NotImplementedError("render_path needs kp or cyls (bounding-cylinder cull)")
    r, dev = _caster_device(render_kwargs["ray_caster"])
    cyls, bboxes, meta = frame_boxes(r, render_poses, H, W, focal, kps=kp, cylinder_params=cyls, ext_scale=ext_scale,
                                     centers=centers)
    valid_idxs = BoxPixelIds(bboxes, [m[1] for m in meta])
    n_box = valid_idxs.counts()
    F = len(meta)
    tasks = plan_tasks(n_box, world, int(chunk))
    per_rank = [sum(t.r1 - t.r0 for t in tasks if t.worker == k) for k in range(world)]
    L = max(max(per_rank), 1)                       # rays per rank in the gathered buffer (padded to the largest share)
    kw = render_kwargs
    r.set_chunk(int(chunk))
    # poses, cylinders and frame-code indices once: a per-task host-to-device copy (or a .item() of a device tensor)
    # would be a blocking call between two launches
    if torch.is_tensor(skts):
        skts = skts.to(dev, dtype=torch.float32)
    cyls = torch.as_tensor(cyls).to(dev, dtype=torch.float32)
    cam_of = None
    if cams is not None:
        cam_h = torch.as_tensor(cams).detach().reshape(-1).float().cpu().tolist()
        cam_of = lambda i: float(cam_h[i % len(cam_h)]) if len(cam_h) > 1 else float(cam_h[0])
    local = torch.zeros(5 * L, device=dev)
    off = 0
    t_first = None
    for t in tasks:
        if t.worker != rank or t.r1 == t.r0:
            continue
        i = t.frame
        h, w, f, c2w_np, center = meta[i]
        n = t.r1 - t.r0
        dst = local[5 * off:5 * (off + n)]
        if t_first is None:
            t_first = time.perf_counter()
        piece = r.render_frame_range(
            h, w, f, c2w_np, bboxes[i], _pick(skts, i), _pick(cyls, i), t.r0, t.r1, center=center,
            cam=None if cam_of is None else cam_of(i), n_samples=kw.get("N_samples"), n_importance=kw.get("N_importance"),
            lindisp=bool(kw.get("lindisp", False)), out=dst)
        if piece.data_ptr() != dst.data_ptr():       # (a renderer that does not write in place)
            dst.copy_(piece.reshape(-1))
        off += n
    if t_first is None:
        t_first = time.perf_counter()
    if dist is not None:                            # also at world size 1: the collective is the path being run
        gathered = torch.empty(world * 5 * L, device=dev)
        dist.all_gather_into_tensor(gathered, local, group=group)
        gathered = gathered.view(world, 5 * L)
    else:
        gathered = local.view(1, 5 * L)
    # every rank: the pieces of each frame (in ray order) -> maps of the whole box -> frame over the background
    offs = [0] * world
    pieces_of = [[] for _ in range(F)]
    for t in tasks:
        pieces_of[t.frame].append((t.r0, t.worker, offs[t.worker], t.r1 - t.r0))
        offs[t.worker] += t.r1 - t.r0
    rgbs, disps, accs = [], [], []
    for i in range(F):
        h, w, f, c2w_np, center = meta[i]
        rm, dm, am = [], [], []
        for _, k, o, n in sorted(pieces_of[i]):
            blk = gathered[k, 5 * o:5 * (o + n)]
            rm.append(blk[:3 * n].view(n, 3)); dm.append(blk[3 * n:4 * n]); am.append(blk[4 * n:])
        cat = lambda xs, shape: (xs[0] if len(xs) == 1 else torch.cat(xs)) if xs else torch.zeros(shape, device=dev)
        bg = None
        if bg_imgs is not None and not white_bkgd:
            import torch.nn.functional as Fn
            bgi = torch.tensor(bg_imgs[bg_indices[i]] if bg_indices is not None else bg_imgs[0])
            bg = Fn.interpolate(bgi.permute(2, 0, 1)[None].float(), size=(h, w), mode="bilinear",
                                align_corners=False)[0].permute(1, 2, 0).reshape(h * w, 3).to(dev)
        rgb, disp, acc = r.compose_frame(h, w, bboxes[i], cat(rm, (0, 3)), cat(dm, (0,)), cat(am, (0,)), bg=bg,
                                         base_bg=1.0 if white_bkgd else 0.0)
        if frame_sink is not None:
            frame_sink(i, rgb, torch.nan_to_num(disp, nan=0.0, posinf=float("inf"), neginf=float("-inf")), acc)   # run_nerf.py:142-143
            continue
        rgbs.append(rgb); disps.append(disp); accs.append(acc)
    if stats is not None:
        now = time.perf_counter()
        stats["host_pre_launch_ms"] = (t_first - t_enter) * 1e3
        stats["host_ms"] = (now - t_enter) * 1e3
    if frame_sink is not None:
        return None, None, None, valid_idxs, bboxes
    e = lambda c: torch.zeros((0, H, W, c), device=dev)
    rgbs, disps, accs = (torch.stack(rgbs), torch.stack(disps), torch.stack(accs)) if F else (e(3), e(1), e(1))
    disps = torch.nan_to_num(disps, nan=0.0, posinf=float("inf"), neginf=float("-inf"))   # run_nerf.py:142-143
    return rgbs, disps, accs, valid_idxs, bboxes


def render_path_distributed(render_poses, hwf, chunk, render_kwargs, group=None, **kw):
    """`render_path` over all ranks of the process group: every rank renders its share of the frames' nanmean
    groups and every rank returns all frames (numpy, like run_nerf.render_path).  Signature of
    render.render_path; in a process without a process group it is the single-device render.  The composed
    frames go to the host through page-locked buffers on a copy stream while the next ones are composed
    (render.FrameDownloader), like the single-device render_path."""
    from .render import FrameDownloader, _caster_device
    keep = ("centers", "kp", "skts", "cyls", "bg_imgs", "bg_indices", "cams", "render_factor", "white_bkgd", "ext_scale")
    args = {k: v for k, v in kw.items() if k in keep}
    ret_acc = kw.get("ret_acc", True)
    _, dev = _caster_device(render_kwargs["ray_caster"])
    H, W = hwf[0], hwf[1]
    scalar = isinstance(H, (int, np.integer)) and isinstance(W, (int, np.integer))
    if len(render_poses) == 0 or torch.device(dev).type != "cuda" or not scalar:
        rgbs, disps, accs, valid_idxs, bboxes = render_frames_distributed(render_poses, hwf, chunk, render_kwargs, group=group, **args)
        packed = torch.cat([rgbs, disps, accs], -1).float().cpu().numpy()
        return packed[..., 0:3], packed[..., 3:4], (packed[..., 4:5] if ret_acc else []), valid_idxs, bboxes
    rf = kw.get("render_factor", 0)
    H, W = (int(H) // rf, int(W) // rf) if rf else (int(H), int(W))
    dl = FrameDownloader(len(render_poses), H, W, ret_acc, dev)
    _, _, _, valid_idxs, bboxes = render_frames_distributed(render_poses, hwf, chunk, render_kwargs, group=group,
                                                            frame_sink=dl.sink, **args)
    res = dl.finish()
    return res[0], res[1], (res[2] if ret_acc else []), valid_idxs, bboxes
