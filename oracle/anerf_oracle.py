"""fp32 CPU restatement of PoseGen's A-NeRF volumetric rendering hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parity is PINNED: every stage
below is checked against golden vectors captured from the real reference
(`tools/gen_golden.py`, fixtures in `tests/golden/`).

All `file:line` citations are relative to the upstream reference repository
(mgholamikn/PoseGen).  Arithmetic is fp32 torch-CPU unless stated; host
geometry (cylinder, 2-D box, kinematics) is float64 numpy like the reference.

Stage map (SURVEY.md section 8 row ids):
  a-18 smpl_l2ws / pose_from_bones        run_gan.py:437-451, 2211-2257
  a-2  bounding_cylinder / cylinder_box_2d / valid_rays / camera_rays
  a-6  near_far_in_cylinder               core/utils/ray_utils.py:292-344
  a-7  coarse_z                           ray_utils.py:204-251
  a-8..a-10 embed_points                  core/encoders.py, core/cutoff_embedder.py
  a-12 mlp_forward                        core/networks/nerf.py:90-148
  a-13 composite                          nerf.py:150-205
  a-14 importance_z                       ray_utils.py:157-201, 255-289
  a-5  render_rays                        core/raycasters.py:361-474
  a-3  render_chunks                      core/trainer.py:64-147
  a-1  render_path                        run_nerf.py:27-147
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

__all__ = [
    "OracleConfig", "SMPL_PARENTS", "SMPL_REST_POSE", "smpl_l2ws", "pose_from_bones",
    "bounding_cylinder", "cylinder_box_2d", "camera_rays", "valid_rays",
    "near_far_in_cylinder", "coarse_z", "embed_points", "mlp_forward", "composite",
    "importance_z", "render_rays", "render_chunks", "render_path", "c2w_to_extrinsic",
    "flops_per_point",
]

# ----------------------------------------------------------------------------
# skeleton constants (core/utils/skeleton_utils.py:83-110, 259-282) -- data
# ----------------------------------------------------------------------------
SMPL_PARENTS = np.array([0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12,
                         13, 14, 16, 17, 18, 19, 20, 21], dtype=np.int64)

SMPL_REST_POSE = np.array([
    [0.00000000e+00, 2.30003661e-09, -9.86228770e-08],
    [1.63832515e-01, -2.17391014e-01, -2.89178602e-02],
    [-1.57855421e-01, -2.14761734e-01, -2.09642015e-02],
    [-7.04505108e-03, 2.50450850e-01, -4.11837511e-02],
    [2.42021069e-01, -1.08830070e+00, -3.14962119e-02],
    [-2.47206554e-01, -1.10715497e+00, -3.06970738e-02],
    [3.95125849e-03, 5.94849110e-01, -4.03754264e-02],
    [2.12680623e-01, -1.99382353e+00, -1.29327580e-01],
    [-2.10857525e-01, -2.01218796e+00, -1.23002514e-01],
    [9.39484313e-03, 7.19204426e-01, 2.06931755e-02],
    [2.63385147e-01, -2.12222481e+00, 1.46775618e-01],
    [-2.51970559e-01, -2.12153077e+00, 1.60450473e-01],
    [3.83779174e-03, 1.22592449e+00, -9.78838727e-02],
    [1.91201791e-01, 1.00385976e+00, -6.21964522e-02],
    [-1.77145526e-01, 9.96228695e-01, -7.55542740e-02],
    [1.68482102e-02, 1.38698268e+00, 2.44048554e-02],
    [4.01985168e-01, 1.07928419e+00, -7.47655183e-02],
    [-3.98825467e-01, 1.07523870e+00, -9.96334553e-02],
    [1.00236952e+00, 1.05217218e+00, -1.35129794e-01],
    [-9.86728609e-01, 1.04515052e+00, -1.40235111e-01],
    [1.56646240e+00, 1.06961894e+00, -1.37338534e-01],
    [-1.56946480e+00, 1.05935931e+00, -1.53905824e-01],
    [1.75282109e+00, 1.04682994e+00, -1.68231070e-01],
    [-1.75758195e+00, 1.04255080e+00, -1.77773550e-01]], dtype=np.float32)


@dataclass
class OracleConfig:
    """Renderer fields of configs/surreal/surreal.txt (run_nerf.py:186-490 defaults)."""
    n_joints: int = 24
    multires: int = 7            # distance embedding frequencies
    multires_views: int = 4      # view embedding frequencies
    net_depth: int = 8
    net_width: int = 256
    skips: Tuple[int, ...] = (4,)
    framecode_ch: int = 0        # 16 with opt_framecode (h36m)
    cutoff_dist: float = 0.5     # cutoff_mm(500) * ext_scale(0.001)
    tau_v: float = 20.0          # CutoffEmbedder.tau of embed_fn
    tau_d: float = 20.0          # CutoffEmbedder.tau of embeddirs_fn
    density_scale: float = 1.0
    rgb_eps: float = 1e-3
    density_type: str = "relu"   # 'relu' | 'softplus' (get_density_fn, core/raycasters.py:230-238)
    softplus_shift: float = 1.0
    # None = the reference's fp32.  'bf16' / 'fp16' / 'bf16x3' / 'fp16x3' / 'fp16c' EMULATE the MFMA
    # operand rounding of the HIP kernel's precision modes (operands rounded to the 16-bit
    # type, or split into hi+lo halves with the lo*lo term dropped; products exact, fp32
    # accumulate) so that layout bugs can be told from rounding.
    quant: Optional[str] = None

    @property
    def ch_v(self) -> int:
        return self.n_joints * (1 + 2 * self.multires)

    @property
    def ch_r(self) -> int:
        return self.n_joints * 3

    @property
    def ch_d(self) -> int:
        return self.n_joints * 3 * (1 + 2 * self.multires_views)

    @property
    def ch_density_in(self) -> int:
        return self.ch_v + self.ch_r


def flops_per_point(cfg: OracleConfig) -> int:
    """Algorithmic FLOP of one MLP point evaluation."""
    W, din = cfg.net_width, cfg.ch_density_in
    mac = din * W
    for i in range(cfg.net_depth - 1):
        mac += (W + din if i in cfg.skips else W) * W
    mac += W                        # alpha_linear
    mac += W * W                    # feature_linear
    mac += (W + cfg.ch_d + cfg.framecode_ch) * (W // 2)
    mac += (W // 2) * 3
    return 2 * mac


# ----------------------------------------------------------------------------
# a-18: forward kinematics (float64 numpy, scipy-free Rodrigues)
# ----------------------------------------------------------------------------
def _rotvec_to_matrix(rv: np.ndarray) -> np.ndarray:
    """Rotation matrices of rotation vectors [...,3] (float64).

    Same map as scipy Rotation.from_rotvec(...).as_matrix() used at
    run_gan.py:2228 (quaternion route there; Rodrigues here, equal to ~1e-16).
    """
    rv = np.asarray(rv, dtype=np.float64)
    ang = np.linalg.norm(rv, axis=-1, keepdims=True)
    small = ang < 1e-3
    ang2 = ang * ang
    # scipy switches to a Taylor series of sin(a/2)/a below 1e-3; mirror that
    scale = np.where(small, 0.5 - ang2 / 48.0 + ang2 * ang2 / 3840.0,
                     np.sin(ang / 2.0) / np.where(small, 1.0, ang))
    q_xyz = rv * scale
    q_w = np.cos(ang / 2.0)[..., 0]
    x, y, z = q_xyz[..., 0], q_xyz[..., 1], q_xyz[..., 2]
    n = np.sqrt(x * x + y * y + z * z + q_w * q_w)
    x, y, z, w = x / n, y / n, z / n, q_w / n
    R = np.empty(rv.shape[:-1] + (3, 3), dtype=np.float64)
    R[..., 0, 0] = x * x - y * y - z * z + w * w
    R[..., 1, 0] = 2 * (x * y + z * w)
    R[..., 2, 0] = 2 * (x * z - y * w)
    R[..., 0, 1] = 2 * (x * y - z * w)
    R[..., 1, 1] = -x * x + y * y - z * z + w * w
    R[..., 2, 1] = 2 * (y * z + x * w)
    R[..., 0, 2] = 2 * (x * z + y * w)
    R[..., 1, 2] = 2 * (y * z - x * w)
    R[..., 2, 2] = -x * x - y * y + z * z + w * w
    return R


def smpl_l2ws(bones: np.ndarray, rest_pose: np.ndarray, scale: float = 1.0) -> np.ndarray:
    """Local-to-world 4x4 per joint for ONE pose (run_gan.py:2211-2257)."""
    rest = np.asarray(rest_pose) * scale
    rots = _rotvec_to_matrix(bones)
    out = np.zeros((len(SMPL_PARENTS), 4, 4), dtype=np.float64)
    for j in range(len(SMPL_PARENTS)):
        rel = np.eye(4, dtype=np.float64)
        rel[:3, :3] = rots[j]
        if j == 0:
            rel[:3, 3] = rest[0]
            out[0] = rel
        else:
            p = SMPL_PARENTS[j]
            rel[:3, 3] = rest[j] - rest[p]
            out[j] = out[p] @ rel
    return out


def pose_from_bones(bones: np.ndarray, rest_pose: np.ndarray):
    """bones [F,24,3] -> kps [F,24,3], skts [F,24,4,4] (run_gan.py:437-451)."""
    l2ws = np.stack([smpl_l2ws(b, rest_pose, 1.0) for b in bones])
    kps = l2ws[..., :3, -1]
    skts = np.linalg.inv(l2ws)
    return kps, skts, l2ws


# ----------------------------------------------------------------------------
# a-2: camera / cylinder / box / rays
# ----------------------------------------------------------------------------
def c2w_to_extrinsic(c2w: np.ndarray) -> np.ndarray:
    """inv(c2w with y,z columns negated) (skeleton_utils.py:529-530, 1401-1410)."""
    m = np.array(c2w, copy=True)
    m[..., 1] = -m[..., 1]
    m[..., 2] = -m[..., 2]
    return np.linalg.inv(m)


def bounding_cylinder(kps: np.ndarray, ext_scale: float, extend_mm: float = 250.0,
                      top_ratio: float = 1.60, bot_ratio: float = 1.10) -> np.ndarray:
    """(cx, cz, radius, top, bot) per pose, head='-y'
    (skeleton_utils.py:635-685 with the constants of ray_utils.py:89-104)."""
    kps = np.asarray(kps)
    root = kps[:, 0, :]
    gdist = np.linalg.norm(kps[..., [0, 2]] - root[:, None, [0, 2]], axis=-1)
    hgt = -kps[..., 1]
    ext = extend_mm * ext_scale
    radius = gdist.max(-1) + ext
    top = -(hgt.max(-1) + ext * top_ratio)
    bot = -(hgt.min(-1) - ext * bot_ratio)
    return np.stack([root[:, 0], root[:, 2], radius, top, bot], axis=-1)


def cylinder_box_2d(cyl: np.ndarray, H: int, W: int, focal, w2c: np.ndarray,
                    center=None):
    """Integer (tl, br) image box of the projected cylinder caps
    (skeleton_utils.py:700-787; intrinsic :1423-1431 is float32)."""
    cyl = np.asarray(cyl)
    ang = np.linspace(0.0, 2.0 * np.pi, 50)
    x = cyl[0] + np.cos(ang) * cyl[2]
    z = cyl[1] + np.sin(ang) * cyl[2]
    one = np.ones_like(x)
    caps = np.concatenate([np.stack([x, cyl[3] * one, z, one], -1),
                           np.stack([x, cyl[4] * one, z, one], -1)], 0)
    f = np.asarray(focal, dtype=np.float64).reshape(-1)
    fx, fy = (f[0], f[0]) if f.size < 2 else (f[0], f[1])
    K = np.array([[fx, 0, 0, 0], [0, fy, 0, 0], [0, 0, 1, 0]], dtype=np.float32)
    cam = caps @ w2c.T
    img = cam @ K.T
    uv = img[:, :2] / img[:, 2:3]
    tl = np.array([np.floor(uv[:, 0].min()), np.floor(uv[:, 1].min())]).astype(np.int32)
    br = np.array([np.ceil(uv[:, 0].max()), np.ceil(uv[:, 1].max())]).astype(np.int32)
    if center is None:
        off = np.array([int(W * .5), int(H * .5)], dtype=np.int32)
    else:
        off = np.array([int(center[0]), int(center[1])], dtype=np.int32)
    tl = tl + off
    br = br + off
    tl[0] = np.clip(tl[0], 0, W - 1)
    br[0] = np.clip(br[0], 0, W - 1)
    tl[1] = np.clip(tl[1], 0, H - 1)
    br[1] = np.clip(br[1], 0, H - 1)
    return tl, br


def camera_rays(H: int, W: int, focal, c2w: torch.Tensor, center=None):
    """Un-normalised pinhole rays of the full frame (ray_utils.py:6-28)."""
    f = np.asarray(focal, dtype=np.float64).reshape(-1)
    fx, fy = (float(f[0]), float(f[0])) if f.size < 2 else (float(f[0]), float(f[1]))
    cx, cy = (W * 0.5, H * 0.5) if center is None else (float(center[0]), float(center[1]))
    col = torch.arange(W, dtype=torch.float32)[None, :].expand(H, W)
    row = torch.arange(H, dtype=torch.float32)[:, None].expand(H, W)
    dirs = torch.stack([(col - cx) / fx, -(row - cy) / fy, -torch.ones(H, W)], -1)
    c2w = torch.as_tensor(c2w, dtype=torch.float32)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def valid_rays(c2ws: torch.Tensor, H: int, W: int, focals, kps: torch.Tensor,
               ext_scale: float, centers=None, cyls: Optional[np.ndarray] = None):
    """Per-frame culled rays, pixel ids, cylinders and boxes (ray_utils.py:83-136)."""
    if cyls is None:
        cyls = bounding_cylinder(kps.cpu().numpy(), ext_scale)
    cyls_t = torch.tensor(np.asarray(cyls), dtype=torch.float32)
    rays, vids, boxes = [], [], []
    for i, c2w in enumerate(c2ws):
        cyl = cyls_t[i % kps.shape[0]].numpy()
        f = focals if isinstance(focals, float) else focals[i]
        ctr = None if centers is None else centers[i]
        ro, rd = camera_rays(H, W, f, c2w, ctr)
        w2c = c2w_to_extrinsic(np.asarray(c2w, dtype=np.float32))
        tl, br = cylinder_box_2d(cyl, H, W, f, w2c, ctr)
        hh = torch.arange(int(tl[1]), int(br[1]))
        ww = torch.arange(int(tl[0]), int(br[0]))
        vid = (hh[:, None] * W + ww[None, :]).reshape(-1)
        rays.append((ro.reshape(-1, 3)[vid], rd.reshape(-1, 3)[vid]))
        vids.append(vid)
        boxes.append((tl, br))
    return rays, vids, cyls_t, boxes


# ----------------------------------------------------------------------------
# a-6 / a-7: near-far and coarse samples
# ----------------------------------------------------------------------------
def near_far_in_cylinder(rays_o, rays_d, cyl, near, far):
    """Ray / circle intersection in the x-z plane with the per-call nanmean
    patch for rays that miss (ray_utils.py:292-344).  near/far: [n,1]."""
    ax = [0, 2]
    p_near = (rays_o + rays_d * near)[..., ax]
    p_far = (rays_o + rays_d * far)[..., ax]
    radius = cyl[..., 2:3]
    to_c = cyl[..., :2] - p_near
    seg = p_far - p_near
    seg_len = torch.norm(seg, dim=-1, p=2)
    scale = torch.norm(rays_d[..., ax], dim=-1, p=2)[..., None]
    cross = to_c[..., 0] * seg[..., 1] - to_c[..., 1] * seg[..., 0]
    dist = (cross.abs() / seg_len)[..., None]
    Q = (radius.pow(2) - dist.pow(2)).pow(0.5)
    K = ((to_c * seg).sum(-1) / seg_len)[..., None]
    inside = (Q < K).float()
    new_near = near + inside * (K - Q) / scale
    new_far = near + (K + Q) / scale
    if torch.isnan(new_near).any():
        miss = torch.isnan(Q)[:, 0]
        m_near = np.nanmean(new_near.numpy())
        new_near[miss] = float(m_near) if not np.isnan(m_near) else near[miss]
        m_far = np.nanmean(new_far.numpy())
        new_far[miss] = float(m_far) if not np.isnan(m_far) else far[miss]
    return new_near, new_far


def coarse_z(near, far, n_samples: int, lindisp: bool = False, t_rand=None):
    """Depth samples (ray_utils.py:204-251): deterministic with perturb=0, else stratified
    with the uniform draws `t_rand` [n,S] the reference makes at ray_utils.py:238-244."""
    t = torch.linspace(0., 1., steps=n_samples).expand(near.shape[0], n_samples)
    if lindisp:
        z = 1. / (1. / near * (1. - t) + 1. / far * t)
    else:
        z = near * (1. - t) + far * t
    if t_rand is not None:
        mids = .5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    return z


# ----------------------------------------------------------------------------
# a-8 .. a-10: bone-relative transform + cutoff positional embedding
# ----------------------------------------------------------------------------
def _cutoff_embed(x, dists, n_freq: int, tau: float, cutoff: float, per_joint: int):
    """CutoffEmbedder with include_input & cutoff_inputs (cutoff_embedder.py:111-174).

    x [..., J*per_joint]; dists [..., J].  Output channel = row*(J*per_joint)+i
    with rows (x, sin 2^0 x, cos 2^0 x, ...), every row times
    w = 1 - sigmoid(tau (dist - cutoff)) of the owning joint.
    """
    freqs = 2. ** torch.linspace(0., n_freq - 1, steps=n_freq)
    if per_joint > 1:
        dists = dists[..., None].expand(*dists.shape, per_joint).flatten(start_dim=-2)
    xf = freqs.view(-1, 1) * x[..., None, :]
    w = 1. - torch.sigmoid((torch.tensor(tau) * (dists - cutoff))[..., None, :])
    rows = torch.stack([torch.sin(xf), torch.cos(xf)], dim=-2).flatten(start_dim=-3, end_dim=-2)
    rows = torch.cat([x[..., None, :], rows], dim=-2) * w
    return rows.flatten(start_dim=-2)


def embed_points(pts, rays_d, skts, cfg: OracleConfig, cams=None):
    """pts [n,s,3], rays_d [n,3], skts [n,J,4,4] -> x [n,s,1080(+1)]
    (encoders.py:8-37, 101-122, 172-193; raycasters.py:476-555)."""
    n, s = pts.shape[:2]
    if skts.shape[0] < n:
        skts = skts.expand(n, *skts.shape[1:])
    pts_h = torch.cat([pts, torch.ones(n, s, 1)], dim=-1)
    # (skt @ [p;1]) for every joint: [n,J,4,4] x [n,J,4,s]
    q = (skts @ pts_h.transpose(1, 2)[:, None].expand(-1, skts.shape[1], -1, -1))
    q = q.permute(0, 3, 1, 2)[..., :3].contiguous()                 # [n,s,J,3]
    dl = (skts[..., :3, :3] @ rays_d[:, None, :, None].expand(-1, skts.shape[1], -1, -1))
    dl = dl.permute(0, 3, 1, 2).contiguous()                        # [n,1,J,3]
    v = torch.norm(q, dim=-1, p=2)                                  # [n,s,J]
    r = F.normalize(q, dim=-1, p=2).flatten(start_dim=2)            # [n,s,3J]
    e = F.normalize(dl, dim=-1, p=2).flatten(start_dim=2).expand(n, s, -1)
    xv = _cutoff_embed(v, v, cfg.multires, cfg.tau_v, cfg.cutoff_dist, 1)
    xd = _cutoff_embed(e, v, cfg.multires_views, cfg.tau_d, cfg.cutoff_dist, 3)
    parts = [xv, r, xd]
    if cams is not None:
        parts.append(cams.view(-1, 1, 1).to(torch.float32).expand(n, s, 1))
    return torch.cat(parts, dim=-1)


# ----------------------------------------------------------------------------
# a-12: NeRF MLP
# ----------------------------------------------------------------------------
def _round16(t, q):
    dt = torch.bfloat16 if q.startswith("bf16") else torch.float16
    return t.to(dt).to(torch.float32)


def _linear(x, w, b, quant: Optional[str]):
    """F.linear with the operand rounding of the kernel's precision modes (see OracleConfig)."""
    if quant is None:
        return F.linear(x, w, b)
    if quant == "fp16c":        # compensated fp16 (PG_PREC_FP16C): 128 w1 x1 + w2 x2, w = W / 129, one fp32 accumulator
        s_ = 129.0
        ws = (w.double() / s_)
        w1 = ws.to(torch.float16).double()
        w2 = (w1 + s_ * (ws - w1)).to(torch.float16).float()
        x1 = x.to(torch.float16).float()
        x2 = (x1 + s_ * (x - x1)).to(torch.float16).float()
        return F.linear(x1, ((s_ - 1) * w1).float()) + F.linear(x2, w2) + (0 if b is None else b)
    xh, wh = _round16(x, quant), _round16(w, quant)
    if not quant.endswith("x3"):
        return F.linear(xh, wh, b)
    xl, wl = _round16(x - xh, quant), _round16(w - wh, quant)
    return F.linear(xh, wh, b) + F.linear(xh, wl) + F.linear(xl, wh)


def mlp_forward(x, weights: Dict[str, torch.Tensor], cfg: OracleConfig):
    """x [P, 1080(+1)] -> raw [P,4] = (rgb_raw, sigma_raw) (nerf.py:94-148)."""
    din, dv = cfg.ch_density_in, cfg.ch_d
    x_in, x_view = x[:, :din], x[:, din:din + dv]
    h = x_in
    for i in range(cfg.net_depth):
        h = F.relu(_linear(h, weights[f"pts_linears.{i}.weight"], weights[f"pts_linears.{i}.bias"], cfg.quant))
        if i in cfg.skips:
            h = torch.cat([x_in, h], -1)
    sigma = _linear(h, weights["alpha_linear.weight"], weights["alpha_linear.bias"], cfg.quant)
    feat = _linear(h, weights["feature_linear.weight"], weights["feature_linear.bias"], cfg.quant)
    if cfg.framecode_ch > 0:
        idx = x[:, din + dv]
        codes = weights["framecodes.codes.weight"]
        if idx.max() < 0:                         # embedding.py:25-26 (eval, no frame)
            code = codes.mean(0, keepdim=True).expand(x.shape[0], -1)
        else:
            code = codes[idx.long()]
        x_view = torch.cat([x_view, code], -1)
    g = F.relu(_linear(torch.cat([feat, x_view], -1),
                       weights["views_linears.0.weight"], weights["views_linears.0.bias"], cfg.quant))
    rgb = _linear(g, weights["rgb_linear.weight"], weights["rgb_linear.bias"], cfg.quant)
    return torch.cat([rgb, sigma], -1)


def _run_mlp(x, weights, cfg, netchunk=65536):
    flat = x.reshape(-1, x.shape[-1])
    out = torch.cat([mlp_forward(flat[i:i + netchunk], weights, cfg)
                     for i in range(0, flat.shape[0], netchunk)], 0)
    return out.reshape(*x.shape[:-1], 4)


# ----------------------------------------------------------------------------
# a-13: alpha compositing
# ----------------------------------------------------------------------------
def composite(raw, z, rays_d, cfg: OracleConfig, noise=None):
    """raw [n,S,4], z [n,S], rays_d [n,3] (nerf.py:150-205); `noise` [n,S] = the term the
    reference adds before the activation when raw_noise_std > 0 (nerf.py:174-184), else 0."""
    delta = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1)
    delta = delta * torch.norm(rays_d[:, None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3]) * (1 + 2 * cfg.rgb_eps) - cfg.rgb_eps
    # act_fn of raw2outputs (nerf.py:164): F.relu or F.softplus(x - shift, beta=1) (raycasters.py:230-238)
    act = F.relu if cfg.density_type == "relu" else (lambda x: F.softplus(x - cfg.softplus_shift, beta=1))
    alpha = 1. - torch.exp(-act(raw[..., 3] / cfg.density_scale + (0. if noise is None else noise)) * delta)
    trans = torch.cumprod(torch.cat([torch.ones(z.shape[0], 1), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * trans
    rgb_map = torch.sum(w[..., None] * rgb, -2)
    depth = torch.sum(w * z, -1)
    wsum = torch.sum(w, -1)
    disp = 1. / torch.max(1e-10 * torch.ones_like(depth), depth / (wsum + 1e-10))
    disp = disp * (~torch.isclose(wsum, torch.tensor(0.))).float()
    acc = torch.minimum(wsum, torch.tensor(1.))
    return {"rgb_map": rgb_map, "disp_map": disp, "acc_map": acc, "weights": w, "alpha": alpha}


# ----------------------------------------------------------------------------
# a-14: deterministic importance samples
# ----------------------------------------------------------------------------
def importance_z(z, weights, n_importance: int, u_rand=None):
    """Inverse-CDF samples over the interior coarse bins merged with the coarse depths
    (ray_utils.py:157-201, 255-289): at u=linspace(0,1,N) (det=True) or at the caller's
    uniform draws `u_rand` [n,N] (det=False, ray_utils.py:166-180)."""
    mids = .5 * (z[:, 1:] + z[:, :-1])
    pw = weights[:, 1:-1] + 1e-5
    pdf = pw / torch.sum(pw, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    if u_rand is None:
        u = torch.linspace(0., 1., steps=n_importance).expand(cdf.shape[0], n_importance).contiguous()
    else:
        u = u_rand.contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = torch.clamp(hi - 1, min=0)
    hi = torch.clamp(hi, max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
    b_lo, b_hi = torch.gather(mids, 1, lo), torch.gather(mids, 1, hi)
    den = c_hi - c_lo
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    z_new = (b_lo + (u - c_lo) / den * (b_hi - b_lo)).detach()     # ray_utils.py:285: no gradient through the samples
    z_all, order = torch.sort(torch.cat([z, z_new], -1), -1)
    return z_all, z_new, order


# ----------------------------------------------------------------------------
# a-5: one chunk
# ----------------------------------------------------------------------------
def render_rays(ray_batch, skts, cyls, cfg: OracleConfig, w_coarse, w_fine,
                n_samples: int, n_importance: int, cams=None, lindisp=False,
                return_extras: bool = False, draws: Optional[Dict[str, torch.Tensor]] = None):
    """One `RayCaster.render_rays` call (raycasters.py:361-474).

    ray_batch [n,11] = (o, d, near, far, viewdir); skts [n|1,J,4,4]; cyls [n|1,5].
    `draws` = None: eval mode (perturb = noise = 0).  Otherwise the random numbers of a
    training-mode call, any subset of: t_rand [n,S] (ray_utils.py:238-244), u_rand [n,N]
    (ray_utils.py:166-180), noise0 [n,S] / noise1 [n,S+N] (nerf.py:174-184, already scaled),
    ray_noise [n,S+N,3] (raycasters.py:660-661 rows [:S], 673-674 rows [S:], already scaled).
    """
    draws = draws or {}
    n = ray_batch.shape[0]
    o, d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    near0, far0 = ray_batch[:, 6:7], ray_batch[:, 7:8]
    if cyls.shape[0] < n:
        cyls = cyls.expand(n, -1)
    near, far = near_far_in_cylinder(o, d, cyls, near0, far0)
    z = coarse_z(near, far, n_samples, lindisp, draws.get("t_rand"))
    pts = o[:, None, :] + d[:, None, :] * z[..., None]
    rn = draws.get("ray_noise")
    if rn is not None:
        pts = pts + rn[:, :n_samples]
    x = embed_points(pts, d, skts, cfg, cams)
    raw = _run_mlp(x, w_coarse, cfg)
    out_c = composite(raw, z, d, cfg, draws.get("noise0"))
    extras = {"near": near, "far": far, "z_coarse": z, "raw_coarse": raw,
              "weights_coarse": out_c["weights"]}
    if return_extras:
        extras["x_coarse"] = x
    out = out_c
    if n_importance > 0:
        z_all, z_new, order = importance_z(z, out_c["weights"], n_importance, draws.get("u_rand"))
        pts_f = o[:, None, :] + d[:, None, :] * z_all[..., None]
        if rn is not None:
            # the reference embeds the noisy coarse and importance points separately and merges the
            # encodings by the depth sort (raycasters.py:451-460): every point keeps its own draw
            pts_f = pts_f + torch.gather(rn, 1, order[..., None].expand(-1, -1, 3))
        x_f = embed_points(pts_f, d, skts, cfg, cams)
        raw_f = _run_mlp(x_f, w_fine, cfg)
        out = composite(raw_f, z_all, d, cfg, draws.get("noise1"))
        extras.update({"z_fine": z_all, "z_new": z_new, "order": order, "raw_fine": raw_f})
    ret = {"rgb_map": out["rgb_map"], "disp_map": out["disp_map"],
           "acc_map": out["acc_map"], "alpha": out["alpha"]}
    if n_importance > 0:
        ret.update({"rgb0": out_c["rgb_map"], "disp0": out_c["disp_map"],
                    "acc0": out_c["acc_map"], "alpha0": out_c["alpha"]})
    if return_extras:
        ret["extras"] = extras
    return ret


# ----------------------------------------------------------------------------
# a-3: chunk loop; a-1: frame loop
# ----------------------------------------------------------------------------
def render_chunks(rays_o, rays_d, skts, cyls, cfg, w_coarse, w_fine, chunk: int,
                  n_samples: int, n_importance: int, cams=None, near=0., far=1.):
    """`render` + `batchify_rays` (trainer.py:64-147): consecutive `chunk`-ray
    slices, each an independent render_rays call (nanmean is per slice)."""
    rays_o = rays_o.reshape(-1, 3).float()
    rays_d = rays_d.reshape(-1, 3).float()
    viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    ones = torch.ones_like(rays_d[:, :1])
    batch = torch.cat([rays_o, rays_d, near * ones, far * ones, viewdirs], -1)
    n = batch.shape[0]
    outs: Dict[str, List[torch.Tensor]] = {}
    for i in range(0, n, chunk):
        sl = slice(i, i + chunk)
        r = render_rays(batch[sl],
                        skts[sl] if skts.shape[0] == n else skts,
                        cyls[sl] if cyls.shape[0] == n else cyls,
                        cfg, w_coarse, w_fine, n_samples, n_importance,
                        cams=None if cams is None else (cams[sl] if cams.shape[0] == n else cams.expand(batch[sl].shape[0])))
        for k, v in r.items():
            outs.setdefault(k, []).append(v)
    return {k: torch.cat(v, 0) for k, v in outs.items()}


def render_path(c2ws, H, W, focals, chunk, cfg, w_coarse, w_fine, kps, skts,
                n_samples, n_importance, ext_scale, cams=None, white_bkgd=True,
                centers=None):
    """Frame driver with bbox cull and background composite (run_nerf.py:27-147).
    Returns rgbs [F,H,W,3], disps [F,H,W,1], accs [F,H,W,1], valid ids, boxes."""
    c2ws = torch.as_tensor(c2ws, dtype=torch.float32)
    kps = torch.as_tensor(kps, dtype=torch.float32)
    skts = torch.as_tensor(skts, dtype=torch.float32)
    rays, vids, cyls, boxes = valid_rays(c2ws, H, W, focals, kps, ext_scale, centers)
    rgbs, disps, accs = [], [], []
    for i in range(c2ws.shape[0]):
        ro, rd = rays[i]
        p = i % kps.shape[0]
        rgb_img = torch.ones(H * W, 3) if white_bkgd else torch.zeros(H * W, 3)
        disp_img = torch.zeros(H * W)
        acc_img = torch.zeros(H * W)
        if ro.shape[0] > 0:
            cam = None if cams is None else torch.as_tensor(cams)[i % len(cams):i % len(cams) + 1].float()
            r = render_chunks(ro, rd, skts[p:p + 1], cyls[p:p + 1], cfg, w_coarse, w_fine,
                              chunk, n_samples, n_importance, cams=cam)
            vid = vids[i]
            bg = (1. - r["acc_map"][:, None]) * rgb_img[vid]
            rgb_img[vid] = r["rgb_map"] + bg
            disp_img[vid] = r["disp_map"]
            acc_img[vid] = r["acc_map"]
        rgbs.append(rgb_img.view(H, W, 3).numpy())
        disps.append(disp_img.view(H, W, 1).numpy())
        accs.append(acc_img.view(H, W, 1).numpy())
    rgbs, disps, accs = np.stack(rgbs), np.stack(disps), np.stack(accs)
    disps[np.isnan(disps)] = 0.
    return rgbs, disps, accs, vids, boxes
