"""Host-side SMPL skeleton geometry feeding the HIP renderer.

Mirrors the reference functions the render path calls on the host
(float64 numpy, exactly like the reference):

* forward kinematics   -- run_gan.py:2211-2257 (= core/utils/skeleton_utils.py:334-376)
* `load_retarget`      -- run_gan.py:437-451
* bounding cylinder    -- core/utils/skeleton_utils.py:635-685 (constants ray_utils.py:89-104)
* cylinder -> 2-D box  -- core/utils/skeleton_utils.py:700-787
* camera conventions   -- skeleton_utils.py:529-530, 1401-1410, 1423-1431
"""
from __future__ import annotations

from collections import namedtuple

import numpy as np

Skeleton = namedtuple("Skeleton", ["joint_names", "joint_trees", "root_id"])

SMPLSkeleton = Skeleton(
    joint_names=[
        "pelvis", "left_hip", "right_hip", "spine1", "left_knee", "right_knee", "spine2",
        "left_ankle", "right_ankle", "spine3", "left_foot", "right_foot", "neck",
        "left_collar", "right_collar", "head", "left_shoulder", "right_shoulder",
        "left_elbow", "right_elbow", "left_wrist", "right_wrist", "left_hand", "right_hand"],
    joint_trees=np.array([0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12,
                          13, 14, 16, 17, 18, 19, 20, 21]),
    root_id=0,
)

# SMPL rest pose in (x, y, z), metres*~2.1 (skeleton_utils.py:259-282) -- data table
smpl_rest_pose = np.array([
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

# SURREAL rest pose scale: load_surreal.py:18 (dataset_ext_scale)
SURREAL_REST_SCALE = 0.714


def rotvec_to_matrix(rotvec) -> np.ndarray:
    """Axis-angle [...,3] -> rotation matrices [...,3,3], float64.

    Numerically the scipy `Rotation.from_rotvec(p).as_matrix()` map used by the
    reference (run_gan.py:2228): rotvec -> unit quaternion -> matrix.
    """
    rv = np.asarray(rotvec, dtype=np.float64)
    theta = np.linalg.norm(rv, axis=-1)
    t2 = theta * theta
    small = theta < 1e-3
    safe = np.where(small, 1.0, theta)
    k = np.where(small, 0.5 - t2 / 48.0 + t2 * t2 / 3840.0, np.sin(0.5 * theta) / safe)
    qx, qy, qz = rv[..., 0] * k, rv[..., 1] * k, rv[..., 2] * k
    qw = np.cos(0.5 * theta)
    nrm = np.sqrt(qx * qx + qy * qy + qz * qz + qw * qw)
    qx, qy, qz, qw = qx / nrm, qy / nrm, qz / nrm, qw / nrm
    xx, yy, zz, ww = qx * qx, qy * qy, qz * qz, qw * qw
    xy, zw, xz, yw, yz, xw = qx * qy, qz * qw, qx * qz, qy * qw, qy * qz, qx * qw
    m = np.empty(rv.shape[:-1] + (3, 3), dtype=np.float64)
    m[..., 0, 0] = xx - yy - zz + ww
    m[..., 0, 1] = 2.0 * (xy - zw)
    m[..., 0, 2] = 2.0 * (xz + yw)
    m[..., 1, 0] = 2.0 * (xy + zw)
    m[..., 1, 1] = -xx + yy - zz + ww
    m[..., 1, 2] = 2.0 * (yz - xw)
    m[..., 2, 0] = 2.0 * (xz - yw)
    m[..., 2, 1] = 2.0 * (yz + xw)
    m[..., 2, 2] = -xx - yy + zz + ww
    return m


def get_smpl_l2ws(pose, rest_pose=None, scale=1., skel_type=SMPLSkeleton) -> np.ndarray:
    """Per-joint local-to-world transforms [24,4,4] of one pose (axis-angle [24,3]).

    Chain: l2w[0] = [R0 | rest0], l2w[j] = l2w[parent] @ [Rj | rest_j - rest_parent].
    """
    if rest_pose is None:
        rest_pose = smpl_rest_pose
    rest = np.asarray(rest_pose) * scale
    parents = skel_type.joint_trees
    rots = rotvec_to_matrix(pose)
    n = rest.shape[0]
    # relative transforms, batched; chain product walks the tree once
    rel = np.zeros((n, 4, 4), dtype=np.float64)
    rel[:, :3, :3] = rots
    rel[:, 3, 3] = 1.0
    rel[0, :3, 3] = rest[0]
    rel[1:, :3, 3] = rest[1:] - rest[parents[1:]]
    l2ws = np.empty_like(rel)
    l2ws[0] = rel[0]
    for j in range(1, n):
        l2ws[j] = l2ws[parents[j]] @ rel[j]
    return l2ws


def bones_to_pose(bones, rest_pose):
    """bones [F,24,3] -> (kps [F,24,3], skts [F,24,4,4]) like `load_retarget`."""
    l2ws = np.stack([get_smpl_l2ws(b, rest_pose, 1.0) for b in np.asarray(bones)])
    return l2ws[..., :3, -1], np.linalg.inv(l2ws), l2ws


def swap_mat(mat):
    """Negate the y and z camera axes (columns 1, 2): NeRF <-> OpenCV convention."""
    out = np.array(mat, copy=True)
    out[..., 1] *= -1
    out[..., 2] *= -1
    return out


def nerf_c2w_to_extrinsic(c2w):
    return np.linalg.inv(swap_mat(c2w))


def nerf_extrinsic_to_c2w(ext):
    return swap_mat(np.linalg.inv(ext))


def focal_to_intrinsic_np(focal):
    f = np.asarray(focal, dtype=np.float64).reshape(-1)
    fx, fy = (f[0], f[0]) if f.size < 2 else (f[0], f[1])
    return np.array([[fx, 0, 0, 0], [0, fy, 0, 0], [0, 0, 1, 0]], dtype=np.float32)


def get_kp_bounding_cylinder(kp, ext_scale=0.001, extend_mm=250., top_expand_ratio=1.60,
                             bot_expand_ratio=1.10, head="-y", skel_type=SMPLSkeleton):
    """Cylinder (cx, cz, radius, top, bot) around each pose; kp [F,24,3].

    Ground plane / height axis from `head` ('-y': ground x-z, up = -y).
    Defaults are the constants `kp_to_valid_rays` passes (ray_utils.py:89-104).
    """
    kp = np.asarray(kp)
    if kp.ndim == 2:
        kp = kp[None]
    if head.endswith("z"):
        g_axes, h_axis = [0, 1], 2
    elif head.endswith("y"):
        g_axes, h_axis = [0, 2], 1
    else:
        raise NotImplementedError(f"Head orientation {head} is not implemented!")
    flip = -1 if head.startswith("-") else 1
    root = kp[:, skel_type.root_id, :]
    reach = np.linalg.norm(kp[..., g_axes] - root[:, None, g_axes], axis=-1).max(-1)
    height = flip * kp[..., h_axis]
    extension = extend_mm * ext_scale
    radius = reach + extension
    top = flip * (height.max(-1) + extension * top_expand_ratio)
    bot = flip * (height.min(-1) - extension * bot_expand_ratio)
    return np.stack([root[:, g_axes[0]], root[:, g_axes[1]], radius, top, bot], axis=-1)


def cylinder_to_box_2d(cylinder_params, hwf, w2c=None, center=None):
    """Integer image box (tl, br) covering the two projected cylinder caps.

    50 points per cap, projected with w2c and the pinhole intrinsic, floor/ceil,
    shifted by the principal point and clipped to the frame.
    """
    H, W, focal = hwf
    cyl = np.asarray(cylinder_params).reshape(-1)
    phi = np.linspace(0., 2 * np.pi, 50)
    ring_x = cyl[0] + np.cos(phi) * cyl[2]
    ring_z = cyl[1] + np.sin(phi) * cyl[2]
    ones = np.ones_like(ring_x)
    pts = np.concatenate([np.stack([ring_x, cyl[3] * ones, ring_z, ones], axis=-1),
                          np.stack([ring_x, cyl[4] * ones, ring_z, ones], axis=-1)], axis=0)
    if w2c is not None:
        pts = pts @ w2c.T
    pts = pts @ focal_to_intrinsic_np(focal).T
    pts_2d = pts[:, :2] / pts[:, 2:3]
    lo = np.floor(pts_2d.min(0)).astype(np.int32)
    hi = np.ceil(pts_2d.max(0)).astype(np.int32)
    if center is None:
        off = np.array([int(W * .5), int(H * .5)], dtype=np.int32)
    else:
        off = np.array([int(center[0]), int(center[1])], dtype=np.int32)
    tl, br = lo + off, hi + off
    lim = np.array([W - 1, H - 1], dtype=np.int32)
    tl = np.clip(tl, 0, lim).astype(np.int32)
    br = np.clip(br, 0, lim).astype(np.int32)
    return tl, br, pts_2d
