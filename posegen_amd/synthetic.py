"""Seeded synthetic model / pose / camera inputs.

There are no datasets or checkpoints in the build environment, so tests,
`bench.py` and `smoke()` all draw from these generators.  The tensor names and
shapes are the reference checkpoint's (core/raycasters.py:752-766,
core/networks/nerf.py:57-88).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from .config import RenderConfig
from .skeleton import (SURREAL_REST_SCALE, bones_to_pose, nerf_extrinsic_to_c2w,
                       smpl_rest_pose)

# fixed camera extrinsic of the GAN loop (run_gan.py:2023-2028)
GAN_EXTRINSIC = np.array(
    [[-5.29919172e-01, -5.56525674e-09, 8.48048140e-01, -1.34771157e-07],
     [1.47262004e-01, 9.84807813e-01, 9.20194958e-02, 1.26640154e-08],
     [-8.35164413e-01, 1.73648166e-01, -5.21868549e-01, 4.28571429e+00],
     [0., 0., 0., 1.]], dtype=np.float32)


def layer_shapes(cfg: RenderConfig):
    """(name, out, in) of every Linear of one NeRF net, state-dict order."""
    W, din = cfg.net_width, cfg.ch_density_in
    shapes = [("pts_linears.0", W, din)]
    for i in range(cfg.net_depth - 1):
        shapes.append((f"pts_linears.{i + 1}", W, W + din if i in cfg.skips else W))
    shapes.append(("alpha_linear", 1, W))
    shapes.append(("views_linears.0", W // 2, cfg.ch_view_in))
    shapes.append(("feature_linear", W, W))
    shapes.append(("rgb_linear", 3, W // 2))
    return shapes


def _density_trunk_np(w: Dict[str, np.ndarray], x: np.ndarray, cfg: RenderConfig) -> np.ndarray:
    """Host fp32 pass through the 8 density layers -- used ONLY to calibrate the
    synthetic alpha bias below (never on the render path)."""
    h = x
    for i in range(cfg.net_depth):
        h = np.maximum(h @ w[f"pts_linears.{i}.weight"].T + w[f"pts_linears.{i}.bias"], 0)
        if i in cfg.skips:
            h = np.concatenate([x, h], -1)
    return h


def make_weights(cfg: RenderConfig, seed: int = 0, tuned: bool = True) -> Dict[str, np.ndarray]:
    """nn.Linear-style U(-1/sqrt(fan_in), 1/sqrt(fan_in)) weights.

    Plain random init is degenerate for rendering (its density ignores the body:
    opacity is 0 or 1 everywhere), so `tuned` plants a trained-like structure:
    neuron 0 of every density layer carries occ = sum_j cos(v_j) w_j (the number
    of joints within the cutoff radius), alpha_linear reads it with gain 3, and
    the alpha bias is calibrated so that empty space (all cutoff weights 0) has
    negative density.  Rendered opacity then spans (0,1): opaque torso, soft
    limbs, empty background -- and importance sampling has surfaces to find.
    """
    rng = np.random.RandomState(seed)
    w: Dict[str, np.ndarray] = {}
    for name, n_out, n_in in layer_shapes(cfg):
        bound = 1.0 / np.sqrt(n_in)
        w[f"{name}.weight"] = rng.uniform(-bound, bound, size=(n_out, n_in)).astype(np.float32)
        w[f"{name}.bias"] = rng.uniform(-bound, bound, size=(n_out,)).astype(np.float32)
    if cfg.framecode_ch > 0:
        std = np.sqrt(2.0 / (cfg.n_framecodes + cfg.framecode_ch))   # xavier_normal_
        w["framecodes.codes.weight"] = (rng.randn(cfg.n_framecodes, cfg.framecode_ch) * std).astype(np.float32)
    if not tuned:
        return w
    J, din = cfg.n_joints, cfg.ch_density_in
    gain, thresh = 3.0, 0.8
    w["pts_linears.0.weight"][0, :] = 0.0
    w["pts_linears.0.weight"][0, 2 * J:3 * J] = 1.0          # row 2 = cos(2^0 v) * w
    w["pts_linears.0.bias"][0] = 0.0
    for i in range(1, cfg.net_depth):
        wi = w[f"pts_linears.{i}.weight"]
        wi[0, :] = 0.0
        wi[0, (din if (i - 1) in cfg.skips else 0)] = 1.0    # pass neuron 0 through
        w[f"pts_linears.{i}.bias"][0] = 0.0
    w["alpha_linear.weight"] *= 20.0
    w["alpha_linear.weight"][0, 0] = gain
    w["rgb_linear.weight"] *= 20.0
    dirs = rng.randn(256, J, 3)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    empty = np.concatenate([np.zeros((256, cfg.ch_v)), dirs.reshape(256, 3 * J)], -1).astype(np.float32)
    s_empty = _density_trunk_np(w, empty, cfg) @ w["alpha_linear.weight"].T
    w["alpha_linear.bias"] = np.array([-(float(s_empty.max()) + 0.05) - gain * thresh], dtype=np.float32)
    return w


def make_model(cfg: RenderConfig, seed: int = 0):
    """(coarse weights, fine weights, tau_v, tau_d): two independent nets."""
    return make_weights(cfg, seed), make_weights(cfg, seed + 1000), 79.6, 79.6


def make_bones(n_frames: int, seed: int = 1, sigma: float = 0.2) -> np.ndarray:
    """Axis-angle poses [F,24,3]: joints ~ N(0, sigma^2), root uniform in [-pi,pi]^3."""
    rng = np.random.RandomState(seed)
    bones = rng.randn(n_frames, 24, 3) * sigma
    bones[:, 0] = rng.uniform(-np.pi, np.pi, size=(n_frames, 3))
    return bones


def make_pose(n_frames: int, seed: int = 1, sigma: float = 0.2):
    """bones, kps [F,24,3] f32, skts [F,24,4,4] f32 with the SURREAL rest pose; sigma = spread of the joint angles (0.2:
    SURVEY.md 8(d); larger values fold the limbs towards the trunk)."""
    rest = smpl_rest_pose * SURREAL_REST_SCALE
    bones = make_bones(n_frames, seed, sigma)
    kps, skts, _ = bones_to_pose(bones, rest)
    return bones.astype(np.float32), kps.astype(np.float32), skts.astype(np.float32)


def make_camera(n_frames: int, H: int = 512, W: int = 512) -> Tuple[np.ndarray, np.ndarray]:
    """(c2ws [F,4,4] f32, focals [F] f32): GAN-loop extrinsic, focal 500 px @ 512."""
    c2w = nerf_extrinsic_to_c2w(GAN_EXTRINSIC.astype(np.float64)).astype(np.float32)
    c2ws = np.repeat(c2w[None], n_frames, axis=0)
    focals = np.full((n_frames,), 500.0 * H / 512.0, dtype=np.float32)
    return c2ws, focals
