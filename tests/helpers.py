"""Shared test helpers: fixture loading and oracle plumbing."""
import os

import numpy as np
import torch

from oracle import anerf_oracle as orc
from posegen_amd import synthetic as syn
from posegen_amd.config import RenderConfig

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False))


def cfg_from_golden(g) -> RenderConfig:
    return RenderConfig(n_samples=int(g["n_samples"]), n_importance=int(g["n_importance"]),
                        framecode_ch=int(g.get("framecode_ch", 0)),
                        n_framecodes=int(g.get("n_framecodes", 0)),
                        density_type="softplus" if int(g.get("density_softplus", 0)) else "relu",
                        softplus_shift=float(g.get("softplus_shift", 1.0)))


def oracle_cfg(cfg: RenderConfig, tau_v, tau_d) -> orc.OracleConfig:
    return orc.OracleConfig(n_joints=cfg.n_joints, multires=cfg.multires,
                            multires_views=cfg.multires_views, net_depth=cfg.net_depth,
                            net_width=cfg.net_width, skips=tuple(cfg.skips),
                            framecode_ch=cfg.framecode_ch, cutoff_dist=cfg.cutoff_dist,
                            tau_v=float(tau_v), tau_d=float(tau_d),
                            density_scale=cfg.density_scale, rgb_eps=cfg.rgb_eps,
                            density_type=cfg.density_type, softplus_shift=cfg.softplus_shift)


def torch_weights(w):
    return {k: torch.tensor(v) for k, v in w.items()}


def model_for(cfg: RenderConfig, seed: int):
    wc, wf, tv, td = syn.make_model(cfg, seed)
    return wc, wf, tv, td


def weights_digest(w):
    import hashlib
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k]).tobytes())
    return h.hexdigest()


DRAW_KEYS = ("t_rand", "u_rand", "noise0", "noise1", "ray_noise")


def golden_draws(g):
    """The training-mode draws a rays_train* fixture holds (None for an eval fixture)."""
    d = {k: torch.tensor(g[k]) for k in DRAW_KEYS if k in g}
    return d or None


def oracle_render_rays(g, cfg, extras=True):
    """Oracle on the inputs stored in a rays_* fixture."""
    wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
    assert weights_digest(wc) == str(g["digest_coarse"]), "synthetic weight recipe drifted"
    ocfg = oracle_cfg(cfg, g["tau_v"], g["tau_d"])
    cams = torch.tensor(g["cams"]) if "cams" in g else None
    return orc.render_rays(torch.tensor(g["ray_batch"]), torch.tensor(g["skts"]),
                           torch.tensor(g["cyl"]), ocfg, torch_weights(wc), torch_weights(wf),
                           cfg.n_samples, cfg.n_importance, cams=cams, return_extras=extras,
                           draws=golden_draws(g))
