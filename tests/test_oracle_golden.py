"""Pin the CPU oracle against golden vectors captured from the real reference
(tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import anerf_oracle as orc
from posegen_amd import synthetic as syn
from tests.helpers import (cfg_from_golden, golden_draws, load_golden, model_for, oracle_cfg,
                           oracle_render_rays, torch_weights, weights_digest)

RAY_CASES = ["rays_surreal", "rays_allhit", "rays_coarse32", "rays_cfg1", "rays_h36m", "rays_softplus"]
# fp32 end to end on both sides; differences are summation-order only
TOL = dict(rtol=2e-5, atol=2e-6)


def test_kinematics_golden():
    g = load_golden("kinematics")
    kps, skts, l2ws = orc.pose_from_bones(g["bones"], g["rest_pose"])
    np.testing.assert_allclose(l2ws, g["l2ws"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(kps, g["kps"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(skts, g["skts"], rtol=0, atol=1e-10)


def test_valid_rays_golden():
    g = load_golden("valid_rays")
    H, W = int(g["H"]), int(g["W"])
    rays, vids, cyls, boxes = orc.valid_rays(torch.tensor(g["c2ws"]), H, W, g["focals"],
                                             torch.tensor(g["kps"]), 0.001)
    np.testing.assert_allclose(cyls.numpy(), g["cyls"], rtol=1e-6, atol=1e-7)
    for i in range(len(rays)):
        assert np.array_equal(np.array([boxes[i][0], boxes[i][1]]), g["boxes"][i])
        assert len(vids[i]) == int(g[f"n_valid_{i}"])
        assert np.array_equal(vids[i][:8].numpy(), g[f"vid_head_{i}"])
        assert np.array_equal(vids[i][-8:].numpy(), g[f"vid_tail_{i}"])
        np.testing.assert_array_equal(rays[i][0][:8].numpy(), g[f"rays_o_head_{i}"])
        np.testing.assert_array_equal(rays[i][1][:8].numpy(), g[f"rays_d_head_{i}"])
        np.testing.assert_array_equal(rays[i][1][-8:].numpy(), g[f"rays_d_tail_{i}"])


@pytest.mark.parametrize("name", RAY_CASES)
def test_render_rays_golden(name):
    g = load_golden(name)
    cfg = cfg_from_golden(g)
    out = oracle_render_rays(g, cfg)
    ex = out["extras"]
    # a-6 / a-7
    np.testing.assert_allclose(ex["near"].numpy(), g["near"], **TOL)
    np.testing.assert_allclose(ex["far"].numpy(), g["far"], **TOL)
    np.testing.assert_allclose(ex["z_coarse"].numpy(), g["z_coarse"], **TOL)
    # a-8..a-10: the 1080(+1)-vector on the stored points
    x = ex["x_coarse"].numpy()[g["x_pick_rays"]][:, g["x_pick_samples"]]
    np.testing.assert_allclose(x, g["x_pick"], rtol=1e-5, atol=1e-5)
    # a-12 / a-13 coarse
    np.testing.assert_allclose(ex["raw_coarse"].numpy(), g["raw_coarse"], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(ex["weights_coarse"].numpy(), g["weights_coarse"], rtol=1e-4, atol=1e-5)
    if cfg.n_importance > 0:
        np.testing.assert_allclose(ex["z_new"].numpy(), g["z_new"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ex["z_fine"].numpy(), g["z_fine"], rtol=1e-4, atol=1e-5)
        assert (ex["order"].numpy() == g["order"]).mean() > 0.999
        np.testing.assert_allclose(ex["raw_fine"].numpy(), g["raw_fine"], rtol=1e-3, atol=2e-3)
        for k in ("rgb0", "disp0", "acc0"):
            np.testing.assert_allclose(out[k].numpy(), g[k], rtol=1e-4, atol=1e-5)
    for k in ("rgb_map", "acc_map"):
        np.testing.assert_allclose(out[k].numpy(), g[k], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["disp_map"].numpy(), g["disp_map"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["alpha"].numpy(), g["alpha"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("name", ["rays_train", "rays_train_coarse"])
def test_render_rays_training_mode_golden(name):
    """Training-mode call of the reference in its own deterministic test mode (pytest=True:
    numpy draws after seed 0; the position noise recorded as drawn): perturb, raw_noise_std,
    ray_noise_std all on (rays_train) / jitter + density noise at N_importance = 0."""
    g = load_golden(name)
    cfg = cfg_from_golden(g)
    out = oracle_render_rays(g, cfg)
    assert float(g["perturb"]) > 0 and float(g["raw_noise_std"]) > 0
    keys = ["rgb_map", "acc_map", "disp_map"] + (["rgb0", "acc0", "disp0"] if cfg.n_importance > 0 else [])
    for k in keys:
        np.testing.assert_allclose(out[k].numpy(), g[k], rtol=1e-4, atol=1e-5, err_msg=k)
    np.testing.assert_allclose(out["alpha"].numpy(), g["alpha"], rtol=1e-3, atol=1e-4)
    # the draws change the result: the same inputs in eval mode are a different image
    g_eval = {k: v for k, v in g.items() if k not in ("t_rand", "u_rand", "noise0", "noise1", "ray_noise")}
    ev = oracle_render_rays(g_eval, cfg)
    assert np.abs(ev["rgb_map"].numpy() - g["rgb_map"]).max() > 1e-3


def test_miss_rays_take_chunk_nanmean():
    """rays_surreal holds cylinder misses: their near/far equal the nanmean of the hits."""
    g = load_golden("rays_surreal")
    b, cyl = g["ray_batch"], g["cyl"][0]
    o, d = b[:, [0, 2]].astype(np.float64), b[:, [3, 5]].astype(np.float64)
    c = cyl[:2] - o
    dist = np.abs(c[:, 0] * d[:, 1] - c[:, 1] * d[:, 0]) / np.linalg.norm(d, axis=-1)
    miss = dist > cyl[2] * (1 + 1e-6)
    assert 0 < miss.sum() < len(miss)
    assert np.all(g["near"][miss] == g["near"][miss][0])
    hits = ~(dist > cyl[2] * (1 - 1e-6))
    np.testing.assert_allclose(g["near"][miss][0], g["near"][hits].mean(), rtol=1e-4)


def test_frame_golden():
    g = load_golden("frame64")
    cfg = cfg_from_golden(g)
    wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
    assert weights_digest(wc) == str(g["digest_coarse"])
    ocfg = oracle_cfg(cfg, g["tau_v"], g["tau_d"])
    H, W = int(g["H"]), int(g["W"])
    rgbs, disps, accs, vids, boxes = orc.render_path(
        g["c2ws"], H, W, g["focals"], int(g["chunk"]), ocfg, torch_weights(wc), torch_weights(wf),
        g["kps"], g["skts"], cfg.n_samples, cfg.n_importance, cfg.ext_scale)
    assert [len(v) for v in vids] == list(g["n_valid"])
    assert np.array_equal(np.array([[b[0], b[1]] for b in boxes]), g["boxes"])
    np.testing.assert_allclose(rgbs, g["rgbs"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(accs, g["accs"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(disps, g["disps"], rtol=1e-4, atol=2e-5)


def test_flops_per_point_matches_survey():
    from posegen_amd.config import surreal_config, h36m_config
    assert surreal_config().flops_per_point() == 1_723_648
    assert h36m_config().flops_per_point() == 1_727_744
    assert orc.flops_per_point(orc.OracleConfig()) == 1_723_648


def _loss_of(out, target):
    """Trainer.compute_loss for the shipped surreal config (core/trainer.py:321-383): MSE of rgb + (1 - acc) * 1 against
    the target, fine + coarse (coarse_weight 1)."""
    loss = torch.mean((out["rgb_map"] + (1. - out["acc_map"])[..., None] - target) ** 2)
    if "rgb0" in out:
        loss = loss + torch.mean((out["rgb0"] + (1. - out["acc0"])[..., None] - target) ** 2)
    return loss


@pytest.mark.parametrize("name", ["train_grads", "train_grads_h36m", "train_grads_softplus", "train_grads_raw"])
def test_oracle_autograd_matches_the_reference_training_gradients(name):
    """The oracle under torch autograd against the reference's own `loss.backward()` (fixture from
    tools/gen_golden.py: Trainer-style MSE on a training-mode call with pytest=True draws): every parameter
    gradient of both nets (+ frame codes) within 1e-4 of its largest entry -- this pins the gradient oracle the
    HIP backward pass is tested against."""
    from tools.gen_golden import grad_sample_index
    g = load_golden(name)
    # train_grads_raw: the un-filtered batch -- bound from the stored sensitivity of the reference's own gradient
    tol = max(1e-4, 4.0 * float(g["grad_sensitivity"])) if name == "train_grads_raw" else 1e-4
    cfg = cfg_from_golden(g)
    wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
    tw = lambda w: {k: torch.tensor(v, requires_grad=True) for k, v in w.items()}
    twc, twf = tw(wc), tw(wf)
    ocfg = oracle_cfg(cfg, g["tau_v"], g["tau_d"])
    cams = torch.tensor(g["cams"]) if "cams" in g else None
    out = orc.render_rays(torch.tensor(g["ray_batch"]), torch.tensor(g["skts"]), torch.tensor(g["cyl"]), ocfg, twc, twf,
                          cfg.n_samples, cfg.n_importance, cams=cams, draws=golden_draws(g))
    loss = _loss_of(out, torch.tensor(g["target"]))
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    n_checked = 0
    for tag, w in (("coarse", twc), ("fine", twf)):
        for k, p in w.items():
            key = k if k != "framecodes.codes.weight" else "framecodes.codes.weight"
            ref_vals, ref_norm = g[f"gval_{tag}_{key}"], float(g[f"gnorm_{tag}_{key}"])
            got = p.grad.numpy().reshape(-1)
            scale = max(float(np.abs(ref_vals).max()), ref_norm / np.sqrt(got.size), 1e-12)
            err = float(np.abs(got[grad_sample_index(got.size)] - ref_vals).max())
            assert err <= tol * scale + 1e-9, (tag, k, err, scale)
            assert abs(float(np.linalg.norm(got.astype(np.float64))) - ref_norm) <= tol * ref_norm + 1e-9, (tag, k)
            n_checked += 1
    assert n_checked == (50 if cfg.framecode_ch else 48)
