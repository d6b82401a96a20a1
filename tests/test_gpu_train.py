"""GPU tests of the training step: `TrainableRayCaster` -- forward in training mode with a
tape, loss on the host side of the ABI, `loss.backward()` through pg_train_backward -- against the reference's own
autograd (fixtures tests/golden/train_grads*.npz, produced by tools/gen_golden.py from the imported reference)."""
import numpy as np
import pytest
import torch

from tests.helpers import cfg_from_golden, golden_draws, load_golden, model_for
from tools.gen_golden import grad_sample_index

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _loss_of(out, target):
    """Trainer.compute_loss for the shipped surreal config (core/trainer.py:321-383)."""
    loss = torch.mean((out["rgb_map"] + (1. - out["acc_map"])[..., None] - target) ** 2)
    if "rgb0" in out:
        loss = loss + torch.mean((out["rgb0"] + (1. - out["acc0"])[..., None] - target) ** 2)
    return loss


def _trainable(g):
    from posegen_amd.raycaster import HipRayCaster
    from posegen_amd.train import TrainableRayCaster
    cfg = cfg_from_golden(g)
    wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
    c = HipRayCaster.from_weights(cfg, wc, wf, float(g["tau_v"]), float(g["tau_d"]), device=DEV, precision="fp32")
    return cfg, TrainableRayCaster(c)


@pytest.mark.parametrize("name", ["train_grads", "train_grads_h36m"])
def test_training_step_gradients_match_the_reference_autograd(name):
    """One training step of the reference on the HIP path: same draws (pytest=True), same loss; the loss, the four
    maps it reads and the gradient of every parameter tensor of both nets (24 each, + the frame codes) within 1e-4 of
    the tensor's largest entry / of its norm."""
    g = load_golden(name)
    cfg, m = _trainable(g)
    m.train()
    cams = torch.tensor(g["cams"]) if "cams" in g else None
    out = m(torch.tensor(g["ray_batch"]), N_samples=cfg.n_samples, skts=torch.tensor(g["skts"]), cyls=torch.tensor(g["cyl"]),
            cams=cams, N_importance=cfg.n_importance, draws=golden_draws(g))
    for k in ("rgb_map", "acc_map", "rgb0", "acc0"):
        assert float(np.abs(out[k].detach().cpu().numpy() - g[k]).max()) <= 2e-5, k
    loss = _loss_of(out, torch.tensor(g["target"], device=DEV))
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    n_checked = 0
    for tag, net in (("coarse", m.network), ("fine", m.network_fine)):
        for key, p in net.items():
            k = key.replace("__", ".")
            ref_vals, ref_norm = g[f"gval_{tag}_{k}"], float(g[f"gnorm_{tag}_{k}"])
            assert p.grad is not None, (tag, k)
            got = p.grad.detach().cpu().numpy().reshape(-1)
            scale = max(float(np.abs(ref_vals).max()), ref_norm / np.sqrt(got.size), 1e-12)
            err = float(np.abs(got[grad_sample_index(got.size)] - ref_vals).max())
            nerr = abs(float(np.linalg.norm(got.astype(np.float64))) - ref_norm)
            assert err <= 1e-4 * scale + 1e-9, (tag, k, err, scale)
            assert nerr <= 1e-4 * ref_norm + 1e-9, (tag, k, nerr, ref_norm)
            n_checked += 1
    assert n_checked == (50 if cfg.framecode_ch else 48)
    m.renderer.close()


def test_training_gradients_of_an_odd_batch_match_the_oracle_autograd():
    """A batch whose point count is not a multiple of 4 (25 rays x 33 + 7 samples) takes the small-tile GEMM with the
    ReLU mask and the bias sums as kernels of their own instead of fused into the large-tile GEMM: same gradients,
    checked against the oracle under torch autograd (itself pinned to the reference's gradients on the fixtures)."""
    from oracle import anerf_oracle as orc
    from posegen_amd import surreal_config
    from posegen_amd.raycaster import HipRayCaster, make_training_draws
    from posegen_amd.train import TrainableRayCaster
    from tests.helpers import oracle_cfg
    g = load_golden("train_grads")
    cfg = surreal_config(n_samples=33, n_importance=7)
    wc, wf, tv, td = model_for(cfg, 4)
    n = 25
    rb, sk, cy = torch.tensor(g["ray_batch"][:n]), torch.tensor(g["skts"]), torch.tensor(g["cyl"])
    target = torch.tensor(g["target"][:n])
    draws = make_training_draws(n, 33, 7, perturb=1., raw_noise_std=1., pytest=True)
    tw = lambda w: {k: torch.tensor(v, requires_grad=True) for k, v in w.items()}
    twc, twf = tw(wc), tw(wf)
    ref = orc.render_rays(rb, sk, cy, oracle_cfg(cfg, tv, td), twc, twf, 33, 7, draws=draws)
    _loss_of(ref, target).backward()
    c = HipRayCaster.from_weights(cfg, wc, wf, float(tv), float(td), device=DEV, precision="fp32")
    m = TrainableRayCaster(c)
    m.train()
    out = m(rb, N_samples=33, skts=sk, cyls=cy, N_importance=7, draws={k: v.to(DEV) for k, v in draws.items()})
    _loss_of(out, target.to(DEV)).backward()
    for tag, net, refw in (("coarse", m.network, twc), ("fine", m.network_fine, twf)):
        for key, p in net.items():
            k = key.replace("__", ".")
            r = refw[k].grad.numpy().reshape(-1)
            got = p.grad.detach().cpu().numpy().reshape(-1)
            scale = max(float(np.abs(r).max()), float(np.linalg.norm(r)) / np.sqrt(r.size), 1e-12)
            assert float(np.abs(got - r).max()) <= 2e-4 * scale + 1e-9, (tag, k, float(np.abs(got - r).max()), scale)
    m.renderer.close()


def test_training_loop_lowers_the_loss_and_syncs_the_inference_kernels():
    """`get_grad_vars` + Adam as in the reference (core/raycasters.py:186-228, lrate 5e-4): a few steps on one batch
    lower the loss; after sync_inference_weights() the fused inference kernels render with the trained weights
    (eval-mode fp32 render == the training-mode forward without noise, 1e-4)."""
    g = load_golden("train_grads")
    cfg, m = _trainable(g)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=5e-4, betas=(0.9, 0.999))
    rb, sk, cy = torch.tensor(g["ray_batch"]), torch.tensor(g["skts"]), torch.tensor(g["cyl"])
    target = torch.tensor(g["target"], device=DEV)
    losses = []
    for it in range(6):
        opt.zero_grad()
        out = m(rb, N_samples=cfg.n_samples, skts=sk, cyls=cy, N_importance=cfg.n_importance, perturb=1., raw_noise_std=0.)
        loss = _loss_of(out, target)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses
    m.sync_inference_weights()
    m.eval()
    with torch.no_grad():
        ev = m(rb, N_samples=cfg.n_samples, skts=sk, cyls=cy, N_importance=cfg.n_importance)
    m.train()
    tr = m(rb, N_samples=cfg.n_samples, skts=sk, cyls=cy, N_importance=cfg.n_importance)       # no draws: the eval arithmetic
    for k in ("rgb_map", "acc_map"):
        assert float((ev[k] - tr[k].detach()).abs().max()) <= 1e-4, k
    m.renderer.close()


def test_backward_without_a_forward_is_refused():
    import ctypes as C
    from posegen_amd import _ffi, surreal_config
    from posegen_amd.raycaster import HipRenderer
    r = HipRenderer(surreal_config(), DEV)
    gr = _ffi.PgNetGrads()
    rc = r.lib.pg_train_backward(r.handle, None, None, None, None, None, C.byref(gr), C.byref(gr))
    assert rc == _ffi.PG_ESTATE and b"pg_train_forward" in r.lib.pg_last_error(r.handle)
    r.close()
