"""GPU tests of the training step: `TrainableRayCaster` -- forward in training mode with a
tape, loss on the host side of the ABI, `loss.backward()` through pg_train_backward -- against the reference's own
autograd (fixtures tests/golden/train_grads*.npz, produced by tools/gen_golden.py from the imported reference)."""
import os

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


def _trainable(g, train_precision="fp32", precision="fp32"):
    from posegen_amd.raycaster import HipRayCaster
    from posegen_amd.train import TrainableRayCaster
    cfg = cfg_from_golden(g)
    wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
    c = HipRayCaster.from_weights(cfg, wc, wf, float(g["tau_v"]), float(g["tau_d"]), device=DEV, precision=precision)
    return cfg, TrainableRayCaster(c, train_precision=train_precision)


@pytest.mark.parametrize("name", ["train_grads", "train_grads_h36m", "train_grads_softplus", "train_grads_raw"])
def test_training_step_gradients_match_the_reference_autograd(name):
    """One training step of the reference on the HIP path: same draws (pytest=True), same loss; the loss, the four
    maps it reads and the gradient of every parameter tensor of both nets (24 each, + the frame codes) within 1e-4 of
    the tensor's largest entry / of its norm.  `train_grads_softplus`: --density_type softplus (its derivative in the
    composite backward).  `train_grads_raw` is the UN-FILTERED h36m batch (a ReLU pre-activation within rounding of
    zero: the reference's own gradient moves by `grad_sensitivity` = 2.5e-4 under 1e-7 weight noise): its bound is
    4 x that stored sensitivity instead of 1e-4."""
    g = load_golden(name)
    tol = max(1e-4, 4.0 * float(g["grad_sensitivity"])) if name == "train_grads_raw" else 1e-4
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
        for k, p in net.named_parameters():          # the reference's parameter names (nerf.py:57-88)
            ref_vals, ref_norm = g[f"gval_{tag}_{k}"], float(g[f"gnorm_{tag}_{k}"])
            assert p.grad is not None, (tag, k)
            got = p.grad.detach().cpu().numpy().reshape(-1)
            scale = max(float(np.abs(ref_vals).max()), ref_norm / np.sqrt(got.size), 1e-12)
            err = float(np.abs(got[grad_sample_index(got.size)] - ref_vals).max())
            nerr = abs(float(np.linalg.norm(got.astype(np.float64))) - ref_norm)
            assert err <= tol * scale + 1e-9, (tag, k, err, scale)
            assert nerr <= tol * ref_norm + 1e-9, (tag, k, nerr, ref_norm)
            n_checked += 1
    assert n_checked == (50 if cfg.framecode_ch else 48)
    m.renderer.close()


@pytest.mark.parametrize("train_precision", ["fp32", "bf16"])
def test_training_gradients_of_an_odd_batch_match_the_oracle_autograd(train_precision):
    """A batch whose point count is not a multiple of 4 (25 rays x 33 + 7 samples) takes the small-tile GEMM with the
    ReLU mask and the bias sums as kernels of their own instead of fused into the large-tile GEMM: same gradients,
    checked against the oracle under torch autograd (itself pinned to the reference's gradients on the fixtures).
    bf16: the same batch in the 16-bit mode -- 825 and 1000 rows end inside a tile of every persistent layer kernel
    (64-row tiles of the 256-wide layers, 32-row tiles of layer 0 and of the skip layer: clamped requests, rows past
    the end never stored) -- within the mode's own bounds (norms 2e-2, entries 0.1 of the tensor's largest)."""
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
    m = TrainableRayCaster(c, train_precision=train_precision)
    m.train()
    out = m(rb, N_samples=33, skts=sk, cyls=cy, N_importance=7, draws={k: v.to(DEV) for k, v in draws.items()})
    _loss_of(out, target.to(DEV)).backward()
    worst = 0.0
    for tag, net, refw in (("coarse", m.network, twc), ("fine", m.network_fine, twf)):
        for k, p in net.named_parameters():
            r = refw[k].grad.numpy().reshape(-1)
            got = p.grad.detach().cpu().numpy().reshape(-1)
            scale = max(float(np.abs(r).max()), float(np.linalg.norm(r)) / np.sqrt(r.size), 1e-12)
            if train_precision == "fp32":
                assert float(np.abs(got - r).max()) <= 2e-4 * scale + 1e-9, (tag, k, float(np.abs(got - r).max()), scale)
            else:
                rn = float(np.linalg.norm(r.astype(np.float64)))
                nerr = abs(float(np.linalg.norm(got.astype(np.float64))) - rn) / max(rn, 1e-12)
                verr = float(np.abs(got - r).max()) / scale
                worst = max(worst, nerr, verr)
                assert np.isfinite(got).all() and nerr <= 2e-2 and verr <= 1e-1, (tag, k, nerr, verr)
    if train_precision != "fp32":
        print(f"odd batch, 16-bit mode: worst relative gradient deviation {worst:.2e}")
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
    rc = r.lib.pg_train_backward(r.handle, None, 1, None, None, None, None, C.byref(gr), C.byref(gr))
    assert rc == _ffi.PG_ESTATE and b"pg_train_forward" in r.lib.pg_last_error(r.handle)
    r.close()


def test_a_second_forward_makes_the_first_tape_stale():
    """the handle holds ONE tape.  Two training-mode forwards, then backward through the FIRST: refused
    (PG_ESTATE, 'overwritten') instead of differentiating the second call's activations; backward through the second
    still works, and its gradients equal those of a lone forward + backward (bitwise: the reductions are ordered)."""
    from posegen_amd._ffi import PgError, PG_ESTATE
    g = load_golden("train_grads")
    cfg, m = _trainable(g)
    m.train()
    rb, sk, cy = torch.tensor(g["ray_batch"]), torch.tensor(g["skts"]), torch.tensor(g["cyl"])
    target = torch.tensor(g["target"], device=DEV)
    call = lambda rays: m(rays, N_samples=cfg.n_samples, skts=sk, cyls=cy, N_importance=cfg.n_importance, draws=None)
    first = _loss_of(call(rb[:24]), target[:24])
    second = _loss_of(call(rb[8:40]), target[8:40])
    with pytest.raises(PgError) as ei:
        first.backward()
    assert ei.value.code == PG_ESTATE and "overwritten" in str(ei.value)
    m.zero_grad()
    second.backward()
    got = [p.grad.clone() for p in m.network.parameters() if p.grad is not None]
    m.zero_grad()
    _loss_of(call(rb[8:40]), target[8:40]).backward()
    again = [p.grad for p in m.network.parameters() if p.grad is not None]
    assert len(got) == 24 and all(torch.equal(a, b) for a, b in zip(got, again)), "gradients are bitwise repeatable"
    m.renderer.close()


def test_checkpoint_round_trip_in_the_reference_layout(tmp_path):
    """TrainableRayCaster.state_dict() is the reference checkpoint (raycasters.py:752-766:
    network_fn_state_dict with pts_linears.0.weight ..., embed_state_dict with cutoff_dist / tau ...): saved with
    torch.save like Trainer.save_nerf (trainer.py:496-507), it is read back by HipRayCaster.load_state_dict /
    load_raycaster and by TrainableRayCaster.load_state_dict, and get_grad_vars' accessors exist."""
    from posegen_amd.raycaster import NET_TENSOR_ORDER, HipRayCaster, load_raycaster
    g = load_golden("train_grads")
    cfg, m = _trainable(g)
    m.train()
    net, fine = m.get_networks()
    embed, embedbones, embeddirs = m.get_embed_fns()
    assert embedbones is None and len(list(net.pts_linears[0].parameters())) == 2
    assert [k for k, _ in net.named_parameters()] == list(NET_TENSOR_ORDER)
    assert not any(p.requires_grad for p in embed.parameters())       # cutoff_dist: requires_grad=False (opt_cutoff off)
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=5e-3)
    rb, sk, cy = torch.tensor(g["ray_batch"]), torch.tensor(g["skts"]), torch.tensor(g["cyl"])
    out = m(rb, N_samples=cfg.n_samples, skts=sk, cyls=cy, N_importance=cfg.n_importance)
    _loss_of(out, torch.tensor(g["target"], device=DEV)).backward()
    opt.step()                                                         # weights now differ from the initial ones
    import argparse
    m.update_embed_fns(150000, argparse.Namespace(cutoff_step=250, cutoff_rate=10.))      # tau: 20 * 10^(150/250) = 79.6
    assert abs(embed.get_tau() - 79.62) < 0.01
    sd = m.state_dict()
    assert set(sd) == {"network_fn_state_dict", "network_fine_state_dict", "embed_state_dict", "embedbones_state_dict", "embeddirs_state_dict"}
    assert set(sd["network_fn_state_dict"]) == set(NET_TENSOR_ORDER) and set(sd["embed_state_dict"]) == {"cutoff_dist", "tau"}
    path = tmp_path / "000001.tar"
    torch.save({"global_step": 1, **sd}, path)
    m.sync_inference_weights()
    m.eval()
    with torch.no_grad():
        want = m(rb, N_samples=cfg.n_samples, skts=sk, cyls=cy, N_importance=cfg.n_importance)
    # (1) a fresh inference caster from the file
    kw = load_raycaster(str(path), cfg, device=DEV, precision="fp32")
    got = kw["ray_caster"](rb, N_samples=cfg.n_samples, skts=sk, cyls=cy, N_importance=cfg.n_importance)
    for k in ("rgb_map", "acc_map", "disp_map"):
        assert torch.equal(got[k], want[k]), k
    # (2) a fresh trainable caster resumes from it
    cfg2, m2 = _trainable(g)
    m2.load_state_dict(torch.load(path, map_location="cpu", weights_only=False))
    for (ka, a), (kb, b) in zip(m.network_fine.named_parameters(), m2.network_fine.named_parameters()):
        assert ka == kb and torch.equal(a, b)
    assert m2.embeddirs_fn.get_tau() == m.embeddirs_fn.get_tau()
    m2.eval()
    with torch.no_grad():
        got2 = m2(rb, N_samples=cfg.n_samples, skts=sk, cyls=cy, N_importance=cfg.n_importance)
    assert torch.equal(got2["rgb_map"], want["rgb_map"])
    for c in (m, m2, kw["ray_caster"]):
        c.renderer.close()


def test_inputs_that_want_a_gradient_are_refused():
    """poses / rays get no gradient on the HIP path (pose optimisation, trainer.py:496-515, is out of
    scope): a tensor that requires one is refused instead of silently left without."""
    g = load_golden("train_grads")
    cfg, m = _trainable(g)
    m.train()
    rb, sk, cy = torch.tensor(g["ray_batch"]), torch.tensor(g["skts"]), torch.tensor(g["cyl"])
    with pytest.raises(NotImplementedError, match="skts requires a gradient"):
        m(rb, N_samples=cfg.n_samples, skts=sk.clone().requires_grad_(True), cyls=cy, N_importance=cfg.n_importance)
    with pytest.raises(TypeError):
        m(rb, N_samples=cfg.n_samples, skts=sk, cyls=cy, N_importance=cfg.n_importance, no_such_argument=1)
    m.renderer.close()


@pytest.mark.parametrize("name", ["train_grads", "train_grads_h36m"])
def test_bf16_training_mode_gradients_are_close_and_repeatable(name):
    """The 16-bit training mode (caster precision bf16: the tape -- embedding rows, activations, activation gradients --
    stored in bf16, bf16 operands in the large forward / dX / dW GEMMs, fp32 accumulation; weights, raw, d_raw and every
    parameter gradient fp32) against the reference's autograd: the loss within 2e-3, the norm of every
    parameter gradient within 1e-2, every sampled entry within 0.1 of its tensor's largest entry (bounds stated: bf16
    operands carry 8 bits; measured 5e-4 on the norms, 5e-2 on single entries), and bitwise the same on a second run."""
    g = load_golden(name)
    cfg, m = _trainable(g, train_precision="bf16")
    m.train()
    cams = torch.tensor(g["cams"]) if "cams" in g else None
    target = torch.tensor(g["target"], device=DEV)

    def run():
        m.zero_grad()
        out = m(torch.tensor(g["ray_batch"]), N_samples=cfg.n_samples, skts=torch.tensor(g["skts"]), cyls=torch.tensor(g["cyl"]),
                cams=cams, N_importance=cfg.n_importance, draws=golden_draws(g))
        loss = _loss_of(out, target)
        loss.backward()
        return float(loss.detach()), {(tag, k): p.grad.clone() for tag, net in (("coarse", m.network), ("fine", m.network_fine))
                                      for k, p in net.named_parameters()}
    loss, grads = run()
    assert abs(loss - float(g["loss"])) <= 2e-3 * max(1.0, abs(float(g["loss"])))
    worst = 0.0
    for (tag, k), gr in grads.items():
        ref_vals, ref_norm = g[f"gval_{tag}_{k}"], float(g[f"gnorm_{tag}_{k}"])
        got = gr.detach().cpu().numpy().reshape(-1)
        scale = max(ref_norm / np.sqrt(got.size), float(np.abs(ref_vals).max()), 1e-12)
        nerr = abs(float(np.linalg.norm(got.astype(np.float64))) - ref_norm) / max(ref_norm, 1e-12)
        verr = float(np.abs(got[grad_sample_index(got.size)] - ref_vals).max()) / scale
        worst = max(worst, nerr, verr)
        assert nerr <= 1e-2 and verr <= 1e-1, (tag, k, nerr, verr)
    print(f"[{name}] bf16 training mode: worst relative gradient deviation {worst:.2e}")
    loss2, grads2 = run()
    assert loss2 == loss and all(torch.equal(grads[k], grads2[k]) for k in grads), "bitwise repeatable"
    m.renderer.close()


def test_training_precision_is_independent_of_the_rendering_precision():
    """wrapping a caster that RENDERS in bf16 (every constructor's default) must not move the training step
    off the reference's fp32 arithmetic: `TrainableRayCaster(caster)` trains in fp32 (gradients within 1e-4 of the
    reference's autograd on the fixture) whatever `set_precision` says, before and after a render-side switch."""
    g = load_golden("train_grads")
    cfg, m = _trainable(g, precision="bf16")
    assert m.train_precision == "fp32"
    m.train()
    for switch in (None, "fp16c"):
        if switch:
            m.renderer.set_precision(switch)
        m.zero_grad()
        out = m(torch.tensor(g["ray_batch"]), N_samples=cfg.n_samples, skts=torch.tensor(g["skts"]), cyls=torch.tensor(g["cyl"]),
                N_importance=cfg.n_importance, draws=golden_draws(g))
        loss = _loss_of(out, torch.tensor(g["target"], device=DEV))
        assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5 * max(1.0, abs(float(g["loss"])))
        loss.backward()
        for tag, net in (("coarse", m.network), ("fine", m.network_fine)):
            for k, p in net.named_parameters():
                ref_vals, ref_norm = g[f"gval_{tag}_{k}"], float(g[f"gnorm_{tag}_{k}"])
                got = p.grad.detach().cpu().numpy().reshape(-1)
                scale = max(float(np.abs(ref_vals).max()), ref_norm / np.sqrt(got.size), 1e-12)
                assert float(np.abs(got[grad_sample_index(got.size)] - ref_vals).max()) <= 1e-4 * scale + 1e-9, (switch, tag, k)
    with pytest.raises(ValueError):
        from posegen_amd.train import TrainableRayCaster
        TrainableRayCaster(m.caster, train_precision="fp16")
    m.renderer.close()


def test_loss_curves_of_both_training_precisions_follow_the_oracle_autograd():
    """more than single-step gradients.  24 Adam steps (lrate 5e-4, the reference's, raycasters.py:186-228) on one
    batch with fixed draws, three ways: the oracle under torch autograd on the CPU (pinned to the reference's gradients),
    the HIP step in fp32 and the HIP step on the bf16 tape.  The fp32 curve stays within 2e-4 (relative) of the oracle's at
    every step (measured 5e-6), the bf16 curve within 5e-2 (measured 2.5e-2 while the loss falls from 0.51 to 0.14), and all three fall."""
    from oracle import anerf_oracle as orc
    from posegen_amd import surreal_config
    from posegen_amd.raycaster import HipRayCaster, make_training_draws
    from posegen_amd.train import TrainableRayCaster
    from tests.helpers import oracle_cfg
    g = load_golden("train_grads")
    cfg = surreal_config(n_samples=32, n_importance=8)
    wc, wf, tv, td = model_for(cfg, 4)
    n, steps = 48, 24
    rb, sk, cy = torch.tensor(g["ray_batch"][:n]), torch.tensor(g["skts"]), torch.tensor(g["cyl"])
    target = torch.tensor(g["target"][:n])
    draws = make_training_draws(n, 32, 8, perturb=1., raw_noise_std=1., pytest=True)
    # the oracle's curve
    tw = lambda w: {k: torch.tensor(v, requires_grad=True) for k, v in w.items()}
    twc, twf = tw(wc), tw(wf)
    opt = torch.optim.Adam(list(twc.values()) + list(twf.values()), lr=5e-4, betas=(0.9, 0.999))
    ref_curve = []
    for it in range(steps):
        opt.zero_grad()
        loss = _loss_of(orc.render_rays(rb, sk, cy, oracle_cfg(cfg, tv, td), twc, twf, 32, 8, draws=draws), target)
        loss.backward()
        opt.step()
        ref_curve.append(float(loss.detach()))
    curves = {}
    for tp in ("fp32", "bf16"):
        c = HipRayCaster.from_weights(cfg, wc, wf, float(tv), float(td), device=DEV, precision="bf16")
        m = TrainableRayCaster(c, train_precision=tp)
        m.train()
        o = torch.optim.Adam(m.parameters(), lr=5e-4, betas=(0.9, 0.999))
        cur = []
        for it in range(steps):
            o.zero_grad()
            out = m(rb, N_samples=32, skts=sk, cyls=cy, N_importance=8, draws={k: v.to(DEV) for k, v in draws.items()})
            loss = _loss_of(out, target.to(DEV))
            loss.backward()
            o.step()
            cur.append(float(loss.detach()))
        curves[tp] = cur
        m.renderer.close()
    dev32 = max(abs(a - b) / max(abs(b), 1e-12) for a, b in zip(curves["fp32"], ref_curve))
    dev16 = max(abs(a - b) / max(abs(b), 1e-12) for a, b in zip(curves["bf16"], ref_curve))
    print(f"loss {ref_curve[0]:.5f} -> {ref_curve[-1]:.5f} (oracle); worst relative deviation of the curve: fp32 {dev32:.2e}, bf16 {dev16:.2e}")
    assert ref_curve[-1] < ref_curve[0] and curves["fp32"][-1] < curves["fp32"][0] and curves["bf16"][-1] < curves["bf16"][0]
    assert dev32 <= 2e-4, dev32
    assert dev16 <= 5e-2, dev16


def test_eval_render_after_a_step_uses_the_trained_weights_and_refuses_unknown_keywords():
    """the eval / no-grad branch.  After a backward pass the inference kernels' packed weights lag the parameters:
    an eval-mode render packs them first (no explicit sync_inference_weights()), so it equals the training-mode forward
    without noise; and it refuses the keywords the kernels do not honour exactly as the training branch does."""
    g = load_golden("train_grads")
    cfg, m = _trainable(g)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=5e-3, betas=(0.9, 0.999))
    rb, sk, cy = torch.tensor(g["ray_batch"]), torch.tensor(g["skts"]), torch.tensor(g["cyl"])
    target = torch.tensor(g["target"], device=DEV)
    for it in range(3):
        opt.zero_grad()
        out = m(rb, N_samples=cfg.n_samples, skts=sk, cyls=cy, N_importance=cfg.n_importance)
        _loss_of(out, target).backward()
        opt.step()
    assert m._stale
    m.eval()
    with torch.no_grad():
        ev = m(rb, N_samples=cfg.n_samples, skts=sk, cyls=cy, N_importance=cfg.n_importance)
        with pytest.raises(TypeError):
            m(rb, N_samples=cfg.n_samples, skts=sk, cyls=cy, N_importance=cfg.n_importance, no_such_argument=1)
        with pytest.raises(NotImplementedError):
            m(rb, N_samples=cfg.n_samples, skts=sk, cyls=cy, N_importance=cfg.n_importance, nerf_type="mipnerf")
    assert not m._stale
    m.train()
    tr = m(rb, N_samples=cfg.n_samples, skts=sk, cyls=cy, N_importance=cfg.n_importance)
    for k in ("rgb_map", "acc_map"):
        assert float((ev[k] - tr[k].detach()).abs().max()) <= 1e-4, k
    m.renderer.close()


@pytest.mark.parametrize("fc", [False, True])
def test_device_side_weight_sync_equals_the_host_packing(fc):
    """TrainableRayCaster.sync_inference_weights on the device (pg_load_weights_device: the packed images of the fast paths
    re-formed by gather kernels from the parameter tensors, feature_linear folded into the view layer on the device) gives
    BITWISE the renders of the host route (pg_load_weights) in every precision mode -- including the modes whose images are
    re-packed from refreshed host copies (fp32, the direct kernels for short rays) -- after the parameters have moved; and
    the state dict a checkpoint would take is the parameters'."""
    import copy
    from posegen_amd import h36m_config, surreal_config, synthetic as syn
    from posegen_amd.raycaster import HipRayCaster
    from posegen_amd.train import TrainableRayCaster
    from bench import full_frame_rays
    cfg = h36m_config(n_samples=64) if fc else surreal_config()
    c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 3), device=DEV, precision="bf16")
    m = TrainableRayCaster(c)
    rb, skts, cyl, *_ = full_frame_rays(96, 96, torch.device(DEV))
    n = rb.shape[0]
    cams = ((torch.arange(n, device=DEV) * 5) % cfg.n_framecodes).float() if fc else None
    m.eval()
    kw = dict(N_samples=64, skts=skts, cyls=cyl, cams=cams, N_importance=16)
    with torch.no_grad():
        m(rb, **kw)                                     # the default images exist
        g = torch.Generator(device="cpu").manual_seed(11)
        for p in m.parameters():                        # "optimiser steps"
            if p.requires_grad:
                p.add_(0.02 * torch.randn(p.shape, generator=g).to(p.device) * p.abs().mean())
    outs = {}
    for route in ("device", "host"):
        m.sync_inference_weights(on_device=route == "device")
        for prec, ns in (("bf16", 64), ("fp16", 64), ("fp16c", 64), ("fp32", 64), ("bf16", 40)):
            c.renderer.set_precision(prec)
            with torch.no_grad():
                o = m(rb, **dict(kw, N_samples=ns))
            outs[(route, prec, ns)] = {k: o[k].clone() for k in ("rgb_map", "acc_map", "disp_map")}
        c.renderer.set_precision("bf16")
    for (route, prec, ns), o in outs.items():
        if route == "device":
            for k, v in o.items():
                assert torch.equal(v, outs[("host", prec, ns)][k]), (prec, ns, k)
    # the parameters did move (the test would pass trivially otherwise), and a checkpoint taken from the inner caster follows them
    sd = c.state_dict()["network_fn_state_dict"]
    want = {k: v.detach().cpu() for k, v in m.network.state_dict().items()}
    assert all(torch.equal(sd[k], want[k]) for k in want)
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m.sync_inference_weights(on_device=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        m.sync_inference_weights(on_device=False)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"sync_inference_weights: on the device {(t1 - t0) / 5 * 1e3:.2f} ms, through the host {(t2 - t1) / 5 * 1e3:.2f} ms")
    c.renderer.close()


def test_persistent_layer_kernel_is_bitwise_the_tile_kernel(tmp_path):
    """The persistent layer GEMM (lgemm256 / lgemm432 / lgemm_skip: weights in registers, rows by LDS-DMA) against the tile GEMM
    it replaces (POSEGEN_LGEMM=0), a child process each, on an odd batch (rows ending inside a tile of every variant): the same
    products summed in the same k order by the same MFMA, so every parameter gradient is BITWISE the same -- except the
    tensors behind dH7, where the alpha head's share now enters as a rank-1 term in the feature GEMM's epilogue (one fma
    instead of a rounded product and an add): there within 1e-6 of the tensor's largest entry."""
    import subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    grads = {}
    for v in ("1", "0"):
        out = str(tmp_path / f"g{v}.pt")
        run = subprocess.run([sys.executable, os.path.join(repo, "tests", "diag", "odd_batch_grads.py"), out], capture_output=True, text=True,
                             timeout=300, env=dict(os.environ, POSEGEN_LGEMM=v), cwd=repo)
        assert run.returncode == 0, run.stderr[-2000:]
        grads[v] = torch.load(out)
    same = differ = 0
    for k, a in grads["1"].items():
        b = grads["0"][k]
        if torch.equal(a, b):
            same += 1
        else:
            differ += 1
            assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max()), k
            assert k.split(".", 1)[1] in ("pts_linears.7.weight", "pts_linears.7.bias") or "pts_linears" in k, k
    print(f"persistent vs tile GEMM: {same} gradient tensors bitwise equal, {differ} within 1e-6 (behind the rank-1 fusion)")
    assert same >= 40
