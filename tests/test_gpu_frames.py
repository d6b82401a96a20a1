"""GPU tests at the frame level: render_path against the reference's own frames (golden
fixture captured through run_nerf.render_path), and size-independent properties at the
BASELINE resolution (determinism, chunk-group independence of rays that hit)."""
import numpy as np
import pytest
import torch

from posegen_amd import PREC_BF16, PREC_FP16, PREC_FP32, synthetic as syn
from tests.helpers import cfg_from_golden, load_golden, model_for

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def caster():
    from posegen_amd.raycaster import HipRayCaster
    g = load_golden("frame64")
    cfg = cfg_from_golden(g)
    wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
    c = HipRayCaster.from_weights(cfg, wc, wf, tv, td, device=DEV, precision=PREC_FP32)
    yield c
    c.renderer.close()


def _render_golden_frames(caster, g):
    from posegen_amd.raycaster import create_raycaster
    from posegen_amd.render import render_path
    cfg = caster.cfg
    kw = {"ray_caster": caster, "perturb": False, "N_importance": cfg.n_importance, "N_samples": cfg.n_samples,
          "use_viewdirs": True, "raw_noise_std": 0., "ray_noise_std": 0., "ext_scale": cfg.ext_scale,
          "preproc_kwargs": {}, "lindisp": False, "nerf_type": "nerf"}
    return render_path(torch.tensor(g["c2ws"]), (int(g["H"]), int(g["W"]), g["focals"]), int(g["chunk"]), kw,
                       kp=torch.tensor(g["kps"]), skts=torch.tensor(g["skts"]), bones=torch.tensor(g["bones"]),
                       cams=None, white_bkgd=True, ret_acc=True, ext_scale=cfg.ext_scale)


@pytest.mark.parametrize("prec,tol", [(PREC_FP32, 1e-4), (PREC_FP16, 1e-3), (PREC_BF16, 5e-3)])
def test_render_path_matches_reference_frames(caster, prec, tol):
    """Two 64x64 frames, bbox cull, chunk boundary inside the frame (chunk=1024), white bg."""
    g = load_golden("frame64")
    caster.renderer.set_precision(prec)
    rgbs, disps, accs, vids, boxes = _render_golden_frames(caster, g)
    assert rgbs.shape == g["rgbs"].shape and disps.shape == g["disps"].shape and accs.shape == g["accs"].shape
    assert [len(v) for v in vids] == list(g["n_valid"])
    assert np.array_equal(np.array([[b[0], b[1]] for b in boxes]), g["boxes"])
    e_rgb = float(np.abs(rgbs - g["rgbs"]).max())
    e_acc = float(np.abs(accs - g["accs"]).max())
    solid = g["accs"] > 1e-3
    e_disp = float(np.abs(disps - g["disps"])[solid].max())
    print(f"prec {prec}: rgb {e_rgb:.2e} acc {e_acc:.2e} disp {e_disp:.2e}")
    assert e_rgb <= tol and e_acc <= tol and e_disp <= tol
    # pixels outside the box keep the white background / zero acc
    assert np.all(rgbs[0, 0, 0] == 1.0) and accs[0, 0, 0, 0] == 0.0


def test_full_frame_512_deterministic_and_consistent(caster):
    """BASELINE config 2 size: 512x512 x (64+16).  Bitwise determinism of the bf16 path and
    agreement of bf16 with the exact fp32 mode within the documented bf16 bound."""
    from bench import full_frame_rays
    rb, skts, cyl, *_ = full_frame_rays(512, 512, torch.device(DEV))
    r = caster.renderer
    r.set_precision(PREC_BF16)
    a = r.render_rays(rb, skts, cyl, want_alpha=False)
    b = r.render_rays(rb, skts, cyl, want_alpha=False)
    for k in ("rgb_map", "disp_map", "acc_map"):
        assert torch.equal(a[k], b[k]), f"{k} differs between two identical launches"
        assert torch.isfinite(a[k]).all()
    r.set_precision(PREC_FP32)
    sub = torch.arange(0, rb.shape[0], 37, device=DEV)[:4096]
    ex = r.render_rays(rb[sub], skts, cyl, want_alpha=False)
    assert float((a["rgb_map"][sub] - ex["rgb_map"]).abs().max()) <= 5e-3
    assert float((a["acc_map"][sub] - ex["acc_map"]).abs().max()) <= 5e-3
    acc = a["acc_map"]
    assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0
    assert 0.02 < float((acc > 0.5).float().mean()) < 0.9       # a body, not an empty or full frame
    # the headline workload against the ORACLE, not only against the repo's own fp32 kernel: a
    # strided 256-ray subset of the full-frame launches (rays that hit are independent of their batch): bf16 within
    # its documented bound and 60 dB, the compensated mode within the north star's 1e-4
    from oracle import anerf_oracle as orc
    from posegen_amd import PREC_FP16C
    from tests.helpers import oracle_cfg, torch_weights
    g = load_golden("frame64")
    cfg = caster.cfg
    wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
    n = rb.shape[0]
    sel = torch.arange(97, n, n // 256)[:256]
    ref = orc.render_rays(rb.cpu()[sel], skts.cpu(), cyl.cpu(), oracle_cfg(cfg, tv, td), torch_weights(wc), torch_weights(wf),
                          cfg.n_samples, cfg.n_importance)
    r.set_precision(PREC_FP16C)
    c = r.render_rays(rb, skts, cyl, want_alpha=False)
    solid = ref["acc_map"] > 1e-3
    for name, got, tol in (("bf16", a, 5e-3), ("fp16c", c, 1e-4)):
        err = max(float((got[k].cpu()[sel] - ref[k]).abs().max()) for k in ("rgb_map", "acc_map"))
        e_disp = float((got["disp_map"].cpu()[sel] - ref["disp_map"])[solid].abs().max())
        mse = float(((got["rgb_map"].cpu()[sel] - ref["rgb_map"]) ** 2).mean())
        print(f"config 2 subset vs oracle, {name}: max |d rgb/acc| {err:.2e}, |d disp| {e_disp:.2e}, rgb MSE {mse:.2e}")
        assert err <= tol and mse <= 1e-6, (name, err, mse)
        if name == "fp16c":
            assert e_disp <= 1e-4, e_disp
    r.set_precision(PREC_FP32)


def test_rays_are_independent_of_batching(caster):
    """A ray that hits the cylinder gives the same result alone, in a slice, or in the frame
    (the only cross-ray coupling is the nanmean patch of rays that miss)."""
    from bench import full_frame_rays
    rb, skts, cyl, *_ = full_frame_rays(128, 128, torch.device(DEV))
    r = caster.renderer
    r.set_precision(PREC_FP32)
    full = r.render_rays(rb, skts, cyl, want_alpha=False)
    part = r.render_rays(rb[5000:5777], skts, cyl, want_alpha=False)
    for k in ("rgb_map", "disp_map", "acc_map"):
        assert torch.equal(full[k][5000:5777], part[k])


@pytest.mark.parametrize("prec", ["bf16", "fp16c"])
def test_large_calls_run_as_ray_range_launches_with_identical_results(caster, prec, monkeypatch):
    """A call with more rays than the record batch (2^19 by default; POSEGEN_REC_BATCH here) runs the per-ray record
    kernel + the fused kernel over consecutive ray ranges, so that the record buffer does not grow with the call:
    bitwise the single-launch result (every hit ray is independent of its neighbours)."""
    from bench import full_frame_rays
    from posegen_amd import PREC_BY_NAME
    rb, skts, cyl, *_ = full_frame_rays(128, 128, torch.device(DEV))
    r = caster.renderer
    r.set_precision(PREC_BY_NAME[prec])
    try:
        one = r.render_rays(rb, skts, cyl, n_samples=64, n_importance=16, want_alpha=False)
        monkeypatch.setenv("POSEGEN_REC_BATCH", "1000")             # 16384 rays -> 17 launches per pass, the last one short
        many = r.render_rays(rb, skts, cyl, n_samples=64, n_importance=16, want_alpha=False)
    finally:
        monkeypatch.delenv("POSEGEN_REC_BATCH", raising=False)
        r.set_precision(PREC_FP32)
    for k in ("rgb_map", "disp_map", "acc_map", "rgb0", "acc0"):
        assert torch.equal(one[k], many[k]), k


def test_render_full_image_from_a_camera_equals_explicit_rays(caster):
    """render(H, W, focal, c2w=...) without rays: the reference's full-image special case
    (trainer.py:109-113) = get_rays of every pixel, result reshaped to [H, W, ...]."""
    from posegen_amd.rays import get_rays
    from posegen_amd.render import render
    g = load_golden("frame64")
    cfg = caster.cfg
    H, W = int(g["H"]), int(g["W"])
    c2w, focal = torch.tensor(g["c2ws"][0]), float(g["focals"][0])
    from posegen_amd.skeleton import get_kp_bounding_cylinder
    cyl = torch.tensor(np.asarray(get_kp_bounding_cylinder(g["kps"][:1], ext_scale=cfg.ext_scale, extend_mm=250,
                                                           top_expand_ratio=1.60, bot_expand_ratio=1.10, head="-y")),
                       dtype=torch.float32)
    kw = dict(ray_caster=caster, N_samples=cfg.n_samples, N_importance=cfg.n_importance, skts=torch.tensor(g["skts"][:1]),
              cyls=cyl, kp_batch=None)
    caster.renderer.set_precision(PREC_FP32)
    a = render(H, W, focal, chunk=4096, c2w=c2w[:3, :4], **kw)
    ro, rd = get_rays(H, W, focal, c2w[:3, :4])
    b = render(H, W, focal, chunk=4096, rays=(ro, rd), **kw)
    assert a["rgb_map"].shape == (H, W, 3) and a["acc_map"].shape == (H, W)
    for k in ("rgb_map", "disp_map", "acc_map"):
        assert torch.equal(a[k], b[k]), k
    assert float(a["acc_map"].max()) > 0.5
    with pytest.raises(ValueError):
        render(H, W, focal, chunk=4096, **kw)


def test_multi_subject_batch_keeps_every_subjects_caster_resident(tmp_path):
    """BASELINE config 4's "multi-subject batch": several subjects = several checkpoints, rendered in one
    batch of frames.  Each subject's caster stays loaded (load_raycaster memoises per file); frames rendered
    interleaved across subjects are bitwise the frames of each subject rendered alone."""
    from posegen_amd.config import h36m_config
    from posegen_amd.raycaster import HipRayCaster, load_raycaster
    from posegen_amd.render import render_path
    cfg = h36m_config()
    H = W = 64
    paths = []
    for s_ in range(3):
        c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 10 + s_), device=DEV, precision=PREC_BF16)
        p = str(tmp_path / f"subject{s_}.tar")
        torch.save(c.state_dict(), p)
        c.renderer.close()
        paths.append(p)
    _, kps, skts = syn.make_pose(4, 3)
    c2ws, focals = syn.make_camera(4, H, W)
    cams = torch.tensor([0., 1., 2., 3.])

    def frames(path, ids):
        kw = load_raycaster(path, cfg, device=DEV, precision="bf16")
        return render_path(torch.tensor(c2ws[ids]), (H, W, focals[ids]), 4096, kw, kp=torch.tensor(kps[ids]),
                           skts=torch.tensor(skts[ids]), cams=cams[ids], white_bkgd=True, ret_acc=True,
                           ext_scale=cfg.ext_scale)[0]

    alone = [frames(p, np.arange(4)) for p in paths]
    casters = [load_raycaster(p, cfg, device=DEV, precision="bf16")["ray_caster"] for p in paths]
    assert len({id(c) for c in casters}) == 3                      # three resident casters, none reloaded
    for f in range(4):                                              # interleaved: frame f of subject f % 3
        s_ = f % 3
        got = frames(paths[s_], np.array([f]))
        assert np.array_equal(got[0], alone[s_][f])
        assert load_raycaster(paths[s_], cfg, device=DEV, precision="bf16")["ray_caster"] is casters[s_]
    assert not np.array_equal(alone[0], alone[1])


def test_render_frame_equals_ray_level_path(caster):
    """pg_render_frame (rays generated, rendered and scattered on the device) against the
    ray-level route the reference takes: kp_to_valid_rays on the host -> render() -> scatter."""
    from posegen_amd.rays import kp_to_valid_rays
    from posegen_amd.render import render
    g = load_golden("frame64")
    cfg = caster.cfg
    r = caster.renderer
    r.set_precision(PREC_FP32)
    r.set_chunk(int(g["chunk"]))
    H, W = int(g["H"]), int(g["W"])
    c2ws, kps, skts = torch.tensor(g["c2ws"]), torch.tensor(g["kps"]), torch.tensor(g["skts"])
    rays, vids, cyls, boxes = kp_to_valid_rays(c2ws, H, W, g["focals"], kps=kps, ext_scale=cfg.ext_scale)
    kw = {"ray_caster": caster, "perturb": False, "N_importance": cfg.n_importance, "N_samples": cfg.n_samples,
          "use_viewdirs": True, "raw_noise_std": 0., "ray_noise_std": 0., "ext_scale": cfg.ext_scale,
          "preproc_kwargs": {}, "lindisp": False, "nerf_type": "nerf", "want_alpha": False}
    for i in range(len(c2ws)):
        ret = render(H, W, g["focals"], rays=rays[i], chunk=int(g["chunk"]), kp_batch=kps[i:i + 1],
                     skts=skts[i:i + 1], cyls=cyls[i:i + 1], cams=None, subject_idxs=None, bones=None, **kw)
        ref_rgb = torch.ones(H * W, 3, device=DEV)
        ref_acc = torch.zeros(H * W, device=DEV)
        vid = vids[i].to(DEV)
        ref_rgb[vid] = ret["rgb_map"] + (1. - ret["acc_map"][..., None]) * ref_rgb[vid]
        ref_acc[vid] = ret["acc_map"]
        rgb, disp, acc, rgb8 = r.render_frame(H, W, g["focals"][i], c2ws[i], boxes[i], skts[i:i + 1], cyls[i:i + 1],
                                              base_bg=1.0, want_uint8=True)
        e = float((rgb.view(-1, 3) - ref_rgb).abs().max())
        print(f"frame {i}: device front/back end vs ray-level route: max |d rgb| {e:.2e}, "
              f"bitwise {bool(torch.equal(rgb.view(-1, 3), ref_rgb))}")
        assert e <= 2e-6
        assert float((acc.view(-1) - ref_acc).abs().max()) <= 2e-6
        assert torch.isfinite(disp).all()
        q = (rgb * 255.0).clamp(0, 255).to(torch.uint8)
        assert torch.equal(rgb8, q)


def test_pose_kinematics_matches_reference_golden(caster):
    """pg_pose_kinematics (device, float64) against the reference's get_smpl_l2ws / inverse
    (golden kinematics fixture) and against the host float64 port on random poses."""
    from posegen_amd.skeleton import bones_to_pose
    g = load_golden("kinematics")
    kps, skts, l2ws = caster.renderer.pose_kinematics(torch.tensor(g["bones"]), g["rest_pose"], want_l2ws=True)
    assert float(np.abs(l2ws.cpu().numpy() - g["l2ws"]).max()) <= 1e-12
    assert float(np.abs(kps.cpu().numpy() - g["kps"]).max()) <= 1e-6
    assert float(np.abs(skts.cpu().numpy() - g["skts"]).max()) <= 1e-6
    rng = np.random.RandomState(5)
    bones = rng.normal(0, 0.6, size=(257, 24, 3))
    bones[0] = 0.0                                   # identity rotations (small-angle branch)
    bones[1] *= 1e-5
    rest = g["rest_pose"]
    k_ref, s_ref, l_ref = bones_to_pose(bones, rest)
    kps, skts, l2ws = caster.renderer.pose_kinematics(torch.tensor(bones), rest, want_l2ws=True)
    assert float(np.abs(l2ws.cpu().numpy() - l_ref).max()) <= 1e-12
    assert np.array_equal(kps.cpu().numpy(), k_ref.astype(np.float32))
    assert float(np.abs(skts.cpu().numpy() - s_ref).max()) <= 1e-6


def test_render_frame_background_image_and_frame_code():
    """pg_render_frame with a background image and with a per-frame frame code (h36m config),
    against the ray-level route: same pixels."""
    from posegen_amd import h36m_config
    from posegen_amd.raycaster import HipRayCaster
    from posegen_amd.rays import kp_to_valid_rays
    cfg = h36m_config()
    wc, wf, tv, td = syn.make_model(cfg, 2)
    c = HipRayCaster.from_weights(cfg, wc, wf, tv, td, device=DEV, precision=PREC_FP32)
    try:
        r = c.renderer
        H = W = 48
        _, kps, skts = syn.make_pose(2, 3)
        c2ws, focals = syn.make_camera(2, H, W)
        kps, skts, c2ws = torch.tensor(kps), torch.tensor(skts), torch.tensor(c2ws)
        rays, vids, cyls, boxes = kp_to_valid_rays(c2ws, H, W, focals, kps=kps, ext_scale=cfg.ext_scale)
        bg = torch.rand(H * W, 3, generator=torch.Generator().manual_seed(0))
        for i, cam in ((0, 3.0), (1, -1.0)):
            ro, rd = rays[i]
            n = ro.shape[0]
            assert n > 0
            vd = rd / torch.norm(rd, dim=-1, keepdim=True)
            rb = torch.cat([ro, rd, torch.zeros(n, 1), torch.ones(n, 1), vd], -1)
            ret = r.render_rays(rb, skts[i:i + 1], cyls[i:i + 1], cams=torch.full((n,), cam), want_alpha=False)
            ref = bg.clone().to(DEV)
            vid = vids[i].to(DEV)
            ref[vid] = ret["rgb_map"] + (1. - ret["acc_map"][..., None]) * ref[vid]
            rgb, disp, acc = r.render_frame(H, W, focals[i], c2ws[i], boxes[i], skts[i:i + 1], cyls[i:i + 1],
                                            cam=cam, bg=bg)
            assert torch.equal(rgb.view(-1, 3), ref), f"frame {i}: max diff {float((rgb.view(-1, 3) - ref).abs().max()):.2e}"
            assert float(acc.max()) > 0.05
    finally:
        c.renderer.close()


def test_checkpoint_round_trip_and_cached_loader(tmp_path):
    """state_dict() in the reference's five-dict key scheme -> .tar -> load_raycaster: the same
    renders, and a second load of the unchanged file returns the resident caster."""
    from posegen_amd import h36m_config
    from posegen_amd.raycaster import HipRayCaster, load_raycaster
    from bench import full_frame_rays
    cfg = h36m_config()
    wc, wf, tv, td = syn.make_model(cfg, 4)
    a = HipRayCaster.from_weights(cfg, wc, wf, tv, td, device=DEV, precision=PREC_FP32)
    path = str(tmp_path / "ckpt.tar")
    sd = a.state_dict()
    assert {"network_fn_state_dict", "network_fine_state_dict", "embed_state_dict", "embeddirs_state_dict",
            "embedbones_state_dict"} <= set(sd)
    torch.save(sd, path)
    kw = load_raycaster(path, cfg, device=DEV, precision=PREC_FP32)
    kw2 = load_raycaster(path, cfg, device=DEV, precision=PREC_FP32)
    assert kw2["ray_caster"] is kw["ray_caster"]
    rb, skts, cyl, *_ = full_frame_rays(32, 32, torch.device(DEV))
    cams = torch.full((rb.shape[0],), 2.0)
    x = a.renderer.render_rays(rb, skts, cyl, cams=cams, want_alpha=False)
    y = kw["ray_caster"].renderer.render_rays(rb, skts, cyl, cams=cams, want_alpha=False)
    for k in ("rgb_map", "disp_map", "acc_map"):
        assert torch.equal(x[k], y[k])
    a.renderer.close()


@pytest.mark.parametrize("n_frames", [1, 3])
def test_in_process_multi_device_frames_are_bitwise_those_of_one_device(n_frames):
    """HipRayCaster(devices=[...]) = the nn.DataParallel replacement (core/raycasters.py:157): render_path
    spreads whole frames (F >= G) or a frame's ray chunks (F < G) over the devices inside ONE call and
    must return exactly the single-device frames.  With one GPU the same device is listed twice (two
    workers, two streams); with more GPUs the real devices are used."""
    from posegen_amd import surreal_config, synthetic as syn
    from posegen_amd.raycaster import HipRayCaster
    from posegen_amd.render import render_path
    cfg = surreal_config()
    model = syn.make_model(cfg, 0)
    ndev = torch.cuda.device_count()
    devices = list(range(ndev)) if ndev >= 2 else [0, 0]
    H = W = 96
    _, kps, skts = syn.make_pose(n_frames, 5)
    c2ws, focals = syn.make_camera(n_frames, H, W)
    kw = dict(kp=torch.tensor(kps), skts=torch.tensor(skts), white_bkgd=True, ret_acc=True, ext_scale=cfg.ext_scale)
    outs = []
    for devs in (None, devices):
        c = HipRayCaster.from_weights(cfg, *model, device="cuda:0", precision="fp32", devices=devs)
        rk = {"ray_caster": c, "N_samples": cfg.n_samples, "N_importance": cfg.n_importance}
        outs.append(render_path(torch.tensor(c2ws), (H, W, focals), 1024, rk, **kw))     # chunk 1024: several groups per box
        c.renderer.close()
    one, multi = outs
    assert sum(len(v) for v in one[3]) > 2 * 1024 * n_frames / 2
    for a, b in zip(one[:3], multi[:3]):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert np.array_equal(np.array(one[4]), np.array(multi[4]))


def test_pose_boxes_on_device_equal_the_reference_boxes():
    """pg_pose_boxes: bounding cylinder (float32) and projected integer box (float64) of device key points --
    bit-for-bit the cylinders and boxes the reference computed for the golden fixture, and those of the host
    restatement on 500 random poses and cameras."""
    from posegen_amd import surreal_config, synthetic as syn
    from posegen_amd.raycaster import HipRenderer
    from posegen_amd.rays import kp_to_boxes
    g = load_golden("valid_rays")
    H, W = int(g["H"]), int(g["W"])
    r = HipRenderer(surreal_config(), DEV)
    cyls, boxes = r.pose_boxes(torch.tensor(g["kps"]), g["c2ws"], H, W, float(g["focals"][0]), 0.001)
    assert np.array_equal(cyls.cpu().numpy(), g["cyls"])
    want = np.array([[b[0][0], b[0][1], b[1][0], b[1][1]] for b in g["boxes"]])
    assert np.array_equal(boxes.cpu().numpy(), want)
    # random poses, one camera each
    n = 500
    _, kps, _ = syn.make_pose(n, 11)
    rng = np.random.RandomState(3)
    c2ws, focals = syn.make_camera(n, 512, 512)
    c2ws = c2ws.copy()
    c2ws[:, :3, 3] += rng.uniform(-0.3, 0.3, size=(n, 3)).astype(np.float32)
    hc, hb, _ = kp_to_boxes(torch.tensor(c2ws), 512, 512, focals, kps=torch.tensor(kps), ext_scale=0.001)
    dc, db = r.pose_boxes(torch.tensor(kps), c2ws, 512, 512, float(focals[0]), 0.001)
    assert np.array_equal(dc.cpu().numpy(), hc.numpy())
    assert np.array_equal(db.cpu().numpy(), np.array([[b[0][0], b[0][1], b[1][0], b[1][1]] for b in hb]))
    r.close()


def test_gan_loop_render_call_on_device_equals_the_host_route():
    """BASELINE config 5: render_for_regressor (kinematics, boxes, frames, crop, normalise, resize all on the
    device) against the reference-shaped route (numpy kinematics and boxes, render_path, uint8 on the host)."""
    from posegen_amd import surreal_config, synthetic as syn
    from posegen_amd.ganloop import render_for_regressor
    from posegen_amd.raycaster import HipRayCaster
    from posegen_amd.render import render_path
    from posegen_amd.skeleton import SURREAL_REST_SCALE, bones_to_pose, smpl_rest_pose
    cfg = surreal_config()
    c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=DEV, precision="fp32")
    H = W = 128
    rest = smpl_rest_pose * SURREAL_REST_SCALE
    c2ws, focals = syn.make_camera(3, H, W)
    bones = syn.make_bones(3, 7)
    img, frames = render_for_regressor(c, torch.tensor(bones, device=DEV), rest, c2ws[0], H, W, float(focals[0]),
                                       ext_scale=cfg.ext_scale, crop=(25, 103), out_res=56, return_frames=True)
    kps, skts, _ = bones_to_pose(bones, rest)
    rk = {"ray_caster": c, "N_samples": cfg.n_samples, "N_importance": cfg.n_importance}
    rgbs, *_ = render_path(torch.tensor(c2ws), (H, W, focals), 4096, rk, kp=torch.tensor(kps.astype(np.float32)),
                           skts=torch.tensor(skts.astype(np.float32)), white_bkgd=True, ext_scale=cfg.ext_scale)
    assert np.array_equal(frames.cpu().numpy(), (rgbs * 255).astype(np.uint8))
    assert img.shape == (3, 3, 56, 56) and torch.isfinite(img).all()
    assert (frames.cpu().numpy() < 255).any(), "something was rendered"
    c.renderer.close()


@pytest.mark.parametrize("prec,tol", [("bf16", 1e-3), ("fp16", 1e-4), ("fp16c", 5e-6)])
def test_limb_skipping_changes_nothing_beyond_rounding(caster, prec, tol):
    """the fused kernels leave out the limbs (groups of four joints; joint pairs in the compensated kernel)
    that a wave's 32 points -- or a whole pass -- are out of cutoff range of: every product left out has a cutoff
    weight below 2^-24.  With the masks off (pg_set_far_skip(0): every limb computed, the on-chip variant computes
    every limb's direction part each pass) the frame is the same except where a 6e-8 change of an fp32 pre-activation
    flips the 16-bit rounding of one activation (measured: 0.04 % of the points in bf16): the MEAN difference of every
    map is below 1e-8, at most 3 % of the rays differ at all, and no ray by more than a few operand roundings of the
    mode (`tol`) -- on the 512 x 512 benchmark frame and on a culled box with rays that miss."""
    from bench import full_frame_rays
    from posegen_amd import PREC_BY_NAME
    from posegen_amd.skeleton import get_kp_bounding_cylinder
    r = caster.renderer
    r.set_precision(PREC_BY_NAME[prec])
    rb, skts, cyl, *_ = full_frame_rays(512, 512, torch.device(DEV))
    _, kps, sk2 = syn.make_pose(1, 9)
    cyl2 = torch.tensor(get_kp_bounding_cylinder(kps, ext_scale=0.001), dtype=torch.float32, device=DEV)
    try:
        for rays, pose, cy in ((rb, skts, cyl), (rb[::3].contiguous(), torch.tensor(sk2, device=DEV), cyl2)):
            r.set_far_skip(True)
            a = r.render_rays(rays, pose, cy, want_alpha=False)
            r.set_far_skip(False)
            b = r.render_rays(rays, pose, cy, want_alpha=False)
            for k in ("rgb_map", "acc_map", "rgb0", "acc0"):
                d = (a[k] - b[k]).abs().reshape(a[k].shape[0], -1)
                assert float(d.max()) <= tol, (prec, k, float(d.max()))
                assert float(d.mean()) <= 1e-8, (prec, k, float(d.mean()))
                assert float((d.amax(-1) > 0).float().mean()) <= 0.03, (prec, k)
            assert torch.isfinite(a["rgb_map"]).all()
    finally:
        r.set_far_skip(True)
        r.set_precision(PREC_FP32)
