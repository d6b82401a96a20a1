"""BASELINE config 5: the render call of the run_gan.py loop (rpi = 20 poses from the generator -> frames ->
crop -> 224 x 224 regressor input) with the HIP renderer, everything on the device
(posegen_amd.ganloop.render_for_regressor), against the same call done the reference's way with the HIP
renderer underneath (poses to the host, numpy kinematics and boxes, render_path -> float frames on the host
-> uint8 -> crop -> normalise -> resize on the host).

    python tools/bench_gan_loop.py [--prec bf16] [--frames 20] [--reps 5]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from posegen_amd import surreal_config, synthetic as syn
from posegen_amd.ganloop import IMG_NORM_MEAN, IMG_NORM_STD, render_for_regressor, resize_antialiased
from posegen_amd.raycaster import HipRayCaster
from posegen_amd.render import render_path
from posegen_amd.skeleton import SURREAL_REST_SCALE, bones_to_pose, smpl_rest_pose

ap = argparse.ArgumentParser()
ap.add_argument("--prec", default="bf16")
ap.add_argument("--frames", type=int, default=20)       # args.rpi of the reference (run_gan.py:104)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = surreal_config()
caster = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision=a.prec)
H = W = 512
rest = smpl_rest_pose * SURREAL_REST_SCALE
c2ws, focals = syn.make_camera(1, H, W)
c2w, focal = c2ws[0], float(focals[0])
bones = torch.tensor(syn.make_bones(a.frames, 7), device=dev)        # stands in for the generator's output (on the GPU)
kw = {"ray_caster": caster, "N_samples": cfg.n_samples, "N_importance": cfg.n_importance}


def device_route():
    return render_for_regressor(caster, bones, rest, c2w, H, W, focal, ext_scale=cfg.ext_scale, return_frames=True)


def host_route():
    b = bones.cpu().numpy()                                             # outputs_axis_angle[kk].cpu().numpy()
    kps, skts, _ = bones_to_pose(b, rest)
    rgbs, _, accs, _, _ = render_path(torch.tensor(np.repeat(c2w[None], a.frames, 0)), (H, W, np.full(a.frames, focal, np.float32)),
                                      4096, kw, kp=torch.tensor(kps.astype(np.float32)), skts=torch.tensor(skts.astype(np.float32)),
                                      white_bkgd=True, ret_acc=True, ext_scale=cfg.ext_scale)
    rgb8 = (rgbs * 255).astype(np.uint8)                                # run_gan.py:2327 (the PNG round trip is skipped)
    img = torch.tensor(rgb8[:, 100:412, 100:412]).permute(0, 3, 1, 2).float() / 255.0
    img = (img - torch.tensor(IMG_NORM_MEAN).view(1, 3, 1, 1)) / torch.tensor(IMG_NORM_STD).view(1, 3, 1, 1)
    return resize_antialiased(img, (224, 224)).to(dev), rgb8


out = {}
for name, fn in (("device", device_route), ("host", host_route)):
    res = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        res = fn()
    torch.cuda.synchronize()
    out[name] = (time.perf_counter() - t0) / a.reps, res
img_d, fr_d = out["device"][1]
img_h, fr_h = out["host"][1]
same = bool(np.array_equal(fr_d.cpu().numpy(), fr_h))
print(json.dumps({"workload": f"run_render call of the GAN loop: {a.frames} poses, {H}x{W}, surreal config, {a.prec}",
                  "device_route_ms_per_call": out["device"][0] * 1e3, "device_route_ms_per_frame": out["device"][0] * 1e3 / a.frames,
                  "host_route_ms_per_call": out["host"][0] * 1e3, "host_route_ms_per_frame": out["host"][0] * 1e3 / a.frames,
                  "uint8_frames_identical": same,
                  "max_abs_diff_regressor_input": float((img_d - img_h).abs().max())}))
