"""Frame-level timing of render_path (pose tensors on host -> frames on host), culled
variant: 512x512 frames, reference bounding-box cull, surreal config, bf16.  Compares the
device frame front/back end (pg_render_frame, what render_path uses) with the ray-level route
(rays built on the host, copied, rendered, scattered with torch ops)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from posegen_amd import surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
from posegen_amd.rays import kp_to_valid_rays
from posegen_amd.render import render, render_path

F, H, W = 8, 512, 512
dev = torch.device("cuda:0")
cfg = surreal_config()
caster = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision="bf16")
_, kps, skts = syn.make_pose(F, 1)
c2ws, focals = syn.make_camera(F, H, W)
kps, skts, c2ws = torch.tensor(kps), torch.tensor(skts), torch.tensor(c2ws)
kw = {"ray_caster": caster, "perturb": False, "N_importance": cfg.n_importance, "N_samples": cfg.n_samples,
      "use_viewdirs": True, "raw_noise_std": 0., "ray_noise_std": 0., "ext_scale": cfg.ext_scale,
      "preproc_kwargs": {}, "lindisp": False, "nerf_type": "nerf"}


def new_route():
    return render_path(c2ws, (H, W, focals), 4096, kw, kp=kps, skts=skts, white_bkgd=True, ret_acc=True,
                       ext_scale=cfg.ext_scale)


def ray_route():
    rays, vids, cyls, boxes = kp_to_valid_rays(c2ws, H, W, focals, kps=kps, ext_scale=cfg.ext_scale)
    out = []
    for i in range(F):
        ret = render(H, W, focals, rays=rays[i], chunk=4096, kp_batch=kps[i:i + 1], skts=skts[i:i + 1],
                     cyls=cyls[i:i + 1], cams=None, subject_idxs=None, bones=None, want_alpha=False, **kw)
        img = torch.ones(H * W, 3, device=dev)
        vid = vids[i].to(dev)
        img[vid] = ret["rgb_map"] + (1. - ret["acc_map"][..., None]) * img[vid]
        out.append(img.view(H, W, 3))
    return torch.stack(out).cpu().numpy(), vids


rgbs, disps, accs, vids, boxes = new_route()
n_valid = sum(len(v) for v in vids)
for name, fn in (("device frame path (render_path)", new_route), ("ray-level route (host rays + torch scatter)", ray_route)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name:46s}: {F} frames {H}x{W}, {n_valid} valid rays: {dt * 1e3 / F:7.2f} ms/frame, "
          f"{n_valid / dt / 1e6:.2f} M valid rays/s, {F * H * W / dt / 1e6:.2f} M pixels/s")
ref, _ = ray_route()
print("max |rgb(device path) - rgb(ray route)| =", float(np.abs(rgbs - ref).max()))
