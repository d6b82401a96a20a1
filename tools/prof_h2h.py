"""Where the host time of bench.py's host_to_host (render_path: poses on the host -> frames on the host) goes."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posegen_amd import surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
from posegen_amd.render import render_path
dev = "cuda:0"
cfg = surreal_config()
caster = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision="bf16")
frames, H, W = 8, 512, 512
_, kps, skts = syn.make_pose(frames, 1)
c2ws, focals = syn.make_camera(frames, H, W)
kps, skts, c2ws = torch.tensor(kps), torch.tensor(skts), torch.tensor(c2ws)
kw = {"ray_caster": caster, "N_importance": cfg.n_importance, "N_samples": cfg.n_samples, "lindisp": False}
run = lambda: render_path(c2ws, (H, W, focals), 4096, kw, kp=kps, skts=skts, white_bkgd=True, ret_acc=True, ext_scale=cfg.ext_scale)
out = run(); torch.cuda.synchronize()
nv = sum(len(v) for v in out[3])
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"ms per frame {dt * 1e3 / frames:.3f}, valid rays/s {nv / dt:.0f}")
pr = cProfile.Profile(); pr.enable(); run(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
