"""Randomised frame-level parity on the GPU: posegen_amd.render.render_path (bbox cull, chunk groups, background
composite, frame codes) against the oracle's render_path on random poses and jittered cameras.  (The all-device
route -- pg_pose_kinematics + pg_pose_boxes -- is held bitwise to this one by tests/test_gpu_frames.py.)

    python tools/frame_sweep.py [--cases 6]

Test infrastructure: uses oracle/.
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import anerf_oracle as orc
from posegen_amd import h36m_config, surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
from posegen_amd.render import render_path
from tests.helpers import oracle_cfg, torch_weights

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=6)
a = ap.parse_args()
rng = np.random.RandomState(11)
dev = "cuda:0"
worst = 0.0
t0 = time.time()
for case in range(a.cases):
    fc = bool(case % 2)
    cfg = (h36m_config if fc else surreal_config)(n_samples=int(rng.choice([32, 64])), n_importance=int(rng.choice([0, 16])))
    H = W = int(rng.choice([48, 64, 80]))
    F = 3
    chunk = int(rng.choice([256, 1000, 4096]))
    wc, wf, tv, td = syn.make_model(cfg, int(rng.randint(0, 1000)))
    _, kps, skts = syn.make_pose(F, int(rng.randint(0, 1000)))
    c2ws, focals = syn.make_camera(F, H, W)
    c2ws[:, :3, 3] += rng.uniform(-0.2, 0.2, size=(F, 3)).astype(np.float32)
    cams = torch.tensor(rng.randint(0, cfg.n_framecodes, size=F).astype(np.float32)) if fc else None
    white = bool(rng.randint(0, 2))
    ref = orc.render_path(c2ws, H, W, focals, chunk, oracle_cfg(cfg, tv, td), torch_weights(wc), torch_weights(wf), kps, skts,
                          cfg.n_samples, cfg.n_importance, cfg.ext_scale, cams=cams, white_bkgd=white)
    c = HipRayCaster.from_weights(cfg, wc, wf, tv, td, device=dev, precision="fp32")
    kw = {"ray_caster": c, "N_samples": cfg.n_samples, "N_importance": cfg.n_importance}
    got = render_path(torch.tensor(c2ws), (H, W, focals), chunk, kw, kp=torch.tensor(kps), skts=torch.tensor(skts), cams=cams,
                      white_bkgd=white, ret_acc=True, ext_scale=cfg.ext_scale)
    e = [float(np.abs(got[k] - ref[k]).max()) for k in range(3)]
    same_boxes = all(np.array_equal(np.asarray(got[4][i]), np.asarray(ref[4][i])) for i in range(F))
    same_ids = all(np.array_equal(np.asarray(got[3][i]), np.asarray(ref[3][i])) for i in range(F))
    worst = max(worst, e[0], e[2])
    print(f"case {case} fc={int(fc)} {H}x{W} S={cfg.n_samples} N={cfg.n_importance} chunk={chunk} white={int(white)} "
          f"valid={[len(v) for v in ref[3]]}: rgb {e[0]:.1e} disp {e[1]:.1e} acc {e[2]:.1e} boxes_equal={same_boxes} ids_equal={same_ids}", flush=True)
    c.renderer.close()
    if not (same_boxes and same_ids):
        sys.exit(2)
print(f"# {a.cases} cases in {time.time() - t0:.0f} s; worst rgb|acc {worst:.2e}")
sys.exit(1 if worst > 3e-4 else 0)
