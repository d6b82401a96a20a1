"""Compensated-fp16 product of single inputs: layer-0 rows that select one input column each (weight 1.0)
or scale it (weight 0.37): pre0[:, r] = w * x_col, against the exact value."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import anerf_oracle as orc
from posegen_amd.raycaster import HipRayCaster
from tests.helpers import cfg_from_golden, load_golden, model_for, oracle_cfg
g = load_golden("rays_surreal"); cfg = cfg_from_golden(g)
wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
wc = {k: v.copy() for k, v in wc.items()}
W0 = wc["pts_linears.0.weight"]; W0[:] = 0; wc["pts_linears.0.bias"][:] = 0
cols = [48 + 7, 48 + 8, 48 + 11, 24 + 8, 0 + 8, 360 + 24, 72 + 8, 336 + 8]      # cos0 j7, j8, j11; sin0 j8; v*w j8; r_x j8; sin1 j8; cos6 j8
for r, cidx in enumerate(cols):
    W0[r, cidx] = 1.0
    W0[32 + r, cidx] = 0.37
c = HipRayCaster.from_weights(cfg, wc, wf, tv, td, device="cuda:0", precision="fp16c")
rb, skts = torch.tensor(g["ray_batch"]), torch.tensor(g["skts"])
z = torch.tensor(g["z_coarse"]); n, S = z.shape
ocfg = oracle_cfg(cfg, tv, td)
pts = rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None]
x = orc.embed_points(pts, rb[:, 3:6], skts, ocfg).reshape(n * S, -1)
for prec in ("fp32", "fp16c"):
    c.renderer.set_precision(prec)
    raw, dbg = c.renderer.stage_eval(0, rb, z, skts, want_dbg=True)
    dbg = dbg.cpu()
    for r, cidx in enumerate(cols):
        xe = x[:, cidx].double()
        for rr, w in ((r, 1.0), (32 + r, 0.37)):
            d = dbg[:, rr].double() - np.float64(np.float32(w)) * xe
            i = int(d.abs().argmax())
            print(f"{prec} col {cidx} w {w}: max err {float(d.abs().max()):.3e} (x = {float(xe[i]):.7f}, got {float(dbg[i, rr]):.7f}); "
                  f"rel to fp16 ulp of x: {float((d.abs() / (xe.abs() * 2**-11 + 1e-30))[xe.abs() > 1e-3].max()):.3f}")
c.renderer.close()
