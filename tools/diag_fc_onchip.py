"""frame codes in the on-chip variant of pg_eval16r.hip: raw values on the golden h36m rays against the fp32 kernel, per channel and per ray"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from posegen_amd import PREC_FP16, PREC_FP32, PREC_BF16
from posegen_amd.raycaster import HipRayCaster
from tests.helpers import cfg_from_golden, load_golden, model_for

g = load_golden("rays_h36m")
cfg = cfg_from_golden(g)
wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
rb, skts = torch.tensor(g["ray_batch"]), torch.tensor(g["skts"])
cams = torch.tensor(g["cams"])
z = torch.tensor(g["z_coarse"])
n, S = z.shape
print("n", n, "S", S, "cams", cams[:8].tolist(), flush=True)
c = HipRayCaster.from_weights(cfg, wc, wf, tv, td, device="cuda:0", precision=PREC_FP32)
ref = c.renderer.stage_eval(0, rb, z, skts, cams=cams).cpu()
for prec in (PREC_FP16, PREC_BF16):
    c.renderer.set_precision(prec)
    for m in (n, 37, 3):
        got = c.renderer.stage_eval(0, rb[:m], z[:m], skts, cams=cams[:m]).cpu()
        e = (got - ref[:m]).abs()
        print(f"prec {prec} rays {m}: per-channel max err", e.reshape(-1, 4).max(0).values.tolist(), "finite", bool(torch.isfinite(got).all()), flush=True)
        per_ray = e.reshape(m, -1).max(1).values
        bad = (per_ray > 0.1).nonzero().flatten().tolist()
        print("   rays off by > 0.1:", len(bad), bad[:20], flush=True)
        if bad:
            r = bad[0]
            print("   ray", r, "per-sample err", e[r].max(1).values[:16].tolist())
    got = c.renderer.stage_eval(0, rb, z, skts, cams=None).cpu()
    c.renderer.set_precision(PREC_FP32)
    ref0 = c.renderer.stage_eval(0, rb, z, skts, cams=None).cpu()
    print(f"prec {prec} no cams: max err", float((got - ref0).abs().max()), flush=True)
