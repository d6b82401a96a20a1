import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import anerf_oracle as orc
from posegen_amd import PREC_NAMES
from posegen_amd.raycaster import HipRayCaster
from tests.helpers import cfg_from_golden, load_golden, model_for, oracle_cfg, torch_weights

for name in ("rays_surreal", "rays_h36m"):
    g = load_golden(name); cfg = cfg_from_golden(g)
    wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
    c = HipRayCaster.from_weights(cfg, wc, wf, tv, td, device="cuda:0", precision=0)
    rb, skts = torch.tensor(g["ray_batch"]), torch.tensor(g["skts"])
    cams = torch.tensor(g["cams"]) if "cams" in g else None
    z = torch.tensor(g["z_coarse"]); n, S = z.shape
    pts = rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None]
    ocfg = oracle_cfg(cfg, tv, td)
    x = orc.embed_points(pts, rb[:, 3:6], skts, ocfg, cams)
    ref = orc.mlp_forward(x.reshape(n * S, -1), torch_weights(wc), ocfg).reshape(n, S, 4)
    for prec in (0, 2, 4):
        c.renderer.set_precision(prec)
        r1 = c.renderer.stage_eval(0, rb, z, skts, cams=cams).cpu()
        r2 = c.renderer.stage_eval(0, rb, z, skts, cams=cams).cpu()
        d = (r1 - ref).abs()[..., :3].amax(-1)
        bad = (d > 1e-3)
        idx = bad.nonzero()
        print(f"{name} {PREC_NAMES[prec]}: deterministic={torch.equal(r1, r2)} bad pts {int(bad.sum())}/{n*S}; max {float(d.max()):.3e}")
        if len(idx):
            flat = (idx[:, 0] * S + idx[:, 1]).numpy()
            print("   flat idx mod 32:", np.bincount(flat % 32, minlength=32).tolist())
            print("   (flat//32) mod 4 (wave):", np.bincount((flat // 32) % 4, minlength=4).tolist())
            print("   first bad:", idx[:10].tolist())
            # how do errors distribute over s?
            print("   per-s count:", bad.sum(0).tolist())
    c.renderer.close()
