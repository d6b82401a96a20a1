import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from posegen_amd import PREC_NAMES
from posegen_amd.raycaster import HipRayCaster
from tests.helpers import cfg_from_golden, load_golden, model_for

for name in ("rays_surreal", "rays_h36m"):
    g = load_golden(name); cfg = cfg_from_golden(g)
    wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
    c = HipRayCaster.from_weights(cfg, wc, wf, tv, td, device="cuda:0", precision=0)
    rb, skts = torch.tensor(g["ray_batch"]), torch.tensor(g["skts"])
    cams = torch.tensor(g["cams"]) if "cams" in g else None
    z = torch.tensor(g["z_coarse"]); n, S = z.shape
    ref = {}
    c.renderer.set_precision(0)
    for st in (8, 9, 10, 11):
        raw, dbg = c.renderer.stage_eval(0, rb, z, skts, cams=cams, want_dbg=True, dbg_stage=st)
        ref[st] = dbg.cpu(); ref['raw'] = raw.cpu()
    for prec in (2, 4):
        c.renderer.set_precision(prec)
        for rep in range(3):
            line = f"{name} {PREC_NAMES[prec]} rep{rep}:"
            for st in (8, 9, 10, 11):
                raw, dbg = c.renderer.stage_eval(0, rb, z, skts, cams=cams, want_dbg=True, dbg_stage=st)
                d = (dbg.cpu() - ref[st]).abs().amax(1)
                bad = (d > 1e-3).nonzero().flatten()
                dr = (raw.cpu() - ref['raw']).abs().amax(-1).flatten()
                badr = (dr > 1e-3).nonzero().flatten()
                line += f" st{st}:{float(d.max()):.1e}/{len(bad)}(raw {len(badr)})"
                if len(bad) and st in (8, 9, 10, 11):
                    p = int(bad[0])
                    dd = (dbg.cpu()[p] - ref[st][p]).abs()
                    ch = (dd > 1e-3).nonzero().flatten().tolist()
                    line += f"[pt {p}=ray {p//S} s {p%S} ch {ch[:12]}{'...' if len(ch)>12 else ''} n={len(ch)}]"
            print(line)
    c.renderer.close()
