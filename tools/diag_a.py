import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from posegen_amd.raycaster import HipRayCaster
from tests.helpers import cfg_from_golden, load_golden, model_for
g = load_golden("rays_surreal"); cfg = cfg_from_golden(g)
wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
c = HipRayCaster.from_weights(cfg, wc, wf, tv, td, device="cuda:0", precision=0)
rb, skts = torch.tensor(g["ray_batch"]), torch.tensor(g["skts"])
z = torch.tensor(g["z_coarse"]); n, S = z.shape
ref = {}
for st in (7, 8, 9):
    raw, dbg = c.renderer.stage_eval(0, rb, z, skts, want_dbg=True, dbg_stage=st); ref[st] = dbg.cpu(); ref['raw'] = raw.cpu()
c.renderer.set_precision(1)
for st in (7, 8, 9):
    raw, dbg = c.renderer.stage_eval(0, rb, z, skts, want_dbg=True, dbg_stage=st)
    d = (dbg.cpu() - ref[st]).abs()
    nch = 128 if st == 9 else 256
    print(f"stage {st}: max diff {float(d[:, :nch].max()):.3e}; per-32ch-tile max {[f'{float(d[:, 32*t:32*t+32].max()):.2e}' for t in range(nch//32)]}; ref max {float(ref[st][:, :nch].abs().max()):.3f}")
dr = (raw.cpu() - ref['raw']).abs()
print("raw per-ch max", dr.amax((0,1)).tolist())
