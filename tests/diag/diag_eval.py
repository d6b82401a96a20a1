"""GPU diagnostic: per-precision error statistics of the fused embed+MLP stage."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import anerf_oracle as orc
from posegen_amd import PREC_NAMES
from posegen_amd.raycaster import HipRayCaster
from tests.helpers import cfg_from_golden, load_golden, model_for, oracle_cfg, torch_weights

g = load_golden("rays_surreal"); cfg = cfg_from_golden(g)
wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
c = HipRayCaster.from_weights(cfg, wc, wf, tv, td, device="cuda:0", precision=0)
rb, skts = torch.tensor(g["ray_batch"]), torch.tensor(g["skts"])
z = torch.tensor(g["z_coarse"]); n, S = z.shape
pts = rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None]
for prec, quant in ((0, None), (1, "bf16"), (3, "fp16")):
    c.renderer.set_precision(prec)
    raw = c.renderer.stage_eval(0, rb, z, skts).cpu()
    for q in (None, quant):
        ocfg = oracle_cfg(cfg, tv, td); ocfg.quant = q
        x = orc.embed_points(pts, rb[:, 3:6], skts, ocfg)
        ref = orc.mlp_forward(x.reshape(n * S, -1), torch_weights(wc), ocfg).reshape(n, S, 4)
        d = (raw - ref).abs()
        i = int(d.argmax()); r, s, ch = np.unravel_index(i, d.shape)
        print(f"{PREC_NAMES[prec]:7s} vs oracle[{q}]: max {d.max():.3e} at ray {r} s {s} ch {ch} (gpu {raw[r,s,ch]:.5f} ref {ref[r,s,ch]:.5f}); per-ch max {[f'{v:.2e}' for v in d.amax((0,1)).tolist()]}; nan {int(torch.isnan(raw).sum())}")
    h = x.reshape(n*S, -1)[:, :432]
    print("   max |x|", float(x.abs().max()))
