import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import anerf_oracle as orc
from posegen_amd.raycaster import HipRayCaster
from tests.helpers import cfg_from_golden, load_golden, model_for, oracle_cfg, torch_weights
g = load_golden("rays_surreal"); cfg = cfg_from_golden(g)
wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
c = HipRayCaster.from_weights(cfg, wc, wf, tv, td, device="cuda:0", precision=1)
rb, skts = torch.tensor(g["ray_batch"]), torch.tensor(g["skts"])
z = torch.tensor(g["z_coarse"]); n, S = z.shape
pts = rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None]
ocfg = oracle_cfg(cfg, tv, td); ocfg.quant = "bf16"
x = orc.embed_points(pts, rb[:, 3:6], skts, ocfg)
ref = orc.mlp_forward(x.reshape(n * S, -1), torch_weights(wc), ocfg).reshape(n, S, 4)
r1 = c.renderer.stage_eval(0, rb, z, skts).cpu()
r2 = c.renderer.stage_eval(0, rb, z, skts).cpu()
d = (r1 - ref).abs()[..., :3].amax(-1)
bad = d > 5e-2
flat = bad.flatten().nonzero().flatten().numpy()
print("deterministic", torch.equal(r1, r2), "bad", len(flat), "of", n * S)
print("lane (flat%32):", np.bincount(flat % 32, minlength=32).tolist())
print("wave ((flat//32)%8):", np.bincount((flat // 32) % 8, minlength=8).tolist())
print("pass (flat//256) first 20:", np.bincount(flat // 256, minlength=64).tolist()[:64])
print("per-s:", bad.sum(0).tolist())
