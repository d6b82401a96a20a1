"""Where does the compensated-fp16 kernel's layer-0 pre-activation differ from the oracle's emulation?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import anerf_oracle as orc
from posegen_amd.raycaster import HipRayCaster
from tests.helpers import cfg_from_golden, load_golden, model_for, oracle_cfg, torch_weights
g = load_golden("rays_surreal"); cfg = cfg_from_golden(g)
wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
c = HipRayCaster.from_weights(cfg, wc, wf, tv, td, device="cuda:0", precision="fp16c")
rb, skts = torch.tensor(g["ray_batch"]), torch.tensor(g["skts"])
z = torch.tensor(g["z_coarse"]); n, S = z.shape
ocfg = oracle_cfg(cfg, tv, td)
pts = rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None]
x = orc.embed_points(pts, rb[:, 3:6], skts, ocfg).reshape(n * S, -1)
W0, b0 = torch.tensor(wc["pts_linears.0.weight"]), torch.tensor(wc["pts_linears.0.bias"])
exact = (x[:, :432].double() @ W0.double().T + b0.double())
emu = orc._linear(x[:, :432], W0, b0, "fp16c")
for prec in ("fp32", "fp16c", "bf16x3"):
    c.renderer.set_precision(prec)
    raw, dbg = c.renderer.stage_eval(0, rb, z, skts, want_dbg=True)
    d = (dbg.cpu().double() - exact).abs()
    i = int(d.argmax()); p, ch = i // 256, i % 256
    print(f"{prec}: max |pre0 - exact| {float(d.max()):.3e} at point {p} channel {ch} (value {float(exact[p, ch]):.4f}); "
          f"per-channel max: ch0 {float(d[:, 0].max()):.3e}, others {float(d[:, 1:].max()):.3e}; rms {float(d.pow(2).mean().sqrt()):.3e}")
de = (emu.double() - exact).abs()
print(f"emulation: max |emu - exact| {float(de.max()):.3e}; ch0 {float(de[:, 0].max()):.3e}, others {float(de[:, 1:].max()):.3e}")
c.renderer.set_precision("fp16c")
raw, dbg = c.renderer.stage_eval(0, rb, z, skts, want_dbg=True)
dd = (dbg.cpu() - emu).abs()
i = int(dd.argmax()); p, ch = i // 256, i % 256
print(f"kernel vs emulation: max {float(dd.max()):.3e} at point {p} ch {ch}: kernel {float(dbg[p, ch]):.6f} emu {float(emu[p, ch]):.6f} exact {float(exact[p, ch]):.6f}")
print("   the 24 cos*w inputs of that point:", x[p, 48:72].tolist())
c.renderer.close()
