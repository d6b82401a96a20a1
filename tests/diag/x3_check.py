import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from posegen_amd import surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
from bench import full_frame_rays
dev = torch.device("cuda:0")
cfg = surreal_config()
c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision="fp32")
rb, skts, cyl, *_ = full_frame_rays(512, 512, dev)
r = c.renderer
ref = r.render_rays(rb, skts, cyl, want_alpha=False)
r.set_precision("bf16x3")
outs = []
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = r.render_rays(rb, skts, cyl, want_alpha=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    outs.append(o)
    d = {k: float((o[k] - ref[k]).abs().max()) for k in ("rgb_map", "acc_map")}
    same = all(torch.equal(o[k], outs[0][k]) for k in ("rgb_map", "acc_map", "disp_map"))
    print(f"run {i}: {dt*1e3:.1f} ms  {rb.shape[0]/dt/1e6:.2f} M rays/s  max|d| vs fp32 kernel {d}  bitwise==run0 {same}")
