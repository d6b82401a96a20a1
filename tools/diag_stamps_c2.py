"""Where a pass of pg_evalc2.hip spends its time: s_memtime stamps of a -DPG_STAMPS build
(FILE=pg_evalc2.hip tools/build_variant.sh stamps_c2 -DPG_STAMPS; POSEGEN_HIP_LIB points at it)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from posegen_amd import surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
from bench import full_frame_rays
dev = torch.device("cuda:0")
cfg = surreal_config()
c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision="fp16c")
rb, skts, cyl, *_ = full_frame_rays(512, 512, dev)
r = c.renderer
S = int(os.environ.get("S", "64"))
nf, z = r.stage_sample_coarse(rb, cyl, S)
for rep in range(2):
    raw, dbg = r.stage_eval(0, rb, z, skts, want_dbg=True, dbg_stage=99)
torch.cuda.synchronize()
n_wg = 256
full = dbg.view(torch.int64).cpu().numpy().reshape(-1)[: n_wg * 8 * 16].reshape(n_wg, 8, 16).astype(np.float64)
passes = full[:, :, 11]
per = full[:, :, :11] / passes[:, :, None]
tot = per.sum(-1)
print(f"S={S}: passes per workgroup {passes.mean():.1f}; ticks per pass: mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}")
names = ["prologue + masks + B1", "L0 x phase", "next (a,b) rows", "layers: a0 a1 (+ beta's conversion)", "layers: barrier X + bias", "layers: retire + barrier Z",
         "layers: b0 a2 b1 a3 | b2 b3 (+ alpha's conversion)", "L5 x phase + run prologues", "alpha + view trunk (with beta's h7 conversion)", "tail: zero, weights, Y", "second stage, G, rgb, store"]
mf = [0, 0, 0, 7 * 64, 0, 0, 7 * 192, 0, 144, 0, 0]
for k, nme in enumerate(names):
    m = per[:, :, k].mean()
    extra = f"   {m / mf[k]:6.1f} ticks per MFMA of the wave ({mf[k]})" if mf[k] else ""
    print(f"{nme:40s} {m:9.0f} ticks  {100 * m / tot.mean():5.1f}%   waves 0-3 {per[:, :4, k].mean():8.0f}  waves 4-7 {per[:, 4:, k].mean():8.0f}{extra}")
