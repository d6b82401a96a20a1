"""Where a pass of the compensated-fp16 kernel (pg_evalc.hip, record variant) spends its time: s_memtime stamps of a
-DPG_STAMPS build (tools/build_variant.sh stamps_c -DPG_STAMPS with FILE=pg_evalc.hip; POSEGEN_LIB points at it)."""
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
full = dbg.view(torch.int64).cpu().numpy().reshape(-1)[: 64 * 8 * 16].reshape(64, 8, 16)[:, :4]
st = full[:, :, :10]
print("per pass per wave: cycles waiting in vmcnt %.0f, in s_barrier %.0f" % (full[1:, :, 10].mean(), full[1:, :, 11].mean()))
print("   by wave: vmcnt", full[1:, :, 10].mean(0).astype(int).tolist(), " barrier", full[1:, :, 11].mean(0).astype(int).tolist())
d = np.diff(st, axis=-1).astype(np.float64)[1:]
tot = (st[1:, :, 9] - st[1:, :, 0]).astype(np.float64)
print("pass total (s_memtime ticks): mean %.0f  min %.0f max %.0f" % (tot.mean(), tot.min(), tot.max()))
names = ["pass prologue", "L0 (x)", "L1-4", "L5 (h)", "L5 (x)", "L6-7", "alpha + view trunk", "view directions", "rgb + store"]
mf = [0, 432, 1024, 256, 432, 512, 160, 16, 16]
for k, nme in enumerate(names):
    m = d[:, :, k].mean()
    print(f"{nme:20s} {m:9.0f} ticks  {100*m/tot.mean():5.1f}%   mfma {mf[k]:4d} -> {m/max(mf[k],1):6.1f} ticks/mfma")
