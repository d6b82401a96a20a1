import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from posegen_amd import surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
from bench import full_frame_rays
dev = torch.device("cuda:0")
cfg = surreal_config()
c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision="bf16")
rb, skts, cyl, *_ = full_frame_rays(512, 512, dev)
r = c.renderer
nf, z = r.stage_sample_coarse(rb, cyl, 64)
for rep in range(2):
    raw, dbg = r.stage_eval(0, rb, z, skts, want_dbg=True, dbg_stage=99)
torch.cuda.synchronize()
full = dbg.view(torch.int64).cpu().numpy().reshape(-1)[: 64 * 8 * 16].reshape(64, 8, 16)
st = full[:, :, :9]
print("per pass per wave: cycles waiting in vmcnt (weight DMA) %.0f, in s_barrier %.0f" % (full[1:, :, 9].mean(), full[1:, :, 10].mean()))
print("   by wave: vmcnt", full[1:, :, 9].mean(0).astype(int).tolist(), " barrier", full[1:, :, 10].mean(0).astype(int).tolist())
d = np.diff(st, axis=-1).astype(np.float64)     # [it, wave, 8 segments]
names = ["ray table+Y stage", "L0 (x)", "L1-4", "L5 (h+x)", "L6-7", "alpha tile", "view (trunk+Y)", "rgb+store"]
tot = (st[:, :, 8] - st[:, :, 0]).astype(np.float64)
print("pass total cycles (s_memtime ticks): mean %.0f  min %.0f max %.0f" % (tot.mean(), tot.min(), tot.max()))
mf = [24, 216, 512, 344, 256, 16, 72, 8]
if True:     # pg_eval16r.hip, in 32x32x16 equivalents
    names = ["pass prologue", "L0 (x)", "L1-4", "L5 (h+x)", "L6-7", "alpha tile", "view (trunk+Y)", "rgb+store"]
    mf = [0, 224, 512, 352, 256, 8, 72, 4]
for k, nme in enumerate(names):
    m = d[:, :, k].mean()
    print(f"{nme:18s} {m:9.0f} cycles  {100*m/tot.mean():5.1f}%   mfma {mf[k]:4d} -> {m/max(mf[k],1):6.1f} cyc/mfma (ideal 64 for 2 waves/SIMD)")
