"""Fourth diagnostic: the view table as the lanes find it in LDS at the end of a pass (tap 16 =
table floats 256..319 of the lane's half), at the points where bf16x3 (SLP build) goes wrong."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from posegen_amd import surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
from bench import full_frame_rays
dev = torch.device("cuda:0")
cfg = surreal_config()
c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision="fp32")
r = c.renderer
rb, skts, cyl, *_ = full_frame_rays(512, 512, dev)
S, nr = 80, 32768
rbs = rb[100000:100000 + nr]
nf, z = r.stage_sample_coarse(rbs, cyl, S)
r.set_precision("fp32")
ref_raw, a = r.stage_eval(0, rbs, z, skts, want_dbg=True, dbg_stage=16)
a = a[:, :128].clone(); ref_raw = ref_raw.clone()
r.set_precision("bf16x3")
for rep in range(2):
    raw, o = r.stage_eval(0, rbs, z, skts, want_dbg=True, dbg_stage=16)
    o = o[:, :128]
    bad = ((raw - ref_raw).abs().amax(-1).reshape(-1) > 1e-3).nonzero().reshape(-1)
    dt = (o - a).abs()
    tb = (dt.amax(-1) > 1e-6).nonzero().reshape(-1)
    print(f"rep {rep}: raw-bad points {len(bad)}; points whose end-of-pass table differs from the fp32 kernel's: {len(tb)}")
    if len(bad):
        p = int(bad[0])
        print("   bad point", p, "lane", p % 32, ": table[h=1][256..263] x3", o[p, 64:72].tolist(), "fp32", a[p, 64:72].tolist())
        print("   table entries differing at bad points:", dt[bad].gt(1e-6).sum(0).nonzero().reshape(-1).tolist())
    if len(tb):
        print("   differing points (pass, wave, lane):", [(int(x // 128), int((x // 32) % 4), int(x % 32)) for x in tb[:10].cpu().numpy()])
c.renderer.close()
