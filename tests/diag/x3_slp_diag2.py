"""Second diagnostic of the split-operand fault (see x3_slp_diag.py): the debug taps on the launch
size at which the fault shows (S = 80, 32768 rays), bf16x3 vs the fp32 kernel, twice."""
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
print("library:", os.environ.get("POSEGEN_HIP_LIB", "(default)"))
S, nr = 80, 32768
rbs = rb[100000:100000 + nr]
nf, z = r.stage_sample_coarse(rbs, cyl, S)
r.set_precision("fp32")
ref_raw = r.stage_eval(0, rbs, z, skts).clone()
for stage, what, width in ((10, "view cutoff weights wd", 24), (11, "view input values (first 16 units per half)", 256),
                           (7, "trunk output h7", 256), (9, "view layer output", 128)):
    r.set_precision("fp32")
    _, a = r.stage_eval(0, rbs, z, skts, want_dbg=True, dbg_stage=stage)
    a = a[:, :width].clone()
    r.set_precision("bf16x3")
    for rep in range(2):
        raw, o = r.stage_eval(0, rbs, z, skts, want_dbg=True, dbg_stage=stage)
        o = o[:, :width]
        d = (o - a).abs()
        dmax = d.amax(-1)
        thr = 1e-3 if stage != 11 else 1e-4
        bad = (dmax > thr).nonzero().reshape(-1)
        braw = ((raw - ref_raw).abs().amax(-1).reshape(-1) > 1e-3).nonzero().reshape(-1)
        print(f"tap {stage} ({what}) rep {rep}: max|x3-fp32| {float(dmax.max()):.3e}; points over {thr:g}: {len(bad)}; "
              f"raw-bad points in this launch: {len(braw)}", flush=True)
        if len(bad):
            b = bad[:6].cpu().numpy()
            ch = d[bad].gt(thr).sum(0).nonzero().reshape(-1).cpu().numpy()
            print(f"   bad points (pass, wave, lane, ray, sample): {[(int(x // 128), int((x // 32) % 4), int(x % 32), int(x // S), int(x % S)) for x in b]}")
            print(f"   channels over threshold ({len(ch)}): {ch[:64].tolist()}")
            p = int(bad[0])
            cc = d[p].gt(thr).nonzero().reshape(-1)[:8]
            print(f"   point {p}: x3 {o[p, cc].tolist()} fp32 {a[p, cc].tolist()}")
        if len(braw):
            both = np.intersect1d(braw.cpu().numpy(), bad.cpu().numpy())
            print(f"   raw-bad points also bad at this tap: {len(both)} of {len(braw)}")
        del raw, o, d
    del a
c.renderer.close()
