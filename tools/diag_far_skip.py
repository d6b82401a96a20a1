import sys, os
sys.path.insert(0, os.getcwd())
import torch
from posegen_amd import surreal_config, synthetic as syn, PREC_BY_NAME
from posegen_amd.raycaster import HipRayCaster
from bench import full_frame_rays
dev = torch.device("cuda:0")
cfg = surreal_config()
c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision="bf16")
r = c.renderer
rb, skts, cyl, *_ = full_frame_rays(512, 512, dev)
for prec in ("bf16", "fp16", "fp16c"):
    r.set_precision(PREC_BY_NAME[prec])
    nf, z = r.stage_sample_coarse(rb, cyl, 64)
    r.set_far_skip(True); ra = r.stage_eval(0, rb, z, skts)
    r.set_far_skip(False); rbb = r.stage_eval(0, rb, z, skts)
    d = (ra - rbb).abs()
    print(prec, "stage_eval raw: max", float(d.max()), "mean", float(d.mean()), "frac points differing", float((d.amax(-1) > 0).float().mean()), "scale", float(ra.abs().mean()))
    r.set_far_skip(True); a = r.render_rays(rb, skts, cyl, want_alpha=False)
    r.set_far_skip(False); b = r.render_rays(rb, skts, cyl, want_alpha=False)
    for k in ("rgb_map", "acc_map", "rgb0", "acc0"):
        dd = (a[k] - b[k]).abs()
        print("   ", k, "max", float(dd.max()), "mean", float(dd.mean()), "frac rays differing", float((dd.reshape(dd.shape[0], -1).amax(-1) > 0).float().mean()))
r.set_far_skip(True)
