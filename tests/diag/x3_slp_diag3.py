"""Third diagnostic of the split-operand fault: dump the view-layer outputs (tap 9), cutoff weights
(tap 10) and view inputs (tap 11) of bad points for offline attribution to an input channel."""
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
out = {}
r.set_precision("fp32")
_, a9 = r.stage_eval(0, rbs, z, skts, want_dbg=True, dbg_stage=9)
a9 = a9[:, :128].clone()
r.set_precision("bf16x3")
_, o9 = r.stage_eval(0, rbs, z, skts, want_dbg=True, dbg_stage=9)
o9 = o9[:, :128].clone()
bad = ((o9 - a9).abs().amax(-1) > 1e-3).nonzero().reshape(-1)
print("bad points:", len(bad))
sel = bad[:256]
near = torch.unique(torch.cat([sel - 3, sel - 2, sel - 1, sel + 1, sel + 2, sel + 3]).clamp(0, nr * S - 1))[:512]
for name, idx in (("bad", sel), ("near", near)):
    out[name + "_idx"] = idx.cpu().numpy()
    out[name + "_x3_9"] = o9[idx].cpu().numpy()
    out[name + "_f32_9"] = a9[idx].cpu().numpy()
del a9, o9
r.set_precision("fp32")
for stage, width in ((10, 24), (11, 256)):
    _, t = r.stage_eval(0, rbs, z, skts, want_dbg=True, dbg_stage=stage)
    for name in ("bad", "near"):
        out[f"{name}_f32_{stage}"] = t[torch.as_tensor(out[name + "_idx"], device=dev), :width].cpu().numpy()
    del t
out["z"] = z.reshape(-1)[sel].cpu().numpy()
out["rays"] = rbs[(sel // S)].cpu().numpy()
out["skts"] = skts.cpu().numpy()
np.savez_compressed("gpurun_out/r2_x3_bad.npz", **out)
c.renderer.close()
