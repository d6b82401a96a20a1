"""Counts the points where bf16x3 differs from the fp32 kernel by > 1e-3 (S = 80, 32768 rays): the
reproducer of the round-1 split-operand fault, for A/B runs over experimental libraries."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from posegen_amd import surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
from bench import full_frame_rays
dev = torch.device("cuda:0")
cfg = surreal_config()
c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision="fp32")
r = c.renderer
rb, skts, cyl, *_ = full_frame_rays(512, 512, dev)
rbs = rb[100000:100000 + 32768]
nf, z = r.stage_sample_coarse(rbs, cyl, 80)
ref = r.stage_eval(0, rbs, z, skts).clone()
r.set_precision("bf16x3")
res = []
for rep in range(4):
    raw = r.stage_eval(0, rbs, z, skts)
    res.append(int(((raw - ref).abs().amax(-1) > 1e-3).sum()))
print(os.path.basename(os.environ.get("POSEGEN_HIP_LIB", "default")), "bad points per run:", res, flush=True)
c.renderer.close()
