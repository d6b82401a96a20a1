"""Kernel time of the compensated-fp16 eval launch (coarse net, 512x512 x 64 samples) for A/B runs over libraries."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posegen_amd import surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
from bench import full_frame_rays
dev = torch.device("cuda:0")
cfg = surreal_config()
c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision=os.environ.get("PREC", "fp16c"))
rb, skts, cyl, *_ = full_frame_rays(512, 512, dev)
r = c.renderer
S = int(os.environ.get("S", "64"))
nf, z = r.stage_sample_coarse(rb, cyl, S)
r.stage_eval(0, rb, z, skts)
torch.cuda.synchronize()
r.profile_enable(True); r.profile_read()
for _ in range(5):
    r.stage_eval(0, rb, z, skts)
n, ms, pts = r.profile_read()
print(os.path.basename(os.environ.get("POSEGEN_HIP_LIB", "default")), f"S={S}: {ms / n:.2f} ms per launch, {pts / n / (ms / n) / 1e6:.2f} G points/s, "
      f"{pts * cfg.flops_per_point() / (ms * 1e-3) / 1e12:.0f} TFLOP/s algorithmic", flush=True)
