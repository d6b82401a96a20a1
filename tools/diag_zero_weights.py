"""Does the clock explain a timing ablation?  The bf16 eval launch on the benchmark frame with (a) the synthetic
weights, (b) ALL-ZERO weights (same instruction stream, same LDS / DMA traffic, but MFMA operands that never toggle).
MI355X lowers its clock under MFMA load by what the data costs in power (MI355X_MICROARCH.md, DVFS give-back 1),
so (b) says how much of a 'wrong results, timing only' ablation that feeds constant operands is the chip, not the code."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from posegen_amd import surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
from bench import full_frame_rays
dev = torch.device("cuda:0")
cfg = surreal_config()
wc, wf, tv, td = syn.make_model(cfg, 0)
rb, skts, cyl, *_ = full_frame_rays(512, 512, dev)
def run(name, w):
    c = HipRayCaster.from_weights(cfg, w, w, tv, td, device=dev, precision="bf16")
    r = c.renderer
    nf, z = r.stage_sample_coarse(rb, cyl, 64)
    for _ in range(2):
        r.stage_eval(0, rb, z, skts)
    torch.cuda.synchronize()
    r.profile_enable(True); r.profile_read()
    for _ in range(6):
        r.stage_eval(0, rb, z, skts)
    n, ms, pts = r.profile_read()
    print(f"{name:14s} {ms / n:.2f} ms per coarse launch (S=64)", flush=True)
    r.close()
run("real weights", wc)
run("zero weights", {k: np.zeros_like(v) for k, v in wc.items()})
run("real weights", wc)
