"""BASELINE config 4's 512 x 512 frame (128 + 16 samples, a frame-code index per ray), FRAMES renders in PREC -- the
program tools/collect_profiles.sh profiles for profiles/r5_h36m_* (PROG=tools/run_h36m.py; POSEGEN_ONCHIP selects the form)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import full_frame_rays
from posegen_amd import h36m_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster

dev = torch.device("cuda:0")
cfg = h36m_config()
c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision=os.environ.get("PREC", "bf16"))
rb, skts, cyl, *_ = full_frame_rays(512, 512, dev)
cams = (torch.arange(rb.shape[0], device=dev) % cfg.n_framecodes).float()
r = c.renderer
r.render_rays(rb, skts, cyl, cams=cams, want_alpha=False)
torch.cuda.synchronize()
n = int(os.environ.get("FRAMES", "3"))
t0 = time.perf_counter()
for _ in range(n):
    r.render_rays(rb, skts, cyl, cams=cams, want_alpha=False)
torch.cuda.synchronize()
print(f"h36m 512x512 {os.environ.get('PREC', 'bf16')} ONCHIP={os.environ.get('POSEGEN_ONCHIP', 'rule')}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per frame", flush=True)
