import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from posegen_amd import h36m_config, surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
from bench import full_frame_rays
dev = torch.device("cuda:0")
for name, cfgf in (("h36m", h36m_config), ("surreal", surreal_config)):
    cfg = cfgf()
    c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision="bf16")
    rb, skts, cyl, *_ = full_frame_rays(512, 512, dev)
    n = rb.shape[0]
    cams = (torch.arange(n, device=dev) % max(cfg.n_framecodes, 1)).float() if cfg.framecode_ch else None
    r = c.renderer
    r.set_chunk(cfg.chunk)
    for rows in (32768, n, n, n, 200000, n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.render_rays(rb[:rows], skts, cyl, cams=None if cams is None else cams[:rows], want_alpha=False)
        torch.cuda.synchronize(); print(name, rows, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    r.close()
