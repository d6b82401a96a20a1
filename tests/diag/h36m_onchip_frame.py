"""Child process of test_config4_on_chip_frame_codes_at_size: BASELINE config 4's 512 x 512 frame in bf16 and fp16 with
the form of the 16x16x32 kernel that POSEGEN_ONCHIP selects (read once per process); saves the maps to argv[1]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from bench import full_frame_rays
from posegen_amd import h36m_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster


def main():
    dev = torch.device("cuda:0")
    rb, skts, cyl, *_ = full_frame_rays(512, 512, dev)
    n = rb.shape[0]
    cfg = h36m_config()
    cams = (torch.arange(n, device=dev) % cfg.n_framecodes).float()
    cams[5::7] = -1.0           # (rays without a frame: the mean code, embedding.py:25-26)
    out = {}
    c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision="bf16")
    for prec in ("bf16", "fp16"):
        c.renderer.set_precision(prec)
        c.renderer.profile_enable(True)
        c.renderer.profile_read(); c.renderer.profile_read_aux()
        a = c.renderer.render_rays(rb, skts, cyl, cams=cams, want_alpha=False)
        torch.cuda.synchronize()
        launches, _, _ = c.renderer.profile_read()
        recs, _ = c.renderer.profile_read_aux()
        c.renderer.profile_enable(False)
        b = c.renderer.render_rays(rb, skts, cyl, cams=cams, want_alpha=False)
        out[prec] = {k: a[k].cpu() for k in ("rgb_map", "acc_map", "disp_map")}
        out[prec]["repeatable"] = all(torch.equal(a[k], b[k]) for k in ("rgb_map", "acc_map", "disp_map"))
        out[prec]["launches"] = (launches, recs)
    c.renderer.close()
    torch.save(out, sys.argv[1])


if __name__ == "__main__":
    main()
