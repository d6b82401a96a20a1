"""Device time of the fused embed+MLP launch and of the per-ray record kernel in front of it (coarse pass, 512x512 x S),
for the whole frame and for slices of it (ROWS env: rays per launch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posegen_amd import surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
from bench import full_frame_rays
dev = torch.device("cuda:0")
cfg = surreal_config()
c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision=os.environ.get("PREC", "bf16"))
rb, skts, cyl, *_ = full_frame_rays(512, 512, dev)
r = c.renderer
for rows in [int(x) for x in os.environ.get("ROWS", "262144,65536,16384").split(",")]:
    for S in (64, 80):
        nf, z = r.stage_sample_coarse(rb[:rows], cyl, S)
        r.stage_eval(0, rb[:rows], z, skts)
        torch.cuda.synchronize()
        r.profile_enable(True); r.profile_read(); r.profile_read_aux()
        for _ in range(max(5, 5 * 262144 // rows // 4)):
            r.stage_eval(0, rb[:rows], z, skts)
        n, ms, pts = r.profile_read()
        na, msa = r.profile_read_aux()
        print(f"rays={rows} S={S}: eval {ms / n:.3f} ms per launch ({pts * cfg.flops_per_point() / (ms * 1e-3) / 1e12:.0f} TFLOP/s algorithmic); "
              f"records {msa / max(na, 1):.3f} ms per launch ({na} launches)", flush=True)
