"""Raw outputs of the compensated kernel's on-chip form against its record form (POSEGEN_ONCHIP=0 in a child process):
where do they differ -- which rays of a pass, which channels."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if os.environ.get("DIAG_CHILD"):
    import torch
    from posegen_amd import surreal_config, synthetic as syn
    from posegen_amd.raycaster import HipRayCaster
    from bench import full_frame_rays
    dev = torch.device("cuda:0")
    cfg = surreal_config()
    c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision=os.environ.get("PREC", "fp16c"))
    rb, skts, cyl, *_ = full_frame_rays(128, 128, dev)
    r = c.renderer
    if os.environ.get("NOSKIP"): r.set_far_skip(False)
    rows = int(os.environ.get("ROWS", "4096"))
    x = rb[6000:6000 + rows].contiguous()
    nf, z = r.stage_sample_coarse(x, cyl, 64)
    raw = r.stage_eval(0, x, z, skts)
    np.save(os.environ["DIAG_CHILD"], raw.cpu().numpy())
    sys.exit(0)

out = {}
for oc in ("1", "0"):
    f = f"/tmp/diag_oc_{oc}.npy"
    subprocess.run([sys.executable, __file__], env=dict(os.environ, DIAG_CHILD=f, POSEGEN_ONCHIP=oc), check=True)
    out[oc] = np.load(f)
a, b = out["1"], out["0"]
d = np.abs(a - b)
print("shape", a.shape, "max |d| per channel", d.reshape(-1, 4).max(0), "mean", d.reshape(-1, 4).mean(0))
per_ray = d[..., :3].max(-1).max(-1)
print("rays differing > 1e-4:", int((per_ray > 1e-4).sum()), "of", per_ray.size)
bad = np.nonzero(per_ray > 1e-4)[0]
print("first bad rays", bad[:32])
if bad.size:
    r0 = bad[0]
    print("ray", r0, "per-sample max |d rgb|", np.round(d[r0, :, :3].max(-1), 5))
    print("values OC ", a[r0, :4], "\nvalues REC", b[r0, :4])
