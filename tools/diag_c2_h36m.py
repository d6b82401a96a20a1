"""BASELINE config 4 (h36m: 128 + 16 samples, frame codes) and per-ray poses in the compensated mode: whole-frame render
time with pg_evalc2.hip (POSEGEN_EVALC2=1, default) against pg_evalc.hip's record variant (POSEGEN_EVALC2=0)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one():
    import torch
    from bench import full_frame_rays, timed_rays
    from posegen_amd import h36m_config, surreal_config, synthetic as syn
    from posegen_amd.raycaster import HipRayCaster
    dev = torch.device("cuda:0")
    rb, skts, cyl, *_ = full_frame_rays(512, 512, dev)
    n = rb.shape[0]
    tag = f"EVALC2={os.environ.get('POSEGEN_EVALC2', '1')}"
    c4 = h36m_config()
    cast4 = HipRayCaster.from_weights(c4, *syn.make_model(c4, 0), device=dev, precision="fp16c")
    cams = (torch.arange(n, device=dev) % c4.n_framecodes).float()
    rs, msf, tf, kms = timed_rays(cast4.renderer, dev, rb, skts, cyl, c4, 2, cams=cams)
    an, ams = cast4.renderer.profile_read_aux()
    print(f"  {tag} h36m 512x512 fp16c: {msf:.2f} ms per frame, {rs / 1e6:.3f} M rays/s, fused kernel {tf / 2500:.3f} of peak, avg launch {kms:.2f} ms", flush=True)
    cast4.renderer.close()
    cfg = surreal_config()
    c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision="fp16c")
    m = 65536
    sk_pp = skts.expand(m, -1, -1, -1).contiguous()
    for name, sk in (("shared pose", skts), ("per-ray poses (materialised)", sk_pp)):
        rs, msf, tf, kms = timed_rays(c.renderer, dev, rb[:m], sk, cyl, cfg, 3)
        print(f"  {tag} surreal {m} rays, {name}: {msf:.2f} ms, {rs / 1e6:.3f} M rays/s", flush=True)
    c.renderer.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one()
    else:
        for v in ("1", "0"):
            subprocess.run([sys.executable, __file__, "one"], env=dict(os.environ, POSEGEN_EVALC2=v))
