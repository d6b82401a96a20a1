"""Where the on-chip variant of pg_eval16r.hip stops paying: whole-frame bf16 render time by samples per ray, with and without
frame codes, POSEGEN_ONCHIP=2 (on-chip whatever the sample count) against POSEGEN_ONCHIP=0 (per-ray records).  The default rule
(pg_api.hip use_onchip: on-chip up to 112 samples per ray) comes from this table."""
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
    tag = f"ONCHIP={os.environ.get('POSEGEN_ONCHIP', 'rule')}"
    for fc in (False, True):
        for ns in (64, 96, 128, 192):
            cfg = h36m_config(n_samples=ns) if fc else surreal_config(n_samples=ns)
            cast = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision="bf16")
            cams = (torch.arange(n, device=dev) % cfg.n_framecodes).float() if fc else None
            rs, msf, tf, kms = timed_rays(cast.renderer, dev, rb, skts, cyl, cfg, 3, cams=cams)
            print(f"  {tag} fc={int(fc)} N_samples={ns}+16: {msf:.2f} ms per frame, fused kernel {tf / 2500:.3f} of peak", flush=True)
            cast.renderer.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one()
    else:
        for v in ("2", "0"):
            subprocess.run([sys.executable, __file__, "one"], env=dict(os.environ, POSEGEN_ONCHIP=v))
