"""BASELINE config 4 (h36m: 128 + 16 samples, frame codes with a per-ray index) in the 16-bit modes: whole-frame render time
with the on-chip variant of pg_eval16r.hip (frame-code rows from the host-made table; POSEGEN_ONCHIP=2: forced, the default takes records from 113 samples per ray on) against
the per-ray-record variant (POSEGEN_ONCHIP=0), and the difference of their maps."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(out):
    import torch
    from bench import full_frame_rays, timed_rays
    from posegen_amd import h36m_config, synthetic as syn
    from posegen_amd.raycaster import HipRayCaster
    dev = torch.device("cuda:0")
    rb, skts, cyl, *_ = full_frame_rays(512, 512, dev)
    n = rb.shape[0]
    tag = f"ONCHIP={os.environ.get('POSEGEN_ONCHIP', 'rule')}"
    c4 = h36m_config()
    maps = {}
    for prec in ("bf16", "fp16"):
        cast4 = HipRayCaster.from_weights(c4, *syn.make_model(c4, 0), device=dev, precision=prec)
        cams = (torch.arange(n, device=dev) % c4.n_framecodes).float()
        cams[5::7] = -1.0           # (the mean code)
        rs, msf, tf, kms = timed_rays(cast4.renderer, dev, rb, skts, cyl, c4, 2, cams=cams)
        print(f"  {tag} h36m 512x512 {prec}: {msf:.2f} ms per frame, {rs / 1e6:.3f} M rays/s, fused kernel {tf / 2500:.3f} of peak, avg launch {kms:.2f} ms", flush=True)
        r = cast4.renderer.render_rays(rb, skts, cyl, cams=cams, want_alpha=False)
        maps[prec] = {k: r[k].cpu() for k in ("rgb_map", "acc_map", "disp_map")}
        cast4.renderer.close()
    torch.save(maps, out)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(sys.argv[2])
    else:
        import torch
        os.makedirs("gpurun_out", exist_ok=True)
        for v in ("2", "0"):
            subprocess.run([sys.executable, __file__, "one", f"gpurun_out/h36m_onchip_{v}.pt"], env=dict(os.environ, POSEGEN_ONCHIP=v))
        a, b = torch.load("gpurun_out/h36m_onchip_2.pt"), torch.load("gpurun_out/h36m_onchip_0.pt")
        for prec in a:
            print(prec, "on-chip vs records:", {k: float((a[prec][k] - b[prec][k]).abs().max()) for k in a[prec]}, flush=True)
        os.remove("gpurun_out/h36m_onchip_2.pt"); os.remove("gpurun_out/h36m_onchip_0.pt")
