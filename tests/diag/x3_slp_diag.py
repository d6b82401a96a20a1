"""One diagnostic run for the intermittent split-operand fault of round 1.

Run it with POSEGEN_HIP_LIB pointing at a library whose pg_eval32.hip was compiled WITH the SLP
vectoriser (tools/build_variant.sh slp32 -fslp-vectorize) -- the build in which wrong view-layer
outputs were seen -- and once with the regular library.  For bf16x3 against the fp32 kernel on the
same inputs it reports, per repetition: max |d raw|, the bad points' (pass, wave, lane, ray, sample)
distribution, run-to-run equality, and which of the debug taps (wd = stage 10, view input = stage 11,
view output = stage 9, trunk output = stage 7) already differs at the bad points.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from posegen_amd import surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
from bench import full_frame_rays

dev = torch.device("cuda:0")
cfg = surreal_config()
c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision="fp32")
r = c.renderer
rb, skts, cyl, *_ = full_frame_rays(512, 512, dev)
print("library:", os.environ.get("POSEGEN_HIP_LIB", "(default)"))
for S, nr in ((64, 32768), (80, 32768)):
    rbs = rb[100000:100000 + nr]
    nf, z = r.stage_sample_coarse(rbs, cyl, S)
    r.set_precision("fp32")
    ref = r.stage_eval(0, rbs, z, skts).clone()
    r.set_precision("bf16x3")
    first = None
    for rep in range(6):
        raw = r.stage_eval(0, rbs, z, skts).clone()
        d = (raw - ref).abs().amax(-1).reshape(-1)
        bad = (d > 1e-3).nonzero().reshape(-1).cpu().numpy()
        same = True if first is None else bool(torch.equal(raw, first))
        first = raw if first is None else first
        msg = f"S={S} rep {rep}: max|d raw| {float(d.max()):.3e} bad {len(bad)}/{d.numel()} bitwise==rep0 {same}"
        if len(bad):
            msg += (f"\n   lane(pt%32) hist {np.bincount(bad % 32, minlength=32).tolist()}"
                    f"\n   wave((pt//32)%4) hist {np.bincount((bad // 32) % 4, minlength=4).tolist()}"
                    f"\n   first bad points (pass, wave, lane, ray, sample): "
                    f"{[(int(b // 128), int((b // 32) % 4), int(b % 32), int(b // S), int(b % S)) for b in bad[:12]]}"
                    f"\n   channel of max diff per bad point (0..2 rgb, 3 sigma): {(raw - ref).abs().reshape(-1, 4)[bad[:12]].argmax(-1).tolist()}")
        print(msg, flush=True)
    # debug taps, both kernels, same points (a separate launch each)
    sub = slice(0, 4096)
    for stage, what in ((7, "trunk output h7"), (10, "view cutoff weights"), (11, "view input values"), (9, "view layer output")):
        r.set_precision("fp32")
        _, a = r.stage_eval(0, rbs[sub], z[sub], skts, want_dbg=True, dbg_stage=stage)
        r.set_precision("bf16x3")
        outs = [r.stage_eval(0, rbs[sub], z[sub], skts, want_dbg=True, dbg_stage=stage)[1] for _ in range(3)]
        dd = [(o - a).abs().amax(-1) for o in outs]
        rep = all(torch.equal(outs[0], o) for o in outs[1:])
        worst = int(torch.stack(dd).amax(0).argmax())
        print(f"S={S} tap {stage} ({what}): max|x3 - fp32| {[f'{float(x.max()):.2e}' for x in dd]} repeatable {rep}; "
              f"worst point {worst} (wave {(worst // 32) % 4}, lane {worst % 32})", flush=True)
c.renderer.close()
