"""pg_evalc2.hip (compensated fp16, out tiles over the waves) against the oracle, the fp32 kernel and pg_evalc.hip:
raw values of the coarse net on the golden ray sets (shared / per-ray poses, frame codes, odd sample counts), then
the launch time on the 512 x 512 benchmark frame.   usage: diag_evalc2.py [check|time]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import anerf_oracle as orc
from posegen_amd import PREC_FP16C, PREC_FP32
from posegen_amd.raycaster import HipRayCaster
from tests.helpers import cfg_from_golden, load_golden, model_for, oracle_cfg, torch_weights

DEV = "cuda:0"


def md(a, b):
    return float(np.nanmax(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def check():
    bad = 0
    for name in ("rays_surreal", "rays_allhit", "rays_h36m"):
        g = load_golden(name)
        cfg = cfg_from_golden(g)
        wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
        c = HipRayCaster.from_weights(cfg, wc, wf, tv, td, device=DEV, precision=PREC_FP16C)
        rb, skts = torch.tensor(g["ray_batch"]), torch.tensor(g["skts"])
        cams = torch.tensor(g["cams"]) if "cams" in g else None
        z = torch.tensor(g["z_coarse"])
        n, S = z.shape
        raw = c.renderer.stage_eval(0, rb, z, skts, cams=cams).cpu()
        ocfg = oracle_cfg(cfg, g["tau_v"], g["tau_d"])
        pts = rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None]
        x = orc.embed_points(pts, rb[:, 3:6], skts, ocfg, cams=cams) if cams is not None else orc.embed_points(pts, rb[:, 3:6], skts, ocfg)
        ref = orc.mlp_forward(x.reshape(n * S, -1), torch_weights(wc), ocfg).reshape(n, S, 4)
        scale = float(ref.abs().max())
        d = md(raw.numpy(), ref.numpy())
        ok = bool(torch.isfinite(raw).all()) and d <= 9e-4 * max(1.0, scale / 10)
        bad += not ok
        print(f"[{name}] n={n} S={S} fc={cfg.framecode_ch}: raw vs oracle {d:.3e} (|raw| max {scale:.1f}) {'ok' if ok else 'FAIL'}", flush=True)
        if not ok:
            e = (raw - ref).abs().reshape(n * S, 4)
            worst = torch.argsort(e.max(1).values, descending=True)[:8]
            for i in worst.tolist():
                print("   point", i, "ray", i // S, "sample", i % S, "err", e[i].tolist(), "ref", ref.reshape(-1, 4)[i].tolist(), "got", raw.reshape(-1, 4)[i].tolist())
            print("   fraction of points off by > 1e-2:", float((e.max(1).values > 1e-2).float().mean()))
        # per-ray poses: the same kernel with the bone rows read per ray
        raw_pp = c.renderer.stage_eval(0, rb, z, skts.expand(n, -1, -1, -1).contiguous(), cams=cams).cpu()
        dpp = md(raw_pp.numpy(), raw.numpy())
        print(f"    per-ray poses vs shared: {dpp:.3e} {'ok' if dpp == 0.0 else 'DIFFERENT'}", flush=True)
        bad += dpp != 0.0
        # odd sample counts against the fp32 kernel
        for S2 in (33, 48, 63, 65, 80, 97, 144, 200):
            m = 37
            rng = np.random.RandomState(S2)
            lo, hi = z[:m, :1], z[:m, -1:]
            z2 = lo + (hi - lo) * torch.tensor(np.sort(rng.uniform(0, 1, size=(m, S2)), axis=1), dtype=torch.float32)
            cm = None if cams is None else cams[:m]
            c.renderer.set_precision(PREC_FP32)
            r32 = c.renderer.stage_eval(0, rb[:m], z2, skts, cams=cm).cpu()
            c.renderer.set_precision(PREC_FP16C)
            r16 = c.renderer.stage_eval(0, rb[:m], z2, skts, cams=cm).cpu()
            d2 = md(r16.numpy(), r32.numpy())
            sc = float(r32.abs().max())
            ok2 = bool(torch.isfinite(r16).all()) and d2 <= 9e-4 * max(1.0, sc / 10)
            bad += not ok2
            print(f"    S={S2}: vs fp32 kernel {d2:.3e} (|raw| max {sc:.1f}) {'ok' if ok2 else 'FAIL'}", flush=True)
        # limb masks off
        c.renderer.set_far_skip(False)
        raw_ns = c.renderer.stage_eval(0, rb, z, skts, cams=cams).cpu()
        c.renderer.set_far_skip(True)
        print(f"    masks off vs on: {md(raw_ns.numpy(), raw.numpy()):.3e}; masks off vs oracle {md(raw_ns.numpy(), ref.numpy()):.3e}", flush=True)
        c.renderer.close()
    print("CHECK", "FAILED" if bad else "OK", flush=True)
    return bad


def time_one():
    from bench import full_frame_rays
    from posegen_amd import surreal_config, synthetic as syn
    cfg = surreal_config()
    c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=DEV, precision=PREC_FP16C)
    rb, skts, cyl, *_ = full_frame_rays(512, 512, torch.device(DEV))
    r = c.renderer
    tot = 0.0
    for S in (64, 80):
        nf, z = r.stage_sample_coarse(rb, cyl, S)
        r.stage_eval(0, rb, z, skts)
        torch.cuda.synchronize()
        r.profile_enable(True); r.profile_read()
        for _ in range(4):
            r.stage_eval(0, rb, z, skts)
        n, ms, pts = r.profile_read()
        tot += ms / n
        print(f"  EVALC2={os.environ.get('POSEGEN_EVALC2', '1')} S={S}: eval {ms / n:.3f} ms, {pts * cfg.flops_per_point() / (ms * 1e-3) / 2.5e15:.3f} of peak", flush=True)
    print(f"  EVALC2={os.environ.get('POSEGEN_EVALC2', '1')} coarse+fine {tot:.3f} ms", flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "check":
        sys.exit(check())
    if mode == "time1":
        time_one()
    else:
        for rnd in range(2):
            for v in ("1", "0"):
                subprocess.run([sys.executable, __file__, "time1"], env=dict(os.environ, POSEGEN_EVALC2=v))
