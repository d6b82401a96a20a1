"""Randomised parity sweep on the GPU: the HIP path against the CPU oracle (itself pinned to the reference's
golden vectors) over random poses, cameras, model seeds, sample counts, inverse-depth sampling and frame codes.

    python tools/parity_sweep.py [--cases 40] [--rays 768]          # prints one line per case + the maxima

`sweep()` is also what tests/test_gpu_configs.py::test_seeded_parity_sweep runs (fewer cases).
Test infrastructure: uses oracle/.
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch

MODES = ("fp32", "fp16c", "bf16x3", "fp16", "bf16")


def sweep(cases=40, rays=768, seed=2026, modes=MODES, dev="cuda:0", verbose=True):
    """One dict per case: the drawn configuration and, per mode, max |error| of rgb / acc / disp (where
    acc > 1e-3) and the rgb MSE against the oracle."""
    from oracle import anerf_oracle as orc
    from posegen_amd import h36m_config, surreal_config, synthetic as syn
    from posegen_amd.raycaster import HipRayCaster
    from tests.helpers import oracle_cfg, torch_weights
    rng = np.random.RandomState(seed)
    out = []
    for case in range(cases):
        fc = bool(rng.randint(0, 2))
        S = int(rng.choice([32, 48, 64, 80, 128]))
        N = int(rng.choice([0, 2, 16, 32]))
        lindisp = bool(rng.randint(0, 4) == 0)
        cfg = (h36m_config if fc else surreal_config)(n_samples=S, n_importance=N)
        seed_m, seed_p = int(rng.randint(0, 1000)), int(rng.randint(0, 1000))
        wc, wf, tv, td = syn.make_model(cfg, seed_m)
        _, kps, skts = syn.make_pose(1, seed_p)
        H = W = 96
        c2ws, focals = syn.make_camera(1, H, W)
        c2ws[0, :3, 3] += rng.uniform(-0.15, 0.15, size=3).astype(np.float32)          # jitter the camera position
        ray_l, vids, cyls, boxes = orc.valid_rays(torch.tensor(c2ws), H, W, focals, torch.tensor(kps), cfg.ext_scale)
        ro, rd = ray_l[0]
        sel = torch.tensor(rng.choice(ro.shape[0], size=min(rays, ro.shape[0]), replace=False))
        ro, rd = ro[sel].float(), rd[sel].float()
        n = ro.shape[0]
        ones = torch.ones(n, 1)
        rb = torch.cat([ro, rd, 0. * ones, 1. * ones, rd / rd.norm(dim=-1, keepdim=True)], -1)
        # per-ray frame-code indices, or all negative = the mean code (embedding.py:21-22; the reference cannot
        # mix the two in one call: a negative index next to valid ones raises in nn.Embedding)
        cams = torch.tensor(rng.randint(0, cfg.n_framecodes, size=n).astype(np.float32)) if fc else None
        if cams is not None and rng.randint(0, 3) == 0:
            cams[:] = -1.0
        ocfg = oracle_cfg(cfg, tv, td)
        ref = orc.render_rays(rb, torch.tensor(skts), cyls, ocfg, torch_weights(wc), torch_weights(wf), S, N, cams=cams,
                              lindisp=lindisp)
        solid = ref["acc_map"] > 1e-3
        c = HipRayCaster.from_weights(cfg, wc, wf, tv, td, device=dev, precision="fp32")
        rec = {"case": case, "fc": fc, "S": S, "N": N, "lindisp": lindisp, "rays": n, "hit": float(solid.float().mean()), "err": {}}
        line = []
        for m in modes:
            c.renderer.set_precision(m)
            got = c.renderer.render_rays(rb, torch.tensor(skts), cyls, cams=cams, n_samples=S, n_importance=N, lindisp=lindisp)
            e = {"rgb": float((got["rgb_map"].cpu() - ref["rgb_map"]).abs().max()),
                 "acc": float((got["acc_map"].cpu() - ref["acc_map"]).abs().max()),
                 "disp": float((got["disp_map"].cpu() - ref["disp_map"])[solid].abs().max()) if solid.any() else 0.0,
                 "mse": float(((got["rgb_map"].cpu() - ref["rgb_map"]) ** 2).mean())}
            rec["err"][m] = e
            line.append(f"{m} {e['rgb']:.1e}/{e['acc']:.1e}")
        c.renderer.close()
        out.append(rec)
        if verbose:
            print(f"case {case:2d} fc={int(fc)} S={S:3d} N={N:2d} lindisp={int(lindisp)} rays={n} hit={rec['hit']:.2f} | "
                  + "  ".join(line), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--rays", type=int, default=768)
    a = ap.parse_args()
    t0 = time.time()
    recs = sweep(a.cases, a.rays)
    print(f"# {a.cases} cases in {time.time() - t0:.0f} s; max |error| vs the oracle (rgb / acc / disp where acc > 1e-3), worst PSNR:")
    worst32 = 0.0
    for m in MODES:
        per = np.array([max(r["err"][m]["rgb"], r["err"][m]["acc"]) for r in recs])
        w = {k: max(r["err"][m][k] for r in recs) for k in ("rgb", "acc", "disp", "mse")}
        q = np.quantile(per, [0.5, 0.9])
        print(f"# {m:7s} {w['rgb']:.2e} / {w['acc']:.2e} / {w['disp']:.2e}  PSNR >= {-10 * np.log10(max(w['mse'], 1e-30)):.1f} dB"
              f"    rgb|acc per case: median {q[0]:.1e}, 90 % {q[1]:.1e}")
        if m == "fp32":
            worst32 = max(w["rgb"], w["acc"])
    # The inverse-cdf importance sampling of the reference is ill-conditioned where the coarse weights are spread thin
    # (random-weight nets at 32-48 coarse samples): a 1e-6 difference in a weight can move a sample across a bin, so
    # even the fp32 kernel and the fp32 oracle (different summation orders) part by up to ~1e-4 there; cases without
    # importance samples stay at 1e-5.  The sweep fails only if the fp32 kernel leaves that envelope.
    sys.exit(1 if worst32 > 3e-4 else 0)


if __name__ == "__main__":
    main()
