#!/usr/bin/env python3
"""Headline benchmark: rendered rays/s of the A-NeRF hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: under torch.distributed.run)

One step = one 512x512 frame per GPU (262 144 rays, 64 coarse + 16 importance samples per
ray -> 144 MLP point evaluations per ray, both nets, synthetic surreal-config weights),
rendered by `pg_render_rays` with the ray batch already resident in HBM.  With N > 1 every
rank renders its own frame (image-parallel, weak scaling) and the frames are reassembled on
every rank by one RCCL all-gather of the packed (rgb, disp, acc) maps inside the timed region.
Rank 0 prints ONE JSON line (see README / DESIGN.md for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from posegen_amd import PREC_BY_NAME, PREC_NAMES, surreal_config, synthetic as syn  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3, "bf16x3": 2500.0, "fp16x3": 2500.0}


def full_frame_rays(H, W, device):
    """All H*W rays of the synthetic camera, cylinder radius 2.5 so that every ray hits
    (SURVEY.md 8(d) 'full' variant).  Ray formula of the reference's get_rays."""
    _, kps, skts = syn.make_pose(1, 1)
    c2ws, focals = syn.make_camera(1, H, W)
    c2w = torch.tensor(c2ws[0])
    f = float(focals[0])
    col = torch.arange(W, dtype=torch.float32)[None, :].expand(H, W)
    row = torch.arange(H, dtype=torch.float32)[:, None].expand(H, W)
    dirs = torch.stack([(col - W * 0.5) / f, -(row - H * 0.5) / f, -torch.ones(H, W)], -1)
    rd = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1).reshape(-1, 3)
    ro = c2w[:3, -1].expand(rd.shape)
    vd = rd / torch.norm(rd, dim=-1, keepdim=True)
    n = rd.shape[0]
    rb = torch.cat([ro, rd, torch.zeros(n, 1), torch.ones(n, 1), vd], -1).contiguous()
    from posegen_amd.skeleton import get_kp_bounding_cylinder
    cyl = torch.tensor(get_kp_bounding_cylinder(kps, ext_scale=0.001), dtype=torch.float32)
    cyl[:, 2] = 2.5
    return rb.to(device), torch.tensor(skts).to(device), cyl.to(device), rb, torch.tensor(skts), cyl


def cpu_baseline(rb_cpu, skts_cpu, cyl_cpu, cfg, model, n_rays):
    """The oracle (CPU port of the reference path) timed on the host cores, on a bounded
    sample of the same workload."""
    from oracle import anerf_oracle as orc
    wc, wf, tv, td = model
    ocfg = orc.OracleConfig(tau_v=tv, tau_d=td)
    tw = lambda w: {k: torch.tensor(v) for k, v in w.items()}
    # a 1-GPU box owns a 16-core share of its host; more threads than that only oversubscribe
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    n = rb_cpu.shape[0]
    start = (n // 2 // 512) * 512 + 128            # rows through the body
    sel = torch.arange(start, start + n_rays)
    sample = rb_cpu[sel]
    with torch.no_grad():
        orc.render_rays(sample[:64], skts_cpu, cyl_cpu, ocfg, tw(wc), tw(wf), cfg.n_samples, cfg.n_importance)
        t0 = time.time()
        ref = orc.render_rays(sample, skts_cpu, cyl_cpu, ocfg, tw(wc), tw(wf), cfg.n_samples, cfg.n_importance)
        dt = time.time() - t0
    return {"value": n_rays / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{n_rays} consecutive rays of the same 512x512 frame, one oracle call, "
                      f"torch {torch.__version__} CPU fp32, {cores} threads, {dt:.1f} s"}, sel, ref


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--prec", default="bf16", choices=list(PREC_BY_NAME))
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--cpu-rays", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-modes", action="store_true", help="skip the per-precision side measurements")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world == 1:
        sys.exit("bench.py --gpus N>1 must run under torch.distributed.run (one process per GPU)")
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from posegen_amd.raycaster import HipRayCaster
    cfg = surreal_config()
    model = syn.make_model(cfg, 0)
    caster = HipRayCaster.from_weights(cfg, *model, device=dev, precision=a.prec)
    H = W = a.res
    rb, skts, cyl, rb_cpu, skts_cpu, cyl_cpu = full_frame_rays(H, W, dev)
    n = rb.shape[0]
    r = caster.renderer
    packed = torch.empty(n, 5, device=dev)
    gathered = torch.empty(world, n, 5, device=dev) if world > 1 else None

    def step():
        out = r.render_rays(rb, skts, cyl, n_samples=cfg.n_samples, n_importance=cfg.n_importance,
                            want_alpha=False)
        if world > 1:           # reassemble the frames of all ranks: one RCCL all-gather
            packed[:, 0:3] = out["rgb_map"]
            packed[:, 3] = out["disp_map"]
            packed[:, 4] = out["acc_map"]
            dist.all_gather_into_tensor(gathered, packed)
        return out

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    sync()
    r.profile_enable(True)
    r.profile_read()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    sync()
    dt = time.perf_counter() - t0
    launches, k_ms, k_pts = r.profile_read()
    r.profile_enable(False)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    flops_pt = cfg.flops_per_point()
    rays_s = world * n * a.steps / dt
    peak = PEAK_TFLOPS[a.prec]
    k_tflops = k_pts * flops_pt / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    result = {
        "metric": "rendered rays/sec (512x512, 64 samples/ray)",
        "value": rays_s, "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.prec, "data": "synthetic",
        "config": {"workload": f"surreal {H}x{W} full frame per GPU ({n} rays, all rays hit: cylinder radius 2.5), "
                               f"{cfg.n_samples} coarse + {cfg.n_importance} importance samples/ray = "
                               f"{cfg.evals_per_ray()} MLP evals/ray (coarse + fine net), seeded synthetic weights/pose",
                   "frames_per_step_per_gpu": 1, "parallelism": f"image-parallel x{world}" if world > 1 else "single GPU",
                   "flop_per_ray": flops_pt * cfg.evals_per_ray()},
        "roofline": {"bound": "mfma", "kernel": "eval16_kernel (fused embed+MLP)" if a.prec in ("bf16", "fp16") else "eval32_kernel",
                     "achieved": k_tflops, "peak": peak, "unit": "TFLOP/s", "frac": k_tflops / peak,
                     "traffic": None, "launches": launches, "avg_launch_ms": k_ms / max(launches, 1),
                     "points_per_launch": k_pts / max(launches, 1), "flop_per_point": flops_pt,
                     "end_to_end_frac": rays_s / world * flops_pt * cfg.evals_per_ray() / 1e12 / peak},
    }
    # the peak re-derived on this box: CUs x 4 SIMDs x 1024 bf16 MFMA FLOP/clk x max clock
    di = r.device_info()
    if di["clock_khz"] > 0 and a.prec in ("bf16", "fp16", "bf16x3"):
        result["roofline"]["peak_derived"] = di["n_cu"] * 4 * 1024 * di["clock_khz"] * 1e3 / 1e12
        result["roofline"]["peak_derived_from"] = f"{di['n_cu']} CUs x 4096 FLOP/clk x {di['clock_khz'] / 1e6:.2f} GHz (hipDeviceProp)"
    # `achieved` counts the ALGORITHMIC flops of the reference network.  The 16-bit
    # kernels execute fewer: feature_linear is folded into the view layer and the view-direction
    # input is factorised over rays, both exact in real arithmetic.  The MFMA
    # flops actually issued are reported beside it.
    q = r.query()
    mfma_flop = {"bf16": 32768, "fp16": 32768, "bf16x3": 32768, "fp32": 4096}.get(a.prec)
    if mfma_flop:
        ex = q["mfma_per_group"] * mfma_flop / 32.0
        result["roofline"]["executed_flop_per_point"] = ex
        result["roofline"]["executed_tflops"] = k_tflops * ex / flops_pt
        result["roofline"]["executed_frac"] = k_tflops * ex / flops_pt / peak

    # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes of this
    # same command (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, profiles/r1_traffic.json);
    # bench.py cannot run the profiler on itself, so the committed measurement is attached.
    tpath = os.path.join(REPO, "profiles", "r1_traffic.json")
    if a.prec == "bf16" and H == 512 and os.path.exists(tpath):
        tj = json.load(open(tpath))
        result["roofline"]["traffic"] = tj["hbm_bytes"]
        result["roofline"]["traffic_unit"] = "bytes per launch (PMC, profiles/r1_traffic.json)"
        result["roofline"]["algorithmic_bytes_per_launch"] = tj["algorithmic_bytes"]

    sel = ref = None
    if not a.no_cpu_baseline:
        base, sel, ref = cpu_baseline(rb_cpu, skts_cpu, cyl_cpu, cfg, model, a.cpu_rays)
        result["cpu_baseline"] = base
        # parity of the measured configuration on the CPU-baseline sample
        got = r.render_rays(rb[sel.to(dev)], skts, cyl, n_samples=cfg.n_samples, n_importance=cfg.n_importance)
        err = {k: float((got[k].cpu() - ref[k]).abs().max()) for k in ("rgb_map", "acc_map")}
        solid = ref["acc_map"] > 1e-3        # disparity of an empty ray is 1/(0/0): noise in the reference too
        err["disp_map(acc>1e-3)"] = float((got["disp_map"].cpu() - ref["disp_map"])[solid].abs().max()) if solid.any() else 0.0
        mse = float(((got["rgb_map"].cpu() - ref["rgb_map"]) ** 2).mean())
        result["parity"] = {"vs": "oracle (fp32 CPU port pinned to the reference's golden vectors)", "rays": int(len(sel)),
                            "max_abs": err, "rgb_rmse": mse ** 0.5,
                            "rgb_psnr_db": -10 * np.log10(max(mse, 1e-30))}

    if not a.no_modes and world == 1:
        modes = {}
        for name in ("fp16", "bf16x3", "fp32"):
            if name == a.prec:
                continue
            r.set_precision(name)
            steps = 3 if name == "fp16" else 1
            r.render_rays(rb[: n // 8], skts, cyl, want_alpha=False)
            torch.cuda.synchronize(dev)
            r.profile_enable(True); r.profile_read()
            t0 = time.perf_counter()
            for _ in range(steps):
                r.render_rays(rb, skts, cyl, want_alpha=False)
            torch.cuda.synchronize(dev)
            mdt = time.perf_counter() - t0
            ml, mms, mpts = r.profile_read()
            r.profile_enable(False)
            m = {"rays_per_s": n * steps / mdt, "ms_per_frame": mdt / steps * 1e3,
                 "kernel_tflops": mpts * flops_pt / (mms * 1e-3) / 1e12, "peak_tflops": PEAK_TFLOPS[name]}
            m["frac"] = m["kernel_tflops"] / m["peak_tflops"]
            if sel is not None:
                got = r.render_rays(rb[sel.to(dev)], skts, cyl)
                m["max_abs_rgb_vs_oracle"] = float((got["rgb_map"].cpu() - ref["rgb_map"]).abs().max())
            modes[name] = m
        r.set_precision(a.prec)
        result["modes"] = modes

    print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
