"""Diagnosis build (PG_DEBUG_Y): the Y image of workgroup 0's first pass against Y computed on the host."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from posegen_amd import surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
from bench import full_frame_rays
PERMC = [1, 7, 2, 8, 16, 20, 17, 21, 0, 6, 12, 13, 4, 10, 5, 11, 18, 22, 19, 23, 3, 9, 15, 14]
dev = torch.device("cuda:0")
cfg = surreal_config()
wc, wf, tv, td = syn.make_model(cfg, 0)
c = HipRayCaster.from_weights(cfg, wc, wf, tv, td, device=dev, precision="fp16c")
rb, skts, cyl, *_ = full_frame_rays(128, 128, dev)
r = c.renderer
r.set_far_skip(False)
x = rb[6000 + 53:6000 + 53 + 512].contiguous()
nf, z = r.stage_sample_coarse(x, cyl, 64)
raw, dbg = r.stage_eval(0, x, z, skts, want_dbg=True, dbg_stage=98)
d = dbg.cpu().numpy().reshape(-1)
print("nrm1, gmask", d[13000], d[13001])
img = d[:3 * 4096].view(np.uint8).reshape(3, 4, 2, 2, 64, 8, 2).copy().view(np.float16)[..., 0].astype(np.float64)   # [ray][o][ku][plane][lane][e]
ab = d[3 * 4096:3 * 4096 + 576].reshape(3, 24, 8)
Wv = wc["views_linears.0.weight"] if "views_linears.0.weight" in wc else None
if Wv is None:
    print([k for k in wc.keys()])
    sys.exit(1)
for ray in range(2):
    worst = 0.0
    for s in range(24):
        h, pj = divmod(s, 12)
        j = PERMC[s]
        b = ab[ray, s, 4:7].astype(np.float64)
        e = b / max(np.linalg.norm(b), 1e-12)
        T = np.zeros(27)
        for k in range(27):
            cc, r9 = divmod(k, 9)
            T[k] = e[cc] if r9 == 0 else (np.sin if (r9 - 1) % 2 == 0 else np.cos)(e[cc] * 2.0 ** ((r9 - 1) // 2))
        cols = [256 + (k % 9) * 72 + 3 * j + k // 9 for k in range(27)]
        Y = Wv[:, cols].astype(np.float64) @ T          # [128]
        ku, ee = divmod(pj, 8)
        for o in range(4):
            p0 = img[ray, o, ku, 0, 32 * h:32 * h + 32, ee]
            p1 = img[ray, o, ku, 1, 32 * h:32 * h + 32, ee]
            y1 = p0 / 128.0
            got = 129.0 * (y1 + (p1 - y1) / 129.0)
            err = np.abs(got - Y[32 * o:32 * o + 32]).max()
            worst = max(worst, err)
            if ray == 0 and s in (0, 12) and o == 0:
                print("slot", s, "want", np.round(Y[:6], 4), "got", np.round(got[:6], 4))
    print("ray", ray, "worst |Y - host|", worst)
    errs = []
    for s in range(24):
        h, pj = divmod(s, 12); j = PERMC[s]
        b = ab[ray, s, 4:7].astype(np.float64); e = b / max(np.linalg.norm(b), 1e-12)
        T = np.array([e[k // 9] if k % 9 == 0 else (np.sin if (k % 9 - 1) % 2 == 0 else np.cos)(e[k // 9] * 2.0 ** ((k % 9 - 1) // 2)) for k in range(27)])
        Y = Wv[:, [256 + (k % 9) * 72 + 3 * j + k // 9 for k in range(27)]].astype(np.float64) @ T
        ku, ee = divmod(pj, 8)
        got = np.concatenate([129.0 * (img[ray, o, ku, 0, 32 * h:32 * h + 32, ee] / 128.0 * (1 - 1 / 129.0) + img[ray, o, ku, 1, 32 * h:32 * h + 32, ee] / 129.0) for o in range(4)])
        errs.append(float(np.abs(got - Y).max()))
    print("per slot:", " ".join("%.1e" % v for v in errs))
    print("e of slots 0..3:", [np.round(ab[ray, s, 4:7] / np.linalg.norm(ab[ray, s, 4:7]), 3) for s in range(4)])
