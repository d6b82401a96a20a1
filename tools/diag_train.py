import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.test_gpu_train import _trainable, _loss_of
from tests.helpers import load_golden, golden_draws
from tools.gen_golden import grad_sample_index
for name in ("train_grads", "train_grads_h36m"):
    g = load_golden(name)
    cfg, m = _trainable(g); m.train()
    cams = torch.tensor(g["cams"]) if "cams" in g else None
    out = m(torch.tensor(g["ray_batch"]), N_samples=cfg.n_samples, skts=torch.tensor(g["skts"]), cyls=torch.tensor(g["cyl"]), cams=cams, N_importance=cfg.n_importance, draws=golden_draws(g))
    loss = _loss_of(out, torch.tensor(g["target"], device="cuda:0")); loss.backward()
    print(name, "loss", float(loss.detach()), float(g["loss"]), {k: float(np.abs(out[k].detach().cpu().numpy()-g[k]).max()) for k in ("rgb_map","acc_map","rgb0","acc0")})
    for tag, net in (("coarse", m.network), ("fine", m.network_fine)):
        for key, p in net.items():
            k = key.replace("__", ".")
            ref_vals, ref_norm = g[f"gval_{tag}_{k}"], float(g[f"gnorm_{tag}_{k}"])
            got = p.grad.detach().cpu().numpy().reshape(-1)
            scale = max(float(np.abs(ref_vals).max()), ref_norm/np.sqrt(got.size), 1e-12)
            err = float(np.abs(got[grad_sample_index(got.size)] - ref_vals).max())
            nerr = abs(float(np.linalg.norm(got.astype(np.float64))) - ref_norm)/max(ref_norm,1e-30)
            print(f"  {tag:6s} {k:28s} rel err {err/scale:.2e} norm rel {nerr:.2e} scale {scale:.2e}")
