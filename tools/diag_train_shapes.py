"""Training step on odd shapes (ray counts and sample counts that are no multiples of the tile sizes, with and without
importance samples / frame codes) in both training precisions: finite gradients, and the 16-bit gradient norm against fp32."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posegen_amd import surreal_config, h36m_config, synthetic as syn
from posegen_amd.train import TrainableRayCaster
from bench import full_frame_rays
dev = torch.device("cuda:0")
for cfgf, n, S, N in ((surreal_config, 1001, 64, 0), (surreal_config, 777, 65, 7), (h36m_config, 333, 80, 16), (surreal_config, 5, 64, 16), (surreal_config, 4097, 64, 16)):
    cfg = cfgf(n_samples=S, n_importance=N)
    outs = {}
    for prec in ("fp32", "bf16"):
        wc, wf, tv, td = syn.make_model(cfg, 0)
        from posegen_amd.raycaster import HipRayCaster
        m = TrainableRayCaster(HipRayCaster.from_weights(cfg, wc, wf, tv, td, device=dev, precision=prec))
        m.train()
        rb, skts, cyl, *_ = full_frame_rays(128, 128, dev)
        x = rb[4000:4000 + n].contiguous()
        cams = (torch.arange(n, device=dev) % max(cfg.n_framecodes, 1)).float() if cfg.framecode_ch else None
        out = m(x, N_samples=S, skts=skts, cyls=cyl, cams=cams, N_importance=N)
        loss = (out["rgb_map"] ** 2).mean() + (out["acc_map"] ** 2).mean()
        loss.backward()
        gn = sum(float(p.grad.double().norm() ** 2) for p in m.parameters() if p.grad is not None) ** 0.5
        ok = all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
        outs[prec] = (float(loss), gn, ok)
        m.renderer.close()
    print(cfgf.__name__, n, S, N, outs, "rel grad-norm diff", abs(outs["bf16"][1] - outs["fp32"][1]) / max(outs["fp32"][1], 1e-30))
