"""Child process of test_persistent_layer_kernel_is_bitwise_the_tile_kernel: the gradients of an odd batch (25 rays x 33 + 7 samples) in
the 16-bit training mode, with the GEMM kernels POSEGEN_LGEMM selects (read once per process); saves them to argv[1]."""
import sys, torch, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from posegen_amd import surreal_config
from posegen_amd.raycaster import HipRayCaster, make_training_draws
from posegen_amd.train import TrainableRayCaster
from tests.helpers import load_golden, model_for
g = load_golden("train_grads")
cfg = surreal_config(n_samples=33, n_importance=7)
wc, wf, tv, td = model_for(cfg, 4)
n = 25
rb, sk, cy = torch.tensor(g["ray_batch"][:n]), torch.tensor(g["skts"]), torch.tensor(g["cyl"])
draws = make_training_draws(n, 33, 7, perturb=1., raw_noise_std=1., pytest=True)
c = HipRayCaster.from_weights(cfg, wc, wf, float(tv), float(td), device="cuda:0", precision="fp32")
m = TrainableRayCaster(c, train_precision="bf16"); m.train()
out = m(rb, N_samples=33, skts=sk, cyls=cy, N_importance=7, draws={k: v.to("cuda:0") for k, v in draws.items()})
(out["rgb_map"].sum() + out["acc_map"].sum()).backward()
torch.save({f"{tag}.{k}": p.grad.cpu() for tag, net in (("coarse", m.network), ("fine", m.network_fine)) for k, p in net.named_parameters()}, sys.argv[1])
