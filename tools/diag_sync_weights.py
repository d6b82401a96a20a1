"""Cost of TrainableRayCaster.sync_inference_weights (parameters -> host -> packed weight streams) and of the first render
behind it (the streams are packed lazily per precision and kernel form)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posegen_amd import surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
from posegen_amd.train import TrainableRayCaster
from bench import full_frame_rays
dev = torch.device("cuda:0")
cfg = surreal_config()
m = TrainableRayCaster(HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision="bf16"))
rb, skts, cyl, *_ = full_frame_rays(128, 128, dev)
m.eval()
with torch.no_grad():
    m(rb, N_samples=64, skts=skts, cyls=cyl, N_importance=16)
torch.cuda.synchronize()
for rep in range(6):
    on_dev = rep >= 3
    t0 = time.perf_counter(); m.sync_inference_weights(on_device=on_dev); th = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"[{'device' if on_dev else 'host'} route] host time of the call {1e3 * (th - t0):.2f} ms;", end=" ")
    with torch.no_grad():
        m(rb, N_samples=64, skts=skts, cyls=cyl, N_importance=16)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    with torch.no_grad():
        m(rb, N_samples=64, skts=skts, cyls=cyl, N_importance=16)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"sync_inference_weights {1e3 * (t1 - t0):.1f} ms, first render behind it {1e3 * (t2 - t1):.1f} ms, the next one {1e3 * (t3 - t2):.1f} ms")
