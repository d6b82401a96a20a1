import sys, os, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import torch
import bench
from posegen_amd import surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
dev = "cuda:0"
cfg = surreal_config()
caster = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device=dev, precision="bf16")
run, _state = bench.strong_workload(caster, cfg, 512, 512, 20)
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); print("ms per frame", (time.perf_counter() - t0) * 1e3 / 20)
pr = cProfile.Profile(); pr.enable(); run(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
