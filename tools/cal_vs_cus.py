"""Bare-MFMA rate per CU against the number of busy CUs (POSEGEN_MAX_WG): how the chip's clock answers to load."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posegen_amd import surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
cfg = surreal_config()
c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device="cuda:0", precision="bf16")
n = int(os.environ.get("POSEGEN_MAX_WG", "256"))
for lds in (False, True):
    r = c.renderer.calibrate_mfma(f16=False, lds_fed=lds)
    print(f"CUs={n} lds_fed={lds}: {r['tflops']:.0f} TFLOP/s = {r['tflops'] / n:.2f} per CU ({r['ms']:.1f} ms)", flush=True)
