"""Sustained MFMA rate of this box by shape (32x32x16 vs 16x16x32), operands in registers or the A operand fed from LDS."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posegen_amd import surreal_config, synthetic as syn
from posegen_amd.raycaster import HipRayCaster
cfg = surreal_config()
c = HipRayCaster.from_weights(cfg, *syn.make_model(cfg, 0), device="cuda:0", precision="bf16")
r = c.renderer
for rnd in range(2):
    for name, fed in (("32x32x16 regs", 0), ("32x32x16 lds-fed", 1), ("16x16x32 regs", 2), ("16x16x32 lds-fed", 3)):
        cal = r.calibrate_mfma(f16=False, lds_fed=fed)
        print(f"{name:18s} {cal['tflops']:.0f} TFLOP/s ({cal['ms']:.1f} ms)", flush=True)
