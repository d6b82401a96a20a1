"""Typed renderer configuration.

Mirrors the renderer-relevant subset of the reference's configargparse flags
(reference run_nerf.py:186-490; values of configs/surreal/surreal.txt and
configs/h36m/h36m_prot2.txt).  Everything else in the reference's ~120 flags
concerns training / data loading and is out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Tuple

# precision modes of the fused embed+MLP kernel (include/posegen_hip.h PG_PREC_*)
PREC_FP32 = 0     # f32-input MFMA, exact fp32 chain (parity mode)
PREC_BF16 = 1     # bf16 operands, fp32 accumulate (throughput mode, BASELINE config 2)
PREC_BF16X3 = 2   # split-bf16: hi*hi + hi*lo + lo*hi, fp32 accumulate
PREC_FP16 = 3     # fp16 operands, fp32 accumulate
PREC_FP16X3 = 4   # split-fp16: hi*hi + hi*lo + lo*hi, fp32 accumulate
PREC_FP16C = 5    # compensated fp16: 128 w1 x1 + w2 x2 (two products, one fp32 accumulator), ~30x below fp16's error
PREC_FP16M = 6    # mixed: fp16c wherever a pass produces the returned maps, plain fp16 for the coarse pass of a
                  # hierarchical render (it only places the importance samples).  NOT a 1e-4 mode: the samples
                  # move with the coarse error (measured rgb 7e-5, acc 1.1e-4); bound 2e-4

PREC_NAMES = {PREC_FP32: "fp32", PREC_BF16: "bf16", PREC_BF16X3: "bf16x3",
              PREC_FP16: "fp16", PREC_FP16X3: "fp16x3", PREC_FP16C: "fp16c", PREC_FP16M: "fp16m"}
PREC_BY_NAME = {v: k for k, v in PREC_NAMES.items()}


@dataclass
class RenderConfig:
    n_joints: int = 24
    multires: int = 7              # --multires          (distance embedding)
    multires_views: int = 4        # --multires_views    (view embedding)
    multires_bones: int = 0        # --multires_bones    (identity bone embedding)
    net_depth: int = 8             # --netdepth
    net_width: int = 256           # --netwidth
    skips: Tuple[int, ...] = (4,)  # raycasters.py:82
    framecode_ch: int = 0          # --framecode_size when --opt_framecode
    n_framecodes: int = 0
    cutoff_mm: float = 500.0       # --cutoff_mm
    ext_scale: float = 0.001       # --ext_scale
    n_samples: int = 64            # --N_samples
    n_importance: int = 16         # --N_importance
    chunk: int = 4096              # --chunk
    density_scale: float = 1.0     # --density_scale
    rgb_eps: float = 1e-3          # nerf.py:151
    density_type: str = "relu"     # --density_type: 'relu' | 'softplus' (get_density_fn, raycasters.py:230-238)
    softplus_shift: float = 1.0    # --softplus_shift: softplus(x - shift)
    lindisp: bool = False
    white_bkgd: bool = True

    @property
    def cutoff_dist(self) -> float:
        return self.cutoff_mm * self.ext_scale

    @property
    def ch_v(self) -> int:
        return self.n_joints * (1 + 2 * self.multires)

    @property
    def ch_r(self) -> int:
        return self.n_joints * 3

    @property
    def ch_d(self) -> int:
        return self.n_joints * 3 * (1 + 2 * self.multires_views)

    @property
    def ch_density_in(self) -> int:
        return self.ch_v + self.ch_r

    @property
    def ch_view_in(self) -> int:
        return self.net_width + self.ch_d + self.framecode_ch

    def flops_per_point(self) -> int:
        """Algorithmic FLOP of one MLP point evaluation."""
        W, din = self.net_width, self.ch_density_in
        mac = din * W
        for i in range(self.net_depth - 1):
            mac += (W + din if i in self.skips else W) * W
        mac += W + W * W + self.ch_view_in * (W // 2) + (W // 2) * 3
        return 2 * mac

    def evals_per_ray(self) -> int:
        return self.n_samples + ((self.n_samples + self.n_importance) if self.n_importance > 0 else 0)

    def to_dict(self):
        return asdict(self)


def surreal_config(**kw) -> RenderConfig:
    """configs/surreal/surreal.txt"""
    return RenderConfig(**kw)


def h36m_config(**kw) -> RenderConfig:
    """configs/h36m/h36m_prot2.txt (frame codes on, 128 coarse samples per BASELINE config 4)."""
    base = dict(framecode_ch=16, n_framecodes=64, n_samples=128)
    base.update(kw)
    return RenderConfig(**base)
