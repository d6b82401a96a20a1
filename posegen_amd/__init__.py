"""posegen_amd -- MI355X-native A-NeRF renderer for the PoseGen loop.

Drop-in for the reference's volumetric rendering hot path
(`render_path` -> `render` -> `RayCaster.render_rays`): Python host code on
PyTorch-ROCm calls hand-written HIP kernels for gfx950 through a C-ABI shared
library (`include/posegen_hip.h`).  There is no CPU fallback: every compute
entry point raises if the HIP library is missing.
"""
from .config import (RenderConfig, surreal_config, h36m_config, PREC_FP32, PREC_BF16,
                     PREC_BF16X3, PREC_FP16, PREC_FP16X3, PREC_FP16C, PREC_FP16M, PREC_NAMES, PREC_BY_NAME)

__all__ = ["RenderConfig", "surreal_config", "h36m_config", "PREC_FP32", "PREC_BF16",
           "PREC_BF16X3", "PREC_FP16", "PREC_FP16X3", "PREC_FP16C", "PREC_FP16M", "PREC_NAMES", "PREC_BY_NAME"]
__version__ = "0.1.0"
