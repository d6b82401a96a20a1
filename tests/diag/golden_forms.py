"""Child process of test_other_kernel_forms_vs_reference_golden: the environment switches that select a kernel form
(POSEGEN_ONCHIP, POSEGEN_EVALC2) are read once per process, so every form renders the golden ray sets in a process of
its own and reports its errors against the reference's vectors as one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from posegen_amd import PREC_BY_NAME
from posegen_amd.raycaster import HipRayCaster
from tests.helpers import cfg_from_golden, load_golden, model_for


def main():
    out = {}
    for name in sys.argv[1].split(","):
        g = load_golden(name)
        cfg = cfg_from_golden(g)
        wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
        c = HipRayCaster.from_weights(cfg, wc, wf, tv, td, device="cuda:0", precision="fp32")
        rb, skts, cyl = torch.tensor(g["ray_batch"]), torch.tensor(g["skts"]), torch.tensor(g["cyl"])
        cams = torch.tensor(g["cams"]) if "cams" in g else None
        for prec in sys.argv[2].split(","):
            c.renderer.set_precision(PREC_BY_NAME[prec])
            c.renderer.profile_enable(True)
            c.renderer.profile_read(); c.renderer.profile_read_aux()
            res = c(rb, N_samples=cfg.n_samples, skts=skts, cyls=cyl, cams=cams, N_importance=cfg.n_importance)
            torch.cuda.synchronize()
            launches, _, _ = c.renderer.profile_read()
            rec_launches, _ = c.renderer.profile_read_aux()
            c.renderer.profile_enable(False)
            keys = ["rgb_map", "acc_map", "disp_map"] + (["rgb0", "acc0"] if cfg.n_importance > 0 else [])
            out[f"{name}:{prec}"] = dict({k: float(np.nanmax(np.abs(res[k].cpu().numpy().astype(np.float64) - g[k]))) for k in keys},
                                         eval_launches=launches, record_launches=rec_launches)
        c.renderer.close()
    print("GOLDEN_FORMS " + json.dumps(out))


if __name__ == "__main__":
    main()
