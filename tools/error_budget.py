"""Per-layer operand-precision error budget of the NeRF MLP (CPU, oracle emulation).

For every golden ray case: render with the fp32 oracle, then with the MFMA operand rounding of a
candidate precision recipe EMULATED layer by layer (products exact, fp32 accumulate), and report
max |d rgb|, |d acc|, |d disp| against the fp32 oracle.  The recipe is a dict
layer -> (activation mode, weight mode), mode in {"f32", "h" (fp16), "hh" (fp16 hi + fp16 lo),
"b" (bf16), "bb", "bbb"}; a layer costs (#x terms) x (#w terms) products minus the lo*lo ones.

    python tools/error_budget.py            # table for DESIGN.md section 3

Test infrastructure: uses oracle/ only.
"""
import itertools
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import anerf_oracle as orc  # noqa: E402
from tests.helpers import cfg_from_golden, load_golden, model_for, oracle_cfg, torch_weights  # noqa: E402

LAYERS = [f"L{i}" for i in range(8)] + ["alpha", "feat", "view", "rgb"]


def split(t, mode):
    """list of terms whose sum approximates t with the mode's operand format"""
    if mode == "f32":
        return [t]
    dt = torch.float16 if mode[0] == "h" else torch.bfloat16
    terms, rest = [], t
    for _ in mode:
        q = rest.to(dt).to(torch.float32)
        terms.append(q)
        rest = rest - q
    return terms


def comp_split(t, s):
    """fp16 pair (t1, t2) of the compensated product: t1 = fp16(t), t2 = fp16(t1 + s (t - t1))"""
    t1 = t.to(torch.float16).to(torch.float32)
    t2 = (t1 + s * (t - t1)).to(torch.float16).to(torch.float32)
    return t1, t2


def linear(x, w, b, mode):
    xm, wm = mode
    if xm == "c":                    # compensated fp16: acc = (s-1) W1 x1 + W2 x2 with W = w / s, one accumulator
        s_ = float(wm)
        w1, w2 = comp_split(w.double().div(s_).float(), s_)
        x1, x2 = comp_split(x, s_)
        return F.linear(x1, (s_ - 1) * w1) + F.linear(x2, w2) + b
    xs, ws = split(x, xm), split(w, wm)
    out = None
    for i, xt in enumerate(xs):
        for j, wt in enumerate(ws):
            if i + j >= max(len(xs), len(ws)) and i + j > 0:
                continue                 # drop lo*lo-order terms
            y = F.linear(xt, wt)
            out = y if out is None else out + y
    return out + b


def make_forward(recipe):
    def mlp_forward(x, weights, cfg):
        din, dv = cfg.ch_density_in, cfg.ch_d
        x_in, x_view = x[:, :din], x[:, din:din + dv]
        h = x_in
        for i in range(cfg.net_depth):
            h = F.relu(linear(h, weights[f"pts_linears.{i}.weight"], weights[f"pts_linears.{i}.bias"], recipe[f"L{i}"]))
            if i in cfg.skips:
                h = torch.cat([x_in, h], -1)
        sigma = linear(h, weights["alpha_linear.weight"], weights["alpha_linear.bias"], recipe["alpha"])
        feat = linear(h, weights["feature_linear.weight"], weights["feature_linear.bias"], recipe["feat"])
        if cfg.framecode_ch > 0:
            idx = x[:, din + dv]
            codes = weights["framecodes.codes.weight"]
            code = codes.mean(0, keepdim=True).expand(x.shape[0], -1) if idx.max() < 0 else codes[idx.long()]
            x_view = torch.cat([x_view, code], -1)
        g = F.relu(linear(torch.cat([feat, x_view], -1), weights["views_linears.0.weight"],
                          weights["views_linears.0.bias"], recipe["view"]))
        rgb = linear(g, weights["rgb_linear.weight"], weights["rgb_linear.bias"], recipe["rgb"])
        return torch.cat([rgb, sigma], -1)
    return mlp_forward


def render(g, cfg, forward=None):
    wc, wf, tv, td = model_for(cfg, int(g["seed_model"]))
    ocfg = oracle_cfg(cfg, g["tau_v"], g["tau_d"])
    cams = torch.tensor(g["cams"]) if "cams" in g else None
    keep = orc.mlp_forward
    if forward is not None:
        orc.mlp_forward = forward
    try:
        return orc.render_rays(torch.tensor(g["ray_batch"]), torch.tensor(g["skts"]), torch.tensor(g["cyl"]),
                               ocfg, torch_weights(wc), torch_weights(wf), cfg.n_samples, cfg.n_importance,
                               cams=cams, return_extras=False)
    finally:
        orc.mlp_forward = keep


def uniform(mode):
    return {k: mode for k in LAYERS}


def errs(ref, out):
    r = {}
    for k in ("rgb_map", "acc_map", "disp_map"):
        d = (out[k] - ref[k]).abs()
        d = d[~torch.isnan(d)]
        r[k] = float(d.max())
    return r


def main():
    cases = [a for a in sys.argv[1:] if not a.startswith("--")] or ["rays_surreal", "rays_h36m", "rays_allhit", "rays_cfg1", "rays_coarse32"]
    H, HH, F32 = ("h", "h"), ("hh", "hh"), ("f32", "f32")
    recipes = {"fp16 all": uniform(H)}
    for lay in (LAYERS if "--layers" in sys.argv else []):                   # fp16 everywhere except one layer exact
        r = uniform(H); r[lay] = F32
        recipes[f"fp16, {lay} exact"] = r
    for lay in (LAYERS if "--layers" in sys.argv else []):                   # exact everywhere except one layer fp16
        r = uniform(F32); r[lay] = H
        recipes[f"exact, {lay} fp16"] = r
    recipes["fp16 x:h w:hh all"] = uniform(("h", "hh"))
    recipes["fp16 x:hh w:h all"] = uniform(("hh", "h"))
    recipes["fp16 x3 all"] = uniform(HH)
    for sc in (17, 33, 65, 129, 257):
        recipes[f"fp16 compensated s={sc}"] = uniform(("c", str(sc)))
    for name, rec in recipes.items():
        line = []
        for c in cases:
            g = load_golden(c)
            cfg = cfg_from_golden(g)
            ref = render(g, cfg)
            e = errs(ref, render(g, cfg, make_forward(rec)))
            line.append(f"{e['rgb_map']:.1e}/{e['acc_map']:.1e}/{e['disp_map']:.1e}")
        print(f"{name:24s} " + "  ".join(line), flush=True)


if __name__ == "__main__":
    main()
