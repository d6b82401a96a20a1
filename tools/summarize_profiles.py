#!/usr/bin/env python3
"""Condense one tools/collect_profiles.sh output directory into the files committed under profiles/:

    python tools/summarize_profiles.py gpurun_out/prof_r2_bf16 profiles/r2_bf16

writes <prefix>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, verbatim),
<prefix>_pmc.csv (per kernel: launches and per-launch mean of every counter, summed over
dimensions/XCDs as rocprofv3 reports them) and <prefix>_traffic.json (HBM bytes per eval launch:
2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md's HBM section).
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def counters(path):
    """{kernel: {counter: (sum over launches, launches)}} of one counter_collection.csv"""
    per = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            per[k][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[k].add(row["Dispatch_Id"])
    return per, {k: len(v) for k, v in launches.items()}


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    stats = sorted(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], prefix + "_kernel_stats.csv")        # (gpurun merges directories: the newest run counts)
    rows = []
    merged = defaultdict(dict)
    nl = {}
    for sub in ("fetch", "write", "mfma", "sq", "lds"):
        for path in sorted(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:
            per, launches = counters(path)
            for k, cs in per.items():
                for c, v in cs.items():
                    merged[k][c] = v / launches[k]
                nl[k] = launches[k]
    names = sorted({c for cs in merged.values() for c in cs})
    with open(prefix + "_pmc.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches_per_pass"] + [n + "_per_launch" for n in names])
        for k in sorted(merged, key=lambda k: -merged[k].get("GRBM_GUI_ACTIVE", 0)):
            w.writerow([k, nl[k]] + [f"{merged[k].get(n, float('nan')):.6g}" for n in names])
    ev = [k for k in merged if "eval" in k and "kernel" in k]
    if ev:
        k = max(ev, key=lambda k: merged[k].get("GRBM_GUI_ACTIVE", 0))
        m = merged[k]
        out = {"kernel": k, "launches_profiled": nl[k],
               "FETCH_SIZE_KB": m.get("FETCH_SIZE"), "WRITE_SIZE_KB": m.get("WRITE_SIZE"),
               "hbm_bytes": (2 * m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) * 1024,
               "formula": "2 x FETCH_SIZE (gfx950 unit correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, "
                          "separate --pmc passes, mean over the eval launches of the pass (coarse + fine)",
               "counters_per_launch": {n: m[n] for n in sorted(m)}}
        # the whole frame: every kernel of the pass, launches per frame = launches / (eval launches / 2)
        frames = nl[k] / 2.0
        per_kernel = {kk: (2 * merged[kk].get("FETCH_SIZE", 0) + merged[kk].get("WRITE_SIZE", 0)) * 1024 * nl[kk] / frames
                      for kk in merged if "mfma_rate" not in kk}
        out["hbm_bytes_per_frame"] = sum(per_kernel.values())
        out["hbm_bytes_per_frame_by_kernel"] = {kk.split("(")[0][-60:]: v for kk, v in sorted(per_kernel.items(), key=lambda kv: -kv[1])}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; BUSY_CYCLES over all SIMDs
            cyc = m["GRBM_GUI_ACTIVE"] / 8
            out["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)
            out["gpu_cycles_per_launch"] = cyc
        json.dump(out, open(prefix + "_traffic.json", "w"), indent=1)
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
