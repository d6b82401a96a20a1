"""A/B of library builds on ONE box, interleaved rounds (cdna guide 5.4 rule 24).
usage: ab_bench.py libA.so libB.so [...]   -> median/min ms of the eval-dominated frame render"""
import sys, os, subprocess, json
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ, POSEGEN_HIP_LIB=os.path.abspath(l), **{k: v for k, v in (kv.split("=") for kv in os.environ.get("AB_ENV", "").split() if kv)})
        out = subprocess.run([sys.executable, "bench.py", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-modes"],
                             capture_output=True, text=True, env=env)
        d = json.loads(out.stdout.strip().split("\n")[-1])
        res[l].append((d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
for l in libs:
    ms = sorted(x[0] for x in res[l]); k = sorted(x[1] for x in res[l])
    print(f"{os.path.basename(l):28s} frame ms median {ms[len(ms)//2]:.2f} min {ms[0]:.2f} | eval launch ms median {k[len(k)//2]:.2f} min {k[0]:.2f} | frac max {max(x[2] for x in res[l]):.3f}")
