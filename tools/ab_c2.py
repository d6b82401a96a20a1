"""A/B of library builds on ONE box for the compensated-fp16 eval launches (coarse S=64 + fine S=80, 512x512 benchmark frame):
interleaved rounds, one child process per build and round.   usage: ab_c2.py libA.so libB.so ..."""
import os, re, subprocess, sys
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(int(os.environ.get("ROUNDS", "2"))):
    for l in libs:
        env = dict(os.environ, POSEGEN_HIP_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, "tools/diag_evalc2.py", "time1"], capture_output=True, text=True, env=env).stdout
        m = re.findall(r"coarse\+fine ([0-9.]+) ms", out)
        res[l].append(float(m[0]) if m else float("nan"))
for l in libs:
    v = sorted(res[l])
    print(f"{os.path.basename(l):28s} coarse+fine eval ms: min {v[0]:.3f} median {v[len(v)//2]:.3f}  ({', '.join('%.3f' % x for x in res[l])})", flush=True)
