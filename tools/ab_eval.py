"""A/B of library builds on ONE box: interleaved rounds of the bf16 eval launch (coarse S=64 and fine S=80 on the 512x512
benchmark frame), one child process per build and round.   usage: ab_eval.py libA.so libB.so ..."""
import sys, os, subprocess, re
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(int(os.environ.get("ROUNDS", "2"))):
    for l in libs:
        env = dict(os.environ, POSEGEN_HIP_LIB=os.path.abspath(l), ROWS="262144")
        out = subprocess.run([sys.executable, "tools/time_eval.py"], capture_output=True, text=True, env=env).stdout
        ms = [float(x) for x in re.findall(r"eval ([0-9.]+) ms", out)]
        res[l].append(sum(ms))
for l in libs:
    v = sorted(res[l])
    print(f"{os.path.basename(l):24s} coarse+fine eval ms: min {v[0]:.3f} median {v[len(v)//2]:.3f}  ({', '.join('%.3f' % x for x in res[l])})")
