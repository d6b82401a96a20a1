#!/bin/bash
# round-3 diagnosis batch: stamps + timing ablations of the bf16 eval launch (one box, back to back)
cd "$(dirname "$0")/.."
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-modes --no-extras --no-strong"
for l in default noread hidden; do
  if [ $l = default ]; then unset POSEGEN_HIP_LIB; else export POSEGEN_HIP_LIB=$PWD/build_ab/lib_$l.so; fi
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$l', 'ms/frame %.2f eval launch ms %.2f frac %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done
POSEGEN_HIP_LIB=$PWD/build_ab/lib_stamps.so python tools/diag_stamps.py
