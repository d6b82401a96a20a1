#!/bin/bash
cd "$(dirname "$0")/.."
python tools/diag_shapes.py
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-modes --no-extras --no-strong"
for v in default mfma16 waves4; do
  unset POSEGEN_MFMA POSEGEN_WAVES
  [ $v = mfma16 ] && export POSEGEN_MFMA=16
  [ $v = waves4 ] && export POSEGEN_WAVES=4
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', 'ms/frame %.2f eval launch ms %.2f frac %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done
