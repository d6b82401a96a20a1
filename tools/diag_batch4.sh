#!/bin/bash
cd "$(dirname "$0")/.."
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-modes --no-extras --no-strong"
for rnd in 1 2; do
for v in records old32 old16; do
  unset POSEGEN_MFMA POSEGEN_RECORDS
  [ $v = old32 ] && export POSEGEN_RECORDS=0
  [ $v = old16 ] && export POSEGEN_RECORDS=0 POSEGEN_MFMA=16
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', 'ms/frame %.2f eval launch ms %.2f frac %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done
done
