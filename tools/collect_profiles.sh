#!/bin/bash
# Round profiles on the GPU box: kernel stats + separate PMC passes of the default bench workload.
# usage: [PROG=tools/x.py] [PASSES="stats fetch write mfma"] tools/collect_profiles.sh <tag> [bench args...]   (writes gpurun_out/prof_<tag>/)
set -e -o pipefail
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# PROG=<script under the repo> profiles that program instead of the bench (tools/run_h36m.py: config 4's frame)
if [ -n "$PROG" ]; then B="python3 $root/$PROG"; else B="python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-modes --no-extras $*"; fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B > $out/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- $B > $out/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- $B > $out/write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $out/mfma -- $B > $out/mfma.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/sq -- $B > $out/sq.log 2>&1
find $out -name "*.csv" | head -40
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out/lds -- $B > $out/lds.log 2>&1
# issue / wait / scalar / instruction-fetch counters (profiles/r4_*_pmc_extra.csv): three more passes, EXTRA=1 to collect
if [ -n "$EXTRA" ]; then
  timeout -k 10 300 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $out/x1 -- $B > $out/x1.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL --kernel-trace --output-format csv -d $out/x2 -- $B > $out/x2.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/x3 -- $B > $out/x3.log 2>&1
fi
