mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2_t1.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r2_t1.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 60 tools/hwtests/trig_err > gpurun_out/r2_trig.log 2>&1; cat gpurun_out/r2_trig.log
POSEGEN_HIP_LIB=$PWD/build_ab/lib_slp32.so timeout -k 10 300 python tests/diag/x3_slp_diag.py > gpurun_out/r2_x3_slp.log 2>&1; rc=$?; echo "x3 slp rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python tests/diag/x3_slp_diag.py > gpurun_out/r2_x3_noslp.log 2>&1; rc=$?; echo "x3 noslp rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python bench.py > gpurun_out/r2_bench0.json 2> gpurun_out/r2_bench0.err; echo "bench rc=$?"; head -c 1500 gpurun_out/r2_bench0.json
