#!/bin/bash
# SQ counters of the f64-MFMA Gram kernel at two C1 shapes (what bounds its 0.6 of peak?): bash tools/xty_f64_pmc.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/xty_f64_pmc.txt
: > $out
for shape in "25600 2305" "6400 4609"; do
  set -- $shape
  timeout -k 10 100 python3 $R/tools/xty_bench.py --m $1 --n $2 --acc f64 --sym --iters 10 >> $out 2>&1 || exit 1
  timeout -s KILL 100 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/p_f64_$1 -- python3 $R/tools/xty_bench.py --m $1 --n $2 --acc f64 --sym --iters 5 > $R/gpurun_out/p_f64_$1.log 2>&1 || exit 1
  python3 $R/tools/pmc_kernel.py $R/gpurun_out/p_f64_$1/*/*counter_collection.csv "xty_kernel" >> $out
  rm -rf $R/gpurun_out/p_f64_$1
done
