#!/bin/bash
# Round-3 evidence for bench.py's roofline object: kernel trace + two PMC passes of the default workload.
# Run on the GPU box from the repo root: bash tools/r3_profiles.sh   (writes under gpurun_out/)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export NIF_BENCH_PREHEAT_MS=300
B="$R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --hard-steps 5"
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_r3_bench -- python3 $B > $R/gpurun_out/p_r3_bench.log 2>&1
cp $R/gpurun_out/p_r3_bench/*/*kernel_stats.csv $R/gpurun_out/r3_f_bench_kernel_stats.csv
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_r3_fetch -- python3 $B > $R/gpurun_out/p_r3_fetch.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/p_r3_write -- python3 $B > $R/gpurun_out/p_r3_write.log 2>&1
cd $R
python3 tools/pmc_traffic.py gpurun_out/p_r3_fetch/*/*counter_collection.csv gpurun_out/p_r3_write/*/*counter_collection.csv \
  "xty_kernel<float, nif::MmaF32" "m=325000,n=384,f32" gpurun_out/r3_xty_traffic.json
head -12 gpurun_out/r3_f_bench_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/p_r3_bench gpurun_out/p_r3_fetch gpurun_out/p_r3_write
