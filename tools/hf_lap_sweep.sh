#!/bin/bash
# tools/hf_lap_stats.py under several settings of the augmenting row reduction (passes cap, chains left, zero bids)
set -e
out=gpurun_out/hf_lap_sweep.txt
: > $out
run() { echo "== $*" >> $out; env "$@" timeout -k 10 120 python tools/hf_lap_stats.py 2>/dev/null | tail -n 10 >> $out; }
run NIF_X=0
run NIF_LAP_ARR_PASSES=1024
run NIF_LAP_ARR_PASSES=4096
run NIF_LAP_ARR_PASSES=8192
run NIF_LAP_ARR_PASSES=8192 NIF_LAP_ARR_MIN_CHAINS=3
run NIF_LAP_ARR_MIN_CHAINS=3
run NIF_LAP_ARR_MIN_CHAINS=10
run NIF_LAP_ARR_MAX_WAIT=2
run NIF_LAP_ARR_MAX_WAIT=8
run NIF_LAP_ARR_PASSES=8192 NIF_LAP_ARR_MAX_WAIT=8
run NIF_LAP_ARR_PASSES=-1
