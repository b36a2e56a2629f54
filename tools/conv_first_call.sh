#!/bin/bash
# first and second fusion of a fresh process with torch's conv (MIOpen immediate mode) and with the im2row + GEMM conv
# of models/conv.py: tools/conv_first_call.sh  ->  gpurun_out/conv_first_call.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/conv_first_call.txt
: > $out
for algo in gr hf otf; do
  for flag in 0 1; do
    echo "== $algo NIF_CONV_GEMM=$flag" >> $out
    NIF_CONV_GEMM=$flag NIF_FAST_TWIN_COPY=1 timeout -k 10 120 python3 $R/tools/e2e_vgg.py --algo $algo >> $out 2>&1 || exit 1
  done
done
