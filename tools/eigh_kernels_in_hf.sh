#!/bin/bash
# duration of the eigenvector kernels inside one HF fusion of two VGG11s (run on the GPU box from the repo root)
set -e
cd /tmp && export TMPDIR=/tmp NIF_FAST_TWIN_COPY=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/invit_t -- python3 $GRAFT_REPO_ROOT/tools/e2e_vgg.py --algo hf > $GRAFT_REPO_ROOT/gpurun_out/invit_t.log 2>&1
grep 'run 1' $GRAFT_REPO_ROOT/gpurun_out/invit_t.log
grep -h "tri_\|bt_\|trd_" $GRAFT_REPO_ROOT/gpurun_out/invit_t/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
