#!/bin/bash
# per-kernel tables of the end-to-end fusions (run on the GPU box from the repo root); prints the top kernels of each
set -e
cd /tmp && export TMPDIR=/tmp NIF_FAST_TWIN_COPY=1
R=$GRAFT_REPO_ROOT
run() {  # name, script, args...
  name=$1; shift
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_$name -- python3 "$@" > $R/gpurun_out/p_$name.log 2>&1
  echo "== $name: $(grep -h 'run 1\|s  (' $R/gpurun_out/p_$name.log | tail -1)"
  python3 - "$R/gpurun_out/p_$name" <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0])))
for r in rows[:16]:
    print('  ', r['Name'][:86].ljust(86), r['Calls'].rjust(6), '%8.2f ms' % (float(r['TotalDurationNs']) / 1e6), '%9.1f us' % (float(r['AverageNs']) / 1e3))
PY
  cp $R/gpurun_out/p_$name/*/*kernel_stats.csv $R/gpurun_out/p_${name}_kernel_stats.csv
  rm -rf $R/gpurun_out/p_$name
}
run kf $R/tools/e2e_vgg.py --algo kf --models 4
run gr $R/tools/e2e_vgg.py --algo gr
run otf $R/tools/e2e_vgg.py --algo otf
run alf $R/tools/e2e_alf_vit.py
