#!/bin/bash
# per-kernel table of one end-to-end tool under rocprofv3: tools/e2e_profile_one.sh <name> <script> [args...]
# (run on the GPU box from the repo root; keeps gpurun_out/p_<name>_kernel_stats.csv)
set -e
cd /tmp && export TMPDIR=/tmp NIF_FAST_TWIN_COPY=1
R=$GRAFT_REPO_ROOT
name=$1; shift
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_$name -- python3 "$@" > $R/gpurun_out/p_$name.log 2>&1
echo "== $name: $(grep -h 'run 1' $R/gpurun_out/p_$name.log | tail -1)"
python3 - "$R/gpurun_out/p_$name" <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0])))
for r in rows[:30]:
    print('  ', r['Name'][:100].ljust(100), r['Calls'].rjust(6), '%8.2f ms' % (float(r['TotalDurationNs']) / 1e6), '%9.1f us' % (float(r['AverageNs']) / 1e3))
PY
cp $R/gpurun_out/p_$name/*/*kernel_stats.csv $R/gpurun_out/p_${name}_kernel_stats.csv
rm -rf $R/gpurun_out/p_$name
