#!/bin/bash
# kernel trace of one tool; prints the kernels before / after every launch whose name contains $1
# usage (GPU box, repo root): tools/trace_around.sh <pattern> <script> [args...]
set -e
cd /tmp && export TMPDIR=/tmp NIF_FAST_TWIN_COPY=1
R=$GRAFT_REPO_ROOT
pat=$1; shift
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_tmp -- python3 "$@" > $R/gpurun_out/tr_tmp.log 2>&1
python3 - "$R/gpurun_out/tr_tmp" "$pat" <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
hits = [i for i, r in enumerate(rows) if sys.argv[2] in r['Kernel_Name']]
print(len(rows), 'launches,', len(hits), 'hits')
for i in hits[:6]:
    for r in rows[max(0, i - 4):i + 3]:
        print('  ', r['Kernel_Name'][:90].ljust(90), r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], '%8.1f us' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
    print('   --')
PY
rm -rf $R/gpurun_out/tr_tmp
