#!/bin/bash
# kernel timeline (start / end in us since the first listed kernel) of the last $1 launches of a tool
# usage (GPU box, repo root): tools/trace_timeline.sh <count> <script> [args...]
set -e
cd /tmp && export TMPDIR=/tmp NIF_FAST_TWIN_COPY=1
R=$GRAFT_REPO_ROOT
cnt=$1; shift
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_tmp -- python3 "$@" > $R/gpurun_out/tl_tmp.log 2>&1
python3 - "$R/gpurun_out/tl_tmp" "$cnt" <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-int(sys.argv[2]):]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    print('%10.1f %10.1f  q%-3s %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:70]))
PY
rm -rf $R/gpurun_out/tl_tmp
