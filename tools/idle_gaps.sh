#!/bin/bash
# GPU idle time of a tool's LAST fusion from a rocprofv3 kernel trace: busy time (union of kernel intervals), idle time, and the
# largest idle gaps with the kernels before / after them.   usage (GPU box): tools/idle_gaps.sh <tail-fraction> <script> [args...]
set -e
cd /tmp && export TMPDIR=/tmp NIF_FAST_TWIN_COPY=1
R=$GRAFT_REPO_ROOT
frac=$1; shift
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ig_tmp -- python3 "$@" > $R/gpurun_out/ig_tmp.log 2>&1
python3 - "$R/gpurun_out/ig_tmp" "$frac" <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t_all0, t_all1 = int(rows[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in rows)
cut = t_all1 - (t_all1 - t_all0) * float(sys.argv[2])
rows = [r for r in rows if int(r['Start_Timestamp']) >= cut]
busy, gaps, end, last = 0, [], None, None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if end is None:
        end = s
    if s > end:
        gaps.append((s - end, last['Kernel_Name'][:60], r['Kernel_Name'][:60]))
        busy += e - s
        end = e
    elif e > end:
        busy += e - end
        end = e
    last = r if (last is None or e >= int(last['End_Timestamp'])) else last
span = end - int(rows[0]['Start_Timestamp'])
print(f'{len(rows)} launches over {span / 1e6:.1f} ms: busy {busy / 1e6:.1f} ms, idle {(span - busy) / 1e6:.1f} ms in {len(gaps)} gaps')
import collections
by = collections.Counter()
for g, a, b in gaps:
    by[(a, b)] += g
print('idle time by (kernel before -> kernel after):')
for (a, b), g in by.most_common(25):
    n = sum(1 for x in gaps if x[1] == a and x[2] == b)
    print(f'  {g / 1e6:8.2f} ms in {n:6d} gaps ({g / n / 1e3:7.1f} us each)  {a}  ->  {b}')
PY
rm -rf $R/gpurun_out/ig_tmp
