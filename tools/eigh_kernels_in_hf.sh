#!/bin/bash
# durations of the eigensolver's kernels inside HF fusions of two VGG11s (two fusions per run; run on the GPU box
# from the repo root)
set -e
cd /tmp && export TMPDIR=/tmp NIF_FAST_TWIN_COPY=1
out=$GRAFT_REPO_ROOT/gpurun_out/eigh_in_hf
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/e2e_vgg.py --algo hf > $out.log 2>&1
grep 'run 1' $out.log
python3 - "$out" <<'PY'
import csv, glob, sys
for r in csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0])):
    if any(k in r['Name'] for k in ('tri_', 'bt_', 'trd_', 'lap_aug')):
        print(r['Name'][:50].ljust(50), r['Calls'].rjust(6), '%8.2f ms' % (float(r['TotalDurationNs']) / 1e6),
              '%8.1f us' % (float(r['AverageNs']) / 1e3))
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0])))
for nm in ('tri_ortho', 'tri_invit', 'tri_eigenvalues'):
    sel = [r for r in rows if nm in r['Kernel_Name']]
    print(nm, 'per layer, us:', [round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in sel][-9:])
PY
