"""Per-level kernel timeline from a rocprofv3 kernel_trace.csv: kernels of the last full level, durations and gaps."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
# find the last two launches of the Gram kernel
idx = [i for i, r in enumerate(rows) if 'xty_kernel' in r['Kernel_Name'] or 'xty_bf16_kernel' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
prev_end = None
for r in rows[a:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{r['Kernel_Name'][:70]:70s} dur {(e - s) / 1e3:7.1f} us  gap {gap:6.1f} us")
    prev_end = e
print('level period', (int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3, 'us')
