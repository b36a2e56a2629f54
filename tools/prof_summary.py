#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short CSV (names cut at 100 chars)."""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
with open(dst, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows:
        name = r['Name']
        if len(name) > 100:
            name = name[:97] + '...'
        w.writerow([name, r['Calls'], r['TotalDurationNs'], f"{float(r['AverageNs']):.0f}", r['Percentage'],
                    r['MinNs'], r['MaxNs']])
