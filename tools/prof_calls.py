"""Per-call kernel breakdown of nif_eigh_kept_f64 from a rocprofv3 kernel trace CSV (calls start at eigh_copy_in)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
calls, cur = [], None
for s, e, n in ev:
    if 'eigh_copy_in' in n:
        cur = {'t0': s, 'k': collections.OrderedDict(), 'tend': e}
        calls.append(cur)
    if cur is not None and 'nif::' in n:
        key = n.replace('nif::', '').replace('void ', '')[:22]
        d = cur['k'].setdefault(key, [0, 0]); d[0] += e - s; d[1] += 1; cur['tend'] = e
for c in calls:
    print('span %.1f ms' % ((c['tend'] - c['t0']) / 1e6), ' '.join(f'{k}={v[0] / 1e6:.2f}/{v[1]}' for k, v in c['k'].items()))
