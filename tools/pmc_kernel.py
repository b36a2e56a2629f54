"""Mean of every PMC counter per kernel name from a rocprofv3 --pmc counter_collection.csv."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
