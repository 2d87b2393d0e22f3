"""Mean of one counter per kernel from a `rocprofv3 --pmc ... --output-format csv` run.
python scripts/pmc_summary.py <counter_collection.csv> [name filter]"""
import collections, csv, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
print("kernel,counter,launches,mean,max")
for k in sorted(rows):
    if flt in k:
        for c, v in sorted(rows[k].items()):
            print('"%s",%s,%d,%.2f,%.2f' % (k[:70], c, len(v), sum(v) / len(v), max(v)))
