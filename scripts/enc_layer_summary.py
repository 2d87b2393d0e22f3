"""Per-position kernel times inside one encoder layer, from a rocprofv3 --kernel-trace CSV of
scripts/prof_encoder.py (last pass): python scripts/enc_layer_summary.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "pack_mel" in r["Kernel_Name"]][-1]
seq = [(r["Kernel_Name"][:44], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows[first:]]
print("last pass: %d kernels, %.1f ms of kernel time" % (len(seq), sum(d for _, d in seq) / 1e3))
marks = [i for i, s in enumerate(seq) if "attn_encoder" in s[0]]
agg = collections.defaultdict(list)
for a, b in zip(marks[:-1], marks[1:]):
    for j, (name, d) in enumerate(seq[a:b]):
        agg[(j, name)].append(d)
for key in sorted(agg):
    v = agg[key]
    print("%d %-46s n=%d mean %8.1f us  sum %6.1f ms" % (key[0], key[1], len(v), sum(v) / len(v), sum(v) / 1e3))
