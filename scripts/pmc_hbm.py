"""HBM bytes per launch and kernel from two separate `rocprofv3 --pmc` runs (FETCH_SIZE, WRITE_SIZE: the
two do not fit one pass) of the same command, written in the format of profiles/*_pmc_hbm.csv.
bytes_corrected: FETCH_SIZE KiB x 1024 x 2 (gfx950 tallies the 128-byte read requests of wide coalesced
loads at 64 bytes each), WRITE_SIZE KiB x 1024.
python scripts/pmc_hbm.py "<what ran>" fetch_counter_collection.csv write_counter_collection.csv [name filter]"""
import collections, csv, sys
what, files = sys.argv[1], sys.argv[2:4]
flt = sys.argv[4] if len(sys.argv) > 4 else ""
print("# %s; two separate rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE)" % what)
print("# bytes_corrected: FETCH_SIZE KiB x 1024 x 2 (gfx950 tallies 128-byte read requests at 64 bytes), WRITE_SIZE KiB x 1024")
print("kernel,counter,launches,mean_per_launch_KiB,bytes_corrected")
for path in files:
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            grid = "x".join(str(int(r[k]) // max(1, int(r[k.replace("Grid", "Workgroup")]))) for k in ("Grid_Size_X", "Grid_Size_Y") if k in r) if "Grid_Size_X" in r else ""
            rows[(r["Kernel_Name"], grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, grid) in sorted(rows):
        if flt in k:
            for c, v in sorted(rows[(k, grid)].items()):
                mean = sum(v) / len(v)
                print('"%s%s",%s,%d,%.2f,%d' % (k[:70], (" [grid %s]" % grid) if grid else "", c, len(v), mean,
                                                 round(mean * 1024 * (2 if c == "FETCH_SIZE" else 1))))
