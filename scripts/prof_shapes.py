import csv, collections, sys
rows = csv.DictReader(open(sys.argv[1]))
d = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    d[(n[:60], r.get('Grid_Size') or 'x'.join(str(int(r[k]) // max(1, int(r.get(k.replace('Grid', 'Workgroup'), 1) or 1))) for k in ('Grid_Size_X', 'Grid_Size_Y', 'Grid_Size_Z')))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    v.sort()
    print('%-62s grid %-14s n %6d  med %7.1f us  total %7.1f ms %5.1f%%' % (k[0], k[1], len(v), v[len(v) // 2], sum(v) / 1e3, 100 * sum(v) / tot))
