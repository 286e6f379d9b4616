import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fps = [i for i, r in enumerate(rows) if 'fps_kernel' in r['Kernel_Name']]
groups = [fps[i:i + 7] for i in range(0, len(fps) - 7, 7)]
a, b = groups[8][6], groups[9][0]
seg = rows[a:b]
out = []
for i, r in enumerate(seg):
    n = r['Kernel_Name']
    if 'rowln_fwd' in n or 'rowln_bwd' in n:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        prev = seg[i - 1]['Kernel_Name'][:40] if i else ''
        out.append("%s %.1f (after %s) grid=%s" % ('F' if 'fwd' in n else 'B', d, prev.replace('(anonymous namespace)::', ''), r.get('Grid_Size', r.get('Grid_Size_X', '?'))))
print("\n".join(out))
