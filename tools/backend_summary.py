"""Kernel time of the trainable back-end (downstream 12 blocks fwd + bwd + optimizer) of one step, by kernel, from a rocprofv3
kernel trace of `bench.py --no-pipeline` (FPS launches are the phase markers, as in phase_summary.py).
    python tools/backend_summary.py <kernel_trace.csv> [step_index]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
fps = [i for i, r in enumerate(rows) if 'fps_kernel' in r['Kernel_Name']]
groups = [fps[i:i + 7] for i in range(0, len(fps) - 7, 7)]
a, b = groups[k][6], groups[k + 1][0]
# skip the downstream group+encoder part: up to and including the 4th gemm_f32_kernel after the marker
g = [i for i in range(a, b) if 'gemm_f32_kernel' in rows[i]['Kernel_Name']]
start = g[3] + 1 if len(g) >= 4 else a
c, t = collections.Counter(), collections.Counter()
for r in rows[start:b]:
    n = re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', r['Kernel_Name'])).split('(')[0][:70]
    c[n] += 1
    t[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(t.values())
wall = (int(rows[b - 1]['End_Timestamp']) - int(rows[start]['Start_Timestamp'])) / 1e3
print("back-end: %d launches, busy %.1f us, wall %.1f us (gaps %.1f us)" % (b - start, tot, wall, wall - tot))
for n, v in t.most_common(28):
    print("  %8.1f us %4d  avg %6.1f  %s" % (v, c[n], v / c[n], n))
