"""Kernel mix of one phase of a training step (phases delimited by the FPS launches, see phase_summary.py).
    python tools/phase_kernels.py <kernel_trace.csv> <phase 0..6> [step_index] [--seq]"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', n))
    return re.sub(r'at::native::', '', n)[:120]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ph = int(sys.argv[2])
    k = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 8
    fps = [i for i, r in enumerate(rows) if 'fps_kernel' in r['Kernel_Name']]
    groups = [fps[i:i + 7] for i in range(0, len(fps) - 7, 7)]
    g, gn = groups[k], groups[k + 1]
    a, b = g[ph], (g[ph + 1] if ph < 6 else gn[0])
    if '--seq' in sys.argv:
        t0 = int(rows[a]['Start_Timestamp'])
        for r in rows[a:b]:
            print('%8.1f %6.1f us  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                                          short(r['Kernel_Name'])))
        return
    c, t = collections.Counter(), collections.Counter()
    for r in rows[a:b]:
        n = short(r['Kernel_Name'])
        c[n] += 1
        t[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    for n, v in t.most_common(40):
        print('%8.1f us %4d  %s' % (v, c[n], n))


if __name__ == '__main__':
    main()
