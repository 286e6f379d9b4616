"""Kernel sequence of one downstream transformer block (forward and backward) from a rocprofv3 kernel trace.
    python tools/block_trace.py <kernel_trace.csv> [step_index] [block_index]"""
import csv
import re
import sys


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', n))
    n = re.sub(r'at::native::', '', n)
    return n[:110]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    blk = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    fps = [i for i, r in enumerate(rows) if 'fps_kernel' in r['Kernel_Name']]
    groups = [fps[i:i + 7] for i in range(0, len(fps) - 7, 7)]
    a, b = groups[k][6], groups[k + 1][0]
    seg = rows[a:b]
    for title, pat in (('forward block', 'attn_fwd_mfma'), ('backward block', 'attn_bwd_mfma')):
        at = [i for i, r in enumerate(seg) if pat in r['Kernel_Name']]
        lo, hi = at[blk], at[blk + 1]
        t0 = int(seg[lo]['Start_Timestamp'])
        print('---- %s: %d kernels, wall %.1f us' % (title, hi - lo, (int(seg[hi]['Start_Timestamp']) - t0) / 1e3))
        for r in seg[lo:hi]:
            print('  %7.1f  %6.1f us  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                                            short(r['Kernel_Name'])))


if __name__ == '__main__':
    main()
