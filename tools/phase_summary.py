"""Per-phase busy time of one training step from a rocprofv3 kernel trace (FPS launches are the phase markers).
    python tools/phase_summary.py <kernel_trace.csv> [step_index]"""
import collections
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    fps = [i for i, r in enumerate(rows) if 'fps_kernel' in r['Kernel_Name']]
    groups = [fps[i:i + 7] for i in range(0, len(fps) - 7, 7)]
    g, gn = groups[k], groups[k + 1]

    def seg(a, b):
        t = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows[a:b])
        return t / 1e3, b - a, (int(rows[b - 1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3

    names = ['rectify: group+enc+3 blocks', 'rectify prompter', 'complete: group+enc+6 blocks+decoder', 'fps 1024->256',
             'fps 1228->1024', 'downstream group+enc', 'downstream 12 blocks fwd+bwd+opt']
    for i in range(7):
        a, b = g[i], (g[i + 1] if i < 6 else gn[0])
        print('%-40s busy %8.1f us  kernels %5d  wall %8.1f us' % ((names[i],) + seg(a, b)))
    a, b = g[6], gn[0]
    fb = next(i for i in range(a, b) if re.search('ackward|_grad|_bwd', rows[i]['Kernel_Name']))
    print('  downstream fwd  busy %.1f us kernels %d wall %.1f' % seg(a, fb))
    print('  bwd + optimizer busy %.1f us kernels %d wall %.1f' % seg(fb, b))
    print('step wall %.1f us' % ((int(rows[gn[0]]['Start_Timestamp']) - int(rows[g[0]]['Start_Timestamp'])) / 1e3))
    for title, lo, hi in (('forward (all phases)', g[0], fb), ('backward+opt', fb, b)):
        c, t = collections.Counter(), collections.Counter()
        for r in rows[lo:hi]:
            n = re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', r['Kernel_Name']))[:86]
            c[n] += 1
            t[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        print(title)
        for n, v in t.most_common(16):
            print('   %7.1f us %4d  %s' % (v, c[n], n))


if __name__ == '__main__':
    main()
