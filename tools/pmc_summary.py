"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, as
MI355X_MICROARCH.md prescribes) of tools/prof_kernels.py, plus the kernel-trace pass for the durations.
    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel_trace.csv> > profiles/r04_pmc_kernels.json
Key = demangled kernel name up to its argument list; values = medians over the launches:
us, fetch_kb_raw (FETCH_SIZE as reported, KB -- bench.py doubles it per the gfx950 note in the guide), write_kb.

Labelled Linear launches: prof_kernels.py puts a MARKER dispatch (transpose_kernel) in front of each; the first linear_sb_kernel /
linear_f32_kernel dispatch behind the i-th marker gets the key 'linear:<label>' (split-bf16 kernel) or 'linear_f32:<label>' (exact-f32
kernel) of LINEAR_ORDER[i % len(LINEAR_ORDER)], and the entry carries the kernel name it was measured on.  Self-check (exit code 2 on failure): the bytes
written by a labelled launch must be its M x N x 4 (x 2 with the GELU' output) within 10 %, and its traffic at least 0.9 x its algorithmic
bytes -- a rotated join (round 3) fails both."""
import collections
import csv
import json
import os
import re
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import LINEAR_SHAPES, STEP_LINEAR_SHAPES, linear_algorithmic_bytes  # noqa: E402
LINEAR_ORDER = [("linear:", s) for s in STEP_LINEAR_SHAPES] + [("linear_f32:", s) for s in LINEAR_SHAPES]      # prof_kernels.py: frozen=True (every block shape), then False (M = 2400)
MARKER = 'transpose_kernel'


def name(r):
    """demangled name without return type and anonymous-namespace qualifier, with its template arguments"""
    return re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', r['Kernel_Name'])).split('(')[0]


def labelled(rows, id_field):
    """(key, row) in dispatch order; see the module docstring for the labelled Linear launches."""
    rows = sorted(rows, key=lambda r: int(r[id_field]))
    markers, pending = 0, None
    for r in rows:
        k = name(r)
        if k.startswith(MARKER):
            pending = LINEAR_ORDER[markers % len(LINEAR_ORDER)]
            markers += 1
            continue
        if pending is not None and k.startswith(('linear_sb_kernel', 'linear_f32_kernel')):
            prefix, shape = pending
            pending = None
            k = prefix + shape[0]
        yield k, r


def counters(path, cname):
    out = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == cname]
    for k, r in labelled(rows, 'Dispatch_Id'):
        out[k].append(float(r['Counter_Value']))
    return out


def main():
    fetch, write = counters(sys.argv[1], 'FETCH_SIZE'), counters(sys.argv[2], 'WRITE_SIZE')
    dur, kern = collections.defaultdict(list), {}
    for k, r in labelled(list(csv.DictReader(open(sys.argv[3]))), 'Dispatch_Id'):
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        kern.setdefault(k, name(r))
    out, bad = {}, []
    for k in sorted(fetch):
        if k.startswith(('at::', '__amd', 'Cijk', 'rocprim', 'elementwise', 'randperm', 'softmax_warp')):
            continue
        out[k] = {'us': statistics.median(dur[k]) if k in dur else None, 'fetch_kb_raw': statistics.median(fetch[k]),
                  'write_kb': statistics.median(write[k]) if k in write else None, 'launches': len(fetch[k])}
    for prefix, (label, M, N, K, epi) in LINEAR_ORDER:
        k = prefix + label
        if k not in out:
            bad.append("%s: no labelled dispatch found" % k)
            continue
        e = out[k]
        e['kernel'] = kern.get(k)
        e['shape'] = [M, N, K, epi]
        split = k.startswith('linear:') and (e['kernel'] or '').startswith('linear_sb_kernel')
        alg = linear_algorithmic_bytes(M, N, K, epi, split)
        e['algorithmic_bytes'] = alg
        e['traffic_bytes'] = (2.0 * e['fetch_kb_raw'] + e['write_kb']) * 1024.0
        want_w = M * N * 4.0 * (2 if epi == 3 else 1) / 1024.0
        if abs(e['write_kb'] - want_w) > 0.10 * want_w:
            bad.append("%s: wrote %.0f KB, the shape writes %.0f KB" % (k, e['write_kb'], want_w))
        if e['traffic_bytes'] < 0.9 * alg:
            bad.append("%s: traffic %.0f B below its algorithmic %.0f B" % (k, e['traffic_bytes'], alg))
    json.dump(out, sys.stdout, indent=1)
    if bad:
        print("pmc_summary: labelled Linear launches fail the self-check:\n  " + "\n  ".join(bad), file=sys.stderr)
        sys.exit(2)


if __name__ == '__main__':
    main()
