"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, as
MI355X_MICROARCH.md prescribes) of tools/prof_kernels.py, plus the kernel-trace pass for the durations.
    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel_trace.csv> > profiles/r01_pmc_kernels.json
Key = demangled kernel name up to its argument list; values = medians over the launches:
us, fetch_kb_raw (FETCH_SIZE as reported, KB -- bench.py doubles it per the gfx950 note in the guide), write_kb."""
import collections
import csv
import json
import re
import statistics
import sys


LINEAR_LABELS = ["qkv", "proj", "fc1_gelu_d", "fc2", "dfc2_mul", "dfc1", "dproj", "dqkv"]      # bench.LINEAR_SHAPES, launch order of prof_kernels.py


def key(r):
    """demangled name without return type, anonymous-namespace qualifier and argument list"""
    n = re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', r['Kernel_Name']))
    return n.split('(')[0]


def labelled(rows, id_field):
    """(key, row) in dispatch order; the i-th linear_f32_kernel dispatch gets the key 'linear:<label of shape i % 8>'."""
    rows = sorted(rows, key=lambda r: int(r[id_field]))
    i = 0
    for r in rows:
        k = key(r)
        if k.startswith('linear_f32_kernel'):
            k = 'linear:' + LINEAR_LABELS[i % len(LINEAR_LABELS)]
            i += 1
        yield k, r


def counters(path, name):
    out = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == name]
    for k, r in labelled(rows, 'Dispatch_Id'):
        out[k].append(float(r['Counter_Value']))
    return out


def main():
    fetch, write = counters(sys.argv[1], 'FETCH_SIZE'), counters(sys.argv[2], 'WRITE_SIZE')
    dur = collections.defaultdict(list)
    for k, r in labelled(list(csv.DictReader(open(sys.argv[3]))), 'Dispatch_Id'):
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    out = {}
    for k in sorted(fetch):
        if k.startswith(('at::', '__amd', 'Cijk', 'rocprim', 'elementwise', 'randperm', 'softmax_warp')):
            continue
        out[k] = {'us': statistics.median(dur[k]) if k in dur else None, 'fetch_kb_raw': statistics.median(fetch[k]),
                  'write_kb': statistics.median(write[k]) if k in write else None, 'launches': len(fetch[k])}
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
