"""Kernels of ONE replayed training step, from a rocprofv3 --kernel-trace CSV: everything between the last two launches of the optimizer
update (start-time order), aggregated by kernel name -- launches, microseconds, and whether the kernel is this library's.  Totals of a short
run divided by its step count also contain construction, warm-up and capture; this is the steady state.
   python tools/step_census.py <kernel_trace.csv> [delimiter-substring = adamw_update_kernel]"""
import collections
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    mark = sys.argv[2] if len(sys.argv) > 2 else "adamw_update_kernel"
    hits = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
    lo, hi = hits[-2] + 1, hits[-1] + 1
    agg = collections.OrderedDict()
    for r in rows[lo:hi]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    span = (int(rows[hi - 1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3
    own = {k: v for k, v in agg.items() if "at::native" not in k and not k.startswith("__amd") and not k.startswith("Cijk") and "rocprim" not in k.lower()}
    other = {k: v for k, v in agg.items() if k not in own}
    tot = sum(v[1] for v in agg.values())
    print("one step: %d launches, %.1f us of kernel time in a %.1f us window" % (sum(v[0] for v in agg.values()), tot, span))
    print("  this library: %d launches, %.1f us (%.1f %%); torch / runtime: %d launches, %.1f us" % (
        sum(v[0] for v in own.values()), sum(v[1] for v in own.values()), 100.0 * sum(v[1] for v in own.values()) / tot,
        sum(v[0] for v in other.values()), sum(v[1] for v in other.values())))
    for title, d in (("this library", own), ("torch / runtime", other)):
        print("-- " + title)
        for k, (c, us) in sorted(d.items(), key=lambda kv: -kv[1][1]):
            print("  %4d x %9.1f us  %s" % (c, us, k[:120]))


if __name__ == "__main__":
    main()
