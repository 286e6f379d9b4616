"""Which kernels run just before / after every launch of a named kernel, from a rocprofv3 --kernel-trace CSV (stream order by start time).
   python tools/around_kernel.py <kernel_trace.csv> <substring> [last N launches]"""
import collections
import csv
import re
import sys


def short(n):
    return re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    pat = sys.argv[2]
    tail = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
    rows = rows[-tail:]
    agg = collections.Counter()
    for i, r in enumerate(rows):
        if pat in r["Kernel_Name"]:
            prev = short(rows[i - 1]["Kernel_Name"]) if i else "-"
            nxt = short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else "-"
            agg[(prev, nxt)] += 1
    for (p, n), c in agg.most_common(40):
        print("%5d  after %-70s before %s" % (c, p, n))


if __name__ == "__main__":
    main()
