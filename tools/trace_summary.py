"""Summarise a rocprofv3 kernel_trace.csv: per-category GPU time, dispatch counts, idle gaps.
    python tools/trace_summary.py <kernel_trace.csv> [n_steps]
"""
import collections
import csv
import re
import sys

CATS = [
    ("fps", r"fps_kernel"), ("knn", r"knn_kernel"), ("upp_other", r"anonymous namespace\)::(?!fps|knn)"),
    ("gemm", r"^Cijk_|gemm|Gemm"), ("conv", r"conv|Conv"), ("batchnorm", r"batch_norm"), ("layernorm", r"layer_norm"),
    ("softmax", r"softmax|Softmax"), ("sort", r"rocprim|sort|Sort"), ("index", r"index|gather|scatter"),
    ("copy", r"direct_copy|copyBuffer|CatArray|copy_"), ("fill", r"FillFunctor|fillBuffer"), ("reduce", r"reduce_kernel"),
    ("optimizer", r"multi_tensor|adam|Adam"), ("rng", r"distribution|philox|random"),
    ("elementwise", r"elementwise|vectorized"),
]


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    tot, cnt = collections.Counter(), collections.Counter()
    names = collections.Counter()
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        n = r["Kernel_Name"]
        for c, pat in CATS:
            if re.search(pat, n):
                break
        else:
            c = "other"
        tot[c] += d
        cnt[c] += 1
        names[(c, re.sub(r"^void ", "", n)[:90])] += d
    T = sum(tot.values())
    span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
    print("dispatches %d  kernel time %.3f ms  (per step: %.0f dispatches, %.3f ms)" % (len(rows), T / 1e6, len(rows) / steps, T / 1e6 / steps))
    for c, v in tot.most_common():
        print("  %-12s %8.3f ms/step %5.1f%%  %6.0f launches/step  avg %.1f us" % (c, v / 1e6 / steps, 100 * v / T, cnt[c] / steps, v / cnt[c] / 1e3))
    print("top kernels:")
    for (c, n), v in names.most_common(25):
        print("  %8.3f ms/step  [%s] %s" % (v / 1e6 / steps, c, n))


if __name__ == "__main__":
    main()
