"""Which Python lines launch the torch glue of a step (device copies, fills, element-wise kernels that are not ours)?

Runs one eager training step (front-end + back-end, no HIP graph) under torch.profiler with stacks and prints, per
(aten op, repo source line), the number of calls and the device time.  python tools/glue_trace.py [--top 40]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "iccv2025-upp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=45)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    tr = bench.Trainer(dev, 32, False, use_graph=False)
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        tr.step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    evs = list(prof.events())
    print("events %d, aten %d, with device time %d, with stack %d" % (
        len(evs), sum(e.name.startswith("aten::") for e in evs), sum(e.device_time_total > 0 for e in evs), sum(bool(e.stack) for e in evs)))
    for ev in evs:
        if not ev.name.startswith("aten::") or ev.self_device_time_total <= 0:
            continue
        stack = ev.stack or ["?"]
        where = next((f for f in stack if "iccv2025-upp_amd" in f or "upp_hip" in f or "bench.py" in f), stack[0])
        where = where.replace(ROOT + "/", "")
        if ev.name in ("aten::mm", "aten::addmm", "aten::bmm"):
            where = str([tuple(x) for x in ev.input_shapes if x])
        key = (ev.name, where)
        agg[key][0] += 1
        agg[key][1] += ev.self_device_time_total
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print("aten leaf ops with device time: %.1f us in %d calls" % (tot, sum(v[0] for _, v in rows)))
    for (name, where), (n, t) in rows[:args.top]:
        print("%8.1f us %4d  %-28s %s" % (t, n, name, where))


if __name__ == "__main__":
    main()
