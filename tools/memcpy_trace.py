"""Which aten ops issue device-to-device memcpys (hipMemcpyAsync nodes, ~3-6 us each in the graph) in one training step?
    python tools/memcpy_trace.py [cls|seg]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "iccv2025-upp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

import bench  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "cls"
dev = torch.device("cuda", 0)
tr = bench.Trainer(dev, 32, False, use_graph=False) if kind == "cls" else bench.RecipeTrainer(kind, dev, 32, use_graph=False)
for _ in range(3):
    tr.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step()
    torch.cuda.synchronize()
evs = list(prof.events())


def chain(e):
    out = []
    while e is not None:
        if e.name.startswith("aten::") or "Function" in e.name or "Backward" in e.name:
            out.append(e.name.replace("aten::", ""))
        e = e.cpu_parent
    return " < ".join(out[:5])


agg = collections.defaultdict(lambda: [0, 0.0])
for e in evs:
    if e.name != "aten::copy_" or e.self_device_time_total <= 0:
        continue
    kernels = [k.name for k in e.kernels]
    if not any("Memcpy" in k or "copyBuffer" in k for k in kernels):
        continue
    key = (chain(e), str([tuple(s) for s in e.input_shapes[:2] if s]))
    agg[key][0] += 1
    agg[key][1] += e.self_device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("D2D memcpys: %d calls, %.1f us" % (sum(v[0] for _, v in rows), sum(v[1] for _, v in rows)))
for (c, shp), (n, t) in rows[:40]:
    print("%7.1f us %3d  %-60s %s" % (t, n, c[:60], shp))
