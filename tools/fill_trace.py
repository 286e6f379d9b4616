"""Which aten ops launch fill / reduce kernels in one eager step of a recipe?  python tools/fill_trace.py pretrain"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "iccv2025-upp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from torch.profiler import profile, ProfilerActivity
import bench
kind = sys.argv[1] if len(sys.argv) > 1 else "pretrain"
dev = torch.device("cuda", 0)
tr = bench.Trainer(dev, 32, False, use_graph=False) if kind == "cls" else bench.RecipeTrainer(kind, dev, 32, use_graph=False)
for _ in range(3):
    tr.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step()
    torch.cuda.synchronize()


def chain(e):
    out = []
    while e is not None:
        out.append(e.name.replace("aten::", ""))
        e = e.cpu_parent
    return " < ".join(out[:6])


agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.self_device_time_total <= 0 or not e.name.startswith("aten::"):
        continue
    if e.name in ("aten::fill_", "aten::zero_", "aten::sum", "aten::copy_"):
        key = (chain(e)[:110], str([tuple(s) for s in e.input_shapes[:1] if s]))
        agg[key][0] += 1
        agg[key][1] += e.self_device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("%d calls, %.1f us" % (sum(v[0] for _, v in rows), sum(v[1] for _, v in rows)))
for (c, shp), (n, t) in rows[:40]:
    print("%7.1f us %3d  %-110s %s" % (t, n, c, shp))
