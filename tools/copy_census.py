"""Which device-to-device copies does one pipelined step issue (torch profiler, eager front-end + back-end)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
import bench
dev = torch.device("cuda", 0)
tr = bench.Trainer(dev, 32, False, use_graph=True, pipeline=True)
ts = tr.ts
for _ in range(2):
    ts._front(0); ts._back(0)
torch.cuda.synchronize()
for name, fn in (("front", lambda: ts._front(0)), ("back", lambda: ts._back(0))):
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
        fn()
        torch.cuda.synchronize()
    rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::copy_", "aten::_foreach_copy_", "aten::contiguous", "aten::clone", "aten::cat", "aten::_to_copy")]
    print("==", name)
    for e in sorted(rows, key=lambda e: -e.count)[:25]:
        print("  %-22s x%-3d %s" % (e.key, e.count, str(e.input_shapes)[:150]))
    k = [e for e in prof.key_averages() if "copyBuffer" in e.key or "Memcpy" in e.key]
    for e in k:
        print("  kernel %-40s x%d  %.1f us total" % (e.key[:40], e.count, e.device_time_total))
