"""Which operators issue device-to-device runtime copies while a TrainStep is constructed and captured (outside the steady-state step)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch, bench
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr = bench.Trainer(torch.device("cuda", 0), 32, False, use_graph=True, pipeline=False)
    torch.cuda.synchronize()
agg = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and any("Memcpy DtoD" in k.name for k in e.kernels):
        if any(c.kernels for c in e.cpu_children if c.name.startswith("aten::")):
            continue
        frames = [f for f in (e.stack or []) if "iccv2025-upp_amd" in f or "bench.py" in f]
        agg[(e.name, str(e.input_shapes)[:80])] += 1
for k, v in agg.most_common(40):
    print(v, k)
