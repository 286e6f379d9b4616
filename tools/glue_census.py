"""Every device kernel of one step that is NOT from libupp_hip.so, with the aten operator that launched it, its input shapes
and the nearest call site inside this package (torch profiler with stacks; eager front-end + back-end of the headline step).
    python tools/glue_census.py [--workload cls]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch  # noqa: E402

import bench  # noqa: E402

OURS = ("_kernel<", "_kernel(", "anonymous namespace")


def main():
    dev = torch.device("cuda", 0)
    tr = bench.Trainer(dev, 32, False, use_graph=True, pipeline=True)
    ts = tr.ts
    for _ in range(2):
        ts._front(0); ts._back(0)
    torch.cuda.synchronize()
    for name, fn in (("front", lambda: ts._front(0)), ("back", lambda: ts._back(0))):
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                                    record_shapes=True, with_stack=True,
                                    experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:      # (without it e.stack stays empty)
            fn()
            torch.cuda.synchronize()
        evs = prof.events()
        cpu = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU]
        agg = collections.OrderedDict()
        for e in cpu:
            ks = [k for k in e.kernels if not ("anonymous namespace" in k.name and "at::native" not in k.name)]
            if not ks or not e.name.startswith("aten::"):
                continue
            # leaf operators only (a parent aten op lists its children's kernels too)
            if any(c.kernels for c in e.cpu_children if c.name.startswith("aten::")):
                continue
            # (frames come as paths relative to their sys.path entry, innermost first: "models/upp_layers.py(1093): select")
            mine = [fr for fr in (e.stack or []) if fr.startswith(("models/", "upp_hip/", "utils/", "extensions/", "bench.py"))]
            site = " < ".join(fr.split(": ")[0] + ":" + fr.split(": ")[-1] for fr in mine[:2]) if mine else "?"
            if not mine:
                p_ = e.cpu_parent
                while p_ is not None and "Backward" not in p_.name:
                    p_ = p_.cpu_parent
                site = p_.name.replace("autograd::engine::evaluate_function: ", "") if p_ is not None else "?"
            key = (e.name, str(e.input_shapes)[:90], site[:110])
            t = sum(k.duration for k in ks)
            a = agg.setdefault(key, [0, 0.0, ks[0].name[:50]])
            a[0] += 1; a[1] += t
        print("== %s: %d operator groups, %d launches, %.1f us" % (name, len(agg), sum(a[0] for a in agg.values()), sum(a[1] for a in agg.values())))
        for (op, shp, site), (n, t, kn) in sorted(agg.items(), key=lambda kv: kv[0][2]):
            print("  %-26s x%-2d %6.1f us  %-90s | %s | %s" % (op, n, t, shp, site, kn))


if __name__ == "__main__":
    main()
