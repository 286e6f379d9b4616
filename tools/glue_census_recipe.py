"""tools/glue_census.py for a secondary recipe: every device kernel of ONE eager forward + backward of `--workload <w>` that is not
ours, with the aten operator and input shapes that launched it (torch profiler).   python tools/glue_census_recipe.py pretrain [--stack]
--stack adds the innermost frame of this package that issued the operator (file:line)."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "pretrain"
    stack = "--stack" in sys.argv
    tr = bench.RecipeTrainer(kind, torch.device("cuda", 0), 32, use_graph=False)
    ts = tr.ts
    for _ in range(2):
        ts._forward_backward()
    torch.cuda.synchronize()
    kw = {"experimental_config": torch._C._profiler._ExperimentalConfig(verbose=True)} if stack else {}      # (without it e.stack stays empty)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True, with_stack=stack, **kw) as prof:
        ts._forward_backward()
        torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for e in prof.events():
        if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"):
            continue
        ks = [k for k in e.kernels if not ("anonymous namespace" in k.name and "at::native" not in k.name)]
        if not ks or any(c.kernels for c in e.cpu_children if c.name.startswith("aten::")):
            continue
        where = ""
        if stack:
            # (frames come as paths relative to their sys.path entry: "models/upp_layers.py(1093): select"; innermost first)
            mine = [f for f in (e.stack or []) if f.startswith(("models/", "upp_hip/", "utils/", "extensions/", "knn_cuda/", "pointnet2_ops/", "emd/", "bench.py"))]
            where = " < ".join(f.split(": ")[0].replace("models/", "").replace("upp_hip/", "") + ":" + f.split(": ")[-1] for f in mine[:3])[:90] if mine else "?"
            if not mine:                          # a backward op: name the autograd node that issued it
                p_ = e.cpu_parent
                while p_ is not None and "Backward" not in p_.name:
                    p_ = p_.cpu_parent
                where = p_.name.replace("autograd::engine::evaluate_function: ", "")[:60] if p_ is not None else "?"
        key = (e.name, str(e.input_shapes)[:60 if stack else 100], (where + " | " if stack else "") + ks[0].name.replace("void at::native::", "")[:30 if stack else 46])
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1; a[1] += sum(k.duration for k in ks)
    print("== %s: %d launches, %.1f us" % (kind, sum(a[0] for a in agg.values()), sum(a[1] for a in agg.values())))
    for (op, shp, kn), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(("  %-26s x%-3d %7.1f us  %-60s | %s" if stack else "  %-26s x%-3d %7.1f us  %-100s | %s") % (op, n, t, shp, kn))


if __name__ == "__main__":
    main()
