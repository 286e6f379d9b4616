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
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True, with_stack=stack) as prof:
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
            mine = [f for f in (e.stack or []) if "iccv2025-upp_amd" in f or "bench.py" in f]
            where = mine[0].replace(ROOT + "/", "").replace("iccv2025-upp_amd/", "")[:60] if mine else "?"
        key = (e.name, str(e.input_shapes)[:100], (where + " " if stack else "") + ks[0].name[:46])
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1; a[1] += sum(k.duration for k in ks)
    print("== %s: %d launches, %.1f us" % (kind, sum(a[0] for a in agg.values()), sum(a[1] for a in agg.values())))
    for (op, shp, kn), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-26s x%-3d %7.1f us  %-100s | %s" % (op, n, t, shp, kn))


if __name__ == "__main__":
    main()
