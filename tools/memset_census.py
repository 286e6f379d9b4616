"""Memset operations (hipMemsetAsync: memset NODES once the step is captured) of one eager forward + backward of a recipe, with the aten
operator, its input shapes and the innermost frames of this package that issued it.  On this stack a captured memset node works in
the first replay of a graph and writes garbage from the second on (NOTEBOOK 12.11): a captured step must contain none.
   python tools/memset_census.py <cls|cls_aux|stage2|pretask|pretrain|seg> [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch  # noqa: E402

import bench  # noqa: E402


def memsets_of(fn):
    """-> list of (operator, input shapes, frames) for every memset issued while fn() runs (torch profiler)."""
    kw = {"experimental_config": torch._C._profiler._ExperimentalConfig(verbose=True)}
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True, with_stack=True, **kw) as prof:
        fn()
        torch.cuda.synchronize()
    out = []
    for e in prof.events():
        if "memset" not in e.name.lower() or e.device_type != torch.autograd.DeviceType.CPU:
            continue
        p_ = e.cpu_parent
        op = None
        while p_ is not None:
            if op is None and (p_.name.startswith("aten::") or "Backward" in p_.name):
                op = p_
            if p_.stack:
                break
            p_ = p_.cpu_parent
        frames = [f for f in ((p_.stack if p_ is not None else None) or []) if f.startswith(("models/", "upp_hip/", "utils/", "extensions/", "bench.py"))]
        out.append((op.name if op is not None else "(no operator: a direct runtime call)", str(op.input_shapes)[:70] if op is not None else "",
                    " < ".join(f.split(": ")[0] + ":" + f.split(": ")[-1] for f in frames[:3])))
    return out


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "pretask"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device("cuda", 0)
    tr = bench.Trainer(dev, batch, False, use_graph=False) if kind == "cls" else bench.RecipeTrainer(kind, dev, batch, use_graph=False)
    for _ in range(2):
        tr.ts._forward_backward()
    torch.cuda.synchronize()
    found = memsets_of(tr.ts._forward_backward)
    print("== %s (B = %d): %d memsets in one forward + backward" % (kind, batch, len(found)))
    for op, shp, where in found:
        print("  %-44s %-70s | %s" % (op[:44], shp, where[:150]))


if __name__ == "__main__":
    main()
