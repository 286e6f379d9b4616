#!/usr/bin/env python3
"""A long run of a captured step: N replays on the rotating synthetic batches, loss printed every N / 10 steps, finiteness of the loss,
the gradient buffer and every parameter at the end.   python tools/micro/soak.py <cls|seg|cls_aux|stage2|pretask|pretrain> [steps = 2000]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch  # noqa: E402

import bench  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "cls"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = torch.device("cuda", 0)
pipeline = kind in ("cls", "cls_aux", "seg")
tr = bench.Trainer(dev, 32, False, pipeline=True) if kind == "cls" else bench.RecipeTrainer(kind, dev, 32, pipeline=pipeline)
out = []
for k in range(steps):
    loss = tr.step()
    if (k + 1) % max(1, steps // 10) == 0:
        torch.cuda.synchronize()
        out.append(float(loss))
torch.cuda.synchronize()
ts = tr.ts
ok = all(v == v and abs(v) < 1e6 for v in out) and bool(torch.isfinite(ts.flat.flat).all()) and all(bool(torch.isfinite(p).all()) for p in tr.model.parameters())
print("%s: %d steps, loss every %d: %s -> %s" % (kind, steps, max(1, steps // 10), ["%.4f" % v for v in out], "finite" if ok else "NOT FINITE"), flush=True)
sys.exit(0 if ok else 1)
