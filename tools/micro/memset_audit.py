#!/usr/bin/env python3
"""Which memset nodes do the captured steps contain?  Run under the shim:
   LD_PRELOAD=tools/micro/bin/memset_shim.so UPP_MEMSET_LOG=/tmp/memsets_<recipe>.txt python tools/micro/memset_audit.py <recipe>
(cls | cls_aux | stage2 | pretask | pretrain | seg; B = 32 as in bench.py, and the segmentation recipe also at B = 4)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch  # noqa: E402

import bench  # noqa: E402

kind = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda", 0)
pipeline = kind in ("cls", "cls_aux", "seg")
tr = bench.Trainer(dev, batch, False, pipeline=True) if kind == "cls" else bench.RecipeTrainer(kind, dev, batch, pipeline=pipeline)
for _ in range(3):
    tr.step()
torch.cuda.synchronize()
print("captured", kind, "B", batch)
