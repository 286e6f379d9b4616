#!/usr/bin/env python3
"""Which torch streams run beside which?  (torch hands out streams from a pool of 32; the runtime maps streams onto hardware queues.)
Prints, for the default stream and the first N pool streams, the pairwise "runs beside" matrix of upp_hip.train._runs_beside.
   python tools/micro/stream_queue_map.py [N = 10]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch  # noqa: E402

from upp_hip.train import _runs_beside  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(n)]
names = ["dflt"] + ["p%02d" % i for i in range(n)]
print("      " + " ".join("%4s" % x for x in names))
for i, a in enumerate(streams):
    row = []
    for j, b in enumerate(streams):
        row.append("   ." if i == j else ("   y" if _runs_beside(a, b) else "   -"))
    print("%4s  " % names[i] + " ".join(row), flush=True)
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(unset)"))
