"""One weight-gradient launch per kernel on one shape (for rocprofv3 --pmc / --kernel-trace):  python tools/micro/wgrad_one.py [M N K]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-upp_amd"))
import torch
from upp_hip import ops
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (65536, 1536, 1024)
g, x = torch.randn(M, N, device='cuda'), torch.randn(M, K, device='cuda')
for split in (False, True):
    ops.WGRAD_SPLIT_BF16 = split
    for _ in range(3):
        ops.linear_wgrad_grouped([(g, x)])
torch.cuda.synchronize()
