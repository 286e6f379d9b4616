"""upp_interp_bwd at the shapes of the recipes (part-segmentation head; the rectify prompter's layers in stage 2 / pre-task).
   python tools/micro/time_interp_bwd.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
from bench import time_kernel
from upp_hip import ops
from models import upp_layers

for B, N, S, C, k in ((32, 2048, 128, 1536, 3), (32, 2048, 128, 1152, 3), (32, 1096, 32, 32, 16), (32, 1096, 32, 64, 16), (32, 32, 16, 32, 6)):
    g = torch.Generator(device='cuda').manual_seed(0)
    xyz1 = torch.rand(B, N, 3, device='cuda', generator=g); xyz2 = torch.rand(B, S, 3, device='cuda', generator=g)
    d, i = upp_layers.square_distance(xyz1, xyz2).sort(dim=-1)
    d, i = d[:, :, :16].contiguous(), i[:, :, :16].contiguous()
    gy = torch.randn(B, N, C, device='cuda', generator=g)
    t = time_kernel(lambda: ops.interp_bwd(d, i, gy, S, k, 1e-4))
    print("B %d N %d S %d C %d k %d: %.1f us  (%.2f TB/s of g)" % (B, N, S, C, k, t * 1e3, B * N * C * 4 / t / 1e9))
