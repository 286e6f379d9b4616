"""A few launches of the tall split-bf16 Linear problems (the segmentation head, the patch embedding) for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
from upp_hip import ops
dev = torch.device('cuda')
for (M, N, K) in ((65536, 1536, 1024), (65536, 1024, 1536), (65536, 1024, 512), (65536, 512, 1024)):
    a = torch.randn(M, K, device=dev)
    w = (torch.randn(N, K, device=dev) * 0.05).requires_grad_(False)
    w._upp_persistent = True
    for _ in range(3):
        c = ops.linear_f32(a, w, frozen=True)
    torch.cuda.synchronize()
print("done")
