"""upp_linear_f32 / upp_linear_wgrad_f32 against the library at the 65,536-row layers of the segmentation head and the patch embedding."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
import torch.nn.functional as F
from bench import time_kernel
from upp_hip import ops
if "--tuned" in sys.argv:            # the library side of the comparison with TunableOp-selected solutions (measurement only)
    os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"; os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"; os.environ["PYTORCH_TUNABLEOP_VERBOSE"] = "0"
    torch.cuda.tunable.enable(True)
M = 65536
for N, K in ((1024, 1536), (512, 1024), (256, 512), (52, 256), (512, 512), (384, 512), (256, 128)):
    a = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * K ** -0.5; g = torch.randn(M, N, device='cuda')
    wt = w.t().contiguous()
    out = torch.empty(M, N, device='cuda'); dx = torch.empty(M, K, device='cuda')
    fl = 2.0 * M * N * K
    t1 = time_kernel(lambda: ops.linear_f32(a, w, out=out), iters=5); t2 = time_kernel(lambda: F.linear(a, w), iters=5)
    t3 = time_kernel(lambda: ops.linear_f32(g, wt, out=dx), iters=5); t4 = time_kernel(lambda: torch.mm(g, w), iters=5)
    def wg():
        p = ops.linear_wgrad(g, a); return p.sum(0) if p.shape[0] > 1 else p
    t5 = time_kernel(wg, iters=5); t6 = time_kernel(lambda: torch.mm(g.t(), a), iters=5)
    print("N=%4d K=%4d | fwd ours %7.1f us (%5.1f TF) lib %7.1f (%5.1f) | dX ours %7.1f lib %7.1f | dW ours(+sum) %7.1f lib %7.1f" % (
        N, K, t1 * 1e3, fl / t1 / 1e9, t2 * 1e3, fl / t2 / 1e9, t3 * 1e3, t4 * 1e3, t5 * 1e3, t6 * 1e3), flush=True)
