import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-upp_amd"))
import torch
from upp_hip import ops
ops.WGRAD_SPLIT_BF16 = True
for (M, N, K) in [(16, 128, 128), (32, 128, 128), (700, 200, 132), (4096, 128, 128)]:
    gen = torch.Generator().manual_seed(1)
    g = torch.randint(-7, 8, (M, N), generator=gen).float().cuda()
    x = torch.randint(-7, 8, (M, K), generator=gen).float().cuda()
    p = ops.linear_wgrad_grouped([(g, x)])[0]
    dw = p.sum(0)
    want = (g.double().t() @ x.double()).float()
    bad = (dw != want).nonzero()
    print(M, N, K, "splits", p.shape[0], "bad", bad.shape[0], "of", dw.numel())
    if bad.shape[0]:
        print("  rows n:", sorted(set(bad[:, 0].tolist()))[:40])
        print("  cols k:", sorted(set(bad[:, 1].tolist()))[:40])
        for s in range(p.shape[0]):
            r = min((s + 1) * ((M + p.shape[0] - 1) // p.shape[0]), M)
print("one-hot probe")
M, N, K = 16, 128, 128
for m0 in (0, 3, 4, 7, 8, 12, 15):
    g = torch.zeros(M, N).cuda(); x = torch.zeros(M, K).cuda()
    g[m0, :] = torch.arange(N).float().cuda() + 1
    x[m0, :] = 1.0
    dw = ops.linear_wgrad_grouped([(g, x)])[0].sum(0)
    want = g.t() @ x
    print(m0, "ok" if torch.equal(dw, want) else ("BAD", dw[:4, :4].tolist(), dw.abs().sum().item(), want.abs().sum().item()))
