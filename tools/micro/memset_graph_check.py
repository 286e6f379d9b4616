import sys
sys.path[:0] = ["/root/repo", "/root/repo/iccv2025-upp_amd"]
import torch
from upp_hip import ops
torch.manual_seed(0)
B, n, m = 4, 256, 256
x1 = torch.rand(B, n, 3, device='cuda'); x2 = torch.rand(B, m, 3, device='cuda')
match = ops.emd_approxmatch(x1, x2)
eager = ops.emd_matchcost(x1, x2, match).clone()
# the memset path of the Chamfer backward: n + m > 5,461
y1 = torch.rand(2, 2048, 3, device='cuda'); y2 = torch.rand(2, 8192, 3, device='cuda')
d1, d2, i1, i2 = ops.chamfer_fwd(y1, y2)
gd1, gd2 = torch.rand_like(d1), torch.rand_like(d2)
e1, e2 = [t.clone() for t in ops.chamfer_bwd(y1, y2, i1, i2, gd1, gd2)]
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    ops.emd_matchcost(x1, x2, match); ops.chamfer_bwd(y1, y2, i1, i2, gd1, gd2)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    c = ops.emd_matchcost(x1, x2, match)
    g1, g2 = ops.chamfer_bwd(y1, y2, i1, i2, gd1, gd2)
for it in range(4):
    g.replay(); torch.cuda.synchronize()
    print("replay", it, "emd cost / eager:", [round(float(v), 4) for v in (c / eager)], " chamfer g1 max ratio %.4f g2 %.4f" % (
        float((g1.abs().max() / e1.abs().max())), float(g2.abs().max() / e2.abs().max())), "equal", torch.equal(g1, e1), torch.equal(g2, e2))
