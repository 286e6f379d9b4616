"""Does a co-running GEMM stream change the result of the FPS kernel?  FPS (B clouds, N -> M) repeatedly on one stream while a second stream
runs Linear kernels back to back; every FPS result is compared with the result computed on an idle GPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd"), os.path.join(ROOT, "tests")]
import torch
import _seeded
from upp_hip import ops, _abi
dev = torch.device("cuda", 0)
B, N, M = 2, 1936, 1536
x = _seeded.noisy_clouds(B, N, seed=5).to(dev).contiguous()
ref = ops.fps(x, M)[0] if isinstance(ops.fps(x, M), tuple) else ops.fps(x, M)
torch.cuda.synchronize()
s2 = torch.cuda.Stream()
def load(kind, Mr, Nc, K):
    a = torch.randn(Mr, K, device=dev); w = torch.randn(Nc, K, device=dev) * K ** -0.5; w._upp_persistent = True
    out = torch.empty(Mr, Nc, device=dev)
    ops.linear_f32(a, w, out=out, frozen=(kind == 'sb'))
    return lambda: ops.linear_f32(a, w, out=out, frozen=(kind == 'sb'))
WAVES = int(sys.argv[1]) if len(sys.argv) > 1 else 0
NN = int(sys.argv[2]) if len(sys.argv) > 2 else N
if NN != N:
    N = NN; M = min(M, N)
    x = _seeded.noisy_clouds(B, N, seed=5).to(dev).contiguous()
ref = ops.fps(x, M, waves=WAVES)
torch.cuda.synchronize()
print("waves", WAVES, "N", N, "M", M)
for kind, shape in (('sb', (4096, 384, 1536)), ('sb', (4096, 384, 1152))):
    bad = 0
    fn = load(kind, *shape) if shape else None
    for it in range(20):
        if fn:
            with torch.cuda.stream(s2):
                for _ in range(60):
                    fn()
        r = ops.fps(x, M, waves=WAVES)
        r = r[0] if isinstance(r, tuple) else r
        torch.cuda.synchronize()
        if not torch.equal(r, ref):
            bad += 1
            d = (r != ref).nonzero()
            first = d[0].tolist()
    print("co-runner %-4s %-20s: %d of 20 FPS results differ%s" % (kind, shape, bad, (" (first at %s)" % first) if bad else ""), flush=True)
