import sys, time, torch
sys.path.insert(0, 'iccv2025-upp_amd')
from upp_hip import _abi, ops
lib = _abi.load()
def bench(N, M, W, B=32, it=20):
    x = torch.rand(B, N, 3, device='cuda') - 0.5
    for _ in range(3): ops.fps(x, M, waves=W)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): ops.fps(x, M, waves=W)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for N, M in ((1228, 1024), (1024, 256), (1024, 64), (1096, 64), (1024, 128)):
    print(N, M, ' '.join('W=%d: %.1f us' % (W, bench(N, M, W)) for W in (0, 1, 2, 4, 8)))
