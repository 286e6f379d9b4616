"""Stand-alone timing of the patch-embed chain (upp_patch_embed_fwd) at the headline shape: python tools/time_patch_embed.py"""
import sys
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'iccv2025-upp_amd'); sys.path.insert(0, 'oracle')
import bench
from utils import synthetic as _seeded
from models.upp_layers import Encoder
from upp_hip import ops
dev = torch.device('cuda', 0)
B = 32
for G, n in ((64, 32), (32, 16)):
    x = _seeded.unit_ball_clouds(B, 1024, seed=7).to(dev)
    _, cen = ops.fps(x, G, want_centers=True)
    _, _, nb = ops.knn(x, cen, n, want_dist=False, want_neigh=True)
    enc = Encoder(384).to(dev).train()
    for p in enc.parameters():
        p.requires_grad_(False)
    for _ in range(3):          # the first measurement of a process runs ~8 % slow (clock ramp): keep the last
        with torch.no_grad():
            t = bench.time_kernel(lambda: enc(nb), iters=5)
    R = B * G * n
    flops = 2.0 * R * (128 * 256 + 256 * 512 + 512 * 384) + 2.0 * (R / n) * 256 * 512 + 2.0 * R * 3 * 128
    print('G=%d n=%d R=%d: %.1f us, %.1f TFLOP/s' % (G, n, R, t * 1e3, flops / t / 1e9))
