import os, sys
ROOT='/root/repo'
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch, bench
from upp_hip import ops
import upp_hip.functional as HF
for kind in ("stage2", "seg", "pretask"):
    ts = bench.RecipeTrainer(kind, torch.device("cuda", 0), 32, use_graph=False).ts
    ts._forward_backward()
    orig = ops.interp_bwd
    log = []
    def spy(dist, idx, g_out, S, k, eps):
        log.append((tuple(g_out.shape), S, k, tuple(dist.shape)))
        return orig(dist, idx, g_out, S, k, eps)
    ops.interp_bwd = spy; HF.ops.interp_bwd = spy
    ts._forward_backward(); torch.cuda.synchronize()
    ops.interp_bwd = orig; HF.ops.interp_bwd = orig
    print(kind, log)
    del ts
