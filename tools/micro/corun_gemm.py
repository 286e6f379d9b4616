"""Two independent chains of Linear launches on two streams against the same chains on one stream: what co-residency of two workgroups
per CU (half-LDS tiles) buys the pipelined step, whose two streams' GEMMs otherwise take turns (every default tile holds 120-160 KB).
Needs a -DUPP_SB_SWEEP build of the library (tools/micro/sb_sweep_gen.py): the half-LDS shapes are not shipped."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
from upp_hip import ops, _abi

dev = torch.device('cuda')
N_CHAIN = 12


def operands(M, N, K, n):
    out = []
    for _ in range(n):
        a = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        out.append((a, ops.PLANES.get(w.requires_grad_(False)), w, torch.empty(M, N, device=dev)))
    return out


def launch(op, M, N, K, tile):
    a, planes, w, c = op
    ops._call(dev, "upp_linear_sb_f32", _abi.ptr(a), a.stride(0), _abi.ptr(planes), None, _abi.ptr(c), N, None, N, M, N, K, 0, tile)


def timed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 10 * 1e3


side = torch.cuda.Stream()
NARROW = [0x422114, 0x422113, 0x422122, 0x421114, 0x412114, 0x442112, 0x442113, 0x422242, 0x423114, 0x423113, 0x412124, 0x422123]
WIDE = [0x444213, 0x444212, 0x424212, 0x424213, 0x424112, 0x424214, 0x443114, 0x442214, 0x442113, 0x422114, 0x423114]
for (M, N, K, cands) in ((2400, 384, 384, NARROW), (2400, 384, 1536, NARROW), (2400, 384, 1152, NARROW), (1120, 384, 384, NARROW), (1120, 384, 1536, NARROW),
                         (2400, 1536, 384, WIDE), (2400, 1152, 384, WIDE), (1120, 1536, 384, WIDE), (1120, 1152, 384, WIDE)):
    opa, opb = operands(M, N, K, N_CHAIN), operands(M, N, K, N_CHAIN)
    print("default %x" % ops.linear_sb_tile(M, N, K))
    for t in cands:
        def chain_a():
            for o in opa:
                launch(o, M, N, K, t)
        def both_par():
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for o in opb:
                    launch(o, M, N, K, t)
            chain_a()
            cur.wait_stream(side)
        try:
            t_a, t_p = timed(chain_a), timed(both_par)
        except RuntimeError as ex:
            print("M %d N %d K %d tile %x: %s" % (M, N, K, t, str(ex)[-40:]), flush=True)
            continue
        print("M %4d N %4d K %4d  tile %x   alone %5.1f us / launch   two chains side by side %5.1f us / launch" % (M, N, K, t, t_a / N_CHAIN, t_p / (2 * N_CHAIN)), flush=True)
