"""Stand-alone time of every tile shape of linear_sb_kernel (a -DUPP_SB_SWEEP build: tools/micro/sb_sweep_gen.py, then
UPP_HIPCC_FLAGS=-DUPP_SB_SWEEP python -m upp_hip.build) on the Linear shapes of the headline step: chains of 12 launches with their own
operands (so no launch finds its inputs in the L2), one HIP graph, per-launch average.  Prints the five fastest shapes per problem and
the shipped choice."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
from upp_hip import ops, _abi

dev = torch.device('cuda')
N_CHAIN = 12
CSRC = os.path.join(ROOT, "iccv2025-upp_amd", "upp_hip", "csrc")


def configs():
    out = []
    for f, macro in (("linear_sb.hip", "UPP_SB_CONFIGS"), ("linear_sb_sweep.h", "UPP_SB_SWEEP_CONFIGS")):
        s = open(os.path.join(CSRC, f)).read()
        m = re.search(r'#define %s\(X\) (.*)\n' % macro, s)
        for t in re.findall(r'X\(([^)]*)\)', m.group(1).replace('UPP_SB_NST44', '3')):
            a, b, c, d, e = (int(v) for v in t.split(','))
            out.append(0x400000 + a * 65536 + b * 4096 + c * 256 + d * 16 + e)
    return out


def timed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            g.replay()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 5 * 1e3)
    return best


SHAPES = [(32, 96, 384), (32, 256, 256), (32, 256, 768), (32, 384, 512), (32, 512, 2432), (32, 768, 256), (32, 2432, 512), (32, 64, 128),
          (1024, 16, 192), (1024, 96, 384), (1024, 128, 384), (1024, 192, 384), (1024, 384, 128),
          (1120, 384, 384), (1120, 384, 1536), (1120, 1152, 384), (1120, 1536, 384),
          (2048, 384, 128), (2048, 384, 384), (2048, 384, 1536), (2048, 1152, 384), (2048, 1536, 384),
          (2080, 384, 384), (2080, 384, 1152), (2080, 384, 1536), (2080, 1152, 384), (2080, 1536, 384),
          (2400, 384, 384), (2400, 384, 1152), (2400, 384, 1536), (2400, 1152, 384), (2400, 1536, 384),
          (4096, 384, 128), (4096, 384, 384), (4096, 384, 1152), (4096, 384, 1536), (4096, 1152, 384), (4096, 1152, 1536), (4096, 1536, 384), (4096, 1536, 1152),
          (4128, 384, 384), (4128, 384, 1152), (4128, 384, 1536), (4128, 1152, 384), (4128, 1536, 384)]
EPI = []
if len(sys.argv) > 1:                       # MxNxK or MxNxK:e (epilogue code of include/upp_hip.h: 1 bias, 2 bias + GELU, 3 bias + GELU + GELU', 4 multiply)
    SHAPES = []
    for a in sys.argv[1:]:
        dims, _, e = a.partition(':')
        SHAPES.append(tuple(int(v) for v in dims.split('x')))
        EPI.append(int(e or 0))
ALL = []
for si, (M, N, K) in enumerate(SHAPES):
    ops_ = []
    for _ in range(N_CHAIN):
        a = torch.randn(M, K, device=dev)
        w = (torch.randn(N, K, device=dev) * 0.05).requires_grad_(False)
        ops_.append((a, ops.PLANES.get(w), w, torch.empty(M, N, device=dev)))
    epi = EPI[si] if EPI else 0
    bias = torch.randn(N, device=dev) if epi in (1, 2, 3) else None
    auxs = [torch.randn(M, N, device=dev) for _ in range(N_CHAIN)] if epi in (3, 4) else [None] * N_CHAIN
    res = []
    for t in configs():
        bmb, bnb, ks, nst = (t >> 16) & 15, (t >> 12) & 15, (t >> 4) & 15, t & 15
        if K % (32 * ks) or K // (32 * ks) < nst:
            continue
        wgs = -(-M // (32 * bmb)) * -(-N // (32 * bnb))
        if wgs > 1024 and (M < 16384 or bmb * bnb < 8):
            continue

        def chain():
            for (a, planes, w, c), x in zip(ops_, auxs):
                ops._call(dev, "upp_linear_sb_f32", _abi.ptr(a), a.stride(0), _abi.ptr(planes), _abi.ptr(bias), _abi.ptr(c), N, _abi.ptr(x), N, M, N, K, epi, t)
        try:
            res.append((timed(chain) / N_CHAIN, t, wgs))
        except RuntimeError:
            pass
    res.sort()
    ALL.append({"M": M, "N": N, "K": K, "epilogue": epi, "shipped": ops.linear_sb_tile(M, N, K), "us": {"%x" % t: us for us, t, _ in res}})
    shipped = ops.linear_sb_tile(M, N, K)
    ship_t = [r for r in res if r[1] == shipped]
    print("%5d x %4d x %4d e%d  shipped %x %s | " % (M, N, K, epi, shipped, ("%.1f us" % ship_t[0][0]) if ship_t else "-")
          + "  ".join("%x %.1f (%d wg)" % (t, us, wgs) for us, t, wgs in res[:6]), flush=True)
import json
json.dump(ALL, open(os.path.join(ROOT, "gpurun_out", "r05", os.environ.get("SB_SWEEP_OUT", "sb_sweep.json")), "w"))
