"""The fitted tile model of csrc/linear_sb.hip on problems OUTSIDE its sweep: the Transformer-block Linear layers at batch sizes 24 and 48
(the sweep of profiles/r05_sb_sweep.json holds B = 8, 16, 32, 64).  Every COMPILED tile shape is timed on every problem (chains of 12
launches with their own operands, one HIP graph: the method of tools/micro/sb_sweep.py) and the model's own choice (option SB_TUNED = 0)
is compared with the measured best.   python tools/micro/sb_model_check.py [B ...]   -> profiles/r06_sb_model_b24_b48.txt"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
from upp_hip import ops, _abi

dev = torch.device('cuda')
N_CHAIN = 12
CSRC = os.path.join(ROOT, "iccv2025-upp_amd", "upp_hip", "csrc")


def compiled():
    out = []
    for f, macro in (("linear_sb.hip", "UPP_SB_CONFIGS"), ("linear_sb_tuned.h", "UPP_SB_TUNED_CONFIGS")):
        m = re.search(r'#define %s\(X\) (.*)\n' % macro, open(os.path.join(CSRC, f)).read())
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


batches = [int(a) for a in sys.argv[1:]] or [24, 48]
shapes = []
for B in batches:
    for L in (75, 65, 35, 64):
        for N, K in ((1152, 384), (384, 384), (1536, 384), (384, 1536), (384, 1152)):
            shapes.append((B * L, N, K))
regs, tot_pick, tot_best, ALL = [], 0.0, 0.0, []
with ops.option("SB_TUNED", 0):
    for M, N, K in shapes:
        chain_ops = []
        for _ in range(N_CHAIN):
            a = torch.randn(M, K, device=dev)
            w = (torch.randn(N, K, device=dev) * 0.05)
            chain_ops.append((a, ops.PLANES._split(w), torch.empty(M, N, device=dev)))
        res = {}
        for t in compiled():
            ks, nst = (t >> 4) & 15, t & 15
            if K % (32 * ks) or K // (32 * ks) < nst:
                continue

            def chain():
                for a, planes, c in chain_ops:
                    ops._call(dev, "upp_linear_sb_f32", _abi.ptr(a), a.stride(0), _abi.ptr(planes), None, _abi.ptr(c), N, None, N, M, N, K, 0, t)
            res[t] = timed(chain) / N_CHAIN
        pick = ops.linear_sb_tile(M, N, K)
        best = min(res, key=res.get)
        reg = res[pick] / res[best] - 1.0
        regs.append(reg); tot_pick += res[pick]; tot_best += res[best]
        ALL.append({"M": M, "N": N, "K": K, "shipped": pick, "us": {"%x" % t: us for t, us in res.items()}})
        print("%5d x %4d x %4d  model %x %.2f us | best %x %.2f us | regret %.1f %%" % (M, N, K, pick, res[pick], best, res[best], 100 * reg), flush=True)
n = len(regs)
print("model-only choice on %d problems outside the sweep (B = %s): within 3 %%: %d, within 5 %%: %d, worst %.1f %%, summed time +%.2f %% over the best"
      % (n, ", ".join(str(b) for b in batches), sum(r <= 0.03 for r in regs), sum(r <= 0.05 for r in regs), 100 * max(regs), 100 * (tot_pick / tot_best - 1)))
import json
os.makedirs(os.path.join(ROOT, "gpurun_out", "r06"), exist_ok=True)
json.dump(ALL, open(os.path.join(ROOT, "gpurun_out", "r06", os.environ.get("SB_CHECK_OUT", "sb_check.json")), "w"))
