"""Timing probe: the captured back-end graph of the headline step beside a side stream that runs only tall split-bf16 GEMMs like the patch
embedding's (65,536 rows; per iteration 3 x [512 -> 512, 512 -> 384]), whole or cut into row chunks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
import bench
from upp_hip import ops

dev = torch.device("cuda", 0)
tr = bench.Trainer(dev, 32, False, use_graph=True, pipeline=True)
for _ in range(4):
    tr.step()
torch.cuda.synchronize()
ts = tr.ts
chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 1
M = 65536
a = torch.randn(M, 512, device=dev)
ws = [(torch.randn(n, 512, device=dev) * 0.05).requires_grad_(False) for n in (512, 384)]
for w in ws:
    w._upp_persistent = True
outs = [torch.empty(M, w.shape[0], device=dev) for w in ws]
s1 = torch.cuda.Stream()


def side():
    for _ in range(3):
        for w, o in zip(ws, outs):
            for c in range(chunks):
                lo, hi = c * (M // chunks), (c + 1) * (M // chunks)
                ops.linear_f32(a[lo:hi], w, frozen=True, out=o[lo:hi])


with torch.cuda.stream(s1):
    side()
torch.cuda.synchronize()
gs = torch.cuda.CUDAGraph()
with torch.cuda.graph(gs, stream=s1):
    side()
torch.cuda.synchronize()
cur = torch.cuda.current_stream()
ev = torch.cuda.Event()


def run(kind, n=40):
    def one():
        if kind == "side":
            gs.replay()
        if kind == "both":
            with torch.cuda.stream(s1):
                gs.replay(); ev.record(s1)
            cur.wait_event(ev)
        if kind != "side":
            ts._g_back[0].replay()
        ts._g_opt.replay()
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(2):
    print("chunks %d:" % chunks, "  ".join("%s %.3f ms" % (k, run(k)) for k in ("back", "side", "both")), flush=True)
