"""Timing probe: the pipelined step as ONE HIP graph per parity -- front-end(k+1) and back-end(k) as two branches of a single capture, the
optimizer behind their join -- against the shipped driver (three graph launches and a cross-stream event fork / join per step)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
import bench

dev = torch.device("cuda", 0)
tr = bench.Trainer(dev, 32, False, use_graph=True, pipeline=True)
for _ in range(6):
    tr.step()
torch.cuda.synchronize()
ts = tr.ts
side = torch.cuda.Stream()
graphs = []
for p in range(2):
    g = torch.cuda.CUDAGraph()
    g.register_generator_state(ts._gen_front)
    with torch.cuda.graph(g):
        cap = torch.cuda.current_stream()
        side.wait_stream(cap)
        with torch.cuda.stream(side):
            ts._front(p)
        ts._back(1 - p)
        cap.wait_stream(side)
        ts._tail()
    graphs.append(g)
torch.cuda.synchronize()


def run(kind, n=60):
    k = [0]
    def one():
        if kind == "shipped":
            tr.step()
        else:
            graphs[k[0] & 1].replay(); k[0] += 1
    for _ in range(6):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(3):
    print("  ".join("%s %.3f ms" % (k, run(k)) for k in ("shipped", "one_graph")), flush=True)
