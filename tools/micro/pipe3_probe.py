"""Timing probe (no data flow between the pieces: same work, static inputs): would a THREE-stage pipeline -- rectify prompter of batch
k+2 || completion prompter + grouping / patch embedding of batch k+1 || back-end of batch k, on three streams -- beat the two-stage one
(whole front-end of batch k+1 || back-end of batch k)?  NOTEBOOK 11.1 inferred it would not; this measures it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
import bench
from models import upp_layers as L

dev = torch.device("cuda", 0)
tr = bench.Trainer(dev, 32, False, use_graph=True, pipeline=True)
for _ in range(4):
    tr.step()
torch.cuda.synchronize()
ts, m = tr.ts, tr.model
pts = tr.batches[0][0].clone()
pn = ts.point_num


def f1():
    with torch.no_grad(), L.use_rng(ts._bank_front):
        L.begin_forward(dev, m.training)
        try:
            return m._rectify(pts, pn)
        finally:
            L.end_forward()


def f2(x):
    with torch.no_grad(), L.use_rng(ts._bank_front):
        L.begin_forward(dev, m.training)
        try:
            return m._front_state(m._complete(x, pn))
        finally:
            L.end_forward()


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s1):
    mid = f1().clone()
    out = f2(mid)
torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph(); g1.register_generator_state(ts._gen_front)
with torch.cuda.graph(g1, stream=s1):
    f1()
g2 = torch.cuda.CUDAGraph(); g2.register_generator_state(ts._gen_front)
with torch.cuda.graph(g2, stream=s2):
    f2(mid)
gboth = torch.cuda.CUDAGraph(); gboth.register_generator_state(ts._gen_front)
with torch.cuda.graph(gboth, stream=s1):
    f2(f1())
torch.cuda.synchronize()
cur = torch.cuda.current_stream()


def run(kind, n=40):
    def one():
        if kind == "two":
            s1.wait_stream(cur)
            with torch.cuda.stream(s1):
                gboth.replay()
            ts._g_back[0].replay()
            cur.wait_stream(s1)
        elif kind == "three":
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                g1.replay()
            with torch.cuda.stream(s2):
                g2.replay()
            ts._g_back[0].replay()
            cur.wait_stream(s1); cur.wait_stream(s2)
        elif kind == "back":
            ts._g_back[0].replay()
        elif kind == "front":
            gboth.replay()
        elif kind == "f1":
            g1.replay()
        elif kind == "f2":
            g2.replay()
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
    print("  ".join("%s %.3f ms" % (k, run(k)) for k in ("back", "front", "f1", "f2", "two", "three")), flush=True)
