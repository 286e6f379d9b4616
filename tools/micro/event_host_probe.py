"""Host time of the calls of one iteration of the pipelined driver's fork (event record on the busy main stream, wait on the side stream) and of
the graph launches: does the wait block the host until the GPU reaches the record?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
import bench

dev = torch.device("cuda", 0)
tr = bench.Trainer(dev, 32, False, use_graph=True, pipeline=True)
for _ in range(4):
    tr.step()
torch.cuda.synchronize()
ts = tr.ts
s1 = torch.cuda.Stream()
cur = torch.cuda.current_stream()
e1 = torch.cuda.Event()
for kind in ("plain", "fork"):
    acc = {"record": 0.0, "wait": 0.0, "back": 0.0, "opt": 0.0}
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    n = 40
    for _ in range(n):
        if kind == "fork":
            t = time.perf_counter(); e1.record(cur); acc["record"] += time.perf_counter() - t
            t = time.perf_counter(); s1.wait_event(e1); acc["wait"] += time.perf_counter() - t
        t = time.perf_counter(); ts._g_back[0].replay(); acc["back"] += time.perf_counter() - t
        t = time.perf_counter(); ts._g_opt.replay(); acc["opt"] += time.perf_counter() - t
    t_enq = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_all
    print(kind, "host enqueue %.3f ms / iteration, wall %.3f ms / iteration;" % (t_enq / n * 1e3, t_tot / n * 1e3),
          "  ".join("%s %.3f ms" % (k, v / n * 1e3) for k, v in acc.items()), flush=True)
