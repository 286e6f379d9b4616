"""Host-side cost of one pipelined step (graph launches + stream edges): time until the Python loop returns against
the time until the device is done.  python tools/host_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "iccv2025-upp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bench
dev = torch.device("cuda", 0)
for pipeline in (True, False):
    tr = bench.Trainer(dev, 32, False, use_graph=True, pipeline=pipeline)
    for _ in range(6):
        tr.step()
    torch.cuda.synchronize()
    for n in (1, 4, 20):
        t0 = time.perf_counter()
        for _ in range(n):
            tr.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("pipeline=%s steps=%d: loop returned after %.2f ms/step, device done after %.2f ms/step" % (pipeline, n, 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n))
    del tr
