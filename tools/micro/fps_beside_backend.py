"""Timing probe: the captured back-end graph of the headline step alone, and beside a second stream that runs only the step's FPS launches
(2 x 1228 -> 1024 on 32 clouds: 32 workgroups for ~0.2 ms each).  A back-end GEMM is one round of 228 workgroups on 256 CUs: with 32 CUs held
by FPS, does every GEMM of that time need a second round?"""
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
g = torch.Generator(device=dev).manual_seed(1)
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 32          # clouds (= workgroups) of the FPS launches
clouds = torch.rand(NB, 1228, 3, device=dev, generator=g) * 2 - 1
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8          # FPS launches per step in the probe: ~1.6 ms of FPS residency
s1 = torch.cuda.Stream()
with torch.cuda.stream(s1):
    ops.fps(clouds, 1024)
torch.cuda.synchronize()
gf = torch.cuda.CUDAGraph()
tiny = torch.zeros(64, device=dev)
with torch.cuda.graph(gf, stream=s1):
    if reps == 0:
        tiny.add_(1.0)                    # (reps = 0: one 2-us kernel -- the cost of the fork / join itself)
    for _ in range(reps):
        ops.fps(clouds, 1024)
torch.cuda.synchronize()
cur = torch.cuda.current_stream()
ev = torch.cuda.Event()


def run(kind, n=40):
    def one():
        if kind in ("both",):
            s1.wait_stream(cur)
            with torch.cuda.stream(s1):
                gf.replay()
        if kind == "forkjoin":             # fork and join with nothing on the side stream
            s1.wait_stream(cur)
        if kind in ("indep", "indep_join"):   # the side stream runs its graph without waiting for the main stream ...
            with torch.cuda.stream(s1):
                gf.replay()
                ev.record(s1)
        if kind == "indep_join":           # ... and the main stream waits for it (the direction a data hand-over needs)
            cur.wait_event(ev)
        if kind == "fps":
            gf.replay()
        else:
            ts._g_back[0].replay()
        if kind in ("both", "forkjoin"):
            cur.wait_stream(s1)
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
    print("  ".join("%s %.3f ms" % (k, run(k)) for k in ("back", "fps", "both", "forkjoin", "indep", "indep_join")), flush=True)
