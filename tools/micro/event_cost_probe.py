"""What a cross-stream fork / join around a graph replay costs on this runtime (the pipelined step driver does one per step): the captured
back-end graph + optimizer graph per iteration, with event operations added one kind at a time."""
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
e1, e2, e3 = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
e3.record(cur)
tiny = torch.zeros(64, device=dev)


def run(kind, n=40):
    def one():
        if kind == "record":
            e1.record(cur)
        if kind in ("fork", "forkjoin", "fork_tiny_join"):
            e1.record(cur); s1.wait_event(e1)
        if kind == "fork_tiny_join":
            with torch.cuda.stream(s1):
                tiny.add_(1.0)
        if kind == "fork2":                # two waits of the side stream on main-stream events (the shipped driver: input copied, state slot read)
            e1.record(cur); s1.wait_event(e1); s1.wait_event(e3)
        if kind == "fork_old":             # the side stream waits on an OLD main-stream event (recorded an iteration ago)
            s1.wait_event(e3)
        ts._g_back[0].replay()
        if kind in ("fork2", "fork_old"):
            e3.record(cur)
        if kind == "join_idle":
            e2.record(s1); cur.wait_event(e2)
        if kind in ("forkjoin", "fork_tiny_join"):
            e2.record(s1); cur.wait_event(e2)
        if kind == "tiny_main":
            tiny.add_(1.0)
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
    print("  ".join("%s %.3f" % (k, run(k)) for k in ("plain", "fork", "fork2", "fork_old", "join_idle", "forkjoin")), flush=True)
