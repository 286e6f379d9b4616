"""Timing probe on the captured graphs of the headline step: what each cross-stream edge of the pipelined driver costs.  Per iteration: the
front-end graph on the side stream, the back-end graph + optimizer graph on the main stream, with
  none       no edge at all (the two streams run free: not a valid schedule, the floor)
  join       main waits for the side stream's event (the data hand-over front(k-1) -> back(k-1))
  fork       side waits for a main-stream event (input copied / flow control)
  fork+join  both (the shipped driver)"""
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
s1 = ts.s_front
cur = torch.cuda.current_stream()
ef, em = torch.cuda.Event(), torch.cuda.Event()


def run(kind, n=60):
    def one():
        if "fork" in kind:
            em.record(cur); s1.wait_event(em)
        with torch.cuda.stream(s1):
            ts._g_front[0].replay()
            ef.record(s1)
        if "join" in kind:
            cur.wait_event(ef)
        ts._g_back[1].replay()
        ts._g_opt.replay()
    for _ in range(6):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(3):
    print("  ".join("%s %.3f ms" % (k, run(k)) for k in ("none", "join", "fork", "fork+join")) + "   shipped step %.3f ms" % (lambda: (lambda t0: ([tr.step() for _ in range(60)], torch.cuda.synchronize(), (time.perf_counter() - t0) / 60 * 1e3)[2])(time.perf_counter()))(), flush=True)
