"""Probe: 2-stage (front || back) against 3-stage (rectify || complete+embed || back) software pipelining."""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'iccv2025-upp_amd'); sys.path.insert(0, 'oracle')
import bench
from utils import synthetic as _seeded
from models import upp_layers as L
dev = torch.device('cuda', 0)
model = bench.build_model(dev).train()
raw = _seeded.noisy_clouds(32, 1024, seed=0).to(dev)
labels = torch.randint(0, 40, (32,), device=dev)
rect = torch.zeros(32, 972, 3, device=dev)
params = [p for p in model.parameters() if p.requires_grad]
with torch.no_grad():
    probe = model.eval().prompt_tokens(raw, True, True, 1024); model.train()
state = [torch.zeros_like(t) for t in probe]
def f1():
    with torch.no_grad():
        L.begin_forward(dev, True)
        try: rect.copy_(model._rectify(raw, 1024))
        finally: L.end_forward()
def f2():
    with torch.no_grad():
        L.begin_forward(dev, True)
        try: st = model._front_state(model._complete(rect, 1024))
        finally: L.end_forward()
        torch._foreach_copy_(state, list(st))
def f12():
    f1(); f2()
def back():
    for p in params: p.grad = None
    loss, _ = model.get_loss_acc(model.forward_tokens(*state), labels)
    loss.backward()
import os
lo, hi = (0, -1) if os.environ.get('PRIO') else (0, 0)
streams = [torch.cuda.Stream(priority=lo), torch.cuda.Stream(priority=lo), torch.cuda.Stream(priority=hi)]
print('priorities', [s.priority for s in streams], flush=True)
for s, fn in ((streams[0], f12), (streams[2], back)):
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize(); print('warm', flush=True)
g1, g2, g12, gb = (torch.cuda.CUDAGraph() for _ in range(4))
with torch.cuda.graph(g1, stream=streams[0]): f1()
with torch.cuda.graph(g2, stream=streams[1]): f2()
with torch.cuda.graph(g12, stream=streams[0]): f12()
with torch.cuda.graph(gb, stream=streams[2]): back()
torch.cuda.synchronize(); print('captured', flush=True)
def t(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def two():
    with torch.cuda.stream(streams[0]): g12.replay()
    with torch.cuda.stream(streams[2]): gb.replay()
def three():
    with torch.cuda.stream(streams[0]): g1.replay()
    with torch.cuda.stream(streams[1]): g2.replay()
    with torch.cuda.stream(streams[2]): gb.replay()
print('alone: rectify %.2f, complete+embed %.2f, back %.2f ms' % (t(g1.replay), t(g2.replay), t(gb.replay)), flush=True)
print('2 streams %.2f ms' % t(two), flush=True)
print('3 streams %.2f ms' % t(three), flush=True)
print('2 streams %.2f ms' % t(two), flush=True)
