"""Probe: which half of the pipelined step is the critical path, and does a CU-masked front-end stream help?
    python tools/cumask_probe.py
Front-end graph (model.prompt_tokens) on stream F, back-end graph (forward_tokens + loss + backward) on stream B.
Prints the stand-alone time of each, then for the concurrent launch the finish time of each half (HIP events) --
for an ordinary F and for F created with hipExtStreamCreateWithCUMask under several masks."""
import ctypes
import sys
import time

import torch

sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'iccv2025-upp_amd'); sys.path.insert(0, 'oracle')
import bench
from utils import synthetic as _seeded
dev = torch.device('cuda', 0)
model = bench.build_model(dev).train()
raw = _seeded.noisy_clouds(32, 1024, seed=0).to(dev)
labels = torch.randint(0, 40, (32,), device=dev)
params = [p for p in model.parameters() if p.requires_grad]
with torch.no_grad():
    probe = model.eval().prompt_tokens(raw, completion_prompt=False, denoise=False, point_num=1024)
    model.train()
state = [torch.zeros_like(t) for t in probe]


def front():
    with torch.no_grad():
        st = model.prompt_tokens(raw, completion_prompt=True, denoise=True, point_num=1024)
        torch._foreach_copy_(state, list(st))


def back():
    for p in params:
        p.grad = None
    loss, _ = model.get_loss_acc(model.forward_tokens(*state), labels)
    loss.backward()


hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(len(words)), arr)
    if rc != 0:
        raise RuntimeError('hipExtStreamCreateWithCUMask -> %d' % rc)
    return torch.cuda.ExternalStream(h.value, device=dev)


def bits(pred):
    words = [0] * 8
    for i in range(256):
        if pred(i):
            words[i // 32] |= 1 << (i % 32)
    return words


sB = torch.cuda.Stream()
s0 = torch.cuda.Stream()
for s, fn in ((s0, front), (sB, back)):
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize(); print('warm', flush=True)
gb = torch.cuda.CUDAGraph()
with torch.cuda.graph(gb, stream=sB):
    back()
torch.cuda.synchronize()


def t(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


def run(name, sF):
    gf = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gf, stream=sF):
        front()
    torch.cuda.synchronize()

    def alone_f():
        with torch.cuda.stream(sF):
            gf.replay()

    def alone_b():
        with torch.cuda.stream(sB):
            gb.replay()

    ta, tb = t(alone_f), t(alone_b)
    cur = torch.cuda.current_stream()
    ef, eb, e0 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    fin_f = fin_b = tot = 0.0
    n = 20
    for it in range(n + 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record(cur)
        sF.wait_stream(cur); sB.wait_stream(cur)
        with torch.cuda.stream(sF):
            gf.replay(); ef.record(sF)
        with torch.cuda.stream(sB):
            gb.replay(); eb.record(sB)
        cur.wait_stream(sF); cur.wait_stream(sB)
        torch.cuda.synchronize()
        if it >= 3:
            tot += (time.perf_counter() - t0) * 1e3
            fin_f += e0.elapsed_time(ef); fin_b += e0.elapsed_time(eb)
    print('%-28s alone: front %.2f back %.2f | together: front done %.2f, back done %.2f, wall %.2f ms'
          % (name, ta, tb, fin_f / n, fin_b / n, tot / n), flush=True)


run('plain stream', torch.cuda.Stream())
for name, pred in (('low 128 bits', lambda i: i < 128), ('even bits', lambda i: i % 2 == 0),
                   ('i%8<4', lambda i: i % 8 < 4), ('low 192 bits', lambda i: i < 192), ('low 64 bits', lambda i: i < 64),
                   ('i%4==0 (64 CUs)', lambda i: i % 4 == 0), ('all 256', lambda i: True)):
    try:
        run(name, masked_stream(bits(pred)))
    except Exception as e:  # noqa: BLE001
        print(name, 'FAILED', repr(e)[:200], flush=True)
