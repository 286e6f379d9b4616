"""Probe: frozen prompting front-end of the next batch on a second stream, concurrent with the trainable back-end."""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'iccv2025-upp_amd'); sys.path.insert(0, 'oracle')
import bench, _seeded
from upp_hip import gemm_tuning
gemm_tuning.enable()
dev = torch.device('cuda', 0)
model = bench.build_model(dev).train()
raw = _seeded.noisy_clouds(32, 1024, seed=0).to(dev)
labels = torch.randint(0, 40, (32,), device=dev)
prompted = torch.zeros(32, 1024, 3, device=dev)
params = [p for p in model.parameters() if p.requires_grad]
def front():
    with torch.no_grad():
        prompted.copy_(model.prompt_points(raw, True, True, 1024))
def back():
    for p in params: p.grad = None
    loss, _ = model.get_loss_acc(model(prompted, completion_prompt=False, denoise=False, point_num=1024), labels)
    loss.backward()
s_front, s_back = torch.cuda.Stream(), torch.cuda.Stream()
for s, fn in ((s_front, front), (s_back, back)):
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize(); print('warm', flush=True)
gf, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(gf, stream=s_front): front()
with torch.cuda.graph(gb, stream=s_back): back()
torch.cuda.synchronize(); print('captured', flush=True)
def t(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def seq():
    gf.replay(); gb.replay()
def par():
    with torch.cuda.stream(s_front): gf.replay()
    with torch.cuda.stream(s_back): gb.replay()
print('front alone %.2f ms, back alone %.2f ms' % (t(gf.replay), t(gb.replay)), flush=True)
print('sequential  %.2f ms' % t(seq), flush=True)
print('two streams %.2f ms' % t(par), flush=True)
