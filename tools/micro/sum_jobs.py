"""The batched_sum jobs of one step (partial matrices per destination, bytes read):  python tools/micro/sum_jobs.py [cls|seg|pretrain|...]"""
import os, sys, collections
ROOT='/root/repo'
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch, bench
from upp_hip import ops
kind = sys.argv[1] if len(sys.argv) > 1 else "pretrain"
dev = torch.device("cuda", 0)
ts = (bench.Trainer(dev, 32, False, use_graph=False) if kind == "cls" else bench.RecipeTrainer(kind, dev, 32, use_graph=False)).ts
for _ in range(2): ts._forward_backward()
orig = ops.batched_sum
log = []
def spy(jobs, *a, **k):
    log.append([(int(j[2]), int(j[3]), bool(j[6]), j[5].data_ptr()) for j in jobs])
    return orig(jobs, *a, **k)
ops.batched_sum = spy
import upp_hip.functional as HF
HF.ops.batched_sum = spy
ts._forward_backward()
torch.cuda.synchronize()
for call in log:
    agg = collections.Counter((n, l) for n, l, a, d in call)
    dsts = collections.Counter(d for n, l, a, d in call)
    print("call: %d jobs, rows x len:" % len(call), sorted(agg.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[1])[:8], "| max jobs per dst", max(dsts.values()),
          "| MB read %.1f" % (sum(n * l for n, l, a, d in call) * 4 / 1e6))
