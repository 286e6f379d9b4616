"""Headline step (one stream) with upp_layers.KPARTS on / off on the same box, alternating: ms per step.   python tools/micro/step_kparts.py
(round 3: 7.145 / 7.153 ms off, 7.207 / 7.220 ms on)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import time  # noqa: E402

import torch  # noqa: E402

import bench  # noqa: E402
from models import upp_layers  # noqa: E402


def run(kparts, pipeline, steps=100):
    upp_layers.KPARTS = kparts
    tr = bench.Trainer(torch.device("cuda", 0), 32, False, use_graph=True, pipeline=pipeline)
    for _ in range(10):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps


if __name__ == "__main__":
    for kp in (False, True, False, True):
        print("KPARTS %-5s %.3f ms" % (kp, run(kp, False)), flush=True)
