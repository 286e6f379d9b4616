#!/usr/bin/env python3
"""Interleaved same-box A/B of a boolean module switch of upp_hip.functional on the bench's own captured step:
     python tools/micro/flag_ab.py ADAPTER_FACTORS [--workloads cls,seg] [--steps 60] [--reps 3]
(ms per step, median of three timings per run; off / on alternate `reps` times; pipelined and one stream)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
import upp_hip.functional as HF  # noqa: E402


def run(flag, value, workload, steps, pipeline):
    setattr(HF, flag, value)
    dev = torch.device("cuda", 0)
    tr = bench.Trainer(dev, 32, False, pipeline=pipeline) if workload == "cls" else bench.RecipeTrainer(workload, dev, 32, pipeline=pipeline)
    for _ in range(8):
        tr.step()
    out = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    del tr
    torch.cuda.empty_cache()
    return sorted(out)[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("flag")
    ap.add_argument("--workloads", default="cls")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    if not isinstance(getattr(HF, a.flag, None), bool):
        raise SystemExit("upp_hip.functional has no boolean switch %r" % a.flag)
    for w in a.workloads.split(","):
        for pipeline in (True, False):
            if pipeline and w not in ("cls", "cls_aux", "seg"):
                continue
            for _ in range(a.reps):
                off = run(a.flag, False, w, a.steps, pipeline)
                on = run(a.flag, True, w, a.steps, pipeline)
                print("%-8s %s  %s = False %.3f ms   True %.3f ms   (%+.2f %%)" % (w, "pipelined " if pipeline else "one stream", a.flag, off, on,
                                                                                100.0 * (on - off) / off), flush=True)


if __name__ == "__main__":
    main()
