#!/usr/bin/env python3
"""Probe for the "more, smaller steps side by side" decomposition (verdict r5, missing #3): would the back-end's light kernels (row
LayerNorms, attention, propagation: ~1 ms per step on a fraction of the chip) hide behind GEMMs if the batch ran as SEVERAL
independent part-batches on streams of their own?

  python tools/micro/half_batch_probe.py [--steps 60] [--parts 1,2,4]

For each P in --parts: P independent step drivers (own model copy, own static buffers, own graphs) of 32 / P clouds each, every one
driven from a stream of its own, K rounds of "one step of each"; the figure is ms per 32 clouds.  P = 1 is the product's step.  The
part-batches take their BatchNorm statistics over 32 / P clouds, so this is a TIMING probe, not a product path: a real form would
exchange the partial statistics between the parts at every BatchNorm (26 joins per step).  Both step drivers are timed: pipelined
(front-end(k+1) || back-end(k), two streams per part) and one stream per part.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "iccv2025-upp_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def run(parts, pipeline, steps, warmup, device):
    # every stream of the probe on a hardware queue of its own (torch's pool streams share the runtime's four queues: two parts on one
    # queue would run one after the other whatever the decomposition is worth) -- as far as there are queues: 2 pipelined parts use all four
    import upp_hip.train as T
    chosen = []

    def pick(device_=None, tries=24):
        s = None
        for _ in range(tries):
            s = torch.cuda.Stream(device=device)
            if all(T._runs_beside(c, s) for c in chosen):
                break
        else:
            print("    (no free hardware queue left for stream %d: it shares one)" % len(chosen))
        chosen.append(s)
        return s

    keep = T._concurrent_stream
    T._concurrent_stream = pick
    try:
        streams, trs = [], []
        for _ in range(parts):
            streams.append(pick())
            with torch.cuda.stream(streams[-1]):
                trs.append(bench.Trainer(device, 32 // parts, False, use_graph=True, pipeline=pipeline))
    finally:
        T._concurrent_stream = keep

    def round_():
        for tr, s in zip(trs, streams):
            with torch.cuda.stream(s):
                tr.step()

    for _ in range(warmup + (1 if pipeline else 0)):
        round_()
    times = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            round_()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / steps * 1e3)
    times.sort()
    del trs
    torch.cuda.empty_cache()
    return times


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--parts", default="1,2")
    a = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    from upp_hip import _abi
    _abi.load()
    print("ms per 32 clouds (median of 3 timings of %d rounds; min / max beside it)" % a.steps)
    for pipeline in (True, False):
        for parts in [int(x) for x in a.parts.split(",")]:
            t = run(parts, pipeline, a.steps, a.warmup, device)
            print("  %-34s %d x B = %-2d   %.3f   (%.3f / %.3f)" % ("pipelined (2 streams per part)" if pipeline else "one stream per part",
                                                                  parts, 32 // parts, t[1], t[0], t[2]), flush=True)


if __name__ == "__main__":
    main()
