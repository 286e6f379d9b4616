#!/usr/bin/env python3
"""Eager / eager, graph / graph and eager / graph trajectories of a recipe with dropout and stochastic depth at 0: loss per step and the
largest gradient difference.  Tells run-to-run noise (atomic adds, remaining random draws) from a replay that computes something else.
   python tools/micro/replay_vs_eager.py <recipe> [batch] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch  # noqa: E402

import bench  # noqa: E402

kind = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)


def run(use_graph):
    tr = bench.RecipeTrainer(kind, dev, batch, use_graph=use_graph, pipeline=False)
    for m in tr.model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0
    out = []
    for _ in range(steps):
        loss = tr.step()
        torch.cuda.synchronize()
        out.append((float(loss), tr.ts.flat.flat.clone()))
    return out


runs = {"eager A": run(False), "eager B": run(False), "graph A": run(True), "graph B": run(True)}
for a, b in (("eager A", "eager B"), ("graph A", "graph B"), ("eager A", "graph A")):
    print("%s vs %s:" % (a, b), "  ".join("step %d: loss %.6f / %.6f, grad diff %.1e of %.1e" % (k, la, lb, float((ga - gb).abs().max()), float(ga.abs().max()))
                                          for k, ((la, ga), (lb, gb)) in enumerate(zip(runs[a], runs[b]))), flush=True)
