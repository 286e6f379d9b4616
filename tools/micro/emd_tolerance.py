"""Measured deviation of the HIP approxmatch / matchcost from the oracle (what tests/test_gpu_parity.py::test_emd_against_oracle's bounds
are set from).   python tools/micro/emd_tolerance.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("oracle", "tests", "iccv2025-upp_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import oracle as O
from upp_hip import ops
import test_gpu_parity as T

for B, n, m in T.EMD_SHAPES:
    if (n, m) == (2, 2):
        continue
    a, b = T.clouds(B, n, "ball", n), T.clouds(B, m, "ball", m + 1)
    wm = O.emd_approxmatch(a, b)
    hm = ops.emd_approxmatch(T.dev(a), T.dev(b)).cpu().numpy()
    wc = O.emd_matchcost(a, b, wm)
    hc = ops.emd_matchcost(T.dev(a), T.dev(b), T.dev(hm)).cpu().numpy()
    big = np.abs(wm) > 1e-3 * np.abs(wm).max()
    print("B %d n %d m %d: match max abs %.2e (scale %.2e), max rel on entries > 1e-3 scale %.2e, cost rel %.2e" % (
        B, n, m, np.abs(hm - wm).max(), np.abs(wm).max(), (np.abs(hm - wm)[big] / np.abs(wm)[big]).max(), np.abs(hc / wc - 1).max()))
