"""Library-GEMM solution selection for the transformer's Linear layers on MI355X.

The plain GEMMs of the model (qkv / proj / fc1 / fc2 and their data-gradients, M = B*L rows of 384..1536 columns) go
to hipBLASLt / rocBLAS through torch.  The libraries' default heuristics pick poor tiles for some of these shapes
(e.g. 2400x1152x384 runs at 65 TFLOP/s with the default and 99 TFLOP/s with the best solution), so the training
entry points switch on PyTorch's TunableOp: every GEMM shape is timed once against all library solutions, in the eager
warm-up steps before the HIP graph is captured, and the winner is used from then on.  `tuned/gemm_gfx950.csv` holds the
selections measured on MI355X for the shapes of the shipped configs (loaded when its validator lines match the
installed ROCm / hipBLASLt build; otherwise the shapes are simply tuned again).
"""
import os
import tempfile

import torch

_TUNED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gemm_gfx950.csv")
_state = {"on": False}


def enable(tune_missing=True, results_file=_TUNED):
    """Turn on GEMM solution selection.  Idempotent.  Returns True when TunableOp is active."""
    if _state["on"]:
        return True
    if not torch.cuda.is_available():
        raise RuntimeError("gemm_tuning.enable() needs a HIP device")
    tun = torch.cuda.tunable
    tun.enable(True)
    # TunableOp dumps its table at exit: keep that out of the working directory
    # (UPP_GEMM_TUNING_OUT=<file> keeps it: that is how tuned/gemm_gfx950.csv is refreshed)
    tun.set_filename(os.environ.get("UPP_GEMM_TUNING_OUT") or os.path.join(tempfile.gettempdir(), "upp_tunableop_%d.csv" % os.getpid()))
    tun.tuning_enable(bool(tune_missing))
    if results_file and os.path.isfile(results_file):
        try:
            tun.read_file(results_file)
        except Exception:       # a results file from another ROCm build: tune afresh
            pass
    _state["on"] = True
    return True


def disable():
    torch.cuda.tunable.enable(False)
    _state["on"] = False
