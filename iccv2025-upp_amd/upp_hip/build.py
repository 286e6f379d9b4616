"""Build libupp_hip.so (gfx950) in-tree with hipcc.

No torch C++ extension, no pybind: the product boundary is the plain C ABI in
include/upp_hip.h, loaded through ctypes (see _abi.py).  hipcc cross-compiles
for gfx950 without a GPU, so this also runs in the CPU-only build container.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libupp_hip.so")
SOURCES = ["abi.hip", "fps.hip", "knn.hip", "group.hip", "chamfer.hip", "emd.hip", "dense.hip", "linear.hip", "linear_sb.hip", "linear_rt.hip", "wgrad_sb.hip", "smallk.hip", "block.hip", "prop.hip", "optim.hip", "attn_flash16.hip", "attn_long.hip", "adapter.hip", "pointwise.hip", "head.hip"]
# -ffp-contract=off: every fma in the kernels is written explicitly so that the
# arithmetic matches the oracle bit for bit (see csrc/common.h sumsq3()).
# -target-feature -packed-fp32-ops: no v_pk_add/mul/fma_f32 in any kernel.  Measured on MI355X (tools/micro/src/lds_canary.cpp, round 4):
# v_pk_add_f32 with op_sel:[0,1] on a VGPR pair returns a - 0 in its low half every so often while a split-bf16 Linear workgroup
# (v_mfma_f32_32x32x16_bf16, 128 KB of LDS) is resident on the same CU -- never on an idle CU.  The FPS round loop used that form and
# picked other points beside the back-end of the pipelined step; hipcc also SLP-packs plain f32 arithmetic of other kernels into it.
# (the x86 pass of hipcc warns that it does not know the feature)
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-Wall", "-Wno-unused-function", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "..", "include", "upp_hip.h"))
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    # the objects on disk belong to ONE flag list: another UPP_HIPCC_FLAGS (an A/B or sweep build) recompiles everything, and so does the way back
    flags_now = " ".join(FLAGS + os.environ.get("UPP_HIPCC_FLAGS", "").split())
    stamp = os.path.join(OBJ, "flags.txt")
    if not (os.path.exists(stamp) and open(stamp).read() == flags_now):
        force = True
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _newer(obj, [src] + headers):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [_hipcc()] + FLAGS + os.environ.get("UPP_HIPCC_FLAGS", "").split() + ["-c", src, "-o", obj]   # (extra -D switches for A/B builds)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    with open(stamp, "w") as f:
        f.write(flags_now)
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in srcs]
    if force or jobs or _newer(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
