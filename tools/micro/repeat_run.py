"""Call one test function several times in ONE process and report every failure:  python tools/micro/repeat_run.py tests/test_gpu_block.py test_name [n]"""
import importlib.util, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("oracle", "tests", "iccv2025-upp_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, sys.argv[1]))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
fn = getattr(mod, sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
bad = 0
for i in range(n):
    try:
        fn()
        print("run %d ok" % i, flush=True)
    except Exception:
        bad += 1
        print("run %d FAILED" % i, flush=True)
        traceback.print_exc(limit=3)
print("%d of %d failed" % (bad, n))
