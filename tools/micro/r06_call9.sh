set -o pipefail
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_model.py tests/test_gpu_parity.py -m gpu -q > gpurun_out/r06/t9.log 2>&1; tail -2 gpurun_out/r06/t9.log
python tools/time_emd.py 2>/dev/null | grep -v amdgpu | tail -6
python bench.py --workload cls_aux --steps 20 --warmup 5 --no-cpu-baseline --no-stage-report 2>/dev/null | cut -c1-330
