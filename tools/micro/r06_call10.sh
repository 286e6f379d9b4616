set -o pipefail
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_pointwise.py tests/test_seg_golden.py tests/test_gpu_declined.py tests/test_gpu_determinism.py -m gpu -x -q > gpurun_out/r06/t10.log 2>&1; tail -3 gpurun_out/r06/t10.log
python bench.py --workload seg --steps 10 --warmup 3 --no-cpu-baseline --no-stage-report 2>/dev/null | cut -c1-330
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06/p_seg -- python3 bench.py --workload seg --steps 5 --warmup 1 --no-cpu-baseline --no-pipeline --no-stage-report > gpurun_out/r06/p_seg.log 2>&1
python3 tools/step_census.py "$(ls gpurun_out/r06/p_seg/*/*kernel_trace.csv | head -1)" > gpurun_out/r06/seg2_step_census.txt; rm -rf gpurun_out/r06/p_seg
head -3 gpurun_out/r06/seg2_step_census.txt
