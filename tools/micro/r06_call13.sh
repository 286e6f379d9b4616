set -o pipefail
mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -q > gpurun_out/r06/t_full.log 2>&1; tail -3 gpurun_out/r06/t_full.log
for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-stage-report 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cls', round(d['ms_per_step'],4), round(d['ms_per_step_min'],4), round(d['ms_per_step_max'],4))"; done
