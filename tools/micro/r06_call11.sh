set -o pipefail
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_pointwise.py tests/test_gpu_model.py tests/test_pretask.py tests/test_point_mae.py tests/test_gpu_declined.py -m gpu -x -q > gpurun_out/r06/t11.log 2>&1; tail -3 gpurun_out/r06/t11.log
for w in stage2 pretask; do python tools/glue_census_recipe.py $w --stack 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r06/glue_${w}_stack2.txt; head -1 gpurun_out/r06/glue_${w}_stack2.txt; done
for w in stage2 pretask pretrain; do python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-stage-report 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['ms_per_step'],4))"; done
