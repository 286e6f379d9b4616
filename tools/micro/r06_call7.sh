set -o pipefail
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_block.py tests/test_gpu_model.py -m gpu -x -q > gpurun_out/r06/t7.log 2>&1; tail -2 gpurun_out/r06/t7.log
python tools/micro/time_ln_adapter.py 2>/dev/null | grep -v amdgpu
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --detail gpurun_out/r06/b7_detail.json > gpurun_out/r06/b7.json 2>/dev/null; python3 -c "
import json; d=json.load(open('gpurun_out/r06/b7.json')); print(d['ms_per_step'], d['ms_per_step_sequential'], {k:v[0] for k,v in d['kernels_brief'].items() if 'rowln' in k or 'adapter' in k})"
