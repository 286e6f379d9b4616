set -o pipefail
mkdir -p gpurun_out/r06
SB_CHECK_OUT=r06_sb_sweep_b24_b48.json python tools/micro/sb_model_check.py 24 48 > gpurun_out/r06/r06_sb_model_b24_b48.txt 2>/dev/null; tail -1 gpurun_out/r06/r06_sb_model_b24_b48.txt
SB_CHECK_OUT=r06_sb_sweep_b12_20_40_56.json python tools/micro/sb_model_check.py 12 20 40 56 > gpurun_out/r06/r06_sb_model_b12_20_40_56.txt 2>/dev/null; tail -1 gpurun_out/r06/r06_sb_model_b12_20_40_56.txt
