# A/B of the FPS launch forms (UPP_FPS_CPW clouds per workgroup, UPP_FPS_EXCL whole-LDS reservation) in the pipelined step; interleaved
mkdir -p gpurun_out/r05; rm -f gpurun_out/r05/fps_cpw.txt
for rep in 1 2 3; do for cfg in "1 0" "2 1" "4 1"; do set -- $cfg; export UPP_FPS_CPW=$1 UPP_FPS_EXCL=$2
python3 bench.py --steps 80 --warmup 5 --no-cpu-baseline --no-stage-report 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cpw $1 excl $2  step', round(d['ms_per_step'],4))" >> gpurun_out/r05/fps_cpw.txt
done; done
cat gpurun_out/r05/fps_cpw.txt
