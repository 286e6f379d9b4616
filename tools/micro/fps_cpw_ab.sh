# A/B of the FPS launch forms (UPP_PIPE_FPS_FORM=<clouds per workgroup>,<whole-LDS reservation> -- a host-side switch of upp_hip/train.py; the library takes the form as an argument of upp_fps_ex) in the pipelined step; interleaved
mkdir -p gpurun_out/r06; rm -f gpurun_out/r06/fps_cpw.txt
for rep in 1 2 3; do for cfg in "1 0" "2 1" "4 1"; do set -- $cfg; export UPP_PIPE_FPS_FORM=$1,$2
python3 bench.py --steps 80 --warmup 5 --no-cpu-baseline --no-stage-report 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cpw $1 excl $2  step', round(d['ms_per_step'],4))" >> gpurun_out/r06/fps_cpw.txt
done; done
cat gpurun_out/r06/fps_cpw.txt
