# A/B of run-time switches on the headline step (interleaved repetitions; one box)
mkdir -p gpurun_out/r05; rm -f gpurun_out/r05/switch_ab.txt
run() { echo "== $*" >> gpurun_out/r05/switch_ab.txt; env "$@" python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-stage-report 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_sequential'] if 'ms_per_step_sequential' in d else '')" >> gpurun_out/r05/switch_ab.txt; }
for r in 1 2 3; do
run UPP_SB_XCD2D=2
run UPP_SB_XCD2D=0
run UPP_SB_XCD2D=4
done
cat gpurun_out/r05/switch_ab.txt
