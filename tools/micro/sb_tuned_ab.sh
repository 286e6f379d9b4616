mkdir -p gpurun_out/r05
rm -f gpurun_out/r05/sb_tuned_ab.txt
run() { flags="$1"; shift; echo "== $* $flags" >> gpurun_out/r05/sb_tuned_ab.txt; env "$@" python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-stage-report $flags 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> gpurun_out/r05/sb_tuned_ab.txt; }
run "" UPP_SB_TUNED=0
run "" UPP_SB_TUNED=1
run "" UPP_SB_TUNED=0
run "" UPP_SB_TUNED=1
run "--no-pipeline" UPP_SB_TUNED=0
run "--no-pipeline" UPP_SB_TUNED=1
run "--workload seg" UPP_SB_TUNED=0
run "--workload seg" UPP_SB_TUNED=1
run "--workload pretrain" UPP_SB_TUNED=0
run "--workload pretrain" UPP_SB_TUNED=1
cat gpurun_out/r05/sb_tuned_ab.txt
