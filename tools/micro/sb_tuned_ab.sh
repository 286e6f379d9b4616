# A/B of the measured tile table (csrc/linear_sb_tuned.h; UPP_SB_TUNED=0: the cost model alone) on whole steps
mkdir -p gpurun_out/r05
rm -f gpurun_out/r05/sb_tuned_ab.txt
run() { flags="$1"; shift; echo "== $* $flags" >> gpurun_out/r05/sb_tuned_ab.txt; env "$@" python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-stage-report $flags 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> gpurun_out/r05/sb_tuned_ab.txt; }
for w in "" "--no-pipeline" "--workload seg" "--workload pretrain" "--workload pretask" "--workload stage2" "--workload cls_aux"; do
run "$w" UPP_SB_TUNED=0
run "$w" UPP_SB_TUNED=1
done
cat gpurun_out/r05/sb_tuned_ab.txt
