mkdir -p gpurun_out/r05; rm -f gpurun_out/r05/wgrad_rounds.txt
for w in seg pretrain; do for r in 3 1 1.5 2 4 6; do echo "== $w rounds $r" >> gpurun_out/r05/wgrad_rounds.txt; UPP_WGRAD_ROUNDS=$r python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-stage-report 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> gpurun_out/r05/wgrad_rounds.txt; done; done
cat gpurun_out/r05/wgrad_rounds.txt
