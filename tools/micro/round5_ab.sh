# The round's run-time-switchable changes off against on, same box, interleaved: tile table, FPS launch forms, long-attention tail
mkdir -p gpurun_out/r05; rm -f gpurun_out/r05/round5_ab.txt
for w in "" "--workload seg" "--workload cls_aux" "--workload pretrain"; do for rep in 1 2; do
UPP_SB_TUNED=0 UPP_PIPE_FPS_FORM=0 UPP_ATTN_FOLD=1 python3 bench.py $w --steps 40 --warmup 5 --no-cpu-baseline --no-stage-report 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w off', round(d['ms_per_step'],4))" >> gpurun_out/r05/round5_ab.txt
python3 bench.py $w --steps 40 --warmup 5 --no-cpu-baseline --no-stage-report 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w on ', round(d['ms_per_step'],4))" >> gpurun_out/r05/round5_ab.txt
done; done
cat gpurun_out/r05/round5_ab.txt
