#!/bin/bash
# Regenerate profiles/r06_* on a GPU box (run from the repo root through gpurun), in three calls (each fits one gpurun limit):
#     gpurun --timeout 1200 -- 'bash tools/make_profiles.sh headline'
#     gpurun --timeout 1200 -- 'bash tools/make_profiles.sh recipes'      (bench lines + kernel summaries of the secondary recipes)
#     gpurun --timeout 1200 -- 'bash tools/make_profiles.sh micro'       (stand-alone kernel timings, stamps, censuses)
# rocprofv3 kernel traces of the headline step (one stream / pipelined), of the stand-alone kernels, two PMC passes
# (FETCH_SIZE, WRITE_SIZE: separate runs, never combined with a trace domain), then the un-profiled bench lines.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06
mkdir -p $O
PART=${1:-headline}
stats() { ls $1/*/*kernel_stats.csv | head -1; }
trace() { ls $1/*/*kernel_trace.csv | head -1; }
if [ "$PART" = "headline" ]; then
# (executions of the step per trace: 2 eager warm-ups + 2 replays + 10 timed = 14)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/seq -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipeline --no-stage-report > $O/seq.log 2>&1
cp "$(stats $O/seq)" $O/r06_bench_sequential_kernel_stats.csv
python3 tools/phase_summary.py "$(trace $O/seq)" > $O/r06_phase_summary.txt
python3 tools/step_census.py "$(trace $O/seq)" > $O/r06_step_census.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pipe -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stage-report > $O/pipe.log 2>&1
cp "$(stats $O/pipe)" $O/r06_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kern -- python3 tools/prof_kernels.py > $O/kern.log 2>&1
cp "$(stats $O/kern)" $O/r06_kernels_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 tools/prof_kernels.py > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 tools/prof_kernels.py > $O/write.log 2>&1
python3 tools/pmc_summary.py "$(ls $O/fetch/*/*counter_collection.csv | head -1)" "$(ls $O/write/*/*counter_collection.csv | head -1)" "$(trace $O/kern)" > $O/r06_pmc_kernels.json || { echo "PMC self-check FAILED (see pmc_summary.py)"; mv $O/r06_pmc_kernels.json $O/r06_pmc_kernels.REJECTED.json; }
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES --output-format csv -d $O/mfma -- python3 tools/prof_kernels.py > $O/mfma.log 2>&1
python3 tools/pmc_mfma_summary.py "$(ls $O/mfma/*/*counter_collection.csv | head -1)" "$(trace $O/kern)" > $O/r06_pmc_mfma.json
# the bench lines below join THESE counters (bench.py reads profiles/): copy them in, and refuse a line whose join lost a label
[ -f $O/r06_pmc_kernels.json ] && cp $O/r06_pmc_kernels.json profiles/r06_pmc_kernels.json
cp $O/r06_pmc_mfma.json profiles/r06_pmc_mfma.json
UPP_BENCH_STRICT=1 python3 bench.py --no-cpu-baseline --no-pipeline --detail $O/r06_bench_sequential_detail.json > $O/r06_bench_sequential.json 2> $O/bench_seq.err
UPP_BENCH_STRICT=1 python3 bench.py --detail $O/r06_bench_detail.json > $O/r06_bench.json 2> $O/bench.err
rm -rf $O/seq $O/pipe $O/kern $O/fetch $O/write $O/mfma
fi
if [ "$PART" = "recipes" ]; then
# secondary recipes: un-profiled lines + one kernel summary for the pre-training step (8 executions: 2 eager + 1 replay + 5 timed)
for w in cls_aux stage2 pretask pretrain seg; do python3 bench.py --workload $w --steps 10 --warmup 3 --detail $O/r06_workload_${w}_detail.json > $O/r06_workload_$w.json 2> $O/w_$w.err; done
python3 bench.py --workload seg --steps 10 --warmup 3 --no-cpu-baseline --no-pipeline --no-stage-report > $O/r06_workload_seg_sequential.json 2>> $O/w_seg.err
# kernel summaries of the recipes that used to run library GEMMs / unfused torch formulations (8 executions of the step each: 2 eager + 1 replay + 5 timed; one stream)
for w in pretrain seg stage2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$w -- python3 bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-pipeline --no-stage-report > $O/p_$w.log 2>&1
  cp "$(stats $O/p_$w)" $O/r06_workload_${w}_kernel_stats.csv
  python3 tools/step_census.py "$(trace $O/p_$w)" > $O/r06_workload_${w}_step_census.txt       # ONE replayed step (the stats file above divides construction and warm-up in)
  rm -rf $O/p_$w
done
fi
if [ "$PART" = "micro" ]; then
set +e          # (a failing helper loses its own file, not the rest)
python3 tools/glue_census.py 2> /dev/null | grep -v amdgpu.ids > $O/r06_glue_census.txt
for w in seg stage2 pretask; do python3 tools/glue_census_recipe.py $w --stack 2> /dev/null | grep -v amdgpu.ids > $O/r06_glue_census_$w.txt; done
python3 tools/linear_calls.py headline 2> /dev/null | grep -v amdgpu.ids > $O/r06_linear_calls_headline.txt
python3 tools/linear_calls.py seg 2> /dev/null | grep -v amdgpu.ids > $O/r06_linear_calls_seg.txt
python3 tools/micro/sb_stamps.py 2> /dev/null | grep -v amdgpu.ids > $O/r06_sb_stamps.txt
SB_CHECK_OUT=r06_sb_sweep_b24_b48.json python3 tools/micro/sb_model_check.py 24 48 2> /dev/null | grep -v amdgpu.ids > $O/r06_sb_model_b24_b48.txt
python3 tools/time_linear_sb.py --tiles --out $O/r06_time_linear_sb.jsonl > /dev/null 2>&1
python3 tools/time_attention.py 2> /dev/null | grep -v amdgpu.ids > $O/r06_time_attention.txt
python3 tools/micro/time_ln_adapter.py 2> /dev/null | grep -v amdgpu.ids >> $O/r06_time_attention.txt
python3 tools/micro/time_attention_long.py 2> /dev/null | grep -v amdgpu.ids > $O/r06_time_attention_long.txt
python3 tools/fps_sweep.py 2> /dev/null | grep -v amdgpu.ids > $O/r06_fps_sweep.txt
# determinism probes of round 4 (packed f32 beside a bf16-MFMA workgroup), their output kept under profiles/ since round 5: FPS beside a
# co-running split-bf16 stream, and the pipelined segmentation step against its serialised self
python3 tools/micro/fps_corun_probe.py 0 1024 2> /dev/null | grep -v amdgpu.ids > $O/r06_fps_corun_probe.txt
python3 tools/micro/pipe_race_probe.py 6 2> /dev/null | grep -v amdgpu.ids > $O/r06_pipe_race_probe.txt
mkdir -p tools/micro/bin
hipcc --offload-arch=gfx950 -O2 tools/micro/src/lds_canary.cpp -o tools/micro/bin/lds_canary -Liccv2025-upp_amd/upp_hip/lib -lupp_hip -Wl,-rpath,$PWD/iccv2025-upp_amd/upp_hip/lib 2> /dev/null
for a in "384 1536 0 0 2" "384 1536 0 0 0" "384 1536 0 0 1" "1536 384 0 0 1"; do ./tools/micro/bin/lds_canary $a 2>&1 | grep -v "^VGPR op_sel" ; done > $O/r06_packed_f32_canary.txt
fi
ls -la $O
