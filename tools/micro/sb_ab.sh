#!/bin/bash
# A/B of build variants of csrc/linear_sb.hip on one box, each variant timed REPS times in rotation.   VARIANTS="flags1|flags2|..."
cd "$(dirname "$0")/../.." || exit 1
IFS='|' read -ra VS <<< "${VARIANTS:-|-DUPP_SB_NO_TIE}"
mkdir -p gpurun_out/ab
i=0
for v in "${VS[@]}"; do
    touch iccv2025-upp_amd/upp_hip/csrc/linear_sb.hip
    UPP_HIPCC_FLAGS="$v" python iccv2025-upp_amd/upp_hip/build.py > /dev/null 2>&1 || { echo "build failed: $v"; exit 1; }
    cp iccv2025-upp_amd/upp_hip/lib/libupp_hip.so gpurun_out/ab/lib_$i.so
    i=$((i+1))
done
for rep in $(seq 1 ${REPS:-2}); do
    i=0
    for v in "${VS[@]}"; do
        cp gpurun_out/ab/lib_$i.so iccv2025-upp_amd/upp_hip/lib/libupp_hip.so
        echo "== rep $rep variant [$v]"
        python tools/time_linear_sb.py --rows ${ROWS:-2400} --out gpurun_out/ab/tmp.jsonl 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('   %-10s sb %6.2f us   f32 %6.2f us   sb+gelu_d %6.2f' % (r['shape'], r['us_sb'], r['us_f32'], r['us_sb_gelu_d']))
"
        i=$((i+1))
    done
done
rm -rf gpurun_out/ab
touch iccv2025-upp_amd/upp_hip/csrc/linear_sb.hip
python iccv2025-upp_amd/upp_hip/build.py > /dev/null 2>&1
