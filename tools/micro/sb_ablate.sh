#!/bin/bash
# Ablation builds of csrc/linear_sb.hip on the GPU box: which phase bounds the split-bf16 loop?  (tools/time_linear_sb.py per variant)
cd "$(dirname "$0")/../.." || exit 1
for v in "" "-DUPP_SB_NO_MFMA" "-DUPP_SB_NO_DMA" "-DUPP_SB_NO_SPLIT" "-DUPP_SB_NO_MFMA -DUPP_SB_NO_SPLIT" "-DUPP_SB_NO_DMA -DUPP_SB_NO_SPLIT" $EXTRA_VARIANTS; do
    echo "== variant: [$v]"
    touch iccv2025-upp_amd/upp_hip/csrc/linear_sb.hip
    UPP_HIPCC_FLAGS="$v" python iccv2025-upp_amd/upp_hip/build.py > /dev/null 2>&1 || { echo build failed; exit 1; }
    python tools/time_linear_sb.py --rows ${ROWS:-2400} --out gpurun_out/sb_ablate_tmp.jsonl 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('   %-10s sb %6.2f us   f32 %6.2f us   sb+gelu_d %6.2f' % (r['shape'], r['us_sb'], r['us_f32'], r['us_sb_gelu_d']))
"
done
touch iccv2025-upp_amd/upp_hip/csrc/linear_sb.hip
python iccv2025-upp_amd/upp_hip/build.py > /dev/null 2>&1
