#!/bin/bash
# Ablation builds of csrc/wgrad_sb.hip on the GPU box: which phase bounds the split-bf16 weight-gradient loop?
cd "$(dirname "$0")/../.." || exit 1
for v in "" "-DUPP_WS_NO_MFMA" "-DUPP_WS_NO_SPLIT" "-DUPP_WS_NO_READS" "-DUPP_WS_NO_SPLIT -DUPP_WS_NO_READS" "-DUPP_WS_NO_MFMA -DUPP_WS_NO_SPLIT" $EXTRA_VARIANTS; do
    echo "== variant: [$v]"
    touch iccv2025-upp_amd/upp_hip/csrc/wgrad_sb.hip
    UPP_HIPCC_FLAGS="$v" python iccv2025-upp_amd/upp_hip/build.py > /dev/null 2>&1 || { echo build failed; exit 1; }
    python tools/micro/time_wgrad.py 2>/dev/null | grep -v amdgpu | head -${LINES_SHOWN:-2}
done
touch iccv2025-upp_amd/upp_hip/csrc/wgrad_sb.hip
python iccv2025-upp_amd/upp_hip/build.py > /dev/null 2>&1
