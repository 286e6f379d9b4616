#!/bin/bash
# Which part of csrc/fps.hip is disturbed by a co-running split-bf16 Linear kernel?  (tools/micro/fps_corun_probe.py per build variant)
cd "$(dirname "$0")/../.." || exit 1
for v in "" "-DUPP_FPS_DIAG_NOASM" "-DUPP_FPS_DIAG_SCALAR" "-DUPP_FPS_DIAG_NOLDS" "-DUPP_FPS_DIAG_NOASM -DUPP_FPS_DIAG_SCALAR -DUPP_FPS_DIAG_NOLDS"; do
    echo "== variant: [$v]"
    touch iccv2025-upp_amd/upp_hip/csrc/fps.hip
    UPP_HIPCC_FLAGS="$v" python iccv2025-upp_amd/upp_hip/build.py > /dev/null 2>&1 || { echo build failed; exit 1; }
    python tools/micro/fps_corun_probe.py 0 1024 2>/dev/null | grep co-runner
done
touch iccv2025-upp_amd/upp_hip/csrc/fps.hip
python iccv2025-upp_amd/upp_hip/build.py > /dev/null 2>&1
