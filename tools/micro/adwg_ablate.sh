#!/bin/bash
# Ablation builds of adapter_wgrad_kernel timed stand-alone (run on the GPU box from the repo root).
mkdir -p tools/micro/bin
for v in "" "-DUPP_ADWG_NO_LOAD" "-DUPP_ADWG_NO_COMPUTE" "-DUPP_ADWG_NO_BIAS" "-DUPP_ADWG_NO_COMPUTE -DUPP_ADWG_NO_BIAS -DUPP_ADWG_SKELETON" "-DUPP_ADWG_NO_LOAD -DUPP_ADWG_NO_COMPUTE -DUPP_ADWG_NO_BIAS -DUPP_ADWG_SKELETON"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Wno-unused-function -Wno-unused-value -Xclang -target-feature -Xclang -packed-fp32-ops \
        -I iccv2025-upp_amd/upp_hip/csrc -I include $v tools/micro/src/adwg_ablate.hip iccv2025-upp_amd/upp_hip/csrc/abi.hip -o tools/micro/bin/adwg_ablate 2> /tmp/adwg_build.log \
        || { echo "build failed: $v"; tail -5 /tmp/adwg_build.log; exit 1; }
    echo "== ${v:-full}"
    ./tools/micro/bin/adwg_ablate ${1:-2080} | tail -1
done
