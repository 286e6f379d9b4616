// Ablation timing of adapter_wgrad_kernel (csrc/adapter.hip) outside the library: the whole file is included, variants by -D.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I iccv2025-upp_amd/upp_hip/csrc -I include [-DUPP_ADWG_NO_LOAD | -DUPP_ADWG_NO_COMPUTE |
//         -DUPP_ADWG_NO_BIAS] tools/micro/src/adwg_ablate.hip iccv2025-upp_amd/upp_hip/csrc/abi.hip -o tools/micro/bin/adwg_ablate
#include "../../../iccv2025-upp_amd/upp_hip/csrc/adapter.hip"
#include <cstdio>
#include <vector>

int main(int argc, char **argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 2080, jobs = 12, D = 384, H = 32;
    const int splits = upp_adapter_wgrad_splits(R);
    std::vector<const float *> xo(jobs), mean(jobs), rstd(jobs), gamma(jobs), beta(jobs), go(jobs), fac(jobs);
    std::vector<float *> part(jobs);
    std::vector<int> rows(jobs, R);
    std::vector<float> scale(jobs, 0.7f);
    auto dev = [&](size_t n) { float *p; hipMalloc(&p, n * sizeof(float)); hipMemset(p, 0, n * sizeof(float)); return p; };
    for (int j = 0; j < jobs; ++j) {
        xo[j] = dev((size_t)R * D); go[j] = dev((size_t)R * D); fac[j] = dev((size_t)R * 2 * H); mean[j] = dev(R); rstd[j] = dev(R);
        gamma[j] = dev(D); beta[j] = dev(D); part[j] = dev((size_t)splits * (2 * H * D + H + D));
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i)
            upp_adapter_wgrad_batched(xo.data(), mean.data(), rstd.data(), gamma.data(), beta.data(), go.data(), fac.data(), rows.data(), scale.data(),
                                      part.data(), jobs, splits, D, H, nullptr);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("R = %d, %d splits: %.1f us per launch\n", R, splits, ms * 1000.0f / 20);
    }
    return 0;
}
