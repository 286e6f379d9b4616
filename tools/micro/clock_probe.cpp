// Does the shader clock depend on how much of the chip a kernel occupies?  A dependent-fmaf spin loop on G workgroups of 256 threads; per
// workgroup: shader cycles (s_memtime) and 100 MHz wall ticks (s_memrealtime) -> GHz, and cycles per dependent fmaf.
//   hipcc --offload-arch=gfx950 -O3 clock_probe.cpp -o bin/clock_probe && ./bin/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CHAINS>
__global__ void spin(float *out, unsigned long long *stamps, int iters) {
    float a[4] = {threadIdx.x * 1e-3f, threadIdx.x * 2e-3f, threadIdx.x * 3e-3f, threadIdx.x * 4e-3f};
    const float b = 1.0001f;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {                  // 4 fmaf per trip: one chain of 4, two chains of 2, or four independent ones
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q % CHAINS] = __builtin_fmaf(a[q % CHAINS], b, 1e-7f);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = w1 - w0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (a[0] + a[1]) + (a[2] + a[3]);
}
int main() {
    const int iters = 200000;
    for (int chains : {1, 2, 4})
    for (int G : {32, 512, 2048}) {
        float *out; unsigned long long *st;
        hipMalloc(&out, sizeof(float) * G * 256); hipMalloc(&st, sizeof(unsigned long long) * 2 * G);
        for (int rep = 0; rep < 3; ++rep) {
            if (chains == 1) hipLaunchKernelGGL(spin<1>, dim3(G), dim3(256), 0, 0, out, st, iters);
            else if (chains == 2) hipLaunchKernelGGL(spin<2>, dim3(G), dim3(256), 0, 0, out, st, iters);
            else hipLaunchKernelGGL(spin<4>, dim3(G), dim3(256), 0, 0, out, st, iters);
        }
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(2 * G);
        hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * G, hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0;
        for (int i = 0; i < G; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
        cyc /= G; wall /= G;
        printf("chains %d  G = %4d: %.0f shader-counter ticks, %.0f wall ticks (100 MHz) -> counter at %.3f GHz; %.2f counter ticks per fmaf, %.2f ns\n", chains, G, cyc, wall,
               cyc / (wall * 10.0), cyc / (4.0 * iters), wall * 10.0 / (4.0 * iters));
        hipFree(out); hipFree(st);
    }
    return 0;
}
