// What does an s_nop cost a lone wave?  4 fmaf per trip, with 0 / 4 / 8 `s_nop 1` between them; and v_cmp -> v_cndmask pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NOPS>
__global__ void spin(float *out, unsigned long long *stamps, int iters) {
    float a = threadIdx.x * 1e-3f, c = threadIdx.x * 2e-3f;
    const float b = 1.0001f;
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        a = __builtin_fmaf(a, b, 1e-7f); if (NOPS >= 4) asm volatile("s_nop 1"); if (NOPS >= 8) asm volatile("s_nop 1");
        c = __builtin_fmaf(c, b, 1e-7f); if (NOPS >= 4) asm volatile("s_nop 1"); if (NOPS >= 8) asm volatile("s_nop 1");
        a = __builtin_fmaf(a, b, 1e-7f); if (NOPS >= 4) asm volatile("s_nop 1"); if (NOPS >= 8) asm volatile("s_nop 1");
        c = __builtin_fmaf(c, b, 1e-7f); if (NOPS >= 4) asm volatile("s_nop 1"); if (NOPS >= 8) asm volatile("s_nop 1");
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) stamps[blockIdx.x] = c1 - c0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + c;
}
int main() {
    const int iters = 200000, G = 32;
    float *out; unsigned long long *st;
    hipMalloc(&out, sizeof(float) * G * 256); hipMalloc(&st, sizeof(unsigned long long) * G);
    for (int nops : {0, 4, 8}) {
        for (int rep = 0; rep < 2; ++rep) {
            if (nops == 0) hipLaunchKernelGGL(spin<0>, dim3(G), dim3(256), 0, 0, out, st, iters);
            else if (nops == 4) hipLaunchKernelGGL(spin<4>, dim3(G), dim3(256), 0, 0, out, st, iters);
            else hipLaunchKernelGGL(spin<8>, dim3(G), dim3(256), 0, 0, out, st, iters);
        }
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(G);
        hipMemcpy(h.data(), st, sizeof(unsigned long long) * G, hipMemcpyDeviceToHost);
        printf("%d s_nop per trip of 4 fmaf: %.1f ticks per trip\n", nops, (double)h[0] / iters);
    }
    return 0;
}
