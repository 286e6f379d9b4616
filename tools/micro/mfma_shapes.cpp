// bare f32 MFMA loops on random operands: 32x32x2 vs 16x16x4, 1..3 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(256) void loop_kernel(const float *in, float *out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(tid * 16 + i) & 0xFFFFF]; b[i] = in[(tid * 16 + 8 + i) & 0xFFFFF]; }
    if (SHAPE == 32) {
        f32x16 acc[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + 1) & 7], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + 1) & 7], b[k], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + 1) & 7], b[(k + 1) & 7], acc[3], 0, 0, 0);
            }
        }
        float s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[tid] = s;
    } else {
        f32x4 acc[16] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + i) & 7], b[(k + j) & 7], acc[i * 4 + j], 0, 0, 0);
        }
        float s = 0; for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
        out[tid] = s;
    }
}
int main() {
    const int n = 1 << 20;
    std::vector<float> h(n); srand(1); for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *in, *out; hipMalloc(&in, n * 4); hipMalloc(&out, 256 * 2048 * 4 * 4);
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int wps = 1; wps <= 3; ++wps) {
        const int blocks = 256 * wps, iters = 4000;
        for (int shape : {32, 16}) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(s);
                for (int q = 0; q < 10; ++q) {
                    if (shape == 32) hipLaunchKernelGGL(loop_kernel<32>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
                    else hipLaunchKernelGGL(loop_kernel<16>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
                }
                hipEventRecord(e); hipEventSynchronize(e);
                hipEventElapsedTime(&ms, s, e);
            }
            const double flops = 10.0 * blocks * 4 * (double)iters * 32 * 4096;
            printf("waves/SIMD %d  shape %2d : %.2f ms, %.1f TFLOP/s\n", wps, shape, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
