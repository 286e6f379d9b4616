// MFMA main-loop structures on LDS-resident random tiles (no global traffic): which shape sustains the matrix pipe?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDT = 132;
// TI x TJ 32x32-tiles per wave, PF = operand prefetch distance (0/1), BAR = k-steps per barrier
template <int TI, int TJ, int PF, int BAR>
__global__ __launch_bounds__(256) void loop_kernel(const float *in, float *out, int ksteps) {
    __shared__ float As[2][16][LDT * 2];
    __shared__ float Ws[2][16][LDT * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 16 * LDT * 2; i += 256) { (&As[0][0][0])[i] = in[i & 0xFFFFF]; (&Ws[0][0][0])[i] = in[(i + 77777) & 0xFFFFF]; }
    __syncthreads();
    const int lr = lane & 31, lk = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[TI][TJ] = {};
    float a[TI], b[TJ];
    auto rd = [&](int buf, int kk, float (&x)[TI], float (&y)[TJ]) {
#pragma unroll
        for (int i = 0; i < TI; ++i) x[i] = As[buf][(kk + lk) & 15][(wm * TI * 32 + i * 32 + lr) % (LDT * 2 - 4)];
#pragma unroll
        for (int j = 0; j < TJ; ++j) y[j] = Ws[buf][(kk + lk) & 15][(wn * TJ * 32 + j * 32 + lr) % (LDT * 2 - 4)];
    };
    if (PF) rd(0, 0, a, b);
    for (int s = 0; s < ksteps; ++s) {
        const int buf = (s / BAR) & 1, kk = (s * 2) & 15;
        float na[TI], nb[TJ];
        if (PF) rd(buf, kk + 2, na, nb); else rd(buf, kk, a, b);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (PF) {
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = na[i];
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = nb[j];
        }
        if ((s + 1) % BAR == 0) __syncthreads();
    }
    float sum = 0;
    for (int i = 0; i < TI; ++i) for (int j = 0; j < TJ; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = sum;
}
template <int TI, int TJ, int PF, int BAR>
void run(const char *name, const float *in, float *out, int wgs_per_cu) {
    const int blocks = 256 * wgs_per_cu * 2, ksteps = 8192;      // two rounds of workgroups
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(s);
        for (int q = 0; q < 3; ++q) hipLaunchKernelGGL((loop_kernel<TI, TJ, PF, BAR>), dim3(blocks), dim3(256), 0, 0, in, out, ksteps);
        hipEventRecord(e); hipEventSynchronize(e);
        hipEventElapsedTime(&ms, s, e);
    }
    const double flops = 3.0 * blocks * 4 * (double)ksteps * TI * TJ * 4096;
    printf("%-34s blocks %4d : %8.2f ms, %.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
}
int main() {
    const int n = 1 << 20;
    std::vector<float> h(n); srand(1); for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *in, *out; hipMalloc(&in, n * 4); hipMalloc(&out, 256 * 4096 * 4);
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int w : {1, 2}) {
        run<2, 2, 0, 8>("2x2 tiles, no prefetch, bar/8", in, out, w);
        run<2, 2, 1, 8>("2x2 tiles, prefetch, bar/8", in, out, w);
        run<2, 2, 1, 1024>("2x2 tiles, prefetch, no barrier", in, out, w);
        run<2, 4, 1, 8>("2x4 tiles, prefetch, bar/8", in, out, w);
    }
    run<4, 4, 1, 8>("4x4 tiles (1 wave/SIMD), prefetch", in, out, 1);
    run<4, 4, 0, 8>("4x4 tiles (1 wave/SIMD), no pf", in, out, 1);
    run<3, 4, 1, 8>("3x4 tiles (1 wave/SIMD), prefetch", in, out, 1);
    return 0;
}
