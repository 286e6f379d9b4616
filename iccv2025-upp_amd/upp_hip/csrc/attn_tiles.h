// attn_tiles.h -- FP32-MFMA tile helpers of attn_long.hip (L <= 160; and of tools/micro/attn_mfma.hip, round 1's L <= 96 kernels).
// Operand tiles live in LDS with a row stride of kLD = 65 floats, so that both the row-major and the transposed
// one-dword-per-lane operand reads of v_mfma_f32_32x32x2_f32 are bank-conflict free.  K (the contraction length) is one of
// 32 / 64 / 96 / 128 / 160.
#pragma once
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kLD = 65;          // row stride of the (L x 64) operand tiles

// acc(32x32) += A(32 x K) . B(K x 32).  Element A(i,k): TA ? a[k*lda + i] : a[i*lda + k];  B(k,j): TB ? b[j*ldb + k] : b[k*ldb + j]
template <bool TA, bool TB, int K>
__device__ __forceinline__ void mfma_tile_k(f32x16 &acc, const float *a, int lda, const float *b, int ldb, int lr, int lk) {
    // A wave runs ONE accumulator chain (64 cycles per MFMA): the operands of the next CH k-steps are fetched from
    // LDS into registers before the current CH MFMAs issue, so the ~100+ cycle ds_read latency hides behind them.
    constexpr int CH = 8, NCH = K / 2 / CH;
    static_assert(K % (2 * CH) == 0, "K must be a multiple of 16");
    float av[CH], bv[CH];
    auto fetch = [&](int c, float (&x)[CH], float (&y)[CH]) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int k0 = (c * CH + i) * 2;
            x[i] = TA ? a[(k0 + lk) * lda + lr] : a[lr * lda + k0 + lk];
            y[i] = TB ? b[lr * ldb + k0 + lk] : b[(k0 + lk) * ldb + lr];
        }
    };
    fetch(0, av, bv);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float an[CH], bn[CH];
        if (c + 1 < NCH) fetch(c + 1, an, bn);
        __builtin_amdgcn_sched_barrier(0);   // hipcc otherwise sinks the reads next to their use (read, wait, mfma, read, ...)
#pragma unroll
        for (int i = 0; i < CH; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NCH) {
#pragma unroll
            for (int i = 0; i < CH; ++i) { av[i] = an[i]; bv[i] = bn[i]; }
        }
    }
}
// NB output tiles that share their A operand: acc[n] += A . B_n.  The NB accumulator chains are independent, so their
// MFMAs interleave in the matrix pipe (a single chain issues a dependent MFMA only every ~2x its pass count: measured
// 2.1-2.5x the MFMA-bound time for one-accumulator tiles), and A is read from LDS once for all NB tiles.
template <bool TA, bool TB, int K, int NB>
__device__ __forceinline__ void mfma_tiles_k(f32x16 (&acc)[NB], const float *a, int lda, const float *const (&b)[NB], int ldb, int lr, int lk) {
    constexpr int CH = 8, NCH = K / 2 / CH;
    static_assert(K % (2 * CH) == 0, "K must be a multiple of 16");
    float av[CH], bv[NB][CH];
    auto fetch = [&](int c, float (&x)[CH], float (&y)[NB][CH]) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int k0 = (c * CH + i) * 2;
            x[i] = TA ? a[(k0 + lk) * lda + lr] : a[lr * lda + k0 + lk];
#pragma unroll
            for (int n = 0; n < NB; ++n) y[n][i] = TB ? b[n][lr * ldb + k0 + lk] : b[n][(k0 + lk) * ldb + lr];
        }
    };
    fetch(0, av, bv);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float an[CH], bn[NB][CH];
        if (c + 1 < NCH) fetch(c + 1, an, bn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[n][i], acc[n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NCH) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                av[i] = an[i];
#pragma unroll
                for (int n = 0; n < NB; ++n) bv[n][i] = bn[n][i];
            }
        }
    }
}
template <bool TA, bool TB, int NB>
__device__ __forceinline__ void mfma_tiles(f32x16 (&acc)[NB], const float *a, int lda, const float *const (&b)[NB], int ldb, int K, int lr, int lk) {
    if (K == 32) mfma_tiles_k<TA, TB, 32, NB>(acc, a, lda, b, ldb, lr, lk);
    else if (K == 64) mfma_tiles_k<TA, TB, 64, NB>(acc, a, lda, b, ldb, lr, lk);
    else if (K == 96) mfma_tiles_k<TA, TB, 96, NB>(acc, a, lda, b, ldb, lr, lk);
    else if (K == 128) mfma_tiles_k<TA, TB, 128, NB>(acc, a, lda, b, ldb, lr, lk);
    else mfma_tiles_k<TA, TB, 160, NB>(acc, a, lda, b, ldb, lr, lk);
}
// K is a compile-time constant so that the operand loop unrolls fully and the LDS reads run ahead of the MFMAs
template <bool TA, bool TB>
__device__ __forceinline__ void mfma_tile(f32x16 &acc, const float *a, int lda, const float *b, int ldb, int K, int lr, int lk) {
    if (K == 32) mfma_tile_k<TA, TB, 32>(acc, a, lda, b, ldb, lr, lk);
    else if (K == 64) mfma_tile_k<TA, TB, 64>(acc, a, lda, b, ldb, lr, lk);
    else if (K == 96) mfma_tile_k<TA, TB, 96>(acc, a, lda, b, ldb, lr, lk);
    else if (K == 128) mfma_tile_k<TA, TB, 128>(acc, a, lda, b, ldb, lr, lk);
    else mfma_tile_k<TA, TB, 160>(acc, a, lda, b, ldb, lr, lk);
}
// exp(x) for x <= 0 as v_exp_f32(x * log2 e): relative error ~ |x| * 6e-8 (the product's rounding), i.e. < 2e-6 for every
// term that matters in a softmax (x > -30); libm's expf costs ~40 VALU instructions per element and made the softmax
// the longest phase of the forward kernel.
__device__ __forceinline__ float exp_neg(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ void zero(f32x16 &a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.0f;
}
__device__ __forceinline__ int tile_row(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }

}  // namespace
