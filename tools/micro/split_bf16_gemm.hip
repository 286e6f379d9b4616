// split_bf16_gemm.hip -- EXPERIMENT (not part of libupp_hip.so): f32 Linear layers on the bf16 matrix pipe of gfx950,
// at f32 accuracy.  Outcome on MI355X (tools/micro/README.md): the arithmetic is as accurate as an f32 GEMM, but the
// kernels only tie the tuned library f32 GEMMs -- see the README for where the time goes.
//
// The reference's Linear layers (Attention.qkv / proj, Mlp.fc1 / fc2: models/Point_MAE_pretask_dev.py:153-196)
// are f32 GEMMs.  gfx950 issues v_mfma_f32_32x32x2_f32 at 1/16 of the bf16 MFMA rate, so an f32-MFMA GEMM is
// bound by a 157 TFLOP/s pipe.  Here every f32 operand is split EXACTLY into three bf16 terms
//     x = hi + mid + lo,   hi = rne_bf16(x), mid = rne_bf16(x - hi), lo = rne_bf16(x - hi - mid)
// (both residuals are exact in f32; three 8-bit significands + signs cover the 24-bit significand), and
//     a*b = ah*bh + (ah*bm + am*bh) + (ah*bl + al*bh + am*bm)  +  O(2^-24 |a*b|)
// is accumulated in f32 by six v_mfma_f32_32x32x16_bf16 per k-block.  Each bf16 x bf16 product is exact in f32;
// the three dropped cross terms (am*bl, al*bm, al*bl) are below one f32 ulp of the product.  The result carries
// the error of an f32 GEMM (split_bf16_probe.py checks both against f64) at 6/16 of the f32-MFMA time.
//
// Weights are frozen in the PEFT recipes, so their three planes are produced once (upp_split_bf16: [3][N][K]
// bf16, optionally of the transpose, which is the operand of the input-gradient GEMM); activations are split
// while they are staged into LDS.
//
// Kernel: C (M,N) = epilogue(A (M,K) . W^T), W planes (3,N,K).  Workgroup = 4 waves (2x2), wave tile
// (32 TM) x (32 TN), k-step 32 (two MFMA k-blocks).  LDS rows are 32 bf16 + 8 pad (80 B): the 16 lanes of one
// ds_read_b128 phase land on 16 distinct 16-byte bank groups (5r mod 16).  One LDS buffer, operands of the next
// k-step prefetched into registers while the MFMAs of the current one run: the small grids of the M = 2400
// token GEMMs keep every workgroup resident, so other workgroups cover the two barriers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#define UPP_E_BADARG (-1)
#define UPP_E_RANGE (-2)
static inline int upp_launch_status(void) { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : (int)e; }

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 32;        // f32 elements of K per LDS stage
constexpr int LDR = 40;       // bf16 elements per LDS row (32 + 8 pad) = 80 bytes

enum { EPI_BIAS = 1, EPI_GELU = 2, EPI_PRE = 4 };

__device__ __forceinline__ float gelu_val(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// two f32 -> packed bf16 pair (round to nearest even: v_cvt_pk_bf16_f32) and back
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float bf16_lo_f32(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi_f32(uint32_t p) { return __uint_as_float(p & 0xFFFF0000u); }

// split 4 consecutive f32 into the three bf16 planes (8 bytes each)
__device__ __forceinline__ void split4(const float4 v, uint2 &hi, uint2 &mid, uint2 &lo) {
    hi.x = pack_bf16(v.x, v.y); hi.y = pack_bf16(v.z, v.w);
    const float r0 = v.x - bf16_lo_f32(hi.x), r1 = v.y - bf16_hi_f32(hi.x);
    const float r2 = v.z - bf16_lo_f32(hi.y), r3 = v.w - bf16_hi_f32(hi.y);
    mid.x = pack_bf16(r0, r1); mid.y = pack_bf16(r2, r3);
    const float s0 = r0 - bf16_lo_f32(mid.x), s1 = r1 - bf16_hi_f32(mid.x);
    const float s2 = r2 - bf16_lo_f32(mid.y), s3 = r3 - bf16_hi_f32(mid.y);
    lo.x = pack_bf16(s0, s1); lo.y = pack_bf16(s2, s3);
}

struct SplitArgs {
    const float *A; int lda;            // (M,K) f32
    const uint16_t *Wp; long long wplane;   // planes (3,N,K) bf16, plane stride in elements
    const float *bias;                  // (N) or null
    float *C; int ldc;                  // (M,N)
    float *pre; int ldpre;              // EPI_PRE: the value before the activation (A.W^T + bias)
    int M, N, K;
};

template <int NA, int NB>
struct Stage {          // the operands of one k-step in registers
    float4 a[NA];
    uint4 w0[NB], w1[NB], w2[NB];
};

template <int NA, int NB>
__device__ __forceinline__ void load_stage(Stage<NA, NB> &st, const float *const (&ap)[NA], const uint16_t *const (&wp)[NB], long long wplane,
                                           int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) st.a[i] = *reinterpret_cast<const float4 *>(ap[i] + k0);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        st.w0[i] = *reinterpret_cast<const uint4 *>(wp[i] + k0);
        st.w1[i] = *reinterpret_cast<const uint4 *>(wp[i] + wplane + k0);
        st.w2[i] = *reinterpret_cast<const uint4 *>(wp[i] + 2 * wplane + k0);
    }
}

template <int TM, int TN, int D, int EPI>
__global__ __launch_bounds__(256) void gemm_split_kernel(SplitArgs g) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int NA = 2 * TM;          // float4 loads of A per thread and k-step
    constexpr int NB = TN;              // 16-byte loads per W plane per thread and k-step
    __shared__ __attribute__((aligned(16))) uint16_t As[3][BM][LDR];
    __shared__ __attribute__((aligned(16))) uint16_t Ws[3][BN][LDR];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order (see dense.hip): a contiguous range of logical tiles per XCD
    const int nwg = gridDim.x * gridDim.y;
    int lin = blockIdx.y * gridDim.x + blockIdx.x;
    if ((nwg & 7) == 0) lin = (lin & 7) * (nwg >> 3) + (lin >> 3);
    const int bx = lin % gridDim.x, by = lin / gridDim.x;
    const int m0 = by * BM, n0 = bx * BN;
    const int M = g.M, N = g.N, K = g.K;

    // staging maps
    const float *ap[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int u = tid + 256 * i, row = min(m0 + (u >> 3), M - 1);     // rows past M are clamped, never stored
        ap[i] = g.A + (size_t)row * g.lda + (u & 7) * 4;
    }
    const uint16_t *wp[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int u = tid + 256 * i, row = min(n0 + (u >> 2), N - 1);
        wp[i] = g.Wp + (size_t)row * K + (u & 3) * 8;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int lr = lane & 31, lh = lane >> 5;
    const int nk = K / BK;

    // register ring: the operands of the next D k-steps are in flight while the current one is multiplied (a k-step is
    // only 12 TM TN MFMAs = 0.16 TM TN us, a global load ~1-2 us).  One named stage per ring slot keeps them in VGPRs.
    Stage<NA, NB> st0, st1, st2, st3;
    load_stage(st0, ap, wp, g.wplane, 0);
    if (D > 1 && 1 < nk) load_stage(st1, ap, wp, g.wplane, BK);
    if (D > 2 && 2 < nk) load_stage(st2, ap, wp, g.wplane, 2 * BK);
    if (D > 3 && 3 < nk) load_stage(st3, ap, wp, g.wplane, 3 * BK);

    auto k_step = [&](Stage<NA, NB> &st, int s) {
        if (s) __syncthreads();              // everyone finished reading the previous tile
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int u = tid + 256 * i, r = u >> 3, c = (u & 7) * 4;
            uint2 hi, mid, lo;
            split4(st.a[i], hi, mid, lo);
            *reinterpret_cast<uint2 *>(&As[0][r][c]) = hi;
            *reinterpret_cast<uint2 *>(&As[1][r][c]) = mid;
            *reinterpret_cast<uint2 *>(&As[2][r][c]) = lo;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int u = tid + 256 * i, r = u >> 2, c = (u & 3) * 8;
            *reinterpret_cast<uint4 *>(&Ws[0][r][c]) = st.w0[i];
            *reinterpret_cast<uint4 *>(&Ws[1][r][c]) = st.w1[i];
            *reinterpret_cast<uint4 *>(&Ws[2][r][c]) = st.w2[i];
        }
        __syncthreads();
        if (s + D < nk) load_stage(st, ap, wp, g.wplane, (s + D) * BK);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int kc = kb * 16 + lh * 8;
            bf16x8 a[TM][3], b[TN][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i][p] = *reinterpret_cast<const bf16x8 *>(&As[p][wm * 32 * TM + i * 32 + lr][kc]);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j][p] = *reinterpret_cast<const bf16x8 *>(&Ws[p][wn * 32 * TN + j * 32 + lr][kc]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];
                    // smallest terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
    };
    for (int s0 = 0; s0 < nk; s0 += D) {
        k_step(st0, s0);
        if (D > 1 && s0 + 1 < nk) k_step(st1, s0 + 1);
        if (D > 2 && s0 + 2 < nk) k_step(st2, s0 + 2);
        if (D > 3 && s0 + 3 < nk) k_step(st3, s0 + 3);
    }

    // epilogue: acc[i][j][r] = C[row][col], col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * 32 * TN + j * 32 + lr;
        const bool cok = col < N;
        const float bias = (EPI & EPI_BIAS) && cok ? g.bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = m0 + wm * 32 * TM + i * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (cok && row < M) {
                    float v = acc[i][j][r] + bias;
                    if (EPI & EPI_PRE) g.pre[(size_t)row * g.ldpre + col] = v;
                    if (EPI & EPI_GELU) v = gelu_val(v);
                    g.C[(size_t)row * g.ldc + col] = v;
                }
            }
        }
    }
}

// ---- variant without LDS: W pre-arranged in MFMA fragment order -------------------------------------------
// Fragment planes of the logical operand B (Nout, Kred):  Wf[nt][kb][p][lane][j] (bf16), nt = n / 32, kb = k / 16,
// lane = (n % 32) + 32 ((k % 16) / 8), j = k % 8 -- one wave-wide 16-byte load per MFMA operand, 1 KB contiguous.
// Every wave owns a (32 TM) x (32 TN) output tile and streams its A rows (f32, split in registers) and its W fragments
// straight from L2: no LDS, no barrier, so the waves of a CU drift apart and cover each other's load and split phases.
template <int TM, int TN>
struct FragStage {
    float4 a[TM][2];
    uint4 w[TN][3];
};

template <int TM, int TN, int D, int EPI>
__global__ __launch_bounds__(256) void gemm_frag_kernel(SplitArgs g) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nwg = gridDim.x * gridDim.y;
    int lin = blockIdx.y * gridDim.x + blockIdx.x;
    if ((nwg & 7) == 0) lin = (lin & 7) * (nwg >> 3) + (lin >> 3);
    const int bx = lin % gridDim.x, by = lin / gridDim.x;
    const int m0 = by * 64 * TM + wm * 32 * TM, n0 = bx * 64 * TN + wn * 32 * TN;
    const int M = g.M, N = g.N, K = g.K;
    const int lr = lane & 31, lh = lane >> 5;
    const int KB = K / 16, NT = (N + 31) / 32;

    const float *ap[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) ap[i] = g.A + (size_t)min(m0 + i * 32 + lr, M - 1) * g.lda + lh * 8;
    const uint16_t *wp[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) wp[j] = g.Wp + ((size_t)min(n0 / 32 + j, NT - 1) * KB * 3 * 64 + lane) * 8;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    auto load = [&](FragStage<TM, TN> &st, int kb) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            st.a[i][0] = *reinterpret_cast<const float4 *>(ap[i] + kb * 16);
            st.a[i][1] = *reinterpret_cast<const float4 *>(ap[i] + kb * 16 + 4);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) st.w[j][p] = *reinterpret_cast<const uint4 *>(wp[j] + ((size_t)kb * 3 + p) * 512);
    };
    auto step = [&](FragStage<TM, TN> &st, int kb) {
        bf16x8 a[TM][3], b[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            uint2 h0, m0_, l0, h1, m1, l1;
            if (g.wplane & 2) {     // probe: no split arithmetic
                h0 = make_uint2(__float_as_uint(st.a[i][0].x), __float_as_uint(st.a[i][0].y)); m0_ = h0; l0 = h0;
                h1 = make_uint2(__float_as_uint(st.a[i][1].x), __float_as_uint(st.a[i][1].y)); m1 = h1; l1 = h1;
            } else {
            split4(st.a[i][0], h0, m0_, l0);
            split4(st.a[i][1], h1, m1, l1);
            }
            a[i][0] = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
            a[i][1] = __builtin_bit_cast(bf16x8, make_uint4(m0_.x, m0_.y, m1.x, m1.y));
            a[i][2] = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[j][p] = __builtin_bit_cast(bf16x8, st.w[j][p]);
        if (kb + D < KB && !(g.wplane & 1)) load(st, kb + D);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                acc[i][j] = c;
            }
    };

    FragStage<TM, TN> st0, st1, st2, st3;
    load(st0, 0);
    if (D > 1 && 1 < KB) load(st1, 1);
    if (D > 2 && 2 < KB) load(st2, 2);
    if (D > 3 && 3 < KB) load(st3, 3);
    for (int kb = 0; kb < KB; kb += D) {
        step(st0, kb);
        if (D > 1 && kb + 1 < KB) step(st1, kb + 1);
        if (D > 2 && kb + 2 < KB) step(st2, kb + 2);
        if (D > 3 && kb + 3 < KB) step(st3, kb + 3);
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + j * 32 + lr;
        const bool cok = col < N;
        const float bias = (EPI & EPI_BIAS) && cok ? g.bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (cok && row < M) {
                    float v = acc[i][j][r] + bias;
                    if (EPI & EPI_PRE) g.pre[(size_t)row * g.ldpre + col] = v;
                    if (EPI & EPI_GELU) v = gelu_val(v);
                    g.C[(size_t)row * g.ldc + col] = v;
                }
            }
        }
    }
}

// logical operand B (Nout, Kred): B[n][k] = W[n * ld + k], or W[k * ld + n] when transpose -> fragment planes
__global__ __launch_bounds__(256) void split_frag_kernel(const float *__restrict__ W, int Nout, int Kred, int ld, int transpose,
                                                         uint16_t *__restrict__ out) {
    const int KB = Kred / 16, NT = (Nout + 31) / 32;
    const long long total = (long long)NT * KB * 512;            // one thread per (nt, kb, lane, j)
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long long blk = i >> 9;
    const int kb = (int)(blk % KB), nt = (int)(blk / KB);
    const int n = nt * 32 + (lane & 31), k = kb * 16 + (lane >> 5) * 8 + j;
    const float x = n < Nout ? (transpose ? W[(size_t)k * ld + n] : W[(size_t)n * ld + k]) : 0.0f;
    const uint32_t h = pack_bf16(x, 0.0f) & 0xFFFFu;
    const float r1 = x - __uint_as_float(h << 16);
    const uint32_t m = pack_bf16(r1, 0.0f) & 0xFFFFu;
    const float r2 = r1 - __uint_as_float(m << 16);
    const uint32_t l = pack_bf16(r2, 0.0f) & 0xFFFFu;
    const size_t base = ((size_t)blk * 3 * 64 + lane) * 8 + j;
    out[base] = (uint16_t)h;
    out[base + 512] = (uint16_t)m;
    out[base + 1024] = (uint16_t)l;
}

template <int TM, int TN, int D>
void launch_frag(const SplitArgs &g, int epi, hipStream_t st) {
    dim3 grid((g.N + 64 * TN - 1) / (64 * TN), (g.M + 64 * TM - 1) / (64 * TM));
    switch (epi) {
    case 0: hipLaunchKernelGGL((gemm_frag_kernel<TM, TN, D, 0>), grid, dim3(256), 0, st, g); break;
    case EPI_BIAS: hipLaunchKernelGGL((gemm_frag_kernel<TM, TN, D, EPI_BIAS>), grid, dim3(256), 0, st, g); break;
    case EPI_BIAS | EPI_GELU: hipLaunchKernelGGL((gemm_frag_kernel<TM, TN, D, EPI_BIAS | EPI_GELU>), grid, dim3(256), 0, st, g); break;
    default: hipLaunchKernelGGL((gemm_frag_kernel<TM, TN, D, EPI_BIAS | EPI_GELU | EPI_PRE>), grid, dim3(256), 0, st, g); break;
    }
}

// W (rows, cols) f32, row stride ld -> planes (3, rows, cols) bf16, or (3, cols, rows) when transpose
__global__ __launch_bounds__(256) void split_planes_kernel(const float *__restrict__ W, int rows, int cols, int ld, int transpose,
                                                           uint16_t *__restrict__ out) {
    const long long total = (long long)rows * cols;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    // i indexes the OUTPUT so that the plane stores are coalesced
    int r, c;
    if (transpose) { c = (int)(i / rows); r = (int)(i % rows); } else { r = (int)(i / cols); c = (int)(i % cols); }
    const float x = W[(size_t)r * ld + c];
    const uint32_t h = pack_bf16(x, 0.0f) & 0xFFFFu;
    const float r1 = x - __uint_as_float(h << 16);
    const uint32_t m = pack_bf16(r1, 0.0f) & 0xFFFFu;
    const float r2 = r1 - __uint_as_float(m << 16);
    const uint32_t l = pack_bf16(r2, 0.0f) & 0xFFFFu;
    out[i] = (uint16_t)h;
    out[total + i] = (uint16_t)m;
    out[2 * total + i] = (uint16_t)l;
}

template <int TM, int TN, int D>
void launch_split(const SplitArgs &g, int epi, hipStream_t st) {
    dim3 grid((g.N + 64 * TN - 1) / (64 * TN), (g.M + 64 * TM - 1) / (64 * TM));
    switch (epi) {
    case 0: hipLaunchKernelGGL((gemm_split_kernel<TM, TN, D, 0>), grid, dim3(256), 0, st, g); break;
    case EPI_BIAS: hipLaunchKernelGGL((gemm_split_kernel<TM, TN, D, EPI_BIAS>), grid, dim3(256), 0, st, g); break;
    case EPI_BIAS | EPI_GELU: hipLaunchKernelGGL((gemm_split_kernel<TM, TN, D, EPI_BIAS | EPI_GELU>), grid, dim3(256), 0, st, g); break;
    default: hipLaunchKernelGGL((gemm_split_kernel<TM, TN, D, EPI_BIAS | EPI_GELU | EPI_PRE>), grid, dim3(256), 0, st, g); break;
    }
}

int g_split_tile = 0;   // 0 = heuristic; 11 / 21 / 22 force a tile (tuning hook)

}  // namespace

extern "C" int upp_gemm_split_set_tile(int t) { g_split_tile = t; return 0; }

extern "C" int upp_split_bf16(const float *W, int rows, int cols, int ld, int transpose, uint16_t *planes, void *stream) {
    if (!W || !planes || rows < 1 || cols < 1 || ld < cols) return UPP_E_BADARG;
    const long long total = (long long)rows * cols;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, rows, cols, ld,
                       transpose, planes);
    return upp_launch_status();
}

extern "C" long long upp_split_frag_elems(int Nout, int Kred) {
    if (Nout < 1 || Kred < 16) return 0;
    return (long long)((Nout + 31) / 32) * (Kred / 16) * 3 * 512;
}

extern "C" int upp_split_bf16_frag(const float *W, int Nout, int Kred, int ld, int transpose, uint16_t *frag, void *stream) {
    if (!W || !frag || Nout < 1 || Kred < 16) return UPP_E_BADARG;
    if (Kred % 16 != 0) return UPP_E_RANGE;
    const long long total = (long long)((Nout + 31) / 32) * (Kred / 16) * 512;
    hipLaunchKernelGGL(split_frag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, Nout, Kred, ld,
                       transpose, frag);
    return upp_launch_status();
}

int g_frag_tile = 0;   // 0 = heuristic; TM*10+TN forces a tile (tuning hook)
extern "C" int upp_gemm_frag_set_tile(int t) { g_frag_tile = t; return 0; }

extern "C" int upp_linear_frag(const float *A, int lda, const uint16_t *Wf, const float *bias, float *C, int ldc, float *pre, int ldpre,
                               int M, int N, int K, int gelu, void *stream) {
    if (!A || !Wf || !C || M < 1 || N < 1 || K < 1) return UPP_E_BADARG;
    if (K % 16 != 0 || lda % 4 != 0 || lda < K || ldc < N || (gelu && !bias) || (pre && (!gelu || ldpre < N))) return UPP_E_RANGE;
    if (((uintptr_t)A | (uintptr_t)Wf) & 15) return UPP_E_RANGE;
    SplitArgs g{};
    g.A = A; g.lda = lda; g.Wp = Wf; g.wplane = g_frag_tile / 100; g.bias = bias; g.C = C; g.ldc = ldc; g.pre = pre; g.ldpre = ldpre;
    g.M = M; g.N = N; g.K = K;
    const int epi = (bias ? EPI_BIAS : 0) | (gelu ? EPI_GELU : 0) | (pre ? EPI_PRE : 0);
    hipStream_t st = (hipStream_t)stream;
    int tile = g_frag_tile % 100;
    if (!tile) {
        const long long t11 = (long long)((M + 63) / 64) * ((N + 63) / 64);
        tile = t11 >= 8 * 256 ? 22 : (t11 >= 2 * 256 ? 12 : 11);
    }
    if (tile == 22) launch_frag<2, 2, 2>(g, epi, st);
    else if (tile == 24) launch_frag<2, 4, 2>(g, epi, st);
    else if (tile == 21) launch_frag<2, 1, 3>(g, epi, st);
    else if (tile == 12) launch_frag<1, 2, 3>(g, epi, st);
    else if (tile == 14) launch_frag<1, 4, 2>(g, epi, st);
    else launch_frag<1, 1, 4>(g, epi, st);
    return upp_launch_status();
}

extern "C" int upp_linear_split(const float *A, int lda, const uint16_t *Wp, const float *bias, float *C, int ldc, float *pre, int ldpre,
                                int M, int N, int K, int gelu, void *stream) {
    if (!A || !Wp || !C || M < 1 || N < 1 || K < 1) return UPP_E_BADARG;
    if (K % BK != 0 || lda % 4 != 0 || lda < K || ldc < N || (gelu && !bias) || (pre && (!gelu || ldpre < N))) return UPP_E_RANGE;
    if (((uintptr_t)A | (uintptr_t)Wp) & 15) return UPP_E_RANGE;
    SplitArgs g{};
    g.A = A; g.lda = lda; g.Wp = Wp; g.wplane = (long long)N * K; g.bias = bias; g.C = C; g.ldc = ldc; g.pre = pre; g.ldpre = ldpre;
    g.M = M; g.N = N; g.K = K;
    const int epi = (bias ? EPI_BIAS : 0) | (gelu ? EPI_GELU : 0) | (pre ? EPI_PRE : 0);
    hipStream_t st = (hipStream_t)stream;
    int tile = g_split_tile;
    if (!tile) {
        // the largest tile that still gives every CU two workgroups
        const long long t11 = (long long)((M + 63) / 64) * ((N + 63) / 64);
        tile = t11 >= 4 * 512 ? 22 : (t11 >= 2 * 512 ? 21 : 11);
    }
    if (tile == 22) launch_split<2, 2, 2>(g, epi, st);
    else if (tile == 21) launch_split<2, 1, 3>(g, epi, st);
    else if (tile == 12) launch_split<1, 2, 3>(g, epi, st);
    else if (tile == 14) launch_split<1, 4, 2>(g, epi, st);
    else launch_split<1, 1, 4>(g, epi, st);
    return upp_launch_status();
}
