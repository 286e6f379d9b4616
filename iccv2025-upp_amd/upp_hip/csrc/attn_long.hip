// attn_long.hip -- attention core on FP32 MFMA for sequences of 97..160 tokens, head dim 64 (the 128-group models:
// Point_MAE_unify_seg runs L = 129 / 139).  Same contract as attn_mfma.hip (L <= 96) and the VALU kernels in block.hip;
// see include/upp_hip.h upp_attn_fwd / upp_attn_bwd and reference models/Point_MAE_pretask_dev.py:186-193.
//
// A 160 x 160 score matrix plus Q, K, V (and dO) does not fit the 160 KB of LDS, so one workgroup (8 waves) per
// (sample, head) keeps K and V resident (2 x 160 x 65 floats) and walks the queries in blocks of 64 rows:
//   forward : per block  S = Q_b K^T (<= 10 tiles) -> row softmax in LDS (lse out) -> O_b = P V (<= 4 tiles)
//   backward: blocks of 32 rows: S, dP = dO_b V^T (one key tile per wave) -> P and dS strips in LDS ->
//             dV += P^T dO_b, dK += dS^T Q_b (accumulator tiles, <= 4 per wave, in the registers of waves 0..5
//             across all blocks, written once at the end) while waves 6, 7 compute dQ_b = dS K.
//             No atomics: results are deterministic.
// Every product is a set of 32x32 tiles of v_mfma_f32_32x32x2_f32 dealt round-robin to the waves (attn_tiles.h).
#include "attn_tiles.h"

namespace {

constexpr int kKP = 160;         // padded key length
constexpr int kSS = 161;         // row stride of the (64 x 160) score strip (161 % 32 == 1: conflict-free transposed reads)
constexpr int kQB = 64;          // query rows per block
constexpr int kLW = 8;           // waves per workgroup (two per SIMD)
constexpr int kBB = 32;          // backward: query rows per block (P and dS strips both in LDS)
constexpr int kAccW = 6;         // backward: waves 0..5 own the dK / dV accumulator tiles, waves 6, 7 compute dQ
constexpr int kAccS = 4;         // accumulator tiles per owner wave: ceil(4 * 5 / 6)
constexpr int kTR = 16;          // FOLD: rows of the tail strips (the queries behind row 128 of a 129 ... 144-token sequence)

// rows [0, valid) x 64 of NARR sources (row stride rs floats) -> dst[a][R][kLD], rows >= valid zero; all loads first
template <int R, int NARR>
__device__ __forceinline__ void stage_block(float *const (&dst)[NARR], const float *const (&src)[NARR], const size_t (&rs)[NARR], int valid) {
    constexpr int IT = (R * 16 + 64 * kLW - 1) / (64 * kLW);
    float4 v[NARR][IT];
#pragma unroll
    for (int a = 0; a < NARR; ++a)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = threadIdx.x + it * 64 * kLW;
            const int r = i >> 4, c = (i & 15) * 4;
            v[a][it] = r < valid ? *reinterpret_cast<const float4 *>(src[a] + (size_t)r * rs[a] + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
    for (int a = 0; a < NARR; ++a)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = threadIdx.x + it * 64 * kLW;
            if (i < R * 16) {
                float *d = dst[a] + (i >> 4) * kLD + (i & 15) * 4;
                d[0] = v[a][it].x; d[1] = v[a][it].y; d[2] = v[a][it].z; d[3] = v[a][it].w;
            }
        }
}

// two row blocks of different heights in one pass (all loads first): rows [0, valid_a) x 64 -> dst[0][RA][kLD], [0, valid_b) -> dst[1][RB][kLD]
template <int RA, int RB>
__device__ __forceinline__ void stage_two(float *const (&dst)[2], const float *const (&src)[2], const size_t (&rs)[2], int valid_a, int valid_b) {
    constexpr int N = (RA + RB) * 16, IT = (N + 64 * kLW - 1) / (64 * kLW);
    float4 v[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int i = threadIdx.x + it * 64 * kLW;
        const int a = i >= RA * 16, ii = a ? i - RA * 16 : i;
        const int r = ii >> 4, c = (ii & 15) * 4;
        const bool ok = i < N && r < (a ? valid_b : valid_a);
        v[it] = ok ? *reinterpret_cast<const float4 *>(src[a] + (size_t)r * rs[a] + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int i = threadIdx.x + it * 64 * kLW;
        if (i < N) {
            const int a = i >= RA * 16, ii = a ? i - RA * 16 : i;
            float *d = dst[a] + (ii >> 4) * kLD + (ii & 15) * 4;
            d[0] = v[it].x; d[1] = v[it].y; d[2] = v[it].z; d[3] = v[it].w;
        }
    }
}

// FOLD (round 5): sequences of 129 ... 144 tokens (the segmentation model: cls + 128 groups, + 10 prompts) used to walk THREE 64-row
// query blocks, the third for 1 or 11 rows (32.8 us at L = 129 against 22.7 at L = 128).  With FOLD the 1 ... 16 rows behind row 128 are a
// second, 16-row query strip (Qt / St) that rides along with the FIRST block in the slots its tile lists leave idle: S has 10 + 5 tiles
// for 2 rounds of 8 waves, P V 4 + 2 tiles for one.  Rows >= 16 of the strips are never written; the 32-row operand reads of the tail
// tiles run on into the arrays behind them (in bounds; such a row only feeds the output row of the same index, which is not stored).
template <bool FOLD>
__global__ __launch_bounds__(64 * kLW) void attn_fwd_long_kernel(const float *__restrict__ qkv, float *__restrict__ ctx,
                                                                 float *__restrict__ lse, int L, int H, float scale) {
    extern __shared__ float sm[];
    float *Ks = sm, *Vs = Ks + kKP * kLD, *Qb = Vs + kKP * kLD, *Qt = Qb + kQB * kLD, *St = Qt + (FOLD ? kTR * kLD : 0), *Ss = St + (FOLD ? kTR * kSS : 0);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 31, lk = lane >> 5;
    const int b = blockIdx.x / H, hh = blockIdx.x - b * H;
    const size_t rs = (size_t)3 * H * 64;
    const float *base = qkv + (size_t)b * L * rs + (size_t)hh * 64;
    {
        float *const dst[2] = {Ks, Vs};
        const float *const src[2] = {base + H * 64, base + 2 * H * 64};
        const size_t strides[2] = {rs, rs};
        stage_block<kKP, 2>(dst, src, strides, L);
    }
    const int nt = (L + 31) / 32;                       // key tiles
    const int Lmain = FOLD ? 2 * kQB : L, tail = FOLD ? L - 2 * kQB : 0;       // FOLD: 128 < L <= 128 + kTR (checked by the host)
    // row softmax of `rows` strip rows (the first `valid` of them are queries q_base + i), lane = key (3 slots cover the 160 columns),
    // four rows per wave and iteration
    auto softmax_rows = [&](float *strip, int rows, int valid, int q_base, int w) {
        for (int i0 = w * 4; i0 < rows; i0 += 4 * kLW) {
            float s[4][3], mx[4], e[4][3], sum[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float *row = strip + min(i0 + q, rows - 1) * kSS;
#pragma unroll
                for (int c = 0; c < 3; ++c) s[q][c] = lane + 64 * c < L ? row[lane + 64 * c] : -__builtin_inff();
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) mx[q] = wave_max_f32(fmaxf(fmaxf(s[q][0], s[q][1]), s[q][2]));
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 3; ++c) e[q][c] = lane + 64 * c < L ? exp_neg(s[q][c] - mx[q]) : 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) sum[q] = wave_sum_f32((e[q][0] + e[q][1]) + e[q][2]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q;
                if (i < rows) {
                    float *row = strip + i * kSS;
                    const float inv = i < valid ? 1.0f / sum[q] : 0.0f;          // padded query rows contribute nothing
#pragma unroll
                    for (int c = 0; c < 3; ++c) if (lane + 64 * c < kKP) row[lane + 64 * c] = e[q][c] * inv;
                    if (lane == 0 && i < valid) lse[((size_t)b * H + hh) * L + q_base + i] = mx[q] + logf(sum[q]);
                }
            }
        }
    };
    for (int q0 = 0; q0 < Lmain; q0 += kQB) {
        const int qv = min(kQB, Lmain - q0), qt = (qv + 31) / 32;
        const bool with_tail = FOLD && q0 == 0;
        if (with_tail) {
            float *const dst[2] = {Qb, Qt};
            const float *const src[2] = {base, base + (size_t)Lmain * rs};
            const size_t strides[2] = {rs, rs};
            stage_two<kQB, kTR>(dst, src, strides, qv, tail);
        } else {
            float *const dst[1] = {Qb};
            const float *const src[1] = {base + (size_t)q0 * rs};
            const size_t strides[1] = {rs};
            stage_block<kQB, 1>(dst, src, strides, qv);
        }
        __syncthreads();
        const int n_s = qt * nt + (with_tail ? nt : 0);
        for (int t = wave; t < n_s; t += kLW) {         // S = Q_b K^T (and S_tail = Q_tail K^T), scaled
            f32x16 acc; zero(acc);
            if (t < qt * nt) {
                const int it = t / nt, jt = t - it * nt;
                mfma_tile<false, true>(acc, Qb + it * 32 * kLD, kLD, Ks + jt * 32 * kLD, kLD, 64, lr, lk);
#pragma unroll
                for (int r = 0; r < 16; ++r) Ss[(it * 32 + tile_row(r, lk)) * kSS + jt * 32 + lr] = acc[r] * scale;
            } else {
                const int jt = t - qt * nt;
                mfma_tile<false, true>(acc, Qt, kLD, Ks + jt * 32 * kLD, kLD, 64, lr, lk);
#pragma unroll
                for (int r = 0; r < 16; ++r) if (tile_row(r, lk) < kTR) St[tile_row(r, lk) * kSS + jt * 32 + lr] = acc[r] * scale;
            }
        }
        __syncthreads();
        softmax_rows(Ss, qt * 32, qv, q0, wave);
        if (with_tail) softmax_rows(St, kTR, tail, Lmain, kLW - 1 - wave);       // (the last waves first: they are the ones with a short first loop)
        __syncthreads();
        const int n_o = qt * 2 + (with_tail ? 2 : 0);
        for (int t = wave; t < n_o; t += kLW) {         // O_b = P V
            const bool is_tail = t >= qt * 2;
            const int u = is_tail ? t - qt * 2 : t;
            const int it = u >> 1, dt = u & 1;
            f32x16 acc; zero(acc);
            mfma_tile<false, false>(acc, is_tail ? St : Ss + it * 32 * kSS, kSS, Vs + dt * 32, kLD, nt * 32, lr, lk);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (is_tail ? Lmain : q0 + it * 32) + tile_row(r, lk);
                if (i < L && (!is_tail || tile_row(r, lk) < kTR)) ctx[((size_t)b * L + i) * (H * 64) + hh * 64 + dt * 32 + lr] = acc[r];
            }
        }
        __syncthreads();
    }
}

// (round 5: folding the rows behind row 128 into the first 32-row block, as the forward kernel does, was built and measured for this
//  kernel too -- 64.1 us against 63.7 at L = 129: the fifth KEY tile (one key of 32) costs what the fifth query block cost, in every
//  block's S / dP phase, accumulator round and dQ contraction; not kept.)
__global__ __launch_bounds__(64 * kLW) void attn_bwd_long_kernel(const float *__restrict__ qkv, const float *__restrict__ ctx,
                                                                 const float *__restrict__ d_ctx, const float *__restrict__ lse,
                                                                 float *__restrict__ d_qkv, int L, int H, float scale) {
    extern __shared__ float sm[];
    float *Ks = sm, *Vs = Ks + kKP * kLD, *Qb = Vs + kKP * kLD, *Gb = Qb + kBB * kLD, *Ps = Gb + kBB * kLD, *Ds = Ps + kBB * kSS;
    float *delta = Ds + kBB * kSS, *lses = delta + kBB;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 31, lk = lane >> 5;
    const int b = blockIdx.x / H, hh = blockIdx.x - b * H;
    const size_t rs = (size_t)3 * H * 64, cs = (size_t)H * 64;
    const float *base = qkv + (size_t)b * L * rs + (size_t)hh * 64;
    float *dbase = d_qkv + (size_t)b * L * rs + (size_t)hh * 64;
    const float *gbase = d_ctx + (size_t)b * L * cs + (size_t)hh * 64;
    const float *obase = ctx + (size_t)b * L * cs + (size_t)hh * 64;
    {
        float *const dst[2] = {Ks, Vs};
        const float *const src[2] = {base + H * 64, base + 2 * H * 64};
        const size_t strides[2] = {rs, rs};
        stage_block<kKP, 2>(dst, src, strides, L);
    }
    const int nt = (L + 31) / 32;
    // dV / dK accumulator tiles: index a = which * 2 nt + jt * 2 + dt, owned by wave a % kAccW, slot a / kAccW
    f32x16 acc[kAccS];
#pragma unroll
    for (int m = 0; m < kAccS; ++m) zero(acc[m]);
    float *const Ks0 = Ks, *const Vs0 = Vs, *const Qb0 = Qb, *const Gb0 = Gb, *const Ps0 = Ps, *const Ds0 = Ds;
    for (int q0 = 0; q0 < L; q0 += kBB) {
        const int qv = min(kBB, L - q0);
        // The operand addresses of all tile products are loop-invariant; hoisted out of this loop they would occupy
        // some 300 VGPRs (and spill).  An opaque zero offset keeps their computation inside the iteration.
        int opq = 0;
        asm volatile("" : "+v"(opq));
        Ks = Ks0 + opq; Vs = Vs0 + opq; Qb = Qb0 + opq; Gb = Gb0 + opq; Ps = Ps0 + opq; Ds = Ds0 + opq;
        {
            float *const dst[2] = {Qb, Gb};
            const float *const src[2] = {base + (size_t)q0 * rs, gbase + (size_t)q0 * cs};
            const size_t strides[2] = {rs, cs};
            stage_block<kBB, 2>(dst, src, strides, qv);
        }
        {   // delta_i = dO_i . O_i ; lse_i  -- 4 rows per wave, all loads first
            constexpr int RW = kBB / kLW;
            float g[RW], o[RW], ls[RW], d[RW];
#pragma unroll
            for (int t = 0; t < RW; ++t) {
                const int i = wave + kLW * t;
                const bool ok = i < qv;
                g[t] = ok ? gbase[(size_t)(q0 + i) * cs + lane] : 0.0f;
                o[t] = ok ? obase[(size_t)(q0 + i) * cs + lane] : 0.0f;
                ls[t] = (ok && lane == 0) ? lse[((size_t)b * H + hh) * L + q0 + i] : 0.0f;
            }
#pragma unroll
            for (int t = 0; t < RW; ++t) d[t] = wave_sum_f32(g[t] * o[t]);
#pragma unroll
            for (int t = 0; t < RW; ++t)
                if (lane == 0) { delta[wave + kLW * t] = d[t]; lses[wave + kLW * t] = ls[t]; }
        }
        __syncthreads();
        // phase 1: wave jt computes S and dP of key tile jt, writes P and dS = P (dP - delta) scale to the two strips
        if (wave < nt) {
            const int jt = wave;
            f32x16 s, dp; zero(s); zero(dp);
            mfma_tile<false, true>(s, Qb, kLD, Ks + jt * 32 * kLD, kLD, 64, lr, lk);
            mfma_tile<false, true>(dp, Gb, kLD, Vs + jt * 32 * kLD, kLD, 64, lr, lk);
            const int j = jt * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = tile_row(r, lk);
                const float p = (i < qv && j < L) ? exp_neg(s[r] * scale - lses[i]) : 0.0f;
                Ps[i * kSS + j] = p;
                Ds[i * kSS + j] = (i < qv && j < L) ? p * (dp[r] - delta[i]) * scale : 0.0f;
            }
        }
        __syncthreads();
        // phase 2: dV += P^T dO_b, dK += dS^T Q_b on the owner waves; dQ_b = dS K on the last two waves
        if (wave < kAccW) {
#pragma unroll
            for (int m = 0; m < kAccS; ++m) {
                const int a = wave + kAccW * m;
                if (a < nt * 4) {
                    const int which = a / (nt * 2), u = a - which * nt * 2;
                    const int jt = u >> 1, dt = u & 1;
                    if (which == 0) mfma_tile_k<true, false, 32>(acc[m], Ps + jt * 32, kSS, Gb + dt * 32, kLD, lr, lk);
                    else mfma_tile_k<true, false, 32>(acc[m], Ds + jt * 32, kSS, Qb + dt * 32, kLD, lr, lk);
                }
            }
        } else {
            const int dt = wave - kAccW;
            f32x16 dq; zero(dq);
            mfma_tile<false, false>(dq, Ds, kSS, Ks + dt * 32, kLD, nt * 32, lr, lk);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = q0 + tile_row(r, lk);
                if (i < L) dbase[(size_t)i * rs + dt * 32 + lr] = dq[r];
            }
        }
        __syncthreads();
    }
    if (wave < kAccW) {
#pragma unroll
        for (int m = 0; m < kAccS; ++m) {
            const int a = wave + kAccW * m;
            if (a < nt * 4) {
                const int which = a / (nt * 2), u = a - which * nt * 2;
                const int jt = u >> 1, dt = u & 1;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = jt * 32 + tile_row(r, lk);
                    if (j < L) dbase[(size_t)j * rs + (which == 0 ? 2 : 1) * H * 64 + dt * 32 + lr] = acc[m][r];
                }
            }
        }
    }
}

constexpr size_t kFwdLongLds = ((size_t)2 * kKP * kLD + (size_t)kQB * kLD + (size_t)kQB * kSS) * sizeof(float);
constexpr size_t kFwdFoldLds = kFwdLongLds + ((size_t)kTR * kLD + (size_t)kTR * kSS) * sizeof(float);
static_assert(kFwdFoldLds <= 160 * 1024, "LDS budget");
constexpr size_t kBwdLongLds = ((size_t)2 * kKP * kLD + (size_t)2 * kBB * kLD + (size_t)2 * kBB * kSS + 2 * kBB) * sizeof(float);
static_assert(kBwdLongLds <= 160 * 1024, "LDS budget");

// UPP_ATTN_FOLD=0 (read once): the three-block walk of rounds 1-4 for 129 ... 144 tokens (A/B timing, tests)
inline bool fold_disabled() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("UPP_ATTN_FOLD"); v = (e && e[0] == '0') ? 1 : 0; }
    return v == 1;
}

}  // namespace

// called by upp_attn_fwd / upp_attn_bwd (block.hip) for 96 < L <= 160
int upp_attn_fwd_long(const float *qkv, float *ctx, float *lse, int B, int L, int H, float scale, hipStream_t st) {
    static std::atomic<bool> raised{false};
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_fwd_long_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLongLds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_fwd_long_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdFoldLds);
        if (e != hipSuccess) return (int)e;
        raised = true;
    }
    if (L > 2 * kQB && L <= 2 * kQB + kTR && !fold_disabled())
        hipLaunchKernelGGL(attn_fwd_long_kernel<true>, dim3(B * H), dim3(64 * kLW), kFwdFoldLds, st, qkv, ctx, lse, L, H, scale);
    else
        hipLaunchKernelGGL(attn_fwd_long_kernel<false>, dim3(B * H), dim3(64 * kLW), kFwdLongLds, st, qkv, ctx, lse, L, H, scale);
    return upp_launch_status();
}

int upp_attn_bwd_long(const float *qkv, const float *ctx, const float *d_ctx, const float *lse, float *d_qkv, int B, int L, int H,
                      float scale, hipStream_t st) {
    static std::atomic<bool> raised{false};
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_long_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdLongLds);
        if (e != hipSuccess) return (int)e;
        raised = true;
    }
    hipLaunchKernelGGL(attn_bwd_long_kernel, dim3(B * H), dim3(64 * kLW), kBwdLongLds, st, qkv, ctx, d_ctx, lse, d_qkv, L, H, scale);
    return upp_launch_status();
}
