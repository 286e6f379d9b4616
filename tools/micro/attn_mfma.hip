// attn_mfma.hip -- attention core (forward and backward) on FP32 MFMA for sequences up to 96 tokens, head dim 64.
// Same contract as attn_fwd / attn_bwd in block.hip (which remain the general path for longer sequences); see
// include/upp_hip.h upp_attn_fwd / upp_attn_bwd and reference models/Point_MAE_pretask_dev.py:186-193.
//
// One workgroup (8 waves) per (sample, head).  Q, K, V (and dO) are staged once in LDS (rows padded to 65 floats so
// that both the row-major and the transposed one-dword-per-lane MFMA operand reads are bank-conflict free), every
// product is a set of 32x32 output tiles of v_mfma_f32_32x32x2_f32 dealt round-robin to the waves:
//   forward : S = Q K^T (9 tiles) -> row softmax in LDS -> O = P V (6 tiles)
//   backward: S and dP = dO V^T (9+9 tiles) -> P = exp(S*scale - lse), dS = P*(dP - delta)*scale in registers ->
//             dV = P^T dO, dK = dS^T Q, dQ = dS K (6 tiles each); P and dS share one 96x97 LDS buffer.
// The (L x L) score matrix never leaves the CU.  delta_i = dO_i . O_i (the usual rewrite of sum_j P_ij dP_ij).
#include "attn_tiles.h"

namespace {

constexpr int kLP = 96;          // padded sequence length
constexpr int kLS = 97;          // row stride of the (L x L) score buffer
constexpr int kNW = 8;           // waves per workgroup (backward): two per SIMD -- a lone wave issues MFMAs at about half the pipe rate
constexpr int kFW = 16;          // forward: 80 VGPRs leave room for four waves per SIMD

// Stage rows [0, L) x 64 of NARR sources (row stride rs floats each) into dst[a][kLP][kLD], zero rows >= L.
// A dependent global load costs ~1 us here, so ALL loads of all arrays are issued before the first LDS write
// (6 float4 per thread and array) instead of load -> wait -> write per iteration.
template <int NARR, int NW>
__device__ __forceinline__ void stage_rows(float *const (&dst)[NARR], const float *const (&src)[NARR], const size_t (&rs)[NARR], int L) {
    constexpr int IT = (kLP * 16 + 64 * NW - 1) / (64 * NW);
    float4 v[NARR][IT];
#pragma unroll
    for (int a = 0; a < NARR; ++a)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = threadIdx.x + it * 64 * NW;
            const int r = i >> 4, c = (i & 15) * 4;
            v[a][it] = r < L ? *reinterpret_cast<const float4 *>(src[a] + (size_t)r * rs[a] + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
    for (int a = 0; a < NARR; ++a)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = threadIdx.x + it * 64 * NW;
            if (i < kLP * 16) {
                float *d = dst[a] + (i >> 4) * kLD + (i & 15) * 4;
                d[0] = v[a][it].x; d[1] = v[a][it].y; d[2] = v[a][it].z; d[3] = v[a][it].w;
            }
        }
}

__global__ __launch_bounds__(64 * kFW) void attn_fwd_mfma_kernel(const float *__restrict__ qkv, float *__restrict__ ctx,
                                                            float *__restrict__ lse, int L, int H, float scale) {
    extern __shared__ float sm[];
    float *Qs = sm, *Ks = Qs + kLP * kLD, *Vs = Ks + kLP * kLD, *Ss = Vs + kLP * kLD;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 31, lk = lane >> 5;
    const int b = blockIdx.x / H, hh = blockIdx.x - b * H;
    const size_t rs = (size_t)3 * H * 64;
    const float *base = qkv + (size_t)b * L * rs + (size_t)hh * 64;
    {
        float *const dst[3] = {Qs, Ks, Vs};
        const float *const src[3] = {base, base + H * 64, base + 2 * H * 64};
        const size_t strides[3] = {rs, rs, rs};
        stage_rows<3, kFW>(dst, src, strides, L);
    }
    __syncthreads();
    const int nt = (L + 31) / 32;                       // tiles along the sequence
    for (int t = wave; t < nt * nt; t += kFW) {         // S = Q K^T, scaled
        const int it = t / nt, jt = t - it * nt;
        f32x16 acc; zero(acc);
        mfma_tile<false, true>(acc, Qs + it * 32 * kLD, kLD, Ks + jt * 32 * kLD, kLD, 64, lr, lk);
#pragma unroll
        for (int r = 0; r < 16; ++r) Ss[(it * 32 + tile_row(r, lk)) * kLS + jt * 32 + lr] = acc[r] * scale;
    }
    __syncthreads();
    // row softmax, lane = key (2 slots cover the 96 columns).  A wave takes FOUR rows per iteration: the reductions
    // are latency chains (DPP + readlane), four independent ones interleave in the pipeline.
    for (int i0 = wave * 4; i0 < L; i0 += 4 * kFW) {
        float s0[4], s1[4], mx[4], e0[4], e1[4], sum[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float *row = Ss + min(i0 + q, L - 1) * kLS;
            s0[q] = lane < L ? row[lane] : -__builtin_inff();
            s1[q] = lane + 64 < L ? row[lane + 64] : -__builtin_inff();
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) mx[q] = wave_max_f32(fmaxf(s0[q], s1[q]));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            e0[q] = lane < L ? exp_neg(s0[q] - mx[q]) : 0.0f;
            e1[q] = lane + 64 < L ? exp_neg(s1[q] - mx[q]) : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) sum[q] = wave_sum_f32(e0[q] + e1[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q;
            if (i < L) {
                float *row = Ss + i * kLS;
                const float inv = 1.0f / sum[q];
                row[lane] = e0[q] * inv;
                if (lane + 64 < kLP) row[lane + 64] = e1[q] * inv;
                if (lane == 0) lse[((size_t)b * H + hh) * L + i] = mx[q] + logf(sum[q]);
            }
        }
    }
    for (int i = L + wave; i < nt * 32; i += kFW) {     // padded query rows contribute nothing
        Ss[i * kLS + lane] = 0.0f;
        if (lane + 64 < kLP) Ss[i * kLS + lane + 64] = 0.0f;
    }
    __syncthreads();
    for (int t = wave; t < nt * 2; t += kFW) {          // O = P V
        const int it = t >> 1, dt = t & 1;
        f32x16 acc; zero(acc);
        mfma_tile<false, false>(acc, Ss + it * 32 * kLS, kLS, Vs + dt * 32, kLD, nt * 32, lr, lk);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = it * 32 + tile_row(r, lk);
            if (i < L) ctx[((size_t)b * L + i) * (H * 64) + hh * 64 + dt * 32 + lr] = acc[r];
        }
    }
}

__global__ __launch_bounds__(64 * kNW) void attn_bwd_mfma_kernel(const float *__restrict__ qkv, const float *__restrict__ ctx,
                                                            const float *__restrict__ d_ctx, const float *__restrict__ lse,
                                                            float *__restrict__ d_qkv, int L, int H, float scale) {
    extern __shared__ float sm[];
    float *Qs = sm, *Ks = Qs + kLP * kLD, *Vs = Ks + kLP * kLD, *Gs = Vs + kLP * kLD, *Ss = Gs + kLP * kLD;
    float *delta = Ss + kLP * kLS, *lses = delta + kLP;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 31, lk = lane >> 5;
    const int b = blockIdx.x / H, hh = blockIdx.x - b * H;
    const size_t rs = (size_t)3 * H * 64, cs = (size_t)H * 64;
    const float *base = qkv + (size_t)b * L * rs + (size_t)hh * 64;
    float *dbase = d_qkv + (size_t)b * L * rs + (size_t)hh * 64;
    const float *gbase = d_ctx + (size_t)b * L * cs + (size_t)hh * 64;
    const float *obase = ctx + (size_t)b * L * cs + (size_t)hh * 64;
    {
        float *const dst[4] = {Qs, Ks, Vs, Gs};
        const float *const src[4] = {base, base + H * 64, base + 2 * H * 64, gbase};
        const size_t strides[4] = {rs, rs, rs, cs};
        stage_rows<4, kNW>(dst, src, strides, L);
    }
    {   // delta_i = dO_i . O_i ; lse_i  -- kLP / kNW rows per wave, all loads first
        constexpr int RW = kLP / kNW;
        float g[RW], o[RW], ls[RW];
#pragma unroll
        for (int t = 0; t < RW; ++t) {
            const int i = wave + kNW * t;
            const bool ok = i < L;
            g[t] = ok ? gbase[(size_t)i * cs + lane] : 0.0f;
            o[t] = ok ? obase[(size_t)i * cs + lane] : 0.0f;
            ls[t] = (ok && lane == 0) ? lse[((size_t)b * H + hh) * L + i] : 0.0f;
        }
#pragma unroll
        for (int t0 = 0; t0 < RW; t0 += 4) {            // four independent reduction chains at a time
            float d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) d[q] = wave_sum_f32(g[t0 + q] * o[t0 + q]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (lane == 0) { delta[wave + kNW * (t0 + q)] = d[q]; lses[wave + kNW * (t0 + q)] = ls[t0 + q]; }
        }
    }
    __syncthreads();
    const int nt = (L + 31) / 32;
    // phase 1: P and dS tiles in registers (a wave owns at most 2 of the <= 9 tiles)
    f32x16 pt[2], dst[2];
    int nown = 0;
    for (int t = wave; t < nt * nt; t += kNW, ++nown) {
        const int it = t / nt, jt = t - it * nt;
        f32x16 s, dp; zero(s); zero(dp);
        mfma_tile<false, true>(s, Qs + it * 32 * kLD, kLD, Ks + jt * 32 * kLD, kLD, 64, lr, lk);
        mfma_tile<false, true>(dp, Gs + it * 32 * kLD, kLD, Vs + jt * 32 * kLD, kLD, 64, lr, lk);
        const int j = jt * 32 + lr;
        float lr_[16], dr_[16];                           // the 16 rows' lse / delta: one batch of LDS reads (rows < kLP always)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const int i = it * 32 + tile_row(r, lk); lr_[r] = lses[i]; dr_[r] = delta[i]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = it * 32 + tile_row(r, lk);
            const float p = (i < L && j < L) ? exp_neg(s[r] * scale - lr_[r]) : 0.0f;
            s[r] = p;
            dp[r] = p * (dp[r] - dr_[r]) * scale;
        }
        // static register indexing of the per-wave tile store
        if (nown == 0) { pt[0] = s; dst[0] = dp; } else { pt[1] = s; dst[1] = dp; }
    }
    // phase 2: P -> LDS, dV = P^T dO
    nown = 0;
    for (int t = wave; t < nt * nt; t += kNW, ++nown) {
        const int it = t / nt, jt = t - it * nt;
        const f32x16 v = nown == 0 ? pt[0] : pt[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) Ss[(it * 32 + tile_row(r, lk)) * kLS + jt * 32 + lr] = v[r];
    }
    __syncthreads();
    for (int t = wave; t < nt * 2; t += kNW) {
        const int jt = t >> 1, dt = t & 1;
        f32x16 acc; zero(acc);
        mfma_tile<true, false>(acc, Ss + jt * 32, kLS, Gs + dt * 32, kLD, nt * 32, lr, lk);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = jt * 32 + tile_row(r, lk);
            if (j < L) dbase[(size_t)j * rs + 2 * H * 64 + dt * 32 + lr] = acc[r];
        }
    }
    __syncthreads();
    // phase 3: dS -> LDS, dK = dS^T Q, dQ = dS K
    nown = 0;
    for (int t = wave; t < nt * nt; t += kNW, ++nown) {
        const int it = t / nt, jt = t - it * nt;
        const f32x16 v = nown == 0 ? dst[0] : dst[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) Ss[(it * 32 + tile_row(r, lk)) * kLS + jt * 32 + lr] = v[r];
    }
    __syncthreads();
    for (int t = wave; t < nt * 4; t += kNW) {
        const int which = t / (nt * 2), u = t - which * nt * 2;   // 0: dK tiles, 1: dQ tiles
        const int rt = u >> 1, dt = u & 1;
        f32x16 acc; zero(acc);
        if (which == 0) mfma_tile<true, false>(acc, Ss + rt * 32, kLS, Qs + dt * 32, kLD, nt * 32, lr, lk);
        else mfma_tile<false, false>(acc, Ss + rt * 32 * kLS, kLS, Ks + dt * 32, kLD, nt * 32, lr, lk);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rt * 32 + tile_row(r, lk);
            if (row < L) dbase[(size_t)row * rs + (which == 0 ? H * 64 : 0) + dt * 32 + lr] = acc[r];
        }
    }
}

constexpr size_t kFwdLds = ((size_t)3 * kLP * kLD + (size_t)kLP * kLS) * sizeof(float);
constexpr size_t kBwdLds = ((size_t)4 * kLP * kLD + (size_t)kLP * kLS + 2 * kLP) * sizeof(float);

}  // namespace

// called by upp_attn_fwd / upp_attn_bwd (block.hip) for L <= 96
int upp_attn_fwd_mfma(const float *qkv, float *ctx, float *lse, int B, int L, int H, float scale, hipStream_t st) {
    static std::atomic<bool> raised{false};
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_fwd_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLds);
        if (e != hipSuccess) return (int)e;
        raised = true;
    }
    hipLaunchKernelGGL(attn_fwd_mfma_kernel, dim3(B * H), dim3(64 * kFW), kFwdLds, st, qkv, ctx, lse, L, H, scale);
    return upp_launch_status();
}

int upp_attn_bwd_mfma(const float *qkv, const float *ctx, const float *d_ctx, const float *lse, float *d_qkv, int B, int L, int H,
                      float scale, hipStream_t st) {
    static std::atomic<bool> raised{false};
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdLds);
        if (e != hipSuccess) return (int)e;
        raised = true;
    }
    hipLaunchKernelGGL(attn_bwd_mfma_kernel, dim3(B * H), dim3(64 * kNW), kBwdLds, st, qkv, ctx, d_ctx, lse, d_qkv, L, H, scale);
    return upp_launch_status();
}
