// attn_flash16.hip -- attention core for L <= 96 tokens, head dim 64, on v_mfma_f32_16x16x4_f32, forward without LDS and
// without a barrier.  Same contract as upp_attn_fwd / upp_attn_bwd (include/upp_hip.h; reference
// models/Point_MAE_pretask_dev.py:186-193).
//
// The sequences of the UPP blocks are 35 / 64 / 65 / 75 tokens: on 32x32 tiles they pad to 64 / 64 / 96 / 96 (up to 1.64x the
// MFMA work), and the round-1 kernels (attn_mfma.hip) spend most of their 13 us in phases -- stage Q, K, V through registers
// into padded LDS rows, barrier, S tiles to LDS, barrier, row softmax from LDS, barrier, P V -- not in their ~2.4 us of MFMA.
// Here one workgroup per (sample, head) has ceil(L/16) waves; wave w owns the 16 queries 16 w .. 16 w + 15 and ALL keys:
//
//   S^T tile t (16 keys x 16 queries) = K_t . Q_w^T      A = K rows, B = Q rows: both loaded straight from global memory as
//                                                        16-byte pieces (the contraction over d may run in any order: lane
//                                                        (i = lane & 15, g = lane >> 4) takes d = 16 m + 4 g + c for the
//                                                        k-step (m, c), so one float4 feeds four MFMAs)
//   softmax over the keys of a query = over the 4 accumulator registers x ceil(L/16) tiles of a lane, then over the four lane
//                                      groups g (two __shfl_xor): in registers, no LDS round trip
//   O_w (16 queries x 64) = P_w . V                      A = P: the accumulator register s of S^T tile t IS the A operand of
//                                                        the k-step "keys 16 t + 4 g + s, g = 0..3" -- no lane movement, no
//                                                        transpose; B = V rows, one dword per lane (64 contiguous bytes per key)
//
// A wave never waits for another wave: no __syncthreads, no LDS.  All loads of a wave (Q, K, then V) are issued before the
// first MFMA.  Masking: keys >= L get p = 0 (their rows are clamped loads), queries >= L are computed and not stored.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float exp2_scaled(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float comp(const float4 &v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

// One wave per workgroup: the NT query tiles of a (sample, head) are NT independent single-wave workgroups, so that the
// dispatcher spreads B H NT waves over the 1,024 SIMDs (five waves of one workgroup on the four SIMDs of a CU run 2:1:1:1).
__device__ __forceinline__ void unit_of(int unit, int NT, int BH, int &bh, int &tile) {
    // XCD x takes the contiguous run of (sample, head) pairs [x BH/8, (x+1) BH/8) -- the token rows it owns in the other kernels of
    // the block (common.h xcd_contiguous) -- with all NT query tiles of a pair on that XCD
    const int lin = xcd_contiguous(unit, BH * NT);
    bh = lin / NT;
    tile = lin - bh * NT;
}

template <int NT>
__global__ __launch_bounds__(64) void attn_fwd16_kernel(const float *__restrict__ qkv, float *__restrict__ ctx,
                                                       float *__restrict__ lse, int L, int H, float scale, int BH) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    int bh, wave;
    unit_of(blockIdx.x, NT, BH, bh, wave);
    const int b = bh / H, hh = bh - b * H;
    const size_t rs = (size_t)3 * H * 64;
    const float *qb = qkv + (size_t)b * L * rs + (size_t)hh * 64, *kb = qb + H * 64, *vb = qb + 2 * H * 64;
    const int q0 = wave * 16;
    if (q0 >= L) return;

    // ---- operand loads: Q (B of S^T), K (A of S^T), V (B of P V); rows clamped to L - 1
    float4 qv[4], kv[NT][4];
    const float *qrow = qb + (size_t)min(q0 + j, L - 1) * rs + 4 * g;
#pragma unroll
    for (int m = 0; m < 4; ++m) qv[m] = *reinterpret_cast<const float4 *>(qrow + 16 * m);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float *krow = kb + (size_t)min(16 * t + j, L - 1) * rs + 4 * g;
#pragma unroll
        for (int m = 0; m < 4; ++m) kv[t][m] = *reinterpret_cast<const float4 *>(krow + 16 * m);
    }
    float vv[NT][4][4];                                  // [key tile][s][d tile]: V[16 t + 4 g + s][16 dt + j]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float *vrow = vb + (size_t)min(16 * t + 4 * g + s, L - 1) * rs + j;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) vv[t][s][dt] = vrow[16 * dt];
        }
    // (measured and dropped: V as 16-byte pieces through a private LDS tile, dword operand reads from there: 8.0 vs 7.7 us)

    // (also dropped: a scheduling barrier that keeps every load in front of the first MFMA -- 8.4 us: the K rows then queue behind
    // the 80 V loads of all 960 waves)

    // ---- S^T tiles
    f32x4 st[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int c = 0; c < 4; ++c) st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(kv[t][m], c), comp(qv[m], c), st[t], 0, 0, 0);
    }
    // ---- softmax over the keys of query q0 + j: registers (4) x tiles (NT) in the lane, then the four lane groups
    const float sl = scale * 1.4426950408889634f;        // exp(x * scale) = exp2(x * scale * log2 e)
    float mx = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool live = 16 * t + 4 * g + s < L;
            st[t][s] = live ? st[t][s] : -__builtin_inff();
            mx = fmaxf(mx, st[t][s]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float p = exp2_scaled((st[t][s] - mx) * sl);    // masked keys: exp2(-inf) = 0
            st[t][s] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    if (g == 0 && q0 + j < L) lse[((size_t)b * H + hh) * L + q0 + j] = mx * scale + logf(sum);
    // ---- O = P V: register s of S^T tile t is the A operand of the k-step over keys 16 t + 4 g + s
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float p = st[t][s] * inv;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, vv[t][s][dt], o[dt], 0, 0, 0);
        }
    // o[dt][reg] = O[q0 + 4 g + reg][16 dt + j]
    float *ob = ctx + (size_t)b * L * (H * 64) + (size_t)hh * 64 + j;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int q = q0 + 4 * g + reg;
        if (q < L) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) ob[(size_t)q * (H * 64) + 16 * dt] = o[dt][reg];
        }
    }
}

// Backward, same partition.  Phase 1, wave w = query tile w, everything in registers:
//   S^T_t = K_t Q_w^T,  dP^T_t = V_t dO_w^T   (A = K / V rows, B = Q / dO rows: float4 loads, four MFMAs each)
//   P^T = exp(S scale - lse),  dS^T = P^T (dP^T - delta) scale,   delta_q = dO_q . O_q  (lane-local partial + two shuffles)
//   dQ_w = dS_w K : the accumulator registers of dS^T are the A operand (as P in the forward), B = K rows (one dword per lane)
// then P^T and dS^T go to the LDS ([key][query], the only exchange of the kernel, ONE barrier) and, phase 2, wave w = key
// tile w computes  dV_w = P^T_w dO  and  dK_w = dS^T_w Q  with A from the LDS (16-byte reads along the queries) and B = dO / Q
// rows from global memory: complete sums over all queries inside one wave -- no atomics, no cross-wave reduction.
// (Measured and dropped: the backward as 2 NT independent single-wave workgroups per (sample, head), the key-tile units
// recomputing S and dP un-transposed instead of reading them from the LDS -- no barrier and a better spread over the SIMDs,
// but 40 % more MFMAs and five dependent load -> MFMA rounds per key-tile wave: 24.3 us against 21.8 us at L = 75.)
template <int NT>
__global__ __launch_bounds__(64 * NT) void attn_bwd16_kernel(const float *__restrict__ qkv, const float *__restrict__ ctx,
                                                            const float *__restrict__ d_ctx, const float *__restrict__ lse,
                                                            float *__restrict__ d_qkv, int L, int H, float scale) {
    constexpr int LP = 16 * NT, LS = LP + 4;             // padded length, LDS row stride (floats; 16-byte aligned rows)
    __shared__ __attribute__((aligned(16))) float PT[LP * LS], DS[LP * LS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 15, g = lane >> 4;
    const int bhx = xcd_contiguous(blockIdx.x, gridDim.x);
    const int b = bhx / H, hh = bhx - b * H;
    const size_t rs = (size_t)3 * H * 64, cs = (size_t)H * 64;
    const float *qb = qkv + (size_t)b * L * rs + (size_t)hh * 64, *kb = qb + H * 64, *vb = qb + 2 * H * 64;
    const float *gb = d_ctx + (size_t)b * L * cs + (size_t)hh * 64, *ob = ctx + (size_t)b * L * cs + (size_t)hh * 64;
    float *dqb = d_qkv + (size_t)b * L * rs + (size_t)hh * 64, *dkb = dqb + H * 64, *dvb = dqb + 2 * H * 64;
    const int q0 = wave * 16;                            // (every wave index < NT has q0 < LP; tiles beyond L are all-masked)
    const float sl = scale * 1.4426950408889634f;

    // ---- phase 1 -------------------------------------------------------------------------------------------------
    {
        const int qr = min(q0 + j, L - 1);
        const bool qlive = q0 + j < L;
        float4 qv[4], gv[4], ov[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            qv[m] = *reinterpret_cast<const float4 *>(qb + (size_t)qr * rs + 16 * m + 4 * g);
            gv[m] = *reinterpret_cast<const float4 *>(gb + (size_t)qr * cs + 16 * m + 4 * g);
            ov[m] = *reinterpret_cast<const float4 *>(ob + (size_t)qr * cs + 16 * m + 4 * g);
        }
        const float lse_q = lse[((size_t)b * H + hh) * L + qr] * 1.4426950408889634f;
        float4 kv[NT][4], vv[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int kr = min(16 * t + j, L - 1);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                kv[t][m] = *reinterpret_cast<const float4 *>(kb + (size_t)kr * rs + 16 * m + 4 * g);
                vv[t][m] = *reinterpret_cast<const float4 *>(vb + (size_t)kr * rs + 16 * m + 4 * g);
            }
        }
        float delta = 0.0f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
            delta += gv[m].x * ov[m].x + gv[m].y * ov[m].y + gv[m].z * ov[m].z + gv[m].w * ov[m].w;
        delta += __shfl_xor(delta, 16);
        delta += __shfl_xor(delta, 32);
        f32x4 st[NT], dp[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(kv[t][m], c), comp(qv[m], c), st[t], 0, 0, 0);
                    dp[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(vv[t][m], c), comp(gv[m], c), dp[t], 0, 0, 0);
                }
        }
        // K rows again as the B operand of dQ = dS K: K[16 t + 4 g + s][16 dt + j]
        float kd[NT][4][4];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float *krow = kb + (size_t)min(16 * t + 4 * g + s, L - 1) * rs + j;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) kd[t][s][dt] = krow[16 * dt];
            }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool live = qlive && 16 * t + 4 * g + s < L;
                const float p = live ? exp2_scaled(st[t][s] * sl - lse_q) : 0.0f;
                st[t][s] = p;
                dp[t][s] = p * (dp[t][s] - delta) * scale;
            }
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dp[t][s], kd[t][s][dt], dq[dt], 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int q = q0 + 4 * g + reg;
            if (q < L) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dqb[(size_t)q * rs + 16 * dt + j] = dq[dt][reg];
            }
        }
        // exchange: [key][query]
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                PT[(16 * t + 4 * g + s) * LS + q0 + j] = st[t][s];
                DS[(16 * t + 4 * g + s) * LS + q0 + j] = dp[t][s];
            }
    }
    __syncthreads();
    // ---- phase 2: wave = key tile ------------------------------------------------------------------------------------
    {
        const int k0 = wave * 16;
        if (k0 >= L) return;
        f32x4 dv[4], dk[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            // A: P^T / dS^T [key k0 + j][queries 16 u + 4 g .. + 3]; B: dO / Q [query 16 u + 4 g + s][16 dt + j]
            const float4 pa = *reinterpret_cast<const float4 *>(&PT[(k0 + j) * LS + 16 * u + 4 * g]);
            const float4 da = *reinterpret_cast<const float4 *>(&DS[(k0 + j) * LS + 16 * u + 4 * g]);
            float gd[4][4], qd[4][4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int qr = min(16 * u + 4 * g + s, L - 1);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    gd[s][dt] = gb[(size_t)qr * cs + 16 * dt + j];
                    qd[s][dt] = qb[(size_t)qr * rs + 16 * dt + j];
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(pa, s), gd[s][dt], dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(da, s), qd[s][dt], dk[dt], 0, 0, 0);
                }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int key = k0 + 4 * g + reg;
            if (key < L) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dvb[(size_t)key * rs + 16 * dt + j] = dv[dt][reg];
                    dkb[(size_t)key * rs + 16 * dt + j] = dk[dt][reg];
                }
            }
        }
    }
}


// Backward with the operand rows staged once per workgroup (round 2, second version).  In the kernel above every wave fetches its
// B operands (K for dQ, dO for dV, Q for dK: one dword per lane and MFMA k-step) straight from global memory -- 240 dword loads per
// wave that the compiler can only keep a handful in flight of, and five waves fetching the same K rows: the kernel was bound by
// load latency, not by its 12 us of MFMA (measured 21.8 us at L = 75).  Here the 4 row blocks K, V, Q, dO of the (sample, head)
// go to the LDS once (coalesced 16-byte loads, rows padded to 68 floats: the 16-byte A-operand reads of 16 rows and the dword
// B-operand reads of 4 rows x 16 columns are both conflict-free), and every operand of every product is an LDS read.
// Same partition and arithmetic as attn_bwd16_kernel: bit-identical results.
template <int NT, bool SPLIT>
__global__ __launch_bounds__(64 * NT * (SPLIT ? 2 : 1)) void attn_bwd16l_kernel(const float *__restrict__ qkv, const float *__restrict__ ctx,
                                                             const float *__restrict__ d_ctx, const float *__restrict__ lse,
                                                             float *__restrict__ d_qkv, int L, int H, float scale) {
    constexpr int LP = 16 * NT, RS = 68, LS = LP + 4, NW = NT * (SPLIT ? 2 : 1);
    constexpr int T0 = SPLIT ? (NT + 1) / 2 : NT;     // SPLIT: waves [0, NT) take key tiles [0, T0) of phase 1, waves [NT, 2 NT) the rest
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *Ks = sm, *Vs = Ks + LP * RS, *Qs = Vs + LP * RS, *Gs = Qs + LP * RS;     // [LP][68] each
    float *PT = Gs + LP * RS, *DS = PT + LP * LS;                                   // [LP][LP + 4] each
    float *DQ = DS + LP * LS;                                                       // (SPLIT) [NT][16][64] partial dQ of the second half
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 15, g = lane >> 4;
    const int bhx = xcd_contiguous(blockIdx.x, gridDim.x);
    const int b = bhx / H, hh = bhx - b * H;
    const size_t rs = (size_t)3 * H * 64, cs = (size_t)H * 64;
    const float *qb = qkv + (size_t)b * L * rs + (size_t)hh * 64, *kb = qb + H * 64, *vb = qb + 2 * H * 64;
    const float *gb = d_ctx + (size_t)b * L * cs + (size_t)hh * 64, *ob = ctx + (size_t)b * L * cs + (size_t)hh * 64;
    float *dqb = d_qkv + (size_t)b * L * rs + (size_t)hh * 64, *dkb = dqb + H * 64, *dvb = dqb + 2 * H * 64;
    const int half = SPLIT ? wave / NT : 0, wq = wave - half * NT;
    const int q0 = wq * 16;
    const float sl = scale * 1.4426950408889634f;

    // ---- stage K, V, Q, dO: LP rows x 16 float4 each; thread -> (row, piece); rows beyond L are clamped copies (masked later)
    {
        constexpr int PER = LP * 16 / (64 * NW);          // float4 per thread and array
        float4 kv[PER], vv[PER], qv[PER], gv[PER];
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int i = threadIdx.x + it * 64 * NW, row = i >> 4, c = (i & 15) * 4;
            const int rr = min(row, L - 1);
            kv[it] = *reinterpret_cast<const float4 *>(kb + (size_t)rr * rs + c);
            vv[it] = *reinterpret_cast<const float4 *>(vb + (size_t)rr * rs + c);
            qv[it] = *reinterpret_cast<const float4 *>(qb + (size_t)rr * rs + c);
            gv[it] = *reinterpret_cast<const float4 *>(gb + (size_t)rr * cs + c);
        }
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int i = threadIdx.x + it * 64 * NW, row = i >> 4, c = (i & 15) * 4;
            *reinterpret_cast<float4 *>(&Ks[row * RS + c]) = kv[it];
            *reinterpret_cast<float4 *>(&Vs[row * RS + c]) = vv[it];
            *reinterpret_cast<float4 *>(&Qs[row * RS + c]) = qv[it];
            *reinterpret_cast<float4 *>(&Gs[row * RS + c]) = gv[it];
        }
    }
    // delta_q = dO_q . O_q for the wave's query tile (O straight from global memory: read once)
    const int qr = min(q0 + j, L - 1);
    const bool qlive = q0 + j < L;
    float4 ov[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) ov[m] = *reinterpret_cast<const float4 *>(ob + (size_t)qr * cs + 16 * m + 4 * g);
    const float lse_q = lse[((size_t)b * H + hh) * L + qr] * 1.4426950408889634f;
    __syncthreads();

    // ---- phase 1: wave = query tile -------------------------------------------------------------------------------------
    {
        float4 qv[4], gv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            qv[m] = *reinterpret_cast<const float4 *>(&Qs[(q0 + j) * RS + 16 * m + 4 * g]);
            gv[m] = *reinterpret_cast<const float4 *>(&Gs[(q0 + j) * RS + 16 * m + 4 * g]);
        }
        float delta = 0.0f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
            delta += gv[m].x * ov[m].x + gv[m].y * ov[m].y + gv[m].z * ov[m].z + gv[m].w * ov[m].w;
        delta += __shfl_xor(delta, 16);
        delta += __shfl_xor(delta, 32);
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < T0; ++tt) {
            const int t = half ? T0 + tt : tt;
            if (t >= NT) break;                            // (wave-uniform: the second half has NT - T0 tiles)
            f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float4 kv = *reinterpret_cast<const float4 *>(&Ks[(16 * t + j) * RS + 16 * m + 4 * g]);
                const float4 vv = *reinterpret_cast<const float4 *>(&Vs[(16 * t + j) * RS + 16 * m + 4 * g]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    st = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(kv, c), comp(qv[m], c), st, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(vv, c), comp(gv[m], c), dp, 0, 0, 0);
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool live = qlive && 16 * t + 4 * g + s < L;
                const float p = live ? exp2_scaled(st[s] * sl - lse_q) : 0.0f;
                const float ds = p * (dp[s] - delta) * scale;
                PT[(16 * t + 4 * g + s) * LS + q0 + j] = p;
                DS[(16 * t + 4 * g + s) * LS + q0 + j] = ds;
                // dQ += dS K: B = K[16 t + 4 g + s][16 dt + j]
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds, Ks[(16 * t + 4 * g + s) * RS + 16 * dt + j], dq[dt], 0, 0, 0);
            }
        }
        if (SPLIT) {
            // the two halves of a query tile meet in the LDS: the second writes its partial dQ, the first adds it after the barrier
            if (half) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) DQ[(wq * 16 + reg * 4 + dt) * 64 + lane] = dq[dt][reg];
            }
            __syncthreads();
            if (!half) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int q = q0 + 4 * g + reg;
                    if (q < L) {
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt)
                            dqb[(size_t)q * rs + 16 * dt + j] = dq[dt][reg] + DQ[(wq * 16 + reg * 4 + dt) * 64 + lane];
                    }
                }
            }
        } else {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int q = q0 + 4 * g + reg;
                if (q < L) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) dqb[(size_t)q * rs + 16 * dt + j] = dq[dt][reg];
                }
            }
            __syncthreads();
        }
    }
    // ---- phase 2: wave = key tile -----------------------------------------------------------------------------------------
    {
        const int k0 = wq * 16;
        if (k0 >= L) return;
        if (SPLIT) {
            // wave (key tile, which): dV = P^T dO (first NT waves) or dK = dS^T Q (the others)
            const float *As = half ? DS : PT, *Bs = half ? Qs : Gs;
            float *dst = half ? dkb : dvb;
            f32x4 acc[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const float4 aa = *reinterpret_cast<const float4 *>(&As[(k0 + j) * LS + 16 * u + 4 * g]);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(aa, s), Bs[(16 * u + 4 * g + s) * RS + 16 * dt + j], acc[dt], 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int key = k0 + 4 * g + reg;
                if (key < L) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) dst[(size_t)key * rs + 16 * dt + j] = acc[dt][reg];
                }
            }
            return;
        }
        f32x4 dv[4], dk[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const float4 pa = *reinterpret_cast<const float4 *>(&PT[(k0 + j) * LS + 16 * u + 4 * g]);
            const float4 da = *reinterpret_cast<const float4 *>(&DS[(k0 + j) * LS + 16 * u + 4 * g]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(pa, s), Gs[(16 * u + 4 * g + s) * RS + 16 * dt + j], dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(da, s), Qs[(16 * u + 4 * g + s) * RS + 16 * dt + j], dk[dt], 0, 0, 0);
                }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int key = k0 + 4 * g + reg;
            if (key < L) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dvb[(size_t)key * rs + 16 * dt + j] = dv[dt][reg];
                    dkb[(size_t)key * rs + 16 * dt + j] = dk[dt][reg];
                }
            }
        }
    }
}

template <int NT, bool SPLIT>
int launch_bwd16l(const float *qkv, const float *ctx, const float *d_ctx, const float *lse, float *d_qkv, int B, int L, int H, float scale,
                  hipStream_t st) {
    constexpr int LP = 16 * NT;
    constexpr size_t lds = (size_t)(4 * LP * 68 + 2 * LP * (LP + 4) + (SPLIT ? NT * 16 * 64 : 0)) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS");
    if (lds > 64 * 1024) {
        static std::atomic<bool> raised{false};
        if (!raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd16l_kernel<NT, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            raised = true;
        }
    }
    hipLaunchKernelGGL((attn_bwd16l_kernel<NT, SPLIT>), dim3(B * H), dim3(64 * NT * (SPLIT ? 2 : 1)), lds, st, qkv, ctx, d_ctx, lse, d_qkv, L, H, scale);
    return upp_launch_status();
}

}  // namespace

int upp_attn_bwd_flash16(const float *qkv, const float *ctx, const float *d_ctx, const float *lse, float *d_qkv, int B, int L, int H,
                         float scale, hipStream_t st) {
    const int nt = (L + 15) / 16;
    // LDS-staged kernel; twice the waves where that evens out the four SIMDs (2, 3, 5 tiles: measured 7.9 -> 5.6, 9.6 -> 8.0,
    // 19.1 -> 16.3 us; at 4 tiles the plain version is as fast: 11.1 vs 11.6 us)
    switch (nt) {
        case 1: return launch_bwd16l<1, false>(qkv, ctx, d_ctx, lse, d_qkv, B, L, H, scale, st);
        case 2: return launch_bwd16l<2, true>(qkv, ctx, d_ctx, lse, d_qkv, B, L, H, scale, st);
        case 3: return launch_bwd16l<3, true>(qkv, ctx, d_ctx, lse, d_qkv, B, L, H, scale, st);
        case 4: return launch_bwd16l<4, false>(qkv, ctx, d_ctx, lse, d_qkv, B, L, H, scale, st);
        case 5: return launch_bwd16l<5, true>(qkv, ctx, d_ctx, lse, d_qkv, B, L, H, scale, st);
        default: break;                          // 81 .. 96 tokens: 181 KB of LDS -- the exchanging kernel above
    }
    if (nt != 6) return UPP_E_RANGE;
    hipLaunchKernelGGL((attn_bwd16_kernel<6>), dim3(B * H), dim3(384), 0, st, qkv, ctx, d_ctx, lse, d_qkv, L, H, scale);
    return upp_launch_status();
}

// called by upp_attn_fwd_ex (block.hip) for L <= 96, variant 0
int upp_attn_fwd_flash16(const float *qkv, float *ctx, float *lse, int B, int L, int H, float scale, hipStream_t st) {
    const int nt = (L + 15) / 16;
    const dim3 grid(B * H * nt);
    switch (nt) {
        case 1: hipLaunchKernelGGL((attn_fwd16_kernel<1>), grid, dim3(64), 0, st, qkv, ctx, lse, L, H, scale, B * H); break;
        case 2: hipLaunchKernelGGL((attn_fwd16_kernel<2>), grid, dim3(64), 0, st, qkv, ctx, lse, L, H, scale, B * H); break;
        case 3: hipLaunchKernelGGL((attn_fwd16_kernel<3>), grid, dim3(64), 0, st, qkv, ctx, lse, L, H, scale, B * H); break;
        case 4: hipLaunchKernelGGL((attn_fwd16_kernel<4>), grid, dim3(64), 0, st, qkv, ctx, lse, L, H, scale, B * H); break;
        case 5: hipLaunchKernelGGL((attn_fwd16_kernel<5>), grid, dim3(64), 0, st, qkv, ctx, lse, L, H, scale, B * H); break;
        case 6: hipLaunchKernelGGL((attn_fwd16_kernel<6>), grid, dim3(64), 0, st, qkv, ctx, lse, L, H, scale, B * H); break;
        default: return UPP_E_RANGE;
    }
    return upp_launch_status();
}
