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
constexpr int kVT = 4;           // MODE 2: at most this many rows behind row 128 (all of their work on the vector ALU)
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
// MODE 2 (1 ... kVT rows behind row 128: cls + 128 groups = 129 tokens): the tail never touches the matrix pipe.  With the strip on MFMA
// the first block's S list is 8 + 4 tiles -- a second round for one query -- and a fifth key tile doubles every block's S rounds; on
// the vector ALU the tail KEYS are `tail` dot products per query (one wave each, lane = query) and a rank-`tail` update of the P V
// accumulators, the tail QUERIES 3 x tail jobs of 64 keys (lane = key) and one P V row per otherwise idle wave (lane = channel).
template <int MODE>
__global__ __launch_bounds__(64 * kLW) void attn_fwd_long_kernel(const float *__restrict__ qkv, float *__restrict__ ctx,
                                                                 float *__restrict__ lse, int L, int H, float scale) {
    constexpr bool FOLD = MODE != 0, VT = MODE == 2;
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
    const int Lmain = FOLD ? 2 * kQB : L, tail = FOLD ? L - 2 * kQB : 0;       // FOLD: 128 < L <= 128 + kTR (checked by the host)
    const int nt = VT ? 4 : (L + 31) / 32;              // key tiles on the matrix pipe
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
        const int n_s = qt * nt + (with_tail && !VT ? nt : 0);
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
        if constexpr (VT) {
            // jobs [0, tail): S column of tail key 128 + j for the block's 64 queries (lane = query); first block, jobs tail + 3 t + g: tail
            // query t against keys 64 g ... 64 g + 63 (lane = key).  (n_s == kLW here: every wave has had one tile.)
            const int n_j = tail + (with_tail ? 3 * tail : 0);
            for (int j = wave; j < n_j; j += kLW) {
                const bool col = j < tail;
                const int t = col ? j : (j - tail) / 3, g = col ? 0 : (j - tail) - 3 * t;
                const int key = col ? Lmain + t : min(g * 64 + lane, kKP - 1);
                const float *ka = Ks + key * kLD, *qa = col ? Qb + lane * kLD : Qt + t * kLD;   // (rows are 65 floats apart: 32-bit reads, no conflicts)
                float a = 0.0f, c = 0.0f;
#pragma unroll
                for (int d0 = 0; d0 < 64; d0 += 16) {
                    float x[16], y[16];
#pragma unroll
                    for (int d = 0; d < 16; ++d) { x[d] = qa[d0 + d]; y[d] = ka[d0 + d]; }
#pragma unroll
                    for (int d = 0; d < 16; d += 2) { a = __builtin_fmaf(x[d], y[d], a); c = __builtin_fmaf(x[d + 1], y[d + 1], c); }
                }
                a += c;
                if (col) Ss[lane * kSS + Lmain + t] = a * scale;
                else if (g * 64 + lane < kKP) St[t * kSS + g * 64 + lane] = a * scale;
            }
        }
        __syncthreads();
        softmax_rows(Ss, qt * 32, qv, q0, wave);
        if (with_tail) softmax_rows(St, VT ? tail : kTR, tail, Lmain, kLW - 1 - wave);       // (the last waves first: they are the ones with a short first loop)
        __syncthreads();
        const int n_o = qt * 2 + (with_tail && !VT ? 2 : 0);
        for (int t = wave; t < n_o; t += kLW) {         // O_b = P V
            const bool is_tail = t >= qt * 2;
            const int u = is_tail ? t - qt * 2 : t;
            const int it = u >> 1, dt = u & 1;
            f32x16 acc; zero(acc);
            const float *prow = is_tail ? St : Ss + it * 32 * kSS;
            mfma_tile<false, false>(acc, prow, kSS, Vs + dt * 32, kLD, nt * 32, lr, lk);
            if constexpr (VT) {                          // + sum over the tail keys of p[row][128 + t] v[128 + t][col]
                for (int t2 = 0; t2 < tail; ++t2) {
                    const float v = Vs[(Lmain + t2) * kLD + dt * 32 + lr];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = __builtin_fmaf(prow[tile_row(r, lk) * kSS + Lmain + t2], v, acc[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (is_tail ? Lmain : q0 + it * 32) + tile_row(r, lk);
                if (i < L && (!is_tail || tile_row(r, lk) < kTR)) ctx[((size_t)b * L + i) * (H * 64) + hh * 64 + dt * 32 + lr] = acc[r];
            }
        }
        if constexpr (VT) {                              // tail query t: one output row on wave 4 + t (the P V list above is 4 tiles), lane = channel
            static_assert(kVT <= kLW - 4, "one idle wave per tail row");
            if (with_tail && wave >= 4 && wave - 4 < tail) {
                const int t = wave - 4;
                const float *p = St + t * kSS;
                float a0 = 0.0f, a1 = 0.0f;
                for (int j0 = 0; j0 < L; j0 += 16) {      // (columns [L, 160) of the strip and rows [L, 160) of V are zero)
                    float pv[16], vv[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) { pv[j] = p[j0 + j]; vv[j] = Vs[(j0 + j) * kLD + lane]; }
#pragma unroll
                    for (int j = 0; j < 16; j += 2) { a0 = __builtin_fmaf(pv[j], vv[j], a0); a1 = __builtin_fmaf(pv[j + 1], vv[j + 1], a1); }
                }
                ctx[((size_t)b * L + Lmain + t) * (H * 64) + hh * 64 + lane] = a0 + a1;
            }
        }
        __syncthreads();
    }
}

// (round 5: folding the rows behind row 128 into the first 32-row block, as the forward kernel does, was built and measured for this
//  kernel too -- 64.1 us against 63.7 at L = 129: the fifth KEY tile (one key of 32) costs what the fifth query block cost, in every
//  block's S / dP phase, accumulator round and dQ contraction; not kept.)
// VT (129 ... 128 + kVT tokens, as the forward's MODE 2): four 32-row blocks and four key tiles -- the walk of a 128-token sequence --
// with the 1 ... kVT tokens behind row 128 on the vector ALU, as keys AND as queries:
//   before the loop   tail queries against all keys (3 tail jobs, lane = key: s, dP, P, dS rows into Pt / Dt); then their rank-1 terms go
//                     into the dV / dK accumulators where those live (registers of the owner waves), their dQ rows on waves 6 / 7
//   phase 1           waves 4 ... 4 + tail - 1: S and dP of one tail key for the block's 32 queries (the two half-waves), P / dS columns
//   phase 2           dQ += dS[:, tail] K[tail] on the dQ waves; dV / dK of the tail keys accumulate in one register per key on waves
//                     4 / 5 (lane = channel; those two own only 2 of the 16 accumulator tiles)
template <bool VT>
__global__ __launch_bounds__(64 * kLW) void attn_bwd_long_kernel(const float *__restrict__ qkv, const float *__restrict__ ctx,
                                                                 const float *__restrict__ d_ctx, const float *__restrict__ lse,
                                                                 float *__restrict__ d_qkv, int L, int H, float scale) {
    extern __shared__ float sm[];
    float *Ks = sm, *Vs = Ks + kKP * kLD, *Qb = Vs + kKP * kLD, *Gb = Qb + kBB * kLD, *Ps = Gb + kBB * kLD, *Ds = Ps + kBB * kSS;
    float *delta = Ds + kBB * kSS, *lses = delta + kBB;
    float *Qt = lses + kBB, *Gt = Qt + kVT * kLD, *Pt = Gt + kVT * kLD, *Dt = Pt + kVT * kSS, *delta_t = Dt + kVT * kSS, *lse_t = delta_t + kVT;   // (VT only)
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
    const int Lmain = VT ? 4 * kBB : L, tail = VT ? L - 4 * kBB : 0;
    const int nt = VT ? 4 : (L + 31) / 32;
    // dV / dK accumulator tiles: index a = which * 2 nt + jt * 2 + dt, owned by wave a % kAccW, slot a / kAccW
    constexpr int NS = VT ? 3 : kAccS;
    f32x16 acc[NS];
#pragma unroll
    for (int m = 0; m < NS; ++m) zero(acc[m]);
    float tacc[kVT] = {0.0f, 0.0f, 0.0f, 0.0f};          // VT: wave 4: dV rows of the tail keys, wave 5: their dK rows (lane = channel)
    static_assert(kVT == 4 && kVT <= kLW - 4, "tail keys: one phase-1 wave each");
    if constexpr (VT) {
        for (int i = threadIdx.x; i < tail * 64; i += 64 * kLW) {
            const int t = i >> 6, d = i & 63;
            Qt[t * kLD + d] = base[(size_t)(Lmain + t) * rs + d];
            Gt[t * kLD + d] = gbase[(size_t)(Lmain + t) * cs + d];
        }
        if (wave < tail) {
            const float dl = wave_sum_f32(gbase[(size_t)(Lmain + wave) * cs + lane] * obase[(size_t)(Lmain + wave) * cs + lane]);
            if (lane == 0) { delta_t[wave] = dl; lse_t[wave] = lse[((size_t)b * H + hh) * L + Lmain + wave]; }
        }
        __syncthreads();
        for (int j = wave; j < 3 * tail; j += kLW) {     // tail query t against keys 64 g ... 64 g + 63
            const int t = j / 3, g = j - 3 * t, key = g * 64 + lane, kc = min(key, kKP - 1);
            const float *qa = Qt + t * kLD, *ga = Gt + t * kLD, *ka = Ks + kc * kLD, *va = Vs + kc * kLD;
            float s0 = 0.0f, s1 = 0.0f, p0 = 0.0f, p1 = 0.0f;
#pragma unroll
            for (int d0 = 0; d0 < 64; d0 += 8) {
                float x[8], y[8], u[8], w[8];
#pragma unroll
                for (int d = 0; d < 8; ++d) { x[d] = qa[d0 + d]; y[d] = ka[d0 + d]; u[d] = ga[d0 + d]; w[d] = va[d0 + d]; }
#pragma unroll
                for (int d = 0; d < 8; d += 2) {
                    s0 = __builtin_fmaf(x[d], y[d], s0); s1 = __builtin_fmaf(x[d + 1], y[d + 1], s1);
                    p0 = __builtin_fmaf(u[d], w[d], p0); p1 = __builtin_fmaf(u[d + 1], w[d + 1], p1);
                }
            }
            if (key < kKP) {
                const float pr = key < L ? exp_neg((s0 + s1) * scale - lse_t[t]) : 0.0f;
                Pt[t * kSS + key] = pr;
                Dt[t * kSS + key] = key < L ? pr * ((p0 + p1) - delta_t[t]) * scale : 0.0f;
            }
        }
        __syncthreads();
        if (wave < kAccW) {                              // rank-1 terms of the tail queries: dV += P_t^T dO_t, dK += dS_t^T Q_t
#pragma unroll
            for (int m = 0; m < NS; ++m) {
                const int a = wave + kAccW * m;
                if (a < nt * 4) {
                    const int which = a / (nt * 2), u = a - which * nt * 2;
                    const int jt = u >> 1, dt = u & 1;
                    const float *rowv = (which == 0 ? Pt : Dt) + jt * 32, *colv = (which == 0 ? Gt : Qt) + dt * 32 + lr;
                    for (int t = 0; t < tail; ++t) {
                        const float c = colv[t * kLD];
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[m][r] = __builtin_fmaf(rowv[t * kSS + tile_row(r, lk)], c, acc[m][r]);
                    }
                }
            }
            if (wave >= 4) {                             // ... and into the tail keys' own rows
                const float *rowv = wave == 4 ? Pt : Dt, *colv = wave == 4 ? Gt : Qt;
                for (int t = 0; t < tail; ++t) {
                    const float c = colv[t * kLD + lane];
#pragma unroll
                    for (int k = 0; k < kVT; ++k) tacc[k] = __builtin_fmaf(k < tail ? rowv[t * kSS + Lmain + k] : 0.0f, c, tacc[k]);
                }
            }
        } else {                                         // dQ rows of the tail queries (lane = channel): wave 6: t = 0, 2; wave 7: t = 1, 3
            for (int t = wave - kAccW; t < tail; t += kLW - kAccW) {
                const float *dsr = Dt + t * kSS;
                float a0 = 0.0f, a1 = 0.0f;
                for (int j0 = 0; j0 < L; j0 += 16) {      // (columns [L, 160) of the strip and rows [L, 160) of K are zero)
                    float x[16], y[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) { x[j] = dsr[j0 + j]; y[j] = Ks[(j0 + j) * kLD + lane]; }
#pragma unroll
                    for (int j = 0; j < 16; j += 2) { a0 = __builtin_fmaf(x[j], y[j], a0); a1 = __builtin_fmaf(x[j + 1], y[j + 1], a1); }
                }
                dbase[(size_t)(Lmain + t) * rs + lane] = a0 + a1;
            }
        }
    }
    float *const Ks0 = Ks, *const Vs0 = Vs, *const Qb0 = Qb, *const Gb0 = Gb, *const Ps0 = Ps, *const Ds0 = Ds;
    for (int q0 = 0; q0 < Lmain; q0 += kBB) {
        const int qv = min(kBB, Lmain - q0);
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
        } else if (VT && wave - 4 < tail) {              // tail key 128 + t: lanes 0..31 S, lanes 32..63 dP of query lr
            const int t = wave - 4;
            const float *x = (lk ? Gb : Qb) + lr * kLD, *y = (lk ? Vs : Ks) + (Lmain + t) * kLD;
            float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
            for (int d0 = 0; d0 < 64; d0 += 16) {
                float xv[16], yv[16];
#pragma unroll
                for (int d = 0; d < 16; ++d) { xv[d] = x[d0 + d]; yv[d] = y[d0 + d]; }
#pragma unroll
                for (int d = 0; d < 16; d += 2) { a0 = __builtin_fmaf(xv[d], yv[d], a0); a1 = __builtin_fmaf(xv[d + 1], yv[d + 1], a1); }
            }
            const float mine = a0 + a1, other = __shfl_xor(mine, 32);
            const float sv = lk ? other : mine, dpv = lk ? mine : other;
            const float p = exp_neg(sv * scale - lses[lr]);
            if (lk) Ds[lr * kSS + Lmain + t] = p * (dpv - delta[lr]) * scale;
            else Ps[lr * kSS + Lmain + t] = p;
        }
        __syncthreads();
        // phase 2: dV += P^T dO_b, dK += dS^T Q_b on the owner waves; dQ_b = dS K on the last two waves
        if (wave < kAccW) {
#pragma unroll
            for (int m = 0; m < NS; ++m) {
                const int a = wave + kAccW * m;
                if (a < nt * 4) {
                    const int which = a / (nt * 2), u = a - which * nt * 2;
                    const int jt = u >> 1, dt = u & 1;
                    if (which == 0) mfma_tile_k<true, false, 32>(acc[m], Ps + jt * 32, kSS, Gb + dt * 32, kLD, lr, lk);
                    else mfma_tile_k<true, false, 32>(acc[m], Ds + jt * 32, kSS, Qb + dt * 32, kLD, lr, lk);
                }
            }
            if (VT && wave >= 4) {                       // rows of the tail keys: += P[:, 128 + k]^T dO_b (wave 4), dS[:, 128 + k]^T Q_b (wave 5)
                const float *strip = (wave == 4 ? Ps : Ds) + Lmain, *opnd = (wave == 4 ? Gb : Qb) + lane;
#pragma unroll
                for (int i0 = 0; i0 < kBB; i0 += 16) {
                    float c[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) c[i] = opnd[(i0 + i) * kLD];
#pragma unroll
                    for (int k = 0; k < kVT; ++k)
                        if (k < tail) {
                            float pv[16];
#pragma unroll
                            for (int i = 0; i < 16; ++i) pv[i] = strip[(i0 + i) * kSS + k];
#pragma unroll
                            for (int i = 0; i < 16; ++i) tacc[k] = __builtin_fmaf(pv[i], c[i], tacc[k]);
                        }
                }
            }
        } else {
            const int dt = wave - kAccW;
            f32x16 dq; zero(dq);
            mfma_tile<false, false>(dq, Ds, kSS, Ks + dt * 32, kLD, nt * 32, lr, lk);
            if constexpr (VT) {
                for (int t = 0; t < tail; ++t) {
                    const float kv = Ks[(Lmain + t) * kLD + dt * 32 + lr];
#pragma unroll
                    for (int r = 0; r < 16; ++r) dq[r] = __builtin_fmaf(Ds[tile_row(r, lk) * kSS + Lmain + t], kv, dq[r]);
                }
            }
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
        for (int m = 0; m < NS; ++m) {
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
        if (VT && wave >= 4) {
#pragma unroll
            for (int k = 0; k < kVT; ++k)
                if (k < tail) dbase[(size_t)(Lmain + k) * rs + (wave == 4 ? 2 : 1) * H * 64 + lane] = tacc[k];
        }
    }
}

constexpr size_t kFwdLongLds = ((size_t)2 * kKP * kLD + (size_t)kQB * kLD + (size_t)kQB * kSS) * sizeof(float);
constexpr size_t kFwdFoldLds = kFwdLongLds + ((size_t)kTR * kLD + (size_t)kTR * kSS) * sizeof(float);
static_assert(kFwdFoldLds <= 160 * 1024, "LDS budget");
constexpr size_t kBwdLongLds = ((size_t)2 * kKP * kLD + (size_t)2 * kBB * kLD + (size_t)2 * kBB * kSS + 2 * kBB) * sizeof(float);
constexpr size_t kBwdTailLds = kBwdLongLds + ((size_t)2 * kVT * kLD + (size_t)2 * kVT * kSS + 2 * kVT) * sizeof(float);
static_assert(kBwdTailLds <= 160 * 1024, "LDS budget");

// 129 ... 144 tokens: 1 ... kVT tail rows on the vector ALU (kernel variant 2), longer tails as a strip on the matrix pipe (variant 1);
// every other length the three-block walk (variant 0).  (round 5 carried an environment switch for A/B timing: profiles/r05_time_attention_long*.txt)
constexpr int fold_mode() { return 2; }

}  // namespace

// called by upp_attn_fwd / upp_attn_bwd (block.hip) for 96 < L <= 160
int upp_attn_fwd_long(const float *qkv, float *ctx, float *lse, int B, int L, int H, float scale, hipStream_t st) {
    static std::atomic<bool> raised{false};
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_fwd_long_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLongLds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_fwd_long_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdFoldLds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_fwd_long_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdFoldLds);
        if (e != hipSuccess) return (int)e;
        raised = true;
    }
    const int mode = (L > 2 * kQB && L <= 2 * kQB + kTR) ? fold_mode() : 0;
    if (mode == 2 && L <= 2 * kQB + kVT)
        hipLaunchKernelGGL(attn_fwd_long_kernel<2>, dim3(B * H), dim3(64 * kLW), kFwdFoldLds, st, qkv, ctx, lse, L, H, scale);
    else if (mode != 0)
        hipLaunchKernelGGL(attn_fwd_long_kernel<1>, dim3(B * H), dim3(64 * kLW), kFwdFoldLds, st, qkv, ctx, lse, L, H, scale);
    else
        hipLaunchKernelGGL(attn_fwd_long_kernel<0>, dim3(B * H), dim3(64 * kLW), kFwdLongLds, st, qkv, ctx, lse, L, H, scale);
    return upp_launch_status();
}

int upp_attn_bwd_long(const float *qkv, const float *ctx, const float *d_ctx, const float *lse, float *d_qkv, int B, int L, int H,
                      float scale, hipStream_t st) {
    static std::atomic<bool> raised{false};
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_long_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdLongLds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_long_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdTailLds);
        if (e != hipSuccess) return (int)e;
        raised = true;
    }
    if (L > 4 * kBB && L <= 4 * kBB + kVT && fold_mode() == 2)
        hipLaunchKernelGGL(attn_bwd_long_kernel<true>, dim3(B * H), dim3(64 * kLW), kBwdTailLds, st, qkv, ctx, d_ctx, lse, d_qkv, L, H, scale);
    else
        hipLaunchKernelGGL(attn_bwd_long_kernel<false>, dim3(B * H), dim3(64 * kLW), kBwdLongLds, st, qkv, ctx, d_ctx, lse, d_qkv, L, H, scale);
    return upp_launch_status();
}
