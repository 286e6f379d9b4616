// adapter.hip -- the bottleneck adapter of every UPP block (reference models/Point_MAE_pretask_dev.py:54-104,
// applied at :312-320):   out = x + 0.7 * ( W2 . dropout(gelu(W1 . ha + b1)) + b2 ),   ha = LayerNorm(x)
// with W1 (32, D), W2 (D, 32).  The LayerNorm itself is produced by rowln_fwd (block.hip); this file fuses the two
// skinny GEMMs, GELU, dropout, bias, the 0.7 scale and the residual into one kernel per direction.
//
// The reference runs 6 kernels forward (2 GEMMs with N=32 / K=32 -- the worst shapes for a GEMM library --, GELU,
// dropout, scale, add) and ~12 in backward.  Here a workgroup (4 waves) owns 32 token rows:
//   forward : S1 = ha . W1^T   split-K over the 4 waves (v_mfma_f32_32x32x2_f32, 48 MFMAs each), reduced through LDS;
//             G = dropout(gelu(S1 + b1)) stays in LDS;  out = x + 0.7 (G . W2^T + b2): 12 column tiles, 3 per wave.
//   backward: gz = 0.7 g_out;  gd = gz . W2 (split-K);  ga = gd * dropout' * gelu'(S1);  g_ha = ga . W1;
//             per-workgroup partials of dW1 = ga^T ha, dW2 = gz^T d, db1, db2 (summed by the caller: deterministic).
// ha, gz and (forward) W1 tiles live in LDS, rows padded to D+1 floats: conflict-free one-dword-per-lane MFMA operand
// reads; W2 is staged per wave as 32x32 blocks; the backward reads W1 / W2 rows (coalesced) straight from L2, always
// a batch of operands ahead of the MFMAs.
// exact GELU (erf), f32 throughout.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kH = 32;      // bottleneck width
constexpr int kRows = 32;   // token rows per workgroup
constexpr int kLG = 33;     // row stride of the 32x32 LDS tiles
constexpr int kAW = 8;      // waves per workgroup: two per SIMD (a lone wave issues MFMAs at about half the pipe rate)
constexpr int kAT = 64 * kAW;

__device__ __forceinline__ int trow(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }
__device__ __forceinline__ void zero16(f32x16 &a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.0f;
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    // d/dx [x Phi(x)] = Phi(x) + x phi(x)
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// acc += A . B over NK k-pairs, operands supplied by functors, fetched CH pairs ahead of the MFMAs
template <int NK, typename FA, typename FB>
__device__ __forceinline__ void mfma_chain(f32x16 &acc, FA fa, FB fb) {
    constexpr int CH = 8;
    static_assert(NK % CH == 0, "k-pairs must be a multiple of 8");
    float av[CH], bv[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) { av[i] = fa(i); bv[i] = fb(i); }
#pragma unroll
    for (int c = 0; c < NK / CH; ++c) {
        float an[CH], bn[CH];
        if (c + 1 < NK / CH) {
#pragma unroll
            for (int i = 0; i < CH; ++i) { an[i] = fa((c + 1) * CH + i); bn[i] = fb((c + 1) * CH + i); }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CH; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NK / CH) {
#pragma unroll
            for (int i = 0; i < CH; ++i) { av[i] = an[i]; bv[i] = bn[i]; }
        }
    }
}

// stage `kRows` rows x D of src (row stride D; rows >= R zero) into dst with row stride D+1; all loads first
template <int D>
__device__ __forceinline__ void stage_tile(float *dst, const float *src, int row0, int R, float mul = 1.0f) {
    constexpr int IT = kRows * D / 4 / kAT;
    static_assert(kRows * D / 4 % kAT == 0, "tile must divide over the workgroup");
    float4 v[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int i = threadIdx.x + it * kAT;
        const int r = i / (D / 4), c = (i % (D / 4)) * 4;
        v[it] = row0 + r < R ? *reinterpret_cast<const float4 *>(src + (size_t)(row0 + r) * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int i = threadIdx.x + it * kAT;
        float *d = dst + (i / (D / 4)) * (D + 1) + (i % (D / 4)) * 4;
        d[0] = v[it].x * mul; d[1] = v[it].y * mul; d[2] = v[it].z * mul; d[3] = v[it].w * mul;
    }
}

// the same tile recomputed from the saved rows and statistics of the LayerNorm: dst = (src - mean) * rstd * gamma + beta
// (the expression of rowln_fwd_kernel / ln_adapter_fwd_kernel: identical values)
template <int D>
__device__ __forceinline__ void stage_tile_ln(float *dst, const float *src, const float *mean, const float *rstd, const float *gamma,
                                              const float *beta, int row0, int R) {
    constexpr int IT = kRows * D / 4 / kAT;
    float4 v[IT], gm[IT], bt[IT];
    float mu[IT], rs[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int i = threadIdx.x + it * kAT;
        const int r = i / (D / 4), c = (i % (D / 4)) * 4;
        const bool ok = row0 + r < R;
        v[it] = ok ? *reinterpret_cast<const float4 *>(src + (size_t)(row0 + r) * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        gm[it] = *reinterpret_cast<const float4 *>(gamma + c);
        bt[it] = *reinterpret_cast<const float4 *>(beta + c);
        mu[it] = ok ? mean[row0 + r] : 0.0f;
        rs[it] = ok ? rstd[row0 + r] : 0.0f;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int i = threadIdx.x + it * kAT;
        const bool ok = row0 + i / (D / 4) < R;
        float *d = dst + (i / (D / 4)) * (D + 1) + (i % (D / 4)) * 4;
        d[0] = ok ? __builtin_fmaf((v[it].x - mu[it]) * rs[it], gm[it].x, bt[it].x) : 0.0f;
        d[1] = ok ? __builtin_fmaf((v[it].y - mu[it]) * rs[it], gm[it].y, bt[it].y) : 0.0f;
        d[2] = ok ? __builtin_fmaf((v[it].z - mu[it]) * rs[it], gm[it].z, bt[it].z) : 0.0f;
        d[3] = ok ? __builtin_fmaf((v[it].w - mu[it]) * rs[it], gm[it].w, bt[it].w) : 0.0f;
    }
}

template <int D>
__global__ __launch_bounds__(kAT) void adapter_fwd_kernel(const float *__restrict__ ha, const float *__restrict__ x,
                                                          const float *__restrict__ W1, const float *__restrict__ b1,
                                                          const float *__restrict__ W2, const float *__restrict__ b2,
                                                          const float *__restrict__ u, float p, float scale,
                                                          float *__restrict__ out, float *__restrict__ s1_out, int R) {
    constexpr int LDH = D + 1, KW = D / kAW;        // K range per wave in the split-K product
    extern __shared__ float sm[];
    float *Hs = sm;                                  // [32][D+1]  ha tile
    float *W1s = Hs + kRows * LDH;                   // [32][D+1]  W1
    float *Part = W1s + kH * LDH;                    // [kAW][32][33] split-K partials, later the per-wave W2 tiles
    float *Gs = Part + kAW * kRows * kLG;            // [32][33]   dropout(gelu(S1))
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 31, lk = lane >> 5;
    const int row0 = blockIdx.x * kRows;
    // Every global load of the kernel is issued here, before the first wait: the three W2 blocks and the x rows /
    // biases of this wave's output tiles, b1 and the dropout uniforms of this thread's four S1 elements.
    constexpr int NTILES = D / 32, NT = (NTILES + kAW - 1) / kAW;   // output column tiles, and per wave (tile = wave + kAW * tt)
    float4 w2v[NT][4];
    float xv[NT][16], b2v[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        const int n0 = min(wave + kAW * tt, NTILES - 1) * 32;        // waves past the last tile load a clamped tile and skip it below
#pragma unroll
        for (int it = 0; it < 4; ++it) w2v[tt][it] = *reinterpret_cast<const float4 *>(W2 + (size_t)n0 * kH + (lane + it * 64) * 4);
        b2v[tt] = b2[n0 + lr];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + trow(r, lk);
            xv[tt][r] = row < R ? x[(size_t)row * D + n0 + lr] : 0.0f;
        }
    }
    constexpr int EQ = kRows * kH / kAT;            // S1 elements per thread
    float b1v[EQ], uv[EQ];
#pragma unroll
    for (int q = 0; q < EQ; ++q) {
        const int e = threadIdx.x + q * kAT, i = e >> 5, j = e & 31;
        b1v[q] = b1[j];
        uv[q] = (u && row0 + i < R) ? u[(size_t)(row0 + i) * kH + j] : 1.0f;
    }
    stage_tile<D>(Hs, ha, row0, R);
    stage_tile<D>(W1s, W1, 0, kH);
    __syncthreads();
    {   // S1 partial over k in [wave*KW, (wave+1)*KW)
        f32x16 acc; zero16(acc);
        const float *hrow = Hs + lr * LDH + wave * KW + lk;
        const float *wrow = W1s + lr * LDH + wave * KW + lk;           // B[k][j] = W1[j][k]
        mfma_chain<KW / 2>(acc, [&](int q) { return hrow[2 * q]; }, [&](int q) { return wrow[2 * q]; });
#pragma unroll
        for (int r = 0; r < 16; ++r) Part[(wave * kRows + trow(r, lk)) * kLG + lr] = acc[r];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < EQ; ++q) {
        const int e = threadIdx.x + q * kAT, i = e >> 5, j = e & 31;
        float sv = 0.0f;
#pragma unroll
        for (int w = 0; w < kAW; w += 2) sv += Part[(w * kRows + i) * kLG + j] + Part[((w + 1) * kRows + i) * kLG + j];   // fixed order
        sv += b1v[q];
        float gq = 0.0f;
        if (row0 + i < R) {
            s1_out[(size_t)(row0 + i) * kH + j] = sv;
            gq = gelu_f(sv) * (u ? (uv[q] >= p ? 1.0f / (1.0f - p) : 0.0f) : 1.0f);
        }
        Gs[i * kLG + j] = gq;
    }
    __syncthreads();
    float *W2t = Part + wave * kRows * kLG;          // Part is free after the reduction above
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {                // out tile columns [n0, n0+32)
        if (wave + kAW * tt >= NTILES) break;        // wave-uniform
        const int n0 = (wave + kAW * tt) * 32;
#pragma unroll
        for (int it = 0; it < 4; ++it) {             // this wave's W2 block -> its LDS tile [n][j] (row stride 33)
            const int e = (lane + it * 64) * 4;
            float *d = W2t + (e >> 5) * kLG + (e & 31);
            d[0] = w2v[tt][it].x; d[1] = w2v[tt][it].y; d[2] = w2v[tt][it].z; d[3] = w2v[tt][it].w;
        }
        __builtin_amdgcn_wave_barrier();
        f32x16 acc; zero16(acc);
        const float *grow = Gs + lr * kLG + lk;
        const float *wrow = W2t + lr * kLG + lk;                       // B[k][n] = W2[n][k]
        mfma_chain<kH / 2>(acc, [&](int q) { return grow[2 * q]; }, [&](int q) { return wrow[2 * q]; });
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + trow(r, lk);
            if (row < R) out[(size_t)row * D + n0 + lr] = xv[tt][r] + scale * (acc[r] + b2v[tt]);
        }
        __builtin_amdgcn_wave_barrier();             // the tile is rewritten in the next iteration
    }
}

// backward.  part layout per workgroup b: dW1 [32][D] | dW2 [D][32] | db1 [32] | db2 [D]
template <int D>
__global__ __launch_bounds__(kAT) void adapter_bwd_kernel(const float *__restrict__ g_out, const float *__restrict__ ha,
                                                          const float *__restrict__ s1, const float *__restrict__ W1,
                                                          const float *__restrict__ W2, const float *__restrict__ u, float p,
                                                          float scale, float *__restrict__ g_ha, float *__restrict__ part, int R,
                                                          const float *__restrict__ ln_mean, const float *__restrict__ ln_rstd,
                                                          const float *__restrict__ ln_gamma, const float *__restrict__ ln_beta) {
    // ln_mean != null: `ha` holds the un-normalised rows saved by ln_adapter_fwd_kernel; the LayerNorm output is rebuilt here
    constexpr int LDH = D + 1, KW = D / kAW;
    extern __shared__ float sm[];
    float *Zs = sm;                                  // [32][D+1]  gz = scale * g_out
    float *Hs = Zs + kRows * LDH;                    // [32][D+1]  ha
    float *Part = Hs + kRows * LDH;                  // [kAW][32][33]
    float *GAs = Part + kAW * kRows * kLG;           // [32][33]   g_a1
    float *Ds = GAs + kRows * kLG;                   // [32][33]   d = dropout(gelu(S1))
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 31, lk = lane >> 5;
    const int row0 = blockIdx.x * kRows;
    float *pw1 = part + (size_t)blockIdx.x * (2 * kH * D + kH + D);
    float *pw2 = pw1 + kH * D, *pb1 = pw2 + D * kH, *pb2 = pb1 + kH;
    constexpr int EQ = kRows * kH / kAT;
    float s1v[EQ], uv[EQ];                           // loads for the element-wise stage, issued before the first wait
#pragma unroll
    for (int q = 0; q < EQ; ++q) {
        const int e = threadIdx.x + q * kAT, i = e >> 5, j = e & 31;
        const bool ok = row0 + i < R;
        s1v[q] = ok ? s1[(size_t)(row0 + i) * kH + j] : 0.0f;
        uv[q] = (u && ok) ? u[(size_t)(row0 + i) * kH + j] : 1.0f;
    }
    // weight operands of the two products that read W2 / W1, held in registers from the start (an operand fetched
    // from global memory inside an MFMA chain is one exposed round trip per chunk)
    constexpr int NTILES = D / 32, NT = (NTILES + kAW - 1) / kAW;
    float w2r[KW / 2], w1r[NT][kH / 2];
    {
        const float *wcol = W2 + (size_t)(wave * KW + lk) * kH + lr;
#pragma unroll
        for (int q = 0; q < KW / 2; ++q) w2r[q] = wcol[(size_t)2 * q * kH];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const float *wrow = W1 + (size_t)lk * D + min(wave + kAW * tt, NTILES - 1) * 32 + lr;
#pragma unroll
            for (int q = 0; q < kH / 2; ++q) w1r[tt][q] = wrow[(size_t)2 * q * D];
        }
    }
    stage_tile<D>(Zs, g_out, row0, R, scale);
    if (ln_mean) stage_tile_ln<D>(Hs, ha, ln_mean, ln_rstd, ln_gamma, ln_beta, row0, R);
    else stage_tile<D>(Hs, ha, row0, R);
    __syncthreads();
    {   // gd partial = gz . W2 over n in [wave*KW, (wave+1)*KW):  A[i][k=n] = gz[i][n], B[k=n][j] = W2[n][j]
        f32x16 acc; zero16(acc);
        const float *zrow = Zs + lr * LDH + wave * KW + lk;
        mfma_chain<KW / 2>(acc, [&](int q) { return zrow[2 * q]; }, [&](int q) { return w2r[q]; });
#pragma unroll
        for (int r = 0; r < 16; ++r) Part[(wave * kRows + trow(r, lk)) * kLG + lr] = acc[r];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < EQ; ++q) {
        const int e = threadIdx.x + q * kAT, i = e >> 5, j = e & 31;
        float gd = 0.0f;
#pragma unroll
        for (int w = 0; w < kAW; w += 2) gd += Part[(w * kRows + i) * kLG + j] + Part[((w + 1) * kRows + i) * kLG + j];   // fixed order
        float ga = 0.0f, d = 0.0f;
        if (row0 + i < R) {
            const float f = u ? (uv[q] >= p ? 1.0f / (1.0f - p) : 0.0f) : 1.0f;
            ga = gd * f * gelu_grad(s1v[q]);
            d = gelu_f(s1v[q]) * f;
        }
        GAs[i * kLG + j] = ga;
        Ds[i * kLG + j] = d;
    }
    __syncthreads();
    // bias partials: db1[j] = sum_i ga[i][j], db2[n] = sum_i gz[i][n]   (row order: deterministic)
    for (int c = threadIdx.x; c < kH + D; c += kAT) {
        float sacc = 0.0f;
        if (c < kH) { for (int i = 0; i < kRows; ++i) sacc += GAs[i * kLG + c]; pb1[c] = sacc; }
        else { const int n = c - kH; for (int i = 0; i < kRows; ++i) sacc += Zs[i * LDH + n]; pb2[n] = sacc; }
    }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        if (wave + kAW * tt >= NTILES) break;        // wave-uniform
        const int n0 = (wave + kAW * tt) * 32;
        {   // g_ha[i][n] = sum_j ga[i][j] W1[j][n]
            f32x16 acc; zero16(acc);
            const float *arow = GAs + lr * kLG + lk;
            mfma_chain<kH / 2>(acc, [&](int q) { return arow[2 * q]; }, [&](int q) { return w1r[tt][q]; });
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + trow(r, lk);
                if (row < R) g_ha[(size_t)row * D + n0 + lr] = acc[r];
            }
        }
        {   // dW2[n][j] = sum_i gz[i][n] d[i][j]:  A[n][k=i] = gz[i][n], B[k=i][j] = d[i][j]
            f32x16 acc; zero16(acc);
            const float *acol = Zs + lk * LDH + n0 + lr;
            const float *brow = Ds + lk * kLG + lr;
            mfma_chain<kRows / 2>(acc, [&](int q) { return acol[2 * q * LDH]; }, [&](int q) { return brow[2 * q * kLG]; });
#pragma unroll
            for (int r = 0; r < 16; ++r) pw2[(size_t)(n0 + trow(r, lk)) * kH + lr] = acc[r];
        }
        {   // dW1[j][n] = sum_i ga[i][j] ha[i][n]:  A[j][k=i] = ga[i][j], B[k=i][n] = ha[i][n]
            f32x16 acc; zero16(acc);
            const float *acol = GAs + lk * kLG + lr;
            const float *brow = Hs + lk * LDH + n0 + lr;
            mfma_chain<kRows / 2>(acc, [&](int q) { return acol[2 * q * kLG]; }, [&](int q) { return brow[2 * q * LDH]; });
#pragma unroll
            for (int r = 0; r < 16; ++r) pw1[(size_t)trow(r, lk) * D + n0 + lr] = acc[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused tail of a block (round 2): the row kernel that closes the MLP residual, strips the prompts and applies the adapter's
// LayerNorm, and the adapter itself, in ONE launch on 16-row workgroups (v_mfma_f32_16x16x4_f32):
//     v   = x[src(t)] + dp_scale(u_b) * (y[src(t)] + ybias)          (rows of the block's output stream, saved: xo)
//     ha  = LayerNorm(v)                                             (never written to memory; mean / rstd saved)
//     out = v + scale * (W2 . dropout(gelu(W1 . ha + b1)) + b2)
// The separate launches (rowln_fwd 6.0 us + adapter_fwd 11.2 us at 2,400 rows) ran 600 and 75 workgroups; the adapter's 32-row
// workgroups left 181 CUs idle behind a chain of dependent phases.  Here 150 workgroups of 4 waves: wave w normalises rows
// 4 w .. 4 w + 3 (lane -> columns lane + 64 e, the arithmetic of rowln_fwd_kernel: bit-identical rows and statistics), then
// takes the k-range [96 w, 96 w + 96) of S1 = ha . W1^T (two 16 x 16 tiles) and the six column tiles 6 w .. 6 w + 5 of the
// second product.  Operands: ha / G from LDS rows padded to stride = 4 (mod 64) floats (16-byte reads, conflict-free:
// lane (r = lane & 15, g = lane >> 4) reads row r, floats 16 i + 4 g .. + 3, one read feeds four MFMAs -- the contraction
// order inside a 16-block is free); W1 / W2 rows straight from L2 as 16-byte pieces in the same k order, all issued at
// kernel entry.  The result goes back through the LDS so that the 16 x 384 tile leaves as full 1,536-byte rows.
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int kFR = 16;                 // rows per workgroup
__device__ __forceinline__ float comp4(const float4 &v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

struct LnAdapterArgs {
    const float *x, *y, *ybias, *u;     // (B, Lin, D) stream, residual branch (or null), its frozen bias (or null), drop-path uniforms
    float keep;
    int mode, P;                        // row map: 0 identity, 3 / 4 strip P prompts (behind the cls token / leading)
    const float *gamma, *beta;
    float eps;
    const float *W1, *b1, *W2, *b2, *ud;  // adapter; ud: (R, 32) dropout uniforms or null
    float p, scale;
    float *xo, *mean, *rstd, *s1, *out;
    int B, Lin, Lout;
    int yparts; long long ystride;      // y = sum of yparts partial matrices ystride floats apart (upp_linear_parts_f32), added in order
};

template <int D, int kFW>
__global__ __launch_bounds__(64 * kFW) void ln_adapter_fwd_kernel(LnAdapterArgs a) {
    constexpr int LDH = D + 4;          // 388 = 4 (mod 64): 16-byte operand reads of 16 rows cover all banks
    constexpr int LDG = kH + 4;         // 36: same property for the 32-wide G rows
    constexpr int E = D / 64;           // elements per lane and row
    constexpr int KW = D / kFW;         // k-range per wave of the first product (96)
    constexpr int NI = KW / 16;         // 16-blocks of k per wave (6)
    constexpr int NT = D / 16 / kFW;    // output column tiles per wave (6)
    constexpr int RW = kFR / kFW;       // rows per wave in the row phase
    static_assert(D % 64 == 0 && KW % 16 == 0 && (D / 16) % kFW == 0, "shape");
    __shared__ __attribute__((aligned(16))) float Hs[kFR * LDH];        // ha
    __shared__ __attribute__((aligned(16))) float Vs[kFR * LDH];        // v, later the output tile
    __shared__ __attribute__((aligned(16))) float Part[kFW * kFR * (kH + 1)];
    __shared__ __attribute__((aligned(16))) float Gs[kFR * LDG];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 15, g = lane >> 4;
    const int R = a.B * a.Lout;
    const int row0 = xcd_contiguous(blockIdx.x, gridDim.x) * kFR;

    // ---- weight operands, issued first (they do not depend on anything this workgroup computes)
    float4 w1v[2][NI], w2v[NT][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < NI; ++i)
            w1v[t][i] = *reinterpret_cast<const float4 *>(a.W1 + (size_t)(16 * t + r) * D + wave * KW + 16 * i + 4 * g);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            w2v[t][i] = *reinterpret_cast<const float4 *>(a.W2 + (size_t)(16 * (wave * NT + t) + r) * kH + 16 * i + 4 * g);

    // ---- rows: residual, prompt strip, LayerNorm (the arithmetic of rowln_fwd_kernel)
    {
        float gv[E], bv[E], ybv[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            gv[e] = a.gamma[lane + 64 * e];
            bv[e] = a.beta[lane + 64 * e];
            ybv[e] = a.ybias ? a.ybias[lane + 64 * e] : 0.0f;
        }
        float xv[RW][E], yv[RW][E];
        int rowi[RW], bi[RW];
#pragma unroll
        for (int q = 0; q < RW; ++q) {
            const int row = min(row0 + wave * RW + q, R - 1);
            const int b = row / a.Lout, t = row - b * a.Lout;
            const int src = a.mode == 3 ? (t == 0 ? 0 : t + a.P) : (a.mode == 4 ? t + a.P : t);
            const size_t off = ((size_t)b * a.Lin + src) * D;
            rowi[q] = row; bi[q] = b;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                xv[q][e] = a.x[off + lane + 64 * e];
                yv[q][e] = a.y ? a.y[off + lane + 64 * e] : 0.0f;
            }
            for (int pp = 1; pp < a.yparts; ++pp) {      // the k-parts of the GEMM that produced y
                float yp[E];
#pragma unroll
                for (int e = 0; e < E; ++e) yp[e] = a.y[(size_t)pp * a.ystride + off + lane + 64 * e];
#pragma unroll
                for (int e = 0; e < E; ++e) yv[q][e] += yp[e];
            }
        }
#pragma unroll
        for (int q = 0; q < RW; ++q) {
            const int rr = wave * RW + q;
            const bool live = row0 + rr < R;
            const float sc = a.y ? (a.u ? floorf(a.keep + a.u[bi[q]]) / a.keep : 1.0f) : 0.0f;
            float v[E];
            float s = 0.0f;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                float val = xv[q][e];
                if (a.y) val = __builtin_fmaf(yv[q][e] + ybv[e], sc, val);
                v[e] = val;
                s += val;
            }
            const float mean = wave_sum_f32(s) / (float)D;
            float qq = 0.0f;
#pragma unroll
            for (int e = 0; e < E; ++e) { const float dv = v[e] - mean; qq = __builtin_fmaf(dv, dv, qq); }
            const float rstd = 1.0f / sqrtf(wave_sum_f32(qq) / (float)D + a.eps);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float h = __builtin_fmaf((v[e] - mean) * rstd, gv[e], bv[e]);
                Hs[rr * LDH + lane + 64 * e] = live ? h : 0.0f;
                Vs[rr * LDH + lane + 64 * e] = v[e];
                if (live) a.xo[(size_t)rowi[q] * D + lane + 64 * e] = v[e];
            }
            if (live && lane == 0) { a.mean[rowi[q]] = mean; a.rstd[rowi[q]] = rstd; }
        }
    }
    __syncthreads();

    // ---- S1 partial over k in [wave KW, wave KW + KW): two 16 x 16 tiles (hidden units 0..15 | 16..31)
    {
        f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float4 av = *reinterpret_cast<const float4 *>(&Hs[r * LDH + wave * KW + 16 * i + 4 * g]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(comp4(av, c), comp4(w1v[0][i], c), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(comp4(av, c), comp4(w1v[1][i], c), acc1, 0, 0, 0);
            }
        }
        // acc[reg] = S1[row 4 g + reg][hidden 16 t + r]
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            Part[(wave * kFR + 4 * g + reg) * (kH + 1) + r] = acc0[reg];
            Part[(wave * kFR + 4 * g + reg) * (kH + 1) + 16 + r] = acc1[reg];
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kFR * kH / (64 * kFW); ++q) {
        const int e = threadIdx.x + q * 64 * kFW, i = e >> 5, jn = e & 31;
        float sv = 0.0f;
#pragma unroll
        for (int w = 0; w < kFW; w += 2) sv += Part[(w * kFR + i) * (kH + 1) + jn] + Part[((w + 1) * kFR + i) * (kH + 1) + jn];   // fixed order
        sv += a.b1[jn];
        float gq = 0.0f;
        if (row0 + i < R) {
            a.s1[(size_t)(row0 + i) * kH + jn] = sv;
            const float f = a.ud ? (a.ud[(size_t)(row0 + i) * kH + jn] >= a.p ? 1.0f / (1.0f - a.p) : 0.0f) : 1.0f;
            gq = gelu_f(sv) * f;
        }
        Gs[i * LDG + jn] = gq;
    }
    __syncthreads();

    // ---- out tile columns 16 (wave NT + t) .. + 15:  Z = G . W2^T, out = v + scale (Z + b2), written over v in the LDS
    {
        const float4 g0 = *reinterpret_cast<const float4 *>(&Gs[r * LDG + 4 * g]);
        const float4 g1 = *reinterpret_cast<const float4 *>(&Gs[r * LDG + 16 + 4 * g]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(comp4(g0, c), comp4(w2v[t][0], c), acc, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(comp4(g1, c), comp4(w2v[t][1], c), acc, 0, 0, 0);
            const int col = 16 * (wave * NT + t) + r;
            const float b2 = a.b2[col];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                float *d = &Vs[(4 * g + reg) * LDH + col];
                *d = *d + a.scale * (acc[reg] + b2);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kFR * (D / 4) / (64 * kFW); ++q) {
        const int idx = threadIdx.x + q * 64 * kFW, i = idx / (D / 4), c4 = idx - i * (D / 4);
        if (row0 + i < R)
            *reinterpret_cast<float4 *>(a.out + (size_t)(row0 + i) * D + 4 * c4) = *reinterpret_cast<const float4 *>(&Vs[i * LDH + 4 * c4]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the fused block tail in ONE launch on 16-row workgroups (was adapter_bwd_kernel on 32-row workgroups, 75 of them,
// 19 us, followed by rowln_bwd_kernel, 5.6 us):
//   gd = scale (g_out . W2)                      split 8 ways over k = the 384 output columns, reduced through the LDS
//   ga = gd * dropout' * gelu'(s1),  d = scale * dropout(gelu(s1))
//   g_ha = ga . W1                               3 column tiles per wave, into an LDS tile
//   dW2 += g_out^T d,  dW1 += ga^T ha            per-workgroup partials (contraction over the 16 rows: 4 k-steps per 16x16 tile)
//   db1 += col-sum ga,  db2 += scale col-sum g_out
//   LayerNorm backward of the adapter's LayerNorm + the residual:  d_row = g_out + rstd (dy - mean(dy) - xhat mean(dy xhat)),
//   dy = g_ha gamma;  g_x[src row] = d_row,  g_y[src row] = dp_scale d_row;  d_gamma / d_beta partials per workgroup;
//   the prompt rows a strip map dropped get their zero gradient from the same launch.
// ha is rebuilt from the saved rows and statistics (the forward does not store it).  LDS tiles of g_out and ha use a row stride
// of 400 floats (= 16 mod 64): the products over the ROWS read them column-wise (lane (r, g) -> row 4 s + g, column 16 t + r), which
// that stride makes conflict-free; their few 16-byte row reads take a 4-way conflict instead.
struct LnAdapterBwdArgs {
    const float *g_out, *xo, *mean, *rstd, *gamma, *beta, *s1, *W1, *W2, *ud, *u;
    float p, scale, keep;
    int mode, P;
    float *g_x, *g_y, *part, *ln_part;     // part: [workgroup][dW1 (32,D) | dW2 (D,32) | db1 (32) | db2 (D)];  ln_part: [workgroup][2][D]
    int B, Lin, Lout;
    float *fac;                            // or (part == null): the FACTORS of the weight gradients per row, [ga (32) | d (32)] -- adapter_wgrad_kernel
};

template <int D, int NW>
__global__ __launch_bounds__(64 * NW) void ln_adapter_bwd_kernel(LnAdapterBwdArgs a) {
    constexpr int LDC = 400, LDR = D + 4, LDG = 48, E = D / 64, RW = kFR / NW;
    constexpr int KW = D / NW, NI = KW / 16, NT = D / 16 / NW;     // k-range of gd per wave (48), its 16-blocks (3), column tiles per wave (3)
    constexpr int TW = 2 * (D / 16) / NW;                          // weight-gradient tiles per wave and matrix (6)
    static_assert(kFR % NW == 0 && KW % 16 == 0 && (D / 16) % NW == 0 && (2 * (D / 16)) % NW == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *Zs = sm;                        // [16][400]  g_out
    float *Hs = Zs + kFR * LDC;            // [16][400]  ha; later the LayerNorm parameter-gradient rows of the waves
    float *Gh = Hs + kFR * LDC;            // [16][388]  g_ha
    float *Part = Gh + kFR * LDR;          // [NW][16][33]
    float *GAs = Part + NW * kFR * (kH + 1);   // [16][48]  ga
    float *Ds = GAs + kFR * LDG;           // [16][48]  scale * dropout(gelu(s1))
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 15, g = lane >> 4;
    const int R = a.B * a.Lout;
    const int wg = xcd_contiguous(blockIdx.x, gridDim.x);
    const int row0 = wg * kFR;
    const size_t psz = (size_t)2 * kH * D + kH + D;
    float *pw1 = a.part ? a.part + (size_t)wg * psz : nullptr;
    float *pw2 = pw1 ? pw1 + kH * D : nullptr, *pb1 = pw1 ? pw2 + D * kH : nullptr, *pb2 = pw1 ? pb1 + kH : nullptr;

    // ---- weight operands of the two products over the feature dimension, one dword per lane and k-step, all issued first
    float w2r[2][NI][4], w1r[NT][2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) w2r[t][i][c] = a.W2[(size_t)(wave * KW + 16 * i + 4 * g + c) * kH + 16 * t + r];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) w1r[t][i][c] = a.W1[(size_t)(16 * i + 4 * g + c) * D + 16 * (wave * NT + t) + r];

    // ---- rows: g_out and ha = LayerNorm(rows) into the LDS
    int rowi[RW];
    float mu[RW], rs[RW];
    {
        float gm[E], bt[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { gm[e] = a.gamma[lane + 64 * e]; bt[e] = a.beta[lane + 64 * e]; }
        float gv[RW][E], xv[RW][E];
#pragma unroll
        for (int q = 0; q < RW; ++q) {
            rowi[q] = min(row0 + wave * RW + q, R - 1);
            mu[q] = a.mean[rowi[q]]; rs[q] = a.rstd[rowi[q]];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                gv[q][e] = a.g_out[(size_t)rowi[q] * D + lane + 64 * e];
                xv[q][e] = a.xo[(size_t)rowi[q] * D + lane + 64 * e];
            }
        }
#pragma unroll
        for (int q = 0; q < RW; ++q) {
            const int rr = wave * RW + q;
            const bool live = row0 + rr < R;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                Zs[rr * LDC + lane + 64 * e] = live ? gv[q][e] : 0.0f;
                Hs[rr * LDC + lane + 64 * e] = live ? __builtin_fmaf((xv[q][e] - mu[q]) * rs[q], gm[e], bt[e]) : 0.0f;
            }
        }
    }
    __syncthreads();

    // ---- gd partial over the columns [wave KW, wave KW + KW): A = g_out rows, B[k = column][j] = W2[column][j]
    {
        f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float4 av = *reinterpret_cast<const float4 *>(&Zs[r * LDC + wave * KW + 16 * i + 4 * g]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(comp4(av, c), w2r[0][i][c], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(comp4(av, c), w2r[1][i][c], acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            Part[(wave * kFR + 4 * g + reg) * (kH + 1) + r] = acc0[reg];
            Part[(wave * kFR + 4 * g + reg) * (kH + 1) + 16 + r] = acc1[reg];
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kFR * kH / (64 * NW); ++q) {
        const int e = threadIdx.x + q * 64 * NW, i = e >> 5, jn = e & 31;
        float gd = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; w += 2) gd += Part[(w * kFR + i) * (kH + 1) + jn] + Part[((w + 1) * kFR + i) * (kH + 1) + jn];   // fixed order
        gd *= a.scale;
        float ga = 0.0f, d = 0.0f;
        if (row0 + i < R) {
            const float sv = a.s1[(size_t)(row0 + i) * kH + jn];
            const float f = a.ud ? (a.ud[(size_t)(row0 + i) * kH + jn] >= a.p ? 1.0f / (1.0f - a.p) : 0.0f) : 1.0f;
            ga = gd * f * gelu_grad(sv);
            d = gelu_f(sv) * f * a.scale;
        }
        GAs[i * LDG + jn] = ga;
        Ds[i * LDG + jn] = d;
        if (a.fac && row0 + i < R) { a.fac[(size_t)(row0 + i) * (2 * kH) + jn] = ga; a.fac[(size_t)(row0 + i) * (2 * kH) + kH + jn] = d; }
    }
    __syncthreads();

    // ---- g_ha tiles (A = ga rows, B = W1 rows) into Gh
    {
        const float4 g0 = *reinterpret_cast<const float4 *>(&GAs[r * LDG + 4 * g]);
        const float4 g1 = *reinterpret_cast<const float4 *>(&GAs[r * LDG + 16 + 4 * g]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(comp4(g0, c), w1r[t][0][c], acc, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(comp4(g1, c), w1r[t][1][c], acc, 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Gh[(4 * g + reg) * LDR + 16 * (wave * NT + t) + r] = acc[reg];
        }
    }
    // ---- weight-gradient partials: contraction over the 16 rows (4 k-steps); tile id -> (tn = column tile of D, tj = half of the 32)
    if (pw1) {
#pragma unroll
        for (int q = 0; q < TW; ++q) {
            const int tile = wave * TW + q, tn = tile >> 1, tj = tile & 1;
            f32x4v acc2 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float zc = Zs[(4 * s + g) * LDC + 16 * tn + r];      // g_out[row][n]   (A of dW2, rows = n)
                const float dc = Ds[(4 * s + g) * LDG + 16 * tj + r];      // d[row][j]       (B of dW2)
                const float gc = GAs[(4 * s + g) * LDG + 16 * tj + r];     // ga[row][j]      (A of dW1, rows = j)
                const float hc = Hs[(4 * s + g) * LDC + 16 * tn + r];      // ha[row][n]      (B of dW1)
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(zc, dc, acc2, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(gc, hc, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                pw2[(size_t)(16 * tn + 4 * g + reg) * kH + 16 * tj + r] = acc2[reg];     // dW2[n][j]
                pw1[(size_t)(16 * tj + 4 * g + reg) * D + 16 * tn + r] = acc1[reg];      // dW1[j][n]
            }
        }
        // bias partials: db1[j] = sum_i ga[i][j], db2[n] = scale sum_i g_out[i][n]   (row order: deterministic)
        for (int c = threadIdx.x; c < kH + D; c += 64 * NW) {
            float sacc = 0.0f;
            if (c < kH) { for (int i = 0; i < kFR; ++i) sacc += GAs[i * LDG + c]; pb1[c] = sacc; }
            else { const int n = c - kH; for (int i = 0; i < kFR; ++i) sacc += Zs[i * LDC + n]; pb2[n] = sacc * a.scale; }
        }
    }
    __syncthreads();

    // ---- LayerNorm backward + residual, row by row (the arithmetic of rowln_bwd_kernel); parameter-gradient rows into Hs
    float *lnp = Hs;                                       // [NW][2][D] (ha is no longer needed)
    {
        float gm[E], pg[E], pb[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { gm[e] = a.gamma[lane + 64 * e]; pg[e] = 0.0f; pb[e] = 0.0f; }
#pragma unroll
        for (int q = 0; q < RW; ++q) {
            const int rr = wave * RW + q;
            if (row0 + rr >= R) continue;                  // (wave-uniform)
            const int row = rowi[q];
            const int b = row / a.Lout, t = row - b * a.Lout;
            const int src = a.mode == 3 ? (t == 0 ? 0 : t + a.P) : (a.mode == 4 ? t + a.P : t);
            float xo[E], gh[E], go[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                xo[e] = a.xo[(size_t)row * D + lane + 64 * e];
                gh[e] = Gh[rr * LDR + lane + 64 * e];
                go[e] = Zs[rr * LDC + lane + 64 * e];
            }
            float dy[E], xh[E], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                dy[e] = gh[e] * gm[e];
                xh[e] = (xo[e] - mu[q]) * rs[q];
                pg[e] += gh[e] * xh[e];
                pb[e] += gh[e];
                s1 += dy[e];
                s2 = __builtin_fmaf(dy[e], xh[e], s2);
            }
            s1 = wave_sum_f32(s1) / (float)D;
            s2 = wave_sum_f32(s2) / (float)D;
            const float sc = a.g_y ? (a.u ? floorf(a.keep + a.u[b]) / a.keep : 1.0f) : 0.0f;
            float *gx = a.g_x ? a.g_x + ((size_t)b * a.Lin + src) * D : nullptr;
            float *gy = a.g_y ? a.g_y + ((size_t)b * a.Lin + src) * D : nullptr;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float dv = go[e] + rs[q] * (dy[e] - s1 - xh[e] * s2);
                if (gx) gx[lane + 64 * e] = dv;
                if (gy) gy[lane + 64 * e] = dv * sc;
            }
        }
        if (a.ln_part) {                                   // (the barrier in front of this phase: every wave is done with ha)
#pragma unroll
            for (int e = 0; e < E; ++e) { lnp[(wave * 2 + 0) * D + lane + 64 * e] = pg[e]; lnp[(wave * 2 + 1) * D + lane + 64 * e] = pb[e]; }
            __syncthreads();
            for (int c = threadIdx.x; c < 2 * D; c += 64 * NW) {
                const int which = c / D, col = c - which * D;
                float sacc = 0.0f;
#pragma unroll
                for (int w = 0; w < NW; w += 2) sacc += lnp[(w * 2 + which) * D + col] + lnp[((w + 1) * 2 + which) * D + col];   // wave order
                a.ln_part[((size_t)wg * 2 + which) * D + col] = sacc;
            }
        }
    }
    // ---- zero gradient for the prompt rows the forward's strip map dropped: B P rows, dealt round-robin to the workgroups
    if (a.P > 0 && (a.mode == 3 || a.mode == 4)) {
        const int first = a.mode == 3 ? 1 : 0;
        for (int z = blockIdx.x; z < a.B * a.P; z += gridDim.x) {
            const int b = z / a.P, t = first + (z - b * a.P);
            const size_t off = ((size_t)b * a.Lin + t) * D;
            for (int c = threadIdx.x; c < D; c += 64 * NW) {
                if (a.g_x) a.g_x[off + c] = 0.0f;
                if (a.g_y) a.g_y[off + c] = 0.0f;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The adapter weight gradients from their FACTORS (round 6).  ln_adapter_bwd_kernel forms dW1 / dW2 per 16-row workgroup -- 24,992
// floats of partial results per workgroup, 130 ... 258 workgroups per block: 153 MB written and read again per headline step, 302 MB
// per segmentation step, most of what upp_batched_sum moves.  With `fac` it writes only ga / d (64 floats per row); this kernel, ONE
// launch for all blocks of a backward pass, then contracts over many rows per workgroup:
//     dW1 (32, D) = sum_rows ga^T . ha,   dW2 (D, 32) = sum_rows g_out^T . d,   db1 = sum_rows ga,   db2 = scale sum_rows g_out
// with ha = LayerNorm(xo) rebuilt from the saved rows and statistics (the expression of the forward).  Workgroup = (job, 64-column tile
// of D, row split): 16 rows per step through two LDS buffers (the next step's rows are in flight while this one is multiplied), wave w
// owns columns 16 w .. 16 w + 15 of the tile: two 16 x 16 tiles of each matrix on v_mfma_f32_16x16x4_f32.  part[job]: `splits` rows of
// the per-workgroup layout above, summed by upp_batched_sum.  Rows ascending inside a split, splits ascending in the sum: deterministic.
constexpr int kWgJobs = 16, kWgLd = 80;       // LDS row stride = 16 mod 64 floats: the column-wise operand reads (lane (r, g) -> row 4 s + g, column 16 t + r) are conflict-free
struct AdapterWgradJobs {
    const float *xo[kWgJobs], *mean[kWgJobs], *rstd[kWgJobs], *gamma[kWgJobs], *beta[kWgJobs], *g_out[kWgJobs], *fac[kWgJobs];
    float *part[kWgJobs];
    int R[kWgJobs];
    float scale[kWgJobs];
    int splits;
};
template <int D>
__global__ __launch_bounds__(256) void adapter_wgrad_kernel(AdapterWgradJobs a) {
    __shared__ __attribute__((aligned(16))) float Zs[2][kFR * kWgLd], Hs[2][kFR * kWgLd], Fs[2][kFR * kWgLd];   // g_out, ha, [ga | d]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    int bx = blockIdx.x;
    const int per_job = (D / 64) * a.splits;
    const int j = bx / per_job;
    bx -= j * per_job;
    const int sp = bx / (D / 64), t = bx - sp * (D / 64);       // the six column tiles of a row range are neighbours: together they read whole rows
    const int R = a.R[j];
    const int per = ((R + a.splits - 1) / a.splits + kFR - 1) / kFR * kFR;
    const int r_lo = sp * per, r_hi = min(R, r_lo + per);
    const int c0 = 64 * t;
    const float *xo = a.xo[j], *go = a.g_out[j], *fac = a.fac[j], *mean = a.mean[j], *rstd = a.rstd[j];
    const int lr = tid >> 4, c4 = (tid & 15) * 4;          // this thread's 16-byte piece of a 16 x 64 tile
    const float4 gm = *reinterpret_cast<const float4 *>(a.gamma[j] + c0 + c4), bt = *reinterpret_cast<const float4 *>(a.beta[j] + c0 + c4);
    // (plain statements, component-wise selects: a lambda that captured the staging registers by reference and selected whole float4s put
    //  them into scratch memory -- 80 bytes of private segment, 31 of the kernel's 46 us)
    float4 vz, vx, vf;
    float mu = 0.0f, rs = 0.0f, keepf = 0.0f;
#define ADWG_LOAD(ROW0)                                                                        \
    {                                                                                          \
        const int row_ = min((ROW0) + lr, R - 1);                                              \
        keepf = (ROW0) + lr < r_hi ? 1.0f : 0.0f;                                              \
        vz = *reinterpret_cast<const float4 *>(go + (size_t)row_ * D + c0 + c4);               \
        vx = *reinterpret_cast<const float4 *>(xo + (size_t)row_ * D + c0 + c4);               \
        vf = *reinterpret_cast<const float4 *>(fac + (size_t)row_ * (2 * kH) + c4);            \
        mu = mean[row_]; rs = rstd[row_];                                                      \
    }
    // rows beyond the split: zeros (select, not multiply: the clamped row may hold anything)
#define ADWG_SEL(V) (keepf != 0.0f ? (V) : 0.0f)
#define ADWG_STORE(BUF)                                                                        \
    {                                                                                          \
        float *z_ = &Zs[BUF][lr * kWgLd + c4], *h_ = &Hs[BUF][lr * kWgLd + c4], *f_ = &Fs[BUF][lr * kWgLd + c4];   \
        const float h0 = __builtin_fmaf((vx.x - mu) * rs, gm.x, bt.x), h1 = __builtin_fmaf((vx.y - mu) * rs, gm.y, bt.y);   \
        const float h2 = __builtin_fmaf((vx.z - mu) * rs, gm.z, bt.z), h3 = __builtin_fmaf((vx.w - mu) * rs, gm.w, bt.w);   \
        *reinterpret_cast<float4 *>(z_) = make_float4(ADWG_SEL(vz.x), ADWG_SEL(vz.y), ADWG_SEL(vz.z), ADWG_SEL(vz.w));      \
        *reinterpret_cast<float4 *>(h_) = make_float4(ADWG_SEL(h0), ADWG_SEL(h1), ADWG_SEL(h2), ADWG_SEL(h3));              \
        *reinterpret_cast<float4 *>(f_) = make_float4(ADWG_SEL(vf.x), ADWG_SEL(vf.y), ADWG_SEL(vf.z), ADWG_SEL(vf.w));      \
    }
    f32x4v w1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, w2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float bsum = 0.0f;                                      // tid < 64: db2 of column c0 + tid; 64 <= tid < 96 (tile 0): db1 of unit tid - 64
    if (r_lo < r_hi) {
        ADWG_LOAD(r_lo)
        ADWG_STORE(0)
        __syncthreads();
        int buf = 0;
        for (int row0 = r_lo; row0 < r_hi; row0 += kFR, buf ^= 1) {
            const bool more = row0 + kFR < r_hi;
#ifndef UPP_ADWG_NO_LOAD          // (diagnostic builds, tools/micro/src/adwg_ablate.hip: wrong results)
            if (more) ADWG_LOAD(row0 + kFR)
#endif
            const float *Z = Zs[buf], *Hh = Hs[buf], *F = Fs[buf];
#ifndef UPP_ADWG_NO_COMPUTE
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int ro = (4 * s4 + g) * kWgLd;
                const float zc = Z[ro + 16 * wave + r];         // g_out[row][n]   (A of dW2, rows = n)
                const float hc = Hh[ro + 16 * wave + r];        // ha[row][n]      (B of dW1)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    const float gc = F[ro + 16 * tj + r];       // ga[row][j]      (A of dW1, rows = j)
                    const float dc = F[ro + kH + 16 * tj + r];  // d[row][j]       (B of dW2)
                    w2[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(zc, dc, w2[tj], 0, 0, 0);
                    w1[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(gc, hc, w1[tj], 0, 0, 0);
                }
            }
#endif
#ifdef UPP_ADWG_SKELETON           // (one LDS read per step keeps the staging alive)
            bsum += Z[tid] + Hh[tid] + F[tid];
#endif
#ifndef UPP_ADWG_NO_BIAS
            if (tid < 64) { for (int i = 0; i < kFR; ++i) bsum += Z[i * kWgLd + tid]; }
            else if (tid < 64 + kH && t == 0) { for (int i = 0; i < kFR; ++i) bsum += F[i * kWgLd + tid - 64]; }
#endif
            if (more) ADWG_STORE(buf ^ 1)
            __syncthreads();
        }
    }
    // ---- the tile leaves through the LDS: dW1 as 32 rows of 256 contiguous bytes, dW2 as ONE contiguous 8 KB block (64 x 32), 16 bytes per
    // lane (straight from the accumulators a wave-store is four 64-byte pieces: the 20 MB of partials took longer than the contraction)
    float *out = a.part[j] + (size_t)sp * ((size_t)2 * kH * D + kH + D);
    float *T1 = &Zs[0][0], *T2 = &Hs[0][0];                  // [32][68], [64][36]   (both buffers of a staging array: 2,560 floats each)
    static_assert(2 * kFR * kWgLd >= 32 * 68 && 2 * kFR * kWgLd >= 64 * 36, "staging tiles fit the operand buffers");
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            T1[(16 * tj + 4 * g + reg) * 68 + 16 * wave + r] = w1[tj][reg];                 // dW1[j][n - c0]
            T2[(16 * wave + 4 * g + reg) * 36 + 16 * tj + r] = w2[tj][reg];                 // dW2[n - c0][j]
        }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = tid + 256 * q;                         // 512 16-byte pieces per matrix
        const int r1 = e >> 4, p1 = (e & 15) * 4;            // dW1: row j = r1 (32 rows), 16 pieces of its 64 columns
        *reinterpret_cast<float4 *>(out + (size_t)r1 * D + c0 + p1) = *reinterpret_cast<const float4 *>(&T1[r1 * 68 + p1]);
        const int r2 = e >> 3, p2 = (e & 7) * 4;             // dW2: row n = c0 + r2 (64 rows), 8 pieces of its 32 columns
        *reinterpret_cast<float4 *>(out + (size_t)kH * D + (size_t)(c0 + r2) * kH + p2) = *reinterpret_cast<const float4 *>(&T2[r2 * 36 + p2]);
    }
    if (tid < 64) out[(size_t)2 * kH * D + kH + c0 + tid] = bsum * a.scale[j];
    else if (tid < 64 + kH && t == 0) out[(size_t)2 * kH * D + tid - 64] = bsum;
#undef ADWG_LOAD
#undef ADWG_SEL
#undef ADWG_STORE
}

template <typename K>
int raise_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

}  // namespace

extern "C" long long upp_adapter_part_floats(int R, int D) {
    if (R < 1 || D < 1) return 0;
    return (long long)((R + kRows - 1) / kRows) * (2LL * kH * D + kH + D);
}

extern "C" int upp_adapter_fwd(const float *ha, const float *x, const float *W1, const float *b1, const float *W2, const float *b2,
                               const float *u, float p, float scale, float *out, float *s1, int R, int D, int H, void *stream) {
    if (!ha || !x || !W1 || !b1 || !W2 || !b2 || !out || !s1 || R < 1) return UPP_E_BADARG;
    if (D != 384 || H != kH) return UPP_E_RANGE;
    const size_t lds = ((size_t)(kRows + kH) * (D + 1) + (kAW + 1) * kRows * kLG) * sizeof(float);
    static std::atomic<bool> raised{false};
    if (!raised) { int rc = raise_lds(adapter_fwd_kernel<384>, lds); if (rc) return rc; raised = true; }
    hipLaunchKernelGGL((adapter_fwd_kernel<384>), dim3((R + kRows - 1) / kRows), dim3(kAT), lds, (hipStream_t)stream, ha, x, W1, b1, W2,
                       b2, u, p, scale, out, s1, R);
    return upp_launch_status();
}

static int ln_adapter_fwd_impl(const float *x, const float *y, int yparts, long long ystride, const float *ybias, const float *u, float keep,
                                        int mode, int P, const float *gamma, const float *beta, float eps, const float *W1, const float *b1,
                                        const float *W2, const float *b2, const float *ud, float p, float scale, float *xo, float *mean,
                                        float *rstd, float *s1, float *out, int B, int Lin, int Lout, int D, int H, void *stream) {
    if (yparts < 1 || (yparts > 1 && (!y || ystride < (long long)B * Lin * D))) return UPP_E_BADARG;
    if (!x || !gamma || !beta || !W1 || !b1 || !W2 || !b2 || !xo || !mean || !rstd || !s1 || !out || B < 0 || Lin < 1 || Lout < 1)
        return UPP_E_BADARG;
    if (ybias && !y) return UPP_E_BADARG;
    if (D != 384 || H != kH) return UPP_E_RANGE;
    if (!(mode == 0 || mode == 3 || mode == 4) || P < 0 || (mode == 0 && Lout != Lin) || (mode != 0 && Lout != Lin - P)) return UPP_E_BADARG;
    if (B == 0) return 0;
    LnAdapterArgs a{x, y, ybias, u, keep, mode, P, gamma, beta, eps, W1, b1, W2, b2, ud, p, scale, xo, mean, rstd, s1, out, B, Lin, Lout, yparts, ystride};
    hipLaunchKernelGGL((ln_adapter_fwd_kernel<384, 8>), dim3((B * Lout + kFR - 1) / kFR), dim3(64 * 8), 0, (hipStream_t)stream, a);
    return upp_launch_status();
}

extern "C" int upp_ln_adapter_fwd(const float *x, const float *y, const float *ybias, const float *u, float keep, int mode, int P,
                                  const float *gamma, const float *beta, float eps, const float *W1, const float *b1, const float *W2,
                                  const float *b2, const float *ud, float p, float scale, float *xo, float *mean, float *rstd, float *s1,
                                  float *out, int B, int Lin, int Lout, int D, int H, void *stream) {
    return ln_adapter_fwd_impl(x, y, 1, 0, ybias, u, keep, mode, P, gamma, beta, eps, W1, b1, W2, b2, ud, p, scale, xo, mean, rstd, s1, out, B, Lin,
                                    Lout, D, H, stream);
}

static int adapter_bwd_launch(const float *g_out, const float *ha, const float *mean, const float *rstd, const float *gamma,
                              const float *beta, const float *s1, const float *W1, const float *W2, const float *u, float p, float scale,
                              float *g_ha, float *part, int R, int D, int H, void *stream) {
    if (!g_out || !ha || !s1 || !W1 || !W2 || !g_ha || !part || R < 1) return UPP_E_BADARG;
    if (D != 384 || H != kH) return UPP_E_RANGE;
    const size_t lds = ((size_t)2 * kRows * (D + 1) + (kAW + 2) * kRows * kLG) * sizeof(float);
    static std::atomic<bool> raised{false};
    if (!raised) { int rc = raise_lds(adapter_bwd_kernel<384>, lds); if (rc) return rc; raised = true; }
    hipLaunchKernelGGL((adapter_bwd_kernel<384>), dim3((R + kRows - 1) / kRows), dim3(kAT), lds, (hipStream_t)stream, g_out, ha, s1, W1,
                       W2, u, p, scale, g_ha, part, R, mean, rstd, gamma, beta);
    return upp_launch_status();
}

extern "C" int upp_adapter_bwd(const float *g_out, const float *ha, const float *s1, const float *W1, const float *W2, const float *u,
                               float p, float scale, float *g_ha, float *part, int R, int D, int H, void *stream) {
    return adapter_bwd_launch(g_out, ha, nullptr, nullptr, nullptr, nullptr, s1, W1, W2, u, p, scale, g_ha, part, R, D, H, stream);
}

extern "C" long long upp_ln_adapter_part_floats(int R, int D) {
    if (R < 1 || D < 1) return 0;
    return (long long)((R + kFR - 1) / kFR) * (2LL * kH * D + kH + D);
}

static int ln_adapter_bwd_fused_launch(const float *g_out, const float *xo, const float *mean, const float *rstd, const float *gamma,
                                       const float *beta, const float *s1, const float *W1, const float *W2, const float *ud, float p,
                                       float scale, const float *u, float keep, int mode, int P, float *g_x, float *g_y, float *part, float *fac,
                                       float *ln_part, int B, int Lin, int Lout, int D, int H, void *stream) {
    if (!g_out || !xo || !mean || !rstd || !gamma || !beta || !s1 || !W1 || !W2 || B < 0 || Lin < 1 || Lout < 1) return UPP_E_BADARG;
    if (D != 384 || H != kH) return UPP_E_RANGE;
    if (!(mode == 0 || mode == 3 || mode == 4) || P < 0 || (mode == 0 && Lout != Lin) || (mode != 0 && Lout != Lin - P)) return UPP_E_BADARG;
    if (B == 0) return 0;
    constexpr int NW = 8;
    const size_t lds = ((size_t)2 * kFR * 400 + kFR * (384 + 4) + NW * kFR * (kH + 1) + 2 * kFR * 48) * sizeof(float);
    static std::atomic<bool> raised{false};
    if (!raised) { int rc = raise_lds(ln_adapter_bwd_kernel<384, NW>, lds); if (rc) return rc; raised = true; }
    LnAdapterBwdArgs a{g_out, xo, mean, rstd, gamma, beta, s1, W1, W2, ud, u, p, scale, keep, mode, P, g_x, g_y, part, ln_part, B, Lin, Lout, fac};
    hipLaunchKernelGGL((ln_adapter_bwd_kernel<384, NW>), dim3((B * Lout + kFR - 1) / kFR), dim3(64 * NW), lds, (hipStream_t)stream, a);
    return upp_launch_status();
}

extern "C" int upp_ln_adapter_bwd_fused(const float *g_out, const float *xo, const float *mean, const float *rstd, const float *gamma,
                                        const float *beta, const float *s1, const float *W1, const float *W2, const float *ud, float p,
                                        float scale, const float *u, float keep, int mode, int P, float *g_x, float *g_y, float *part,
                                        float *ln_part, int B, int Lin, int Lout, int D, int H, void *stream) {
    return ln_adapter_bwd_fused_launch(g_out, xo, mean, rstd, gamma, beta, s1, W1, W2, ud, p, scale, u, keep, mode, P, g_x, g_y, part, nullptr, ln_part,
                                       B, Lin, Lout, D, H, stream);
}

extern "C" int upp_ln_adapter_bwd_factors(const float *g_out, const float *xo, const float *mean, const float *rstd, const float *gamma,
                                          const float *beta, const float *s1, const float *W1, const float *W2, const float *ud, float p,
                                          float scale, const float *u, float keep, int mode, int P, float *g_x, float *g_y, float *fac,
                                          float *ln_part, int B, int Lin, int Lout, int D, int H, void *stream) {
    if (!fac) return UPP_E_BADARG;
    return ln_adapter_bwd_fused_launch(g_out, xo, mean, rstd, gamma, beta, s1, W1, W2, ud, p, scale, u, keep, mode, P, g_x, g_y, nullptr, fac, ln_part,
                                       B, Lin, Lout, D, H, stream);
}

extern "C" int upp_adapter_wgrad_splits(int R) {
    if (R < 1) return 0;
    const int s = (R + 127) / 128;
    return s > 64 ? 64 : s;
}

extern "C" int upp_adapter_wgrad_batched(const float *const *xo, const float *const *mean, const float *const *rstd, const float *const *gamma,
                                         const float *const *beta, const float *const *g_out, const float *const *fac, const int *R,
                                         const float *scale, float *const *part, int jobs, int splits, int D, int H, void *stream) {
    if (jobs < 0 || (jobs > 0 && (!xo || !mean || !rstd || !gamma || !beta || !g_out || !fac || !R || !scale || !part))) return UPP_E_BADARG;
    if (D != 384 || H != kH || splits < 1 || splits > 64) return UPP_E_RANGE;
    for (int j = 0; j < jobs; ++j) {
        if (!xo[j] || !mean[j] || !rstd[j] || !gamma[j] || !beta[j] || !g_out[j] || !fac[j] || !part[j] || R[j] < 1) return UPP_E_BADARG;
        if ((reinterpret_cast<uintptr_t>(xo[j]) | reinterpret_cast<uintptr_t>(g_out[j]) | reinterpret_cast<uintptr_t>(fac[j]) |
             reinterpret_cast<uintptr_t>(gamma[j]) | reinterpret_cast<uintptr_t>(beta[j])) & 15) return UPP_E_RANGE;
    }
    for (int j0 = 0; j0 < jobs; j0 += kWgJobs) {
        AdapterWgradJobs a{};
        const int n = jobs - j0 < kWgJobs ? jobs - j0 : kWgJobs;
        for (int j = 0; j < n; ++j) {
            a.xo[j] = xo[j0 + j]; a.mean[j] = mean[j0 + j]; a.rstd[j] = rstd[j0 + j]; a.gamma[j] = gamma[j0 + j]; a.beta[j] = beta[j0 + j];
            a.g_out[j] = g_out[j0 + j]; a.fac[j] = fac[j0 + j]; a.part[j] = part[j0 + j]; a.R[j] = R[j0 + j]; a.scale[j] = scale[j0 + j];
        }
        a.splits = splits;
        hipLaunchKernelGGL((adapter_wgrad_kernel<384>), dim3((unsigned)(n * (D / 64) * splits)), dim3(256), 0, (hipStream_t)stream, a);
    }
    return upp_launch_status();
}

extern "C" int upp_ln_adapter_bwd(const float *g_out, const float *xo, const float *mean, const float *rstd, const float *gamma,
                                  const float *beta, const float *s1, const float *W1, const float *W2, const float *u, float p,
                                  float scale, float *g_ha, float *part, int R, int D, int H, void *stream) {
    if (!mean || !rstd || !gamma || !beta) return UPP_E_BADARG;
    return adapter_bwd_launch(g_out, xo, mean, rstd, gamma, beta, s1, W1, W2, u, p, scale, g_ha, part, R, D, H, stream);
}
