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
__device__ __forceinline__ float drop_factor(const float *u, size_t i, float p) {
    // torch dropout: keep with probability 1-p, scale kept values by 1/(1-p)
    return u ? (u[i] >= p ? 1.0f / (1.0f - p) : 0.0f) : 1.0f;
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

// one wave copies the contiguous 32x32 block W2[n0 .. n0+32][0..32) into its own LDS tile [32][33] (coalesced loads)
__device__ __forceinline__ void stage_w2_tile(float *dst, const float *W2, int n0, int lane) {
    float4 v[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) v[it] = *reinterpret_cast<const float4 *>(W2 + (size_t)n0 * kH + (lane + it * 64) * 4);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int e = (lane + it * 64) * 4;
        float *d = dst + (e >> 5) * kLG + (e & 31);
        d[0] = v[it].x; d[1] = v[it].y; d[2] = v[it].z; d[3] = v[it].w;
    }
    __builtin_amdgcn_wave_barrier();
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
                                                          float scale, float *__restrict__ g_ha, float *__restrict__ part, int R) {
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
    stage_tile<D>(Hs, ha, row0, R);
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

extern "C" int upp_adapter_bwd(const float *g_out, const float *ha, const float *s1, const float *W1, const float *W2, const float *u,
                               float p, float scale, float *g_ha, float *part, int R, int D, int H, void *stream) {
    if (!g_out || !ha || !s1 || !W1 || !W2 || !g_ha || !part || R < 1) return UPP_E_BADARG;
    if (D != 384 || H != kH) return UPP_E_RANGE;
    const size_t lds = ((size_t)2 * kRows * (D + 1) + (kAW + 2) * kRows * kLG) * sizeof(float);
    static std::atomic<bool> raised{false};
    if (!raised) { int rc = raise_lds(adapter_bwd_kernel<384>, lds); if (rc) return rc; raised = true; }
    hipLaunchKernelGGL((adapter_bwd_kernel<384>), dim3((R + kRows - 1) / kRows), dim3(kAT), lds, (hipStream_t)stream, g_out, ha, s1, W1,
                       W2, u, p, scale, g_ha, part, R);
    return upp_launch_status();
}
