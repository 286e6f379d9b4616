// pointwise.hip -- the small row operators of the frozen prompter branches, for gfx950.
//
// The rectify prompter (reference models/Point_MAE_pretask_dev.py:386-517) is a PointNet++-style network on a few
// tens of thousands of rows with 3..64 channels.  In the reference every layer is a chain of tiny element-wise /
// reduction kernels (BatchNorm statistics, update, transform, ReLU; reciprocal, sum, divide, gather, multiply,
// reduce; sin / cos per frequency; concatenations).  At these sizes each of them costs one launch (~5 us on MI355X)
// and nothing else, so the chains are folded into three operators:
//   upp_bn_rows_fwd : BatchNorm over the rows of a channels-last matrix (+ optional ReLU), batch statistics via
//                     per-slab (sum, M2) partials -> one finalize workgroup per 64 channels -> apply   [3 launches]
//   upp_interp_fwd  : inverse-distance interpolation from the k nearest of a sorted neighbour table   [1 launch]
//   upp_posenc_fwd  : (x, sin(f x), cos(f x))_f positional embedding                                  [1 launch]
// interp / posenc write into a column window of a wider row-major buffer, so the reference's torch.cat is free.
#include <atomic>

#include "common.h"

namespace {

constexpr int kBnWaves = 16;

// ---- BatchNorm over rows ----------------------------------------------------------------------------------------
// Partial statistics of one slab of `per` rows: part[(slab*2+0)*C + c] = sum_r x[r][c],
// part[(slab*2+1)*C + c] = M2 = sum_r (x[r][c] - slab mean)^2, computed from sums shifted by the slab's first row
// (shifted-data algorithm: no cancellation when |mean| >> std).  Thread (rl, col): col = tid % Cp, rl = tid / Cp.
__global__ __launch_bounds__(256) void bn_rows_partial_kernel(const float *__restrict__ x, int R, int C, int Cp, int per,
                                                              float *__restrict__ part) {
    __shared__ float s1[256], s2[256];
    const int tid = threadIdx.x;
    const int col = blockIdx.y * 256 + tid % Cp, rl = tid / Cp, RL = 256 / Cp;
    const int r0 = blockIdx.x * per, r1 = min(R, r0 + per);
    const int cc = min(col, C - 1);
    const float shift = x[(size_t)r0 * C + cc];
    float a1 = 0.0f, a2 = 0.0f;
    for (int rb = r0 + rl; rb < r1; rb += RL * 8) {         // 8 independent row loads in flight
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = x[(size_t)min(rb + t * RL, r1 - 1) * C + cc];
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (rb + t * RL < r1) { const float d = v[t] - shift; a1 += d; a2 = __builtin_fmaf(d, d, a2); }
    }
    s1[tid] = a1; s2[tid] = a2;
    __syncthreads();
    if (rl == 0 && col < C) {
        for (int q = 1; q < RL; ++q) { a1 += s1[q * Cp + tid]; a2 += s2[q * Cp + tid]; }   // fixed order
        const float n = (float)(r1 - r0);
        part[((size_t)blockIdx.x * 2 + 0) * C + col] = __builtin_fmaf(shift, n, a1);
        part[((size_t)blockIdx.x * 2 + 1) * C + col] = a2 - a1 * a1 / n;
    }
}

// mean / rstd from the slab partials (training) or the running statistics (eval); running statistics updated in
// training mode.  grid = ceil(C / 64), 16 waves; lane = column; wave w owns slabs w, w+16, ...
// Two passes over the (register-resident) partials, both in f64 and in a fixed order:
//   mean = sum_i S_i / rows;   M2 = sum_i [ M2_i + n_i (S_i / n_i - mean)^2 ]       (Chan et al., all slabs at once)
__global__ __launch_bounds__(64 * kBnWaves) void bn_finalize_rows_kernel(const float *__restrict__ part, int slabs, int per, int rows,
                                                                         int C, int training, float momentum, float eps,
                                                                         float *__restrict__ running_mean, float *__restrict__ running_var,
                                                                         float *__restrict__ mean_out, float *__restrict__ rstd_out) {
    __shared__ double sh[kBnWaves][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int cc = min(c, C - 1);
    if (!training) {
        if (wave == 0 && c < C) { mean_out[c] = running_mean[c]; rstd_out[c] = 1.0f / sqrtf(running_var[c] + eps); }
        return;
    }
    constexpr int T = 16;                                   // slabs held in registers per wave and sweep
    const bool one_sweep = slabs <= kBnWaves * T;
    float ps[T], pq[T];
    double s = 0.0;
    for (int w0 = wave; w0 < slabs; w0 += kBnWaves * T) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int w = min(w0 + kBnWaves * t, slabs - 1);
            ps[t] = part[((size_t)w * 2 + 0) * C + cc]; pq[t] = part[((size_t)w * 2 + 1) * C + cc];
        }
#pragma unroll
        for (int t = 0; t < T; ++t) if (w0 + kBnWaves * t < slabs) s += (double)ps[t];
    }
    sh[wave][lane] = s;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < kBnWaves; ++w) tot += sh[w][lane];
    const double mean = tot / (double)rows;
    __syncthreads();
    const double inv_per = 1.0 / (double)per;
    double q = 0.0;
    for (int w0 = wave; w0 < slabs; w0 += kBnWaves * T) {
        if (!one_sweep) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int w = min(w0 + kBnWaves * t, slabs - 1);
                ps[t] = part[((size_t)w * 2 + 0) * C + cc]; pq[t] = part[((size_t)w * 2 + 1) * C + cc];
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int w = w0 + kBnWaves * t;
            if (w < slabs) {
                const int nb = min(per, rows - w * per);
                const double mb = nb == per ? (double)ps[t] * inv_per : (double)ps[t] / (double)nb;
                const double d = mb - mean;
                q += (double)pq[t] + (double)nb * d * d;
            }
        }
    }
    sh[wave][lane] = q;
    __syncthreads();
    if (wave == 0 && c < C) {
        double m2 = 0.0;
#pragma unroll
        for (int w = 0; w < kBnWaves; ++w) m2 += sh[w][lane];
        const double n = (double)rows;
        const float var = (float)(m2 / n);
        mean_out[c] = (float)mean;
        rstd_out[c] = 1.0f / sqrtf(var + eps);
        if (running_mean) {
            const float unbiased = (float)(m2 / (n > 1.0 ? n - 1.0 : 1.0));
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
        }
    }
}

// Dropout behind BatchNorm + ReLU (round 6; reference models/Point_MAE_unify_segment.py:424-427 `Conv1d, BatchNorm1d, ReLU, Dropout(0.5)`):
// the mask of element i is a counter-based hash of (seed, i) -- recomputed by the backward kernels, never stored, no uniform tensor --
// with seed = the BatchNorm layer's own num_batches_tracked (a device scalar the step bumps once per forward, inside the captured graph)
// mixed with a per-site salt.  thresh = p * 2^32 (0: no dropout); kept values are scaled by 1 / (1 - p) as torch.nn.Dropout does.  The
// stream is this library's own (lowbias32 mixer), not torch's Philox: dropout is a stochastic regulariser, parity tests run it at p = 0.
struct BnDrop { const long long *seed; long long seed_add; unsigned salt, thresh; float scale; };
__device__ __forceinline__ unsigned drop_seed(const BnDrop &d) {
    if (!d.thresh) return 0u;
    unsigned h = (unsigned)(*d.seed + d.seed_add) * 0x9E3779B1u + d.salt;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ float drop_factor(const BnDrop &d, unsigned seed, long long elem) {
    unsigned h = ((unsigned)elem ^ __builtin_rotateleft32((unsigned)(elem >> 32), 13)) ^ seed;       // (the high word only matters beyond 2^32 elements)
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;      // lowbias32: two quarter-rate multiplies per element
    return h >= d.thresh ? d.scale : 0.0f;
}

// y = ((x - mean) * rstd) * gamma + beta, optionally max(., 0); 4 consecutive elements per thread
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float *p) {
    const v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float *p, float a, float b, float c, float d) {
    __builtin_nontemporal_store(v4f_t{a, b, c, d}, reinterpret_cast<v4f_t *>(p));
}
constexpr long long kStreamElems = 32ll << 20;   // 128 MB of f32: beyond this a tensor is streamed, not cached

__global__ __launch_bounds__(256) void bn_rows_apply_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                                            const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, int relu, float *__restrict__ y,
                                                            long long total, int C, int nt, BnDrop dr) {
    // nt: streams larger than the last-level cache (the 65,536-row activations of the per-point heads) are read and written
    // with non-temporal accesses -- a written-once stream that allocates in L2 costs a third of the HBM write rate
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= total) return;
    float v[4];
    int c[4];
    if ((C & 3) == 0) {
        const float4 t = nt ? nt_load4(x + i0) : *reinterpret_cast<const float4 *>(x + i0);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        const int cb = (int)(i0 % C);
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = cb + q;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const long long i = min(i0 + q, total - 1); v[q] = x[i]; c[q] = (int)(i % C); }
    }
    float mu[4], rs[4], ga[4], be[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { mu[q] = mean[c[q]]; rs[q] = rstd[c[q]]; ga[q] = gamma ? gamma[c[q]] : 1.0f; be[q] = beta ? beta[c[q]] : 0.0f; }
    float o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        o[q] = ((v[q] - mu[q]) * rs[q]) * ga[q] + be[q];
        if (relu) o[q] = fmaxf(o[q], 0.0f);
    }
    if (dr.thresh) {
        const unsigned sd = drop_seed(dr);
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] *= drop_factor(dr, sd, i0 + q);
    }
    if ((C & 3) == 0) { if (nt) nt_store4(y + i0, o[0], o[1], o[2], o[3]); else *reinterpret_cast<float4 *>(y + i0) = make_float4(o[0], o[1], o[2], o[3]); }
    else {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (i0 + q < total) y[i0 + q] = o[q];
    }
}

// ---- backward of BatchNorm(+ReLU) over rows (trainable heads: batch statistics) --------------------------------------
// With xh = (x - mean) * rstd, o = xh * gamma + beta, gm = g * [o > 0] (ReLU folded in; gm = g without it):
//   g_beta = sum_r gm,  g_gamma = sum_r gm * xh,  g_x = gamma * rstd * (gm - (g_beta + xh * g_gamma) / R).
// Same three-launch shape as the forward: per-slab column partials -> one finalize workgroup per 64 channels -> apply.
// The ReLU mask is recomputed from x with the forward's exact expression, so no activation is saved.
__global__ __launch_bounds__(256) void bn_rows_bwd_partial_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                                  const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                  const float *__restrict__ gamma, const float *__restrict__ beta, int relu,
                                                                  int R, int C, int Cp, int per, float *__restrict__ part, BnDrop dr) {
    __shared__ float s1[256], s2[256];
    const int tid = threadIdx.x;
    const int col = blockIdx.y * 256 + tid % Cp, rl = tid / Cp, RL = 256 / Cp;
    const int r0 = blockIdx.x * per, r1 = min(R, r0 + per);
    const int cc = min(col, C - 1);
    const float mu = mean[cc], rs = rstd[cc], ga = gamma ? gamma[cc] : 1.0f, be = beta ? beta[cc] : 0.0f;
    const unsigned sd = drop_seed(dr);
    float a1 = 0.0f, a2 = 0.0f;
    for (int rb = r0 + rl; rb < r1; rb += RL * 8) {         // 2 x 8 independent row loads in flight
        float xv[8], gv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const size_t o = (size_t)min(rb + t * RL, r1 - 1) * C + cc;
            xv[t] = x[o]; gv[t] = g[o];
            if (dr.thresh) gv[t] *= drop_factor(dr, sd, (long long)o);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (rb + t * RL < r1) {
                const float xh = (xv[t] - mu) * rs;
                const float gm = (relu && !(xh * ga + be > 0.0f)) ? 0.0f : gv[t];
                a1 += gm; a2 = __builtin_fmaf(gm, xh, a2);
            }
    }
    s1[tid] = a1; s2[tid] = a2;
    __syncthreads();
    if (rl == 0 && col < C) {
        for (int q = 1; q < RL; ++q) { a1 += s1[q * Cp + tid]; a2 += s2[q * Cp + tid]; }   // fixed order
        part[((size_t)blockIdx.x * 2 + 0) * C + col] = a1;
        part[((size_t)blockIdx.x * 2 + 1) * C + col] = a2;
    }
}

// g_beta / g_gamma = column sums of the slab partials, f64, fixed order.  grid = ceil(C / 64), 16 waves.
__global__ __launch_bounds__(64 * kBnWaves) void bn_rows_bwd_finalize_kernel(const float *__restrict__ part, int slabs, int C,
                                                                             float *__restrict__ g_gamma, float *__restrict__ g_beta) {
    __shared__ double sh[2][kBnWaves][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int cc = min(c, C - 1);
    constexpr int T = 16;
    double b = 0.0, gq = 0.0;
    for (int w0 = wave; w0 < slabs; w0 += kBnWaves * T) {
        float pb[T], pg[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int w = min(w0 + kBnWaves * t, slabs - 1);
            pb[t] = part[((size_t)w * 2 + 0) * C + cc]; pg[t] = part[((size_t)w * 2 + 1) * C + cc];
        }
#pragma unroll
        for (int t = 0; t < T; ++t) if (w0 + kBnWaves * t < slabs) { b += (double)pb[t]; gq += (double)pg[t]; }
    }
    sh[0][wave][lane] = b; sh[1][wave][lane] = gq;
    __syncthreads();
    if (wave == 0 && c < C) {
        double tb = 0.0, tg = 0.0;
#pragma unroll
        for (int w = 0; w < kBnWaves; ++w) { tb += sh[0][w][lane]; tg += sh[1][w][lane]; }
        g_beta[c] = (float)tb; g_gamma[c] = (float)tg;
    }
}

__global__ __launch_bounds__(256) void bn_rows_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                                const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                const float *__restrict__ g_gamma, const float *__restrict__ g_beta, int relu,
                                                                float inv_rows, float *__restrict__ g_x, long long total, int C, int nt, BnDrop dr) {
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= total) return;
    float xv[4], gv[4];
    int c[4];
    if ((C & 3) == 0) {
        const float4 t = nt ? nt_load4(x + i0) : *reinterpret_cast<const float4 *>(x + i0);
        const float4 u = nt ? nt_load4(g + i0) : *reinterpret_cast<const float4 *>(g + i0);
        xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
        gv[0] = u.x; gv[1] = u.y; gv[2] = u.z; gv[3] = u.w;
        const int cb = (int)(i0 % C);
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = cb + q;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const long long i = min(i0 + q, total - 1); xv[q] = x[i]; gv[q] = g[i]; c[q] = (int)(i % C); }
    }
    if (dr.thresh) {
        const unsigned sd = drop_seed(dr);
#pragma unroll
        for (int q = 0; q < 4; ++q) gv[q] *= drop_factor(dr, sd, i0 + q);
    }
    float mu[4], rs[4], ga[4], be[4], gg[4], gb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        mu[q] = mean[c[q]]; rs[q] = rstd[c[q]]; ga[q] = gamma ? gamma[c[q]] : 1.0f; be[q] = beta ? beta[c[q]] : 0.0f;
        gg[q] = g_gamma[c[q]]; gb[q] = g_beta[c[q]];
    }
    float o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float xh = (xv[q] - mu[q]) * rs[q];
        const float gm = (relu && !(xh * ga[q] + be[q] > 0.0f)) ? 0.0f : gv[q];
        o[q] = (ga[q] * rs[q]) * (gm - (gb[q] + xh * gg[q]) * inv_rows);
    }
    if ((C & 3) == 0) { if (nt) nt_store4(g_x + i0, o[0], o[1], o[2], o[3]); else *reinterpret_cast<float4 *>(g_x + i0) = make_float4(o[0], o[1], o[2], o[3]); }
    else {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (i0 + q < total) g_x[i0 + q] = o[q];
    }
}

// Tall matrices (the 65,536-row activations of the per-point heads), C % 256 == 0 (or C = 64 / 128: `rpw` = 4 / 2 whole rows per
// wave-load; the flat kernel ran the head's 128-channel layer at 2 TB/s): a wave owns 256 consecutive columns -- one 1 KB
// row segment per load instruction -- and walks down the rows with the column parameters in registers (the flat kernels above spend
// 16-24 dword parameter loads per 16 bytes of payload and read at a third of the HBM rate).  Block = 4 waves on 4 interleaved rows
// of the same 256 columns, kTallRows rows per block, 8 row loads in flight per lane.  Same expressions: same bits as the flat kernels.
constexpr int kTallRows = 32;

template <bool NT>
__global__ __launch_bounds__(256) void bn_rows_apply_tall_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                                                 const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, int relu, float *__restrict__ y, int R,
                                                                 int C, int rpw, BnDrop dr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = C >= 256 ? blockIdx.y * 256 + lane * 4 : (lane * 4) % C;          // C = 64 / 128: a wave-load covers `rpw` whole rows
    const int sub = C >= 256 ? 0 : (lane * 4) / C;
    const float4 mu = *reinterpret_cast<const float4 *>(mean + c0), rs = *reinterpret_cast<const float4 *>(rstd + c0);
    const float4 ga = gamma ? *reinterpret_cast<const float4 *>(gamma + c0) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    const float4 be = beta ? *reinterpret_cast<const float4 *>(beta + c0) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int r0 = (blockIdx.x * kTallRows + wave) * rpw + sub, rs4 = 4 * rpw;
    const unsigned sd = drop_seed(dr);
    constexpr int U = kTallRows / 4;
    float4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const size_t o = (size_t)min(r0 + rs4 * j, R - 1) * C + c0;
        v[j] = NT ? nt_load4(x + o) : *reinterpret_cast<const float4 *>(x + o);
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
        if (r0 + rs4 * j >= R) continue;
        float o0 = ((v[j].x - mu.x) * rs.x) * ga.x + be.x, o1 = ((v[j].y - mu.y) * rs.y) * ga.y + be.y;
        float o2 = ((v[j].z - mu.z) * rs.z) * ga.z + be.z, o3 = ((v[j].w - mu.w) * rs.w) * ga.w + be.w;
        if (relu) { o0 = fmaxf(o0, 0.0f); o1 = fmaxf(o1, 0.0f); o2 = fmaxf(o2, 0.0f); o3 = fmaxf(o3, 0.0f); }
        if (dr.thresh) {
            const long long e0 = (long long)(r0 + rs4 * j) * C + c0;
            o0 *= drop_factor(dr, sd, e0); o1 *= drop_factor(dr, sd, e0 + 1); o2 *= drop_factor(dr, sd, e0 + 2); o3 *= drop_factor(dr, sd, e0 + 3);
        }
        float *dst = y + (size_t)(r0 + rs4 * j) * C + c0;
        if (NT) nt_store4(dst, o0, o1, o2, o3); else *reinterpret_cast<float4 *>(dst) = make_float4(o0, o1, o2, o3);
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void bn_rows_bwd_apply_tall_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                                     const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                     const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                     const float *__restrict__ g_gamma, const float *__restrict__ g_beta,
                                                                     int relu, float inv_rows, float *__restrict__ g_x, int R, int C, int rpw, BnDrop dr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = C >= 256 ? blockIdx.y * 256 + lane * 4 : (lane * 4) % C;
    const int sub = C >= 256 ? 0 : (lane * 4) / C;
    const float4 mu4 = *reinterpret_cast<const float4 *>(mean + c0), rs4 = *reinterpret_cast<const float4 *>(rstd + c0);
    const float4 ga4 = gamma ? *reinterpret_cast<const float4 *>(gamma + c0) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    const float4 be4 = beta ? *reinterpret_cast<const float4 *>(beta + c0) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 gg4 = *reinterpret_cast<const float4 *>(g_gamma + c0), gb4 = *reinterpret_cast<const float4 *>(g_beta + c0);
    const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, rs[4] = {rs4.x, rs4.y, rs4.z, rs4.w}, ga[4] = {ga4.x, ga4.y, ga4.z, ga4.w};
    const float be[4] = {be4.x, be4.y, be4.z, be4.w}, gg[4] = {gg4.x, gg4.y, gg4.z, gg4.w}, gb[4] = {gb4.x, gb4.y, gb4.z, gb4.w};
    const int r0 = (blockIdx.x * kTallRows + wave) * rpw + sub, rstep = 4 * rpw;
    const unsigned sd = drop_seed(dr);
    constexpr int U = kTallRows / 8;                           // two streams: 2 x 4 row loads in flight per lane, twice
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 xv[U], gv[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const size_t o = (size_t)min(r0 + rstep * (h * U + j), R - 1) * C + c0;
            xv[j] = NT ? nt_load4(x + o) : *reinterpret_cast<const float4 *>(x + o);
            gv[j] = NT ? nt_load4(g + o) : *reinterpret_cast<const float4 *>(g + o);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int r = r0 + rstep * (h * U + j);
            if (r >= R) continue;
            const float xs[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
            float gs[4] = {gv[j].x, gv[j].y, gv[j].z, gv[j].w};
            if (dr.thresh) {
                const long long e0 = (long long)r * C + c0;
#pragma unroll
                for (int q = 0; q < 4; ++q) gs[q] *= drop_factor(dr, sd, e0 + q);
            }
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float xh = (xs[q] - mu[q]) * rs[q];
                const float gm = (relu && !(xh * ga[q] + be[q] > 0.0f)) ? 0.0f : gs[q];
                o[q] = (ga[q] * rs[q]) * (gm - (gb[q] + xh * gg[q]) * inv_rows);
            }
            float *dst = g_x + (size_t)r * C + c0;
            if (NT) nt_store4(dst, o[0], o[1], o[2], o[3]); else *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// ---- k nearest source points per query, in the reference's square_distance form ----------------------------------
// dist[row][0..k) / idx[row][0..k) = the k smallest  d = |a|^2 + |b|^2 - 2 a.b  over the S source points of the query's
// cloud, ascending (d, index) -- what `square_distance(xyz1, xyz2).sort(dim=-1)` followed by `[:, :, :k]` yields in the
// reference (models/modules.py:13-32, models/Point_MAE_unify.py:22-48, Point_MAE_pretask_dev.py:443-449), without the
// (B,N,S) matrix, its matmul, five element-wise passes and the full sort.  d is evaluated in the operation order of the
// reference formula (dist = -2 a.b; dist += |a|^2; dist += |b|^2).  One wavefront per query; lane l holds the
// candidates l, l+64, ... (SL per lane); k selection rounds of one wave-wide u32 min on an order-preserving image.
template <int SL>
__global__ __launch_bounds__(256) void sqdist_topk_kernel(const float *__restrict__ q, const float *__restrict__ src, int N, int S, int k,
                                                          float *__restrict__ dist, int64_t *__restrict__ idx, long long rows) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= rows) return;
    const long long b = row / N;
    const float *a = q + row * 3;
    const float ax = a[0], ay = a[1], az = a[2];
    float bx[SL], by[SL], bz[SL];
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
        const int j = min(lane + 64 * sl, S - 1);
        const float *p = src + (b * S + j) * 3;
        bx[sl] = p[0]; by[sl] = p[1]; bz[sl] = p[2];
    }
    const float aa = (ax * ax + ay * ay) + az * az;
    uint32_t img[SL];
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
        float d = -2.0f * __builtin_fmaf(az, bz[sl], __builtin_fmaf(ay, by[sl], ax * bx[sl]));
        d += aa;
        d += (bx[sl] * bx[sl] + by[sl] * by[sl]) + bz[sl] * bz[sl];
        const uint32_t bits = __float_as_uint(d);
        img[sl] = lane + 64 * sl < S ? ((bits & 0x80000000u) ? ~bits : (bits | 0x80000000u)) : 0xFFFFFFFFu;
    }
    float myd = 0.0f;
    int myj = 0;
    if (SL == 1 && k > 4) {
        // one candidate per lane and many neighbours wanted: a bitonic sort of the 64 (distance image, index) keys across the
        // lanes (21 compare-exchange stages) is ~3x fewer instructions than k rounds of wave-min + ballot
        unsigned long long key = ((unsigned long long)img[0] << 32) | (unsigned)lane;
#pragma unroll
        for (int k2 = 2; k2 <= 64; k2 <<= 1)
#pragma unroll
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                const unsigned long long other = __shfl_xor(key, j);
                const bool take_min = ((lane & k2) == 0) == ((lane & j) == 0);
                key = take_min ? (other < key ? other : key) : (other > key ? other : key);
            }
        const uint32_t m = (uint32_t)(key >> 32);
        if (lane < k) {
            dist[row * k + lane] = __uint_as_float((m & 0x80000000u) ? (m & 0x7FFFFFFFu) : ~m);
            idx[row * k + lane] = (int64_t)(key & 63u);
        }
        return;
    }
    for (int r = 0; r < k; ++r) {
        uint32_t lm = img[0];
#pragma unroll
        for (int sl = 1; sl < SL; ++sl) lm = min(lm, img[sl]);
        const uint32_t m = wave_min_u32(lm);
        int wsl = 0, wl = 0;
        bool found = false;
#pragma unroll
        for (int sl = 0; sl < SL; ++sl) {              // lowest index among equal distances: lowest slot, then lowest lane
            const unsigned long long mask = __ballot(img[sl] == m);
            if (!found && mask) { found = true; wsl = sl; wl = __builtin_ctzll(mask); }
        }
#pragma unroll
        for (int sl = 0; sl < SL; ++sl) if (sl == wsl && lane == wl) img[sl] = 0xFFFFFFFFu;
        if (lane == r) { myd = __uint_as_float((m & 0x80000000u) ? (m & 0x7FFFFFFFu) : ~m); myj = wl + 64 * wsl; }
    }
    if (lane < k) { dist[row * k + lane] = myd; idx[row * k + lane] = (int64_t)myj; }
}

// ---- inverse-distance interpolation ----------------------------------------------------------------------------------
// out[row][col0 + c] = sum_{j<k} w_j * feat[b][idx[row][j]][c],  w_j = (1/(d_j+eps)) / sum_j (1/(d_j+eps)),
// (d, idx) = the first k entries of row `row` of a neighbour table sorted by distance (row stride ld_tab).
// PointNetFeaturePropagation.forward, reference models/Point_MAE_pretask_dev.py:443-462.
constexpr int kInterpK = 16;
// KK = compile-time bound of the neighbour count (k <= KK): the gathers are unrolled KK deep, so a 3-neighbour
// interpolation must not pay for 16.
template <int KK>
__global__ __launch_bounds__(256) void interp_kernel(const float *__restrict__ dist, const int64_t *__restrict__ idx, int ld_tab,
                                                     const float *__restrict__ feat, int S, int C, int Cp, int k, float eps,
                                                     float *__restrict__ out, int ld_out, int col0, int rows, int N) {
    const int tid = threadIdx.x;
    const int col = blockIdx.y * 256 + tid % Cp, rl = tid / Cp, RL = 256 / Cp;
    const int row = blockIdx.x * RL + rl;
    if (row >= rows) return;
    const int b = row / N;
    const int cc = min(col, C - 1);
    float d[KK];
    int id[KK];
#pragma unroll
    for (int j = 0; j < KK; ++j) {
        const int jj = min(j, k - 1);
        d[j] = dist[(size_t)row * ld_tab + jj];
        id[j] = (int)idx[(size_t)row * ld_tab + jj];
    }
    float f[KK];
#pragma unroll
    for (int j = 0; j < KK; ++j) f[j] = feat[((size_t)b * S + id[j]) * C + cc];
    float w[KK], norm = 0.0f;
#pragma unroll
    for (int j = 0; j < KK; ++j) { w[j] = j < k ? 1.0f / (d[j] + eps) : 0.0f; norm += w[j]; }
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < KK; ++j) acc += f[j] * (w[j] / norm);
    if (col < C) out[(size_t)row * ld_out + col0 + col] = acc;
}

// Wide rows (C >= 256, C % 4 == 0, k <= 4): one wave per target row walks the row in 1 KB pieces (a float4 per lane), so the
// k table entries are read once per row instead of once per 256 columns and every gather is a contiguous kilobyte.
// Optional affine term (the part-segmentation head's commuted first layer, upp_layers._forward_commuted):
//   out[row][c] += x[row][0] * wt[0][c] + x[row][1] * wt[1][c] + x[row][2] * wt[2][c],   x (rows,3), wt (3,C).
// Same weights and summation order over the neighbours as interp_kernel.
__global__ __launch_bounds__(256) void interp_wide_kernel(const float *__restrict__ dist, const int64_t *__restrict__ idx, int ld_tab,
                                                          const float *__restrict__ feat, int S, int C, int k, float eps,
                                                          const float *__restrict__ x3, const float *__restrict__ wt,
                                                          float *__restrict__ out, int ld_out, int col0, int rows, int N) {
    const int lane = threadIdx.x & 63;
    // XCD-aware order: consecutive workgroups go to different XCDs, so every L2 would fetch the source rows of every sample
    // (PMC: 200 MB for 25 MB of features); give every XCD a contiguous range of target rows = whole samples instead
    int blk = blockIdx.x;
    if ((gridDim.x & 7) == 0) blk = (blk & 7) * (gridDim.x >> 3) + (blk >> 3);
    const int row = blk * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / N;
    float d[4];
    int id[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int jj = min(j, k - 1);
        d[j] = dist[(size_t)row * ld_tab + jj];
        id[j] = (int)idx[(size_t)row * ld_tab + jj];
    }
    float px = 0.0f, py = 0.0f, pz = 0.0f;
    if (x3) { px = x3[(size_t)row * 3 + 0]; py = x3[(size_t)row * 3 + 1]; pz = x3[(size_t)row * 3 + 2]; }
    float w[4], norm = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { w[j] = j < k ? 1.0f / (d[j] + eps) : 0.0f; norm += w[j]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = w[j] / norm;
    const float *f0 = feat + ((size_t)b * S + id[0]) * C, *f1 = feat + ((size_t)b * S + id[1]) * C;
    const float *f2 = feat + ((size_t)b * S + id[2]) * C, *f3 = feat + ((size_t)b * S + id[3]) * C;
    float *o = out + (size_t)row * ld_out + col0;
    for (int c0 = 0; c0 < C; c0 += 512) {          // two pieces per trip: 8 + 6 independent loads in flight
        const int ca = c0 + lane * 4, cb = ca + 256;
        const int la = min(ca, C - 4), lb = min(cb, C - 4);
        const float4 a0 = *reinterpret_cast<const float4 *>(f0 + la), a1 = *reinterpret_cast<const float4 *>(f1 + la);
        const float4 a2 = *reinterpret_cast<const float4 *>(f2 + la), a3 = *reinterpret_cast<const float4 *>(f3 + la);
        const float4 b0 = *reinterpret_cast<const float4 *>(f0 + lb), b1 = *reinterpret_cast<const float4 *>(f1 + lb);
        const float4 b2 = *reinterpret_cast<const float4 *>(f2 + lb), b3 = *reinterpret_cast<const float4 *>(f3 + lb);
        float4 ra, rb;
#define UPP_MIX(r, v0, v1, v2, v3, m)  r.m = v0.m * w[0]; r.m += v1.m * w[1]; r.m += v2.m * w[2]; r.m += v3.m * w[3];
        UPP_MIX(ra, a0, a1, a2, a3, x) UPP_MIX(ra, a0, a1, a2, a3, y) UPP_MIX(ra, a0, a1, a2, a3, z) UPP_MIX(ra, a0, a1, a2, a3, w)
        UPP_MIX(rb, b0, b1, b2, b3, x) UPP_MIX(rb, b0, b1, b2, b3, y) UPP_MIX(rb, b0, b1, b2, b3, z) UPP_MIX(rb, b0, b1, b2, b3, w)
#undef UPP_MIX
        if (x3) {
            const float4 ua = *reinterpret_cast<const float4 *>(wt + la), va = *reinterpret_cast<const float4 *>(wt + C + la);
            const float4 ta = *reinterpret_cast<const float4 *>(wt + 2 * (size_t)C + la);
            const float4 ub = *reinterpret_cast<const float4 *>(wt + lb), vb = *reinterpret_cast<const float4 *>(wt + C + lb);
            const float4 tb = *reinterpret_cast<const float4 *>(wt + 2 * (size_t)C + lb);
#define UPP_AFF(r, u, v, t, m)  r.m += (px * u.m + py * v.m) + pz * t.m;
            UPP_AFF(ra, ua, va, ta, x) UPP_AFF(ra, ua, va, ta, y) UPP_AFF(ra, ua, va, ta, z) UPP_AFF(ra, ua, va, ta, w)
            UPP_AFF(rb, ub, vb, tb, x) UPP_AFF(rb, ub, vb, tb, y) UPP_AFF(rb, ub, vb, tb, z) UPP_AFF(rb, ub, vb, tb, w)
#undef UPP_AFF
        }
        typedef float v4f __attribute__((ext_vector_type(4)));
        if (ca < C) __builtin_nontemporal_store(v4f{ra.x, ra.y, ra.z, ra.w}, reinterpret_cast<v4f *>(o + ca));
        if (cb < C) __builtin_nontemporal_store(v4f{rb.x, rb.y, rb.z, rb.w}, reinterpret_cast<v4f *>(o + cb));
    }
}

// Backward of the interpolation w.r.t. the features (the neighbour table is constant):
//   g_feat[b][s][c] = sum over the (row, j) with idx[row][j] == s of  w_j(row) * g_out[row][col0 + c].
// One workgroup per (sample, source row s): it scans the sample's N*k table entries in order, compacts the hits
// (row, weight) into LDS with ballots (ascending entry order -> deterministic sums, no atomics), then accumulates the
// referenced gradient rows; entries are split over 256/Cp thread groups whose partial sums are combined in a fixed order.
constexpr int kInterpMaxRows = 4096;
constexpr int kInterpSeg = kInterpMaxRows / 4 + 72;   // hits of one wave's quarter of the entries: at most one per row
__global__ __launch_bounds__(256) void interp_bwd_kernel(const float *__restrict__ dist, const int64_t *__restrict__ idx, int ld_tab,
                                                         const float *__restrict__ g_out, int ld_g, int col0, int S, int C, int Cp, int k,
                                                         float eps, int N, float *__restrict__ g_feat) {
    __shared__ int l_row[4][kInterpSeg];
    __shared__ float l_w[4][kInterpSeg];
    __shared__ int cnt[4];
    __shared__ float red[256];
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so the S workgroups of one sample -- which read the
    // same gradient rows, k times in total -- would sit on 8 different L2s and fetch every row from memory k times (PMC:
    // 1.13 GB for 403 MB of gradients).  Give every XCD whole samples instead.
    int lin = blockIdx.x;
    if ((gridDim.x & 7) == 0) lin = (lin & 7) * (gridDim.x >> 3) + (lin >> 3);
    const int b = lin / S, s = lin - b * S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t base = (size_t)b * N;
    const int E = N * k;
    {   // scan: wave w owns the contiguous entry range [w * per, (w+1) * per) and compacts its hits into its own LDS
        // segment with ballots -- no workgroup barrier inside the loop; four 64-entry chunks of loads in flight
        const int per = (((E + 3) >> 2) + 63) & ~63;
        const int e_lo = wave * per, e_hi = min(E, e_lo + per);
        int c = 0;
        for (int e0 = e_lo; e0 < e_hi; e0 += 256) {
            int row[4], j[4];
            long long id[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int e = e0 + 64 * t + lane;
                row[t] = e / k; j[t] = e - row[t] * k;
                id[t] = e < e_hi ? idx[(base + row[t]) * ld_tab + j[t]] : -1;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool hit = id[t] == (long long)s;
                const unsigned long long m = __ballot(hit);
                if (hit) {
                    const float *dr = dist + (base + row[t]) * ld_tab;
                    float norm = 0.0f, mine = 0.0f;
                    for (int q = 0; q < k; ++q) { const float r = 1.0f / (dr[q] + eps); norm += r; if (q == j[t]) mine = r; }
                    const int pos = c + __popcll(m & ((1ull << lane) - 1ull));
                    l_row[wave][pos] = row[t]; l_w[wave][pos] = mine / norm;
                }
                c += __popcll(m);
            }
        }
        if (lane == 0) cnt[wave] = c;
    }
    __syncthreads();
    const int p1 = cnt[0], p2 = p1 + cnt[1], p3 = p2 + cnt[2], count = p3 + cnt[3];
    // entry i of the concatenated list (segments in wave order = ascending entry order)
    auto seg_of = [&](int i, int &w, int &o) { w = (i >= p1) + (i >= p2) + (i >= p3); o = i - (w == 0 ? 0 : (w == 1 ? p1 : (w == 2 ? p2 : p3))); };
    if (Cp == 256 && C <= 2048) {
        // wide rows: walk the hits ONCE, row-major (a whole gradient row = C / 256 coalesced kilobyte loads per thread group),
        // so the k workgroups that read a row do so at about the same point of their ascending lists and share it in L2
        // (column-chunk-major order re-read every hit row C / 256 times, each time as a separate 1 KB piece).  The sum over
        // the hits of a column keeps its ascending order.
        const int nch = (C + 255) >> 8;
        float acc[8];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) acc[ch] = 0.0f;
        for (int i = 0; i < count; i += 4) {
            float v[4][8], w[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int sw, so;
                seg_of(min(i + t, count - 1), sw, so);
                w[t] = i + t < count ? l_w[sw][so] : 0.0f;
                const float *gr = g_out + (base + l_row[sw][so]) * ld_g + col0;
#pragma unroll
                for (int ch = 0; ch < 8; ++ch)
                    if (ch < nch) v[t][ch] = gr[min(ch * 256 + tid, C - 1)];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ch = 0; ch < 8; ++ch)
                    if (ch < nch) acc[ch] = __builtin_fmaf(w[t], v[t][ch], acc[ch]);
        }
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            const int c = ch * 256 + tid;
            if (ch < nch && c < C) g_feat[((size_t)b * S + s) * C + c] = acc[ch];
        }
        return;
    }
    const int cq = tid % Cp, part = tid / Cp, RL = 256 / Cp;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + cq, cc = min(c, C - 1);
        float acc = 0.0f;
        for (int i = part; i < count; i += RL * 8) {
            float v[8], w[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                int sw, so;
                seg_of(min(i + t * RL, count - 1), sw, so);
                w[t] = i + t * RL < count ? l_w[sw][so] : 0.0f;
                v[t] = g_out[(base + l_row[sw][so]) * ld_g + col0 + cc];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) acc = __builtin_fmaf(w[t], v[t], acc);
        }
        if (RL > 1) {
            red[tid] = acc;
            __syncthreads();
            if (part == 0) for (int q = 1; q < RL; ++q) acc += red[q * Cp + cq];
            __syncthreads();
        }
        if (part == 0 && c < C) g_feat[((size_t)b * S + s) * C + c] = acc;
    }
}

// Few source rows, narrow features (the rectify prompter's 1096 <- 32, k = 16, C = 32 layer in stage 2 / the pre-task recipe): the pull
// kernel above spends its time scanning -- each of its B S workgroups reads the sample's whole N k table (17,536 entries) to find its
// ~550 hits: 125 us for 4.5 MB of gradients.  Here ONE workgroup per sample walks the rows in tiles of TR: the tile's table entries are
// read coalesced into the LDS, turned into weights by one thread per row (the forward's expression: 1 / (d + eps) summed in neighbour order)
// and scattered into a dense (TR x S) weight matrix (k plain stores per row: the k neighbours of a row are distinct); the tile's gradient
// rows are staged beside it, and  g_feat[s][c] += sum_r Wd[r][s] * g[r][c]  runs out of the LDS: a thread owns four source rows x four
// columns (two 16-byte LDS reads per 16 fmaf), the four waves take every fourth row of the tile and their partial sums are added in wave
// order at the end.  Deterministic; (S + C + 2 kInterpK) TR floats of LDS; S % 4 == 0, C % 4 == 0, S C <= 4,096.
template <int NW>
__global__ __launch_bounds__(64 * NW) void interp_bwd_dense_kernel(const float *__restrict__ dist, const int64_t *__restrict__ idx, int ld_tab,
                                                               const float *__restrict__ g_out, int ld_g, int col0, int S, int C, int k,
                                                               float eps, int N, int TR, float *__restrict__ g_feat) {
    constexpr int NT = 64 * NW;              // (a lone wave per SIMD issues one VALU instruction per ~10 cycles: NW = 8 puts two on each)
    extern __shared__ float lds_f[];
    float *wd = lds_f;                                            // [TR][S]
    float *gl = wd + (size_t)TR * S;                              // [TR][C]
    float *td = gl + (size_t)TR * C;                              // [TR][k] distances
    int *ti = reinterpret_cast<int *>(td + (size_t)TR * kInterpK);   // [TR][k] source rows
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t base = (size_t)b * N;
    const int quads = C >> 2, sq = S >> 2, units = sq * quads;    // unit = (s-quad, c-quad); lane owns units lane, lane + 64, ... (<= kDU)
    constexpr int kDU = 4;
    float4 acc[kDU][4];                                           // acc[j][e] = g_feat[o_s + e][o_c .. o_c + 3] of unit j, this wave's rows
#pragma unroll
    for (int j = 0; j < kDU; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int t0 = 0; t0 < N; t0 += TR) {
        const int rows = min(TR, N - t0);
        __syncthreads();                                          // the previous tile's readers are done
        for (int i = tid; i < rows * S; i += NT) wd[i] = 0.0f;
        for (int i0 = tid; i0 < rows * quads; i0 += NT * 8) {     // 8 global loads in flight per thread (a load -> store loop pays ~1 us each)
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + NT * u, rows * quads - 1);
                const int r = i / quads, q = i - r * quads;
                v[u] = *reinterpret_cast<const float4 *>(g_out + (base + t0 + r) * ld_g + col0 + 4 * q);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + NT * u;
                if (i < rows * quads) { const int r = i / quads, q = i - r * quads; *reinterpret_cast<float4 *>(gl + r * C + 4 * q) = v[u]; }
            }
        }
        for (int i0 = tid; i0 < rows * k; i0 += NT * 4) {         // the tile's table entries, consecutive lanes on consecutive entries
            float dv[4];
            int iv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + NT * u, rows * k - 1);
                const int r = i / k, q = i - r * k;
                dv[u] = dist[(base + t0 + r) * ld_tab + q]; iv[u] = (int)idx[(base + t0 + r) * ld_tab + q];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + NT * u;
                if (i < rows * k) { const int r = i / k, q = i - r * k; td[r * kInterpK + q] = dv[u]; ti[r * kInterpK + q] = iv[u]; }
            }
        }
        __syncthreads();
        if (tid < rows) {
            float rc[kInterpK];
            float norm = 0.0f;
#pragma unroll
            for (int q = 0; q < kInterpK; ++q) { rc[q] = 1.0f / (td[tid * kInterpK + min(q, k - 1)] + eps); if (q < k) norm += rc[q]; }
#pragma unroll
            for (int q = 0; q < kInterpK; ++q) if (q < k) wd[tid * S + ti[tid * kInterpK + q]] = rc[q] / norm;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kDU; ++j) {
            const int unit = lane + 64 * j;
            if (unit >= units) break;
            const int o_s = (unit / quads) * 4, o_c = (unit % quads) * 4;
            for (int r0 = wave; r0 < rows; r0 += NW * 4) {         // this wave's rows r0, r0 + NW, ...: four rows of reads in flight
                float4 w[4], v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = min(r0 + NW * u, rows - 1);
                    w[u] = *reinterpret_cast<const float4 *>(wd + r * S + o_s);
                    v[u] = *reinterpret_cast<const float4 *>(gl + r * C + o_c);
                    if (r0 + NW * u >= rows) w[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float ws[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[j][e].x = __builtin_fmaf(ws[e], v[u].x, acc[j][e].x); acc[j][e].y = __builtin_fmaf(ws[e], v[u].y, acc[j][e].y);
                        acc[j][e].z = __builtin_fmaf(ws[e], v[u].z, acc[j][e].z); acc[j][e].w = __builtin_fmaf(ws[e], v[u].w, acc[j][e].w);
                    }
                }
            }
        }
    }
    // the four waves' partial sums, added in wave order (through the weight-matrix area: S C floats per wave <= 4 KB)
    __syncthreads();
    float *red = lds_f;
#pragma unroll
    for (int j = 0; j < kDU; ++j) {
        const int unit = lane + 64 * j;
        if (unit >= units) break;
        const int o_s = (unit / quads) * 4, o_c = (unit % quads) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<float4 *>(red + (size_t)wave * S * C + (o_s + e) * C + o_c) = acc[j][e];
    }
    __syncthreads();
    for (int i = tid; i < S * quads; i += NT) {
        const int srow = i / quads, q = i - srow * quads;
        const float *p0 = red + srow * C + 4 * q;
        float4 a = *reinterpret_cast<const float4 *>(p0);
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) {
            const float4 t = *reinterpret_cast<const float4 *>(p0 + (size_t)w2 * S * C);
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        *reinterpret_cast<float4 *>(g_feat + ((size_t)b * S + srow) * C + 4 * q) = a;
    }
}

// ---- positional embedding ---------------------------------------------------------------------------------------
// out[row][col0 + ...] = (x, sin(f0 x), cos(f0 x), sin(f1 x), cos(f1 x), ...) for x (rows, 3)
// (reference models/Point_MAE_pretask_dev.py:22-52).  One thread per (row, coordinate).
struct Freqs { float f[8]; };
__global__ __launch_bounds__(256) void posenc_kernel(const float *__restrict__ x, Freqs fr, int F, float *__restrict__ out, int ld_out,
                                                     int col0, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long row = i / 3;
    const int c = (int)(i - row * 3);
    const float v = x[i];
    float *o = out + (size_t)row * ld_out + col0 + c;
    o[0] = v;
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (q < F) { const float a = fr.f[q] * v; o[3 + 6 * q] = sinf(a); o[6 + 6 * q] = cosf(a); }
}

int bn_per(int R) { int per = (R + 255) / 256; return per < 8 ? 8 : per; }
int pow2_at_least(int c) { int p = 4; while (p < c && p < 256) p <<= 1; return p; }
// the row-walking apply kernels: whole 1 KB wave-loads -- C a multiple of 256, or 64 / 128 (4 / 2 whole rows per load)
bool bn_tall(int R, int C) { return R >= 4096 && (C % 256 == 0 || C == 128 || C == 64); }


// Backward of the interpolation w.r.t. the GEOMETRY (the queries xyz1 and the sources xyz2; stage 2 of the recipe and the pre-task
// recipe differentiate through the prompters' interpolations: reference models/Point_MAE_unify.py:22-48 / models/modules.py:13-32 under
// autograd -- about forty element-wise / sort / index kernels per site there).  out[row] = sum_j w_j feat[idx_j], w_j = r_j / R,
// r_j = 1 / (d_j + eps), d_j = |x1 - x2_j|^2:
//     g_w_j = <g_out[row], feat[idx_j]>,  g_r_j = (g_w_j - sum_l g_w_l w_l) / R,  g_d_j = -g_r_j r_j^2,
//     g_x1 = sum_j 2 g_d_j (x1 - x2_j),   g_x2[idx_j] -= 2 g_d_j (x1 - x2_j)
// One wave per query row: the k dot products as wave reductions, the rest on uniform values.  The source-side terms leave as one
// (row, j) contribution each (contrib (B*N, k, 3)); interp_geo_src_kernel adds them per source point in (row, j) order: deterministic.
__global__ __launch_bounds__(256) void interp_geo_bwd_kernel(const float *__restrict__ dist, const int64_t *__restrict__ idx, int ld_tab,
                                                            const float *__restrict__ feat, const float *__restrict__ g_out, int ld_g, int col0,
                                                            const float *__restrict__ xyz1, const float *__restrict__ xyz2, int B, int N, int S, int C,
                                                            int k, float eps, float *__restrict__ g_xyz1, float *__restrict__ contrib) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (row >= B * N) return;
    const int b = row / N;
    const float *d = dist + (size_t)row * ld_tab;
    const int64_t *ix = idx + (size_t)row * ld_tab;
    const float *g = g_out + (size_t)row * ld_g + col0;
    float r[kInterpK], gw[kInterpK];
    int src[kInterpK];
    float R = 0.0f;
#pragma unroll
    for (int j = 0; j < kInterpK; ++j) {
        if (j < k) { r[j] = 1.0f / (d[j] + eps); src[j] = (int)ix[j]; R += r[j]; } else { r[j] = 0.0f; src[j] = 0; }
    }
#pragma unroll
    for (int j = 0; j < kInterpK; ++j) {
        float acc = 0.0f;
        if (j < k) {
            const float *f = feat + ((size_t)b * S + src[j]) * C;
            for (int c = lane; c < C; c += 64) acc = __builtin_fmaf(g[c], f[c], acc);
        }
        gw[j] = wave_sum_f32(acc);
    }
    if (lane != 0) return;
    float mix = 0.0f;
#pragma unroll
    for (int j = 0; j < kInterpK; ++j) if (j < k) mix = __builtin_fmaf(gw[j], r[j] / R, mix);
    const float x = xyz1[(size_t)row * 3], y = xyz1[(size_t)row * 3 + 1], z = xyz1[(size_t)row * 3 + 2];
    float gx = 0.0f, gy = 0.0f, gz = 0.0f;
#pragma unroll
    for (int j = 0; j < kInterpK; ++j) {
        if (j < k) {
            const float gd2 = -2.0f * ((gw[j] - mix) / R) * r[j] * r[j];            // 2 g_d_j
            const float *q = xyz2 + ((size_t)b * S + src[j]) * 3;
            const float dx = (x - q[0]) * gd2, dy = (y - q[1]) * gd2, dz = (z - q[2]) * gd2;
            gx += dx; gy += dy; gz += dz;
            if (contrib) { float *o = contrib + ((size_t)row * k + j) * 3; o[0] = -dx; o[1] = -dy; o[2] = -dz; }
        }
    }
    if (g_xyz1) { g_xyz1[(size_t)row * 3] = gx; g_xyz1[(size_t)row * 3 + 1] = gy; g_xyz1[(size_t)row * 3 + 2] = gz; }
}

// g_xyz2[b][s] = sum over the (row, j) entries of sample b whose source is s, in (row, j) order: one wave per (b, s).
__global__ __launch_bounds__(256) void interp_geo_src_kernel(const int64_t *__restrict__ idx, int ld_tab, const float *__restrict__ contrib, int B, int N,
                                                            int S, int k, float *__restrict__ g_xyz2) {
    const int lane = threadIdx.x & 63;
    const int bs = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (bs >= B * S) return;
    const int b = bs / S, s = bs - b * S;
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    const int total = N * k;
    for (int e = lane; e < total; e += 64) {                        // (lane-strided: a fixed assignment of entries to lanes, fixed lane order below)
        const int n = e / k, j = e - n * k;
        const size_t row = (size_t)b * N + n;
        if ((int)idx[row * ld_tab + j] == s) { const float *c = contrib + (row * k + j) * 3; ax += c[0]; ay += c[1]; az += c[2]; }
    }
    ax = wave_sum_f32(ax); ay = wave_sum_f32(ay); az = wave_sum_f32(az);
    if (lane == 0) { g_xyz2[(size_t)bs * 3] = ax; g_xyz2[(size_t)bs * 3 + 1] = ay; g_xyz2[(size_t)bs * 3 + 2] = az; }
}

// ---- max-pool over the k rows of every group, with its arg-max; the backward writes the whole gradient in one pass -----------------
// x (R, k, C): out[r][c] = max_j x[r][j][c], amax[r][c] = the FIRST j that attains it (k <= 255).  One thread per (r, four columns).
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void group_max_fwd_kernel(const float *__restrict__ x, int R, int k, int C, float *__restrict__ out,
                                                            unsigned char *__restrict__ amax) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int C4 = C >> 2;
    if (t >= (long long)R * C4) return;
    const long long r = t / C4;
    const int c = (int)(t - r * C4) * 4;
    const float *p = x + (r * k) * C + c;
    f32x4 best = *reinterpret_cast<const f32x4 *>(p);
    int bi[4] = {0, 0, 0, 0};
    for (int j0 = 1; j0 < k; j0 += 8) {                       // eight independent row loads in flight
        f32x4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const f32x4 *>(p + (long long)min(j0 + q, k - 1) * C);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (j0 + q < k) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (v[q][e] > best[e] || (v[q][e] != v[q][e] && best[e] == best[e])) { best[e] = v[q][e]; bi[e] = j0 + q; }   // (NaN propagates: the FIRST NaN of a column wins, as torch.max(dim))
            }
    }
    *reinterpret_cast<f32x4 *>(out + r * C + c) = best;
    *reinterpret_cast<uchar4 *>(amax + r * C + c) = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
}

// g_x[r][j][c] = amax[r][c] == j ? g[r][c] : 0 -- every element of g_x written once (no zero-fill launch, no scatter)
__global__ __launch_bounds__(256) void group_max_bwd_kernel(const float *__restrict__ g, const unsigned char *__restrict__ amax, int R, int k, int C,
                                                            float *__restrict__ g_x) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int C4 = C >> 2;
    if (t >= (long long)R * k * C4) return;
    const long long rj = t / C4;
    const int c = (int)(t - rj * C4) * 4;
    const long long r = rj / k;
    const int j = (int)(rj - r * k);
    const f32x4 gv = *reinterpret_cast<const f32x4 *>(g + r * C + c);
    const uchar4 a = *reinterpret_cast<const uchar4 *>(amax + r * C + c);
    f32x4 o;
    o[0] = a.x == j ? gv[0] : 0.0f; o[1] = a.y == j ? gv[1] : 0.0f; o[2] = a.z == j ? gv[2] : 0.0f; o[3] = a.w == j ? gv[3] : 0.0f;
    *reinterpret_cast<f32x4 *>(g_x + rj * C + c) = o;
}

}  // namespace

extern "C" int upp_group_max_fwd(const float *x, int R, int k, int C, float *out, unsigned char *amax, void *stream) {
    if (!x || !out || !amax || R < 1 || k < 1 || C < 1) return UPP_E_BADARG;
    if (k > 255 || C % 4 != 0 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) || (reinterpret_cast<uintptr_t>(amax) & 3)) return UPP_E_RANGE;
    const long long n = (long long)R * (C / 4);
    hipLaunchKernelGGL(group_max_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, R, k, C, out, amax);
    return upp_launch_status();
}

extern "C" int upp_group_max_bwd(const float *g, const unsigned char *amax, int R, int k, int C, float *g_x, void *stream) {
    if (!g || !amax || !g_x || R < 1 || k < 1 || C < 1) return UPP_E_BADARG;
    if (k > 255 || C % 4 != 0 || ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(g_x)) & 15) || (reinterpret_cast<uintptr_t>(amax) & 3)) return UPP_E_RANGE;
    const long long n = (long long)R * k * (C / 4);
    if ((n + 255) / 256 > 0x7fffffffLL) return UPP_E_RANGE;
    hipLaunchKernelGGL(group_max_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, amax, R, k, C, g_x);
    return upp_launch_status();
}

// shared with prop.hip: finalize of (sum, M2) slab partials
int upp_bn_finalize_launch(const float *part, int slabs, int per, int rows, int C, int training, float momentum, float eps,
                           float *running_mean, float *running_var, float *mean, float *rstd, hipStream_t st) {
    hipLaunchKernelGGL(bn_finalize_rows_kernel, dim3((C + 63) / 64), dim3(64 * kBnWaves), 0, st, part, slabs, per, rows, C, training, momentum,
                       eps, running_mean, running_var, mean, rstd);
    return 0;
}

extern "C" long long upp_bn_rows_part_floats(int R, int C) {
    if (R < 1 || C < 1) return 0;
    const int per = bn_per(R);
    return (long long)((R + per - 1) / per) * 2 * C;
}

static int bn_drop_args(float p, const long long *seed, long long seed_add, unsigned salt, BnDrop &d) {
    d = BnDrop{nullptr, 0, 0u, 0u, 1.0f};
    if (!(p >= 0.0f) || p >= 1.0f) return UPP_E_BADARG;
    if (p > 0.0f) {
        if (!seed) return UPP_E_BADARG;
        const double t = (double)p * 4294967296.0;
        d.seed = seed; d.seed_add = seed_add; d.salt = salt; d.thresh = t < 1.0 ? 1u : (t > 4294967295.0 ? 4294967295u : (unsigned)t); d.scale = 1.0f / (1.0f - p);
    }
    return 0;
}

static int bn_rows_fwd_impl(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                            float momentum, float eps, int training, int relu, const BnDrop &dr, float *part, float *mean, float *rstd, float *y,
                            int R, int C, void *stream) {
    if (!x || !mean || !rstd || !y || R < 1 || C < 1) return UPP_E_BADARG;
    if (training && !part) return UPP_E_BADARG;
    if (!training && (!running_mean || !running_var)) return UPP_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int per = bn_per(R), slabs = (R + per - 1) / per;
    if (training) {
        const int Cp = pow2_at_least(C);
        hipLaunchKernelGGL(bn_rows_partial_kernel, dim3(slabs, (C + 255) / 256), dim3(256), 0, st, x, R, C, Cp, per, part);
    }
    upp_bn_finalize_launch(part, slabs, per, R, C, training, momentum, eps, running_mean, running_var, mean, rstd, st);
    const long long total = (long long)R * C;
    if (bn_tall(R, C)) {
        const int rpw = C >= 256 ? 1 : 256 / C;
        const dim3 grid((unsigned)((R + kTallRows * rpw - 1) / (kTallRows * rpw)), (unsigned)(C >= 256 ? C / 256 : 1));
        if (total >= kStreamElems) hipLaunchKernelGGL(bn_rows_apply_tall_kernel<true>, grid, dim3(256), 0, st, x, mean, rstd, gamma, beta, relu, y, R, C, rpw, dr);
        else hipLaunchKernelGGL(bn_rows_apply_tall_kernel<false>, grid, dim3(256), 0, st, x, mean, rstd, gamma, beta, relu, y, R, C, rpw, dr);
        return upp_launch_status();
    }
    hipLaunchKernelGGL(bn_rows_apply_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, st, x, mean, rstd, gamma, beta, relu, y,
                       total, C, total >= kStreamElems ? 1 : 0, dr);
    return upp_launch_status();
}

extern "C" int upp_bn_rows_fwd(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                               float momentum, float eps, int training, int relu, float *part, float *mean, float *rstd, float *y,
                               int R, int C, void *stream) {
    return bn_rows_fwd_impl(x, gamma, beta, running_mean, running_var, momentum, eps, training, relu, BnDrop{nullptr, 0, 0u, 0u, 1.0f}, part, mean, rstd,
                            y, R, C, stream);
}

extern "C" int upp_bn_rows_drop_fwd(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                                    float momentum, float eps, int relu, float p, const long long *seed, long long seed_add, unsigned salt, float *part,
                                    float *mean, float *rstd, float *y, int R, int C, void *stream) {
    BnDrop dr;
    const int rc = bn_drop_args(p, seed, seed_add, salt, dr);
    if (rc) return rc;
    return bn_rows_fwd_impl(x, gamma, beta, running_mean, running_var, momentum, eps, 1, relu, dr, part, mean, rstd, y, R, C, stream);
}

static int bn_rows_bwd_impl(const float *x, const float *g, const float *mean, const float *rstd, const float *gamma, const float *beta,
                            int relu, const BnDrop &dr, float *part, float *g_gamma, float *g_beta, float *g_x, int R, int C, void *stream) {
    if (!x || !g || !mean || !rstd || !part || !g_gamma || !g_beta || R < 1 || C < 1) return UPP_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int per = bn_per(R), slabs = (R + per - 1) / per, Cp = pow2_at_least(C);
    hipLaunchKernelGGL(bn_rows_bwd_partial_kernel, dim3(slabs, (C + 255) / 256), dim3(256), 0, st, x, g, mean, rstd, gamma, beta, relu, R, C, Cp,
                       per, part, dr);
    hipLaunchKernelGGL(bn_rows_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(64 * kBnWaves), 0, st, part, slabs, C, g_gamma, g_beta);
    if (g_x) {
        const long long total = (long long)R * C;
        if (bn_tall(R, C)) {
            const int rpw = C >= 256 ? 1 : 256 / C;
            const dim3 grid((unsigned)((R + kTallRows * rpw - 1) / (kTallRows * rpw)), (unsigned)(C >= 256 ? C / 256 : 1));
            if (total >= kStreamElems)
                hipLaunchKernelGGL(bn_rows_bwd_apply_tall_kernel<true>, grid, dim3(256), 0, st, x, g, mean, rstd, gamma, beta, g_gamma, g_beta, relu,
                                   1.0f / (float)R, g_x, R, C, rpw, dr);
            else
                hipLaunchKernelGGL(bn_rows_bwd_apply_tall_kernel<false>, grid, dim3(256), 0, st, x, g, mean, rstd, gamma, beta, g_gamma, g_beta, relu,
                                   1.0f / (float)R, g_x, R, C, rpw, dr);
            return upp_launch_status();
        }
        hipLaunchKernelGGL(bn_rows_bwd_apply_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, st, x, g, mean, rstd, gamma, beta,
                           g_gamma, g_beta, relu, 1.0f / (float)R, g_x, total, C, total >= kStreamElems ? 1 : 0, dr);
    }
    return upp_launch_status();
}

extern "C" int upp_bn_rows_bwd(const float *x, const float *g, const float *mean, const float *rstd, const float *gamma, const float *beta,
                               int relu, float *part, float *g_gamma, float *g_beta, float *g_x, int R, int C, void *stream) {
    return bn_rows_bwd_impl(x, g, mean, rstd, gamma, beta, relu, BnDrop{nullptr, 0, 0u, 0u, 1.0f}, part, g_gamma, g_beta, g_x, R, C, stream);
}

extern "C" int upp_bn_rows_drop_bwd(const float *x, const float *g, const float *mean, const float *rstd, const float *gamma, const float *beta,
                                    int relu, float p, const long long *seed, long long seed_add, unsigned salt, float *part, float *g_gamma, float *g_beta,
                                    float *g_x, int R, int C, void *stream) {
    BnDrop dr;
    const int rc = bn_drop_args(p, seed, seed_add, salt, dr);
    if (rc) return rc;
    return bn_rows_bwd_impl(x, g, mean, rstd, gamma, beta, relu, dr, part, g_gamma, g_beta, g_x, R, C, stream);
}

extern "C" int upp_sqdist_topk(const float *q, const float *src, float *dist, int64_t *idx, int B, int N, int S, int k, void *stream) {
    if (!q || !src || !dist || !idx || B < 1 || N < 1 || S < 1 || k < 1) return UPP_E_BADARG;
    if (S > 256 || k > S || k > 64) return UPP_E_RANGE;
    const long long rows = (long long)B * N;
    const dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (S <= 64) hipLaunchKernelGGL(sqdist_topk_kernel<1>, grid, dim3(256), 0, st, q, src, N, S, k, dist, idx, rows);
    else if (S <= 128) hipLaunchKernelGGL(sqdist_topk_kernel<2>, grid, dim3(256), 0, st, q, src, N, S, k, dist, idx, rows);
    else hipLaunchKernelGGL(sqdist_topk_kernel<4>, grid, dim3(256), 0, st, q, src, N, S, k, dist, idx, rows);
    return upp_launch_status();
}

extern "C" int upp_interp_fwd(const float *dist, const int64_t *idx, int ld_tab, const float *feat, float *out, int ld_out, int col0,
                              int B, int N, int S, int C, int k, float eps, void *stream) {
    if (!dist || !idx || !feat || !out || B < 1 || N < 1 || S < 1 || C < 1 || k < 1 || ld_tab < k || ld_out < col0 + C || col0 < 0)
        return UPP_E_BADARG;
    if (k > kInterpK || k > S) return UPP_E_RANGE;
    const int Cp = pow2_at_least(C), RL = 256 / Cp, rows = B * N;
    if (k <= 4 && C >= 256 && C % 4 == 0 && ld_out % 4 == 0 && col0 % 4 == 0 && (((uintptr_t)feat | (uintptr_t)out) & 15) == 0) {
        hipLaunchKernelGGL(interp_wide_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dist, idx, ld_tab, feat, S, C, k, eps,
                           (const float *)nullptr, (const float *)nullptr, out, ld_out, col0, rows, N);
        return upp_launch_status();
    }
    const dim3 grid((rows + RL - 1) / RL, (C + 255) / 256);
    if (k <= 4) hipLaunchKernelGGL(interp_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, dist, idx, ld_tab, feat, S, C, Cp, k, eps, out, ld_out, col0, rows, N);
    else if (k <= 8) hipLaunchKernelGGL(interp_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, dist, idx, ld_tab, feat, S, C, Cp, k, eps, out, ld_out, col0, rows, N);
    else hipLaunchKernelGGL(interp_kernel<kInterpK>, grid, dim3(256), 0, (hipStream_t)stream, dist, idx, ld_tab, feat, S, C, Cp, k, eps, out, ld_out, col0, rows, N);
    return upp_launch_status();
}

extern "C" int upp_interp_affine_fwd(const float *dist, const int64_t *idx, int ld_tab, const float *feat, const float *x3, const float *wt,
                                     float *out, int B, int N, int S, int C, int k, float eps, void *stream) {
    if (!dist || !idx || !feat || !x3 || !wt || !out || B < 1 || N < 1 || S < 1 || C < 1 || k < 1 || ld_tab < k) return UPP_E_BADARG;
    if (k > 4 || k > S || C < 256 || C % 4 != 0 || (((uintptr_t)feat | (uintptr_t)out | (uintptr_t)wt) & 15) != 0) return UPP_E_RANGE;
    const int rows = B * N;
    hipLaunchKernelGGL(interp_wide_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dist, idx, ld_tab, feat, S, C, k, eps, x3, wt,
                       out, C, 0, rows, N);
    return upp_launch_status();
}

extern "C" int upp_interp_bwd(const float *dist, const int64_t *idx, int ld_tab, const float *g_out, int ld_g, int col0, float *g_feat,
                              int B, int N, int S, int C, int k, float eps, void *stream) {
    if (!dist || !idx || !g_out || !g_feat || B < 1 || N < 1 || S < 1 || C < 1 || k < 1 || ld_tab < k || ld_g < col0 + C || col0 < 0)
        return UPP_E_BADARG;
    if (k > kInterpK || k > S || N > kInterpMaxRows || (long long)B * S > 0x7fffffffLL) return UPP_E_RANGE;
    {   // few source rows and narrow, 16-byte aligned features: the dense-in-LDS kernel (see interp_bwd_dense_kernel)
        const bool aligned = C % 4 == 0 && S % 4 == 0 && ld_g % 4 == 0 && col0 % 4 == 0 &&
                             ((reinterpret_cast<uintptr_t>(g_out) | reinterpret_cast<uintptr_t>(g_feat)) & 15) == 0;
        int TR = (int)((60 * 1024) / ((size_t)(S + C + 2 * kInterpK) * sizeof(float)));
        TR = TR > 256 ? 256 : TR / 8 * 8;
        if (aligned && TR >= 64 && (long long)(S / 4) * (C / 4) <= 256 && (size_t)4 * S * C <= (size_t)TR * (S + C)) {
            const size_t lds = (size_t)TR * (S + C + 2 * kInterpK) * sizeof(float);
            if ((size_t)8 * S * C <= (size_t)TR * (S + C))      // eight waves when their partial sums fit the weight / gradient area
                hipLaunchKernelGGL(interp_bwd_dense_kernel<8>, dim3((unsigned)B), dim3(512), lds, (hipStream_t)stream, dist, idx, ld_tab, g_out, ld_g,
                                   col0, S, C, k, eps, N, TR, g_feat);
            else
                hipLaunchKernelGGL(interp_bwd_dense_kernel<4>, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, dist, idx, ld_tab, g_out, ld_g,
                                   col0, S, C, k, eps, N, TR, g_feat);
            return upp_launch_status();
        }
    }
    // (round 6 measured a PUSH form for the wide rows of the segmentation head -- a workgroup streams a sample's gradient rows once and
    //  accumulates into an [S][columns] LDS tile: 693 us as register read-modify-writes (a 6,144-long LDS latency chain per wave), 1,780 us
    //  with ds_add_f32, against 226 us for the pull kernel below, whose re-reads are served by the L2: NOTEBOOK section 12)
    hipLaunchKernelGGL(interp_bwd_kernel, dim3((unsigned)(B * S)), dim3(256), 0, (hipStream_t)stream, dist, idx, ld_tab, g_out, ld_g, col0, S, C,
                       pow2_at_least(C), k, eps, N, g_feat);
    return upp_launch_status();
}

extern "C" int upp_posenc_fwd(const float *x, const float *freqs, int F, float *out, int ld_out, int col0, long long rows, void *stream) {
    if (!x || !out || rows < 1 || F < 0 || (F > 0 && !freqs) || col0 < 0 || ld_out < col0 + 3 * (2 * F + 1)) return UPP_E_BADARG;
    if (F > 8) return UPP_E_RANGE;
    Freqs fr;
    for (int q = 0; q < 8; ++q) fr.f[q] = q < F ? freqs[q] : 0.0f;       // freqs is a HOST array (a handful of scalars)
    const long long total = rows * 3;
    hipLaunchKernelGGL(posenc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, fr, F, out, ld_out, col0, total);
    return upp_launch_status();
}

extern "C" int upp_interp_geo_bwd(const float *dist, const int64_t *idx, int ld_tab, const float *feat, const float *g_out, int ld_g, int col0,
                                  const float *xyz1, const float *xyz2, int B, int N, int S, int C, int k, float eps, float *g_xyz1,
                                  float *g_xyz2, float *contrib, void *stream) {
    if (!dist || !idx || !feat || !g_out || !xyz1 || !xyz2 || B < 0 || N < 1 || S < 1 || C < 1 || k < 1) return UPP_E_BADARG;
    if (k > kInterpK || k > S || ld_tab < k || ld_g < col0 + C) return UPP_E_RANGE;
    if (!g_xyz1 && !g_xyz2) return UPP_E_BADARG;
    if (g_xyz2 && !contrib) return UPP_E_BADARG;
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(interp_geo_bwd_kernel, dim3((unsigned)((B * N + 3) / 4)), dim3(256), 0, st, dist, idx, ld_tab, feat, g_out, ld_g, col0, xyz1, xyz2,
                       B, N, S, C, k, eps, g_xyz1, g_xyz2 ? contrib : nullptr);
    if (g_xyz2)
        hipLaunchKernelGGL(interp_geo_src_kernel, dim3((unsigned)((B * S + 3) / 4)), dim3(256), 0, st, idx, ld_tab, contrib, B, N, S, k, g_xyz2);
    return upp_launch_status();
}
