// emd.hip -- approximate earth mover's distance (auction-style soft matching) for gfx950.
// Replaces emd_cuda.approxmatch_forward / matchcost_forward / matchcost_backward of the
// reference extension (extensions/emd/cuda/emd_kernel.cu:24-157, :199-242, :285-354;
// bindings emd.cpp:23-27).
//
// The CUDA kernel runs ONE 512-thread block per cloud through 10 levels x 3 passes and
// read-modify-writes the (m x n) match matrix in every level.  Here:
//   * every pass is its own launch over (point tile, cloud) -> thousands of waves;
//     pass 3 of level i and pass 1 of level i+1 share their pair loop (one distance,
//     two exponentials), so approxmatch is 1 + 2 + 9*2 + 1 = 22 launches;
//   * match is never read-modify-written: the per-level row/column factors
//     ratioL_i[k], ratioR_i[l] (10*(n+m) floats per cloud) are kept in the work
//     buffer and match[l,k] = sum_i exp(level_i*d) * ratioL_i[k] * ratioR_i[l] is
//     accumulated in registers in level order and stored exactly once
//     (B*n*m*4 bytes of HBM writes instead of 20x that).
// Sums over the other cloud are split over the 4 waves of a workgroup and combined
// in wave order, so results are run-to-run deterministic but the float summation
// order differs from the CUDA kernel's: EMD parity is tolerance-based (the
// reference itself uses the approximate __expf, as does this file).
#include "common.h"

namespace {

constexpr int kLevels = 10;
constexpr int kT = 1024;  // points of the "other" cloud per LDS tile

__device__ __forceinline__ float level_of(int it) {  // -4^(7-it), last level 0  (emd_kernel.cu:45-49)
    return it == kLevels - 1 ? 0.0f : (it == kLevels - 2 ? -0.25f : -(float)(1 << (2 * (7 - it))));
}

struct Work {
    float *remainL, *remainR, *ratioL, *ratioR;  // [B][n], [B][m], [10][B][n], [10][B][m]
};
__host__ __device__ inline Work carve(float *w, int B, int n, int m) {
    Work k;
    k.remainL = w;
    k.remainR = k.remainL + (size_t)B * n;
    k.ratioL = k.remainR + (size_t)B * m;
    k.ratioR = k.ratioL + (size_t)kLevels * B * n;
    return k;
}

__global__ void emd_init_kernel(float *remainL, float *remainR, long long nl, long long nr, float multiL, float multiR) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nl + nr; i += (long long)gridDim.x * blockDim.x) {
        if (i < nl) remainL[i] = multiL; else remainR[i - nl] = multiR;
    }
}

// Row pass, lane = point k of xyz1.  If it > 0 first finishes level it-1
// (pass 3, emd_kernel.cu:120-153: remainL -= sum_l w) then does pass 1 of level
// `it` (:50-83: ratioL = remainL / (1e-9 + sum_l exp(level*d) * remainR[l])).
__global__ __launch_bounds__(256) void emd_row_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2, Work wk,
                                                      int B, int n, int m, int it) {
    __shared__ float4 tile[kT];   // x, y, z, remainR
    __shared__ float tr[kT];      // ratioR of the previous level
    __shared__ float part[2][4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: loop counters and LDS offsets in SGPRs
    int tile_x, b;
    xcd_cloud_tile(tile_x, b);                            // whole clouds per XCD (common.h)
    const int k = tile_x * 64 + lane;
    const int kc = k < n ? k : n - 1;
    const float *a = xyz1 + ((size_t)b * n + kc) * 3;
    const float x1 = a[0], y1 = a[1], z1 = a[2];
    const float lev = level_of(it);
    const float levp = it > 0 ? level_of(it - 1) : 0.0f;
    const float rlp = it > 0 ? wk.ratioL[((size_t)(it - 1) * B + b) * n + kc] : 0.0f;
    const float *rR = wk.remainR + (size_t)b * m;
    const float *ratRp = it > 0 ? wk.ratioR + ((size_t)(it - 1) * B + b) * m : nullptr;
    float s1 = 0.0f, s3 = 0.0f;
    for (int c0 = 0; c0 < m; c0 += kT) {
        const int len = min(kT, m - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < len; i += 256) {
            const float *s = xyz2 + ((size_t)b * m + c0 + i) * 3;
            tile[i] = make_float4(s[0], s[1], s[2], rR[c0 + i]);
            tr[i] = ratRp ? ratRp[c0 + i] : 0.0f;
        }
        __syncthreads();
        const int seg = (len + 3) >> 2, l0 = wave * seg, l1 = min(len, l0 + seg);
        if (it > 0) {
#pragma unroll 2
            for (int l = l0; l < l1; ++l) {
                const float4 p = tile[l];
                const float d = sumsq3(p.x - x1, p.y - y1, p.z - z1);
                s1 = __builtin_fmaf(__expf(lev * d), p.w, s1);
                s3 = __builtin_fmaf(__expf(levp * d) * rlp, tr[l], s3);
            }
        } else {
#pragma unroll 2
            for (int l = l0; l < l1; ++l) {
                const float4 p = tile[l];
                const float d = sumsq3(p.x - x1, p.y - y1, p.z - z1);
                s1 = __builtin_fmaf(__expf(lev * d), p.w, s1);
            }
        }
    }
    part[0][wave][lane] = s1;
    part[1][wave][lane] = s3;
    __syncthreads();
    if (wave == 0 && k < n) {
        const float suml = (((1e-9f + part[0][0][lane]) + part[0][1][lane]) + part[0][2][lane]) + part[0][3][lane];
        float rem = wk.remainL[(size_t)b * n + k];
        if (it > 0) {
            const float sw = ((part[1][0][lane] + part[1][1][lane]) + part[1][2][lane]) + part[1][3][lane];
            rem = fmaxf(0.0f, rem - sw);
            wk.remainL[(size_t)b * n + k] = rem;
        }
        wk.ratioL[((size_t)it * B + b) * n + k] = rem / suml;
    }
}

// Column pass (pass 2, emd_kernel.cu:85-118), lane = point l of xyz2.
__global__ __launch_bounds__(256) void emd_col_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2, Work wk,
                                                      int B, int n, int m, int it) {
    __shared__ float4 tile[kT];  // x, y, z, ratioL
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: loop counters and LDS offsets in SGPRs
    int tile_x, b;
    xcd_cloud_tile(tile_x, b);                            // whole clouds per XCD (common.h)
    const int l = tile_x * 64 + lane;
    const int lc = l < m ? l : m - 1;
    const float *a = xyz2 + ((size_t)b * m + lc) * 3;
    const float x2 = a[0], y2 = a[1], z2 = a[2];
    const float lev = level_of(it);
    const float *ratL = wk.ratioL + ((size_t)it * B + b) * n;
    float s = 0.0f;
    for (int c0 = 0; c0 < n; c0 += kT) {
        const int len = min(kT, n - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < len; i += 256) {
            const float *p = xyz1 + ((size_t)b * n + c0 + i) * 3;
            tile[i] = make_float4(p[0], p[1], p[2], ratL[c0 + i]);
        }
        __syncthreads();
        const int seg = (len + 3) >> 2, k0 = wave * seg, k1 = min(len, k0 + seg);
#pragma unroll 2
        for (int k = k0; k < k1; ++k) {
            const float4 p = tile[k];
            const float d = sumsq3(x2 - p.x, y2 - p.y, z2 - p.z);
            s = __builtin_fmaf(__expf(lev * d), p.w, s);
        }
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && l < m) {
        float sumr = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
        const float rem = wk.remainR[(size_t)b * m + l];
        sumr *= rem;
        const float consumption = fminf(rem / (sumr + 1e-9f), 1.0f);
        wk.ratioR[((size_t)it * B + b) * m + l] = consumption * rem;
        wk.remainR[(size_t)b * m + l] = fmaxf(0.0f, rem - sumr);
    }
}

// match[b,l,k] = sum_it (exp(level_it * d) * ratioL_it[k]) * ratioR_it[l], level order,
// stored once.  lane = k (coalesced rows of match), each wave walks its own rows l.
constexpr int kMT = 64;  // rows l per workgroup
__global__ __launch_bounds__(256) void emd_match_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2, Work wk,
                                                        float *__restrict__ match, int B, int n, int m) {
    __shared__ float4 pt[kMT];
    __shared__ float rr[kLevels][kMT];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: loop counters and LDS offsets in SGPRs
    // (3-D grid: whole clouds per XCD, as xcd_cloud_tile does for the 2-D ones)
    const int per_cloud = (int)(gridDim.x * gridDim.y);
    const int lin = xcd_contiguous((int)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x), per_cloud * (int)gridDim.z);
    const int b = lin / per_cloud, in_cloud = lin - b * per_cloud;
    const int tile_y = in_cloud / (int)gridDim.x, tile_x = in_cloud - tile_y * (int)gridDim.x;
    const int k = tile_x * 64 + lane;
    const int kc = k < n ? k : n - 1;
    const int l_base = tile_y * kMT;
    const int len = min(kMT, m - l_base);
    const float *a = xyz1 + ((size_t)b * n + kc) * 3;
    const float x1 = a[0], y1 = a[1], z1 = a[2];
    float rl[kLevels];
#pragma unroll
    for (int it = 0; it < kLevels; ++it) rl[it] = wk.ratioL[((size_t)it * B + b) * n + kc];
    for (int i = threadIdx.x; i < len; i += 256) {
        const float *s = xyz2 + ((size_t)b * m + l_base + i) * 3;
        pt[i] = make_float4(s[0], s[1], s[2], 0.0f);
    }
    for (int i = threadIdx.x; i < kLevels * len; i += 256) {
        const int it = i / len, li = i - it * len;
        rr[it][li] = wk.ratioR[((size_t)it * B + b) * m + l_base + li];
    }
    __syncthreads();
    for (int li = wave; li < len; li += 4) {
        const float4 p = pt[li];
        const float d = sumsq3(p.x - x1, p.y - y1, p.z - z1);
        float acc = 0.0f;
#pragma unroll
        for (int it = 0; it < kLevels; ++it) acc = __builtin_fmaf(__expf(level_of(it) * d) * rl[it], rr[it][li], acc);
        if (k < n) match[((size_t)b * m + l_base + li) * n + k] = acc;
    }
}

// cost[b] += sum_{k,l} d(k,l) * match[l,k]   (emd_kernel.cu:199-242).  lane = k.
__global__ __launch_bounds__(256) void emd_cost_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                       const float *__restrict__ match, float *__restrict__ cost, int n, int m) {
    __shared__ float4 tile[kT];
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: loop counters and LDS offsets in SGPRs
    int tile_x, b;
    xcd_cloud_tile(tile_x, b);                            // whole clouds per XCD (common.h)
    const int k = tile_x * 64 + lane;
    const int kc = k < n ? k : n - 1;
    const float *a = xyz1 + ((size_t)b * n + kc) * 3;
    const float x1 = a[0], y1 = a[1], z1 = a[2];
    const float *mt = match + (size_t)b * n * m;
    float s = 0.0f;
    for (int c0 = 0; c0 < m; c0 += kT) {
        const int len = min(kT, m - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < len; i += 256) {
            const float *p = xyz2 + ((size_t)b * m + c0 + i) * 3;
            tile[i] = make_float4(p[0], p[1], p[2], 0.0f);
        }
        __syncthreads();
        const int seg = (len + 3) >> 2, l0 = wave * seg, l1 = min(len, l0 + seg);
#pragma unroll 4
        for (int l = l0; l < l1; ++l) {
            const float4 p = tile[l];
            const float d = sumsq3(p.x - x1, p.y - y1, p.z - z1);
            s = __builtin_fmaf(d, mt[(size_t)(c0 + l) * n + kc], s);
        }
    }
    if (k >= n) s = 0.0f;
    const float ws = wave_sum_f32(s);
    if (lane == 0) part[wave] = ws;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&cost[b], (part[0] + part[1]) + (part[2] + part[3]));
}

// grad1[b,l] = gc[b] * sum_k 2*match[k,l] * (p1_l - p2_k)   (matchcostgrad1, :332-354).  lane = l.
__global__ __launch_bounds__(256) void emd_grad1_kernel(const float *__restrict__ gc, const float *__restrict__ xyz1,
                                                        const float *__restrict__ xyz2, const float *__restrict__ match,
                                                        float *__restrict__ grad1, int n, int m) {
    __shared__ float4 tile[kT];
    __shared__ float part[3][4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: loop counters and LDS offsets in SGPRs
    int tile_x, b;
    xcd_cloud_tile(tile_x, b);                            // whole clouds per XCD (common.h)
    const int l = tile_x * 64 + lane;
    const int lc = l < n ? l : n - 1;
    const float *a = xyz1 + ((size_t)b * n + lc) * 3;
    const float x1 = a[0], y1 = a[1], z1 = a[2];
    const float *mt = match + (size_t)b * n * m;
    float dx = 0.0f, dy = 0.0f, dz = 0.0f;
    for (int c0 = 0; c0 < m; c0 += kT) {
        const int len = min(kT, m - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < len; i += 256) {
            const float *p = xyz2 + ((size_t)b * m + c0 + i) * 3;
            tile[i] = make_float4(p[0], p[1], p[2], 0.0f);
        }
        __syncthreads();
        const int seg = (len + 3) >> 2, k0 = wave * seg, k1 = min(len, k0 + seg);
#pragma unroll 4
        for (int k = k0; k < k1; ++k) {
            const float4 p = tile[k];
            const float d = mt[(size_t)(c0 + k) * n + lc] * 2;
            dx = __builtin_fmaf(x1 - p.x, d, dx);
            dy = __builtin_fmaf(y1 - p.y, d, dy);
            dz = __builtin_fmaf(z1 - p.z, d, dz);
        }
    }
    part[0][wave][lane] = dx; part[1][wave][lane] = dy; part[2][wave][lane] = dz;
    __syncthreads();
    if (wave == 0 && l < n) {
        const float g = gc[b];
        float *o = grad1 + ((size_t)b * n + l) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            o[c] = (((part[c][0][lane] + part[c][1][lane]) + part[c][2][lane]) + part[c][3][lane]) * g;
    }
}

// grad2[b,k] = gc[b] * sum_j 2*match[k,j] * (p2_k - p1_j)   (matchcostgrad2, :285-326).
// One wave per row k of match; lanes stride the contiguous j.
__global__ __launch_bounds__(256) void emd_grad2_kernel(const float *__restrict__ gc, const float *__restrict__ xyz1,
                                                        const float *__restrict__ xyz2, const float *__restrict__ match,
                                                        float *__restrict__ grad2, int n, int m) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: loop counters and LDS offsets in SGPRs
    int tile_x, b;
    xcd_cloud_tile(tile_x, b);                            // whole clouds per XCD (common.h)
    const int k = tile_x * 4 + wave;
    if (k >= m) return;
    const float *q = xyz2 + ((size_t)b * m + k) * 3;
    const float x2 = q[0], y2 = q[1], z2 = q[2];
    const float *row = match + ((size_t)b * m + k) * n;
    const float *p1 = xyz1 + (size_t)b * n * 3;
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    for (int j = lane; j < n; j += 64) {
        const float d = row[j] * 2;
        sx = __builtin_fmaf(x2 - p1[j * 3 + 0], d, sx);
        sy = __builtin_fmaf(y2 - p1[j * 3 + 1], d, sy);
        sz = __builtin_fmaf(z2 - p1[j * 3 + 2], d, sz);
    }
    sx = wave_sum_f32(sx); sy = wave_sum_f32(sy); sz = wave_sum_f32(sz);
    if (lane == 0) {
        const float g = gc[b];
        float *o = grad2 + ((size_t)b * m + k) * 3;
        o[0] = sx * g; o[1] = sy * g; o[2] = sz * g;
    }
}

}  // namespace

extern "C" long long upp_emd_work_floats(int B, int n, int m) {
    if (B < 0 || n < 0 || m < 0) return 0;
    return (long long)B * ((long long)n + m) * (kLevels + 1);
}

extern "C" int upp_emd_approxmatch(const float *xyz1, const float *xyz2, float *match, float *work, int B, int n, int m,
                                   void *stream) {
    if (!xyz1 || !xyz2 || !match || !work || B < 0 || n < 1 || m < 1) return UPP_E_BADARG;
    if (B == 0) return 0;
    if (B > 65535) return UPP_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    const Work wk = carve(work, B, n, m);
    float multiL, multiR;  // integer division, emd_kernel.cu:28-34
    if (n >= m) { multiL = 1.0f; multiR = (float)(n / m); } else { multiL = (float)(m / n); multiR = 1.0f; }
    const long long nl = (long long)B * n, nr = (long long)B * m;
    long long ig = (nl + nr + 255) / 256; if (ig > 2048) ig = 2048;
    hipLaunchKernelGGL(emd_init_kernel, dim3((int)ig), dim3(256), 0, st, wk.remainL, wk.remainR, nl, nr, multiL, multiR);
    const dim3 grow((n + 63) / 64, B), gcol((m + 63) / 64, B);
    for (int it = 0; it < kLevels; ++it) {
        hipLaunchKernelGGL(emd_row_kernel, grow, dim3(256), 0, st, xyz1, xyz2, wk, B, n, m, it);
        hipLaunchKernelGGL(emd_col_kernel, gcol, dim3(256), 0, st, xyz1, xyz2, wk, B, n, m, it);
    }
    // (pass 3 of the last level only updates remainL, which nothing reads afterwards)
    hipLaunchKernelGGL(emd_match_kernel, dim3((n + 63) / 64, (m + kMT - 1) / kMT, B), dim3(256), 0, st, xyz1, xyz2, wk, match, B, n, m);
    return upp_launch_status();
}

extern "C" int upp_emd_matchcost(const float *xyz1, const float *xyz2, const float *match, float *cost, int B, int n, int m,
                                 void *stream) {
    if (!xyz1 || !xyz2 || !match || !cost || B < 0 || n < 1 || m < 1) return UPP_E_BADARG;
    if (B == 0) return 0;
    if (B > 65535) return UPP_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    upp_zero_async(cost, B, st);                                     // (a kernel, not a memset node: common.h)
    hipLaunchKernelGGL(emd_cost_kernel, dim3((n + 63) / 64, B), dim3(256), 0, st, xyz1, xyz2, match, cost, n, m);
    return upp_launch_status();
}

extern "C" int upp_emd_matchcost_bwd(const float *grad_cost, const float *xyz1, const float *xyz2, const float *match,
                                     float *grad1, float *grad2, int B, int n, int m, void *stream) {
    if (!grad_cost || !xyz1 || !xyz2 || !match || !grad1 || !grad2 || B < 0 || n < 1 || m < 1) return UPP_E_BADARG;
    if (B == 0) return 0;
    if (B > 65535) return UPP_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(emd_grad1_kernel, dim3((n + 63) / 64, B), dim3(256), 0, st, grad_cost, xyz1, xyz2, match, grad1, n, m);
    hipLaunchKernelGGL(emd_grad2_kernel, dim3((m + 3) / 4, B), dim3(256), 0, st, grad_cost, xyz1, xyz2, match, grad2, n, m);
    return upp_launch_status();
}
