// optim.hip -- gradient clipping + AdamW over ONE flat parameter / gradient buffer.
//
// The reference step (tools/runner_module.py:202-207, tools/builder.py:40-55) is
//   clip_grad_norm_(parameters, 10, norm_type=2); AdamW(two groups: weight decay 0 / 0.05).step()
// On 122 small trainable tensors torch's graph-capturable AdamW issues ~260 kernels per step (244 of them
// 0-dim divisions for the per-parameter bias corrections).  All trainable parameters, gradients and both
// moments live in flat buffers here (no-decay parameters first, then the decayed ones), so the step is:
//   sumsq partials -> (1 block) total norm, clip coefficient, step += 1, bias corrections -> fused update.
// The update is torch.optim.AdamW's (decoupled decay, lerp first moment, eps added after the bias-corrected
// sqrt), written so that every rounding happens where torch's foreach implementation has it.
#include "common.h"

namespace {

constexpr int kRedBlocks = 1024;

// sum of squares: 16-byte loads, four independent chains per thread (a single fmaf chain over one load at a time is latency-bound:
// 108 us for the 22 M gradients of the pre-training recipe), fixed combination order -> deterministic
__global__ __launch_bounds__(256) void sumsq_kernel(const float *__restrict__ g, long long n, float *__restrict__ part) {
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    const long long n4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? n / 4 : 0;
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        const float4 a = g4[i], b = g4[i + stride];
        s0 = __builtin_fmaf(a.x, a.x, s0); s1 = __builtin_fmaf(a.y, a.y, s1); s2 = __builtin_fmaf(a.z, a.z, s2); s3 = __builtin_fmaf(a.w, a.w, s3);
        s0 = __builtin_fmaf(b.x, b.x, s0); s1 = __builtin_fmaf(b.y, b.y, s1); s2 = __builtin_fmaf(b.z, b.z, s2); s3 = __builtin_fmaf(b.w, b.w, s3);
    }
    for (; i < n4; i += stride) {
        const float4 a = g4[i];
        s0 = __builtin_fmaf(a.x, a.x, s0); s1 = __builtin_fmaf(a.y, a.y, s1); s2 = __builtin_fmaf(a.z, a.z, s2); s3 = __builtin_fmaf(a.w, a.w, s3);
    }
    for (long long k = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) s0 = __builtin_fmaf(g[k], g[k], s0);
    float s = (s0 + s1) + (s2 + s3);
    __shared__ float red[4];
    s = wave_sum_f32(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// state: [0] step (float), [1] grad norm, [2] clip coefficient, [3] bias_correction1, [4] sqrt(bias_correction2)
__global__ __launch_bounds__(256) void adamw_prepare_kernel(const float *__restrict__ part, int nparts, float *__restrict__ state,
                                                            float max_norm, float beta1, float beta2) {
    __shared__ float red[4];
    float s = 0.0f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = wave_sum_f32(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
        const float step = state[0] + 1.0f;
        state[0] = step;
        state[1] = norm;
        state[2] = max_norm > 0.0f ? fminf(max_norm / (norm + 1e-6f), 1.0f) : 1.0f;   // clip_grad_norm_
        state[3] = 1.0f - powf(beta1, step);
        state[4] = sqrtf(1.0f - powf(beta2, step));
    }
}

__global__ __launch_bounds__(256) void adamw_update_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                           float *__restrict__ v, long long n, long long split,
                                                           const float *__restrict__ state, float lr, float beta1, float beta2,
                                                           float eps, float wd) {
    const float clip = state[2], bc1 = state[3], sbc2 = state[4];
    if (lr < 0.0f) { lr = state[5]; wd = state[6]; }        // schedule-driven values live on the device (HIP-graph replays see updates)
    const float step_size = lr / bc1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * clip;                       // the clipped gradient is what stays in the buffer
        g[i] = gi;
        float pi = p[i] * (1.0f - lr * (i >= split ? wd : 0.0f));
        const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);        // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = __builtin_fmaf(gi * gi, 1.0f - beta2, v[i] * beta2);  // mul_(beta2).addcmul_(g, g, 1 - beta2)
        const float denom = sqrtf(vi) / sbc2 + eps;
        pi = pi - step_size * (mi / denom);                  // addcdiv_(exp_avg, denom, value=-step_size)
        m[i] = mi; v[i] = vi; p[i] = pi;
    }
}

// Column sums of many small partial-result matrices in ONE launch: job j reduces the n_j rows (row stride ld_j) of
// src_j into dst_j (len_j), rows added in ascending order; dst is overwritten or accumulated into.  Jobs that share a
// destination form a GROUP served by the same workgroups, one job after the other in submission order -- so several
// partial matrices (e.g. the cls-position gradient of 12 blocks) can target one buffer without a race, and the
// result is deterministic.  The parameter-gradient partials of a backward pass (adapter weight partials per workgroup,
// LayerNorm gamma/beta partials per chunk, per-sample prompt gradients) are summed here, after the pass, instead of by
// ~3 tiny reduce launches per transformer block.
constexpr int kMaxSumJobs = 64;
struct SumJobs {
    const float *src[kMaxSumJobs];
    int n[kMaxSumJobs], ld[kMaxSumJobs];
    float *dst[kMaxSumJobs];                   // per group
    int len[kMaxSumJobs], acc[kMaxSumJobs];    // per group
    int dw[kMaxSumJobs], dpitch[kMaxSumJobs];  // per group: dw > 0: the destination is a (len / dw, dw) WINDOW of rows dpitch floats apart --
                                               // flat element c lives at (c / dw) * dpitch + c % dw (a column range of a wider gradient matrix)
    int first[kMaxSumJobs + 1];                // per group: its jobs are [first[g], first[g+1])
    int wg0[kMaxSumJobs + 1];                  // per group: first workgroup (256 columns per workgroup)
    int groups;
};
// V = 4: long rows whose sources, strides and destination are 16-byte aligned (the partial tiles of the grouped weight gradients: 1-3
// partial matrices of 147,456 ... 589,824 elements each in the pre-training step) -- 1,024 columns per workgroup, 16 bytes per lane; the
// one-dword-per-lane form spent its time on workgroup start-up (2,304 workgroups of one to three loads per thread: 0.7 TB/s).  Same sums.
template <int V>
__global__ __launch_bounds__(256) void batched_sum_kernel(SumJobs t) {
    typedef float vec_t __attribute__((ext_vector_type(V)));
    int g = 0;
    while (g + 1 < t.groups && (int)blockIdx.x >= t.wg0[g + 1]) ++g;          // wave-uniform scan of <= 64 entries
    const int c = (((int)blockIdx.x - t.wg0[g]) * 256 + threadIdx.x) * V;
    const int len = t.len[g];
    if (c >= len) return;
    const int dw = t.dw[g];
    float *dptr = t.dst[g] + (dw > 0 ? (size_t)(c / dw) * t.dpitch[g] + c % dw : (size_t)c);      // (V = 4: dw % 4 == 0, dpitch % 4 == 0 -- checked by the host)
    vec_t acc = t.acc[g] ? *reinterpret_cast<const vec_t *>(dptr) : vec_t(0.0f);
    for (int j = t.first[g]; j < t.first[g + 1]; ++j) {
        const float *src = t.src[j] + c;
        const int n = t.n[j], ld = t.ld[j];
        constexpr int U = V == 1 ? 16 : 8;                                      // independent loads in flight
        for (int i0 = 0; i0 < n; i0 += U) {
            vec_t v[U];
#pragma unroll
            for (int q = 0; q < U; ++q) v[q] = *reinterpret_cast<const vec_t *>(src + (size_t)min(i0 + q, n - 1) * ld);
#pragma unroll
            for (int q = 0; q < U; ++q) if (i0 + q < n) acc += v[q];
        }
    }
    *reinterpret_cast<vec_t *>(dptr) = acc;
}

// Tall jobs (more than kTallRows rows: the bias gradients of trainable Linear layers are column sums of (B L, N) output
// gradients): 64 columns per workgroup and the four waves split the rows in 16-row chunks (wave w takes chunks w, w+4, ...);
// the four partial sums are combined in wave order, so the result is deterministic.
constexpr int kTallRows = 512;
__global__ __launch_bounds__(256) void batched_sum_tall_kernel(SumJobs t) {
    __shared__ float part[4][64];
    int g = 0;
    while (g + 1 < t.groups && (int)blockIdx.x >= t.wg0[g + 1]) ++g;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = ((int)blockIdx.x - t.wg0[g]) * 64 + lane;
    const int len = t.len[g];
    const int cc = min(c, len - 1);
    float acc = 0.0f;
    for (int j = t.first[g]; j < t.first[g + 1]; ++j) {
        const float *src = t.src[j] + cc;
        const int n = t.n[j], ld = t.ld[j];
        for (int i0 = wave * 16; i0 < n; i0 += 64) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = src[(size_t)min(i0 + q, n - 1) * ld];
#pragma unroll
            for (int q = 0; q < 16; ++q) if (i0 + q < n) acc += v[q];
        }
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < len) {
        const float s = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
        const int dw = t.dw[g];
        float *dptr = t.dst[g] + (dw > 0 ? (size_t)(c / dw) * t.dpitch[g] + c % dw : (size_t)c);
        *dptr = t.acc[g] ? *dptr + s : s;
    }
}

// Column sums of ONE very tall matrix (the bias gradient of a trainable Linear over 65,536 point rows is the column sum of its
// output gradient): chunk ch of the rows -> dst[ch][c]; the caller sums the `chunks` partial rows (upp_batched_sum).  grid =
// (ceil(len / 64), chunks); the four waves take the chunk's rows in 16-row batches and are combined in wave order.
// (accumulated in f64: these sums replace torch reductions -- ops.sum_rows -- whose pairwise order loses less to cancellation than a
//  sequential f32 sum; the kernel is bound by its loads)
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float *__restrict__ src, long long ld, int n, int len, float *__restrict__ dst) {
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x * 64 + lane, cc = min(c, len - 1);
    const int chunks = gridDim.y, per = (n + chunks - 1) / chunks;
    const int r0 = blockIdx.y * per, r1 = min(n, r0 + per);
    double acc = 0.0;
    for (int i0 = r0 + wave * 16; i0 < r1; i0 += 64) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = src[(size_t)min(i0 + q, r1 - 1) * ld + cc];
#pragma unroll
        for (int q = 0; q < 16; ++q) if (i0 + q < r1) acc += (double)v[q];
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < len) dst[(size_t)blockIdx.y * len + c] = (float)(((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]);
}

// Weighted column sums of ONE very tall matrix: dst[ch][w][c] = sum over the rows r of chunk ch of wts[r][w] * src[r][c], w < W <= 4 --
// the weight gradient of a rank-W update  y += x (rows, W) . wt (W, C)  over the 65,536 label points of the segmentation head
// (the xyz columns of the commuted first feature-propagation layer: reference models/Point_MAE_unify_segment.py:420,605 under
// autograd) is x^T . g: W weighted column sums of g.  HBM-bound: every element of g is read once, 16 bytes per lane; grid =
// (ceil(len / 256), chunks); lane -> 4 columns, the four waves take every fourth row and are combined in wave order.
template <int W>
__global__ __launch_bounds__(256) void wcolsum_partials_kernel(const float *__restrict__ src, long long ld, const float *__restrict__ wts,
                                                               long long ldw, int n, int len, float *__restrict__ dst) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ f32x4 part[4][W][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x * 256 + 4 * lane, cc = min(c, len - 4);
    const int chunks = gridDim.y, per = (n + chunks - 1) / chunks;
    const int r0 = blockIdx.y * per, r1 = min(n, r0 + per);
    f32x4 acc[W];
#pragma unroll
    for (int w = 0; w < W; ++w) acc[w] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int i0 = r0 + wave; i0 < r1; i0 += 4 * 8) {
        f32x4 v[8];
        float x[8][W];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = min(i0 + 4 * q, r1 - 1);
            v[q] = *reinterpret_cast<const f32x4 *>(src + (size_t)i * ld + cc);
#pragma unroll
            for (int w = 0; w < W; ++w) x[q][w] = wts[(size_t)i * ldw + w];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (i0 + 4 * q < r1) {
#pragma unroll
                for (int w = 0; w < W; ++w) acc[w] += x[q][w] * v[q];
            }
    }
#pragma unroll
    for (int w = 0; w < W; ++w) part[wave][w][lane] = acc[w];
    __syncthreads();
    if (wave == 0 && c < len) {
#pragma unroll
        for (int w = 0; w < W; ++w)
            *reinterpret_cast<f32x4 *>(dst + ((size_t)blockIdx.y * W + w) * len + c) = ((part[0][w][lane] + part[1][w][lane]) + part[2][w][lane]) + part[3][w][lane];
    }
}

// Many small device-to-device copies in one launch (the 16 hand-over tensors a pipelined training step passes from its front-end graph
// to its back-end graph: tokens, positions, centres, index lists -- 1.7 MB in 16 runtime copies of ~4.7 us each, one launch here).
// Every job is a byte range; 16-byte pieces when source, destination and length allow it.  grid = (chunks of 64 KB, jobs).
constexpr int kMaxCopies = 64;
struct CopyJobs { const char *src[kMaxCopies]; char *dst[kMaxCopies]; long long bytes[kMaxCopies]; };
__global__ __launch_bounds__(256) void copy_batched_kernel(CopyJobs t) {
    const int j = blockIdx.y;
    const long long n = t.bytes[j], c0 = (long long)blockIdx.x * 65536;
    if (c0 >= n) return;
    const char *src = t.src[j] + c0;
    char *dst = t.dst[j] + c0;
    const long long len = n - c0 < 65536 ? n - c0 : 65536;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const long long v = len >> 4;
        for (long long i = threadIdx.x; i < v; i += 256) reinterpret_cast<f32x4 *>(dst)[i] = reinterpret_cast<const f32x4 *>(src)[i];
        for (long long i = (v << 4) + threadIdx.x; i < len; i += 256) dst[i] = src[i];
    } else {
        for (long long i = threadIdx.x; i < len; i += 256) dst[i] = src[i];
    }
}

}  // namespace

extern "C" int upp_wcolsum_partials(const float *src, long long ld, const float *wts, long long ldw, int W, int n, int len, int chunks,
                                    float *dst, void *stream) {
    if (!src || !wts || !dst || n < 1 || len < 1 || chunks < 1 || ld < len || W < 1 || ldw < W) return UPP_E_BADARG;
    if (chunks > 65535 || W > 4 || len % 4 || ld % 4 || (reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) return UPP_E_RANGE;
    const dim3 grid((len + 255) / 256, chunks);
    hipStream_t st = (hipStream_t)stream;
    switch (W) {
        case 1: hipLaunchKernelGGL(wcolsum_partials_kernel<1>, grid, dim3(256), 0, st, src, ld, wts, ldw, n, len, dst); break;
        case 2: hipLaunchKernelGGL(wcolsum_partials_kernel<2>, grid, dim3(256), 0, st, src, ld, wts, ldw, n, len, dst); break;
        case 3: hipLaunchKernelGGL(wcolsum_partials_kernel<3>, grid, dim3(256), 0, st, src, ld, wts, ldw, n, len, dst); break;
        default: hipLaunchKernelGGL(wcolsum_partials_kernel<4>, grid, dim3(256), 0, st, src, ld, wts, ldw, n, len, dst); break;
    }
    return upp_launch_status();
}

extern "C" int upp_batched_sum(const float *const *src, float *const *dst, const int *n, const int *len, const int *ld,
                               const int *accumulate, const int *dst_width, const int *dst_pitch, int jobs, void *stream) {
    if (jobs < 0 || (jobs > 0 && (!src || !dst || !n || !len || !ld || !accumulate))) return UPP_E_BADARG;
    if ((dst_width == nullptr) != (dst_pitch == nullptr)) return UPP_E_BADARG;
    for (int j = 0; j < jobs; ++j) {
        if (!src[j] || !dst[j] || n[j] < 1 || len[j] < 1 || ld[j] < len[j]) return UPP_E_BADARG;
        if (dst_width && dst_width[j] != 0 && (dst_width[j] < 1 || len[j] % dst_width[j] != 0 || dst_pitch[j] < dst_width[j])) return UPP_E_BADARG;
    }
    // group the jobs by destination (first-seen order; submission order inside a group), then launch in batches
    int order[4096], gstart[4096 + 1];
    if (jobs > 4096) return UPP_E_RANGE;
    int ngroups = 0, filled = 0;
    bool taken[4096];
    for (int j = 0; j < jobs; ++j) taken[j] = false;
    for (int j = 0; j < jobs; ++j) {
        if (taken[j]) continue;
        gstart[ngroups++] = filled;
        for (int k = j; k < jobs; ++k)
            if (!taken[k] && dst[k] == dst[j]) {
                if (len[k] != len[j] || accumulate[k] != accumulate[j]) return UPP_E_BADARG;   // one destination, one shape
                if (dst_width && (dst_width[k] != dst_width[j] || dst_pitch[k] != dst_pitch[j])) return UPP_E_BADARG;
                taken[k] = true; order[filled++] = k;
            }
    }
    gstart[ngroups] = filled;
    // two passes over the groups: the short ones (<= kTallRows rows per job, 256 columns per workgroup), then the tall ones
    bool tall[4096];
    for (int g = 0; g < ngroups; ++g) {
        tall[g] = false;
        for (int q = gstart[g]; q < gstart[g + 1]; ++q) tall[g] = tall[g] || n[order[q]] > kTallRows;
    }
    // ... and the short ones with long, 16-byte aligned rows (pass 2: 1,024 columns per workgroup, 16 bytes per lane)
    int cls[4096];
    for (int g = 0; g < ngroups; ++g) {
        const int head = order[gstart[g]];
        bool wide = !tall[g] && len[head] >= 4096 && len[head] % 4 == 0 && (reinterpret_cast<uintptr_t>(dst[head]) & 15) == 0;
        if (dst_width && dst_width[head] > 0) wide = wide && dst_width[head] % 4 == 0 && dst_pitch[head] % 4 == 0;
        for (int q = gstart[g]; q < gstart[g + 1] && wide; ++q) {
            const int k = order[q];
            wide = ld[k] % 4 == 0 && (reinterpret_cast<uintptr_t>(src[k]) & 15) == 0;
        }
        cls[g] = tall[g] ? 1 : (wide ? 2 : 0);
    }
    for (int pass = 0; pass < 3; ++pass) {
        const int cols = pass == 1 ? 64 : (pass == 2 ? 1024 : 256);
        int g0 = 0;
        while (g0 < ngroups) {
            SumJobs t;
            int g = 0, nj = 0, wg = 0, used = 0;
            while (g0 + used < ngroups && g < kMaxSumJobs) {
                const int gi = g0 + used;
                if (cls[gi] != pass) { ++used; continue; }
                if (nj + (gstart[gi + 1] - gstart[gi]) > kMaxSumJobs) break;
                const int lo = gstart[gi], hi = gstart[gi + 1], head = order[lo];
                t.first[g] = nj;
                for (int q = lo; q < hi; ++q) { const int k = order[q]; t.src[nj] = src[k]; t.n[nj] = n[k]; t.ld[nj] = ld[k]; ++nj; }
                t.dst[g] = dst[head]; t.len[g] = len[head]; t.acc[g] = accumulate[head]; t.wg0[g] = wg;
                t.dw[g] = dst_width ? dst_width[head] : 0; t.dpitch[g] = dst_width ? dst_pitch[head] : 0;
                wg += (len[head] + cols - 1) / cols;
                ++g; ++used;
            }
            if (g == 0) {
                if (g0 + used >= ngroups) break;             // nothing of this pass left
                return UPP_E_RANGE;                          // a single destination with more than 64 partial matrices
            }
            t.first[g] = nj; t.wg0[g] = wg; t.groups = g;
            if (pass == 1) hipLaunchKernelGGL(batched_sum_tall_kernel, dim3(wg), dim3(256), 0, (hipStream_t)stream, t);
            else if (pass == 2) hipLaunchKernelGGL(batched_sum_kernel<4>, dim3(wg), dim3(256), 0, (hipStream_t)stream, t);
            else hipLaunchKernelGGL(batched_sum_kernel<1>, dim3(wg), dim3(256), 0, (hipStream_t)stream, t);
            g0 += used;
        }
    }
    return upp_launch_status();
}

extern "C" int upp_colsum_partials(const float *src, long long ld, int n, int len, int chunks, float *dst, void *stream) {
    if (!src || !dst || n < 1 || len < 1 || chunks < 1 || ld < len) return UPP_E_BADARG;
    if (chunks > 65535) return UPP_E_RANGE;
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((len + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, src, ld, n, len, dst);
    return upp_launch_status();
}

extern "C" long long upp_adamw_scratch_floats(void) { return kRedBlocks; }

extern "C" int upp_adamw_flat(float *p, float *g, float *m, float *v, long long n, long long split, float *state, float *scratch,
                              float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm, void *stream) {
    if (!p || !g || !m || !v || !state || !scratch || n < 1 || split < 0 || split > n) return UPP_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    long long blocks = (n + 255) / 256;
    const int rb = (int)(blocks < kRedBlocks ? blocks : kRedBlocks);
    hipLaunchKernelGGL(sumsq_kernel, dim3(rb), dim3(256), 0, st, g, n, scratch);
    hipLaunchKernelGGL(adamw_prepare_kernel, dim3(1), dim3(256), 0, st, scratch, rb, state, max_norm, beta1, beta2);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adamw_update_kernel, dim3((int)blocks), dim3(256), 0, st, p, g, m, v, n, split, state, lr, beta1, beta2, eps,
                       weight_decay);
    return upp_launch_status();
}

extern "C" int upp_copy_batched(const void *const *src, void *const *dst, const long long *bytes, int count, void *stream) {
    if (count < 0 || (count > 0 && (!src || !dst || !bytes))) return UPP_E_BADARG;
    for (int j0 = 0; j0 < count; j0 += kMaxCopies) {
        CopyJobs t;
        const int n = count - j0 < kMaxCopies ? count - j0 : kMaxCopies;
        long long most = 0;
        for (int j = 0; j < n; ++j) {
            if (!src[j0 + j] || !dst[j0 + j] || bytes[j0 + j] < 0) return UPP_E_BADARG;
            t.src[j] = static_cast<const char *>(src[j0 + j]); t.dst[j] = static_cast<char *>(dst[j0 + j]); t.bytes[j] = bytes[j0 + j];
            most = bytes[j0 + j] > most ? bytes[j0 + j] : most;
        }
        if (most == 0) continue;
        if ((most + 65535) / 65536 > 65535) return UPP_E_RANGE;
        hipLaunchKernelGGL(copy_batched_kernel, dim3((unsigned)((most + 65535) / 65536), n), dim3(256), 0, (hipStream_t)stream, t);
    }
    return upp_launch_status();
}
