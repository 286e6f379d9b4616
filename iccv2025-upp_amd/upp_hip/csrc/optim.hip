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

constexpr int kRedBlocks = 256;

__global__ __launch_bounds__(256) void sumsq_kernel(const float *__restrict__ g, long long n, float *__restrict__ part) {
    float s = 0.0f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s = __builtin_fmaf(g[i], g[i], s);
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
// src_j over its first dimension into dst_j (len_j), rows added in ascending order (deterministic); dst is overwritten
// or accumulated into.  The parameter-gradient partials of a
// backward pass (adapter weight partials per workgroup, LayerNorm gamma/beta partials per chunk, per-sample prompt
// gradients) are summed here, after the pass, instead of by ~3 tiny reduce launches per transformer block.
constexpr int kMaxSumJobs = 64;
struct SumJobs {
    const float *src[kMaxSumJobs];
    float *dst[kMaxSumJobs];
    int n[kMaxSumJobs], len[kMaxSumJobs], ld[kMaxSumJobs], acc[kMaxSumJobs];
    int wg0[kMaxSumJobs + 1];                  // first workgroup of each job (256 columns per workgroup)
    int jobs;
};
__global__ __launch_bounds__(256) void batched_sum_kernel(SumJobs t) {
    int j = 0;
    while (j + 1 < t.jobs && (int)blockIdx.x >= t.wg0[j + 1]) ++j;             // wave-uniform scan of <= 64 entries
    const int c = ((int)blockIdx.x - t.wg0[j]) * 256 + threadIdx.x;
    const int len = t.len[j], n = t.n[j], ld = t.ld[j];
    if (c >= len) return;
    const float *src = t.src[j] + c;
    float acc = 0.0f;
    for (int i0 = 0; i0 < n; i0 += 16) {                                       // 16 independent loads in flight
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = src[(size_t)min(i0 + q, n - 1) * ld];
#pragma unroll
        for (int q = 0; q < 16; ++q) if (i0 + q < n) acc += v[q];
    }
    t.dst[j][c] = t.acc[j] ? t.dst[j][c] + acc : acc;
}

}  // namespace

extern "C" int upp_batched_sum(const float *const *src, float *const *dst, const int *n, const int *len, const int *ld,
                               const int *accumulate, int jobs, void *stream) {
    if (jobs < 0 || (jobs > 0 && (!src || !dst || !n || !len || !ld || !accumulate))) return UPP_E_BADARG;
    for (int j0 = 0; j0 < jobs; j0 += kMaxSumJobs) {
        SumJobs t;
        t.jobs = jobs - j0 < kMaxSumJobs ? jobs - j0 : kMaxSumJobs;
        int wg = 0;
        for (int j = 0; j < t.jobs; ++j) {
            if (!src[j0 + j] || !dst[j0 + j] || n[j0 + j] < 1 || len[j0 + j] < 1 || ld[j0 + j] < len[j0 + j]) return UPP_E_BADARG;
            t.src[j] = src[j0 + j]; t.dst[j] = dst[j0 + j]; t.n[j] = n[j0 + j]; t.len[j] = len[j0 + j];
            t.ld[j] = ld[j0 + j]; t.acc[j] = accumulate[j0 + j];
            t.wg0[j] = wg;
            wg += (len[j0 + j] + 255) / 256;
        }
        t.wg0[t.jobs] = wg;
        hipLaunchKernelGGL(batched_sum_kernel, dim3(wg), dim3(256), 0, (hipStream_t)stream, t);
    }
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
