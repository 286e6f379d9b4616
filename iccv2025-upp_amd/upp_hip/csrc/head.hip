// head.hip -- the classification tail of Point_MAE_unify.forward / get_loss_acc (reference
// models/Point_MAE_unify.py:650-655, :499-503) as three launches instead of ~30 tiny torch kernels:
//   upp_cls_pool_fwd / bwd : LayerNorm of the block output, then feature = [cls row | max over the other rows]
//                            (self.norm + torch.cat([x[:, 0], x[:, 1:].max(1)[0]], -1)); backward routes the feature
//                            gradient back through the arg-max rows and the LayerNorm.
//   upp_ce_acc             : mean cross-entropy, top-1 accuracy (x100) and d loss / d logits in one pass.
// One workgroup per sample (pooling) / one wave per sample (loss): B is the batch size (32), everything is latency.
#include "common.h"

namespace {

constexpr int kMaxE = 8;     // columns per lane -> D <= 512
constexpr int kPW = 8;       // waves per workgroup in the pooling kernels

// feat[b][0:D] = LN(x[b][0]);  feat[b][D:2D] = max_{t>=1} LN(x[b][t]) (first maximum), amax[b][c] = its row
__global__ __launch_bounds__(64 * kPW) void cls_pool_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, float eps, float *__restrict__ feat,
                                                                int32_t *__restrict__ amax, float *__restrict__ mean,
                                                                float *__restrict__ rstd, int L, int D) {
    __shared__ float smx[kPW][64 * kMaxE];
    __shared__ int sam[kPW][64 * kMaxE];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    int cc[kMaxE];
    float gv[kMaxE], bv[kMaxE], mx[kMaxE];
    int am[kMaxE];
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) {
        cc[e] = min(lane + 64 * e, D - 1);
        gv[e] = gamma[cc[e]]; bv[e] = beta[cc[e]];
        mx[e] = -__builtin_inff(); am[e] = 1;
    }
    for (int t0 = wave; t0 < L; t0 += kPW * 2) {                 // two rows per iteration: their reductions interleave
        float v[2][kMaxE], s[2], q[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int t = min(t0 + r * kPW, L - 1);
#pragma unroll
            for (int e = 0; e < kMaxE; ++e) v[r][e] = x[((size_t)b * L + t) * D + cc[e]];
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float a = 0.0f;
#pragma unroll
            for (int e = 0; e < kMaxE; ++e) a += lane + 64 * e < D ? v[r][e] : 0.0f;
            s[r] = a;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) s[r] = wave_sum_f32(s[r]) / (float)D;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float a = 0.0f;
#pragma unroll
            for (int e = 0; e < kMaxE; ++e) { const float dv = lane + 64 * e < D ? v[r][e] - s[r] : 0.0f; a = __builtin_fmaf(dv, dv, a); }
            q[r] = a;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) q[r] = 1.0f / sqrtf(wave_sum_f32(q[r]) / (float)D + eps);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int t = t0 + r * kPW;
            if (t >= L) continue;
            if (lane == 0) { mean[(size_t)b * L + t] = s[r]; rstd[(size_t)b * L + t] = q[r]; }
#pragma unroll
            for (int e = 0; e < kMaxE; ++e) {
                const float h = __builtin_fmaf((v[r][e] - s[r]) * q[r], gv[e], bv[e]);
                if (t == 0) { if (lane + 64 * e < D) feat[(size_t)b * 2 * D + lane + 64 * e] = h; }
                else if (h > mx[e]) { mx[e] = h; am[e] = t; }      // rows visited in increasing t: strict '>' keeps the first maximum
            }
        }
    }
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) { smx[wave][lane + 64 * e] = mx[e]; sam[wave][lane + 64 * e] = am[e]; }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 64 * kPW) {
        float best = smx[0][c];
        int bi = sam[0][c];
#pragma unroll
        for (int w = 1; w < kPW; ++w) {
            const float v = smx[w][c];
            const int i = sam[w][c];
            if (v > best || (v == best && i < bi)) { best = v; bi = i; }
        }
        feat[(size_t)b * 2 * D + D + c] = best;
        amax[(size_t)b * D + c] = bi;
    }
}

// g_x[b][t] = LayerNormBackward(row t) of  g_h[t][c] = [t == 0] g_feat[b][c] + [amax[b][c] == t] g_feat[b][D + c]
__global__ __launch_bounds__(64 * kPW) void cls_pool_bwd_kernel(const float *__restrict__ g_feat, const float *__restrict__ x,
                                                                const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                const float *__restrict__ gamma, const int32_t *__restrict__ amax,
                                                                float *__restrict__ g_x, int L, int D) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    int cc[kMaxE], am[kMaxE];
    float gv[kMaxE], g0[kMaxE], g1[kMaxE];
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) {
        cc[e] = min(lane + 64 * e, D - 1);
        gv[e] = gamma[cc[e]];
        g0[e] = g_feat[(size_t)b * 2 * D + cc[e]];
        g1[e] = g_feat[(size_t)b * 2 * D + D + cc[e]];
        am[e] = amax[(size_t)b * D + cc[e]];
    }
    for (int t0 = wave; t0 < L; t0 += kPW * 2) {
        float xv[2][kMaxE], mu[2], rs[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int t = min(t0 + r * kPW, L - 1);
#pragma unroll
            for (int e = 0; e < kMaxE; ++e) xv[r][e] = x[((size_t)b * L + t) * D + cc[e]];
            mu[r] = mean[(size_t)b * L + t]; rs[r] = rstd[(size_t)b * L + t];
        }
        float dy[2][kMaxE], xh[2][kMaxE], s1[2], s2[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int t = t0 + r * kPW;
            float a1 = 0.0f, a2 = 0.0f;
#pragma unroll
            for (int e = 0; e < kMaxE; ++e) {
                const bool ok = lane + 64 * e < D;
                const float gh = (t == 0 ? g0[e] : 0.0f) + (am[e] == t ? g1[e] : 0.0f);
                dy[r][e] = ok ? gh * gv[e] : 0.0f;
                xh[r][e] = ok ? (xv[r][e] - mu[r]) * rs[r] : 0.0f;
                a1 += dy[r][e];
                a2 = __builtin_fmaf(dy[r][e], xh[r][e], a2);
            }
            s1[r] = a1; s2[r] = a2;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) { s1[r] = wave_sum_f32(s1[r]) / (float)D; s2[r] = wave_sum_f32(s2[r]) / (float)D; }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int t = t0 + r * kPW;
            if (t >= L) continue;
#pragma unroll
            for (int e = 0; e < kMaxE; ++e)
                if (lane + 64 * e < D) g_x[((size_t)b * L + t) * D + lane + 64 * e] = rs[r] * (dy[r][e] - s1[r] - xh[r][e] * s2[r]);
        }
    }
}

// One wave per sample: log-softmax over C <= 512 classes; out[0] = mean loss, out[1] = accuracy * 100 (top-1, first maximum);
// dlogits = (softmax - onehot) / B.  A single workgroup sums the B per-sample terms in sample order (deterministic).
__global__ __launch_bounds__(1024) void ce_acc_kernel(const float *__restrict__ logits, const int64_t *__restrict__ labels,
                                                      float *__restrict__ out, float *__restrict__ dlogits, int B, int C) {
    __shared__ float sl[1024], sa[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float lsum = 0.0f, asum = 0.0f;                                  // lane 0 of each wave accumulates its samples
    for (int b = wave; b < B; b += nw) {
        float v[kMaxE];
        float mx = -__builtin_inff();
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) {
            const int c = lane + 64 * e;
            v[e] = c < C ? logits[(size_t)b * C + c] : -__builtin_inff();
            mx = fmaxf(mx, v[e]);
        }
        mx = wave_max_f32(mx);
        float se = 0.0f;
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) se += lane + 64 * e < C ? expf(v[e] - mx) : 0.0f;
        se = wave_sum_f32(se);
        const float lse = mx + logf(se);
        const int y = (int)labels[b];
        // first index holding the maximum (torch.argmax)
        int first = C;
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) if (lane + 64 * e < C && v[e] == mx) first = min(first, lane + 64 * e);
        first = (int)wave_min_u32((uint32_t)first);
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) {
            const int c = lane + 64 * e;
            if (c < C) dlogits[(size_t)b * C + c] = (expf(v[e] - lse) - (c == y ? 1.0f : 0.0f)) / (float)B;
        }
        const float vy = (y >= 0 && y < C) ? logits[(size_t)b * C + y] : 0.0f;
        lsum += lse - vy;
        asum += first == y ? 1.0f : 0.0f;
    }
    sl[threadIdx.x] = lsum; sa[threadIdx.x] = asum;
    __syncthreads();
    if (threadIdx.x == 0) {
        // per-sample terms in sample order: sample b was handled by wave b % nw in its (b / nw)-th iteration; each wave kept a
        // running sum, so add the waves' sums in wave order
        float l = 0.0f, a = 0.0f;
        for (int w = 0; w < nw; ++w) { l += sl[w * 64]; a += sa[w * 64]; }
        out[0] = l / (float)B;
        out[1] = a / (float)B * 100.0f;
    }
}

}  // namespace

extern "C" int upp_cls_pool_fwd(const float *x, const float *gamma, const float *beta, float eps, float *feat, int32_t *amax, float *mean,
                                float *rstd, int B, int L, int D, void *stream) {
    if (!x || !gamma || !beta || !feat || !amax || !mean || !rstd || B < 1 || L < 2 || D < 1) return UPP_E_BADARG;
    if (D > 64 * kMaxE) return UPP_E_RANGE;
    hipLaunchKernelGGL(cls_pool_fwd_kernel, dim3(B), dim3(64 * kPW), 0, (hipStream_t)stream, x, gamma, beta, eps, feat, amax, mean, rstd, L, D);
    return upp_launch_status();
}

extern "C" int upp_cls_pool_bwd(const float *g_feat, const float *x, const float *mean, const float *rstd, const float *gamma,
                                const int32_t *amax, float *g_x, int B, int L, int D, void *stream) {
    if (!g_feat || !x || !mean || !rstd || !gamma || !amax || !g_x || B < 1 || L < 2 || D < 1) return UPP_E_BADARG;
    if (D > 64 * kMaxE) return UPP_E_RANGE;
    hipLaunchKernelGGL(cls_pool_bwd_kernel, dim3(B), dim3(64 * kPW), 0, (hipStream_t)stream, g_feat, x, mean, rstd, gamma, amax, g_x, L, D);
    return upp_launch_status();
}

extern "C" int upp_ce_acc(const float *logits, const int64_t *labels, float *out2, float *dlogits, int B, int C, void *stream) {
    if (!logits || !labels || !out2 || !dlogits || B < 1 || C < 1) return UPP_E_BADARG;
    if (C > 64 * kMaxE) return UPP_E_RANGE;
    int waves = B < 16 ? B : 16;
    hipLaunchKernelGGL(ce_acc_kernel, dim3(1), dim3(64 * waves), 0, (hipStream_t)stream, logits, labels, out2, dlogits, B, C);
    return upp_launch_status();
}
