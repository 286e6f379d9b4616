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

// ---- per-point log-softmax + NLL of the segmentation head (round 6; reference models/Point_MAE_unify_segment.py:433 `F.log_softmax`,
// :20-25 get_loss = F.nll_loss): C <= 64 classes over tens of thousands of point rows.  One wave per row, kLsRows rows per wave; the row
// arrives as a lane-per-class load from a matrix whose rows may be wider than C (the 50-class layer's output lives in a 52-column
// matrix: the GEMM kernels want N % 4 == 0) and the class bias is added here (the GEMM then runs bias-free on the un-padded weight's
// plane image).  torch formulation replaced: softmax_warp_forward / _backward, gather, mean, neg, div, zero-fill + scatter_add, and the
// zero-fill + copy of the [:, :C] slice's backward.
constexpr int kLsRows = 8;
__global__ __launch_bounds__(256) void logsoftmax_rows_fwd_kernel(const float *__restrict__ y, long long ld_y, const float *__restrict__ bias,
                                                                  long long R, int C, float *__restrict__ logp) {
    const int lane = threadIdx.x & 63;
    const long long r0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * kLsRows;
    const float bv = (bias && lane < C) ? bias[lane] : 0.0f;
    float v[kLsRows];
#pragma unroll
    for (int i = 0; i < kLsRows; ++i) v[i] = (lane < C && r0 + i < R) ? y[(r0 + i) * ld_y + lane] : 0.0f;
#pragma unroll
    for (int i = 0; i < kLsRows; ++i) {
        if (r0 + i >= R) break;
        const float x = lane < C ? v[i] + bv : -__builtin_inff();
        const float mx = wave_max_f32(x);
        const float se = wave_sum_f32(lane < C ? expf(x - mx) : 0.0f);
        if (lane < C) logp[(r0 + i) * C + lane] = (x - mx) - logf(se);
    }
}

// g_y[r][c] = g[r][c] - exp(logp[r][c]) * sum_c g[r][c]  (c < C);  0 for C <= c < Cpad (the pad columns of the producing GEMM's output)
__global__ __launch_bounds__(256) void logsoftmax_rows_bwd_kernel(const float *__restrict__ g, const float *__restrict__ logp, long long R, int C,
                                                                  float *__restrict__ g_y, long long ld_gy, int Cpad) {
    const int lane = threadIdx.x & 63;
    const long long r0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * kLsRows;
    float gv[kLsRows], lv[kLsRows];
#pragma unroll
    for (int i = 0; i < kLsRows; ++i) {
        const bool ok = lane < C && r0 + i < R;
        gv[i] = ok ? g[(r0 + i) * C + lane] : 0.0f;
        lv[i] = ok ? logp[(r0 + i) * C + lane] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < kLsRows; ++i) {
        if (r0 + i >= R) break;
        const float sg = wave_sum_f32(gv[i]);
        if (lane < Cpad) g_y[(r0 + i) * ld_gy + lane] = lane < C ? gv[i] - expf(lv[i]) * sg : 0.0f;
    }
}

// part[wg] = sum over the workgroup's rows of logp[r][target[r]] (rows in order inside a wave, waves in order); the second launch adds the
// partials in workgroup order and scales by -1 / R: out[0] = F.nll_loss(logp, target) (mean), deterministic.
constexpr int kNllRows = 64;          // rows per wave
__global__ __launch_bounds__(256) void nll_partial_kernel(const float *__restrict__ logp, const int64_t *__restrict__ target, long long R, int C,
                                                          float *__restrict__ part) {
    __shared__ float sw[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long r = ((long long)blockIdx.x * 4 + wave) * kNllRows + lane;
    float v = 0.0f;
    if (r < R) { const long long t = target[r]; v = (t >= 0 && t < C) ? logp[r * C + t] : 0.0f; }
    v = wave_sum_f32(v);
    if (lane == 0) sw[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((sw[0] + sw[1]) + sw[2]) + sw[3];
}
__global__ __launch_bounds__(1024) void nll_final_kernel(const float *__restrict__ part, int n, float scale, float *__restrict__ out) {
    __shared__ float sh[1024];
    float acc = 0.0f;
    const int per = (n + 1023) / 1024, i0 = (int)threadIdx.x * per;           // thread t owns a contiguous run: order = index order
    for (int i = i0; i < min(n, i0 + per); ++i) acc += part[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.0f;
        for (int t = 0; t < 1024; ++t) tot += sh[t];
        out[0] = tot * scale;
    }
}
// g_logp[r][c] = c == target[r] ? -g_loss / R : 0  -- every element written once (no zero-fill + scatter)
__global__ __launch_bounds__(256) void nll_bwd_kernel(const float *__restrict__ g_loss, const int64_t *__restrict__ target, long long R, int C,
                                                      float *__restrict__ g_logp) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * C) return;
    const long long r = i / C;
    const int c = (int)(i - r * C);
    g_logp[i] = (long long)c == target[r] ? -g_loss[0] / (float)R : 0.0f;
}

// ---- noise-vector supervision of the pre-task recipe (round 6; reference models/Point_MAE_pretask_dev.py:685-692:
//   positive = mean(norm(pred_noise - noise_vector, dim=-1, keepdim=True) ** 2),  negative = mean(norm(pred_pure, dim=-1, keepdim=True) ** 2),
//   score = norm(pred, dim=-1)) -- ~30 element-wise torch launches forward and backward (norm, pow, mean, the zero guards of norm's backward).
// pred (B,P,3): the first pn points of a cloud are shape points (target 0), the rest noise points (target nv (B, P - pn, 3)).
// Forward: per-workgroup partial sums in a fixed order -> nll_final_kernel adds them; score (B,P) = |pred| beside it.
__global__ __launch_bounds__(256) void noise_loss_partial_kernel(const float *__restrict__ pred, const float *__restrict__ nv, int B, int P, int pn,
                                                                 float w_pure, float w_noise, float *__restrict__ part, float *__restrict__ score) {
    __shared__ float sw[4];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x, total = (long long)B * P;
    float v = 0.0f;
    if (i < total) {
        const int b = (int)(i / P), t = (int)(i - (long long)b * P);
        const float x = pred[i * 3], y = pred[i * 3 + 1], z = pred[i * 3 + 2];
        if (score) score[i] = sqrtf(sumsq3(x, y, z));
        if (t < pn) v = w_pure * sumsq3(x, y, z);
        else {
            const float *q = nv + ((long long)b * (P - pn) + (t - pn)) * 3;
            v = w_noise * sumsq3(x - q[0], y - q[1], z - q[2]);
        }
    }
    v = wave_sum_f32(v);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((sw[0] + sw[1]) + sw[2]) + sw[3];
}
__global__ __launch_bounds__(256) void noise_loss_bwd_kernel(const float *__restrict__ g_loss, const float *__restrict__ pred, const float *__restrict__ nv,
                                                             int B, int P, int pn, float w_pure, float w_noise, float *__restrict__ g_pred) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x, total = (long long)B * P;
    if (i >= total) return;
    const int b = (int)(i / P), t = (int)(i - (long long)b * P);
    const float g = g_loss[0];
    float dx = pred[i * 3], dy = pred[i * 3 + 1], dz = pred[i * 3 + 2], w = 2.0f * w_pure * g;
    if (t >= pn) {
        const float *q = nv + ((long long)b * (P - pn) + (t - pn)) * 3;
        dx -= q[0]; dy -= q[1]; dz -= q[2]; w = 2.0f * w_noise * g;
    }
    g_pred[i * 3] = w * dx; g_pred[i * 3 + 1] = w * dy; g_pred[i * 3 + 2] = w * dz;
}

// BatchNorm1d (batch statistics over the R rows) + ReLU + Dropout of a small (R, C) matrix, one workgroup per 64 columns:
// lane = column, the 4 waves stride the rows; two passes over the column (mean, then M2 about it) from registers/L2.
// Forward saves mean / rstd; the dropout mask is (u >= p) from caller-supplied uniforms, kept values scaled by 1/(1-p).
constexpr int kBW = 4;
__global__ __launch_bounds__(64 * kBW) void bn_relu_drop_fwd_kernel(const float *__restrict__ z, const float *__restrict__ gamma,
                                                                    const float *__restrict__ beta, float *__restrict__ running_mean,
                                                                    float *__restrict__ running_var, float momentum, float eps,
                                                                    int training, const float *__restrict__ u, float p,
                                                                    float *__restrict__ a, float *__restrict__ mean_out,
                                                                    float *__restrict__ rstd_out, int R, int C) {
    __shared__ float red[kBW][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int cc = min(c, C - 1);
    float mean, rstd;
    if (training) {
        float s = 0.0f;
        for (int r = wave; r < R; r += kBW) s += z[(size_t)r * C + cc];
        red[wave][lane] = s;
        __syncthreads();
        mean = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) / (float)R;
        __syncthreads();
        float q = 0.0f;
        for (int r = wave; r < R; r += kBW) { const float d = z[(size_t)r * C + cc] - mean; q = __builtin_fmaf(d, d, q); }
        red[wave][lane] = q;
        __syncthreads();
        const float m2 = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        rstd = 1.0f / sqrtf(m2 / (float)R + eps);
        if (wave == 0 && c < C && running_mean) {
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (m2 / (float)(R > 1 ? R - 1 : 1));
        }
    } else {
        mean = running_mean[cc];
        rstd = 1.0f / sqrtf(running_var[cc] + eps);
    }
    if (wave == 0 && c < C) { mean_out[c] = mean; rstd_out[c] = rstd; }
    const float g = gamma[cc], bt = beta[cc], keep = 1.0f / (1.0f - p);
    if (c < C)
        for (int r = wave; r < R; r += kBW) {
            float v = fmaxf(((z[(size_t)r * C + c] - mean) * rstd) * g + bt, 0.0f);
            if (u) v = u[(size_t)r * C + c] >= p ? v * keep : 0.0f;
            a[(size_t)r * C + c] = v;
        }
}

// backward of the above (training statistics): g_z, g_gamma, g_beta from g_a; ReLU / dropout masks are recomputed.
__global__ __launch_bounds__(64 * kBW) void bn_relu_drop_bwd_kernel(const float *__restrict__ g_a, const float *__restrict__ z,
                                                                    const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                    const float *__restrict__ mean_in, const float *__restrict__ rstd_in,
                                                                    int training, const float *__restrict__ u, float p,
                                                                    float *__restrict__ g_z, float *__restrict__ g_gamma,
                                                                    float *__restrict__ g_beta, int R, int C) {
    __shared__ float r1[kBW][64], r2[kBW][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int cc = min(c, C - 1);
    const float mean = mean_in[cc], rstd = rstd_in[cc], g = gamma[cc], bt = beta[cc], keep = 1.0f / (1.0f - p);
    auto dy_of = [&](int r, float &xh) {          // gradient w.r.t. the BatchNorm output of element (r, c)
        xh = (z[(size_t)r * C + cc] - mean) * rstd;
        float d = g_a[(size_t)r * C + cc];
        if (u) d = u[(size_t)r * C + cc] >= p ? d * keep : 0.0f;
        return (xh * g + bt) > 0.0f ? d : 0.0f;
    };
    float sb = 0.0f, sg = 0.0f;
    for (int r = wave; r < R; r += kBW) { float xh; const float d = dy_of(r, xh); sb += d; sg = __builtin_fmaf(d, xh, sg); }
    r1[wave][lane] = sb; r2[wave][lane] = sg;
    __syncthreads();
    sb = (r1[0][lane] + r1[1][lane]) + (r1[2][lane] + r1[3][lane]);
    sg = (r2[0][lane] + r2[1][lane]) + (r2[2][lane] + r2[3][lane]);
    if (wave == 0 && c < C) { g_beta[c] = sb; g_gamma[c] = sg; }
    if (c < C)
        for (int r = wave; r < R; r += kBW) {
            float xh;
            const float d = dy_of(r, xh);
            g_z[(size_t)r * C + c] = training ? g * rstd * (d - sb / (float)R - xh * sg / (float)R) : g * rstd * d;
        }
}

// ---- the tail of the denoising ("rectify") prompter -----------------------------------------------------------------------------
// reference models/Point_MAE_pretask_dev.py:491-512 (RectifyPrompter.score_head: Linear(32, 64) -> ReLU -> Dropout(0.2) -> Linear(64, 3))
// and models/Point_MAE_unify.py:553-559: score = ||pred||_2, order = argsort(score, descending), pts += 0.2 pred, keep the last
// int(0.95 point_num) entries of the order.  Two launches instead of two library GEMMs, ReLU, dropout, scale, norm, a radix sort,
// add and gather:
//   rectify_score_kernel : one thread per point -- the 32 -> 64 -> 3 head from the LDS copies of its weights, the nudged point and
//                          its score.  hidden = relu(W0 f + b0) * keep/(1-p) is an ascending-k fmaf chain per output (as
//                          upp_linear_smallk_f32), pred = W1 hidden + b1 likewise.
//   rectify_select_kernel: one workgroup per (cloud, 256 points) -- the cloud's scores in the LDS, rank of a point = how many
//                          points precede it in the stable descending order (greater score, or equal score and lower index: what
//                          torch's stable radix sort returns); the points ranked >= N - keep are written at rank - (N - keep).
constexpr int kRsIn = 32, kRsHid = 64;

__global__ __launch_bounds__(256) void rectify_score_kernel(const float *__restrict__ feat, const float *__restrict__ W0, const float *__restrict__ b0,
                                                            const float *__restrict__ W1, const float *__restrict__ b1, const float *__restrict__ u,
                                                            float p, float factor, const float *__restrict__ pts, float nudge, int rows,
                                                            float *__restrict__ pred, float *__restrict__ moved, float *__restrict__ score) {
    __shared__ float w0[kRsHid][kRsIn + 1], w1[3][kRsHid], bb0[kRsHid];
    for (int i = threadIdx.x; i < kRsHid * kRsIn; i += 256) w0[i / kRsIn][i % kRsIn] = W0[i];
    for (int i = threadIdx.x; i < 3 * kRsHid; i += 256) w1[i / kRsHid][i % kRsHid] = W1[i];
    if (threadIdx.x < kRsHid) bb0[threadIdx.x] = b0[threadIdx.x];
    __syncthreads();
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float f[kRsIn];
    const float4 *fp = reinterpret_cast<const float4 *>(feat + (size_t)row * kRsIn);
#pragma unroll
    for (int q = 0; q < kRsIn / 4; ++q) { const float4 v = fp[q]; f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w; }
    const float keep_scale = 1.0f / (1.0f - p);
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
    for (int hh = 0; hh < kRsHid; ++hh) {
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < kRsIn; ++k) a = __builtin_fmaf(f[k], w0[hh][k], a);
        a = fmaxf(a + bb0[hh], 0.0f);
        if (u) a = u[(size_t)row * kRsHid + hh] >= p ? a * keep_scale : 0.0f;
        o0 = __builtin_fmaf(a, w1[0][hh], o0); o1 = __builtin_fmaf(a, w1[1][hh], o1); o2 = __builtin_fmaf(a, w1[2][hh], o2);
    }
    o0 = (o0 + b1[0]) * factor; o1 = (o1 + b1[1]) * factor; o2 = (o2 + b1[2]) * factor;
    if (pred) { pred[(size_t)row * 3] = o0; pred[(size_t)row * 3 + 1] = o1; pred[(size_t)row * 3 + 2] = o2; }
    moved[(size_t)row * 3] = pts[(size_t)row * 3] + o0 * nudge;
    moved[(size_t)row * 3 + 1] = pts[(size_t)row * 3 + 1] + o1 * nudge;
    moved[(size_t)row * 3 + 2] = pts[(size_t)row * 3 + 2] + o2 * nudge;
    score[row] = sqrtf(o0 * o0 + o1 * o1 + o2 * o2);
}

__global__ __launch_bounds__(256) void rectify_select_kernel(const float *__restrict__ score, const float *__restrict__ moved, int N, int keep,
                                                             float *__restrict__ out, int64_t *__restrict__ order) {
    extern __shared__ float sc[];
    const int b = blockIdx.y;
    const float *s = score + (size_t)b * N;
    // (a NaN score compares false to everything: every NaN point would get the rank of the largest score and collide with it, leaving
    // slots of `order` / `out` unwritten.  torch.argsort -- the reference, models/Point_MAE_unify.py:553-559 -- sorts NaN as the LARGEST
    // value: NaN -> +inf here, ties in index order like every other tie, so the ranks stay a permutation.)
    for (int i = threadIdx.x; i < N; i += 256) { const float v = s[i]; sc[i] = v != v ? __builtin_inff() : v; }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float mine = sc[i];
    int rank = 0;
    int j = 0;
    for (; j + 4 <= N; j += 4) {                          // (every lane reads the same address: LDS broadcast)
        const float4 v = *reinterpret_cast<const float4 *>(sc + j);
        rank += (v.x > mine || (v.x == mine && j < i)) + (v.y > mine || (v.y == mine && j + 1 < i)) + (v.z > mine || (v.z == mine && j + 2 < i)) +
                (v.w > mine || (v.w == mine && j + 3 < i));
    }
    for (; j < N; ++j) rank += sc[j] > mine || (sc[j] == mine && j < i);
    if (order) order[(size_t)b * N + rank] = i;
    const int slot = rank - (N - keep);
    if (slot >= 0) {
        const float *q = moved + ((size_t)b * N + i) * 3;
        float *o = out + ((size_t)b * keep + slot) * 3;
        o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
    }
}

// order[b][rank] = i: the stable argsort of every row of `key` (B,N) by rank counting, as rectify_select_kernel ranks its scores -- one
// workgroup per (row, 256 elements), the row in the LDS, rank of an element = how many elements precede it (smaller key -- greater, if
// descending -- or equal key and lower index: the order of torch's stable sort).  The comparison runs on a TOTAL-ORDER image of the
// floats: -0 == +0 (ties by index), NaN above +inf -- last ascending, first descending, as torch.sort ranks it; always a permutation.
__device__ __forceinline__ uint32_t sort_key_f32(float v) {
    if (v != v) return 0xFFFFFFFFu;                        // NaN: above +inf (0xFF800000)
    const uint32_t u = __float_as_uint(v + 0.0f);          // (-0 + 0 = +0)
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <bool DESC>
__global__ __launch_bounds__(256) void argsort_rows_kernel(const float *__restrict__ key, long long B, int N, int64_t *__restrict__ order) {
    extern __shared__ uint32_t sk[];
    const long long b = (long long)blockIdx.z * 65535 + blockIdx.y;          // (rows beyond the 65,535 of grid.y: grid.z)
    if (b >= B) return;
    const float *s = key + (size_t)b * N;
    for (int i = threadIdx.x; i < N; i += 256) sk[i] = sort_key_f32(s[i]);
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t mine = sk[i];
    int rank = 0;
    int j = 0;
    auto before = [&](uint32_t v, int jj) { return (DESC ? v > mine : v < mine) || (v == mine && jj < i); };
    for (; j + 4 <= N; j += 4) {                          // (every lane reads the same address: LDS broadcast)
        const uint4 v = *reinterpret_cast<const uint4 *>(sk + j);
        rank += before(v.x, j) + before(v.y, j + 1) + before(v.z, j + 2) + before(v.w, j + 3);
    }
    for (; j < N; ++j) rank += before(sk[j], j);
    order[(size_t)b * N + rank] = i;
}

}  // namespace

extern "C" int upp_argsort_rows(const float *key, int B, int N, int descending, int64_t *order, void *stream) {
    if (!key || !order || B < 1 || N < 1) return UPP_E_BADARG;
    if (N > 16384) return UPP_E_RANGE;
    const dim3 grid((N + 255) / 256, B < 65535 ? B : 65535, (B + 65534) / 65535);
    const size_t lds = (size_t)((N + 3) / 4 * 4) * sizeof(float);
    if (descending) hipLaunchKernelGGL(argsort_rows_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, key, (long long)B, N, order);
    else hipLaunchKernelGGL(argsort_rows_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, key, (long long)B, N, order);
    return upp_launch_status();
}

extern "C" int upp_bn_relu_drop_fwd(const float *z, const float *gamma, const float *beta, float *running_mean, float *running_var,
                                    float momentum, float eps, int training, const float *u, float p, float *a, float *mean,
                                    float *rstd, int R, int C, void *stream) {
    if (!z || !gamma || !beta || !a || !mean || !rstd || R < 1 || C < 1 || p < 0.0f || p >= 1.0f) return UPP_E_BADARG;
    if (!training && (!running_mean || !running_var)) return UPP_E_BADARG;
    hipLaunchKernelGGL(bn_relu_drop_fwd_kernel, dim3((C + 63) / 64), dim3(64 * kBW), 0, (hipStream_t)stream, z, gamma, beta, running_mean,
                       running_var, momentum, eps, training, u, p, a, mean, rstd, R, C);
    return upp_launch_status();
}

extern "C" int upp_bn_relu_drop_bwd(const float *g_a, const float *z, const float *gamma, const float *beta, const float *mean,
                                    const float *rstd, int training, const float *u, float p, float *g_z, float *g_gamma,
                                    float *g_beta, int R, int C, void *stream) {
    if (!g_a || !z || !gamma || !beta || !mean || !rstd || !g_z || !g_gamma || !g_beta || R < 1 || C < 1 || p < 0.0f || p >= 1.0f)
        return UPP_E_BADARG;
    hipLaunchKernelGGL(bn_relu_drop_bwd_kernel, dim3((C + 63) / 64), dim3(64 * kBW), 0, (hipStream_t)stream, g_a, z, gamma, beta, mean, rstd,
                       training, u, p, g_z, g_gamma, g_beta, R, C);
    return upp_launch_status();
}

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

extern "C" int upp_logsoftmax_rows_fwd(const float *y, long long ld_y, const float *bias, long long R, int C, float *logp, void *stream) {
    if (!y || !logp || R < 1 || C < 1 || ld_y < C) return UPP_E_BADARG;
    if (C > 64 || (R + 4 * kLsRows - 1) / (4 * kLsRows) > 0x7fffffffLL) return UPP_E_RANGE;
    hipLaunchKernelGGL(logsoftmax_rows_fwd_kernel, dim3((unsigned)((R + 4 * kLsRows - 1) / (4 * kLsRows))), dim3(256), 0, (hipStream_t)stream, y, ld_y, bias, R, C, logp);
    return upp_launch_status();
}

extern "C" int upp_logsoftmax_rows_bwd(const float *g_logp, const float *logp, long long R, int C, float *g_y, long long ld_gy, int Cpad, void *stream) {
    if (!g_logp || !logp || !g_y || R < 1 || C < 1 || Cpad < C || ld_gy < Cpad) return UPP_E_BADARG;
    if (Cpad > 64 || (R + 4 * kLsRows - 1) / (4 * kLsRows) > 0x7fffffffLL) return UPP_E_RANGE;
    hipLaunchKernelGGL(logsoftmax_rows_bwd_kernel, dim3((unsigned)((R + 4 * kLsRows - 1) / (4 * kLsRows))), dim3(256), 0, (hipStream_t)stream, g_logp, logp, R, C,
                       g_y, ld_gy, Cpad);
    return upp_launch_status();
}

extern "C" long long upp_nll_mean_part_floats(long long R) { return R < 1 ? 0 : (R + 4 * kNllRows - 1) / (4 * kNllRows); }

extern "C" int upp_nll_mean_fwd(const float *logp, const int64_t *target, long long R, int C, float *part, float *out, void *stream) {
    if (!logp || !target || !part || !out || R < 1 || C < 1) return UPP_E_BADARG;
    const long long n = (R + 4 * kNllRows - 1) / (4 * kNllRows);
    if (n > 0x7fffffffLL) return UPP_E_RANGE;
    hipLaunchKernelGGL(nll_partial_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, logp, target, R, C, part);
    hipLaunchKernelGGL(nll_final_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, part, (int)n, -1.0f / (float)R, out);
    return upp_launch_status();
}

extern "C" int upp_nll_mean_bwd(const float *g_loss, const int64_t *target, long long R, int C, float *g_logp, void *stream) {
    if (!g_loss || !target || !g_logp || R < 1 || C < 1) return UPP_E_BADARG;
    const long long blocks = (R * C + 255) / 256;
    if (blocks > 0x7fffffffLL) return UPP_E_RANGE;
    hipLaunchKernelGGL(nll_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g_loss, target, R, C, g_logp);
    return upp_launch_status();
}

extern "C" long long upp_noise_loss_part_floats(int B, int P) { return (B < 1 || P < 1) ? 0 : ((long long)B * P + 255) / 256; }

extern "C" int upp_noise_loss_fwd(const float *pred, const float *noise_vector, int B, int P, int pn, float *part, float *loss, float *score,
                                  void *stream) {
    if (!pred || !noise_vector || !part || !loss || B < 1 || P < 2 || pn < 1 || pn >= P) return UPP_E_BADARG;
    const long long n = ((long long)B * P + 255) / 256;
    if (n > 0x7fffffffLL) return UPP_E_RANGE;
    hipLaunchKernelGGL(noise_loss_partial_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, pred, noise_vector, B, P, pn,
                       1.0f / ((float)B * (float)pn), 1.0f / ((float)B * (float)(P - pn)), part, score);
    hipLaunchKernelGGL(nll_final_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, part, (int)n, 1.0f, loss);
    return upp_launch_status();
}

extern "C" int upp_noise_loss_bwd(const float *g_loss, const float *pred, const float *noise_vector, int B, int P, int pn, float *g_pred, void *stream) {
    if (!g_loss || !pred || !noise_vector || !g_pred || B < 1 || P < 2 || pn < 1 || pn >= P) return UPP_E_BADARG;
    const long long n = ((long long)B * P + 255) / 256;
    if (n > 0x7fffffffLL) return UPP_E_RANGE;
    hipLaunchKernelGGL(noise_loss_bwd_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, g_loss, pred, noise_vector, B, P, pn,
                       1.0f / ((float)B * (float)pn), 1.0f / ((float)B * (float)(P - pn)), g_pred);
    return upp_launch_status();
}

extern "C" int upp_rectify_select(const float *feature, const float *W0, const float *b0, const float *W1, const float *b1, const float *u, float p,
                                  float factor, const float *pts, float nudge, int B, int N, int keep, float *pred, float *moved, float *score,
                                  float *out, int64_t *order, void *stream) {
    if (!feature || !W0 || !b0 || !W1 || !b1 || !pts || !moved || !score || !out || B < 1 || N < 1 || keep < 1) return UPP_E_BADARG;
    if (keep > N || N > 16384 || p < 0.0f || p >= 1.0f || (reinterpret_cast<uintptr_t>(feature) & 15)) return UPP_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    const int rows = B * N;
    hipLaunchKernelGGL(rectify_score_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, feature, W0, b0, W1, b1, u, p, factor, pts, nudge, rows, pred,
                       moved, score);
    hipLaunchKernelGGL(rectify_select_kernel, dim3((N + 255) / 256, B), dim3(256), (size_t)((N + 3) / 4 * 4) * sizeof(float), st, score, moved, N, keep,
                       out, order);
    return upp_launch_status();
}
