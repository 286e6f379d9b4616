// block.hip -- fused kernels for the Transformer Block of UPP
// (reference models/Point_MAE_pretask_dev.py: Attention :172-196, Block :199-321).
//
// The reference block issues ~60 small kernels per forward (pos add, prompt concat, LayerNorm,
// qkv permutes, q.k^T, scale, softmax, attn.v, transpose copy, drop-path mask arithmetic, residual
// adds, ...) and about twice that in backward; at B*L = 2400 tokens every one of them sits on the
// ~5 us small-kernel floor.  Two kernel families replace the glue around the GEMMs:
//
//  * rowln_{fwd,bwd}: "gather rows (+pos) (+ per-sample-scaled residual branch) -> LayerNorm".
//    One wavefront per token row (D <= 512: up to 8 elements per lane), statistics by DPP wave
//    reductions.  Covers: pos-add + prompt insertion + norm1; drop-path residual + norm2; drop-path
//    residual + prompt strip + adapter LayerNorm.  The stochastic-depth factor floor(keep+u)/keep is
//    computed inline from a per-sample uniform, so no mask kernels exist.
//  * attn_{fwd,bwd}: softmax(q k^T * scale) v for one (sample, head) per workgroup, reading q/k/v
//    straight out of the packed qkv GEMM output (B, L, 3, H, 64) and writing the context in the
//    (B, L, H*64) layout the projection GEMM consumes -- no permute copies, no (B,H,L,L) tensors in
//    HBM.  Keys live in VGPRs (lane = key), values in LDS; backward recomputes the probabilities from
//    the saved log-sum-exp and accumulates dK / dV in registers (lane = channel).
// All arithmetic f32 (parity bar 1e-5 rel against the reference's unfused ops).
#include "common.h"

namespace {

constexpr int kMaxE = 8;  // elements per lane in a row kernel -> D <= 512

__device__ __forceinline__ float wave_sum(float v) { return wave_sum_f32(v); }

struct RowLnArgs {
    const float *x;        // (B, Lin, D) source rows
    const float *add;      // (B, Lin, D) added to x rows (positional embedding) or null
    const float *prompts;  // (P, D) rows selected by negative row_src values, or null
    int mode, P;           // row map (see row_src): 0 identity, 1/2 insert P prompts (with/without cls), 3/4 strip them
    const float *y;        // (B, Lin, D) residual branch, row-aligned with x: added as scale_b * (y + ybias), or null
    const float *ybias;    // (D) bias of the Linear that produced y (its GEMM then runs bias-free), or null
    const float *u;        // (B) uniforms for stochastic depth (scale_b = floor(keep+u)/keep), or null (scale 1)
    float keep;
    const float *gamma, *beta;  // LayerNorm affine (null gamma: no LayerNorm, only xo is produced)
    float eps;
    float *xo;             // (B, Lout, D) assembled rows (may be null)
    float *h;              // (B, Lout, D) LayerNorm output
    float *mean, *rstd;    // (B, Lout) saved statistics
    int B, Lin, Lout, D;
    int yparts; long long ystride;   // y = sum of yparts partial matrices ystride floats apart (upp_linear_parts_f32), added in order
};

// Source row of output row t; -(p+1) selects prompt p.  Token layouts: [cls | prompts | tokens] or [prompts | tokens].
__device__ __forceinline__ int row_src(int t, int mode, int P) {
    switch (mode) {
        case 1: return t == 0 ? 0 : (t <= P ? -t : t - P);   // insert P prompts after the cls token
        case 2: return t < P ? -(t + 1) : t - P;             // insert P prompts in front
        case 3: return t == 0 ? 0 : t + P;                   // strip the prompts that follow the cls token
        case 4: return t + P;                                // strip leading prompts
        default: return t;
    }
}

__device__ __forceinline__ float dp_scale(const float *u, float keep, int b) {
    return u ? floorf(keep + u[b]) / keep : 1.0f;
}

// Loads of a row kernel are issued in one batch on clamped column indices (no divergent branch around a load):
// a load inside `if (c < D)` followed by its use costs one full memory round trip per element slot.
__global__ __launch_bounds__(256) void rowln_fwd_kernel(RowLnArgs a) {
    const int lane = threadIdx.x & 63;
    const int wg = xcd_contiguous(blockIdx.x, gridDim.x);
    const int row = wg * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (row >= a.B * a.Lout) return;
    const int b = row / a.Lout, t = row - b * a.Lout;
    const int src = row_src(t, a.mode, a.P);
    const int D = a.D;
    const float *xs = src >= 0 ? a.x + ((size_t)b * a.Lin + src) * D : a.prompts + (size_t)(-src - 1) * D;
    const bool has_ad = src >= 0 && a.add, has_y = src >= 0 && a.y, has_ln = a.gamma != nullptr, has_yb = has_y && a.ybias;
    const float *ad = has_ad ? a.add + ((size_t)b * a.Lin + src) * D : xs;
    const float *ys = has_y ? a.y + ((size_t)b * a.Lin + src) * D : xs;
    const float *gm = has_ln ? a.gamma : xs, *bt = has_ln ? a.beta : xs;
    int cc[kMaxE];
    float xv[kMaxE], av[kMaxE], yv[kMaxE], gv[kMaxE], bv[kMaxE];
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) { cc[e] = min(lane + 64 * e, D - 1); xv[e] = xs[cc[e]]; }
    if (has_ad) {
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) av[e] = ad[cc[e]];
    }
    if (has_y) {
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) yv[e] = ys[cc[e]];
        for (int p = 1; p < a.yparts; ++p) {        // the k-parts of the GEMM that produced y: added here, in part order
            float yp[kMaxE];
#pragma unroll
            for (int e = 0; e < kMaxE; ++e) yp[e] = ys[(size_t)p * a.ystride + cc[e]];
#pragma unroll
            for (int e = 0; e < kMaxE; ++e) yv[e] += yp[e];
        }
    }
    if (has_yb) {
        float yb[kMaxE];
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) yb[e] = a.ybias[cc[e]];
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) yv[e] += yb[e];
    }
    if (has_ln) {
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) { gv[e] = gm[cc[e]]; bv[e] = bt[cc[e]]; }
    }
    const float sc = has_y ? dp_scale(a.u, a.keep, b) : 0.0f;
    float v[kMaxE];
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) {
        float val = xv[e];
        if (has_ad) val += av[e];
        if (has_y) val = __builtin_fmaf(yv[e], sc, val);
        v[e] = lane + 64 * e < D ? val : 0.0f;
        s += v[e];
    }
    if (a.xo) {
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) { const int c = lane + 64 * e; if (c < D) a.xo[(size_t)row * D + c] = v[e]; }
    }
    if (!has_ln) return;
    const float mean = wave_sum(s) / (float)D;
    float q = 0.0f;
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) { const int c = lane + 64 * e; const float dv = c < D ? v[e] - mean : 0.0f; q = __builtin_fmaf(dv, dv, q); }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + a.eps);
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) {
        const int c = lane + 64 * e;
        if (c < D) a.h[(size_t)row * D + c] = __builtin_fmaf((v[e] - mean) * rstd, gv[e], bv[e]);
    }
    if (lane == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
}

struct RowLnBwdArgs {
    const float *g_xo;     // (B, Lout, D) gradient w.r.t. the assembled rows (may be null)
    const float *g_h;      // (B, Lout, D) gradient w.r.t. the LayerNorm output (null when no LayerNorm)
    const float *xo;       // (B, Lout, D) saved assembled rows
    const float *mean, *rstd, *gamma;
    int mode;
    const float *u; float keep;
    float *g_x;            // (B, Lin, D): every row is written (zeros for the prompt rows a strip map drops)
    float *g_prompt;       // (B, P, D) per-sample gradient of the prompt rows, or null
    float *g_y;            // (B, Lin, D) gradient of the residual branch (same rows as g_x), or null
    float *ln_part;        // (workgroups, 2, D) partial d_gamma / d_beta per workgroup of 4 rows, or null
    int B, Lin, Lout, D, P;
    int gparts; long long gstride;   // g_h = sum of gparts partial matrices gstride floats apart (upp_linear_parts_f32), added in order
};

__global__ __launch_bounds__(256) void rowln_bwd_kernel(RowLnBwdArgs a) {
    __shared__ float lnp[2][4][64 * kMaxE];   // per-wave LayerNorm parameter-gradient contributions (only used with ln_part)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = xcd_contiguous(blockIdx.x, gridDim.x);
    const int it = wg * 4 + wave;   // wave-uniform
    const int D = a.D;
    // Insert / identity maps are walked by OUTPUT row; strip maps by INPUT row, so that the prompt rows the forward
    // dropped get their zero gradient here (no separate fill launch).
    const bool by_input = a.mode == 3 || a.mode == 4;
    const int Lg = by_input ? a.Lin : a.Lout;
    bool live = it < a.B * Lg;
    const int b = live ? it / Lg : 0, t = live ? it - b * Lg : 0;
    int src = 0, t_out = 0;
    if (live) {
        if (by_input) {
            src = t;
            t_out = a.mode == 3 ? (t == 0 ? 0 : (t <= a.P ? -1 : t - a.P)) : (t < a.P ? -1 : t - a.P);
            if (t_out < 0) {
                float *zx = a.g_x ? a.g_x + ((size_t)b * a.Lin + t) * D : nullptr;
                float *zy = a.g_y ? a.g_y + ((size_t)b * a.Lin + t) * D : nullptr;
#pragma unroll
                for (int e = 0; e < kMaxE; ++e) {
                    const int c = lane + 64 * e;
                    if (c < D) { if (zx) zx[c] = 0.0f; if (zy) zy[c] = 0.0f; }
                }
                live = false;
            }
        } else {
            t_out = t;
            src = row_src(t, a.mode, a.P);
        }
    }
    float pgam[kMaxE], pbet[kMaxE];         // this row's g_h * xhat and g_h (LayerNorm parameter gradients)
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) { pgam[e] = 0.0f; pbet[e] = 0.0f; }
    if (live) {
        const int row = b * a.Lout + t_out;
        const bool has_gxo = a.g_xo != nullptr, has_ln = a.g_h != nullptr;
        // one batch of loads (clamped columns), then the arithmetic
        int cc[kMaxE];
        float d[kMaxE], gh[kMaxE], gm[kMaxE], xo[kMaxE];
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) cc[e] = min(lane + 64 * e, D - 1);
        if (has_gxo) {
#pragma unroll
            for (int e = 0; e < kMaxE; ++e) d[e] = a.g_xo[(size_t)row * D + cc[e]];
        }
        float mean = 0.0f, rstd = 0.0f;
        if (has_ln) {
#pragma unroll
            for (int e = 0; e < kMaxE; ++e) { gh[e] = a.g_h[(size_t)row * D + cc[e]]; gm[e] = a.gamma[cc[e]]; xo[e] = a.xo[(size_t)row * D + cc[e]]; }
            for (int p = 1; p < a.gparts; ++p) {    // the k-parts of the data-gradient GEMM that produced g_h
                float gp[kMaxE];
#pragma unroll
                for (int e = 0; e < kMaxE; ++e) gp[e] = a.g_h[(size_t)p * a.gstride + (size_t)row * D + cc[e]];
#pragma unroll
                for (int e = 0; e < kMaxE; ++e) gh[e] += gp[e];
            }
            mean = a.mean[row]; rstd = a.rstd[row];
        }
        float *gx = nullptr;
        if (src >= 0) { if (a.g_x) gx = a.g_x + ((size_t)b * a.Lin + src) * D; }
        else if (a.g_prompt) gx = a.g_prompt + ((size_t)b * a.P + (-src - 1)) * D;
        float *gy = (src >= 0 && a.g_y) ? a.g_y + ((size_t)b * a.Lin + src) * D : nullptr;
        const float sc = gy ? dp_scale(a.u, a.keep, b) : 0.0f;
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) d[e] = (has_gxo && lane + 64 * e < D) ? d[e] : 0.0f;
        if (has_ln) {
            // dx = rstd * (dy - mean(dy) - xhat * mean(dy * xhat)),  dy = g_h * gamma
            float dy[kMaxE], xh[kMaxE], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int e = 0; e < kMaxE; ++e) {
                const bool ok = lane + 64 * e < D;
                dy[e] = ok ? gh[e] * gm[e] : 0.0f;
                xh[e] = ok ? (xo[e] - mean) * rstd : 0.0f;
                pgam[e] = ok ? gh[e] * xh[e] : 0.0f;
                pbet[e] = ok ? gh[e] : 0.0f;
                s1 += dy[e];
                s2 = __builtin_fmaf(dy[e], xh[e], s2);
            }
            s1 = wave_sum(s1) / (float)D;
            s2 = wave_sum(s2) / (float)D;
#pragma unroll
            for (int e = 0; e < kMaxE; ++e) d[e] += rstd * (dy[e] - s1 - xh[e] * s2);
        }
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) {
            const int c = lane + 64 * e;
            if (c < D) {
                if (gx) gx[c] = d[e];
                if (gy) gy[c] = d[e] * sc;
            }
        }
    }
    if (a.ln_part) {
        // LayerNorm parameter gradients: the workgroup's 4 rows are summed in wave order into one partial row pair;
        // the caller sums the partials over the workgroups (upp_batched_sum): deterministic, no extra pass over g_h / xo.
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) { lnp[0][wave][lane + 64 * e] = pgam[e]; lnp[1][wave][lane + 64 * e] = pbet[e]; }
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256) {
            a.ln_part[((size_t)wg * 2 + 0) * D + c] = (lnp[0][0][c] + lnp[0][1][c]) + (lnp[0][2][c] + lnp[0][3][c]);
            a.ln_part[((size_t)wg * 2 + 1) * D + c] = (lnp[1][0][c] + lnp[1][1][c]) + (lnp[1][2][c] + lnp[1][3][c]);
        }
    }
}

// Partial LayerNorm parameter gradients (only for trainable LayerNorms: the adapter's):
// part[0][chunk][c] = sum over the chunk's rows of g_h * xhat, part[1][chunk][c] = sum of g_h.
// grid = (ceil(D/64), chunks); lane = column, the 4 waves stride the chunk's rows and are combined in wave
// order; the caller sums the `chunks` partials (fixed order: deterministic).
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const float *__restrict__ g_h, const float *__restrict__ xo,
                                                            const float *__restrict__ mean, const float *__restrict__ rstd,
                                                            float *__restrict__ part, int rows, int D) {
    __shared__ float pg[4][64], pb[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int chunks = gridDim.y, per = (rows + chunks - 1) / chunks;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    float sg = 0.0f, sb = 0.0f;
    if (c < D)
        for (int r = r0 + wave; r < r1; r += 4) {
            const float g = g_h[(size_t)r * D + c];
            sg = __builtin_fmaf(g, (xo[(size_t)r * D + c] - mean[r]) * rstd[r], sg);
            sb += g;
        }
    pg[wave][lane] = sg; pb[wave][lane] = sb;
    __syncthreads();
    if (wave == 0 && c < D) {
        part[((size_t)0 * chunks + blockIdx.y) * D + c] = (pg[0][lane] + pg[1][lane]) + (pg[2][lane] + pg[3][lane]);
        part[((size_t)1 * chunks + blockIdx.y) * D + c] = (pb[0][lane] + pb[1][lane]) + (pb[2][lane] + pb[3][lane]);
    }
}

// h = GELU(z + b) (erf form) of a Linear's bias-free GEMM output z (rows, C), and its backward g_z = g_h * GELU'(z + b).
// The GEMM runs without its bias epilogue (the bias-free library kernels are 1.5-3 us faster at these shapes) and the
// bias costs nothing here.  C % 4 == 0.
__device__ __forceinline__ float gelu_val(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_der(float v) {
    return 0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * 0.39894228040143267794f * expf(-0.5f * v * v);
}
__global__ __launch_bounds__(256) void bias_gelu_fwd_kernel(const float *__restrict__ z, const float *__restrict__ b,
                                                            float *__restrict__ h, long long total4, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const float4 v = reinterpret_cast<const float4 *>(z)[i];
    const float4 bb = *reinterpret_cast<const float4 *>(b + (int)((i * 4) % C));
    reinterpret_cast<float4 *>(h)[i] = make_float4(gelu_val(v.x + bb.x), gelu_val(v.y + bb.y), gelu_val(v.z + bb.z), gelu_val(v.w + bb.w));
}
// forward that also leaves d = GELU'(z + b) behind: the backward is then one multiply per element (g_z = g_h * d) instead of
// a second erf + exp per element (the erf is shared between the value and the derivative here)
__global__ __launch_bounds__(256) void bias_gelu_fwd_d_kernel(const float *__restrict__ z, const float *__restrict__ b,
                                                              float *__restrict__ h, float *__restrict__ d, long long total4, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const float4 v = reinterpret_cast<const float4 *>(z)[i];
    const float4 bb = *reinterpret_cast<const float4 *>(b + (int)((i * 4) % C));
    const float x[4] = {v.x + bb.x, v.y + bb.y, v.z + bb.z, v.w + bb.w};
    float hv[4], dv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { hv[q] = gelu_val(x[q]); dv[q] = gelu_der(x[q]); }
    reinterpret_cast<float4 *>(h)[i] = make_float4(hv[0], hv[1], hv[2], hv[3]);
    reinterpret_cast<float4 *>(d)[i] = make_float4(dv[0], dv[1], dv[2], dv[3]);
}
__global__ __launch_bounds__(256) void bias_gelu_bwd_kernel(const float *__restrict__ g_h, const float *__restrict__ z,
                                                            const float *__restrict__ b, float *__restrict__ g_z, long long total4, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const float4 g = reinterpret_cast<const float4 *>(g_h)[i];
    const float4 v = reinterpret_cast<const float4 *>(z)[i];
    const float4 bb = *reinterpret_cast<const float4 *>(b + (int)((i * 4) % C));
    reinterpret_cast<float4 *>(g_z)[i] = make_float4(g.x * gelu_der(v.x + bb.x), g.y * gelu_der(v.y + bb.y), g.z * gelu_der(v.z + bb.z),
                                                     g.w * gelu_der(v.w + bb.w));
}

}  // namespace

static int rowln_fwd_impl(const float *x, const float *add, const float *prompts, int mode, int P, const float *y, int yparts,
                                   long long ystride, const float *ybias, const float *u, float keep, const float *gamma, const float *beta,
                                   float eps, float *xo, float *h, float *mean, float *rstd, int B, int Lin, int Lout, int D, void *stream) {
    if (yparts < 1 || (yparts > 1 && (!y || ystride < (long long)B * Lin * D))) return UPP_E_BADARG;
    if (!x || B < 0 || Lin < 1 || Lout < 1 || D < 1) return UPP_E_BADARG;
    if (gamma && (!beta || !h || !mean || !rstd)) return UPP_E_BADARG;
    if (!gamma && !xo) return UPP_E_BADARG;
    if (D > 64 * kMaxE) return UPP_E_RANGE;
    if (B == 0) return 0;
    if (mode < 0 || mode > 4 || P < 0 || ((mode == 1 || mode == 2) && (Lout != Lin + P || (P > 0 && !prompts))) ||
        ((mode == 3 || mode == 4) && Lout != Lin - P) || (mode == 0 && Lout != Lin))
        return UPP_E_BADARG;
    if (ybias && !y) return UPP_E_BADARG;
    RowLnArgs a{x, add, prompts, mode, P, y, ybias, u, keep, gamma, beta, eps, xo, h, mean, rstd, B, Lin, Lout, D, yparts, ystride};
    hipLaunchKernelGGL(rowln_fwd_kernel, dim3((B * Lout + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    return upp_launch_status();
}

extern "C" int upp_rowln_fwd(const float *x, const float *add, const float *prompts, int mode, int P, const float *y,
                             const float *ybias, const float *u, float keep, const float *gamma, const float *beta, float eps, float *xo, float *h,
                             float *mean, float *rstd, int B, int Lin, int Lout, int D, void *stream) {
    return rowln_fwd_impl(x, add, prompts, mode, P, y, 1, 0, ybias, u, keep, gamma, beta, eps, xo, h, mean, rstd, B, Lin, Lout, D, stream);
}

extern "C" long long upp_rowln_part_floats(int B, int Lin, int Lout, int D, int mode) {
    if (B < 1 || Lin < 1 || Lout < 1 || D < 1) return 0;
    const int Lg = (mode == 3 || mode == 4) ? Lin : Lout;
    return (long long)((B * Lg + 3) / 4) * 2 * D;
}

static int rowln_bwd_impl(const float *g_xo, const float *g_h, int gparts, long long gstride, const float *xo, const float *mean,
                                   const float *rstd, const float *gamma, int mode, const float *u, float keep, float *g_x, float *g_prompt,
                                   float *g_y, float *ln_part, int B, int Lin, int Lout, int D, int P, void *stream) {
    if (gparts < 1 || (gparts > 1 && (!g_h || gstride < (long long)B * Lout * D))) return UPP_E_BADARG;
    if ((!g_xo && !g_h) || B < 0 || Lin < 1 || Lout < 1 || D < 1) return UPP_E_BADARG;
    if (g_h && (!xo || !mean || !rstd || !gamma)) return UPP_E_BADARG;
    if (D > 64 * kMaxE) return UPP_E_RANGE;
    if (B == 0) return 0;
    if (mode < 0 || mode > 4 || P < 0) return UPP_E_BADARG;
    if (ln_part && !g_h) return UPP_E_BADARG;
    RowLnBwdArgs a{g_xo, g_h, xo, mean, rstd, gamma, mode, u, keep, g_x, g_prompt, g_y, ln_part, B, Lin, Lout, D, P, gparts, gstride};
    const int Lg = (mode == 3 || mode == 4) ? Lin : Lout;      // strip maps are walked by input row (see the kernel)
    hipLaunchKernelGGL(rowln_bwd_kernel, dim3((B * Lg + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    return upp_launch_status();
}

extern "C" int upp_rowln_bwd(const float *g_xo, const float *g_h, const float *xo, const float *mean, const float *rstd,
                             const float *gamma, int mode, const float *u, float keep, float *g_x, float *g_prompt,
                             float *g_y, float *ln_part, int B, int Lin, int Lout, int D, int P, void *stream) {
    return rowln_bwd_impl(g_xo, g_h, 1, 0, xo, mean, rstd, gamma, mode, u, keep, g_x, g_prompt, g_y, ln_part, B, Lin, Lout, D, P, stream);
}

extern "C" int upp_ln_param_grad(const float *g_h, const float *xo, const float *mean, const float *rstd, float *part,
                                 int rows, int D, int chunks, void *stream) {
    if (!g_h || !xo || !mean || !rstd || !part || rows < 1 || D < 1 || chunks < 1) return UPP_E_BADARG;
    hipLaunchKernelGGL(ln_param_grad_kernel, dim3((D + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, g_h, xo, mean, rstd,
                       part, rows, D);
    return upp_launch_status();
}

// attention core, head dim 64: qkv (B, L, 3, H, 64); ctx (B, L, H*64); lse (B, H, L).  L <= 96: attn_flash16.hip (operands in registers);
// L <= 160: attn_long.hip.  (Rounds 1-3 also shipped a VALU pair and the LDS-staged 32x32x2 pair of attn_mfma.hip behind a `variant`
// argument, for measurements: removed from the library in round 4 -- tools/micro/attn_mfma.hip keeps the latter reproducible.)
int upp_attn_fwd_long(const float *qkv, float *ctx, float *lse, int B, int L, int H, float scale, hipStream_t st);
int upp_attn_bwd_long(const float *qkv, const float *ctx, const float *d_ctx, const float *lse, float *d_qkv, int B, int L, int H,
                      float scale, hipStream_t st);
int upp_attn_fwd_flash16(const float *qkv, float *ctx, float *lse, int B, int L, int H, float scale, hipStream_t st);
int upp_attn_bwd_flash16(const float *qkv, const float *ctx, const float *d_ctx, const float *lse, float *d_qkv, int B, int L, int H,
                         float scale, hipStream_t st);
extern "C" int upp_attn_fwd(const float *qkv, float *ctx, float *lse, int B, int L, int H, int head_dim, float scale, void *stream) {
    if (!qkv || !ctx || !lse || B < 0 || L < 1 || H < 1) return UPP_E_BADARG;
    if (head_dim != 64 || L > 160) return UPP_E_RANGE;
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    return L <= 96 ? upp_attn_fwd_flash16(qkv, ctx, lse, B, L, H, scale, st) : upp_attn_fwd_long(qkv, ctx, lse, B, L, H, scale, st);
}

extern "C" int upp_attn_bwd(const float *qkv, const float *ctx, const float *d_ctx, const float *lse, float *d_qkv, int B, int L,
                            int H, int head_dim, float scale, void *stream) {
    if (!qkv || !ctx || !d_ctx || !lse || !d_qkv || B < 0 || L < 1 || H < 1) return UPP_E_BADARG;
    if (head_dim != 64 || L > 160) return UPP_E_RANGE;
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    return L <= 96 ? upp_attn_bwd_flash16(qkv, ctx, d_ctx, lse, d_qkv, B, L, H, scale, st) : upp_attn_bwd_long(qkv, ctx, d_ctx, lse, d_qkv, B, L, H, scale, st);
}

extern "C" int upp_bias_gelu_fwd(const float *z, const float *bias, float *h, long long rows, int C, void *stream) {
    if (!z || !bias || !h || rows < 1 || C < 4) return UPP_E_BADARG;
    if (C % 4 != 0) return UPP_E_RANGE;
    const long long total4 = rows * C / 4;
    hipLaunchKernelGGL(bias_gelu_fwd_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, bias, h, total4, C);
    return upp_launch_status();
}

extern "C" int upp_bias_gelu_fwd_d(const float *z, const float *bias, float *h, float *d, long long rows, int C, void *stream) {
    if (!z || !bias || !h || !d || rows < 1 || C < 4) return UPP_E_BADARG;
    if (C % 4 != 0) return UPP_E_RANGE;
    const long long total4 = rows * C / 4;
    hipLaunchKernelGGL(bias_gelu_fwd_d_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, bias, h, d, total4,
                       C);
    return upp_launch_status();
}

extern "C" int upp_bias_gelu_bwd(const float *g_h, const float *z, const float *bias, float *g_z, long long rows, int C, void *stream) {
    if (!g_h || !z || !bias || !g_z || rows < 1 || C < 4) return UPP_E_BADARG;
    if (C % 4 != 0) return UPP_E_RANGE;
    const long long total4 = rows * C / 4;
    hipLaunchKernelGGL(bias_gelu_bwd_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g_h, z, bias, g_z, total4,
                       C);
    return upp_launch_status();
}
