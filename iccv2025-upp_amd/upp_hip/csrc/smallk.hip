// smallk.hip -- the Linear layers the matrix-core kernel (linear.hip) does not take: contraction lengths that are not a multiple of
// 4 or rows that are not 16-byte aligned -- the first layer of every position MLP (K = 3, reference models/Point_MAE_unify.py:
// `pos_embed = Linear(3,128) - GELU - Linear(128,384)`), the first point-wise layer of the rectify prompter's feature propagation
// (K = 59, models/Point_MAE_pretask_dev.py:475-517 via PointNetFeaturePropagation) -- with bias and activation in the same pass;
// and the transposed copy of a TRAINABLE weight that the data-gradient GEMM needs (dX = dY . W = upp_linear_f32(dY, W^T)).
//
//   y (M,N) = act( x (M,K) . W (N,K)^T + bias ),   K <= 64, N <= 256,   act: 0 none, 1 ReLU, 2 GELU (erf)
//
// VALU kernel: lanes run along n (coalesced stores), a thread owns TR = 4 rows of one output column.  W^T [k][n] and the row tile
// live in the LDS: per 4 values of k a thread does 4 conflict-free dword reads of W^T and TR 16-byte broadcast reads of x for
// 4 TR FMAs.  Sums run over k in ascending order (fmaf chain): the oracle is the plain loop.
#include "common.h"

namespace {

constexpr int kTR = 4;

__device__ __forceinline__ float act_f(float v, int act) {
    if (act == 1) return fmaxf(v, 0.0f);
    if (act == 2) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    return v;
}

__global__ __launch_bounds__(256) void linear_smallk_kernel(const float *__restrict__ x, long long ldx, const float *__restrict__ W,
                                                           long long ldw, const float *__restrict__ bias, float *__restrict__ y,
                                                           long long ldy, int M, int N, int K, int act, float *__restrict__ d, long long ldd) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Kp = (K + 3) & ~3;
    const int groups = 256 / N;                       // row groups per workgroup (N <= 256)
    const int RB = groups * kTR;                      // rows per workgroup
    float *Wt = sm;                                   // [Kp][N]
    float *Xs = sm + Kp * N;                          // [RB][Kp]   (Kp * N is a multiple of 4: 16-byte aligned)
    const int row0 = blockIdx.x * RB;
    for (int i = threadIdx.x; i < Kp * N; i += 256) {
        const int k = i / N, n = i - k * N;
        Wt[i] = k < K ? W[(long long)n * ldw + k] : 0.0f;
    }
    for (int i = threadIdx.x; i < RB * Kp; i += 256) {
        const int r = i / Kp, k = i - r * Kp;
        Xs[i] = (k < K && row0 + r < M) ? x[(long long)(row0 + r) * ldx + k] : 0.0f;
    }
    __syncthreads();
    const int grp = threadIdx.x / N, n = threadIdx.x - grp * N;
    if (grp >= groups) return;
    float acc[kTR];
#pragma unroll
    for (int t = 0; t < kTR; ++t) acc[t] = 0.0f;
    const float *xr = Xs + grp * kTR * Kp;
    for (int k = 0; k < Kp; k += 4) {
        const float w0 = Wt[(k + 0) * N + n], w1 = Wt[(k + 1) * N + n], w2 = Wt[(k + 2) * N + n], w3 = Wt[(k + 3) * N + n];
#pragma unroll
        for (int t = 0; t < kTR; ++t) {
            const float4 xv = *reinterpret_cast<const float4 *>(xr + t * Kp + k);
            acc[t] = __builtin_fmaf(xv.x, w0, acc[t]);
            acc[t] = __builtin_fmaf(xv.y, w1, acc[t]);
            acc[t] = __builtin_fmaf(xv.z, w2, acc[t]);
            acc[t] = __builtin_fmaf(xv.w, w3, acc[t]);
        }
    }
    const float b = bias ? bias[n] : 0.0f;
#pragma unroll
    for (int t = 0; t < kTR; ++t) {
        const int row = row0 + grp * kTR + t;
        if (row < M) {
            y[(long long)row * ldy + n] = act_f(acc[t] + b, act);
            if (d) {                                  // GELU'(z) beside GELU(z): the factor of the backward pass (upp_linear_smallk_gelu_d_f32)
                const float z = acc[t] + b;
                d[(long long)row * ldd + n] = 0.5f * (1.0f + erff(z * 0.70710678118654752440f)) + z * 0.39894228040143267794f * expf(-0.5f * z * z);
            }
        }
    }
}

// dst (cols, rows) = src (rows, cols)^T, 32 x 32 tiles through the LDS; blockIdx.z selects one of `count` equally shaped matrices
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ src, long long lds_, float *__restrict__ dst, long long ldd,
                                                       int rows, int cols) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < rows && c < cols) ? src[(long long)r * lds_ + c] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < cols && r < rows) dst[(long long)c * ldd + r] = tile[tx][ty + 8 * i];
    }
}

// Weight gradient of a small Linear over MANY rows:  part[chunk][n][k] = sum over the chunk's rows m of G[m][n] * X[m][k]
// (N K <= 2048 outputs, any alignment).  The library's GEMM heuristics pick dreadful solutions for such shapes (186 us for the
// 32 x 27 gradient over 34,432 rows of the pre-task recipe: 60 MFLOP).  Here a workgroup owns a chunk of rows, stages 32 rows of G
// and X at a time in the LDS, and thread t keeps the outputs t, t + 256, ... in registers (rows added in ascending order); the
// caller sums the chunks (upp_batched_sum).
constexpr int kSwRows = 32, kSwOut = 8;
__global__ __launch_bounds__(256) void smallk_wgrad_kernel(const float *__restrict__ G, long long ldg, const float *__restrict__ X, long long ldx,
                                                          float *__restrict__ part, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *Gs = sm, *Xs = sm + kSwRows * N;                 // [32][N], [32][K]
    const int chunks = gridDim.x, per = (M + chunks - 1) / chunks;
    const int r0 = blockIdx.x * per, r1 = min(M, r0 + per);
    const int O = N * K;
    int on[kSwOut], ok[kSwOut];
    float acc[kSwOut];
#pragma unroll
    for (int j = 0; j < kSwOut; ++j) {
        const int o = min((int)threadIdx.x + 256 * j, O - 1);
        on[j] = o / K; ok[j] = o - on[j] * K; acc[j] = 0.0f;
    }
    for (int b0 = r0; b0 < r1; b0 += kSwRows) {
        const int nr = min(kSwRows, r1 - b0);
        for (int i = threadIdx.x; i < kSwRows * N; i += 256) { const int r = i / N, c = i - r * N; Gs[i] = r < nr ? G[(long long)(b0 + r) * ldg + c] : 0.0f; }
        for (int i = threadIdx.x; i < kSwRows * K; i += 256) { const int r = i / K, c = i - r * K; Xs[i] = r < nr ? X[(long long)(b0 + r) * ldx + c] : 0.0f; }
        __syncthreads();
        for (int r = 0; r < kSwRows; ++r) {
#pragma unroll
            for (int j = 0; j < kSwOut; ++j) acc[j] = __builtin_fmaf(Gs[r * N + on[j]], Xs[r * K + ok[j]], acc[j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < kSwOut; ++j) {
        const int o = threadIdx.x + 256 * j;
        if (o < O) part[(size_t)blockIdx.x * O + o] = acc[j];
    }
}

// several matrices in one launch: blockIdx.z selects the matrix (grid x / y sized for the largest one)
constexpr int kMaxTr = 96;
struct TrJobs { const float *src[kMaxTr]; float *dst[kMaxTr]; int rows[kMaxTr], cols[kMaxTr]; };
__global__ __launch_bounds__(256) void transpose_batched_kernel(TrJobs t) {
    __shared__ float tile[32][33];
    const int j = blockIdx.z, rows = t.rows[j], cols = t.cols[j];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    if (c0 >= cols || r0 >= rows) return;                              // (workgroup-uniform)
    const float *src = t.src[j];
    float *dst = t.dst[j];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < rows && c < cols) ? src[(long long)r * cols + c] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < cols && r < rows) dst[(long long)c * rows + r] = tile[tx][ty + 8 * i];
    }
}

}  // namespace

extern "C" int upp_linear_smallk_wgrad_f32(const float *G, long long ldg, const float *X, long long ldx, float *partials, int M, int N, int K,
                                           int chunks, void *stream) {
    if (!G || !X || !partials || M < 1 || N < 1 || K < 1 || chunks < 1) return UPP_E_BADARG;
    if (N * K > 256 * kSwOut || ldg < N || ldx < K || chunks > 65535) return UPP_E_RANGE;
    const size_t lds = (size_t)kSwRows * (N + K) * sizeof(float);
    if (lds > 64 * 1024) return UPP_E_RANGE;
    hipLaunchKernelGGL(smallk_wgrad_kernel, dim3(chunks), dim3(256), lds, (hipStream_t)stream, G, ldg, X, ldx, partials, M, N, K);
    return upp_launch_status();
}

extern "C" int upp_transpose_batched_f32(const float *const *src, float *const *dst, const int *rows, const int *cols, int count, void *stream) {
    if (count < 0 || (count > 0 && (!src || !dst || !rows || !cols))) return UPP_E_BADARG;
    for (int j0 = 0; j0 < count; j0 += kMaxTr) {
        TrJobs t;
        const int n = count - j0 < kMaxTr ? count - j0 : kMaxTr;
        int mr = 1, mc = 1;
        for (int j = 0; j < n; ++j) {
            if (!src[j0 + j] || !dst[j0 + j] || rows[j0 + j] < 1 || cols[j0 + j] < 1) return UPP_E_BADARG;
            t.src[j] = src[j0 + j]; t.dst[j] = dst[j0 + j]; t.rows[j] = rows[j0 + j]; t.cols[j] = cols[j0 + j];
            mr = rows[j0 + j] > mr ? rows[j0 + j] : mr; mc = cols[j0 + j] > mc ? cols[j0 + j] : mc;
        }
        hipLaunchKernelGGL(transpose_batched_kernel, dim3((mc + 31) / 32, (mr + 31) / 32, n), dim3(256), 0, (hipStream_t)stream, t);
    }
    return upp_launch_status();
}

extern "C" int upp_linear_smallk_f32(const float *x, long long ldx, const float *W, long long ldw, const float *bias, float *y, long long ldy,
                                     int M, int N, int K, int act, void *stream) {
    if (!x || !W || !y || M < 1 || N < 1 || K < 1) return UPP_E_BADARG;
    if (K > 64 || N > 256 || ldx < K || ldw < K || ldy < N || act < 0 || act > 2) return UPP_E_RANGE;
    const int Kp = (K + 3) & ~3, RB = (256 / N) * kTR;
    const size_t lds = (size_t)(Kp * N + RB * Kp) * sizeof(float);     // <= 64 * 256 * 4 + ... < 160 KB; above 64 KB only for N > 192
    if (lds > 64 * 1024) {
        static std::atomic<bool> raised{false};
        if (!raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(linear_smallk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            raised = true;
        }
    }
    hipLaunchKernelGGL(linear_smallk_kernel, dim3((M + RB - 1) / RB), dim3(256), lds, (hipStream_t)stream, x, ldx, W, ldw, bias, y, ldy, M, N, K, act,
                       (float *)nullptr, 0LL);
    return upp_launch_status();
}

// y = GELU(x . W^T + bias) and d = GELU'(x . W^T + bias) in one pass: the first layer of a TRAINABLE position MLP (Linear(3,128) - GELU -
// Linear(128,D)) keeps the derivative for its backward instead of a torch gelu / gelu_backward pair (stage 2, the pre-task recipe).
extern "C" int upp_linear_smallk_gelu_d_f32(const float *x, long long ldx, const float *W, long long ldw, const float *bias, float *y, long long ldy,
                                            float *d, long long ldd, int M, int N, int K, void *stream) {
    if (!x || !W || !y || !d || M < 1 || N < 1 || K < 1) return UPP_E_BADARG;
    if (K > 64 || N > 256 || ldx < K || ldw < K || ldy < N || ldd < N) return UPP_E_RANGE;
    const int Kp = (K + 3) & ~3, RB = (256 / N) * kTR;
    const size_t lds = (size_t)(Kp * N + RB * Kp) * sizeof(float);
    if (lds > 64 * 1024) {
        static std::atomic<bool> raised{false};
        if (!raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(linear_smallk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            raised = true;
        }
    }
    hipLaunchKernelGGL(linear_smallk_kernel, dim3((M + RB - 1) / RB), dim3(256), lds, (hipStream_t)stream, x, ldx, W, ldw, bias, y, ldy, M, N, K, 2, d, ldd);
    return upp_launch_status();
}

extern "C" int upp_transpose_f32(const float *src, long long ld_src, float *dst, long long ld_dst, int rows, int cols, void *stream) {
    if (!src || !dst || rows < 1 || cols < 1) return UPP_E_BADARG;
    if (ld_src < cols || ld_dst < rows) return UPP_E_RANGE;
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, src, ld_src, dst, ld_dst, rows, cols);
    return upp_launch_status();
}
