// dense.hip -- the mini-PointNet patch embedding (reference Encoder,
// models/Point_MAE_unify.py:191-222) as a chain of FP32-MFMA GEMMs for gfx950.
//
//   x (R,3) --Conv1d(3,128)--BN--ReLU--Conv1d(128,256)--> f (R,256)
//   fg = max over the n points of a group                 (R/n,256)
//   cat([fg, f]) --Conv1d(512,512)--BN--ReLU--Conv1d(512,C)--> max over the group --> (R/n, C)
//
// The reference runs this as 4 cuDNN/cuBLAS 1x1 convs + 2 BatchNorms + ReLUs + 2 max-pools +
// a concat, every intermediate (up to R x 512 f32 = 134 MB at B=32) making several HBM round
// trips.  Here one tiled GEMM kernel, C = epilogue(prologue(A) . W^T), carries everything else
// in its prologue / epilogue:
//   * layer 1 (K=3) and its BatchNorm+ReLU are generated on the fly while staging the A tile
//     of layer 2; BN1's batch statistics follow analytically from the 3x3 second-moment
//     matrix of the input points (mean_c = w_c.mu + b_c, var_c = w_c^T Cov w_c), so the
//     (R,128) activation never exists in memory;
//   * the 512->512 layer is split: the "global" half of its input is constant over a group,
//     so it is a (R/n x 256 x 512) GEMM whose result is added per group in the epilogue of the
//     "local" (R x 256 x 512) GEMM -- 25 % fewer FLOPs than the concat formulation;
//   * BatchNorm statistics of that layer come out of the same epilogue as per-slab partial sums
//     (summed in slab order in f64 by a one-block finalize kernel: deterministic);
//     the normalisation + ReLU are applied in the prologue of the last GEMM;
//   * both max-pools are epilogues (a 32x32 MFMA tile holds whole groups of 16 or 32 rows).
// HBM traffic: f written+read once (R*256*4*2), the pre-BN activation written+read once
// (R*512*4*2), nothing else above O(R*3).
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fma chain; the 1e-5 parity bar rules
// out bf16).  Workgroup = 4 waves (2x2), tile 128x128x16, each wave 64x64 = 2x2 MFMA tiles
// (64 accumulator VGPRs); A and W tiles are staged k-major in LDS ([k][row], rows padded) so
// that the one-dword-per-lane MFMA operands are conflict-free ds_read_b32; global loads of
// the next k-step are issued before the MFMAs of the current one (register double buffering,
// two LDS buffers, one barrier per k-step).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDT = BM + 4;  // padded row length of the k-major LDS tiles

enum { PRO_NONE = 0, PRO_BNRELU = 1, PRO_POINT3 = 2 };
enum { EPI_BIAS = 1, EPI_ROWGROUP = 2, EPI_STORE = 4, EPI_STATS = 8, EPI_GMAX = 16 };

struct GemmArgs {
    const float *A; int lda;          // (M,K) row-major (unused for PRO_POINT3)
    const float *W; int ldw;          // (N,K) row-major: C = A . W^T
    const float *bias;                // (N)
    float *C; int ldc;                // (M,N)
    int M, N, K;
    const float *pro_mean, *pro_a, *pro_b;  // per-K: a' = relu((a - mean) * a_k + b_k)
    const float *pts, *w1, *b1;             // PRO_POINT3: a = pts[m] . w1[k] + b1[k]  (K = 128)
    const float *rowgroup; int ldg;         // C[m][:] += rowgroup[m / n][:]
    int n;                                  // rows per group (16 or 32)
    float *stat_part;                       // [2][M/64][N] per-(64-row slab) column sums / sums of squares of C
    float *gmax; int ldgmax;                // gmax[m / n][:] = max over the group of C
};

template <int PRO, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    __shared__ float As[2][BK][LDT];
    __shared__ float Ws[2][BK][LDT];
    __shared__ float pmean[PRO == PRO_NONE ? 1 : 512], pa[PRO == PRO_NONE ? 1 : 512], pb[PRO == PRO_NONE ? 1 : 512];
    __shared__ float pw1[PRO == PRO_POINT3 ? 128 * 3 : 1], pb1[PRO == PRO_POINT3 ? 128 : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;           // wave position in the 2x2 grid
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so the
    // column tiles that share one 128-row A panel would land on different L2s and each fetch the panel again
    // (PMC: FETCH_SIZE = 3-4x the panel bytes).  Give every XCD a contiguous range of logical tiles instead.
    const int nwg = gridDim.x * gridDim.y;
    int lin = blockIdx.y * gridDim.x + blockIdx.x;
    if ((nwg & 7) == 0) lin = (lin & 7) * (nwg >> 3) + (lin >> 3);
    const int bx = lin % gridDim.x, by = lin / gridDim.x;
    const int m0 = by * BM, n0 = bx * BN;
    const int M = g.M, N = g.N, K = g.K;

    if (PRO != PRO_NONE) {
        for (int i = tid; i < K; i += 256) { pmean[i] = g.pro_mean[i]; pa[i] = g.pro_a[i]; pb[i] = g.pro_b[i]; }
        if (PRO == PRO_POINT3) {
            for (int i = tid; i < 128 * 3; i += 256) pw1[i] = g.w1[i];
            for (int i = tid; i < 128; i += 256) pb1[i] = g.b1[i];
        }
        __syncthreads();
    }

    // staging map: thread -> (row = tid / 4 [+64], 4 consecutive k)
    const int srow = tid >> 2, sk = (tid & 3) * 4;
    float px[2][3];
    if (PRO == PRO_POINT3) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + srow + 64 * h;
            const int mc = m < M ? m : M - 1;
            px[h][0] = g.pts[(size_t)mc * 3 + 0]; px[h][1] = g.pts[(size_t)mc * 3 + 1]; px[h][2] = g.pts[(size_t)mc * 3 + 2];
        }
    }

    float4 ra[2], rw[2];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + srow + 64 * h, n = n0 + srow + 64 * h;
            if (PRO != PRO_POINT3)
                ra[h] = (m < M) ? *reinterpret_cast<const float4 *>(g.A + (size_t)m * g.lda + k0 + sk) : make_float4(0, 0, 0, 0);
            rw[h] = (n < N) ? *reinterpret_cast<const float4 *>(g.W + (size_t)n * g.ldw + k0 + sk) : make_float4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&](int buf, int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float v[4];
            if (PRO == PRO_POINT3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = k0 + sk + i;
                    // Conv1d(3,128): bias + w.x accumulated in channel order, then BatchNorm + ReLU
                    float a = pb1[k];
                    a = __builtin_fmaf(px[h][0], pw1[k * 3 + 0], a);
                    a = __builtin_fmaf(px[h][1], pw1[k * 3 + 1], a);
                    a = __builtin_fmaf(px[h][2], pw1[k * 3 + 2], a);
                    v[i] = fmaxf(__builtin_fmaf(a - pmean[k], pa[k], pb[k]), 0.0f);
                }
            } else {
                v[0] = ra[h].x; v[1] = ra[h].y; v[2] = ra[h].z; v[3] = ra[h].w;
                if (PRO == PRO_BNRELU) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = k0 + sk + i;
                        v[i] = fmaxf(__builtin_fmaf(v[i] - pmean[k], pa[k], pb[k]), 0.0f);
                    }
                }
            }
            const int r = srow + 64 * h;
            As[buf][sk + 0][r] = v[0]; As[buf][sk + 1][r] = v[1]; As[buf][sk + 2][r] = v[2]; As[buf][sk + 3][r] = v[3];
            Ws[buf][sk + 0][r] = rw[h].x; Ws[buf][sk + 1][r] = rw[h].y; Ws[buf][sk + 2][r] = rw[h].z; Ws[buf][sk + 3][r] = rw[h].w;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = K / BK;
    load_tile(0);
    store_tile(0, 0);
    __syncthreads();
    const int lr = lane & 31, lk = lane >> 5;
    for (int s = 0; s < nk; ++s) {
        const int buf = s & 1;
        if (s + 1 < nk) load_tile((s + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a0 = As[buf][kk + lk][wm * 64 + lr], a1 = As[buf][kk + lk][wm * 64 + 32 + lr];
            const float b0 = Ws[buf][kk + lk][wn * 64 + lr], b1 = Ws[buf][kk + lk][wn * 64 + 32 + lr];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (s + 1 < nk) {
            store_tile(buf ^ 1, (s + 1) * BK);   // the other buffer: its readers finished before the last barrier
            __syncthreads();
        }
    }

    // ---- epilogue.  acc[i][j][r] = C[row][col], col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + lr;
        const bool cok = col < N;
        const float bias = (EPI & EPI_BIAS) && cok ? g.bias[col] : 0.0f;
        float ssum = 0.0f, ssq = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rbase = m0 + wm * 64 + i * 32;
            float add_lo = bias, add_hi = bias;      // rows 0-15 / 16-31 of the tile (two groups when n = 16)
            if ((EPI & EPI_ROWGROUP) && cok) {
                const int glo = min(rbase, M - 1) / g.n, ghi = min(rbase + 16, M - 1) / g.n;
                add_lo += g.rowgroup[(size_t)glo * g.ldg + col];
                add_hi += g.rowgroup[(size_t)ghi * g.ldg + col];
            }
            float mx_lo = -__builtin_inff(), mx_hi = -__builtin_inff();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const float v = acc[i][j][r] + ((r >> 3) ? add_hi : add_lo);
                const bool ok = cok && row < M;
                if ((EPI & EPI_STORE) && ok) g.C[(size_t)row * g.ldc + col] = v;
                if ((EPI & EPI_STATS) && ok) { ssum += v; ssq = __builtin_fmaf(v, v, ssq); }
                if (EPI & EPI_GMAX) {
                    if (r >> 3) mx_hi = fmaxf(mx_hi, ok ? v : -__builtin_inff());
                    else mx_lo = fmaxf(mx_lo, ok ? v : -__builtin_inff());
                }
            }
            if (EPI & EPI_GMAX) {
                // rows of the other lane half: lane ^ 32 holds the same column
                mx_lo = fmaxf(mx_lo, __shfl_xor(mx_lo, 32));
                mx_hi = fmaxf(mx_hi, __shfl_xor(mx_hi, 32));
                if (cok && lk == 0) {
                    if (g.n == 16) {
                        if (rbase < M) g.gmax[(size_t)(rbase / 16) * g.ldgmax + col] = mx_lo;
                        if (rbase + 16 < M) g.gmax[(size_t)(rbase / 16 + 1) * g.ldgmax + col] = mx_hi;
                    } else if (rbase < M) {   // n == 32
                        g.gmax[(size_t)(rbase / 32) * g.ldgmax + col] = fmaxf(mx_lo, mx_hi);
                    }
                }
            }
        }
        if (EPI & EPI_STATS) {
            // one partial per 64-row slab and column, summed in slab order by bn_finalize: deterministic
            ssum += __shfl_xor(ssum, 32);
            ssq += __shfl_xor(ssq, 32);
            if (cok && lk == 0) {
                const size_t slab = (size_t)by * 2 + wm, slabs = (size_t)gridDim.y * 2;
                g.stat_part[slab * N + col] = ssum;
                g.stat_part[(slabs + slab) * N + col] = ssq;
            }
        }
    }
}

// 9 second moments + 3 first moments of the input rows, f64: per workgroup b, mom[b][0..2] = sum x,y,z; mom[b][3..8] = sum xx, xy, xz, yy,
// yz, zz over its rows.  (Round 4: partial sums per workgroup, added in workgroup order by bn_finalize_kernel<1> -- no f64 atomics any more
// (their order made BN1's statistics depend on scheduling in the last bit), no zero-fill launch in front.)
constexpr int kMomBlocks = 64;
__global__ __launch_bounds__(256) void moments3_kernel(const float *__restrict__ pts, int R, double *__restrict__ mom) {
    double s[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) s[i] = 0.0;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < R; r += gridDim.x * 256) {
        const double x = pts[(size_t)r * 3 + 0], y = pts[(size_t)r * 3 + 1], z = pts[(size_t)r * 3 + 2];
        s[0] += x; s[1] += y; s[2] += z;
        s[3] += x * x; s[4] += x * y; s[5] += x * z; s[6] += y * y; s[7] += y * z; s[8] += z * z;
    }
    __shared__ double red[9][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        double v = s[i];
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[i][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) mom[(size_t)blockIdx.x * 9 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// BatchNorm parameters for a prologue: mean[c], a[c] = gamma * rsqrt(var + eps), and the running-stat update
// (momentum, unbiased variance) of nn.BatchNorm1d in training mode; running stats as-is in eval mode.
// MODE 0: statistics from `slabs` partial per-channel sums / sums of squares (stat_part [2][slabs][C]).
// MODE 1: statistics of the affine map h_c = w_c . x + b_c from the moments of x (w (C,3), b (C)).
template <int MODE>
__global__ __launch_bounds__(256) void bn_finalize_kernel(int C, double count, const float *stat_part, int slabs, const double *mom,
                                                          const float *w, const float *b, const float *gamma, float *rmean, float *rvar,
                                                          float momentum, float eps, int training, float *out_mean, float *out_a,
                                                          const float *beta = nullptr, float *out_shift = nullptr) {
    // 16 channels per workgroup; the 16 thread-rows split the slab partials, then sum in row order (deterministic)
    __shared__ double rs[16][16], rq[16][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tx;
    const bool live = c < C;
    if (MODE == 0 && training) {
        double s = 0.0, q = 0.0;
        if (live)
            for (int i0 = ty; i0 < slabs; i0 += 16 * 8) {       // 16 independent loads in flight (a dependent load is ~1 us)
                float ps[8], pq[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int i = min(i0 + 16 * t, slabs - 1);
                    ps[t] = stat_part[(size_t)i * C + c];
                    pq[t] = stat_part[((size_t)slabs + i) * C + c];
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) if (i0 + 16 * t < slabs) { s += (double)ps[t]; q += (double)pq[t]; }
            }
        rs[ty][tx] = s; rq[ty][tx] = q;
        __syncthreads();
    }
    __shared__ double m9s[9];
    if (MODE == 1 && training) {                            // the moments: thread i sums component i over the workgroups of moments3_kernel, ascending
        if (threadIdx.x < 9) {
            double acc9 = 0.0;
            for (int b0 = 0; b0 < slabs; b0 += 8) {
                double v[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = mom[(size_t)min(b0 + t, slabs - 1) * 9 + threadIdx.x];
#pragma unroll
                for (int t = 0; t < 8; ++t) if (b0 + t < slabs) acc9 += v[t];
            }
            m9s[threadIdx.x] = acc9;
        }
        __syncthreads();
    }
    if (ty != 0 || !live) return;
    double mean, var;
    if (!training) {
        mean = rmean[c]; var = rvar[c];
    } else {
        if (MODE == 0) {
            double s = 0.0, q = 0.0;
            for (int i = 0; i < 16; ++i) { s += rs[i][tx]; q += rq[i][tx]; }
            mean = s / count;
            var = q / count - mean * mean;
        } else {
            const double *m9 = m9s;
            const double mx = m9[0] / count, my = m9[1] / count, mz = m9[2] / count;
            const double cxx = m9[3] / count - mx * mx, cxy = m9[4] / count - mx * my, cxz = m9[5] / count - mx * mz;
            const double cyy = m9[6] / count - my * my, cyz = m9[7] / count - my * mz, czz = m9[8] / count - mz * mz;
            const double w0 = w[c * 3 + 0], w1 = w[c * 3 + 1], w2 = w[c * 3 + 2];
            mean = w0 * mx + w1 * my + w2 * mz + (double)b[c];
            var = w0 * (w0 * cxx + 2 * (w1 * cxy + w2 * cxz)) + w1 * (w1 * cyy + 2 * w2 * cyz) + w2 * w2 * czz;
        }
        if (var < 0) var = 0;
        const double unbiased = count > 1 ? var * count / (count - 1) : var;
        rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
        rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unbiased);
    }
    out_mean[c] = (float)mean;
    out_a[c] = gamma[c] * (float)(1.0 / sqrt(var + (double)eps));
    if (out_shift) out_shift[c] = (float)((double)beta[c] - mean * (double)out_a[c]);      // relu((h - mean) a + beta) = relu(h a + shift)
}

template <int PRO, int EPI>
void launch_gemm(const GemmArgs &g, hipStream_t st) {
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM);
    hipLaunchKernelGGL((gemm_f32_kernel<PRO, EPI>), grid, dim3(256), 0, st, g);
}

struct EmbedWork {
    float *f, *h3, *fg, *hg, *mean1, *a1, *mean3, *a3, *part3, *shift3;
    unsigned char *planes3, *planes4;       // split-bf16 images of W3[:, 256:] and W4 (linear_sb.hip), rebuilt by every call
    double *mom;
};
inline size_t align64(size_t v) { return (v + 63) & ~(size_t)63; }
inline size_t slabs_of(int R) { return (size_t)((R + BM - 1) / BM) * 2; }
inline size_t stat_floats(int R) { return align64((size_t)2 * ((R + 31) / 32) * 512); }     // (per 32-row block: the split-bf16 chain; >= the 64-row slabs of this file's kernel)
constexpr int kMaxEmbedC = 1024;
inline size_t plane_floats(int N, int K) { return align64((size_t)((N + 31) / 32) * ((K + 31) / 32) * 6144 / 4); }
inline EmbedWork carve_embed(float *w, int R, int n) {
    EmbedWork k;
    const size_t G = (size_t)R / n;
    size_t o = 0;
    k.mom = reinterpret_cast<double *>(w + o); o += align64((size_t)kMomBlocks * 9 * 2);            // [workgroup][9] doubles
    k.part3 = w + o; o += stat_floats(R);
    k.mean1 = w + o; o += 128;  k.a1 = w + o; o += 128;
    k.mean3 = w + o; o += 512;  k.a3 = w + o; o += 512;  k.shift3 = w + o; o += 512;
    k.planes3 = reinterpret_cast<unsigned char *>(w + o); o += plane_floats(512, 256);
    k.planes4 = reinterpret_cast<unsigned char *>(w + o); o += plane_floats(kMaxEmbedC, 512);
    k.fg = w + o; o += align64(G * 256);
    k.hg = w + o; o += align64(G * 512);
    k.f = w + o; o += align64((size_t)R * 256);
    k.h3 = w + o; o += align64((size_t)R * 512);
    return k;
}

}  // namespace

extern "C" int upp_linear_sb_prep_batched(const float *const *W, const long long *ldw, const int *N, const int *K, const int *transposed,
                                          void *const *planes, int count, void *stream);
__attribute__((visibility("hidden"))) int upp_detail_linear_sb_chain(const float *A, long long lda, const void *planes, const float *bias, int bias_shift,
                                                                     float *C, long long ldc, int M, int N, int K, const float *pro_scale,
                                                                     const float *pro_shift, float *gmax, int ldgmax, int gshift, float *stat_part,
                                                                     void *stream);

extern "C" long long upp_patch_embed_work_floats(int R, int n) {
    if (R <= 0 || n <= 0) return 0;
    const size_t G = (size_t)R / n;
    return (long long)(align64((size_t)kMomBlocks * 9 * 2) + stat_floats(R) + 256 + 1536 + plane_floats(512, 256) + plane_floats(kMaxEmbedC, 512) + align64(G * 256) + align64(G * 512) +
                       align64((size_t)R * 256) + align64((size_t)R * 512));
}

extern "C" int upp_patch_embed_fwd(const float *pts, int R, int n, const float *w1, const float *b1, const float *bn1_gamma,
                                   const float *bn1_beta, float *bn1_rmean, float *bn1_rvar, const float *w2, const float *b2,
                                   const float *w3, const float *b3, const float *bn3_gamma, const float *bn3_beta,
                                   float *bn3_rmean, float *bn3_rvar, const float *w4, const float *b4, int C, float momentum,
                                   float eps, int training, float *work, float *out, void *stream) {
    if (!pts || !w1 || !b1 || !bn1_gamma || !bn1_beta || !bn1_rmean || !bn1_rvar || !w2 || !b2 || !w3 || !b3 || !bn3_gamma ||
        !bn3_beta || !bn3_rmean || !bn3_rvar || !w4 || !b4 || !work || !out || R < 1 || C < 1)
        return UPP_E_BADARG;
    if ((n != 16 && n != 32) || R % n != 0 || C % 4 != 0) return UPP_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    const EmbedWork k = carve_embed(work, R, n);
    const int G = R / n;

    // BN1 statistics from the moments of the input rows
    int blocks = (R + 255) / 256; if (blocks > kMomBlocks) blocks = kMomBlocks;
    if (training) hipLaunchKernelGGL(moments3_kernel, dim3(blocks), dim3(256), 0, st, pts, R, k.mom);
    hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3(8), dim3(256), 0, st, 128, (double)R, nullptr, blocks, k.mom, w1, b1,
                       bn1_gamma, bn1_rmean, bn1_rvar, momentum, eps, training, k.mean1, k.a1);

    GemmArgs g{};
    // f = relu(bn1(conv1(x))) . W2^T + b2 ; fg = group max
    g = GemmArgs{};
    g.W = w2; g.ldw = 128; g.bias = b2; g.C = k.f; g.ldc = 256; g.M = R; g.N = 256; g.K = 128;
    g.pro_mean = k.mean1; g.pro_a = k.a1; g.pro_b = bn1_beta; g.pts = pts; g.w1 = w1; g.b1 = b1;
    g.n = n; g.gmax = k.fg; g.ldgmax = 256;
    launch_gemm<PRO_POINT3, EPI_BIAS | EPI_STORE | EPI_GMAX>(g, st);
    // hg = fg . W3[:, :256]^T + b3      (once per group)
    g = GemmArgs{};
    g.A = k.fg; g.lda = 256; g.W = w3; g.ldw = 512; g.bias = b3; g.C = k.hg; g.ldc = 512; g.M = G; g.N = 512; g.K = 256;
    // (a G x 512 x 256 product: 64 tiles of this file's 128 x 128 kernel are 16 latency-bound k-steps on a quarter of the CUs, 31 us at
    // G = 2048; upp_linear_f32 picks a one-round decomposition of the same product: 8 us)
    {
        const int rc = upp_linear_f32(k.fg, 256, w3, 512, b3, k.hg, 512, nullptr, 0, G, 512, 256, /*LEPI_BIAS*/ 1, 0, stream);
        if (rc) return rc;
    }
    // h3 = f . W3[:, 256:]^T + hg[group] ; BN3 statistics ;  out = group max( relu(bn3(h3)) . W4^T + b4 )
    const bool split_bf16 = upp_option(UPP_OPT_EMBED_SPLIT_BF16) != 0;       // (read per call: tests switch between the two chains inside one process)
    if (split_bf16 && C <= kMaxEmbedC) {
        // the two large products on the BF16 matrix pipe (linear_sb.hip: f32 operands split into three bf16 terms, six products; error of an
        // f32 GEMM), with this chain's fusions as that kernel's prologue / epilogues: 100 -> 180-200 TFLOP/s on 65,536 rows
        const float *ws[2] = {w3 + 256, w4};
        const long long ldws[2] = {512, 512};
        const int ns[2] = {512, C}, ks[2] = {256, 512}, trs[2] = {0, 0};
        void *pls[2] = {k.planes3, k.planes4};
        int rc = upp_linear_sb_prep_batched(ws, ldws, ns, ks, trs, pls, 2, stream);
        if (rc) return rc;
        const int gshift = n == 16 ? 4 : 5;
        rc = upp_detail_linear_sb_chain(k.f, 256, k.planes3, k.hg, gshift, k.h3, 512, R, 512, 256, nullptr, nullptr, nullptr, 0, 0,
                                        training ? k.part3 : nullptr, stream);
        if (rc) return rc;
        hipLaunchKernelGGL((bn_finalize_kernel<0>), dim3(32), dim3(256), 0, st, 512, (double)R, k.part3, (R + 31) / 32, nullptr,
                           nullptr, nullptr, bn3_gamma, bn3_rmean, bn3_rvar, momentum, eps, training, k.mean3, k.a3, bn3_beta, k.shift3);
        rc = upp_detail_linear_sb_chain(k.h3, 512, k.planes4, b4, 0, nullptr, 0, R, C, 512, k.a3, k.shift3, out, C, gshift, nullptr, stream);
        if (rc) return rc;
        return upp_launch_status();
    }
    g = GemmArgs{};
    g.A = k.f; g.lda = 256; g.W = w3 + 256; g.ldw = 512; g.C = k.h3; g.ldc = 512; g.M = R; g.N = 512; g.K = 256;
    g.rowgroup = k.hg; g.ldg = 512; g.n = n; g.stat_part = k.part3;
    if (training) launch_gemm<PRO_NONE, EPI_ROWGROUP | EPI_STORE | EPI_STATS>(g, st);
    else launch_gemm<PRO_NONE, EPI_ROWGROUP | EPI_STORE>(g, st);
    hipLaunchKernelGGL((bn_finalize_kernel<0>), dim3(32), dim3(256), 0, st, 512, (double)R, k.part3, (int)slabs_of(R), nullptr,
                       nullptr, nullptr, bn3_gamma, bn3_rmean, bn3_rvar, momentum, eps, training, k.mean3, k.a3);
    // out = group max( relu(bn3(h3)) . W4^T + b4 )
    g = GemmArgs{};
    g.A = k.h3; g.lda = 512; g.W = w4; g.ldw = 512; g.bias = b4; g.M = R; g.N = C; g.K = 512;
    g.pro_mean = k.mean3; g.pro_a = k.a3; g.pro_b = bn3_beta; g.n = n; g.gmax = out; g.ldgmax = C;
    launch_gemm<PRO_BNRELU, EPI_BIAS | EPI_GMAX>(g, st);
    return upp_launch_status();
}
