/*
 * upp_oracle.c -- CPU restatement of the UPP hot-path operators.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 *
 * Every function restates, in plain scalar C, the algorithm of one reference
 * kernel, with the same arithmetic order, so that integer outputs (FPS indices,
 * kNN neighbour lists, Chamfer argmins) can be compared bit-for-bit and float
 * outputs to a stated tolerance.
 *
 * PARITY PINNING
 *  - Chamfer, EMD: restated line by line from the in-tree CUDA sources
 *    (extensions/chamfer_dist/chamfer.cu, extensions/emd/cuda/emd_kernel.cu).
 *    Pinned by the reference's own known answers: extensions/emd/test_emd_loss.py
 *    (cost 0.71 / analytic grads) and extensions/chamfer_dist/test.py (gradcheck),
 *    see tests/test_oracle_golden.py.
 *  - FPS, gather, kNN: the algorithm lives in third-party packages that are NOT
 *    vendored under /root/reference:
 *        pointnet2_ops 3.0.0 (erikwijmans/Pointnet2_PyTorch, git master, unpinned;
 *                             reference README.md:73)  sampling_gpu.cu
 *        KNN_CUDA 0.2        (unlimblue/KNN_CUDA wheel; reference README.md:76)  knn.cu
 *    Their published algorithms are restated here.  The reference holds no test
 *    or golden vector for the operators themselves, but its tree states the same
 *    algorithms in plain numpy / torch (datasets/ModelNetDataset.py:29-49
 *    farthest_point_sample, models/Transformer_utils.py:17-29 knn_point); their
 *    outputs, produced here by oracle/gen_golden_ops.py, are tests/golden/ref_ops.npz
 *    and pin the FPS index sequence and the kNN neighbour sets on clouds without
 *    exact ties and without points inside |p|^2 <= 1e-3.  Still "PARITY UNPINNED":
 *    the CUDA kernels' tie order, the 1e-3 skip rule of FPS, and the order inside a
 *    KNN_CUDA neighbour list -- anchored only on upstream memory and the reference
 *    call sites utils/misc.py:18-19 and models/Point_MAE_unify.py:56,69.
 *
 * FLOATING-POINT CONTRACTION.  The CUDA sources are compiled by nvcc with
 * -fmad=true, so  a*a + b*b + c*c  is contracted.  nvcc (NVPTX DAG combiner,
 * fadd(fmul x y, z) -> fma x y z, left operand first) produces
 *     t = b*b ; t = fma(a,a,t) ; t = fma(c,c,t)
 * for the left-associated three-term sum, and fma(t,t,acc) for a running
 * "acc += t*t" loop.  Those two forms are written out explicitly below with
 * fmaf(); this file must be compiled with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* (x*x) + (y*y) + (z*z) as nvcc contracts it. */
static inline float sumsq3(float x, float y, float z) {
    float t = y * y;
    t = fmaf(x, x, t);
    t = fmaf(z, z, t);
    return t;
}

/* ------------------------------------------------------------------------ */
/* FPS  -- pointnet2_ops sampling_gpu.cu furthest_point_sampling_kernel      */
/*         (call site: reference utils/misc.py:18)                           */
/* ------------------------------------------------------------------------ */

/* pointnet2_ops cuda_utils.h opt_n_threads(): largest power of two <= work
 * size, computed through double log, clamped to [1, 512]. */
int oracle_fps_block_size(int n) {
    const int pow_2 = (int)(log((double)n) / log(2.0));
    int t = 1 << pow_2;
    if (t > 512) t = 512;
    if (t < 1) t = 1;
    return t;
}

/* xyz (B,N,3) f32, idx (B,M) int32.  temp (B,N) scratch may be NULL.
 * One CUDA block of T threads per cloud is simulated literally: per-thread
 * strided scan with strict '>' (lowest k of a thread wins), then the
 * shared-memory tree  __update(): dists_i[i1] = v2 > v1 ? i2 : i1. */
int oracle_fps(const float *xyz, int B, int N, int M, int32_t *idx, float *temp_in) {
    if (M <= 0) return 0;
    const int T = oracle_fps_block_size(N);
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        const float *p = xyz + (size_t)b * N * 3;
        int32_t *out = idx + (size_t)b * M;
        float *temp = temp_in ? temp_in + (size_t)b * N : (float *)malloc(sizeof(float) * N);
        float *dists = (float *)malloc(sizeof(float) * T);
        int *dists_i = (int *)malloc(sizeof(int) * T);
        for (int k = 0; k < N; ++k) temp[k] = 1e10f;
        int old = 0;
        out[0] = 0;
        for (int j = 1; j < M; ++j) {
            const float x1 = p[old * 3 + 0], y1 = p[old * 3 + 1], z1 = p[old * 3 + 2];
            for (int tid = 0; tid < T; ++tid) {
                int besti = 0;
                float best = -1.0f;
                for (int k = tid; k < N; k += T) {
                    const float x2 = p[k * 3 + 0], y2 = p[k * 3 + 1], z2 = p[k * 3 + 2];
                    const float mag = sumsq3(x2, y2, z2);
                    if ((double)mag <= 1e-3) continue; /* literal 1e-3 is a double */
                    const float d = sumsq3(x2 - x1, y2 - y1, z2 - z1);
                    const float d2 = d < temp[k] ? d : temp[k]; /* min(d, temp[k]) */
                    temp[k] = d2;
                    besti = d2 > best ? k : besti;
                    best = d2 > best ? d2 : best;
                }
                dists[tid] = best;
                dists_i[tid] = besti;
            }
            for (int s = T / 2; s >= 1; s >>= 1) {
                for (int tid = 0; tid < s; ++tid) {
                    const float v1 = dists[tid], v2 = dists[tid + s];
                    const int i1 = dists_i[tid], i2 = dists_i[tid + s];
                    dists[tid] = v1 > v2 ? v1 : v2; /* max(v1, v2) */
                    dists_i[tid] = v2 > v1 ? i2 : i1;
                }
            }
            old = dists_i[0];
            out[j] = old;
        }
        free(dists);
        free(dists_i);
        if (!temp_in) free(temp);
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* gather_operation -- pointnet2_ops sampling_gpu.cu gather_points_kernel    */
/*         (call site: reference utils/misc.py:19)                           */
/* ------------------------------------------------------------------------ */
/* feat (B,C,N), idx (B,M) -> out (B,C,M) */
void oracle_gather(const float *feat, const int32_t *idx, float *out, int B, int C, int N, int M) {
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int j = 0; j < M; ++j)
                out[((size_t)b * C + c) * M + j] = feat[((size_t)b * C + c) * N + idx[(size_t)b * M + j]];
}

/* grad_out (B,C,M), idx (B,M) -> grad_feat (B,C,N), accumulated in j order
 * (the CUDA kernel uses atomicAdd: order unspecified). */
void oracle_gather_grad(const float *grad_out, const int32_t *idx, float *grad_feat, int B, int C, int N, int M) {
    memset(grad_feat, 0, sizeof(float) * (size_t)B * C * N);
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int j = 0; j < M; ++j)
                grad_feat[((size_t)b * C + c) * N + idx[(size_t)b * M + j]] += grad_out[((size_t)b * C + c) * M + j];
}

/* ------------------------------------------------------------------------ */
/* kNN -- KNN_CUDA 0.2 knn.cu: cuComputeDistanceGlobal + cuInsertionSort +   */
/*        cuParallelSqrt  (call site: reference models/Point_MAE_unify.py:69,*/
/*        KNN(k, transpose_mode=True))                                       */
/* ------------------------------------------------------------------------ */
/* ref (B,N,3), query (B,Q,3) -> dist (B,Q,K) f32 (Euclidean, may be NULL),
 * idx (B,Q,K) int64 0-based.  Requires 1 <= K <= N. */
int oracle_knn(const float *ref, const float *query, int B, int N, int Q, int K, float *dist, int64_t *idx) {
    if (K < 1 || K > N) return -1;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        float *col = (float *)malloc(sizeof(float) * N);
        int64_t *ind = (int64_t *)malloc(sizeof(int64_t) * N);
        for (int q = 0; q < Q; ++q) {
            const float *qp = query + ((size_t)b * Q + q) * 3;
            /* cuComputeDistanceGlobal: ssd = 0; for d: tmp = A - B; ssd += tmp*tmp
             * (A = ref, B = query; dims padded to 16 with zeros, which add +0). */
            for (int r = 0; r < N; ++r) {
                const float *rp = ref + ((size_t)b * N + r) * 3;
                float ssd = 0.0f;
                for (int d = 0; d < 3; ++d) {
                    const float tmp = rp[d] - qp[d];
                    ssd = fmaf(tmp, tmp, ssd);
                }
                col[r] = ssd;
            }
            /* cuInsertionSort, one query column, 1-based indices. */
            float max_dist = col[0];
            ind[0] = 1;
            for (int l = 1; l < K; ++l) {
                const float curr = col[l];
                if (curr < max_dist) {
                    int i = l - 1;
                    for (int a = 0; a < l - 1; ++a)
                        if (col[a] > curr) { i = a; break; }
                    for (int j = l; j > i; --j) { col[j] = col[j - 1]; ind[j] = ind[j - 1]; }
                    col[i] = curr;
                    ind[i] = l + 1;
                } else {
                    ind[l] = l + 1;
                }
                max_dist = col[l];
            }
            for (int l = K; l < N; ++l) {
                const float curr = col[l];
                if (curr < max_dist) {
                    int i = K - 1;
                    for (int a = 0; a < K - 1; ++a)
                        if (col[a] > curr) { i = a; break; }
                    for (int j = K - 1; j > i; --j) { col[j] = col[j - 1]; ind[j] = ind[j - 1]; }
                    col[i] = curr;
                    ind[i] = l + 1;
                    max_dist = col[K - 1];
                }
            }
            for (int j = 0; j < K; ++j) {
                if (dist) dist[((size_t)b * Q + q) * K + j] = sqrtf(col[j]); /* cuParallelSqrt */
                idx[((size_t)b * Q + q) * K + j] = ind[j] - 1;               /* wrapper: i -= 1 */
            }
        }
        free(col);
        free(ind);
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Group gather + centre subtraction -- reference models/Point_MAE_unify.py  */
/* :73-88 (flat gather xyz.view(B*N,3)[idx] then neighborhood - center)      */
/* ------------------------------------------------------------------------ */
/* xyz (B,N,3), center (B,G,3), idx (B,G,K) int64 in [0,N) -> out (B,G,K,3) */
void oracle_group(const float *xyz, const float *center, const int64_t *idx, float *out, int B, int N, int G, int K) {
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g)
            for (int k = 0; k < K; ++k) {
                const int64_t r = idx[((size_t)b * G + g) * K + k];
                for (int c = 0; c < 3; ++c)
                    out[(((size_t)b * G + g) * K + k) * 3 + c] =
                        xyz[((size_t)b * N + r) * 3 + c] - center[((size_t)b * G + g) * 3 + c];
            }
}

/* ------------------------------------------------------------------------ */
/* Chamfer -- reference extensions/chamfer_dist/chamfer.cu                   */
/* ------------------------------------------------------------------------ */
/* One direction (chamfer.cu:15-145): for every point j of xyz1 (B,n,3) the
 * min squared distance to xyz2 (B,m,3) and its argmin.  Tiles of 512 as in
 * the kernel: strict '<' inside a tile (:47,57), strict '>' across tiles
 * (:137) ==> lowest index among equal minima. */
void oracle_chamfer_dir(const float *xyz1, const float *xyz2, int B, int n, int m, float *dist, int32_t *idx) {
    const int batch = 512;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < B; ++i) {
        for (int k2 = 0; k2 < m; k2 += batch) {
            const int end_k = (m < k2 + batch ? m : k2 + batch) - k2;
            const float *buf = xyz2 + ((size_t)i * m + k2) * 3;
            for (int j = 0; j < n; ++j) {
                const float x1 = xyz1[((size_t)i * n + j) * 3 + 0];
                const float y1 = xyz1[((size_t)i * n + j) * 3 + 1];
                const float z1 = xyz1[((size_t)i * n + j) * 3 + 2];
                float best = 0.0f;
                int besti = 0;
                for (int k = 0; k < end_k; ++k) {
                    const float x2 = buf[k * 3 + 0] - x1;
                    const float y2 = buf[k * 3 + 1] - y1;
                    const float z2 = buf[k * 3 + 2] - z1;
                    const float d = sumsq3(x2, y2, z2); /* x2*x2 + y2*y2 + z2*z2, :45 */
                    if (k == 0 || d < best) { best = d; besti = k + k2; }
                }
                if (k2 == 0 || dist[(size_t)i * n + j] > best) {
                    dist[(size_t)i * n + j] = best;
                    idx[(size_t)i * n + j] = besti;
                }
            }
        }
    }
}

/* chamfer.cu:147-171 -- both directions. */
void oracle_chamfer_fwd(const float *xyz1, const float *xyz2, int B, int n, int m,
                        float *dist1, float *dist2, int32_t *idx1, int32_t *idx2) {
    oracle_chamfer_dir(xyz1, xyz2, B, n, m, dist1, idx1);
    oracle_chamfer_dir(xyz2, xyz1, B, m, n, dist2, idx2);
}

/* One launch of chamfer_dist_grad_kernel (chamfer.cu:173-201), sequential in
 * j (the CUDA kernel's atomicAdd order is unspecified).  Accumulates. */
static void chamfer_grad_dir(const float *xyz1, const float *xyz2, int B, int n, int m,
                             const float *grad_dist1, const int32_t *idx1, float *g1, float *g2) {
    for (int i = 0; i < B; ++i)
        for (int j = 0; j < n; ++j) {
            const float x1 = xyz1[((size_t)i * n + j) * 3 + 0];
            const float y1 = xyz1[((size_t)i * n + j) * 3 + 1];
            const float z1 = xyz1[((size_t)i * n + j) * 3 + 2];
            const int j2 = idx1[(size_t)i * n + j];
            const float x2 = xyz2[((size_t)i * m + j2) * 3 + 0];
            const float y2 = xyz2[((size_t)i * m + j2) * 3 + 1];
            const float z2 = xyz2[((size_t)i * m + j2) * 3 + 2];
            const float g = grad_dist1[(size_t)i * n + j] * 2;
            g1[((size_t)i * n + j) * 3 + 0] += g * (x1 - x2);
            g1[((size_t)i * n + j) * 3 + 1] += g * (y1 - y2);
            g1[((size_t)i * n + j) * 3 + 2] += g * (z1 - z2);
            g2[((size_t)i * m + j2) * 3 + 0] += -(g * (x1 - x2));
            g2[((size_t)i * m + j2) * 3 + 1] += -(g * (y1 - y2));
            g2[((size_t)i * m + j2) * 3 + 2] += -(g * (z1 - z2));
        }
}

/* chamfer.cu:203-229 */
void oracle_chamfer_bwd(const float *xyz1, const float *xyz2, const int32_t *idx1, const int32_t *idx2,
                        const float *gd1, const float *gd2, int B, int n, int m, float *g1, float *g2) {
    memset(g1, 0, sizeof(float) * (size_t)B * n * 3);
    memset(g2, 0, sizeof(float) * (size_t)B * m * 3);
    chamfer_grad_dir(xyz1, xyz2, B, n, m, gd1, idx1, g1, g2);
    chamfer_grad_dir(xyz2, xyz1, B, m, n, gd2, idx2, g2, g1);
}

/* ------------------------------------------------------------------------ */
/* EMD -- reference extensions/emd/cuda/emd_kernel.cu                        */
/* ------------------------------------------------------------------------ */
/* approxmatch (emd_kernel.cu:24-157).  xyz1 (B,n,3), xyz2 (B,m,3) ->
 * match (B,m,n).  The per-thread sums run sequentially over the other cloud
 * in index order, so the summation order is independent of the launch shape.
 * __expf is the CUDA fast intrinsic, ex2.approx(x * log2(e)) with the product rounded to f32; it is restated as
 * exp2f(x * 1.44269504f) -- the same product rounding, a correctly rounded 2^y where the hardware (NVIDIA's ex2.approx, gfx950's
 * v_exp_f32) is within ~1-2 ulp -- so the remaining difference to a device kernel is the exponential unit's own approximation,
 * not the argument's (|x| log2(e) 2^-24 relative: 1e-5 at x = -100); EMD parity stays tolerance-based (see tests). */
static inline float oracle_fast_expf(float x) { return exp2f(x * 1.44269504088896340736f); }
void oracle_emd_approxmatch(const float *xyz1, const float *xyz2, int B, int n, int m, float *match) {
    float multiL, multiR;
    if (n >= m) { multiL = 1; multiR = (float)(n / m); }   /* integer division, :28-34 */
    else        { multiL = (float)(m / n); multiR = 1; }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < B; ++i) {
        float *remainL = (float *)malloc(sizeof(float) * n);
        float *remainR = (float *)malloc(sizeof(float) * m);
        float *ratioL = (float *)malloc(sizeof(float) * n);
        float *ratioR = (float *)malloc(sizeof(float) * m);
        float *mt = match + (size_t)i * n * m;
        const float *p1 = xyz1 + (size_t)i * n * 3, *p2 = xyz2 + (size_t)i * m * 3;
        for (size_t j = 0; j < (size_t)n * m; ++j) mt[j] = 0;
        for (int j = 0; j < n; ++j) remainL[j] = multiL;
        for (int j = 0; j < m; ++j) remainR[j] = multiR;
        for (int j = 7; j >= -2; --j) {
            float level = -powf(4.0f, (float)j);
            if (j == -2) level = 0;
            for (int k = 0; k < n; ++k) {               /* pass 1, :50-83 */
                const float x1 = p1[k * 3 + 0], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2];
                float suml = 1e-9f;
                for (int l = 0; l < m; ++l) {
                    const float x2 = p2[l * 3 + 0], y2 = p2[l * 3 + 1], z2 = p2[l * 3 + 2];
                    const float d = level * sumsq3(x2 - x1, y2 - y1, z2 - z1);
                    suml = fmaf(oracle_fast_expf(d), remainR[l], suml); /* w = e*remainR; suml += w (contracted) */
                }
                ratioL[k] = remainL[k] / suml;
            }
            for (int l = 0; l < m; ++l) {               /* pass 2, :85-118 */
                const float x2 = p2[l * 3 + 0], y2 = p2[l * 3 + 1], z2 = p2[l * 3 + 2];
                float sumr = 0;
                for (int k = 0; k < n; ++k) {
                    const float x1 = p1[k * 3 + 0], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2];
                    sumr = fmaf(oracle_fast_expf(level * sumsq3(x2 - x1, y2 - y1, z2 - z1)), ratioL[k], sumr);
                }
                sumr *= remainR[l];
                const float consumption = fminf(remainR[l] / (sumr + 1e-9f), 1.0f);
                ratioR[l] = consumption * remainR[l];
                remainR[l] = fmaxf(0.0f, remainR[l] - sumr);
            }
            for (int k = 0; k < n; ++k) {               /* pass 3, :120-153 */
                const float x1 = p1[k * 3 + 0], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2];
                const float rl = ratioL[k];
                float suml = 0;
                for (int l = 0; l < m; ++l) {
                    const float x2 = p2[l * 3 + 0], y2 = p2[l * 3 + 1], z2 = p2[l * 3 + 2];
                    /* w = e*rl*ratioR; match += w; suml += w -- NVPTX fuses aggressively,
                     * so both adds become fma(e*rl, ratioR, acc). */
                    const float er = oracle_fast_expf(level * sumsq3(x2 - x1, y2 - y1, z2 - z1)) * rl;
                    mt[(size_t)l * n + k] = fmaf(er, ratioR[l], mt[(size_t)l * n + k]);
                    suml = fmaf(er, ratioR[l], suml);
                }
                remainL[k] = fmaxf(0.0f, remainL[k] - suml);
            }
        }
        free(remainL); free(remainR); free(ratioL); free(ratioR);
    }
}

/* matchcost (emd_kernel.cu:199-242): cost[b] = sum d2(k,l) * match[l,k],
 * 512 virtual threads (k = tid, tid+512, ...) then the kernel's shared-memory
 * tree ((tid & j) == 0 && tid + j < 512). */
void oracle_emd_matchcost(const float *xyz1, const float *xyz2, const float *match, int B, int n, int m, float *cost) {
    const int T = 512;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < B; ++i) {
        float allsum[512];
        const float *p1 = xyz1 + (size_t)i * n * 3, *p2 = xyz2 + (size_t)i * m * 3;
        const float *mt = match + (size_t)i * n * m;
        for (int tid = 0; tid < T; ++tid) {
            float subsum = 0;
            for (int k = tid; k < n; k += T) {
                const float x1 = p1[k * 3 + 0], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2];
                for (int l = 0; l < m; ++l) {
                    const float x2 = p2[l * 3 + 0], y2 = p2[l * 3 + 1], z2 = p2[l * 3 + 2];
                    const float d = sumsq3(x2 - x1, y2 - y1, z2 - z1);
                    subsum = fmaf(d, mt[(size_t)l * n + k], subsum); /* subsum += d*match, contracted */
                }
            }
            allsum[tid] = subsum;
        }
        for (int j = 1; j < T; j <<= 1)
            for (int tid = 0; tid < T; ++tid)
                if ((tid & j) == 0 && tid + j < T) allsum[tid] += allsum[tid + j];
        cost[i] = allsum[0];
    }
}

/* matchcostgrad1 (:332-354) and matchcostgrad2 (:285-326).  match is a
 * constant w.r.t. the clouds.  grad2 uses 256 virtual threads + tree. */
void oracle_emd_matchcost_grad(const float *grad_cost, const float *xyz1, const float *xyz2, const float *match,
                               int B, int n, int m, float *grad1, float *grad2) {
    const int T = 256;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < B; ++i) {
        const float *p1 = xyz1 + (size_t)i * n * 3, *p2 = xyz2 + (size_t)i * m * 3;
        const float *mt = match + (size_t)i * n * m;
        for (int l = 0; l < n; ++l) {
            const float x1 = p1[l * 3 + 0], y1 = p1[l * 3 + 1], z1 = p1[l * 3 + 2];
            float dx = 0, dy = 0, dz = 0;
            for (int k = 0; k < m; ++k) {
                const float d = mt[(size_t)k * n + l] * 2;
                dx = fmaf(x1 - p2[k * 3 + 0], d, dx);
                dy = fmaf(y1 - p2[k * 3 + 1], d, dy);
                dz = fmaf(z1 - p2[k * 3 + 2], d, dz);
            }
            grad1[((size_t)i * n + l) * 3 + 0] = dx * grad_cost[i];
            grad1[((size_t)i * n + l) * 3 + 1] = dy * grad_cost[i];
            grad1[((size_t)i * n + l) * 3 + 2] = dz * grad_cost[i];
        }
        for (int k = 0; k < m; ++k) {
            float sg[256 * 3];
            const float x2 = p2[k * 3 + 0], y2 = p2[k * 3 + 1], z2 = p2[k * 3 + 2];
            for (int tid = 0; tid < T; ++tid) {
                float sx = 0, sy = 0, sz = 0;
                for (int j = tid; j < n; j += T) {
                    const float d = mt[(size_t)k * n + j] * 2;
                    sx = fmaf(x2 - p1[j * 3 + 0], d, sx);
                    sy = fmaf(y2 - p1[j * 3 + 1], d, sy);
                    sz = fmaf(z2 - p1[j * 3 + 2], d, sz);
                }
                sg[tid * 3 + 0] = sx; sg[tid * 3 + 1] = sy; sg[tid * 3 + 2] = sz;
            }
            for (int j = 1; j < T; j <<= 1)
                for (int tid = 0; tid < T; ++tid)
                    if ((tid & j) == 0 && tid + j < T) {
                        sg[tid * 3 + 0] += sg[(tid + j) * 3 + 0];
                        sg[tid * 3 + 1] += sg[(tid + j) * 3 + 1];
                        sg[tid * 3 + 2] += sg[(tid + j) * 3 + 2];
                    }
            grad2[((size_t)i * m + k) * 3 + 0] = sg[0] * grad_cost[i];
            grad2[((size_t)i * m + k) * 3 + 1] = sg[1] * grad_cost[i];
            grad2[((size_t)i * m + k) * 3 + 2] = sg[2] * grad_cost[i];
        }
    }
}

/* ---- Linear layers of the Transformer blocks ---------------------------------------------------
 * Reference: nn.Linear -> F.linear (models/Point_MAE_pretask_dev.py:153-196: Mlp.fc1/fc2, Attention.qkv/proj), i.e.
 * C = A . W^T (+ bias); the reference's GEMM library fixes no summation order.  This is the order of the product under test
 * (upp_linear_f32, csrc/linear.hip), written as plain fmaf chains so that the kernel can be checked BIT FOR BIT:
 *   - v_mfma_f32_32x32x2_f32 adds its two products to the accumulator as fma(a1, b1, fma(a0, b0, c)), lower lane half (k index 0)
 *     first, one rounding per product;
 *   - inside a group of 32 k values the kernel feeds k = 8 i + j to the lower and 8 i + 4 + j to the upper lane half for
 *     i = 0..3, j = 0..3 (16-byte operand granules);
 *   - the contraction is cut into stages of 32 KS KC values; wave group g of KS owns sub-chunks g KC .. g KC + KC - 1 of every
 *     stage and keeps its own accumulator; the KS accumulators are summed in group order starting from 0.0f;
 *   - epilogues: 0 none, 1 + bias[n], 4 * aux[m][n]  (the GELU epilogues use the hardware's v_exp_f32 / v_rcp_f32 approximations
 *     and are checked against torch within a tolerance instead).  */
void oracle_linear_f32(const float *A, const float *W, const float *bias, const float *aux, float *C, int M, int N, int K,
                       int ks, int kc, int epilogue) {
    const int stage = 32 * ks * kc, nst = K / stage;
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        for (int n = 0; n < N; ++n) {
            const float *a = A + (size_t)m * K, *w = W + (size_t)n * K;
            float total = 0.0f;
            for (int g = 0; g < ks; ++g) {
                float acc = 0.0f;
                for (int c = 0; c < nst; ++c)
                    for (int q = 0; q < kc; ++q) {
                        const int k0 = c * stage + (g * kc + q) * 32;
                        for (int i = 0; i < 4; ++i)
                            for (int j = 0; j < 4; ++j) {
                                acc = fmaf(a[k0 + 8 * i + j], w[k0 + 8 * i + j], acc);
                                acc = fmaf(a[k0 + 8 * i + 4 + j], w[k0 + 8 * i + 4 + j], acc);
                            }
                    }
                total = ks > 1 ? total + acc : acc;
            }
            if (epilogue == 1) total += bias[n];
            if (epilogue == 4) total *= aux[(size_t)m * N + n];
            C[(size_t)m * N + n] = total;
        }
    }
}

/* upp_linear_wgrad_grouped_f32 (csrc/linear_rt.hip): partial weight gradients of a Linear, P (splits, N, K) with
 * P[s][n][k] = sum over the rows m of run s (rows s*rows .. min(M, (s+1)*rows) - 1) of G[m][n] * X[m][k] as ONE ascending-row
 * fmaf chain from 0 (the FP32 MFMA adds its two products as fma(g1, x1, fma(g0, x0, c)); rows beyond the run contribute
 * fma(0, x, c) = c).  Reference: AddmmBackward's G^T . X in torch -- no summation order is specified there. */
void oracle_linear_wgrad(const float *G, const float *X, float *P, int M, int N, int K, int rows) {
    const int splits = (M + rows - 1) / rows;
#pragma omp parallel for schedule(static) collapse(2)
    for (int s = 0; s < splits; ++s)
        for (int n = 0; n < N; ++n) {
            const int m0 = s * rows, m1 = m0 + rows < M ? m0 + rows : M;
            float *out = P + ((size_t)s * N + n) * K;
            for (int k = 0; k < K; ++k) out[k] = 0.0f;
            for (int m = m0; m < m1; ++m) {
                const float gv = G[(size_t)m * N + n];
                const float *x = X + (size_t)m * K;
                for (int k = 0; k < K; ++k) out[k] = fmaf(gv, x[k], out[k]);
            }
        }
}

/* upp_linear_smallk_f32 (csrc/smallk.hip): one ascending-k fmaf chain per output, k padded to a multiple of 4 with zeros,
 * then + bias, then act (0 none, 1 ReLU; GELU goes through erff and is compared within a tolerance instead).
 * Stands for torch.nn.functional.linear at the reference's K = 3 / K = 59 layers (models/Point_MAE_pretask_dev.py:395-399, 475-517). */
void oracle_linear_smallk(const float *x, const float *W, const float *bias, float *y, int M, int N, int K, int act) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) acc = fmaf(x[(size_t)m * K + k], W[(size_t)n * K + k], acc);
            acc += bias ? bias[n] : 0.0f;
            y[(size_t)m * N + n] = act == 1 ? (acc > 0.0f ? acc : 0.0f) : acc;
        }
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
