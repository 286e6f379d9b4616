// knn.hip -- batched brute-force k nearest neighbours (+ fused grouping) for gfx950.
//
// Replaces KNN_CUDA 0.2 (knn.cu: cuComputeDistanceGlobal + cuInsertionSort +
// cuParallelSqrt, driven by a Python loop over the batch) behind
// knn_cuda.KNN(k, transpose_mode=True), reference models/Point_MAE_unify.py:56,69.
// The whole batch is ONE launch, no (N x Q) distance matrix ever reaches memory.
//
// Mapping: one wavefront per query.  The reference cloud is staged through LDS
// in chunks shared by the 4 waves of a workgroup; a wave streams 64 reference
// points per step (one per lane, conflict-free stride-3 LDS reads) and keeps the
// current k best as a SORTED LIST SPREAD OVER ITS LANES (lane j = j-th nearest;
// k <= 64).  A candidate survives only if its distance bits are below the list's
// k-th entry (one v_cmp + ballot per 64 points); a survivor is inserted with one
// DPP wave-shift: lanes whose entry is greater move one lane up, the first such
// lane takes the newcomer.  That is exactly cuInsertionSort's rule (insert
// before the first strictly greater entry, drop on ties with the k-th), so the
// output order is (distance, index) ascending -- bit-identical neighbour lists.
//
// A cheap prefilter bounds the survivors: the k-th smallest of the 64 per-lane
// minima is an upper bound of the k-th distance, found by a 32-step bitwise
// search on ballots; only points <= that bound are ever offered to the list.
#include "common.h"

namespace {

constexpr int kKnnChunk = 4096;   // reference points staged per LDS chunk (48 KiB)
constexpr int kKnnWaves = 4;

__device__ __forceinline__ uint32_t ballot_count(bool p) { return (uint32_t)__popcll(__ballot(p)); }

template <bool PREFILTER>
__global__ __launch_bounds__(64 * kKnnWaves) void knn_kernel(const float *__restrict__ ref, const float *__restrict__ query,
                                                             float *__restrict__ dist, int64_t *__restrict__ idx,
                                                             float *__restrict__ neigh, int N, int Q, int K) {
    __shared__ float chunk[3 * kKnnChunk];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tile_x, b;
    xcd_cloud_tile(tile_x, b);                            // whole clouds per XCD: the reference cloud is fetched into ONE L2
    const int q_raw = tile_x * kKnnWaves + wave;
    const bool q_live = q_raw < Q;
    const int q = q_live ? q_raw : Q - 1;  // surplus waves shadow the last query (they must reach the barriers)
    const float *rp = ref + (size_t)b * N * 3;
    const float *qp = query + ((size_t)b * Q + q) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];

    uint32_t ld = 0xFFFFFFFFu;  // lane j: distance bits of the j-th nearest so far
    uint32_t lr = 0;            //         its reference index
    uint32_t thr = 0xFFFFFFFFu; // bits of the k-th entry
    uint32_t bound = 0xFFFFFFFFu;

    const bool single = N <= kKnnChunk;  // the whole cloud fits one chunk: stage it once
    auto stage = [&](int c0, int len) {
        __syncthreads();
        stage_floats(chunk, rp + (size_t)c0 * 3, 3 * len, threadIdx.x, 64 * kKnnWaves);
        __syncthreads();
    };
    // cuComputeDistanceGlobal: tmp = ref - query; ssd += tmp*tmp over d.  Bits of
    // the squared distance of chunk point r (0xFFFFFFFF past the end).
    auto dist_bits = [&](int r, int len) -> uint32_t {
        const int rc = r < len ? r : len - 1;
        const float d = ssd3(chunk[rc * 3 + 0] - qx, chunk[rc * 3 + 1] - qy, chunk[rc * 3 + 2] - qz);
        return r < len ? __float_as_uint(d) : 0xFFFFFFFFu;
    };
    if (single) stage(0, N);

    // ---- optional pass 0: bound = k-th smallest of the per-lane minima --------
    if (PREFILTER && N >= 256) {
        uint32_t lmin = 0xFFFFFFFFu;
        for (int c0 = 0; c0 < N; c0 += kKnnChunk) {
            const int len = min(kKnnChunk, N - c0);
            if (!single) stage(c0, len);
            for (int s0 = 0; s0 < len; s0 += 64) lmin = min(lmin, dist_bits(s0 + lane, len));
        }
        // smallest v with count(lmin <= v) >= K, built from the top bit down
        uint32_t v = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t trial = v | ((1u << bit) - 1u);
            if (ballot_count(lmin <= trial) < (uint32_t)K) v |= (1u << bit);
        }
        bound = v;
    }

    // ---- main pass: stream candidates into the lane-resident sorted list ------
    for (int c0 = 0; c0 < N; c0 += kKnnChunk) {
        const int len = min(kKnnChunk, N - c0);
        if (!single) stage(c0, len);
        for (int s0 = 0; s0 < len; s0 += 64) {
            const uint32_t db = dist_bits(s0 + lane, len);
            unsigned long long mask = __ballot(db < thr && db <= bound);
            while (mask) {
                const int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                const uint32_t dc = readlane_u32(db, l);
                if (dc < thr) {  // re-test: thr shrinks while the batch is consumed
                    const uint32_t rn = (uint32_t)(c0 + s0 + l);
                    const uint32_t ld_left = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ld, DPP_WAVE_SHR1, 0xF, 0xF, false);
                    const uint32_t lr_left = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lr, DPP_WAVE_SHR1, 0xF, 0xF, false);
                    const bool gt = ld > dc;            // my entry moves up one lane
                    const bool gtl = ld_left > dc;      // so does my left neighbour's (lane 0: 0 > dc is false)
                    ld = gt ? (gtl ? ld_left : dc) : ld;
                    lr = gt ? (gtl ? lr_left : rn) : lr;
                    thr = readlane_u32(ld, K - 1);
                }
            }
        }
    }

    if (q_live && lane < K) {
        const size_t o = ((size_t)b * Q + q) * K + lane;
        idx[o] = (int64_t)lr;
        if (dist) dist[o] = sqrtf(__uint_as_float(ld));  // cuParallelSqrt
        if (neigh) {
            const float *nb = rp + (size_t)lr * 3;
            neigh[o * 3 + 0] = nb[0] - qx;
            neigh[o * 3 + 1] = nb[1] - qy;
            neigh[o * 3 + 2] = nb[2] - qz;
        }
    }
}

}  // namespace

extern "C" int upp_knn_ex(const float *ref, const float *query, float *dist, int64_t *idx, float *neigh, int B, int N, int Q,
                          int K, int prefilter, void *stream) {
    if (!ref || !query || !idx || B < 0 || N < 1 || Q < 0 || K < 1) return UPP_E_BADARG;
    if (K > N) return UPP_E_KGTN;
    if (K > 64) return UPP_E_RANGE;
    if (B == 0 || Q == 0) return 0;
    if (B > 65535) return UPP_E_RANGE;
    dim3 grid((Q + kKnnWaves - 1) / kKnnWaves, B);
    hipStream_t st = (hipStream_t)stream;
    if (prefilter)
        hipLaunchKernelGGL((knn_kernel<true>), grid, dim3(64 * kKnnWaves), 0, st, ref, query, dist, idx, neigh, N, Q, K);
    else
        hipLaunchKernelGGL((knn_kernel<false>), grid, dim3(64 * kKnnWaves), 0, st, ref, query, dist, idx, neigh, N, Q, K);
    return upp_launch_status();
}

extern "C" int upp_knn(const float *ref, const float *query, float *dist, int64_t *idx, float *neigh, int B, int N, int Q,
                       int K, void *stream) {
    return upp_knn_ex(ref, query, dist, idx, neigh, B, N, Q, K, 1, stream);
}
