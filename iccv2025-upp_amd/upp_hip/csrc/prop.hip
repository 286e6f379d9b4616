// prop.hip -- the prompt-propagation step of Block.forward
// (reference models/Point_MAE_pretask_dev.py:275-303, `pooling` per SURVEY D.3, `propagate`
// models/Point_MAE_unify.py:22-48) as four row kernels for gfx950.
//
// Reference data flow per block (6 blocks per forward), on the token matrix X (B, L', D) whose rows are
// [cls | P prompts | 64 centre tokens]:
//   nb   = X[i1]                      8 "level-1 neighbour" rows for each of the B*32 level-2 groups
//   nb   = nb + drop_path(nb)         per-group stochastic-depth factor
//   pool = max_k nb + mean_k nb   ->  BatchNorm1d over the B*32 rows  -> lc
//   c2   = lc + 0.3 * X[i2]           the level-2 centre's own token
//   X[tok i] += 0.3 * sum_k w8[i,k] * c2[idx8[i,k]]        (8 nearest level-2 centres, inverse-distance weights)
// The reference does this with ~35 kernels forward and ~100 in backward (advanced indexing, index_put with
// sorts, a full sort of the distances in every block).  Here: idx8 / w8 depend only on the centres and are
// computed once per forward by the host; the row indices are handed over as absolute rows of X (the host
// converts the reference's flat / per-sample index conventions, including its stride-64-into-stride-74
// behaviour, into that form); pool and interpolate are one kernel each, one wavefront per row, and the
// backward scatter-adds are done as deterministic "scan the index list for my row" gathers, no atomics.
#include "common.h"

int upp_bn_finalize_launch(const float *part, int slabs, int per, int rows, int C, int training, float momentum, float eps,
                           float *running_mean, float *running_var, float *mean, float *rstd, hipStream_t st);   // pointwise.hip

namespace {

constexpr int kMaxE = 8;   // D <= 512
constexpr int kNb = 8;     // neighbours per level-2 group and per interpolation (reference: group_size=8, de_neighbors=8)

// pooled[g][c] = max_k v_k + (sum_k v_k) / 8,  v_k = X[i1[g][k]][c] * (1 + s_g);  amax[g][c] = arg max
__global__ __launch_bounds__(256) void prop_pool_fwd_kernel(const float *__restrict__ X, const int32_t *__restrict__ i1,
                                                            const float *__restrict__ u, float keep, float *__restrict__ pooled,
                                                            uint8_t *__restrict__ amax, int groups, int D) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (g >= groups) return;
    const float f = 1.0f + (u ? floorf(keep + u[g]) / keep : 1.0f);   // x + drop_path(x): factor 1 + s
    float mx[kMaxE], sm[kMaxE];
    int am[kMaxE];
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) { mx[e] = -__builtin_inff(); sm[e] = 0.0f; am[e] = 0; }
    // index loads, then all 8 x kMaxE row loads, then the arithmetic: a dependent global load is ~1 us
    int rows8[kNb];
#pragma unroll
    for (int k = 0; k < kNb; ++k) rows8[k] = i1[g * kNb + k];
    float val[kNb][kMaxE];
#pragma unroll
    for (int k = 0; k < kNb; ++k)
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) val[k][e] = X[(size_t)rows8[k] * D + min(lane + 64 * e, D - 1)];   // clamped: no branch
#pragma unroll
    for (int k = 0; k < kNb; ++k)
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) {
            const float v = val[k][e] * f;
            if (v > mx[e]) { mx[e] = v; am[e] = k; }   // first maximum wins, as torch.max
            sm[e] += v;
        }
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) {
        const int c = lane + 64 * e;
        if (c < D) { pooled[(size_t)g * D + c] = mx[e] + sm[e] * 0.125f; amax[(size_t)g * D + c] = (uint8_t)am[e]; }
    }
}

// Positions q in [q0, n) with list[q] == key, appended to hits[] (LDS, per wave) until `cap` could overflow.
// Returns the number collected and advances q0 (wave-uniform).  list lives in LDS.
__device__ __forceinline__ int collect_hits(const int32_t *list, int n, int key, int *hits, int cap, int lane, int &q0) {
    int cnt = 0;
    int q = q0;
    for (; q < n && cnt + 64 <= cap; q += 64) {
        const int me = q + lane;
        const bool pred = me < n && list[me] == key;
        const unsigned long long mask = __ballot(pred);
        if (mask) {
            const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            if (pred) hits[pos] = me;
            cnt += __popcll(mask);
        }
    }
    q0 = q;
    __builtin_amdgcn_wave_barrier();
    return cnt;
}

constexpr int kHitCap = 256;

// g_X[r][c] = sum over (g,k) with i1[g][k] == r of  (1 + s_g) * g_pooled[g][c] * (1/8 + [amax[g][c] == k])
// The whole index list is staged in LDS once per workgroup; each wave scans it for its row, collects the hits and
// then accumulates them four at a time (independent row loads in flight), in ascending list order: deterministic.
__global__ __launch_bounds__(256) void prop_pool_bwd_kernel(const float *__restrict__ g_pooled, const uint8_t *__restrict__ amax,
                                                            const int32_t *__restrict__ i1, const float *__restrict__ u, float keep,
                                                            float *__restrict__ g_X, int rows, int groups, int D) {
    extern __shared__ int32_t lds_i[];
    const int n = groups * kNb;
    int32_t *list = lds_i;                               // [n]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int *hits = lds_i + n + wave * kHitCap;
    for (int i0 = threadIdx.x; i0 < n; i0 += 256 * 8) {   // 8 independent loads in flight per thread
        int32_t t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = i0 + q * 256 < n ? i1[i0 + q * 256] : 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) if (i0 + q * 256 < n) list[i0 + q * 256] = t[q];
    }
    __syncthreads();
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    float acc[kMaxE];
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) acc[e] = 0.0f;
    int q0 = 0;
    while (q0 < n) {
        const int cnt = collect_hits(list, n, r, hits, kHitCap, lane, q0);
        for (int h0 = 0; h0 < cnt; h0 += 4) {
            int g[4], k[4];
            float f[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool live = h0 + t < cnt;
                const int hit = hits[live ? h0 + t : h0];
                g[t] = hit / kNb; k[t] = hit - g[t] * kNb;
                f[t] = live ? 1.0f + (u ? floorf(keep + u[g[t]]) / keep : 1.0f) : 0.0f;
            }
            float gp[kMaxE][4]; int am[kMaxE][4];
#pragma unroll
            for (int e = 0; e < kMaxE; ++e) {
                const int c = min(lane + 64 * e, D - 1);
#pragma unroll
                for (int t = 0; t < 4; ++t) { gp[e][t] = g_pooled[(size_t)g[t] * D + c]; am[e][t] = amax[(size_t)g[t] * D + c]; }
            }
#pragma unroll
            for (int e = 0; e < kMaxE; ++e)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[e] += gp[e][t] * f[t] * (0.125f + (am[e][t] == k[t] ? 1.0f : 0.0f));
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) { const int c = lane + 64 * e; if (c < D) g_X[(size_t)r * D + c] = acc[e]; }
}

// out[b][t] = X[b][t]                                                     for t < L' - T   (cls, prompts)
//           = X[b][t] + 0.3 * sum_k w8[b][i][k] * (lc[b][j] + 0.3 * X[i2[b][j]]),  j = idx8[b][i][k], i = t - (L' - T)
__global__ __launch_bounds__(256) void prop_interp_fwd_kernel(const float *__restrict__ X, const float *__restrict__ lc,
                                                              const int32_t *__restrict__ i2, const int32_t *__restrict__ idx8,
                                                              const float *__restrict__ w8, float *__restrict__ out, int B, int Lp,
                                                              int T, int G2, int D) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (r >= B * Lp) return;
    const int b = r / Lp, t = r - b * Lp;
    const int i = t - (Lp - T);
    int cc[kMaxE];
    float xv[kMaxE], acc[kMaxE];
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) { cc[e] = min(lane + 64 * e, D - 1); xv[e] = X[(size_t)r * D + cc[e]]; acc[e] = 0.0f; }
    if (i >= 0) {
        // three dependent hops (idx8 -> i2 -> X row): each hop's loads are issued for all 8 neighbours at once
        int j[kNb], cr[kNb];
        float w[kNb];
#pragma unroll
        for (int k = 0; k < kNb; ++k) { j[k] = idx8[((size_t)b * T + i) * kNb + k]; w[k] = w8[((size_t)b * T + i) * kNb + k]; }
#pragma unroll
        for (int k = 0; k < kNb; ++k) cr[k] = i2[b * G2 + j[k]];
#pragma unroll
        for (int h = 0; h < kMaxE; h += 4) {          // 4 column slots x 8 neighbours x 2 arrays = 64 loads in flight
            float lv[4][kNb], cv[4][kNb];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int k = 0; k < kNb; ++k) { lv[e][k] = lc[((size_t)b * G2 + j[k]) * D + cc[h + e]]; cv[e][k] = X[(size_t)cr[k] * D + cc[h + e]]; }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int k = 0; k < kNb; ++k) acc[h + e] += (lv[e][k] + 0.3f * cv[e][k]) * w[k];
            if (64 * (h + 4) >= D) break;             // wave-uniform: no live column beyond
        }
    }
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) {
        const int c = lane + 64 * e;
        if (c < D) out[(size_t)r * D + c] = xv[e] + (i >= 0 ? 0.3f * acc[e] : 0.0f);
    }
}

// g_c2[b][j][c] = 0.3 * sum over (i,k) with idx8[b][i][k] == j of w8[b][i][k] * g_out[b][L'-T+i][c]
// A workgroup serves 4 consecutive level-2 centres of one sample (G2 % 4 == 0 is checked by the host), stages that
// sample's idx8 / w8 in LDS, collects the hits per wave and accumulates four token rows at a time.
__global__ __launch_bounds__(256) void prop_interp_bwd_c2_kernel(const float *__restrict__ g_out, const int32_t *__restrict__ idx8,
                                                                 const float *__restrict__ w8, float *__restrict__ g_c2, int B, int Lp,
                                                                 int T, int G2, int D) {
    extern __shared__ int32_t lds_i[];
    const int n = T * kNb;
    int32_t *list = lds_i;                                   // [n]
    float *wl = reinterpret_cast<float *>(lds_i + n);        // [n]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int *hits = lds_i + 2 * n + wave * kHitCap;
    const int gj0 = blockIdx.x * 4;
    const int b = gj0 / G2;
    for (int i0 = threadIdx.x; i0 < n; i0 += 256 * 4) {     // 8 independent loads in flight per thread
        int32_t ti[4]; float tw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int i = min(i0 + q * 256, n - 1); ti[q] = idx8[(size_t)b * n + i]; tw[q] = w8[(size_t)b * n + i]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) if (i0 + q * 256 < n) { list[i0 + q * 256] = ti[q]; wl[i0 + q * 256] = tw[q]; }
    }
    __syncthreads();
    const int gj = gj0 + wave;
    if (gj >= B * G2) return;
    const int j = gj - b * G2;
    float acc[kMaxE];
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) acc[e] = 0.0f;
    int q0 = 0;
    while (q0 < n) {
        const int cnt = collect_hits(list, n, j, hits, kHitCap, lane, q0);
        for (int h0 = 0; h0 < cnt; h0 += 8) {                // 8 token rows (x kMaxE column slots) in flight
            const float *grow[8];
            float w[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const bool live = h0 + t < cnt;
                const int hit = hits[live ? h0 + t : h0];
                w[t] = live ? wl[hit] : 0.0f;
                grow[t] = g_out + ((size_t)b * Lp + (Lp - T) + hit / kNb) * D;
            }
            float gv[kMaxE][8];
#pragma unroll
            for (int e = 0; e < kMaxE; ++e) {
                const int c = min(lane + 64 * e, D - 1);
#pragma unroll
                for (int t = 0; t < 8; ++t) gv[e][t] = grow[t][c];
            }
#pragma unroll
            for (int e = 0; e < kMaxE; ++e)
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[e] += gv[e][t] * w[t];
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) { const int c = lane + 64 * e; if (c < D) g_c2[(size_t)gj * D + c] = 0.3f * acc[e]; }
}

// g_X[r] = g_out[r] + 0.3 * g_c2[m]  for the (unique) m with i2[m] == r, else g_out[r]
__global__ __launch_bounds__(256) void prop_interp_bwd_x_kernel(const float *__restrict__ g_out, const float *__restrict__ g_c2,
                                                                const int32_t *__restrict__ i2, float *__restrict__ g_X, int rows,
                                                                int groups, int D) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (r >= rows) return;
    int cc[kMaxE];
    float acc[kMaxE];
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) { cc[e] = min(lane + 64 * e, D - 1); acc[e] = g_out[(size_t)r * D + cc[e]]; }
    // the whole index list is compared first (independent loads), then the matching rows are added in ascending order
    for (int base = 0; base < groups; base += 64 * 8) {
        int v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int m = base + q * 64 + lane; v[q] = i2[min(m, groups - 1)]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int m0 = base + q * 64;
            unsigned long long mask = __ballot(m0 + lane < groups && v[q] == r);
            while (mask) {
                const int m = m0 + __builtin_ctzll(mask);
                mask &= mask - 1;
#pragma unroll
                for (int e = 0; e < kMaxE; ++e) acc[e] += 0.3f * g_c2[(size_t)m * D + cc[e]];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) { const int c = lane + 64 * e; if (c < D) g_X[(size_t)r * D + c] = acc[e]; }
}

inline dim3 rows_grid(long long rows) { return dim3((unsigned)((rows + 3) / 4)); }

// ---------------------------------------------------------------------------------------------
// Fused propagation step (pool -> BatchNorm1d -> interpolate) with inverted index lists.
//
// The index lists (i1, i2, idx8) depend only on the centres, so they are the same for every block of a forward:
// the host inverts each list ONCE per forward into a CSR (csr_build_kernel) and the backward kernels walk
// "the entries that reference my row" directly instead of scanning the whole list in every block.
// BatchNorm runs inside the step: pool emits per-workgroup (sum, M2) column partials, a one-workgroup-per-64-columns
// kernel (pointwise.hip) combines them (Chan's update, f64) into mean / rstd and updates the running statistics, and interpolate
// normalises the pooled rows as it gathers them.  Backward mirrors it: the c2 kernel emits the column partials of
// sum(g_lc) and sum(g_lc * xhat), a finalize kernel reduces them to g_beta / g_gamma, and ONE row kernel produces the
// whole g_X (identity path + i2 scatter + BatchNorm-backward + pool-backward), so autograd adds nothing on top.

// Stable CSR of an index list: perm[start[r] .. start[r+1]) = the positions q (ascending) whose key equals r, where
// key(q) = (q / seg_len) * seg_rows + keys[q]  (seg_rows = 0: keys are absolute).  Keys outside [0, rows) are skipped.
// One workgroup of 1024 threads; counts and cursors live in LDS.
constexpr int kCsrThreads = 1024;
template <bool PERM_IN_LDS>
__global__ __launch_bounds__(kCsrThreads) void csr_build_kernel(const int32_t *__restrict__ keys, int n, int seg_len, int seg_rows,
                                                                int rows, int32_t *__restrict__ start, int32_t *__restrict__ perm_out) {
    extern __shared__ int32_t lds_i[];
    int32_t *cnt = lds_i;                   // [rows]
    int32_t *scan = lds_i + rows;           // [kCsrThreads]
    // the permutation is filled by atomics and then sorted per row: in LDS when it fits (an insertion sort on global
    // memory is a chain of ~1 us round trips), written out coalesced at the end
    int32_t *perm = PERM_IN_LDS ? scan + kCsrThreads : perm_out;
    int32_t *prow = perm + n;               // PERM_IN_LDS: the row of every filled slot
    const int tid = threadIdx.x;
    for (int r = tid; r < rows; r += kCsrThreads) cnt[r] = 0;
    __syncthreads();
    auto key_of = [&](int q, int raw) { const int k = (q / seg_len) * seg_rows + raw; return (k >= 0 && k < rows) ? k : -1; };
    for (int q0 = tid; q0 < n; q0 += kCsrThreads * 8) {          // 8 independent loads in flight per thread
        int raw[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) raw[t] = keys[min(q0 + t * kCsrThreads, n - 1)];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int q = q0 + t * kCsrThreads;
            const int k = q < n ? key_of(q, raw[t]) : -1;
            if (k >= 0) atomicAdd(&cnt[k], 1);
        }
    }
    __syncthreads();
    const int per = (rows + kCsrThreads - 1) / kCsrThreads;      // thread t owns rows [t*per, (t+1)*per)
    const int lo = min(rows, tid * per), hi = min(rows, lo + per);
    int sum = 0;
    for (int r = lo; r < hi; ++r) sum += cnt[r];
    scan[tid] = sum;
    __syncthreads();
    for (int off = 1; off < kCsrThreads; off <<= 1) {             // inclusive Hillis-Steele scan
        const int v = tid >= off ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    const int first = scan[tid] - sum;
    {
        int base = first;
        for (int r = lo; r < hi; ++r) { const int c = cnt[r]; start[r] = base; cnt[r] = base; base += c; }
    }
    if (tid == kCsrThreads - 1) start[rows] = scan[tid];
    __syncthreads();
    for (int q0 = tid; q0 < n; q0 += kCsrThreads * 8) {
        int raw[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) raw[t] = keys[min(q0 + t * kCsrThreads, n - 1)];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int q = q0 + t * kCsrThreads;
            const int k = q < n ? key_of(q, raw[t]) : -1;
            if (k >= 0) {
                const int pos = atomicAdd(&cnt[k], 1);
                perm[pos] = q;
                if (PERM_IN_LDS) prow[pos] = k;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    if (PERM_IN_LDS) {
        // The atomics filled each row's segment in arbitrary order.  Rank sort, one thread per ENTRY: its final slot is
        // segment start + the number of smaller positions in its segment -- independent LDS reads (no dependent
        // read-modify-write chain as in an insertion sort) and balanced over the threads whatever the segment lengths.
        const int total = scan[kCsrThreads - 1];      // inclusive scan of the row counts: number of valid entries
        for (int p = tid; p < total; p += kCsrThreads) {
            const int v = perm[p], k = prow[p];
            const int e = cnt[k], sgm = k > 0 ? cnt[k - 1] : 0;   // after the fill cnt[k] is the END of row k
            int rank = 0;
            for (int bq = sgm; bq < e; ++bq) rank += perm[bq] < v ? 1 : 0;
            perm_out[sgm + rank] = v;
        }
        return;
    }
    {   // global-memory variant (lists too long for LDS): insertion sort per row
        int base = first;
        for (int r = lo; r < hi; ++r) {
            const int end = cnt[r];
            for (int a = base + 1; a < end; ++a) {
                const int v = perm[a];
                int bpos = a - 1;
                while (bpos >= base && perm[bpos] > v) { perm[bpos + 1] = perm[bpos]; --bpos; }
                perm[bpos + 1] = v;
            }
            base = end;
        }
    }
}

// pool forward + BatchNorm column partials.  part[(wg*2+0)*D + c] = sum of the workgroup's (<= 4) pooled values of
// column c, part[(wg*2+1)*D + c] = their M2 about the workgroup mean.
__global__ __launch_bounds__(256) void prop_pool_stats_kernel(const float *__restrict__ X, const int32_t *__restrict__ i1,
                                                              const float *__restrict__ u, float keep, float *__restrict__ pooled,
                                                              uint8_t *__restrict__ amax, float *__restrict__ part, int groups, int D) {
    const int wgx = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);   // XCD x owns a contiguous eighth of the rows / groups (common.h)
    __shared__ float sh[4][64 * kMaxE];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = wgx * 4 + wave;
    if (g < groups) {
        const float f = 1.0f + (u ? floorf(keep + u[g]) / keep : 1.0f);
        int rows8[kNb];
#pragma unroll
        for (int k = 0; k < kNb; ++k) rows8[k] = i1[g * kNb + k];
        float val[kNb][kMaxE];
#pragma unroll
        for (int k = 0; k < kNb; ++k)
#pragma unroll
            for (int e = 0; e < kMaxE; ++e) val[k][e] = X[(size_t)rows8[k] * D + min(lane + 64 * e, D - 1)];
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) {
            float mx = -__builtin_inff(), sm = 0.0f;
            int am = 0;
#pragma unroll
            for (int k = 0; k < kNb; ++k) {
                const float v = val[k][e] * f;
                if (v > mx) { mx = v; am = k; }
                sm += v;
            }
            const int c = lane + 64 * e;
            const float pv = mx + sm * 0.125f;
            if (c < D) { pooled[(size_t)g * D + c] = pv; amax[(size_t)g * D + c] = (uint8_t)am; sh[wave][c] = pv; }
        }
    }
    __syncthreads();
    const int nw = min(4, groups - (int)wgx * 4);
    for (int c = threadIdx.x; c < D; c += 256) {
        float sum = 0.0f;
        for (int w = 0; w < nw; ++w) sum += sh[w][c];
        const float mean = sum / (float)nw;
        float m2 = 0.0f;
        for (int w = 0; w < nw; ++w) { const float dv = sh[w][c] - mean; m2 = __builtin_fmaf(dv, dv, m2); }
        part[((size_t)wgx * 2 + 0) * D + c] = sum;
        part[((size_t)wgx * 2 + 1) * D + c] = m2;
    }
}

// interpolate forward reading the PRE-BatchNorm pooled rows: lc = ((pooled - mean) * rstd) * gamma + beta on the fly
__global__ __launch_bounds__(256) void prop_interp_bn_fwd_kernel(const float *__restrict__ X, const float *__restrict__ pooled,
                                                                 const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 const int32_t *__restrict__ i2, const int32_t *__restrict__ idx8,
                                                                 const float *__restrict__ w8, float *__restrict__ out, int B, int Lp,
                                                                 int T, int G2, int D) {
    const int wgx = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);   // XCD x owns a contiguous eighth of the rows / groups (common.h)
    const int lane = threadIdx.x & 63;
    const int r = wgx * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (r >= B * Lp) return;
    const int b = r / Lp, t = r - b * Lp;
    const int i = t - (Lp - T);
    int cc[kMaxE];
    float xv[kMaxE], acc[kMaxE];
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) { cc[e] = min(lane + 64 * e, D - 1); xv[e] = X[(size_t)r * D + cc[e]]; acc[e] = 0.0f; }
    if (i >= 0) {
        int j[kNb], cr[kNb];
        float w[kNb];
#pragma unroll
        for (int k = 0; k < kNb; ++k) { j[k] = idx8[((size_t)b * T + i) * kNb + k]; w[k] = w8[((size_t)b * T + i) * kNb + k]; }
        float mu[kMaxE], rs[kMaxE], ga[kMaxE], be[kMaxE];
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) { mu[e] = mean[cc[e]]; rs[e] = rstd[cc[e]]; ga[e] = gamma[cc[e]]; be[e] = beta[cc[e]]; }
#pragma unroll
        for (int k = 0; k < kNb; ++k) cr[k] = i2[b * G2 + j[k]];
#pragma unroll
        for (int h = 0; h < kMaxE; h += 4) {
            float lv[4][kNb], cv[4][kNb];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int k = 0; k < kNb; ++k) { lv[e][k] = pooled[((size_t)b * G2 + j[k]) * D + cc[h + e]]; cv[e][k] = X[(size_t)cr[k] * D + cc[h + e]]; }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int k = 0; k < kNb; ++k) {
                    const float lcv = ((lv[e][k] - mu[h + e]) * rs[h + e]) * ga[h + e] + be[h + e];
                    acc[h + e] += (lcv + 0.3f * cv[e][k]) * w[k];
                }
            if (64 * (h + 4) >= D) break;
        }
    }
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) {
        const int c = lane + 64 * e;
        if (c < D) out[(size_t)r * D + c] = xv[e] + (i >= 0 ? 0.3f * acc[e] : 0.0f);
    }
}

// g_c2[gj] = 0.3 * sum over the CSR entries q of row gj of w8[q] * g_out[token row of q]; plus the BatchNorm-backward
// column partials of the workgroup's 4 rows: part[(wg*2+0)*D+c] = sum g_c2, part[(wg*2+1)*D+c] = sum g_c2 * xhat.
__global__ __launch_bounds__(256) void prop_c2_csr_kernel(const float *__restrict__ g_out, const int32_t *__restrict__ start8,
                                                          const int32_t *__restrict__ perm8, const float *__restrict__ w8,
                                                          const float *__restrict__ pooled, const float *__restrict__ mean,
                                                          const float *__restrict__ rstd, float *__restrict__ g_c2,
                                                          float *__restrict__ part, int B, int Lp, int T, int G2, int D) {
    const int wgx = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);   // XCD x owns a contiguous eighth of the rows / groups (common.h)
    __shared__ float sh[2][4][64 * kMaxE];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gj = wgx * 4 + wave;
    const int groups = B * G2;
    if (gj < groups) {
        int cc[kMaxE];
        float acc[kMaxE], pv[kMaxE], mu[kMaxE], rs[kMaxE];
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) {
            cc[e] = min(lane + 64 * e, D - 1); acc[e] = 0.0f;
            pv[e] = pooled[(size_t)gj * D + cc[e]]; mu[e] = mean[cc[e]]; rs[e] = rstd[cc[e]];
        }
        const int s = start8[gj], eend = start8[gj + 1];
        for (int h0 = s; h0 < eend; h0 += 8) {                // 8 token rows x kMaxE column slots in flight
            int q[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) q[t] = perm8[min(h0 + t, eend - 1)];
            float w[8];
            const float *grow[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int tok = q[t] / kNb;                   // b * T + i
                const int bb = tok / T, ii = tok - bb * T;
                w[t] = w8[q[t]];
                grow[t] = g_out + ((size_t)bb * Lp + (Lp - T) + ii) * D;
            }
            float gv[kMaxE][8];
#pragma unroll
            for (int e = 0; e < kMaxE; ++e)
#pragma unroll
                for (int t = 0; t < 8; ++t) gv[e][t] = grow[t][cc[e]];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float wt = h0 + t < eend ? w[t] : 0.0f;
#pragma unroll
                for (int e = 0; e < kMaxE; ++e) acc[e] += gv[e][t] * wt;
            }
        }
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) {
            const int c = lane + 64 * e;
            if (c < D) {
                const float gl = 0.3f * acc[e];
                g_c2[(size_t)gj * D + c] = gl;
                sh[0][wave][c] = gl;
                sh[1][wave][c] = gl * ((pv[e] - mu[e]) * rs[e]);
            }
        }
    }
    __syncthreads();
    const int nw = min(4, groups - (int)wgx * 4);
    for (int c = threadIdx.x; c < D; c += 256) {
        float s0 = 0.0f, s1 = 0.0f;
        for (int w = 0; w < nw; ++w) { s0 += sh[0][w][c]; s1 += sh[1][w][c]; }
        part[((size_t)wgx * 2 + 0) * D + c] = s0;
        part[((size_t)wgx * 2 + 1) * D + c] = s1;
    }
}

// g_beta[c] = sum of the "sum g_lc" partials, g_gamma[c] = sum of the "sum g_lc * xhat" partials (fixed order)
__global__ __launch_bounds__(256) void prop_bn_grad_kernel(const float *__restrict__ part, int wgs, int D, float *__restrict__ g_gamma,
                                                           float *__restrict__ g_beta) {
    __shared__ float sb[4][64], sg[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int cc = min(c, D - 1);
    float ab = 0.0f, ag = 0.0f;
    for (int w0 = wave; w0 < wgs; w0 += 4 * 8) {
        float pb[8], pg[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int w = min(w0 + 4 * t, wgs - 1);
            pb[t] = part[((size_t)w * 2 + 0) * D + cc]; pg[t] = part[((size_t)w * 2 + 1) * D + cc];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) if (w0 + 4 * t < wgs) { ab += pb[t]; ag += pg[t]; }
    }
    sb[wave][lane] = ab; sg[wave][lane] = ag;
    __syncthreads();
    if (wave == 0 && c < D) {
        g_beta[c] = (sb[0][lane] + sb[1][lane]) + (sb[2][lane] + sb[3][lane]);
        g_gamma[c] = (sg[0][lane] + sg[1][lane]) + (sg[2][lane] + sg[3][lane]);
    }
}

// The whole g_X of the step, one wavefront per row r of X:
//   g_X[r] = g_out[r] + 0.3 * sum_{m: i2[m] == r} g_c2[m]
//          + sum_{(g,k): i1[g][k] == r} (1 + s_g) * g_pooled[g] * (1/8 + [amax[g] == k])
//   g_pooled[g] = gamma * rstd * (g_c2[g] - g_beta/n - xhat[g] * g_gamma/n)      (training; eval: gamma * rstd * g_c2[g])
__global__ __launch_bounds__(256) void prop_x_csr_kernel(const float *__restrict__ g_out, const float *__restrict__ g_c2,
                                                         const float *__restrict__ pooled, const uint8_t *__restrict__ amax,
                                                         const float *__restrict__ mean, const float *__restrict__ rstd,
                                                         const float *__restrict__ gamma, const float *__restrict__ g_gamma,
                                                         const float *__restrict__ g_beta, const float *__restrict__ u, float keep,
                                                         const int32_t *__restrict__ start1, const int32_t *__restrict__ perm1,
                                                         const int32_t *__restrict__ start2, const int32_t *__restrict__ perm2,
                                                         float *__restrict__ g_X, int rows, int groups, int D, int training) {
    const int wgx = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);   // XCD x owns a contiguous eighth of the rows / groups (common.h)
    const int lane = threadIdx.x & 63;
    const int r = wgx * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (r >= rows) return;
    const int s1 = start1[r], e1 = start1[r + 1], s2 = start2[r], e2 = start2[r + 1];
    int cc[kMaxE];
    float acc[kMaxE], mu[kMaxE], rs[kMaxE], ga[kMaxE], c1[kMaxE], c2[kMaxE];
    const float inv_n = 1.0f / (float)groups;
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) {
        cc[e] = min(lane + 64 * e, D - 1);
        acc[e] = g_out[(size_t)r * D + cc[e]];
        mu[e] = mean[cc[e]]; rs[e] = rstd[cc[e]]; ga[e] = gamma[cc[e]];
        c1[e] = 0.0f; c2[e] = 0.0f;
    }
    if (training) {                                            // one uniform branch around the whole batch of loads
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) { c1[e] = g_beta[cc[e]]; c2[e] = g_gamma[cc[e]]; }
    }
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) { c1[e] *= inv_n; c2[e] *= inv_n; }
    for (int h0 = s2; h0 < e2; h0 += 2) {                     // level-2 centres whose own token is this row (usually <= 1)
        const int m0 = perm2[h0], m1 = perm2[min(h0 + 1, e2 - 1)];
        float a0[kMaxE], a1[kMaxE];
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) { a0[e] = g_c2[(size_t)m0 * D + cc[e]]; a1[e] = g_c2[(size_t)m1 * D + cc[e]]; }
        const float f1 = h0 + 1 < e2 ? 0.3f : 0.0f;
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) { acc[e] += 0.3f * a0[e]; acc[e] += f1 * a1[e]; }
    }
    for (int h0 = s1; h0 < e1; h0 += 4) {                     // groups that list this row as a neighbour
        int g[4], k[4];
        float f[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { const int q = perm1[min(h0 + t, e1 - 1)]; g[t] = q / kNb; k[t] = q - g[t] * kNb; }
#pragma unroll
        for (int t = 0; t < 4; ++t) f[t] = h0 + t < e1 ? 1.0f + (u ? floorf(keep + u[g[t]]) / keep : 1.0f) : 0.0f;
        float gl[kMaxE][4], pv[kMaxE][4];
        int am[kMaxE][4];
#pragma unroll
        for (int e = 0; e < kMaxE; ++e)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const size_t o = (size_t)g[t] * D + cc[e];
                gl[e][t] = g_c2[o]; pv[e][t] = pooled[o]; am[e][t] = amax[o];
            }
#pragma unroll
        for (int e = 0; e < kMaxE; ++e)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float xh = (pv[e][t] - mu[e]) * rs[e];
                const float gp = (ga[e] * rs[e]) * (gl[e][t] - c1[e] - xh * c2[e]);
                acc[e] += gp * f[t] * (0.125f + (am[e][t] == k[t] ? 1.0f : 0.0f));
            }
    }
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) { const int c = lane + 64 * e; if (c < D) g_X[(size_t)r * D + c] = acc[e]; }
}

// Index lists of the propagation step for one forward, in ONE launch (the reference spends ~30 small torch kernels on
// them: index arithmetic, square_distance, a full sort, the weight normalisation):
//   i1a / i2a : absolute rows of the (B*L') token matrix for the level-1 neighbour lists / level-2 centres, from the
//               reference's index tensors (flat with batch offsets b*T when gather_idx == 0 -- re-interpreted in the
//               (B*G)-row view G = L' - off exactly as models/Point_MAE_pretask_dev.py:291-292 does -- or per-sample);
//   idx8 / w8 : for every level-1 centre its 8 nearest level-2 centres by  d = |a|^2 + |b|^2 - 2 a.b  (the reference's
//               square_distance form, models/modules.py:13-32), ascending (d, index), w = (1/(d+eps)) / sum.
// One wavefront per (sample, level-1 centre); lane = level-2 centre (G2 <= 64).
__global__ __launch_bounds__(256) void prop_index_kernel(const float *__restrict__ c1, const float *__restrict__ c2,
                                                         const int64_t *__restrict__ i1, const int64_t *__restrict__ i2, int gather_idx,
                                                         int B, int T, int G2, int Lp, int off, float eps,
                                                         int32_t *__restrict__ i1a, int32_t *__restrict__ i2a,
                                                         int32_t *__restrict__ idx8, float *__restrict__ w8) {
    const int lane = threadIdx.x & 63;
    const int wg = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    {   // integer part: every wave converts 64 consecutive entries of [i1 | i2]
        const int n1 = B * G2 * kNb, n2 = B * G2;
        const int e = wg * 64 + lane;
        if (e < n1 + n2) {
            const bool first = e < n1;
            const long long v = first ? i1[e] : i2[e - n1];
            long long r;
            if (gather_idx) {
                const int bsmp = first ? e / (G2 * kNb) : (e - n1) / G2;
                r = (long long)bsmp * Lp + off + v;
            } else {
                const int G = Lp - off;
                r = (v / G) * Lp + off + v % G;
            }
            if (first) i1a[e] = (int32_t)r; else i2a[e - n1] = (int32_t)r;
        }
    }
    if (wg >= B * T) return;
    const int b = wg / T;
    const float *a = c1 + (size_t)wg * 3;
    const float ax = a[0], ay = a[1], az = a[2];
    const int j = min(lane, G2 - 1);
    const float *q = c2 + ((size_t)b * G2 + j) * 3;
    const float bx = q[0], by = q[1], bz = q[2];
    // dist = -2 * (a . b); dist += |a|^2; dist += |b|^2   (operation order of the reference formula)
    float d = -2.0f * __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, ax * bx));
    d += (ax * ax + ay * ay) + az * az;
    d += (bx * bx + by * by) + bz * bz;
    const uint32_t bits = __float_as_uint(d);
    uint32_t img = lane < G2 ? ((bits & 0x80000000u) ? ~bits : (bits | 0x80000000u)) : 0xFFFFFFFFu;   // order-preserving image
    float dk[kNb];
    int jk[kNb];
#pragma unroll
    for (int k = 0; k < kNb; ++k) {
        const uint32_t m = wave_min_u32(img);
        const int win = __builtin_ctzll(__ballot(img == m));          // lowest index among equal distances
        jk[k] = win;
        dk[k] = __uint_as_float((m & 0x80000000u) ? (m & 0x7FFFFFFFu) : ~m);
        if (lane == win) img = 0xFFFFFFFFu;
    }
    float rc[kNb], sum = 0.0f;
#pragma unroll
    for (int k = 0; k < kNb; ++k) { rc[k] = 1.0f / (dk[k] + eps); sum += rc[k]; }
    if (lane < kNb) {
        float wv = 0.0f; int jv = 0;
#pragma unroll
        for (int k = 0; k < kNb; ++k) if (lane == k) { wv = rc[k] / sum; jv = jk[k]; }
        idx8[(size_t)wg * kNb + lane] = jv;
        w8[(size_t)wg * kNb + lane] = wv;
    }
}

// Gradient of the step w.r.t. the interpolation weights (stage 2 of the recipe: the weights are functions of centres that carry a
// gradient back to the prompters):  g_w8[b,i,k] = 0.3 * < g_out[b, L'-T+i, :], ctr[b, idx8[b,i,k], :] >,
//   ctr[gj] = lc[gj] + 0.3 * X[i2[gj]],  lc = BatchNorm(pooled) (mean != null: pooled are pre-BatchNorm rows) or `pooled` itself.
// One wavefront per (sample, level-1 centre); fixed summation order (lane partials in column order, then the wave tree).
__global__ __launch_bounds__(256) void prop_w8_grad_kernel(const float *__restrict__ g_out, const float *__restrict__ X,
                                                           const float *__restrict__ pooled, const float *__restrict__ mean,
                                                           const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, const int32_t *__restrict__ i2,
                                                           const int32_t *__restrict__ idx8, float *__restrict__ g_w8, int B, int Lp,
                                                           int T, int G2, int D) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (tok >= B * T) return;
    const int b = tok / T, i = tok - b * T;
    const float *grow = g_out + ((size_t)b * Lp + (Lp - T) + i) * D;
    int cc[kMaxE];
    float gv[kMaxE], mu[kMaxE], rs[kMaxE], ga[kMaxE], be[kMaxE];
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) {
        const int c = lane + 64 * e;
        cc[e] = min(c, D - 1);
        gv[e] = c < D ? grow[cc[e]] : 0.0f;
        mu[e] = 0.0f; rs[e] = 1.0f; ga[e] = 1.0f; be[e] = 0.0f;
    }
    if (mean) {
#pragma unroll
        for (int e = 0; e < kMaxE; ++e) { mu[e] = mean[cc[e]]; rs[e] = rstd[cc[e]]; ga[e] = gamma[cc[e]]; be[e] = beta[cc[e]]; }
    }
    int j[kNb], cr[kNb];
#pragma unroll
    for (int k = 0; k < kNb; ++k) j[k] = idx8[(size_t)tok * kNb + k];
#pragma unroll
    for (int k = 0; k < kNb; ++k) cr[k] = i2[b * G2 + j[k]];
    float dot[kNb];
#pragma unroll
    for (int k = 0; k < kNb; ++k) dot[k] = 0.0f;
#pragma unroll
    for (int h = 0; h < kMaxE; h += 4) {
        float lv[4][kNb], cv[4][kNb];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < kNb; ++k) { lv[e][k] = pooled[((size_t)b * G2 + j[k]) * D + cc[h + e]]; cv[e][k] = X[(size_t)cr[k] * D + cc[h + e]]; }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < kNb; ++k) {
                const float lcv = ((lv[e][k] - mu[h + e]) * rs[h + e]) * ga[h + e] + be[h + e];
                dot[k] = __builtin_fmaf(gv[h + e], lcv + 0.3f * cv[e][k], dot[k]);
            }
        if (64 * (h + 4) >= D) break;
    }
    float mine = 0.0f;
#pragma unroll
    for (int k = 0; k < kNb; ++k) { const float s = wave_sum_f32(dot[k]); if (lane == k) mine = 0.3f * s; }
    if (lane < kNb) g_w8[(size_t)tok * kNb + lane] = mine;
}

// Backward of the weights themselves: w_k = r_k / S, r_k = 1 / (d_k + eps), S = sum_k r_k, d_k = |a - b_k|^2 in the reference's
// square_distance form (a = level-1 centre, b_k = its k-th nearest level-2 centre).  With G_k the incoming gradient of w_k:
//   dL/dd_k = -r_k^2 * (G_k - sum_j G_j w_j) / S,   dL/da = sum_k dL/dd_k * 2 (a - b_k),   dL/db_j = - sum over (i,k) with idx8 == j.
// One workgroup of 64 threads per sample: phase 1 thread = level-1 centre (writes g_c1, leaves dL/dd in LDS), phase 2 thread =
// level-2 centre, which walks the T*8 list in index order -- a fixed summation order, no atomics.
__global__ __launch_bounds__(64) void prop_weights_bwd_kernel(const float *__restrict__ c1, const float *__restrict__ c2,
                                                              const int32_t *__restrict__ idx8, const float *__restrict__ g_w8, float eps,
                                                              float *__restrict__ g_c1, float *__restrict__ g_c2, int T, int G2) {
    extern __shared__ float sh[];
    float *gd = sh;                                    // [T * 8]
    int *jj = reinterpret_cast<int *>(sh + (size_t)T * kNb);
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < T; i += 64) {
        const float *a = c1 + ((size_t)b * T + i) * 3;
        const float ax = a[0], ay = a[1], az = a[2];
        float rc[kNb], G[kNb], dx[kNb], dy[kNb], dz[kNb], S = 0.0f;
#pragma unroll
        for (int k = 0; k < kNb; ++k) {
            const int j = idx8[((size_t)b * T + i) * kNb + k];
            jj[i * kNb + k] = j;
            const float *q = c2 + ((size_t)b * G2 + j) * 3;
            const float bx = q[0], by = q[1], bz = q[2];
            float d = -2.0f * __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, ax * bx));     // the forward's operation order (prop_index_kernel)
            d += (ax * ax + ay * ay) + az * az;
            d += (bx * bx + by * by) + bz * bz;
            rc[k] = 1.0f / (d + eps); S += rc[k];
            dx[k] = ax - bx; dy[k] = ay - by; dz[k] = az - bz;
            G[k] = g_w8[((size_t)b * T + i) * kNb + k];
        }
        float gw = 0.0f;
#pragma unroll
        for (int k = 0; k < kNb; ++k) gw = __builtin_fmaf(G[k], rc[k] / S, gw);
        float gx = 0.0f, gy = 0.0f, gz = 0.0f;
#pragma unroll
        for (int k = 0; k < kNb; ++k) {
            const float t = -(rc[k] * rc[k]) * ((G[k] - gw) / S) * 2.0f;
            gd[i * kNb + k] = t;
            gx = __builtin_fmaf(t, dx[k], gx); gy = __builtin_fmaf(t, dy[k], gy); gz = __builtin_fmaf(t, dz[k], gz);
        }
        float *o = g_c1 + ((size_t)b * T + i) * 3;
        o[0] = gx; o[1] = gy; o[2] = gz;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < G2; j += 64) {
        const float *q = c2 + ((size_t)b * G2 + j) * 3;
        const float bx = q[0], by = q[1], bz = q[2];
        float gx = 0.0f, gy = 0.0f, gz = 0.0f;
        for (int e = 0; e < T * kNb; ++e) {
            if (jj[e] != j) continue;
            const float *a = c1 + ((size_t)b * T + e / kNb) * 3;
            const float t = gd[e];
            gx = __builtin_fmaf(-t, a[0] - bx, gx); gy = __builtin_fmaf(-t, a[1] - by, gy); gz = __builtin_fmaf(-t, a[2] - bz, gz);
        }
        float *o = g_c2 + ((size_t)b * G2 + j) * 3;
        o[0] = gx; o[1] = gy; o[2] = gz;
    }
}

}  // namespace

extern "C" int upp_prop_index(const float *c1, const float *c2, const int64_t *i1, const int64_t *i2, int gather_idx, int B, int T,
                              int G2, int Lp, int off, float eps, int32_t *i1a, int32_t *i2a, int32_t *idx8, float *w8, void *stream) {
    if (!c1 || !c2 || !i1 || !i2 || !i1a || !i2a || !idx8 || !w8 || B < 1 || T < 1 || G2 < kNb || Lp < 1 || off < 0 || off >= Lp)
        return UPP_E_BADARG;
    if (G2 > 64) return UPP_E_RANGE;
    const long long waves_idx = ((long long)B * G2 * (kNb + 1) + 63) / 64, waves = waves_idx > (long long)B * T ? waves_idx : (long long)B * T;
    hipLaunchKernelGGL(prop_index_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, c1, c2, i1, i2, gather_idx, B, T,
                       G2, Lp, off, eps, i1a, i2a, idx8, w8);
    return upp_launch_status();
}

extern "C" int upp_prop_pool_fwd(const float *X, const int32_t *i1, const float *u, float keep, float *pooled, uint8_t *amax,
                                 int groups, int D, void *stream) {
    if (!X || !i1 || !pooled || !amax || groups < 1 || D < 1) return UPP_E_BADARG;
    if (D > 64 * kMaxE) return UPP_E_RANGE;
    hipLaunchKernelGGL(prop_pool_fwd_kernel, rows_grid(groups), dim3(256), 0, (hipStream_t)stream, X, i1, u, keep, pooled, amax, groups, D);
    return upp_launch_status();
}

extern "C" int upp_prop_pool_bwd(const float *g_pooled, const uint8_t *amax, const int32_t *i1, const float *u, float keep,
                                 float *g_X, int rows, int groups, int D, void *stream) {
    if (!g_pooled || !amax || !i1 || !g_X || rows < 1 || groups < 1 || D < 1) return UPP_E_BADARG;
    if (D > 64 * kMaxE) return UPP_E_RANGE;
    const size_t lds = ((size_t)groups * kNb + 4 * kHitCap) * sizeof(int32_t);
    if (lds > 64 * 1024) return UPP_E_RANGE;    // the index list must fit LDS (groups <= ~1900)
    hipLaunchKernelGGL(prop_pool_bwd_kernel, rows_grid(rows), dim3(256), lds, (hipStream_t)stream, g_pooled, amax, i1, u, keep, g_X, rows, groups, D);
    return upp_launch_status();
}

extern "C" int upp_prop_interp_fwd(const float *X, const float *lc, const int32_t *i2, const int32_t *idx8, const float *w8,
                                   float *out, int B, int Lp, int T, int G2, int D, void *stream) {
    if (!X || !lc || !i2 || !idx8 || !w8 || !out || B < 1 || Lp < T || T < 1 || G2 < 1 || D < 1) return UPP_E_BADARG;
    if (D > 64 * kMaxE) return UPP_E_RANGE;
    hipLaunchKernelGGL(prop_interp_fwd_kernel, rows_grid((long long)B * Lp), dim3(256), 0, (hipStream_t)stream, X, lc, i2, idx8, w8, out, B, Lp, T, G2, D);
    return upp_launch_status();
}

extern "C" int upp_prop_interp_bwd(const float *g_out, const int32_t *i2, const int32_t *idx8, const float *w8, float *g_c2,
                                   float *g_X, int B, int Lp, int T, int G2, int D, void *stream) {
    if (!g_out || !i2 || !idx8 || !w8 || !g_c2 || !g_X || B < 1 || Lp < T || T < 1 || G2 < 1 || D < 1) return UPP_E_BADARG;
    if (D > 64 * kMaxE) return UPP_E_RANGE;
    const size_t lds = ((size_t)2 * T * kNb + 4 * kHitCap) * sizeof(int32_t);
    if (G2 % 4 != 0 || lds > 64 * 1024) return UPP_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(prop_interp_bwd_c2_kernel, rows_grid((long long)B * G2), dim3(256), lds, st, g_out, idx8, w8, g_c2, B, Lp, T, G2, D);
    hipLaunchKernelGGL(prop_interp_bwd_x_kernel, rows_grid((long long)B * Lp), dim3(256), 0, st, g_out, g_c2, i2, g_X, B * Lp, B * G2, D);
    return upp_launch_status();
}

extern "C" int upp_csr_build(const int32_t *keys, int n, int seg_len, int seg_rows, int rows, int32_t *start, int32_t *perm,
                             void *stream) {
    if (!keys || !start || !perm || n < 1 || rows < 1 || seg_len < 1 || seg_rows < 0) return UPP_E_BADARG;
    const size_t lds = ((size_t)rows + kCsrThreads) * sizeof(int32_t);
    if (lds > 64 * 1024) return UPP_E_RANGE;    // counts of every row live in LDS (rows <= 15360)
    const size_t lds_perm = lds + (size_t)2 * n * sizeof(int32_t);      // + permutation and its row ids
    if (lds_perm <= 150 * 1024) {
        static std::atomic<bool> raised{false};
        if (!raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(csr_build_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (e != hipSuccess) return (int)e;
            raised = true;
        }
        hipLaunchKernelGGL(csr_build_kernel<true>, dim3(1), dim3(kCsrThreads), lds_perm, (hipStream_t)stream, keys, n, seg_len, seg_rows, rows,
                           start, perm);
    } else {
        hipLaunchKernelGGL(csr_build_kernel<false>, dim3(1), dim3(kCsrThreads), lds, (hipStream_t)stream, keys, n, seg_len, seg_rows, rows,
                           start, perm);
    }
    return upp_launch_status();
}

extern "C" long long upp_prop_part_floats(int groups, int D) {
    if (groups < 1 || D < 1) return 0;
    return (long long)((groups + 3) / 4) * 2 * D;
}

extern "C" int upp_prop_fwd(const float *X, const int32_t *i1, const float *u, float keep, const int32_t *i2, const int32_t *idx8,
                            const float *w8, const float *gamma, const float *beta, float *running_mean, float *running_var,
                            float momentum, float eps, int training, float *pooled, uint8_t *amax, float *part, float *mean,
                            float *rstd, float *out, int B, int Lp, int T, int G2, int D, void *stream) {
    if (!X || !i1 || !i2 || !idx8 || !w8 || !gamma || !beta || !pooled || !amax || !part || !mean || !rstd || !out) return UPP_E_BADARG;
    if (B < 1 || Lp < T || T < 1 || G2 < 1 || D < 1) return UPP_E_BADARG;
    if (!training && (!running_mean || !running_var)) return UPP_E_BADARG;
    if (D > 64 * kMaxE) return UPP_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    const int groups = B * G2, wgs = (groups + 3) / 4;
    hipLaunchKernelGGL(prop_pool_stats_kernel, dim3(wgs), dim3(256), 0, st, X, i1, u, keep, pooled, amax, part, groups, D);
    upp_bn_finalize_launch(part, wgs, 4, groups, D, training, momentum, eps, running_mean, running_var, mean, rstd, st);
    hipLaunchKernelGGL(prop_interp_bn_fwd_kernel, rows_grid((long long)B * Lp), dim3(256), 0, st, X, pooled, mean, rstd, gamma, beta, i2,
                       idx8, w8, out, B, Lp, T, G2, D);
    return upp_launch_status();
}

extern "C" int upp_prop_bwd(const float *g_out, const float *pooled, const uint8_t *amax, const float *mean, const float *rstd,
                            const float *gamma, const float *u, float keep, const float *w8, const int32_t *start1,
                            const int32_t *perm1, const int32_t *start2, const int32_t *perm2, const int32_t *start8,
                            const int32_t *perm8, int training, float *g_c2, float *part, float *g_gamma, float *g_beta, float *g_X,
                            int B, int Lp, int T, int G2, int D, void *stream) {
    if (!g_out || !pooled || !amax || !mean || !rstd || !gamma || !w8 || !start1 || !perm1 || !start2 || !perm2 || !start8 || !perm8 ||
        !g_c2 || !part || !g_gamma || !g_beta || !g_X)
        return UPP_E_BADARG;
    if (B < 1 || Lp < T || T < 1 || G2 < 1 || D < 1) return UPP_E_BADARG;
    if (D > 64 * kMaxE) return UPP_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    const int groups = B * G2, wgs = (groups + 3) / 4;
    hipLaunchKernelGGL(prop_c2_csr_kernel, dim3(wgs), dim3(256), 0, st, g_out, start8, perm8, w8, pooled, mean, rstd, g_c2, part, B, Lp, T,
                       G2, D);
    hipLaunchKernelGGL(prop_bn_grad_kernel, dim3((D + 63) / 64), dim3(256), 0, st, part, wgs, D, g_gamma, g_beta);
    hipLaunchKernelGGL(prop_x_csr_kernel, rows_grid((long long)B * Lp), dim3(256), 0, st, g_out, g_c2, pooled, amax, mean, rstd, gamma,
                       g_gamma, g_beta, u, keep, start1, perm1, start2, perm2, g_X, B * Lp, groups, D, training);
    return upp_launch_status();
}

extern "C" int upp_prop_w8_grad(const float *g_out, const float *X, const float *pooled, const float *mean, const float *rstd,
                                const float *gamma, const float *beta, const int32_t *i2, const int32_t *idx8, float *g_w8, int B, int Lp,
                                int T, int G2, int D, void *stream) {
    if (!g_out || !X || !pooled || !i2 || !idx8 || !g_w8 || B < 1 || Lp < T || T < 1 || G2 < 1 || D < 1) return UPP_E_BADARG;
    if (mean && (!rstd || !gamma || !beta)) return UPP_E_BADARG;
    if (D > 64 * kMaxE) return UPP_E_RANGE;
    hipLaunchKernelGGL(prop_w8_grad_kernel, dim3((unsigned)(((long long)B * T + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g_out, X, pooled,
                       mean, rstd, gamma, beta, i2, idx8, g_w8, B, Lp, T, G2, D);
    return upp_launch_status();
}

extern "C" int upp_prop_weights_bwd(const float *c1, const float *c2, const int32_t *idx8, const float *g_w8, float eps, float *g_c1,
                                    float *g_c2, int B, int T, int G2, void *stream) {
    if (!c1 || !c2 || !idx8 || !g_w8 || !g_c1 || !g_c2 || B < 1 || T < 1 || G2 < 1) return UPP_E_BADARG;
    const size_t lds = (size_t)T * kNb * (sizeof(float) + sizeof(int));
    if (lds > 64 * 1024) return UPP_E_RANGE;
    hipLaunchKernelGGL(prop_weights_bwd_kernel, dim3((unsigned)B), dim3(64), lds, (hipStream_t)stream, c1, c2, idx8, g_w8, eps, g_c1, g_c2, T, G2);
    return upp_launch_status();
}
