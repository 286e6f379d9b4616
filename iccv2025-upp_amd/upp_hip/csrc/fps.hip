// fps.hip -- furthest point sampling for gfx950.
//
// Replaces pointnet2_ops furthest_point_sampling_kernel (sampling_gpu.cu), the
// operator behind reference utils/misc.py:18.  Same results, different machine
// mapping:
//
//  * One workgroup of W wavefronts per cloud (W = 1,2,4,8).  Every lane keeps
//    its points AND their running min-distances in VGPRs for the whole call
//    (S "slots" per lane), so a round touches no memory except one broadcast
//    LDS read of the winner's coordinates: HBM traffic is the algorithmic
//    minimum (the cloud once in, the indices once out).  A round is pure
//    latency (M rounds are inherently serial), so everything is arranged to
//    shorten the dependent chain:
//      - the distance update runs on packed-f32 VALU ops, two slots per
//        instruction (v_pk_add/mul/fma_f32);
//      - the arg-max is ONE wave reduction (u32 max of the distance bits on DPP
//        row operations) + a ballot: points are dealt to lanes so that the
//        reference's tie-break order is (wave, lane, slot) lexicographic, hence
//        the winner among equal maxima is simply the first set bit of the
//        ballot and the first such slot inside that lane;
//      - with W = 1 a round has no barrier at all; with W > 1 the waves
//        exchange one 8-byte record through LDS and meet at a single barrier
//        per round (double-buffered records);
//      - winners are collected in LDS and written out coalesced at the end (a
//        global store per round would sit in front of every barrier).
//
// Tie-breaking.  The CUDA kernel runs T = min(512, 2^floor(log2 N)) threads per
// cloud; thread t scans k = t, t+T, ... with a strict '>', then a shared-memory
// tree  dists_i[a] = v[b] > v[a] ? i[b] : i[a]  folds t+s into t for
// s = T/2 .. 1.  Among equal maxima the survivor is therefore the candidate
// with the smallest  rank(k) = bitrev_log2T(k mod T) * ceil(N/T) + k div T.
// Here global lane g (= 64 * wave + lane) owns the U = T / (64 W) "virtual
// threads" whose bit-reversed ids are v = g*U .. g*U + U-1, slot s = u*Q + q
// holding point k = bitrev(v) + T*q:  rank(k) = v*Q + q = g*(U*Q) + s.
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct FpsGeom {
    int N, M;
    int T, log2T;  // CUDA block size of the reference kernel and its log2
    int Q;         // ceil(N / T): points per virtual thread
    int U;         // virtual threads per lane = max(1, T / L)
};

template <int S, int W, bool USE_LDS>
__global__ __launch_bounds__(64 * W) void fps_kernel(const float *__restrict__ xyz, int32_t *__restrict__ idx,
                                                     float *__restrict__ centers, FpsGeom g) {
    static_assert(S % 2 == 0, "slots are processed in packed pairs");
    constexpr int L = 64 * W;
    extern __shared__ float lds[];  // 2*W*2 words of wave records, a [3*N] copy of the cloud, then the M winners
    const int N = g.N, M = g.M;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const float *p = xyz + (size_t)b * N * 3;
    uint32_t *rec = reinterpret_cast<uint32_t *>(lds);
    float *cloud = lds + 4 * W;
    int *won = reinterpret_cast<int *>(cloud + 3 * N);
    // Coordinates of a point: from the LDS copy or, for clouds that do not fit LDS, straight from global memory.
    auto coord = [&](int k, int c) -> float { return USE_LDS ? cloud[k * 3 + c] : p[k * 3 + c]; };
    if (USE_LDS) {
        stage_floats(cloud, p, 3 * N, tid, L);
        __syncthreads();
    }

    f32x2 px[S / 2], py[S / 2], pz[S / 2];
    float tmp[S];
    int kk[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int u = s / g.Q, q = s - u * g.Q;
        const int v = tid * g.U + u;
        const int t = g.log2T ? (int)(__builtin_bitreverse32((uint32_t)v) >> (32 - g.log2T)) : 0;
        const int k = t + g.T * q;
        const bool valid = (u < g.U) && (v < g.T) && (k < N);
        const int kc = valid ? k : 0;
        const float x = coord(kc, 0), y = coord(kc, 1), z = coord(kc, 2);
        px[s >> 1][s & 1] = x; py[s >> 1][s & 1] = y; pz[s >> 1][s & 1] = z;
        kk[s] = kc;
        // |p|^2 <= 1e-3 (double compare, as the literal in the CUDA source is a
        // double) -> never a candidate.  A slot with tmp = -1 can never beat
        // best (init -1, strict '>') and min(d, -1) keeps it at -1.
        const float mag = sumsq3(x, y, z);
        tmp[s] = (valid && !((double)mag <= 1e-3)) ? 1e10f : -1.0f;
    }

    float x1 = coord(0, 0), y1 = coord(0, 1), z1 = coord(0, 2);
    if (tid == 0) {
        if (USE_LDS) won[0] = 0;
        else {
            idx[(size_t)b * M] = 0;
            if (centers) {
                float *c = centers + (size_t)b * M * 3;
                c[0] = x1; c[1] = y1; c[2] = z1;
            }
        }
    }
    const int lane = tid & 63, wave = tid >> 6;

    for (int j = 1; j < M; ++j) {
        float best = -1.0f;
        int bk = 0;
        const f32x2 X1 = {x1, x1}, Y1 = {y1, y1}, Z1 = {z1, z1};
#pragma unroll
        for (int h = 0; h < S / 2; ++h) {
            // sumsq3 on two slots at once: t = dy*dy; t = fma(dx,dx,t); t = fma(dz,dz,t)   (v_pk_* f32, IEEE per element)
            const f32x2 dx = px[h] - X1, dy = py[h] - Y1, dz = pz[h] - Z1;
            f32x2 t = dy * dy;
            t = __builtin_elementwise_fma(dx, dx, t);
            t = __builtin_elementwise_fma(dz, dz, t);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int s = 2 * h + e;
                const float d2 = fminf(t[e], tmp[s]);
                tmp[s] = d2;
                const bool gt = d2 > best;
                bk = gt ? kk[s] : bk;
                best = gt ? d2 : best;
            }
        }
        // distance bits are monotone for d >= 0; +1 so that "no candidate" = 0
        const uint32_t kb = best < 0.0f ? 0u : __float_as_uint(best) + 1u;
        uint32_t m = wave_max_u32(kb);
        // lanes are in rank order: the first lane that holds the maximum wins the wave
        const int wl = __builtin_ctzll(__ballot(kb == m));
        uint32_t wk = readlane_u32((uint32_t)bk, wl);
        if (W > 1) {
            uint32_t *r = rec + (j & 1) * (2 * W);
            if (lane == 0) { r[2 * wave] = m; r[2 * wave + 1] = wk; }
            __syncthreads();
            m = r[0]; wk = r[1];
#pragma unroll
            for (int w = 1; w < W; ++w) {            // waves are in rank order too: strict '>' keeps the lower wave
                const uint32_t mw = r[2 * w], kw = r[2 * w + 1];
                const bool take = mw > m;
                m = take ? mw : m;
                wk = take ? kw : wk;
            }
        }
        const int old = (m == 0u) ? 0 : (int)wk;
        x1 = coord(old, 0); y1 = coord(old, 1); z1 = coord(old, 2);
        if (tid == 0) {
            if (USE_LDS) won[j] = old;
            else {
                idx[(size_t)b * M + j] = old;
                if (centers) {
                    float *c = centers + ((size_t)b * M + j) * 3;
                    c[0] = x1; c[1] = y1; c[2] = z1;
                }
            }
        }
    }
    if (USE_LDS) {
        __syncthreads();
        for (int j = tid; j < M; j += L) idx[(size_t)b * M + j] = won[j];
        if (centers)
            for (int e = tid; e < 3 * M; e += L) centers[(size_t)b * M * 3 + e] = cloud[won[e / 3] * 3 + (e % 3)];
    }
}

int ilog2_floor(int v) { int l = 0; while ((1 << (l + 1)) <= v) ++l; return l; }

// pointnet2_ops cuda_utils.h opt_n_threads(): through double log, as upstream.
int fps_block_size(int n) {
    const int pow_2 = (int)(log((double)n) / log(2.0));
    int t = 1 << pow_2;
    if (t > 512) t = 512;
    if (t < 1) t = 1;
    return t;
}

constexpr int kFpsLdsBytes = 160 * 1024 - 256;  // cloud copy + winner list + wave records must fit the CU's LDS

template <int S, int W>
int launch(const float *xyz, int32_t *idx, float *centers, int B, const FpsGeom &g, hipStream_t st) {
    const size_t lds_bytes = (size_t)(4 * W + 3 * g.N + g.M) * 4;
    if (lds_bytes <= (size_t)kFpsLdsBytes) {
        if (lds_bytes > 64 * 1024) {
            static std::atomic<bool> raised{false};  // one attribute call per instantiation
            if (!raised) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fps_kernel<S, W, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, kFpsLdsBytes);
                if (e != hipSuccess) return (int)e;
                raised = true;
            }
        }
        hipLaunchKernelGGL((fps_kernel<S, W, true>), dim3(B), dim3(64 * W), lds_bytes, st, xyz, idx, centers, g);
    } else {
        hipLaunchKernelGGL((fps_kernel<S, W, false>), dim3(B), dim3(64 * W), (size_t)(4 * W) * 4, st, xyz, idx, centers, g);
    }
    return upp_launch_status();
}

template <int W>
int dispatch_s(int slots, const float *xyz, int32_t *idx, float *centers, int B, const FpsGeom &g, hipStream_t st) {
    if (slots <= 2) return launch<2, W>(xyz, idx, centers, B, g, st);
    if (slots <= 4) return launch<4, W>(xyz, idx, centers, B, g, st);
    if (slots <= 6) return launch<6, W>(xyz, idx, centers, B, g, st);
    if (slots <= 8) return launch<8, W>(xyz, idx, centers, B, g, st);
    if (slots <= 10) return launch<10, W>(xyz, idx, centers, B, g, st);
    if (slots <= 12) return launch<12, W>(xyz, idx, centers, B, g, st);
    if (slots <= 16) return launch<16, W>(xyz, idx, centers, B, g, st);
    if (slots <= 20) return launch<20, W>(xyz, idx, centers, B, g, st);
    if (slots <= 24) return launch<24, W>(xyz, idx, centers, B, g, st);
    if (slots <= 32) return launch<32, W>(xyz, idx, centers, B, g, st);
    if (slots <= 64) return launch<64, W>(xyz, idx, centers, B, g, st);
    return UPP_E_RANGE;
}

}  // namespace

extern "C" int upp_fps_ex(const float *xyz, int32_t *idx, float *centers, int B, int N, int M, int waves, void *stream) {
    if (!xyz || !idx || B < 0 || N < 1 || M < 1) return UPP_E_BADARG;
    if (waves != 0 && waves != 1 && waves != 2 && waves != 4 && waves != 8) return UPP_E_BADARG;
    if (N > 32768) return UPP_E_RANGE;  // 15-bit point ids, 64 slots x 512 lanes
    if (B == 0) return 0;
    FpsGeom g;
    g.N = N; g.M = M;
    g.T = fps_block_size(N);
    g.log2T = ilog2_floor(g.T);
    g.Q = (N + g.T - 1) / g.T;
    // waves per cloud: never more lanes than virtual threads
    int W = waves ? waves : (N <= 128 ? 1 : (N <= 512 ? 2 : 4));
    while (W > 1 && 64 * W > g.T) W >>= 1;
    while (W < 8 && 64 * W < g.T && (g.T / (64 * W)) * g.Q > 64) W <<= 1;  // at most 64 slots per lane
    const int L = 64 * W;
    g.U = g.T / L > 0 ? g.T / L : 1;
    const int slots = g.U * g.Q;
    hipStream_t st = (hipStream_t)stream;
    switch (W) {
        case 1: return dispatch_s<1>(slots, xyz, idx, centers, B, g, st);
        case 2: return dispatch_s<2>(slots, xyz, idx, centers, B, g, st);
        case 4: return dispatch_s<4>(slots, xyz, idx, centers, B, g, st);
        default: return dispatch_s<8>(slots, xyz, idx, centers, B, g, st);
    }
}

extern "C" int upp_fps(const float *xyz, int32_t *idx, float *centers, int B, int N, int M, void *stream) {
    return upp_fps_ex(xyz, idx, centers, B, N, M, 0, stream);
}
