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
//      - the distance update is written on two-slot vectors; since round 4 the library is built with -packed-fp32-ops OFF
//        (build.py: v_pk_add/mul/fma_f32 with op_sel misbehaved beside bf16-MFMA workgroups), so the compiler
//        emits scalar f32 VALU ops for them (347 -> 389 us at 1228 -> 1024; tests/test_abi.py asserts no v_pk_*_f32 in any code object);
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

// Wave-wide max of a signed 32-bit key, result in an SGPR: six v_max_i32 with the DPP permute ON the max instruction (quad swaps, half-row and
// row mirrors, then row_bcast 15 / 31 carry the row maxima down to lane 63) and one v_readlane -- 13 instructions with their hazard nops;
// common.h's wave_max_u32 (update_dpp + max per step, four readlanes, SALU / VALU combine) compiles to 24, and a lone wave issues one
// instruction per ~10 cycles (tools/micro/clock_probe.cpp): the round loop below is priced by its instruction COUNT.
__device__ __forceinline__ int wave_max_i32(int v) {
    asm volatile("s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}

struct FpsGeom {
    int N, M;
    int T, log2T;  // CUDA block size of the reference kernel and its log2
    int Q;         // ceil(N / T): points per virtual thread
    int U;         // virtual threads per lane = max(1, T / L)
};

// CPW clouds per workgroup (round 5): the W waves of cloud blockIdx.x * CPW + c are waves c W ... c W + W - 1, with their own LDS region;
// the per-round barrier spans the workgroup (every cloud runs the same M - 1 rounds).  Nothing else changes -- a cloud's waves never
// look at another cloud's records -- so the indices are those of CPW = 1.  Why: 32 clouds on 32 CUs keep a one-round GEMM of the other
// stream (228 workgroups on 256 CUs) from fitting, and every such GEMM then runs two rounds for as long as FPS is resident (back-end
// chain 3.26 -> 3.61 ms beside 0.8 ms of FPS: tools/micro/fps_beside_backend.py); 8 CUs leave room.
template <int S, int W, bool USE_LDS, int CPW = 1>
__global__ __launch_bounds__(64 * W * CPW) void fps_kernel(const float *__restrict__ xyz, int32_t *__restrict__ idx,
                                                           float *__restrict__ centers, FpsGeom g) {
    static_assert(S % 2 == 0, "slots are processed in packed pairs");
    static_assert(CPW == 1 || USE_LDS, "several clouds per workgroup: the LDS form only");
    constexpr int L = 64 * W;
    extern __shared__ float lds_all[];  // per cloud: 2*W*2 words of wave records, a [3*N] copy of the cloud, then the M winners
    const int N = g.N, M = g.M;
    const int cslot = CPW > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / L) : 0;
    const int tid = (int)threadIdx.x - cslot * L;
    const int b = blockIdx.x * CPW + cslot;
    float *lds = lds_all + (size_t)cslot * (size_t)((4 * W + 3 * N + M + 3) / 4 * 4);
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
    int tmp[S];          // running min-distance of the slot as its IEEE bit pattern: every value is >= +0 or exactly -1.0f (never a
                         // candidate), for which signed-integer order IS float order -- v_min_i32 / v_max3_i32 need no NaN canonicalisation
                         // (fminf costs a v_max x,x,x per operand) and no compare + select pairs
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
        tmp[s] = __float_as_int((valid && !((double)mag <= 1e-3)) ? 1e10f : -1.0f);
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

    constexpr int kNone = (int)0xBF800000u;                  // bits of -1.0f: negative as an integer, below every real distance
    for (int j = 1; j < M; ++j) {
        const f32x2 X1 = {x1, x1}, Y1 = {y1, y1}, Z1 = {z1, z1};
        int d2[S];
#pragma unroll
        for (int h = 0; h < S / 2; ++h) {
            // sumsq3 on two slots at once: t = dy*dy; t = fma(dx,dx,t); t = fma(dz,dz,t)   (two-slot vectors; scalar f32 VALU in the shipped build, IEEE per element)
#ifdef UPP_FPS_DIAG_SCALAR
            f32x2 t;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float dx = px[h][e] - x1, dy = py[h][e] - y1, dz = pz[h][e] - z1;
                float tt = dy * dy;
                asm volatile("" : "+v"(tt));
                tt = __builtin_fmaf(dx, dx, tt);
                asm volatile("" : "+v"(tt));
                tt = __builtin_fmaf(dz, dz, tt);
                t[e] = tt;
            }
#else
            const f32x2 dx = px[h] - X1, dy = py[h] - Y1, dz = pz[h] - Z1;
            f32x2 t = dy * dy;
            t = __builtin_elementwise_fma(dx, dx, t);
            t = __builtin_elementwise_fma(dz, dz, t);
#endif
            d2[2 * h] = min(__float_as_int(t[0]), tmp[2 * h]); d2[2 * h + 1] = min(__float_as_int(t[1]), tmp[2 * h + 1]);
            tmp[2 * h] = d2[2 * h]; tmp[2 * h + 1] = d2[2 * h + 1];
        }
        int best = d2[0];
#pragma unroll
        for (int sl = 1; sl < S; ++sl) best = max(best, d2[sl]);         // (v_max3_i32 pairs)
        best = max(best, kNone);
        int bk = 0;                                               // the FIRST slot that holds the maximum (strict '>' of the reference scan)
        int m_wave;
#ifdef UPP_FPS_DIAG_NOASM
        if constexpr (false) {
#else
        if constexpr (S == 6 || S == 4) {
#endif
            // The wave reduction's six DPP steps need two wait states each before the next one reads the register, and every
            // v_cmp -> v_cndmask pair of the slot search two as well; an s_nop costs a lone wave 8 cycles (tools/micro/nop_probe.cpp), as
            // much as a real instruction.  So the two chains are interleaved by hand: each hazard shadow holds two instructions of the
            // other chain -- compares first (one SGPR pair per slot), selects in descending slot order afterwards.
            int v = best;
            unsigned long long e0, e1, e2, e3, e4, e5;           // (ONE asm statement: between separate ones hipcc puts an s_nop of its own)
#define UPP_DPP(CTRL) "v_max_i32_dpp %[v], %[v], %[v] " CTRL " row_mask:0xf bank_mask:0xf\n\t"
#define UPP_DPPB(CTRL) "v_max_i32_dpp %[v], %[v], %[v] " CTRL " bank_mask:0xf\n\t"
#define UPP_CMP(E, D) "v_cmp_eq_u32_e64 %[" E "], %[" D "], %[b]\n\t"
#define UPP_SEL(E, K) "v_cndmask_b32_e64 %[bk], %[bk], %[" K "], %[" E "]\n\t"
            if constexpr (S == 6) {
                asm volatile("s_nop 1\n\t"
                             UPP_DPP("quad_perm:[1,0,3,2]") UPP_CMP("e5", "d5") UPP_CMP("e4", "d4")
                             UPP_DPP("quad_perm:[2,3,0,1]") UPP_CMP("e3", "d3") UPP_CMP("e2", "d2")
                             UPP_DPP("row_half_mirror") UPP_CMP("e1", "d1") UPP_CMP("e0", "d0")
                             UPP_DPP("row_mirror") UPP_SEL("e5", "k5") UPP_SEL("e4", "k4")
                             UPP_DPPB("row_bcast:15 row_mask:0xa") UPP_SEL("e3", "k3") UPP_SEL("e2", "k2")
                             UPP_DPPB("row_bcast:31 row_mask:0xc") UPP_SEL("e1", "k1") UPP_SEL("e0", "k0")
                             : [v] "+&v"(v), [bk] "+&v"(bk), [e0] "=&s"(e0), [e1] "=&s"(e1), [e2] "=&s"(e2), [e3] "=&s"(e3), [e4] "=&s"(e4), [e5] "=&s"(e5)
                             : [b] "v"(best), [d0] "v"(d2[0]), [d1] "v"(d2[1]), [d2] "v"(d2[2]), [d3] "v"(d2[3]), [d4] "v"(d2[S > 4 ? 4 : 0]),
                               [d5] "v"(d2[S > 5 ? 5 : 0]), [k0] "v"(kk[0]), [k1] "v"(kk[1]), [k2] "v"(kk[2]), [k3] "v"(kk[3]),
                               [k4] "v"(kk[S > 4 ? 4 : 0]), [k5] "v"(kk[S > 5 ? 5 : 0]));
            } else {
                asm volatile("s_nop 1\n\t"
                             UPP_DPP("quad_perm:[1,0,3,2]") UPP_CMP("e3", "d3") UPP_CMP("e2", "d2")
                             UPP_DPP("quad_perm:[2,3,0,1]") UPP_CMP("e1", "d1") UPP_CMP("e0", "d0")
                             UPP_DPP("row_half_mirror") UPP_SEL("e3", "k3") UPP_SEL("e2", "k2")
                             UPP_DPP("row_mirror") UPP_SEL("e1", "k1") UPP_SEL("e0", "k0")
                             UPP_DPPB("row_bcast:15 row_mask:0xa") "s_nop 1\n\t"
                             UPP_DPPB("row_bcast:31 row_mask:0xc") "s_nop 1\n\t"
                             : [v] "+&v"(v), [bk] "+&v"(bk), [e0] "=&s"(e0), [e1] "=&s"(e1), [e2] "=&s"(e2), [e3] "=&s"(e3)
                             : [b] "v"(best), [d0] "v"(d2[0]), [d1] "v"(d2[1]), [d2] "v"(d2[2]), [d3] "v"(d2[3]), [k0] "v"(kk[0]), [k1] "v"(kk[1]),
                               [k2] "v"(kk[2]), [k3] "v"(kk[3]));
                (void)e4; (void)e5;
            }
#undef UPP_DPP
#undef UPP_DPPB
#undef UPP_CMP
#undef UPP_SEL
            m_wave = __builtin_amdgcn_readlane(v, 63);
        } else {
#pragma unroll
            for (int sl = S - 1; sl >= 0; --sl) bk = d2[sl] == best ? kk[sl] : bk;
#ifdef UPP_FPS_DIAG_NOASM
            m_wave = (int)(wave_max_u32((uint32_t)best ^ 0x80000000u) ^ 0x80000000u);
#else
            m_wave = wave_max_i32(best);
#endif
        }
        // lanes are in rank order: the first lane that holds the maximum wins the wave
        const int wl = __builtin_ctzll(__ballot(best == m_wave));
        int m = m_wave;
        uint32_t wk = readlane_u32((uint32_t)bk, wl);
        if (W > 1) {
            uint32_t *r = rec + (j & 1) * (2 * W);
            if (lane == 0) { r[2 * wave] = (uint32_t)m; r[2 * wave + 1] = wk; }
            __syncthreads();
#ifdef UPP_FPS_DIAG_NOASM
            if constexpr (false) {
#else
            if constexpr (W == 4) {
#endif
                // max of the four keys, then the LOWEST wave that holds it (waves are in rank order), all compares ahead of all selects:
                // no hazard nops; a key < 0 (no candidate anywhere) selects point 0
                const uint4 ra = *reinterpret_cast<const uint4 *>(r), rb = *reinterpret_cast<const uint4 *>(r + 4);
                unsigned long long q0, q1, q2, qn;
                int mm;
                uint32_t sel = rb.w;                              // wave 3's point unless a lower wave holds the maximum
                asm volatile("v_max3_i32 %[m], %[r0], %[r1], %[r2]\n\t"
                             "v_max_i32_e32 %[m], %[m], %[r3]\n\t"
                             "v_cmp_eq_u32_e64 %[q2], %[r2], %[m]\n\t"
                             "v_cmp_eq_u32_e64 %[q1], %[r1], %[m]\n\t"
                             "v_cmp_eq_u32_e64 %[q0], %[r0], %[m]\n\t"
                             "v_cmp_gt_i32_e64 %[qn], 0, %[m]\n\t"
                             "v_cndmask_b32_e64 %[s], %[s], %[k2], %[q2]\n\t"
                             "v_cndmask_b32_e64 %[s], %[s], %[k1], %[q1]\n\t"
                             "v_cndmask_b32_e64 %[s], %[s], %[k0], %[q0]\n\t"
                             "v_cndmask_b32_e64 %[s], %[s], 0, %[qn]\n\t"
                             : [m] "=&v"(mm), [s] "+v"(sel), [q0] "=&s"(q0), [q1] "=&s"(q1), [q2] "=&s"(q2), [qn] "=&s"(qn)
                             : [r0] "v"(ra.x), [r1] "v"(ra.z), [r2] "v"(rb.x), [r3] "v"(rb.z), [k0] "v"(ra.y), [k1] "v"(ra.w), [k2] "v"(rb.y));
                m = 0;                                            // (the "no candidate" case is already folded into sel)
                wk = sel;
            } else {
                m = (int)r[0]; wk = r[1];
#pragma unroll
                for (int w = 1; w < W; ++w) {        // waves are in rank order too: strict '>' keeps the lower wave
                    const int mw = (int)r[2 * w];
                    const uint32_t kw = r[2 * w + 1];
                    const bool take = mw > m;
                    m = take ? mw : m;
                    wk = take ? kw : wk;
                }
            }
        }
        const int old = m < 0 ? 0 : (int)wk;                      // (no candidate at all: every slot at -1)
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
int launch(const float *xyz, int32_t *idx, float *centers, int B, const FpsGeom &g, int form, hipStream_t st) {
    const size_t lds_bytes = (size_t)(4 * W + 3 * g.N + g.M) * 4;
#ifdef UPP_FPS_DIAG_NOLDS
    if (false) {
#else
    if (lds_bytes <= (size_t)kFpsLdsBytes) {
#endif
        if (lds_bytes > 64 * 1024) {
            static std::atomic<bool> raised{false};  // one attribute call per instantiation
            if (!raised) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fps_kernel<S, W, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, kFpsLdsBytes);
                if (e != hipSuccess) return (int)e;
                raised = true;
            }
        }
        // clouds per workgroup: as many of {4, 2} as divide B and fit 1,024 threads and the LDS (W = 4, the shapes of the recipes: S <= 8)
        if constexpr (W == 4 && S <= 8) {
            const size_t stride = (size_t)((4 * W + 3 * g.N + g.M + 3) / 4 * 4) * 4;
            const int want = (form & 0xF) ? (form & 0xF) : 1;                          // (the caller's form: upp_fps_ex `waves` bits 8-12; default one cloud per workgroup)
            const bool excl = (form & 0xF) && (form & 0x10) != 0;
            if (want >= 4 && B % 4 == 0 && 4 * stride <= (size_t)kFpsLdsBytes) {
                static std::atomic<bool> raised4{false};
                if (!raised4) {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fps_kernel<S, W, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, kFpsLdsBytes);
                    if (e != hipSuccess) return (int)e;
                    raised4 = true;
                }
                hipLaunchKernelGGL((fps_kernel<S, W, true, 4>), dim3(B / 4), dim3(64 * W * 4), excl ? (size_t)kFpsLdsBytes : 4 * stride, st, xyz, idx, centers, g);
                return upp_launch_status();
            }
            if (want >= 2 && B % 2 == 0 && 2 * stride <= (size_t)kFpsLdsBytes) {
                static std::atomic<bool> raised2{false};
                if (!raised2) {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fps_kernel<S, W, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, kFpsLdsBytes);
                    if (e != hipSuccess) return (int)e;
                    raised2 = true;
                }
                hipLaunchKernelGGL((fps_kernel<S, W, true, 2>), dim3(B / 2), dim3(64 * W * 2), excl ? (size_t)kFpsLdsBytes : 2 * stride, st, xyz, idx, centers, g);
                return upp_launch_status();
            }
        }
        hipLaunchKernelGGL((fps_kernel<S, W, true>), dim3(B), dim3(64 * W), lds_bytes, st, xyz, idx, centers, g);
    } else {
        hipLaunchKernelGGL((fps_kernel<S, W, false>), dim3(B), dim3(64 * W), (size_t)(4 * W) * 4, st, xyz, idx, centers, g);
    }
    return upp_launch_status();
}

template <int W>
int dispatch_s(int slots, const float *xyz, int32_t *idx, float *centers, int B, const FpsGeom &g, int form, hipStream_t st) {
    if (slots <= 2) return launch<2, W>(xyz, idx, centers, B, g, form, st);
    if (slots <= 4) return launch<4, W>(xyz, idx, centers, B, g, form, st);
    if (slots <= 6) return launch<6, W>(xyz, idx, centers, B, g, form, st);
    if (slots <= 8) return launch<8, W>(xyz, idx, centers, B, g, form, st);
    if (slots <= 10) return launch<10, W>(xyz, idx, centers, B, g, form, st);
    if (slots <= 12) return launch<12, W>(xyz, idx, centers, B, g, form, st);
    if (slots <= 16) return launch<16, W>(xyz, idx, centers, B, g, form, st);
    if (slots <= 20) return launch<20, W>(xyz, idx, centers, B, g, form, st);
    if (slots <= 24) return launch<24, W>(xyz, idx, centers, B, g, form, st);
    if (slots <= 32) return launch<32, W>(xyz, idx, centers, B, g, form, st);
    if (slots <= 64) return launch<64, W>(xyz, idx, centers, B, g, form, st);
    return UPP_E_RANGE;
}

}  // namespace

extern "C" int upp_fps_ex(const float *xyz, int32_t *idx, float *centers, int B, int N, int M, int waves, void *stream) {
    if (!xyz || !idx || B < 0 || N < 1 || M < 1) return UPP_E_BADARG;
    // bits 8-11 of `waves`: clouds per workgroup (0: the library's default form, 1 / 2 / 4), bit 12: the launch reserves its CUs' whole LDS
    const int cpw = (waves >> 8) & 0xF;
    if ((waves & ~0x1FFF) || (cpw != 0 && cpw != 1 && cpw != 2 && cpw != 4)) return UPP_E_BADARG;
    const int form = cpw | (((waves >> 12) & 1) << 4);
    waves &= 0xFF;
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
        case 1: return dispatch_s<1>(slots, xyz, idx, centers, B, g, form, st);
        case 2: return dispatch_s<2>(slots, xyz, idx, centers, B, g, form, st);
        case 4: return dispatch_s<4>(slots, xyz, idx, centers, B, g, form, st);
        default: return dispatch_s<8>(slots, xyz, idx, centers, B, g, form, st);
    }
}

extern "C" int upp_fps(const float *xyz, int32_t *idx, float *centers, int B, int N, int M, void *stream) {
    return upp_fps_ex(xyz, idx, centers, B, N, M, 0, stream);
}
