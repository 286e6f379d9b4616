// wgrad_sb.hip -- the grouped weight gradients  dW_p (N_p,K_p) = G_p^T . X_p  of linear_rt.hip (reference: AddmmBackward's second GEMM of
// every trainable nn.Linear / 1x1 Conv1d -- models/Point_MAE_cp.py:369-465, models/Point_MAE_unify_segment.py:420-433) at f32 accuracy
// on the BF16 matrix pipe, the arithmetic of linear_sb.hip:
//     x = x1 + x2 + x3 (three bf16 terms, exact),   g x = g1 x1 + (g1 x2 + g2 x1) + (g2 x2 + g1 x3 + g3 x1)  + [<= 2^-25 |g x|, dropped]
// Both operands are activations: nothing can be split ahead of the launch, and the contraction runs over the ROWS of both matrices, so
// an MFMA fragment (8 consecutive values of the contraction index for one output row / column) is a COLUMN piece of G / X.
//   * a workgroup owns one work unit (WgGroup, as the f32 kernel: a tile of dW x a run of rows; 256 x 256 tiles on 8 waves, or 128 x 128
//     on 4 waves with two workgroups per CU) and walks the run 16 rows at a time: 16 x [G columns | X columns] f32 arrive in registers by
//     16-byte loads (issued a whole step ahead), are split there (11 VALU instructions per pair) and leave as three bf16 planes of 16
//     rows into one of two LDS stages;
//   * the fragments come back TRANSPOSED: ds_read_b64_tr_b16 hands each lane 4 consecutive rows of its column, two reads per plane and
//     32-column block.  Row pitch = 64 B mod 256 B (16 banks mod 64): the 4 rows x 64 B of a 32-lane half cover the 64 banks once;
//   * per step and wave of the wide tile: 36 transposed reads, 48 v_mfma_f32_32x32x16_bf16 (2 x 4 blocks x 6 products), 12 ds_write_b64,
//     4 loads, one barrier.
// Each output is a fixed-order sum (rows ascending in steps of 16, six products per step): deterministic, not bit-comparable with a
// scalar chain -- parity by tolerance against float64 like linear_sb.hip (tests/test_gpu_linear_sb.py); the exact-f32 kernel stays.
#include "common.h"
#include <type_traits>

namespace {

#include "linear_shared.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));        // v_cvt_pk_bf16_f32 (round to nearest even)
}

// 4 f32 -> 4 bf16 of each plane (x - x1 and (x - x1) - x2 are exact in f32; x3 needs no rounding)
__device__ __forceinline__ void split4(const f32x4 x, u32x2 &p1, u32x2 &p2, u32x2 &p3) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float x0 = x[2 * q], x1 = x[2 * q + 1];
        const uint32_t u = pack_bf16(x0, x1);
        const float r0 = x0 - __uint_as_float(u << 16), r1 = x1 - __uint_as_float(u & 0xFFFF0000u);
        const uint32_t v = pack_bf16(r0, r1);
        const float t0 = r0 - __uint_as_float(v << 16), t1 = r1 - __uint_as_float(v & 0xFFFF0000u);
        p1[q] = u; p2[q] = v; p3[q] = pack_bf16(t0, t1);
    }
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// WM x WN waves of BA x BB blocks (32 x 32 each): a (32 WM BA) x (32 WN BB) tile of dW per workgroup.
template <int WM, int WN, int BA, int BB>
struct WsGeom {
    static constexpr int TN = 32 * WM * BA, TK = 32 * WN * BB, T = 64 * WM * WN;
    static constexpr int PIECES = (TN + TK) / 4;          // 16-byte pieces of one [G | X] row
    static constexpr int RPP = T / PIECES;                // rows per load pass
    static constexpr int PASSES = 16 / RPP;
    static constexpr int PITCH = (TN + TK) * 2 + 64;      // bytes of one LDS row (bf16): = 64 B mod 256 B
    static constexpr int PLANE = 16 * PITCH, STAGE = 3 * PLANE;
    static_assert(T % PIECES == 0 && 16 % RPP == 0 && PITCH % 256 == 64, "geometry");
};

template <int WM, int WN, int BA, int BB, int OCC>
__global__ __launch_bounds__(64 * WM * WN, OCC) void wgrad_sb_kernel(WgGroup g) {
    using Z = WsGeom<WM, WN, BA, BB>;
    constexpr int TN = Z::TN, TK = Z::TK, PITCH = Z::PITCH, PLANE = Z::PLANE, STAGE = Z::STAGE, PASSES = Z::PASSES, RPP = Z::RPP;
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    const int u = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    int p = 0;
    while (u >= g.unit0[p + 1]) ++p;                // (scalar) at most kMaxWgProblems steps
    const int local = u - g.unit0[p], tiles = g.tiles[p];
    const int split = local / tiles, tile = local - split * tiles;
    const int tn = tile / g.tiles_k[p], tk = tile - tn * g.tiles_k[p];
    const int n0 = tn * TN, k0 = tk * TK;
    const int M = g.M[p], N = g.N[p], K = g.K[p];
    const int ms = split * g.rows[p], me = min(M, ms + g.rows[p]);
    const int nst = (me - ms + 15) >> 4;
    const int wm = wave / WN, wn = wave - wm * WN;

    // ---- operand stream: thread -> 16-byte piece `piece` of the [G | X] row, rows rg + RPP i of a step.  Columns beyond N / K are
    // clamped (their outputs are never stored); rows beyond the run are loaded from a clamped row and zeroed.
    const int piece = (int)threadIdx.x % Z::PIECES, rg = (int)threadIdx.x / Z::PIECES;
    const bool isG = piece < TN / 4;
    const int col = isG ? min(n0 + 4 * piece, N - 4) : min(k0 + 4 * (piece - TN / 4), K - 4);
    const float *src = (isG ? g.G[p] : g.X[p]) + col;
    const long long ld = isG ? g.ldg[p] : g.ldx[p];
    f32x4 raw[PASSES];
    auto load = [&](int c) {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int row = min(ms + 16 * c + rg + RPP * i, M - 1);
            raw[i] = *reinterpret_cast<const f32x4 *>(src + (long long)row * ld);
        }
    };
    char *wbase = lds + rg * PITCH + piece * 8;
    auto split_store = [&](int c) {
        char *wb = wbase + (c & 1) * STAGE;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            f32x4 v = raw[i];
            if (ms + 16 * c + rg + RPP * i >= me) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            u32x2 p1, p2, p3;
            split4(v, p1, p2, p3);
            *reinterpret_cast<u32x2 *>(wb + RPP * i * PITCH) = p1;
            *reinterpret_cast<u32x2 *>(wb + RPP * i * PITCH + PLANE) = p2;
            *reinterpret_cast<u32x2 *>(wb + RPP * i * PITCH + 2 * PLANE) = p3;
        }
    };

    // ---- transposed fragment reads: 16-lane group grp = lane / 16 takes the 4 rows 8 (grp / 2) + 4 e + {0..3} of the 16 columns
    // 16 (grp & 1) .. + 15 of a 32-column block; lane 4 q + pp of the group supplies the address of row q, columns 4 pp .. 4 pp + 3.
    const int grp = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
    const char *rbase = lds + (8 * (grp >> 1) + q4) * PITCH + (16 * (grp & 1) + 4 * pp) * 2;
    const char *rG = rbase + wm * (BA * 64), *rX = rbase + TN * 2 + wn * (BB * 64);
    auto frag = [&](const char *base, int off) -> s16x8 {
        typedef __attribute__((address_space(3))) s16x4 *lp;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base + off + 4 * PITCH));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    f32x16 acc[BA][BB];
#pragma unroll
    for (int a = 0; a < BA; ++a)
#pragma unroll
        for (int b = 0; b < BB; ++b)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[a][b][t] = 0.0f;

    // One step, in issue order (pinned: a sched_barrier behind every MFMA slot).  Plane 1 of the step's fragments is in registers when it
    // starts; the products run in the order g1 x1, g1 x2 | g2 x1, g1 x3, g3 x1 | g2 x2:
    //   behind the first 2 NB MFMAs  the reads of planes 2 and 3 of stage c & 1, the X fragments first (a fragment is read >= 4 MFMA slots
    //                                 ahead of its first use: the LDS latency is not exposed);
    //   behind the first 4 NB         the split of step c + 1 (three dependent stages per pair of values), its LDS writes into the other
    //                                 stage and the loads of step c + 2, in chunks;
    //   the step's ONE barrier        (every wave has read stage c & 1 and written stage (c + 1) & 1: the writes of the next step go to
    //                                 stage c & 1 again, behind it);
    //   behind g3 x1 and g2 x2        the reads of plane 1 of step c + 1 (G, then X), into registers their last readers are done with.
    s16x8 fg[BA][3], fx[BB][3];
    auto body = [&](int c, auto st_c, auto ld_c) {
        constexpr bool ST = decltype(st_c)::value, LD = decltype(ld_c)::value;
        constexpr int NB = BA * BB, NF = BA + BB, NP = 2 * PASSES, NC = 7 * PASSES + 1, NS = 4 * NB;
        const int sb = (c & 1) * STAGE, sn = ((c + 1) & 1) * STAGE;
        auto read_frag = [&](int stage_off, auto plc, auto fc) {
            constexpr int pl = decltype(plc)::value, f = decltype(fc)::value;
            if constexpr (f < BA) fg[f][pl] = frag(rG, stage_off + pl * PLANE + f * 64);
            else fx[f - BA][pl] = frag(rX, stage_off + pl * PLANE + (f - BA) * 64);
        };
        float x0[NP], x1[NP];
        uint32_t pu[NP], pv[NP], pw[NP], live[PASSES];
        if constexpr (ST) {
#pragma unroll
            for (int i = 0; i < PASSES; ++i) live[i] = ms + 16 * (c + 1) + rg + RPP * i < me ? 0xFFFFFFFFu : 0u;
        }
        char *wb = wbase + sn;
        // chunks: 2 PASSES first stages (they consume `raw`), the loads of step c + 2 (a whole step ahead of their use), then per 16-byte
        // piece the second and third stages of its two pairs and its three LDS writes
        auto chunk = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k == NP) {
                if constexpr (LD) load(c + 2);
            } else if constexpr (ST) {
                if constexpr (k < NP) {                        // rows beyond the run: zeros (bitwise: the clamped row may hold anything)
                    constexpr int j = k, i = j / 2, e = 2 * (j & 1);
                    const float a0 = __uint_as_float(__float_as_uint(raw[i][e]) & live[i]);
                    const float a1 = __uint_as_float(__float_as_uint(raw[i][e + 1]) & live[i]);
                    pu[j] = pack_bf16(a0, a1);
                    x0[j] = a0 - __uint_as_float(pu[j] << 16); x1[j] = a1 - __uint_as_float(pu[j] & 0xFFFF0000u);
                } else {
                    constexpr int i = (k - NP - 1) / 5, q = (k - NP - 1) % 5;
                    if constexpr (q == 4) {
                        *reinterpret_cast<u32x2 *>(wb + RPP * i * PITCH) = u32x2{pu[2 * i], pu[2 * i + 1]};
                        *reinterpret_cast<u32x2 *>(wb + RPP * i * PITCH + PLANE) = u32x2{pv[2 * i], pv[2 * i + 1]};
                        *reinterpret_cast<u32x2 *>(wb + RPP * i * PITCH + 2 * PLANE) = u32x2{pw[2 * i], pw[2 * i + 1]};
                    } else {
                        constexpr int j = 2 * i + (q & 1);
                        if constexpr (q < 2) {
                            pv[j] = pack_bf16(x0[j], x1[j]);
                            x0[j] -= __uint_as_float(pv[j] << 16); x1[j] -= __uint_as_float(pv[j] & 0xFFFF0000u);
                        } else {
                            pw[j] = pack_bf16(x0[j], x1[j]);
                        }
                    }
                }
            }
        };
        static_for<0, 6 * NB>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr int pr = m / NB, blk = m % NB, a = blk / BB, b = blk % BB;
            constexpr int gp = pr == 0 ? 0 : pr == 1 ? 0 : pr == 2 ? 1 : pr == 3 ? 0 : pr == 4 ? 2 : 1;
            constexpr int xp = pr == 0 ? 0 : pr == 1 ? 1 : pr == 2 ? 0 : pr == 3 ? 2 : pr == 4 ? 0 : 1;
            if constexpr (m == 4 * NB) {
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
#ifndef UPP_WS_NO_MFMA        // (diagnostic builds, tools/micro/ws_ablate.sh: wrong results)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fg[a][gp]), __builtin_bit_cast(bf16x8, fx[b][xp]), acc[a][b], 0, 0, 0);
#else
            acc[a][b][m % 16] += __builtin_bit_cast(float, (int)fg[a][gp][m % 8] | ((int)fx[b][xp][m % 8] << 16));
#endif
            constexpr int sl = m % NB;
            // fragment f of a plane in READ order: the X fragments first (the next product needs them), then the G fragments
            auto rd = [&](int stage_off, auto plc, auto oc) {
                constexpr int o = decltype(oc)::value;
                read_frag(stage_off, plc, std::integral_constant<int, (o < BB ? BA + o : o - BB)>{});
            };
#ifndef UPP_WS_NO_READS
            if constexpr (m < 2 * NB) {                     // the reads of plane 2 (m < NB), then of plane 3
                static_for<sl * NF / NB, (sl + 1) * NF / NB>([&](auto oc) { rd(sb, std::integral_constant<int, 1 + m / NB>{}, oc); });
            }
#endif
            if constexpr (m < 4 * NB) {
#ifndef UPP_WS_NO_SPLIT
                static_for<m * NC / NS, (m + 1) * NC / NS>([&](auto kc) { chunk(kc); });
#endif
            } else if constexpr (ST) {                      // plane 1 of the next step: G behind g3 x1 (the last reader of this step's G plane 1
                                                            // was g1 x3), X in the first slots behind g2 x2 (last reader: g3 x1)
                if constexpr (m < 5 * NB && sl < BA) read_frag(sn, std::integral_constant<int, 0>{}, std::integral_constant<int, sl>{});
                if constexpr (m >= 5 * NB && sl < BB) read_frag(sn, std::integral_constant<int, 0>{}, std::integral_constant<int, BA + sl>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    if (nst > 0) {
        load(0);
        split_store(0);
        if (nst > 1) load(1);
        __syncthreads();
        static_for<0, BA + BB>([&](auto fc) {
            constexpr int f = decltype(fc)::value;
            if constexpr (f < BA) fg[f][0] = frag(rG, f * 64);
            else fx[f - BA][0] = frag(rX, (f - BA) * 64);
        });
        int c = 0;
        for (; c + 2 < nst; ++c) body(c, std::true_type{}, std::true_type{});
        if (c + 1 < nst) { body(c, std::true_type{}, std::false_type{}); ++c; }
        body(c, std::false_type{}, std::false_type{});
    }

    // ---- partial tile: block (a, b) register t = dW[n][k], n = n0 + 32 (wm BA + a) + (t & 3) + 8 (t >> 2) + 4 h, k = k0 + 32 (wn BB + b) + r
    // (a swapped problem -- N, K, G, X are the caller's K, N, X, G -- stores its tile transposed: the partial stays (caller's N, caller's K))
    float *out = g.P[p] + (long long)split * N * K;
    if (g.tr[p]) {                  // out[k][n]: the four registers t & 3 of a lane are four consecutive n of its row k
#pragma unroll
        for (int a = 0; a < BA; ++a)
#pragma unroll
            for (int b = 0; b < BB; ++b) {
                const int k = k0 + 32 * (wn * BB + b) + r;
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const int n = n0 + 32 * (wm * BA + a) + 8 * t4 + 4 * h;
                    const f32x4 v = {acc[a][b][4 * t4], acc[a][b][4 * t4 + 1], acc[a][b][4 * t4 + 2], acc[a][b][4 * t4 + 3]};
                    if (n < N && k < K) *reinterpret_cast<f32x4 *>(out + (long long)k * N + n) = v;         // (N % 4 == 0: n + 3 < N)
                }
            }
        return;
    }
#pragma unroll
    for (int a = 0; a < BA; ++a)
#pragma unroll
        for (int b = 0; b < BB; ++b) {
            const int k = k0 + 32 * (wn * BB + b) + r;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int n = n0 + 32 * (wm * BA + a) + (t & 3) + 8 * (t >> 2) + 4 * h;
                if (n < N && k < K) out[(long long)n * K + k] = acc[a][b][t];
            }
        }
}


// (round 4's wave-specialised variant -- 4 producer + 4 consumer waves on 256 x 128 tiles -- was measured slower than the kernel above and
// left the library in round 6: NOTEBOOK section 10.4, profiles/r04_*)

}  // namespace

// Tile classes.  A problem whose dW has both dimensions above 128 runs on 256 x 256 tiles (8 waves, one workgroup per CU: 10.7 B/clk/CU
// of operand stream at the matrix pipe's pace against 21 on 128 x 128 tiles) or on 128 x 384 tiles -- as they are or with G and X
// swapped (the kernel is symmetric in its operands: the swapped problem computes dW^T and stores it transposed) -- whichever pads the
// least (the Transformer's 384 / 1152 / 1536 are multiples of 384, not of 256); a class that does not make one round of 512-row units on
// every CU, and every other problem, runs on 128 x 128 tiles (4 waves, two workgroups per CU).  At most three launches.
enum { WS_NARROW = 0, WS_WIDE = 1, WS_TALL = 2, WS_TALL_SWAPPED = 3 };
struct WsTile { int tn, tk; };
static WsTile ws_tile(int cls) { return cls == WS_WIDE ? WsTile{256, 256} : cls == WS_TALL ? WsTile{128, 384} : cls == WS_TALL_SWAPPED ? WsTile{384, 128} : WsTile{128, 128}; }
static long long ws_tiles(int cls, int N, int K) { const WsTile t = ws_tile(cls); return (long long)((N + t.tn - 1) / t.tn) * ((K + t.tk - 1) / t.tk); }
static void ws_classify(int count, const int *M, const int *N, const int *K, int *cls) {
    double work[4] = {0.0, 0.0, 0.0, 0.0};
    for (int p = 0; p < count; ++p) {
        cls[p] = WS_NARROW;
        if (N[p] > 128 && K[p] > 128) {
            double best = 0.0;
            for (int c = WS_WIDE; c <= WS_TALL_SWAPPED; ++c) {
                const WsTile t = ws_tile(c);
                const double padded = (double)ws_tiles(c, N[p], K[p]) * t.tn * t.tk / (c == WS_WIDE ? 1.0 : 0.92);     // (the wide tile is the faster one at equal padding)
                if (c == WS_WIDE || padded < best) { best = padded; cls[p] = c; }
            }
        }
        work[cls[p]] += (double)ws_tiles(cls[p], N[p], K[p]) * M[p];
    }
    work[WS_TALL] += work[WS_TALL_SWAPPED];
    for (int p = 0; p < count; ++p) {
        const int c = cls[p] == WS_TALL_SWAPPED ? WS_TALL : cls[p];
        if (c != WS_NARROW && work[c] < 256.0 * 512.0) cls[p] = WS_NARROW;
    }
}

// Rows per split for upp_linear_wgrad_grouped_sb (the plan of upp_linear_wgrad_grouped_rows, per launch): about three rounds of equal
// units on the resident workgroups, at least 256 rows per unit, whole multiples of 32.
extern "C" int upp_linear_wgrad_grouped_sb_rows(int count, const int *M, const int *N, const int *K, int *rows) {
    if (count < 1 || count > 4096 || !M || !N || !K || !rows) return UPP_E_BADARG;
    for (int p = 0; p < count; ++p)
        if (M[p] < 1 || N[p] < 1 || K[p] < 1) return UPP_E_BADARG;
    int cls[4096];
    ws_classify(count, M, N, K, cls);
    for (int launch = 0; launch < 3; ++launch) {             // narrow, wide, tall (both orientations)
        double work = 0.0;
        for (int p = 0; p < count; ++p)
            if ((cls[p] == WS_TALL_SWAPPED ? WS_TALL : cls[p]) == launch) work += (double)ws_tiles(cls[p], N[p], K[p]) * M[p];
        const double rounds = 3.0;                           // (A/B: profiles/r05_wgrad_rounds.txt)
        long long per_unit = (long long)(work / (launch == WS_NARROW ? 512.0 * rounds : 256.0 * rounds));
        per_unit = (per_unit + 31) / 32 * 32;
        if (per_unit < 256) per_unit = 256;
        for (int p = 0; p < count; ++p) {
            if ((cls[p] == WS_TALL_SWAPPED ? WS_TALL : cls[p]) != launch) continue;
            const long long m32 = ((long long)M[p] + 31) / 32 * 32;
            long long rws = per_unit < m32 ? per_unit : m32;
            const long long splits = (M[p] + rws - 1) / rws;            // even the splits out
            rws = ((M[p] + splits - 1) / splits + 31) / 32 * 32;
            rows[p] = (int)rws;
        }
    }
    return 0;
}

// Same work list, limits and partial layout as upp_linear_wgrad_grouped_f32 (linear_rt.hip); see include/upp_hip.h.
extern "C" int upp_linear_wgrad_grouped_sb(const float *const *G, const long long *ldg, const float *const *X, const long long *ldx,
                                           float *const *partials, const int *M, const int *N, const int *K, const int *rows, int count,
                                           void *stream) {
    if (count < 1 || count > 4096 || !G || !ldg || !X || !ldx || !partials || !M || !N || !K || !rows) return UPP_E_BADARG;
    for (int p = 0; p < count; ++p) {
        if (!G[p] || !X[p] || !partials[p] || M[p] < 1 || N[p] < 1 || K[p] < 1) return UPP_E_BADARG;
        if (N[p] % 4 || K[p] % 4 || ldg[p] % 4 || ldx[p] % 4 || ldg[p] < N[p] || ldx[p] < K[p] || rows[p] < 32 || rows[p] % 32) return UPP_E_RANGE;
        if (ldg[p] > 0x7fffffffLL || ldx[p] > 0x7fffffffLL) return UPP_E_RANGE;
        if ((reinterpret_cast<uintptr_t>(G[p]) | reinterpret_cast<uintptr_t>(X[p]) | reinterpret_cast<uintptr_t>(partials[p])) & 15) return UPP_E_RANGE;
    }
    int cls[4096];
    ws_classify(count, M, N, K, cls);
    for (int launch = 0; launch < 3; ++launch) {
        WgGroup g{};
        int np = 0;
        long long units = 0;
        auto flush = [&]() -> int {
            if (!np) return 0;
            g.unit0[np] = (int)units;
            for (int q = np + 1; q <= kMaxWgProblems; ++q) g.unit0[q] = 0x7fffffff;
            if (launch == WS_WIDE) hipLaunchKernelGGL((wgrad_sb_kernel<4, 2, 2, 4, 1>), dim3((unsigned)units), dim3(512), 0, (hipStream_t)stream, g);
            else if (launch == WS_TALL) hipLaunchKernelGGL((wgrad_sb_kernel<2, 4, 2, 3, 1>), dim3((unsigned)units), dim3(512), 0, (hipStream_t)stream, g);
            else hipLaunchKernelGGL((wgrad_sb_kernel<2, 2, 2, 2, 2>), dim3((unsigned)units), dim3(256), 0, (hipStream_t)stream, g);
            np = 0;
            units = 0;
            return upp_launch_status();
        };
        for (int p = 0; p < count; ++p) {
            if ((cls[p] == WS_TALL_SWAPPED ? WS_TALL : cls[p]) != launch) continue;
            const int q = np++;
            const bool sw = cls[p] == WS_TALL_SWAPPED;
            g.G[q] = sw ? X[p] : G[p]; g.X[q] = sw ? G[p] : X[p]; g.P[q] = partials[p];
            g.ldg[q] = (int)(sw ? ldx[p] : ldg[p]); g.ldx[q] = (int)(sw ? ldg[p] : ldx[p]);
            g.M[q] = M[p]; g.N[q] = sw ? K[p] : N[p]; g.K[q] = sw ? N[p] : K[p]; g.rows[q] = rows[p]; g.tr[q] = sw ? 1 : 0;
            const WsTile t = ws_tile(sw ? WS_TALL : cls[p]);
            g.tiles_k[q] = (g.K[q] + t.tk - 1) / t.tk;
            g.tiles[q] = ((g.N[q] + t.tn - 1) / t.tn) * g.tiles_k[q];
            g.unit0[q] = (int)units;
            units += (long long)g.tiles[q] * ((M[p] + rows[p] - 1) / rows[p]);
            if (units > 0x3fffffffLL) return UPP_E_RANGE;
            if (np == kMaxWgProblems) {
                const int rc = flush();
                if (rc) return rc;
            }
        }
        const int rc = flush();
        if (rc) return rc;
    }
    return 0;
}
