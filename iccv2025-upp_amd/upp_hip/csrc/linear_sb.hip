// linear_sb.hip -- the token-matrix Linear layers (reference models/Point_MAE_pretask_dev.py:153-169 Mlp.fc1 / fc2, :172-196
// Attention.qkv / proj) and their data gradients at f32 accuracy on the BF16 matrix pipe of gfx950:
//
//     C (M,N) = epilogue( A (M,K) . W (N,K)^T )      A f32, W a FROZEN f32 weight handed over as three bf16 planes
//
// v_mfma_f32_32x32x2_f32 issues at 64 FLOP/clk/SIMD -- 1/16 of v_mfma_f32_32x32x16_bf16.  linear.hip (the exact-f32 kernel) runs its
// k-loop at 98 % of that pipe on the wide shapes and is bound by it.  Here every f32 operand is split EXACTLY into three bf16 terms
//     x = x1 + x2 + x3,   x1 = rne_bf16(x), x2 = rne_bf16(x - x1), x3 = x - x1 - x2         (both residuals are exact in f32; 3 x 8
//                                                                                            significand bits + signs = the 24 of f32)
// and a product is accumulated in f32 from the six terms of weight >= 2^-16:
//     a w = a1 w1 + (a1 w2 + a2 w1) + (a2 w2 + a1 w3 + a3 w1)  +  [a2 w3 + a3 w2 + a3 w3: |.| <= 2^-25 |a w|, dropped]
// Each bf16 x bf16 product is exact in f32, so the result carries the error of an f32 GEMM (measured against float64 in
// tests/test_gpu_linear_sb.py: not larger than that of the f32 fmaf chain of linear.hip) for 6/16 of its matrix-pipe time.  With the
// matrix pipe 2.7 x cheaper the loop is bound by operand delivery (L2 -> LDS, ~25 B/clk/CU), so:
//
//   * W is split ONCE per weight version (upp_linear_sb_prep) into the exact LDS image of the kernel: [32-row block][32-wide k-stage]
//     [plane][16-byte granule of 8 k][row] -- every LDS-DMA instruction copies 1 KB of contiguous global memory, a lane's fragment of a
//     k-step is one conflict-free ds_read_b128 per plane, and no swizzle arithmetic is left in the kernel for W.
//   * A arrives as f32 (4 B per element, the compact form) by LDS-DMA, 8 rows x 128 B per instruction with the granule swizzle of
//     linear.hip applied to the source address, and is split in registers after the fragment read: 11 VALU instructions per pair of
//     values (v_cvt_pk_bf16_f32, shift / and, subtract), amortised over the RN column blocks a wave owns.
//   * Three (four) LDS stages with COUNTED vmcnt: the DMA of k-stage c + NST - 1 is issued while k-stage c is multiplied, one raw
//     s_barrier per k-stage, DMA in flight across it.
//   * One workgroup per CU in one round, tiles as linear.hip; the contraction may be cut over KS = 2 wave groups (narrow N);
//     partial tiles, bias / GELU / GELU' / multiply epilogues and 16-byte stores through the LDS exactly as linear.hip.
//
// Not bit-reproducible against a scalar restatement (the internal summation of the bf16 MFMA is not documented): parity is by
// tolerance against float64 (oracle/oracle.py linear_f64) -- the exact-f32 kernel stays the bit-pinned one and serves every weight that
// is not frozen, every shape this file does not take (K % 32, unaligned rows, more than one round) and every caller that asks for it.
#include "common.h"

namespace {

#include "linear_shared.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int SB_CHUNK = 6144;          // bytes of one (32-row block, 32-wide k-stage) of the planes: 3 planes x 4 granules x 32 rows x 16 B

struct SbArgs {
    LinArgs l;                          // A, C, bias, aux, M, N, K, epi, tiles_n as linear.hip (W unused)
    const unsigned char *planes;        // upp_linear_sb_prep image of W
    int nblocks, kstages;               // ceil(N / 32), ceil(K / 32) of that image
    // the patch embedding's fused layers (dense.hip; PRO = 1 instantiations serve pro_*, every instantiation the epilogues):
    const float *pro_scale, *pro_shift; // a' = max(a * scale[k] + shift[k], 0) applied to the A fragment in front of its split (K <= 512)
    float *gmax; int ldgmax, gshift;    // gmax[row >> gshift][col] = max over the 2^gshift (16 | 32) rows of a group of C (+ bias); C is NOT stored
    float *stat_part;                   // [2][ceil(M / 32)][N]: column sums / sums of squares of C over each 32-row block (rows < M)
    int xcd_gc;                         // > 1: 2-D XCD map with gc column groups (launch_sb decides; the grid is then 8 equal regions)
};

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));        // v_cvt_pk_bf16_f32 (round to nearest even)
}

// 8 f32 -> the three bf16 planes (8 bf16 = 4 VGPRs each).  x - x1 and (x - x1) - x2 are exact, x3 needs no rounding.
// Non-finite values: x1 = rne_bf16(x) is +-inf for x = +-inf and for |x| > 0x7F7F0000 (3.39e38, the largest bf16), and x - x1 is then
// inf - inf = NaN.  SAFE (the one-time split of a WEIGHT, upp_linear_sb_prep) zeroes the residual terms when x1 is not finite; the
// in-loop split of the A operand does not pay those 16 instructions per 8 values.  Either way a non-finite operand makes every output
// it reaches NON-FINITE, but NaN where the exact-f32 kernel has +-inf: the six-product sum pairs the infinite term with residual terms
// of the other operand, which carry both signs (inf - inf).  Stated in include/upp_hip.h; tests/test_gpu_linear_sb.py
// test_non_finite_operands_stay_visible.
template <bool SAFE = false>
__device__ __forceinline__ void split8(const f32x4 lo, const f32x4 hi, u32x4 &p1, u32x4 &p2, u32x4 &p3) {
#ifdef UPP_SB_NO_SPLIT            // diagnostic build: no split arithmetic (wrong results)
    p1 = __builtin_bit_cast(u32x4, lo); p2 = __builtin_bit_cast(u32x4, hi); p3 = p1 ^ p2;
    return;
#endif
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = q < 2 ? lo[2 * q] : hi[2 * q - 4], x1 = q < 2 ? lo[2 * q + 1] : hi[2 * q - 3];
        const uint32_t u = pack_bf16(x0, x1);
        float r0 = x0 - __uint_as_float(u << 16), r1 = x1 - __uint_as_float(u & 0xFFFF0000u);
        if constexpr (SAFE) {
            if (((u << 16) & 0x7F800000u) == 0x7F800000u) r0 = 0.0f;
            if ((u & 0x7F800000u) == 0x7F800000u) r1 = 0.0f;
        }
        const uint32_t v = pack_bf16(r0, r1);
        const float t0 = r0 - __uint_as_float(v << 16), t1 = r1 - __uint_as_float(v & 0xFFFF0000u);
        p1[q] = u; p2[q] = v; p3[q] = pack_bf16(t0, t1);
    }
}

template <int BMB, int BNB, int RN, int KS, int NST, int PRO = 0>      // PRO = 1: BatchNorm + ReLU applied to the A fragment (pro_scale / pro_shift)
__global__ __launch_bounds__(BMB *(BNB / RN) * KS * 64) void linear_sb_kernel(SbArgs ga) {
    static_assert(BNB % RN == 0 && (RN == 1 || RN == 2 || RN == 4), "wave tile: 1 x RN blocks");
    static_assert(!PRO || ((RN == 2 || RN == 4) && KS == 1 && BMB * (BNB / RN) >= 8), "the prologue variant: 8-wave tiles with two or four blocks per wave");
    constexpr int WPG = BMB * (BNB / RN), NW = WPG * KS, BM = BMB * 32, BN = BNB * 32;
    constexpr int AG = BM * 128, GB = AG + BNB * SB_CHUNK;          // bytes of one wave group's share of a k-stage: [A rows | W blocks]
    constexpr int STAGE = KS * GB, TG = GB / 1024, T = KS * TG;     // DMA wave-instructions per stage (1 KB each)
    constexpr int TPW = (T + NW - 1) / NW;
    constexpr bool PADDED = TPW * NW != T;                          // some waves issue a dummy instruction into a scratch KB
    constexpr int RED = NW * RN * 4096;
    constexpr int TABOFF = NST * STAGE + (PADDED ? 1024 : 0);      // PRO: [scale 512 | shift 512] floats behind the stages
    constexpr int LDS_BYTES = (TABOFF + (PRO ? 4096 : 0)) > RED ? (TABOFF + (PRO ? 4096 : 0)) : RED;
    static_assert(LDS_BYTES <= 160 * 1024, "stages exceed the 160 KB of LDS");
    static_assert(NST >= 2 && NST <= 4, "LDS stages");
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];

    const LinArgs &g = ga.l;
    UPP_STAMP(0)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    const int M = g.M, N = g.N;
    int by, bx;
    if (ga.xcd_gc > 1) {
        // 2-D XCD map (wide outputs): the 8 XCDs form a (8 / gc) x gc grid of tile regions, so an XCD's L2 holds 1 / gc of W and gc / 8 of A
        // instead of all of W and 1 / 8 of A (row-major ranges).  Block b runs on XCD b % 8 (observed dispatch; for speed only): it takes
        // tile b / 8 of that XCD's region; the grid is padded to 8 equal regions and the surplus workgroups leave at once.
        const int tiles_m = (M + BM - 1) / BM, gc = ga.xcd_gc, gr = 8 / gc;
        const int x = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3;
        const int rows_per = (tiles_m + gr - 1) / gr, cols_per = g.tiles_n / gc;
        const int ri = i / cols_per, ci = i - ri * cols_per;
        by = (x / gc) * rows_per + ri; bx = (x % gc) * cols_per + ci;
        if (ri >= rows_per || by >= tiles_m) return;
    } else {
        const int lin = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
        by = lin / g.tiles_n; bx = lin - by * g.tiles_n;
    }
    const int m0 = by * BM, n0 = bx * BN;
    const int nsc = g.K / (32 * KS);                                 // k-stages per wave group (K % (32 KS) == 0: checked by the host)
    const int ks = wave / WPG, wb = wave - ks * WPG;
    const int bm = wb / (BNB / RN), bnp = wb - bm * (BNB / RN);

    // ---- DMA sources.  Instruction t of a stage fills KB t of it: group t / TG, then [A: 8 rows x 128 B | W: block j, piece 0..5].
    const char *src[TPW];
    int adv[TPW];                                                    // (scalar) bytes from one k-stage to the next
    unsigned dst[TPW];                                               // (scalar) LDS offset inside a stage
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int t = wave + q * NW;
        if (PADDED && t >= T) {                                      // dummy: 1 KB of the planes into the scratch KB, every stage
            src[q] = reinterpret_cast<const char *>(ga.planes) + lane * 16; adv[q] = 0; dst[q] = NST * STAGE;
            continue;
        }
        const int grp = t / TG, idx = t - grp * TG;
        dst[q] = grp * GB + idx * 1024;
        if (idx < BM / 8) {
            const int rt = idx * 8 + (lane >> 3);                    // row of the tile
            const int row = min(m0 + rt, M - 1);
            const int ko = grp * (g.K / KS) + 4 * ((lane & 7) ^ ((rt >> 1) & 7));
            src[q] = reinterpret_cast<const char *>(g.A + (long long)row * g.lda + ko); adv[q] = 128;
        } else {
            const int wi = idx - BM / 8, j = wi / 6, piece = wi - j * 6;
            const int nb = min(n0 / 32 + j, ga.nblocks - 1);
            src[q] = reinterpret_cast<const char *>(ga.planes) + ((long long)nb * ga.kstages + (long long)grp * nsc) * SB_CHUNK + piece * 1024 + lane * 16;
            adv[q] = SB_CHUNK;
        }
    }
    auto issue1 = [&](int q, int stage, int c) {
#ifdef UPP_SB_NO_DMA          // diagnostic build: no operand stream (the loop runs on whatever the LDS holds)
        (void)q; (void)stage; (void)c;
        return;
#endif
        const unsigned off = (PADDED && wave + q * NW >= T) ? dst[q] : stage * STAGE + dst[q];
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[q] + (long long)c * adv[q]), (lds_ptr_t)(lds + off), 16, 0, 0);
    };
    auto issue = [&](int stage, int c) {
#pragma unroll
        for (int q = 0; q < TPW; ++q) issue1(q, stage, c);
    };

    // ---- fragment addresses.  A: row image (bm 32 + r) of the group, granules 4 s + 2 h + {0, 1} of k-step s, XOR-swizzled;
    //      W: block (bnp RN + jj), plane p, granule 2 s + h, row r  ->  one base + immediates.
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;
    const int sw = (r >> 1) & 7;
    unsigned adrA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) adrA[i] = lds0 + ks * GB + (bm * 32 + r) * 128 + (((4 * (i >> 1) + 2 * h + (i & 1)) ^ sw) << 4);
    const unsigned adrW = lds0 + ks * GB + AG + bnp * RN * SB_CHUNK + h * 512 + r * 16;

    f32x16 acc[RN];
#pragma unroll
    for (int jj = 0; jj < RN; ++jj)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[jj][t] = 0.0f;

    // Fragment reads are inline asm (to hipcc an LDS-DMA is a pending LDS write that any compiler-visible ds_read may alias: it would wait
    // vmcnt(0) in front of each); the wait that retires them names the destination registers as read-write operands, which orders every
    // consumer behind it without a scheduling barrier.
    struct WSet { u32x4 w[RN][3]; };
    struct ASplit { u32x4 p1, p2, p3; };
    f32x4 ra0, ra1;                       // the f32 A fragment of the step being fetched (split as soon as it has landed)
    f32x4 sc0, sc1, sh0, sh1;             // PRO: scale / shift of its 8 values of k
    const unsigned adrT = lds0 + TABOFF + h * 32;
    if constexpr (PRO) {                  // the table: plain stores, visible behind the barrier in front of the first fragment read
        float *tab = reinterpret_cast<float *>(lds + TABOFF);
        for (int i = threadIdx.x; i < 512; i += NW * 64) {
            tab[i] = i < g.K ? ga.pro_scale[i] : 0.0f;
            tab[512 + i] = i < g.K ? ga.pro_shift[i] : 0.0f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#define SB_RDA(DST, ADR) asm volatile("ds_read_b128 %0, %1" : "=v"(DST) : "v"(ADR));
#define SB_RDW(DST, ADR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADR), "n"(OFF));
#define SB_READ(WS, S, SA, TA)                                                                 \
    SB_RDA(ra0, adrA[2 * (S)] + (SA)) SB_RDA(ra1, adrA[2 * (S) + 1] + (SA))                    \
    if constexpr (PRO) { SB_RDW(sc0, adrT + (TA), 0) SB_RDW(sc1, adrT + (TA), 16) SB_RDW(sh0, adrT + (TA), 2048) SB_RDW(sh1, adrT + (TA), 2064) } \
    SB_RDW(WS.w[0][0], adrW + (SA), (S) * 1024) SB_RDW(WS.w[0][1], adrW + (SA), 2048 + (S) * 1024) SB_RDW(WS.w[0][2], adrW + (SA), 4096 + (S) * 1024) \
    if constexpr (RN == 2) {                                                                   \
        SB_RDW(WS.w[1][0], adrW + (SA), SB_CHUNK + (S) * 1024) SB_RDW(WS.w[1][1], adrW + (SA), SB_CHUNK + 2048 + (S) * 1024) \
        SB_RDW(WS.w[1][2], adrW + (SA), SB_CHUNK + 4096 + (S) * 1024)                          \
    }                                                                                          \
    if constexpr (RN == 4) {                                                                   \
        SB_RDW(WS.w[1][0], adrW + (SA), SB_CHUNK + (S) * 1024) SB_RDW(WS.w[1][1], adrW + (SA), SB_CHUNK + 2048 + (S) * 1024) \
        SB_RDW(WS.w[1][2], adrW + (SA), SB_CHUNK + 4096 + (S) * 1024)                          \
        SB_RDW(WS.w[2][0], adrW + (SA), 2 * SB_CHUNK + (S) * 1024) SB_RDW(WS.w[2][1], adrW + (SA), 2 * SB_CHUNK + 2048 + (S) * 1024) \
        SB_RDW(WS.w[2][2], adrW + (SA), 2 * SB_CHUNK + 4096 + (S) * 1024)                      \
        SB_RDW(WS.w[3][0], adrW + (SA), 3 * SB_CHUNK + (S) * 1024) SB_RDW(WS.w[3][1], adrW + (SA), 3 * SB_CHUNK + 2048 + (S) * 1024) \
        SB_RDW(WS.w[3][2], adrW + (SA), 3 * SB_CHUNK + 4096 + (S) * 1024)                      \
    }
    // acc[0] among the wait's operands pins the first half's matrix instructions IN FRONT of the wait (otherwise hipcc may sink them behind
    // it and the fragment reads are not covered): k-loop of the 128 x 128 tile 21,400 -> 19,700 cycles; at ONE wave per SIMD the same pin
    // costs 25 % (nothing else can issue while the lone wave sits in the wait), so only workgroups of >= 8 waves carry it.
#define SB_WAIT(WS)                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    if constexpr (RN == 4)                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra0), "+v"(ra1), "+v"(WS.w[0][0]), "+v"(WS.w[0][1]), "+v"(WS.w[0][2]), \
                     "+v"(WS.w[1][0]), "+v"(WS.w[1][1]), "+v"(WS.w[1][2]), "+v"(WS.w[2][0]), "+v"(WS.w[2][1]), "+v"(WS.w[2][2]), \
                     "+v"(WS.w[3][0]), "+v"(WS.w[3][1]), "+v"(WS.w[3][2]), "+v"(acc[0]));       \
    if constexpr (RN == 4 && PRO) asm volatile("" : "+v"(sc0), "+v"(sc1), "+v"(sh0), "+v"(sh1));      /* (30-operand limit: the table fragments ride on a second statement) */ \
    if constexpr (RN == 4) {}                                                                  \
    else if constexpr (PRO)                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra0), "+v"(ra1), "+v"(WS.w[0][0]), "+v"(WS.w[0][1]), "+v"(WS.w[0][2]), \
                     "+v"(WS.w[RN - 1][0]), "+v"(WS.w[RN - 1][1]), "+v"(WS.w[RN - 1][2]), "+v"(acc[0]), "+v"(sc0), "+v"(sc1), "+v"(sh0), "+v"(sh1)); \
    else if constexpr (RN > 1 && NW >= 8)                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra0), "+v"(ra1), "+v"(WS.w[0][0]), "+v"(WS.w[0][1]), "+v"(WS.w[0][2]), \
                     "+v"(WS.w[RN - 1][0]), "+v"(WS.w[RN - 1][1]), "+v"(WS.w[RN - 1][2]), "+v"(acc[0])); \
    else if constexpr (NW >= 8)                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra0), "+v"(ra1), "+v"(WS.w[0][0]), "+v"(WS.w[0][1]), "+v"(WS.w[0][2]), "+v"(acc[0])); \
    else if constexpr (RN > 1)                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra0), "+v"(ra1), "+v"(WS.w[0][0]), "+v"(WS.w[0][1]), "+v"(WS.w[0][2]), \
                     "+v"(WS.w[RN - 1][0]), "+v"(WS.w[RN - 1][1]), "+v"(WS.w[RN - 1][2]));     \
    else                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra0), "+v"(ra1), "+v"(WS.w[0][0]), "+v"(WS.w[0][1]), "+v"(WS.w[0][2])); \
    __builtin_amdgcn_sched_barrier(0);
#ifdef UPP_SB_NO_MFMA          // diagnostic build: operands delivered, read and split, no matrix instructions
#define SB_MFMA(AV, WV, JJ) asm volatile("" ::"v"(AV), "v"(WV));
#else
#define SB_MFMA(AV, WV, JJ) acc[JJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, AV), __builtin_bit_cast(bf16x8, WV), acc[JJ], 0, 0, 0);
#endif
    // the six products of one block and k-step, smallest terms first.  SB_M1: one matrix instruction pinned in program order; SB_D: the
    // q-th DMA instruction of this wave for k-stage CC (into LDS stage ST) pinned behind it -- issued as ONE burst behind the barrier the
    // DMA instructions cost every wave 60-180 issue cycles each at the same moment with the matrix pipe idle (measured: the loop took
    // delivery-alone + multiply-alone); one at a time between matrix instructions they are covered.
#define SB_M1(AV, WV, JJ) SB_MFMA(AV, WV, JJ) __builtin_amdgcn_sched_barrier(0);
#define SB_D(Q, ST, CC) if constexpr ((Q) < TPW) { issue1(Q, ST, CC); __builtin_amdgcn_sched_barrier(0); }
    // first half of a k-step's matrix work: runs while the NEXT step's fragments are on their way from the LDS.  DO: issue DMA
    // instructions Q0 ... Q0 + 3 (those below QE) of k-stage CC behind the first matrix instructions.
#define SB_HALF1(X, WS, DO, Q0, QE, ST, CC)                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    SB_M1(X.p3, WS.w[0][0], 0) if constexpr (DO && (Q0) < (QE)) { SB_D(Q0, ST, CC) }           \
    SB_M1(X.p1, WS.w[0][2], 0) if constexpr (DO && (Q0) + 1 < (QE)) { SB_D((Q0) + 1, ST, CC) } \
    SB_M1(X.p2, WS.w[0][1], 0) if constexpr (DO && (Q0) + 2 < (QE)) { SB_D((Q0) + 2, ST, CC) } \
    if constexpr (RN > 1) { SB_M1(X.p2, WS.w[0][0], 0) if constexpr (DO && (Q0) + 3 < (QE)) { SB_D((Q0) + 3, ST, CC) } SB_M1(X.p1, WS.w[0][1], 0) SB_M1(X.p1, WS.w[0][0], 0) } \
    if constexpr (DO) { _Pragma("unroll") for (int q_ = (Q0) + QSLOT; q_ < (QE); ++q_) issue1(q_, ST, CC); }     /* (more instructions than slots: the few-wave tiles) */
    // second half, interleaved with the split of the next step's A fragment (one matrix instruction, then its share of the ~44 VALU)
#define SB_PROD_LO(X, WS, JJ) SB_MFMA(X.p3, WS.w[JJ][0], JJ) SB_MFMA(X.p1, WS.w[JJ][2], JJ) SB_MFMA(X.p2, WS.w[JJ][1], JJ)
#define SB_PROD_HI(X, WS, JJ) SB_MFMA(X.p2, WS.w[JJ][0], JJ) SB_MFMA(X.p1, WS.w[JJ][1], JJ) SB_MFMA(X.p1, WS.w[JJ][0], JJ)
#define SB_PRO()                                                                               \
    if constexpr (PRO) {                                                                       \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                     \
            ra0[e_] = fmaxf(__builtin_fmaf(ra0[e_], sc0[e_], sh0[e_]), 0.0f);                  \
            ra1[e_] = fmaxf(__builtin_fmaf(ra1[e_], sc1[e_], sh1[e_]), 0.0f);                  \
        }                                                                                      \
    }
#define SB_HALF2(X, WS, XN)                                                                    \
    SB_PRO()                                                                                   \
    split8(ra0, ra1, XN.p1, XN.p2, XN.p3);                                                     \
    if constexpr (RN == 4) { SB_PROD_LO(X, WS, 1) SB_PROD_HI(X, WS, 1) SB_PROD_LO(X, WS, 2) SB_PROD_HI(X, WS, 2) SB_PROD_LO(X, WS, 3) SB_PROD_HI(X, WS, 3) } \
    else if constexpr (RN > 1) { SB_PROD_LO(X, WS, RN - 1) SB_PROD_HI(X, WS, RN - 1) } else { SB_PROD_HI(X, WS, 0) } \
    asm volatile("" : "+v"(XN.p1), "+v"(XN.p2), "+v"(XN.p3));           /* (the split belongs HERE: not sunk to its first use) */ \
    _Pragma("unroll") for (int i_ = 0; i_ < (RN == 4 ? 18 : RN > 1 ? 6 : 3); ++i_) {          \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
        __builtin_amdgcn_sched_group_barrier(0x002, RN == 4 ? 3 : RN > 1 ? 8 : 15, 0);         \
    }                                                                                          \
    __builtin_amdgcn_sched_barrier(0);

    // ---- main loop.  NST LDS stages, all of them filled ahead: k-stage c + NST goes into the buffer of k-stage c as soon as the barrier
    // in the middle of iteration c has shown that every wave is done reading it -- its first QH instructions behind the matrix
    // instructions of step (c, 1), the rest behind those of step (c + 1, 0).  The matrix work is software-pipelined over 16-wide k-steps
    // (two per stage): while a step is multiplied, the next step's fragments are read from the LDS (first half of its matrix
    // instructions) and its A fragment is split (second half); the barrier that opens the next stage sits between the two steps of a
    // stage, where every wave still has a step's worth of matrix work whose operands are in registers.
    constexpr int QSLOT = RN > 1 ? 4 : 3;                              // DMA slots behind the first matrix instructions of a k-step
    constexpr int QH = (TPW + 1) / 2 > QSLOT ? QSLOT : (TPW + 1) / 2;   // instructions of a k-stage issued behind step 1, the rest behind step 0
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(s, s);
#pragma unroll
    for (int q = 0; q < QH; ++q) issue1(q, NST - 1, NST - 1);
    wait_vmcnt<(NST - 2) * TPW + QH>();
    __builtin_amdgcn_s_barrier();
    WSet w0, w1;
    ASplit x0, x1;
    SB_READ(w0, 0, 0u, 0u)
    SB_WAIT(w0)
    SB_PRO()
    split8(ra0, ra1, x0.p1, x0.p2, x0.p3);
    // one iteration = k-stage c.  D1: the rest of k-stage c + NST - 1 is issued behind step 0; D2: the head of k-stage c + NST behind step 1.
#define SB_ITER(D1, D2, VM_STMT)                                                               \
    {                                                                                          \
        const unsigned sa = stage * STAGE;                                                     \
        const int nxt = stage == NST - 1 ? 0 : stage + 1, prv = stage == 0 ? NST - 1 : stage - 1; \
        SB_READ(w1, 1, sa, (unsigned)(c * 128 + 64))                                           \
        SB_HALF1(x0, w0, D1, QH, TPW, prv, c + NST - 1)                                        \
        SB_WAIT(w1)                                                                            \
        SB_HALF2(x0, w0, x1)                                                                   \
        VM_STMT                                                                                \
        __builtin_amdgcn_s_barrier();         /* k-stage c + 1 is complete; everyone is done reading k-stage c */ \
        SB_READ(w0, 0, nxt * STAGE, (unsigned)(c * 128 + 128))                                 \
        SB_HALF1(x1, w1, D2, 0, QH, stage, c + NST)                                            \
        SB_WAIT(w0)                                                                            \
        SB_HALF2(x1, w1, x0)                                                                   \
        stage = nxt;                                                                           \
    }
    UPP_STAMP(1)
    int stage = 0, c = 0;
    for (; c < nsc - NST; ++c) SB_ITER(true, true, wait_vmcnt<(NST - 2) * TPW>();)
    SB_ITER(true, false, wait_vmcnt<(NST - 2) * TPW>();)                                      // c = nsc - NST (nsc >= NST: checked by the host)
    ++c;
    for (; c + 1 < nsc; ++c)
        SB_ITER(false, false, if (NST >= 4 && nsc - 2 - c == 1) wait_vmcnt<TPW>(); else wait_vmcnt<0>();)
    {   // the last k-stage: nothing left to fetch behind step 1
        const unsigned sa = stage * STAGE;
        SB_READ(w1, 1, sa, (unsigned)(c * 128 + 64))
        SB_HALF1(x0, w0, false, 0, 0, 0, 0)
        SB_WAIT(w1)
        SB_HALF2(x0, w0, x1)
        SB_HALF1(x1, w1, false, 0, 0, 0, 0)
        if constexpr (RN == 4) { SB_PROD_LO(x1, w1, 1) SB_PROD_HI(x1, w1, 1) SB_PROD_LO(x1, w1, 2) SB_PROD_HI(x1, w1, 2) SB_PROD_LO(x1, w1, 3) SB_PROD_HI(x1, w1, 3) }
        else if constexpr (RN > 1) { SB_PROD_LO(x1, w1, RN - 1) SB_PROD_HI(x1, w1, RN - 1) } else { SB_PROD_HI(x1, w1, 0) }
    }
#undef SB_ITER
#undef SB_D
#undef SB_M1
#undef SB_HALF2
#undef SB_PRO
#undef SB_HALF1
#undef SB_PROD_HI
#undef SB_PROD_LO
#undef SB_MFMA
#undef SB_WAIT
#undef SB_READ
#undef SB_RDW
#undef SB_RDA

    // ---- tile -> memory (as linear.hip): every block through the LDS as [slot][t][lane], slot = ks (BMB BNB) + wb RN + jj; the KS partial
    // tiles are summed in wave-group order from 0.0f, bias / activation / factor act on four columns per lane, 16-byte stores.
    UPP_STAMP(2)
    constexpr int TN = 16 / KS;
    const int T0 = ks * TN;
    __syncthreads();                                         // all fragment reads done (and no DMA in flight): the stages may be overwritten
    float *red = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int jj = 0; jj < RN; ++jj)
#pragma unroll
        for (int t = 0; t < 16; ++t) red[((wave * RN + jj) * 16 + t) * 64 + lane] = acc[jj][t];
    if (KS > 1) __syncthreads();
    const int rb = m0 + bm * 32;
#pragma unroll
    for (int jj = 0; jj < RN; ++jj) {
        const int cb = n0 + (bnp * RN + jj) * 32;
        const int col = cb + 4 * (lane & 7);
        const bool col_ok = col < N;
        f32x4 bias4 = {0.0f, 0.0f, 0.0f, 0.0f};
        if (g.epi != LEPI_NONE && g.epi != LEPI_MUL && col_ok && g.bias_shift == 0) bias4 = *reinterpret_cast<const f32x4 *>(g.bias + col);
        // the patch embedding's epilogues (KS = 1 tiles): group max over 16 | 32 rows instead of the store, column sums of the stored values
        const bool want_max = KS == 1 && ga.gmax != nullptr, want_stats = KS == 1 && ga.stat_part != nullptr;
        const float ninf = -__builtin_inff();
        f32x4 mx[2] = {{ninf, ninf, ninf, ninf}, {ninf, ninf, ninf, ninf}}, sm = {0.0f, 0.0f, 0.0f, 0.0f}, sq = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int pass = 0; pass < TN / 4; ++pass) {
            const int ridx = pass * 8 + (lane >> 3), t = T0 + (ridx >> 1);
            const int row = rb + (t & 3) + 8 * (t >> 2) + 4 * (ridx & 1);
            if (g.bias_shift > 0 && col_ok)                      // a bias per group of 2^bias_shift rows (upp_linear_sb_group_bias_f32)
                bias4 = *reinterpret_cast<const f32x4 *>(g.bias + (long long)(min(row, M - 1) >> g.bias_shift) * N + col);
            const float *sp = red + ((wb * RN + jj) * 16 + T0) * 64 + (pass * 64 + lane) * 4;
            f32x4 v = *reinterpret_cast<const f32x4 *>(sp);
            if (KS > 1) {
                const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
                v = zero + v;
#pragma unroll
                for (int k2 = 1; k2 < KS; ++k2) v += *reinterpret_cast<const f32x4 *>(sp + k2 * (BMB * BNB) * 1024);
            }
            if (want_max || want_stats) {                        // (epilogue NONE / BIAS, checked by the host)
                const f32x4 val = v + bias4;
                const bool live = row < M;
                if (want_stats) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { sm[e] += live ? val[e] : 0.0f; sq[e] += live ? val[e] * val[e] : 0.0f; }
                }
                if (want_max) {
                    const int gi = ga.gshift == 4 ? pass >> 1 : 0;          // rows 0..15 of a block are passes 0, 1
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float c = live ? val[e] : ninf;
                        if (gi == 0) mx[0][e] = fmaxf(mx[0][e], c); else mx[1][e] = fmaxf(mx[1][e], c);
                    }
                    continue;
                }
            }
            epilogue_store4(g, g.epi, v, bias4, row, col, col_ok && row < M);
        }
        if (want_max || want_stats) {                            // the 8 lanes lane & 7 + 8 i hold the 32 rows of these four columns
#pragma unroll
            for (int off = 8; off < 64; off <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (want_max) { mx[0][e] = fmaxf(mx[0][e], __shfl_xor(mx[0][e], off)); mx[1][e] = fmaxf(mx[1][e], __shfl_xor(mx[1][e], off)); }
                    if (want_stats) { sm[e] += __shfl_xor(sm[e], off); sq[e] += __shfl_xor(sq[e], off); }
                }
            if ((lane >> 3) == 0 && col_ok && rb < M) {
                if (want_stats) {
                    const long long nblk = (M + 31) / 32;
                    *reinterpret_cast<f32x4 *>(ga.stat_part + (long long)(rb >> 5) * N + col) = sm;
                    *reinterpret_cast<f32x4 *>(ga.stat_part + (nblk + (rb >> 5)) * N + col) = sq;
                }
                if (want_max) {
                    *reinterpret_cast<f32x4 *>(ga.gmax + (long long)(rb >> ga.gshift) * ga.ldgmax + col) = mx[0];
                    if (ga.gshift == 4 && rb + 16 < M) *reinterpret_cast<f32x4 *>(ga.gmax + (long long)((rb >> 4) + 1) * ga.ldgmax + col) = mx[1];
                }
            }
        }
    }
    UPP_STAMP(3)
}

// W (N,K) f32 -> the plane image [block][k-stage][plane][granule][row][8 bf16], zero beyond N and K.  One thread per (row, granule).
// `transposed`: the operand is W^T of a row-major (K,N) matrix `W` (the B operand of a data-gradient GEMM, dX = dY . W): element (n,k) is
// read from W[k ldw + n] -- the 32 rows of a block are 32 consecutive floats of a source row -- so no f32 copy of W^T is ever made.
constexpr int kMaxPrep = 48;
struct PrepJobs { const float *W[kMaxPrep]; long long ldw[kMaxPrep]; unsigned char *out[kMaxPrep]; int N[kMaxPrep], K[kMaxPrep], tr[kMaxPrep], first[kMaxPrep + 1]; int count; };

__device__ __forceinline__ void prep_chunk(const float *__restrict__ W, long long ldw, int N, int K, int transposed, int kstages, int chunk, unsigned char *__restrict__ out) {
    const int nb = chunk / kstages, kst = chunk - nb * kstages;
    const int row = threadIdx.x & 31, gq = threadIdx.x >> 5;
    const int n = nb * 32 + row, k0 = kst * 32 + gq * 8;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (n < N && k0 + e < K) ? (transposed ? W[(long long)(k0 + e) * ldw + n] : W[(long long)n * ldw + k0 + e]) : 0.0f;
    const f32x4 lo = {x[0], x[1], x[2], x[3]}, hi = {x[4], x[5], x[6], x[7]};
    u32x4 p1, p2, p3;
    split8<true>(lo, hi, p1, p2, p3);
    unsigned char *base = out + (long long)chunk * SB_CHUNK + gq * 512 + row * 16;
    *reinterpret_cast<u32x4 *>(base) = p1;
    *reinterpret_cast<u32x4 *>(base + 2048) = p2;
    *reinterpret_cast<u32x4 *>(base + 4096) = p3;
}

__global__ __launch_bounds__(128) void linear_sb_prep_kernel(const float *__restrict__ W, long long ldw, int N, int K, int transposed, int kstages, unsigned char *__restrict__ out) {
    prep_chunk(W, ldw, N, K, transposed, kstages, (int)blockIdx.x, out);
}

// several weights in one launch (the trainable weights of a step driver, re-split at the start of every step): workgroup -> (job, chunk)
__global__ __launch_bounds__(128) void linear_sb_prep_batched_kernel(PrepJobs t) {
    const int b = blockIdx.x;
    int j = 0;
    while (j + 1 < t.count && b >= t.first[j + 1]) ++j;                // (scalar: at most kMaxPrep compares)
    prep_chunk(t.W[j], t.ldw[j], t.N[j], t.K[j], t.tr[j], (t.K[j] + 31) / 32, b - t.first[j], t.out[j]);
}

#ifndef UPP_SB_NST44
#define UPP_SB_NST44 3
#endif
#define UPP_SB_CONFIGS(X) X(8, 4, 4, 1, 2) X(8, 4, 2, 1, 2) X(4, 4, 2, 1, UPP_SB_NST44) X(4, 3, 1, 1, 4) X(3, 4, 2, 1, 4) X(2, 4, 2, 1, 4) X(2, 3, 1, 1, 4) X(2, 2, 1, 2, 3) X(2, 2, 2, 4, 2) X(1, 2, 1, 2, 4)
// tiles that pick_sb never chooses, reachable by an explicit tile code: the whole shape space in a -DUPP_SB_SWEEP build (tools/micro/sb_sweep.py)
#ifdef UPP_SB_SWEEP
#include "linear_sb_sweep.h"
#define UPP_SB_EXTRA_CONFIGS(X) UPP_SB_SWEEP_CONFIGS(X)
#else
#define UPP_SB_EXTRA_CONFIGS(X)
#endif
#include "linear_sb_tuned.h"
// (tile code: hex digits 4 BMB BNB RN KS NST)

// The measured choice (linear_sb_tuned.h) for the swept problems on which the fitted model is still behind: EXACT (M, N, K) matches only --
// round 5's table, the main source of choices then, also answered for M within 1 / 16 of a row; with nine residual rows a neighbouring
// problem is better served by the model (a row's tile cost a neighbour 6.8 % in the sweep).  Option UPP_OPT_SB_TUNED = 0: the model alone.
int sb_tuned(int M, int N, int K) {
    if (!upp_option(UPP_OPT_SB_TUNED)) return 0;
    for (const SbTuned &t : kSbTuned)
        if (t.M == M && t.N == N && t.K == K) return t.code;
    return 0;
}

// The cost model (round 6): per compiled tile shape a seven-coefficient time estimate FITTED to the tile sweep of round 5 (197 Linear
// problems of the six recipes x every shape that fits the LDS: profiles/r05_sb_sweep.json, tools/micro/sb_model_fit.py ->
// linear_sb_model.h) --  t = c0 + c1 R + c2 R s + c3 Rc + c4 Rc s + c5 R s u + c6 u  with R = ceil(wgs / 256) rounds of workgroups,
// Rc = max(1, wgs / 256) (a partial last round costs less than a whole one), s = K / (32 KS) k-stages per wave group and u the idle share
// of the chip.  The per-tile constants carry what round 5's analytic model left out (the k-split's LDS reduction, prologue and store burst
// per round, the operand split at its measured weight): that model's choice was within 3 % of the measured best on 23 of the 197 swept
// problems, this one's on 188, and on 95 % of problems held out of the fit (5-fold cross-validation).  0: not a problem for this file.
#include "linear_sb_model.h"
int pick_sb_model(int M, int N, int K) {
    if (K % 32 != 0 || K < 64) return 0;
    int best = 0;
    double best_t = 0.0;
    for (const SbModel &m : kSbModel) {
        const int bmb = (m.code >> 16) & 15, bnb = (m.code >> 12) & 15, ks = (m.code >> 4) & 15, nst = m.code & 15;
        if (K % (32 * ks) != 0 || K / (32 * ks) < nst) continue;          // (every LDS stage is filled before the loop starts)
        const long long wgs = (long long)((M + 32 * bmb - 1) / (32 * bmb)) * ((N + 32 * bnb - 1) / (32 * bnb));
        const double R = (double)((wgs + 255) / 256), Rc = wgs > 256 ? (double)wgs / 256.0 : 1.0, u = wgs < 256 ? 1.0 - (double)wgs / 256.0 : 0.0;
        const double s = (double)(K / (32 * ks));
        double t = m.c[0] + m.c[1] * R + m.c[2] * R * s + m.c[3] * Rc + m.c[4] * Rc * s + m.c[5] * R * s * u + m.c[6] * u;
        if (t < 0.5) t = 0.5;                                            // (an extrapolation far outside the sweep must not go negative)
        if (!best || t < best_t) { best = m.code; best_t = t; }
    }
    return best;
}

// the measured choice where the table has one, the cost model's otherwise
int pick_sb(int M, int N, int K) {
    if (K % 32 != 0 || K < 64) return 0;
    const int t = sb_tuned(M, N, K);
    return t ? t : pick_sb_model(M, N, K);
}

// option UPP_OPT_SB_XCD2D = gc (0: row-major XCD ranges; default 2): column groups of the 2-D XCD map for one-round launches of wide outputs
inline int sb_xcd_cols() { return upp_option(UPP_OPT_SB_XCD2D); }

template <int BMB, int BNB, int RN, int KS, int NST>
int launch_sb(const SbArgs &g0, hipStream_t st) {
    SbArgs g = g0;
    const int tiles_m = (g.l.M + BMB * 32 - 1) / (BMB * 32);
    g.l.tiles_n = (g.l.N + BNB * 32 - 1) / (BNB * 32);
    unsigned grid = (unsigned)(tiles_m * g.l.tiles_n);
    g.xcd_gc = 0;
    const int gc = sb_xcd_cols();
    if (gc > 1 && g.l.tiles_n % gc == 0 && g.l.tiles_n >= 4 * gc && tiles_m >= 8 / gc && tiles_m * g.l.tiles_n <= 256) {
        const int gr = 8 / gc, rows_per = (tiles_m + gr - 1) / gr, cols_per = g.l.tiles_n / gc;
        if (8 * rows_per * cols_per <= 256) { g.xcd_gc = gc; grid = (unsigned)(8 * rows_per * cols_per); }
    }
    hipLaunchKernelGGL((linear_sb_kernel<BMB, BNB, RN, KS, NST>), dim3(grid), dim3(BMB * (BNB / RN) * KS * 64), 0, st, g);
    return upp_launch_status();
}

}  // namespace

#ifdef UPP_LIN_STAMPS
extern "C" void upp_linear_sb_set_stamps(unsigned long long *p) { g_lin_stamps = p; }
#endif

extern "C" int upp_linear_sb_tile(int M, int N, int K) {
    if (M < 1 || N < 1 || K < 1) return UPP_E_BADARG;
    return pick_sb(M, N, K);
}

extern "C" long long upp_linear_sb_planes_bytes(int N, int K) {
    if (N < 1 || K < 1) return UPP_E_BADARG;
    return (long long)((N + 31) / 32) * ((K + 31) / 32) * SB_CHUNK;
}

extern "C" int upp_linear_sb_prep(const float *W, long long ldw, int N, int K, int transposed, void *planes, void *stream) {
    if (!W || !planes || N < 1 || K < 1) return UPP_E_BADARG;
    if (ldw < (transposed ? N : K) || (reinterpret_cast<uintptr_t>(planes) & 15)) return UPP_E_RANGE;
    const int kstages = (K + 31) / 32, nblocks = (N + 31) / 32;
    hipLaunchKernelGGL(linear_sb_prep_kernel, dim3((unsigned)(nblocks * kstages)), dim3(128), 0, (hipStream_t)stream, W, ldw, N, K, transposed ? 1 : 0, kstages,
                       reinterpret_cast<unsigned char *>(planes));
    return upp_launch_status();
}

extern "C" int upp_linear_sb_prep_batched(const float *const *W, const long long *ldw, const int *N, const int *K, const int *transposed,
                                          void *const *planes, int count, void *stream) {
    if (count < 0 || (count > 0 && (!W || !ldw || !N || !K || !transposed || !planes))) return UPP_E_BADARG;
    for (int j0 = 0; j0 < count; j0 += kMaxPrep) {
        PrepJobs t;
        const int n = count - j0 < kMaxPrep ? count - j0 : kMaxPrep;
        int total = 0;
        for (int j = 0; j < n; ++j) {
            const int i = j0 + j;
            if (!W[i] || !planes[i] || N[i] < 1 || K[i] < 1) return UPP_E_BADARG;
            if (ldw[i] < (transposed[i] ? N[i] : K[i]) || (reinterpret_cast<uintptr_t>(planes[i]) & 15)) return UPP_E_RANGE;
            t.W[j] = W[i]; t.ldw[j] = ldw[i]; t.out[j] = reinterpret_cast<unsigned char *>(planes[i]); t.N[j] = N[i]; t.K[j] = K[i]; t.tr[j] = transposed[i] ? 1 : 0;
            t.first[j] = total;
            total += ((N[i] + 31) / 32) * ((K[i] + 31) / 32);
        }
        t.first[n] = total; t.count = n;
        hipLaunchKernelGGL(linear_sb_prep_batched_kernel, dim3((unsigned)total), dim3(128), 0, (hipStream_t)stream, t);
    }
    return upp_launch_status();
}

static int linear_sb_launch(const float *A, long long lda, const void *planes, const float *bias, int bias_shift, float *C, long long ldc, float *aux,
                            long long ldaux, int M, int N, int K, int epilogue, int tile, void *stream) {
    if (!A || !planes || !C || M < 1 || N < 1 || K < 1) return UPP_E_BADARG;
    if (K % 32 != 0 || lda % 4 != 0 || lda < K || ldc < N || N % 4 != 0 || ldc % 4 != 0) return UPP_E_RANGE;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(planes) | reinterpret_cast<uintptr_t>(C)) & 15) return UPP_E_RANGE;
    if (ldc > (1LL << 24) || ldaux > (1LL << 24)) return UPP_E_RANGE;
    if (epilogue < LEPI_NONE || epilogue > LEPI_BIAS_RELU) return UPP_E_RANGE;
    if ((epilogue == LEPI_BIAS || epilogue == LEPI_BIAS_GELU || epilogue == LEPI_BIAS_GELU_D || epilogue == LEPI_BIAS_RELU) && (!bias || (reinterpret_cast<uintptr_t>(bias) & 15))) return UPP_E_BADARG;
    if ((epilogue == LEPI_BIAS_GELU_D || epilogue == LEPI_MUL) && (!aux || ldaux < N || ldaux % 4 != 0 || (reinterpret_cast<uintptr_t>(aux) & 15))) return UPP_E_BADARG;
    if (tile <= 0) tile = pick_sb(M, N, K);
    if (tile <= 0) return UPP_E_RANGE;
    SbArgs g{};
    g.l.A = A; g.l.lda = lda; g.l.C = C; g.l.ldc = ldc; g.l.bias = bias; g.l.aux = aux; g.l.ldaux = ldaux;
    g.l.M = M; g.l.N = N; g.l.K = K; g.l.epi = epilogue; g.l.bias_shift = bias_shift; g.l.wt = upp_store_policy();
#ifdef UPP_LIN_STAMPS
    g.l.stamps = g_lin_stamps;
#endif
    g.planes = reinterpret_cast<const unsigned char *>(planes);
    g.nblocks = (N + 31) / 32; g.kstages = (K + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
#define UPP_SB_CASE(a, b, c, d, e)                                                                    \
    case 0x400000 + a * 65536 + b * 4096 + c * 256 + d * 16 + e:                                          \
        if (K % (32 * d) != 0 || K / (32 * d) < e) return UPP_E_RANGE;                                \
        return launch_sb<a, b, c, d, e>(g, st);
    switch (tile) {
        UPP_SB_CONFIGS(UPP_SB_CASE)
#ifndef UPP_SB_SWEEP                                      // (the sweep build's list contains these)
        UPP_SB_TUNED_CONFIGS(UPP_SB_CASE)
#endif
        UPP_SB_EXTRA_CONFIGS(UPP_SB_CASE)
        default: return UPP_E_RANGE;
    }
#undef UPP_SB_CASE
}

// The two large products of the patch embedding (dense.hip upp_patch_embed_fwd) on this kernel, with the chain's prologue and epilogues:
//   C = A . W^T + bias (a bias per column, or per group of 2^bias_shift rows);  pro_scale / pro_shift: A is max(A * scale + shift, 0) (the
//   BatchNorm + ReLU between the layers; the 128 x 128 tile); stat_part: column sums / sums of squares of C per 32-row block (the next
//   BatchNorm's batch statistics); gmax: the max-pool over each group of 2^gshift rows replaces the store of C.
__attribute__((visibility("hidden"))) int upp_detail_linear_sb_chain(const float *A, long long lda, const void *planes, const float *bias, int bias_shift,
                                                                     float *C, long long ldc, int M, int N, int K, const float *pro_scale,
                                                                     const float *pro_shift, float *gmax, int ldgmax, int gshift, float *stat_part,
                                                                     void *stream) {
    if (!A || !planes || !bias || (!C && !gmax) || M < 1 || N < 1 || K < 96) return UPP_E_BADARG;
    if (K % 32 != 0 || K > 512 || lda % 4 != 0 || lda < K || N % 4 != 0 || (C && (ldc < N || ldc % 4 != 0))) return UPP_E_RANGE;
    if (gmax && ((gshift != 4 && gshift != 5) || ldgmax < N || ldgmax % 4 != 0 || M % (1 << gshift) != 0)) return UPP_E_RANGE;
    if ((pro_scale == nullptr) != (pro_shift == nullptr)) return UPP_E_BADARG;
    SbArgs g{};
    g.l.A = A; g.l.lda = lda; g.l.C = C; g.l.ldc = ldc; g.l.bias = bias; g.l.M = M; g.l.N = N; g.l.K = K;
    g.l.epi = bias_shift > 0 ? LEPI_NONE : LEPI_BIAS; g.l.bias_shift = bias_shift;
#ifdef UPP_LIN_STAMPS
    g.l.stamps = nullptr;
#endif
    g.planes = reinterpret_cast<const unsigned char *>(planes);
    g.nblocks = (N + 31) / 32; g.kstages = (K + 31) / 32;
    g.pro_scale = pro_scale; g.pro_shift = pro_shift; g.gmax = gmax; g.ldgmax = ldgmax; g.gshift = gshift; g.stat_part = stat_part;
    hipStream_t st = (hipStream_t)stream;
    const long long wgs256 = (long long)((M + 255) / 256) * ((N + 127) / 128);
    if (pro_scale) {
        g.l.tiles_n = (N + 127) / 128;
        if (wgs256 >= 192) {                       // (the 256-row tile with 1 x 4 blocks per wave: 383 -> 366 us for the chain at 65,536 rows)
            hipLaunchKernelGGL((linear_sb_kernel<8, 4, 4, 1, 2, 1>), dim3((unsigned)(((M + 255) / 256) * g.l.tiles_n)), dim3(512), 0, st, g);
        } else {
            hipLaunchKernelGGL((linear_sb_kernel<4, 4, 2, 1, UPP_SB_NST44, 1>), dim3((unsigned)(((M + 127) / 128) * g.l.tiles_n)), dim3(512), 0, st, g);
        }
        return upp_launch_status();
    }
    if (wgs256 >= 192) return launch_sb<8, 4, 4, 1, 2>(g, st);
    return launch_sb<4, 4, 2, 1, UPP_SB_NST44>(g, st);
}

extern "C" int upp_linear_sb_f32(const float *A, long long lda, const void *planes, const float *bias, float *C, long long ldc, float *aux,
                                 long long ldaux, int M, int N, int K, int epilogue, int tile, void *stream) {
    return linear_sb_launch(A, lda, planes, bias, 0, C, ldc, aux, ldaux, M, N, K, epilogue, tile, stream);
}

// C (M,N) = A . W^T + bias[m >> group_shift][:] -- a bias per GROUP of 2^group_shift >= 5 consecutive rows, as upp_linear_group_bias_f32.
extern "C" int upp_linear_sb_group_bias_f32(const float *A, long long lda, const void *planes, const float *bias, int group_shift, float *C,
                                            long long ldc, int M, int N, int K, void *stream) {
    if (!bias) return UPP_E_BADARG;
    if (group_shift < 5 || group_shift > 30) return UPP_E_RANGE;
    return linear_sb_launch(A, lda, planes, bias, group_shift, C, ldc, nullptr, 0, M, N, K, LEPI_BIAS, 0, stream);
}
