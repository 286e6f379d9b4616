// linear.hip -- the token-matrix Linear layers of the Transformer blocks (reference
// models/Point_MAE_pretask_dev.py:153-169 Mlp.fc1/fc2, :172-196 Attention.qkv/proj) and their
// data gradients as ONE exact-f32 MFMA GEMM family for gfx950:
//
//     C (M,N) = epilogue( A (M,K) . W (N,K)^T )            v_mfma_f32_32x32x2_f32, k-ordered fma chains
//
// These GEMMs have M = B*L = 1,120 ... 2,400 token rows and N, K in {384, 1152, 1536}: 420 ... 3,600 output blocks
// of 32x32 for the chip's 1,024 SIMDs, i.e. 0.4 ... 3.5 blocks per SIMD.  At that size the decomposition decides
// the time: every SIMD should own the same small number of blocks, and 2-4 waves must share a SIMD so that the
// matrix pipe keeps issuing while a wave waits for operands.  Structure:
//
//   * ONE 32x32 output block per wave.  A workgroup owns BMB x BNB blocks and, for the narrow-N shapes, splits the
//     contraction KS ways over wave groups: BMB*BNB*KS <= 16 waves (4 per SIMD).  The host picks (BMB, BNB, KS) per
//     (M, N, K) so that the workgroups fit the 256 CUs in one round with the fewest blocks per SIMD (pick_config).
//   * Operands reach the LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write): every wave-instruction
//     fetches 8 rows x 128 contiguous bytes (whole cache lines -- fragment-shaped loads straight into VGPRs,
//     32 lines per instruction, were measured TA-bound at ~10 B/clk/CU: 31-45 us where this kernel takes 22-27).
//     Two LDS stages: the DMA of k-stage c+1 is issued behind the barrier that opens k-stage c, between its MFMAs.
//   * Both operands are K-contiguous (nn.Linear stores W as (out, in)) and a sum may run in any order, so a lane
//     reads its MFMA operands as 16-byte granules (ds_read_b128): lane (h = lane >> 5, r = lane & 31) reads
//     A[r][8 i + 4 h ... + 3] and four consecutive MFMAs contract k = 8 i + j of the lower lane half with
//     8 i + 4 + j of the upper one (W alike).  The 16-byte granules of a row are XOR-swizzled with (row >> 1) & 7
//     -- applied to the SOURCE address of the DMA, whose LDS side is lane-linear -- so that every 16-lane group
//     of a ds_read_b128 covers all 64 banks.
//   * K-split partial tiles are summed through the LDS in wave-group order (deterministic); every wave group
//     finishes a quarter / half of the rows, so the epilogue (bias, exact-erf GELU and its derivative) is spread
//     over all waves and overlaps the other waves' MFMAs.
//   * Workgroups are renumbered so that each XCD (own L2) gets a contiguous run of tiles in row-panel order: the
//     big operand A is fetched by about one XCD per row panel, only the small W is read by all eight.
//   * Epilogues: bias; bias + GELU, optionally storing GELU' for the backward pass; multiply by a saved GELU'
//     (the data gradient of fc2 then IS the gradient w.r.t. the fc1 pre-activation).
#include "common.h"

namespace {

#include "linear_shared.h"

template <int BMB, int BNB, int KS, int KC, bool TAIL>
__global__ __launch_bounds__(BMB *BNB *KS * 64) void linear_f32_kernel(LinArgs g) {
    constexpr int NW = BMB * BNB * KS, BM = BMB * 32, BN = BNB * 32;
    constexpr int ROWS = (BM + BN) * KS * KC;       // 128-byte row images per stage: [ks][kc][A rows | W rows]
    constexpr int STAGE = ROWS * 128;
    constexpr int T = ROWS / 8;                     // DMA wave-instructions per stage (8 rows each)
    constexpr int TPW = (T + NW - 1) / NW;
    constexpr int RED = NW * 4096;                  // every wave's block, turned through the LDS on its way out
    constexpr int LDS_BYTES = 2 * STAGE > RED ? 2 * STAGE : RED;
    static_assert(LDS_BYTES <= 160 * 1024, "stages exceed the 160 KB of LDS");
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];     // (the only LDS object of the kernel)

    UPP_STAMP(0)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware, bijective renumbering: workgroups b and b + 8 share an XCD; give every XCD a contiguous run of tiles
    const int nwg = gridDim.x;
    const int orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, rem = nwg & 7;
    const int lin = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + (orig >> 3);
    const int by = lin / g.tiles_n, bx = lin - by * g.tiles_n;       // row-panel order: neighbours share the A rows of their panel
    const int m0 = by * BM, n0 = bx * BN;
    const int M = g.M, N = g.N;
    const int ks = wave / (BMB * BNB), wb = wave - ks * (BMB * BNB);
    const int bm = wb / BNB, bn = wb - bm * BNB;

    // ---- DMA sources: instruction t fills row images 8t .. 8t+7; lane -> (row image 8t + lane/8, granule lane%8).
    // Everything that depends on t only (sub-image, A or W, first row) is wave-uniform: scalar code, few VALU instructions.
    const float *src[TPW];
    int koff[TAIL ? TPW : 1];                                          // (TAIL) k of the lane's granule inside a k-stage
    const int nsc = TAIL ? (g.K + 32 * KS * KC - 1) / (32 * KS * KC) : g.K / (32 * KS * KC);
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int t = wave + q * NW;
        const int rho0 = (t < T ? t : 0) * 8;                         // (scalar) first row image of the instruction
        const int sk = rho0 / (BM + BN), rr0 = rho0 - sk * (BM + BN);  // (scalar) sub-image, first row in it: 8 | BM, so all 8 rows
        const bool isA = rr0 < BM;                                     //          of an instruction are on the same side
        const float *base = isA ? g.A : g.W;
        const long long ld = isA ? g.lda : g.ldw;
        const int first = isA ? m0 + rr0 : n0 + rr0 - BM, last = isA ? M - 1 : N - 1;
        const int row = min(first + (lane >> 3), last);
        const int ko = sk * 32 + 4 * ((lane & 7) ^ (((rho0 >> 1) + (lane >> 4)) & 7));
        if constexpr (TAIL) koff[q] = ko;
        src[q] = base + row * ld + ko;
    }
    static_assert(KC != 1 || TPW <= 6, "DMA slots of the interleaved schedule");
    auto issue1 = [&](int q, int stage, int c) {            // q-th DMA instruction of this wave for k-stage c
        const int t = wave + q * NW;
#ifndef UPP_LIN_NO_DMA        // diagnostic build without the operand stream (the MFMAs then run on whatever the LDS holds)
        if (TPW * NW == T || t < T) {
            const float *p = src[q < TPW ? q : 0] + (long long)c * (32 * KS * KC);
            // (contraction lengths that are not a multiple of the k-stage: granules beyond K come from a page of zeros)
            if constexpr (TAIL) {
                if (c == nsc - 1 && koff[q < TPW ? q : 0] + c * (32 * KS * KC) >= g.K) p = g_lin_zeros;
            }
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)p, (lds_ptr_t)(lds + stage * STAGE + t * 1024), 16, 0, 0);
        }
#else
        (void)t; (void)stage; (void)c;
#endif
    };
    auto issue = [&](int stage, int c) {
#pragma unroll
        for (int q = 0; q < TPW; ++q) issue1(q, stage, c);
    };

    // ---- fragment addresses: row image of lane's A row / W row, granule (2 i + h) ^ swizzle
    const int sw = (r >> 1) & 7;
    const int rowA = (ks * KC * (BM + BN) + bm * 32 + r) * 128, rowW = (ks * KC * (BM + BN) + BM + bn * 32 + r) * 128;
    int offA[4], offW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { offA[i] = rowA + (((2 * i + h) ^ sw) << 4); offW[i] = rowW + (((2 * i + h) ^ sw) << 4); }

    f32x16 acc;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = 0.0f;

    // The fragment reads are inline asm: to hipcc an LDS-DMA is a pending LDS write that any ds_read may alias, so it puts
    // s_waitcnt vmcnt(0) in front of compiler-visible reads -- which would expose the latency of the DMA just issued for a
    // LATER k-stage on every iteration.  Ordering is by hand (vmcnt / lgkmcnt / barriers below).
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;
    unsigned adrA[4], adrW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { adrA[i] = lds0 + offA[i]; adrW[i] = lds0 + offW[i]; }
#define UPP_READ_FRAG(F, SO)                                                          \
    asm volatile("ds_read_b128 %0, %1" : "=v"(F.a0) : "v"(adrA[0] + (SO)));           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(F.b0) : "v"(adrW[0] + (SO)));           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(F.a1) : "v"(adrA[1] + (SO)));           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(F.b1) : "v"(adrW[1] + (SO)));           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(F.a2) : "v"(adrA[2] + (SO)));           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(F.b2) : "v"(adrW[2] + (SO)));           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(F.a3) : "v"(adrA[3] + (SO)));           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(F.b3) : "v"(adrW[3] + (SO)));
#ifdef UPP_LIN_NO_MFMA
#define UPP_MFMA4(AV, BV) asm volatile("" ::"v"(AV), "v"(BV));
#else
#define UPP_MFMA4(AV, BV)                                                             \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[0], BV[0], acc, 0, 0, 0);           \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[1], BV[1], acc, 0, 0, 0);           \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[2], BV[2], acc, 0, 0, 0);           \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[3], BV[3], acc, 0, 0, 0);
#endif
#ifdef UPP_LIN_NO_MFMA      // diagnostic build: operands delivered and read, no matrix instructions (the values stay live)
#define UPP_M1(AV, BV) asm volatile("" ::"v"(AV), "v"(BV)); __builtin_amdgcn_sched_barrier(0);
#else
#define UPP_M1(AV, BV) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, BV, acc, 0, 0, 0); __builtin_amdgcn_sched_barrier(0);
#endif
#define UPP_RD1(DST, ADR) asm volatile("ds_read_b128 %0, %1" : "=v"(DST) : "v"(ADR));
#define UPP_LGKM(N_LEFT) asm volatile("s_waitcnt lgkmcnt(" #N_LEFT ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    struct Frag { f32x4 a0, a1, a2, a3, b0, b1, b2, b3; };

    // Two LDS stages.  Iteration c: "my share of k-stage c has landed" (vmcnt) + barrier -- behind it k-stage c is complete
    // in the LDS and every wave has finished reading the other stage, which is refilled with k-stage c + 1 while the MFMAs of
    // k-stage c run.  The DMA instructions go BETWEEN the first MFMAs: issued as one burst behind the barrier they cost every
    // wave of a SIMD 60-180 issue cycles each at the same moment, with the matrix pipe idle (64 x 64 tiles: 32.3 -> 29.0 us
    // at K = 1536); the 128 x 128 tile with 64-wide k-stages measured better with the burst (its iterations are twice as long).
    // Measured and dropped (tools/micro/lin_stamps.py): three / four stages with the DMA two / three k-stages
    // ahead (no gain: the DMA is not late), and operands of k-stage c + 1 read into a second register set during the MFMAs
    // of k-stage c (k-loop 82 -> 89 % MFMA duty, but the epilogue then bunches up: same launch time).
    issue(0, 0);
    for (int c = 0; c < nsc; ++c) {
        wait_vmcnt<0>();                                 // this wave's share of k-stage c has landed
        __builtin_amdgcn_s_barrier();                    // ... everyone's has; and everyone is done reading the other stage
        const bool fill = c + 1 < nsc;
        const int fst = (c + 1) & 1;
        if (KC >= 2 && fill) issue(fst, c + 1);
        if (c == 0) { UPP_STAMP(1) }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const unsigned so = (c & 1) * STAGE + kc * ((BM + BN) * 128);
            Frag f;
            UPP_READ_FRAG(f, so)
            UPP_LGKM(6)
            UPP_M1(f.a0[0], f.b0[0]) if (KC == 1 && fill) issue1(0, fst, c + 1);
            UPP_M1(f.a0[1], f.b0[1]) if (KC == 1 && fill && TPW > 1) issue1(1, fst, c + 1);
            UPP_M1(f.a0[2], f.b0[2]) if (KC == 1 && fill && TPW > 2) issue1(2, fst, c + 1);
            UPP_M1(f.a0[3], f.b0[3]) if (KC == 1 && fill && TPW > 3) issue1(3, fst, c + 1);
            UPP_LGKM(4)
            UPP_M1(f.a1[0], f.b1[0]) if (KC == 1 && fill && TPW > 4) issue1(4, fst, c + 1);
            UPP_M1(f.a1[1], f.b1[1]) if (KC == 1 && fill && TPW > 5) issue1(5, fst, c + 1);
            UPP_M1(f.a1[2], f.b1[2]) UPP_M1(f.a1[3], f.b1[3])
            UPP_LGKM(2) UPP_MFMA4(f.a2, f.b2)
            UPP_LGKM(0) UPP_MFMA4(f.a3, f.b3)
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef UPP_READ_FRAG
#undef UPP_MFMA4
#undef UPP_LGKM
#undef UPP_M1
#undef UPP_RD1

    UPP_STAMP(2)
    // ---- tile -> memory.  Register t of a block = C[row][col]: col = lane & 31, row = (t & 3) + 8 (t >> 2) + 4 (lane >> 5): stored
    // from the accumulators a block is 16 dword stores of 2 x 128 bytes; the store burst of 16 waves at the end of a launch was
    // 3.5 us of the 30 us fc1 launch (round 2 stamps).  Every block goes through the LDS instead -- which the K-split shapes need
    // anyway for their partial tiles -- as [wave][t][lane]: for a fixed row, 4 consecutive columns are 16 contiguous bytes, and a
    // wave group's share of a block (registers [T0, T0 + TN): 2 TN rows) is one lane-linear ds_read_b128 sweep.  The KS partial
    // tiles are summed in wave-group order from 0.0f on the way (deterministic; oracle_linear_f32), bias / activation / factor
    // act on four columns per lane, and a wave stores 4 x fewer, 16-byte-per-lane instructions.
    constexpr int TN = 16 / KS;
    const int T0 = ks * TN;
    const int rb = m0 + bm * 32, cb = n0 + bn * 32;                         // (scalar) block origin
    const bool wide = ((g.ldc | N | g.ldaux) & 3) == 0 && ((reinterpret_cast<uintptr_t>(g.C) | reinterpret_cast<uintptr_t>(g.aux)) & 15) == 0;
    if (wide) {
        __syncthreads();                                     // all fragment reads done: the stages may be overwritten
        float *red = reinterpret_cast<float *>(lds);
#pragma unroll
        for (int t = 0; t < 16; ++t) red[(wave * 16 + t) * 64 + lane] = acc[t];
        if (KS > 1) __syncthreads();
        const int col = cb + 4 * (lane & 7);
        const bool col_ok = col < N;
        f32x4 bias4 = {0.0f, 0.0f, 0.0f, 0.0f};
        if (g.epi != LEPI_NONE && g.epi != LEPI_MUL && col_ok) bias4 = *reinterpret_cast<const f32x4 *>(g.bias + col);
#pragma unroll
        for (int pass = 0; pass < TN / 4; ++pass) {
            const int ridx = pass * 8 + (lane >> 3), t = T0 + (ridx >> 1);
            const int row = rb + (t & 3) + 8 * (t >> 2) + 4 * (ridx & 1);
            const float *src = red + (wb * 16 + T0) * 64 + (pass * 64 + lane) * 4;
            f32x4 v = *reinterpret_cast<const f32x4 *>(src);
            if (KS > 1) {
                const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
                v = zero + v;
#pragma unroll
                for (int k2 = 1; k2 < KS; ++k2) v += *reinterpret_cast<const f32x4 *>(src + k2 * (BMB * BNB) * 1024);
            }
            epilogue_store4(g, g.epi, v, bias4, row, col, col_ok && row < M);
        }
        UPP_STAMP(3)
        return;
    }
    // rows that are not 16-byte aligned (N = 3, 50, ...): dword stores from the accumulators
    float outv[TN];
    if (KS > 1) {
        __syncthreads();                                     // all fragment reads done: the stages may be overwritten
        float *red = reinterpret_cast<float *>(lds);
#pragma unroll
        for (int t = 0; t < 16; ++t) red[(wave * 16 + t) * 64 + lane] = acc[t];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < TN; ++u) {
            float s = 0.0f;
#pragma unroll
            for (int k2 = 0; k2 < KS; ++k2) s += red[((k2 * (BMB * BNB) + wb) * 16 + T0 + u) * 64 + lane];
            outv[u] = s;
        }
    } else {
#pragma unroll
        for (int u = 0; u < TN; ++u) outv[u] = acc[u];
    }

    // ---- epilogue.  register t of a block = C[row][col]: col = lane & 31, row = (t & 3) + 8 (t >> 2) + 4 (lane >> 5).
    // Written for instruction COUNT: 16 waves share the CU's four issue ports, a wave's epilogue runs beside the other waves'
    // last MFMAs, and the first version (one runtime switch and 64-bit address arithmetic per element: ~1,000 instructions
    // per wave) took 3.8 us of the 31 us fc1 launch WITHOUT its stores.  Now: a wave-uniform block base (scalar), one 32-bit
    // lane offset, a switch outside the element loop, and per-element guards only on edge tiles.
    const bool full = rb + 32 <= M && cb + 32 <= N;                          // (scalar)
    const int ldc = (int)g.ldc, ldx = (int)g.ldaux;
    float *cblk = g.C + (long long)rb * g.ldc + cb;
    float *xblk = g.aux ? g.aux + (long long)rb * g.ldaux + cb : nullptr;
    const int row_l = 4 * h, lc = row_l * ldc + r, lx = row_l * ldx + r;     // lane offsets inside the block
    const bool col_ok = cb + r < N;
    const float bias = (g.epi == LEPI_BIAS || g.epi == LEPI_BIAS_GELU || g.epi == LEPI_BIAS_GELU_D || g.epi == LEPI_BIAS_RELU) ? g.bias[min(cb + r, N - 1)] : 0.0f;
#define UPP_ROWOF(u) (((T0 + (u)) & 3) + 8 * ((T0 + (u)) >> 2))
#define UPP_EPI_LOOP(...)                                                                  \
    if (full) {                                                                             \
        _Pragma("unroll") for (int u = 0; u < TN; ++u) { const int rr = UPP_ROWOF(u); __VA_ARGS__ } \
    } else {                                                                                \
        _Pragma("unroll") for (int u = 0; u < TN; ++u) {                                    \
            const int rr = UPP_ROWOF(u);                                                    \
            if (col_ok && rb + row_l + rr < M) { __VA_ARGS__ }                                     \
        }                                                                                   \
    }
    switch (g.epi) {
        case LEPI_NONE:
        case LEPI_BIAS:
            UPP_EPI_LOOP(cblk[lc + rr * ldc] = outv[u] + bias;)
            break;
        case LEPI_BIAS_GELU:
            UPP_EPI_LOOP(float gv, dv; gelu_pair(outv[u] + bias, gv, dv); cblk[lc + rr * ldc] = gv;)
            break;
        case LEPI_BIAS_GELU_D:
            UPP_EPI_LOOP(float gv, dv; gelu_pair(outv[u] + bias, gv, dv); cblk[lc + rr * ldc] = gv; xblk[lx + rr * ldx] = dv;)
            break;
        case LEPI_BIAS_RELU:
            UPP_EPI_LOOP(cblk[lc + rr * ldc] = fmaxf(outv[u] + bias, 0.0f);)
            break;
        default: {                                               // LEPI_MUL: all factor loads in flight before the first use
            float fac[TN];
#pragma unroll
            for (int u = 0; u < TN; ++u) {
                const int rr = UPP_ROWOF(u);
                fac[u] = (full || (col_ok && rb + row_l + rr < M)) ? xblk[lx + rr * ldx] : 0.0f;
            }
            UPP_EPI_LOOP(cblk[lc + rr * ldc] = outv[u] * fac[u];)
        }
    }
#undef UPP_EPI_LOOP
#undef UPP_ROWOF
    UPP_STAMP(3)
}

template <int BMB, int BNB, int KS, int KC>
int launch_linear(const LinArgs &g0, hipStream_t st) {
    // (the zero-tail variant is a separate instantiation: its extra registers and branch cost the common case 15-20 % when
    // compiled into the same kernel -- measured: 22.0 -> 26.3 us per launch on the narrow-N shapes)
    LinArgs g = g0;
    const int tiles_m = (g.M + BMB * 32 - 1) / (BMB * 32);
    g.tiles_n = (g.N + BNB * 32 - 1) / (BNB * 32);
    const dim3 grid((unsigned)(tiles_m * g.tiles_n));
    if (g.ktail) hipLaunchKernelGGL((linear_f32_kernel<BMB, BNB, KS, KC, true>), grid, dim3(BMB * BNB * KS * 64), 0, st, g);
    else hipLaunchKernelGGL((linear_f32_kernel<BMB, BNB, KS, KC, false>), grid, dim3(BMB * BNB * KS * 64), 0, st, g);
    return upp_launch_status();
}

struct LinConfig { int bmb, bnb, ks, kc; };
#define UPP_LIN_CONFIGS(X) X(4, 4, 1, 2) X(4, 3, 1, 1) X(3, 4, 1, 1) X(2, 4, 2, 1) X(2, 3, 2, 1) X(2, 2, 4, 1) X(1, 2, 4, 1) X(2, 2, 2, 1)
#define UPP_LIN_ENTRY(a, b, c, d) {a, b, c, d},
constexpr LinConfig kConfigs[] = {UPP_LIN_CONFIGS(UPP_LIN_ENTRY)};
#undef UPP_LIN_ENTRY
constexpr int kNumConfigs = sizeof(kConfigs) / sizeof(kConfigs[0]);
inline int config_code(const LinConfig &c) { return c.bmb * 4096 + c.bnb * 256 + c.ks * 16 + c.kc; }

// Choice of the decomposition (measured on MI355X, tools/time_linear.py): a workgroup per CU in ONE
// round beats everything else at these sizes, so among the shapes whose workgroups fit the 256 CUs take the one with the
// fewest MFMAs per SIMD -- ceil(waves / 4) waves per SIMD, each with 1 / KS of a block --, then the one with more waves per
// SIMD (they cover each other's barrier and LDS latency), then fewer idle CUs.  Larger problems
// (the 65,536-row layers of the segmentation head) run the 128 x 128 tile in several rounds.
int pick_config(int M, int N, int K) {
    const int mb = (M + 31) / 32, nb = (N + 31) / 32;
    int best = -1;
    long long best_cost = 0;
    for (int i = 0; i < kNumConfigs; ++i) {
        const LinConfig c = kConfigs[i];
        const long long wgs = (long long)((mb + c.bmb - 1) / c.bmb) * ((nb + c.bnb - 1) / c.bnb);
        const int waves = c.bmb * c.bnb * c.ks;
        const long long rounds = (wgs + 255) / 256;
        const long long ksteps = (K + 32 * c.ks * c.kc - 1) / (32 * c.ks * c.kc) * c.kc;         // 32-wide k-steps per wave, padding included
        const long long quarters = rounds * ((waves + 3) / 4) * ksteps;                          // MFMA work per SIMD (x 16 MFMAs)
        const long long traffic = 100LL * (c.bmb + c.bnb) / (c.bmb * c.bnb);                    // staged rows per block
        long long cost;
        if (rounds == 1) cost = quarters * 1000000LL + (16 - waves) * 10000LL + (256 - wgs);
        else cost = 1000000000LL + quarters * 1000000LL + traffic * 1000LL;
        if (best < 0 || cost < best_cost) { best = i; best_cost = cost; }
    }
    // ((2,2,2,64) -- the 64 x 64 tile with the contraction split 2 ways -- is never the one-round choice: 64 KB of LDS lets a workgroup of the
    // other stream's GEMM onto the same CU, measured 0.28 ms slower per pipelined step; pick_multi_round takes it for 4,000 ... 4,500 rows)
    return best;
}

// Tall matrices (thousands of 128 x 128 tiles: the B N point rows of the segmentation head, a trainable patch embedding) go to the
// register-tiled kernel of linear_rt.hip: 2 x 2 waves of 2 x 2 blocks, two workgroups per CU.  Measured at 65,536 x 1024 x 1536:
// 139 TFLOP/s against 126 for the (4,4,1,64) shape above (one block per wave, one workgroup per CU, prologue and store burst of
// every tile exposed).  Code: 0x1000000 NST + 0x100000 (4 (RM - 1) + RN - 1) + 0x10000 + 4096 WM + 256 WN + 16 + KC -- the low 16 bits
// read like a (BMB, BNB, KS = 1, KC) code of this file, which is also the summation order (oracle_linear_f32 with ks = 1).
constexpr int kRtTall = 0x2000000 + 0x100000 * 5 + 0x10000 + 4096 * 2 + 256 * 2 + 16 + 1;
int pick_rt(int M, int N, int K) {
    const long long tiles = (long long)((M + 127) / 128) * ((N + 127) / 128);
    return tiles >= 1024 && N >= 128 && K >= 32 ? kRtTall : 0;      // at least two rounds of the 512 resident workgroups
}

// Problems that fit no decomposition into one round of 256 workgroups and are not tall enough for pick_rt (the Transformer blocks at
// 2,700 ... 9,000 token rows: part segmentation 4,416, the auxiliary-task recipes 4,128 / 4,448): estimate, per candidate,
//   rounds x (MFMA time of the SIMD's share of a CU's resident workgroups + prologue / epilogue that nothing covers),
// rounds = ceil(workgroups / (256 x resident workgroups per CU)) -- 64 KB workgroups (8 waves, and linear_rt.hip's 4 waves of 2 x 2
// blocks) sit two to a CU and cover each other's prologue / epilogue, 96-128 KB ones sit alone and pay ~5 us per round.  Fitted to
// tools/time_linear.py --tiles --rows 2720,3072,4128,4416,6144,8832 (ranks the measured order within ~10 %); the cost model of
// pick_config -- made for one round -- took (1,2,4,128) at fc1 x 4,416 rows: 76.5 us against 51.8 for (2,2,2,64) chosen here.
int pick_multi_round(int M, int N, int K) {
    const int mb = (M + 31) / 32, nb = (N + 31) / 32;
    int best = 0;
    double best_t = 0.0;
    auto consider = [&](int code, long long wgs, int per_cu, int waves_per_simd, int blocks_per_wave, long long ksteps, double overhead_us) {
        const long long rounds = (wgs + 256LL * per_cu - 1) / (256LL * per_cu);
        const double mfma_us = (double)per_cu * waves_per_simd * blocks_per_wave * (double)ksteps * 1024.0 / 2400.0;   // 16 MFMAs x 64 cycles per k-step
        const double t = (double)rounds * (mfma_us + overhead_us);
        if (!best || t < best_t) { best = code; best_t = t; }
    };
    for (int i = 0; i < kNumConfigs; ++i) {
        const LinConfig c = kConfigs[i];
        const int waves = c.bmb * c.bnb * c.ks;
        const long long wgs = (long long)((mb + c.bmb - 1) / c.bmb) * ((nb + c.bnb - 1) / c.bnb);
        const long long ksteps = (K + 32 * c.ks * c.kc - 1) / (32 * c.ks * c.kc) * c.kc;
        // measured per-round residue: 5 us alone on the CU, 3 us with a partner, 5.5 us for the two-block tile whose partner is as short
        consider(config_code(c), wgs, waves <= 8 ? 2 : 1, (waves + 3) / 4, 1, ksteps, waves > 8 ? 5.0 : (c.bmb * c.bnb >= 4 ? 3.0 : 5.5));
    }
    if (N >= 128 && K >= 32) consider(kRtTall, (long long)((M + 127) / 128) * ((N + 127) / 128), 2, 1, 4, (K + 31) / 32, 3.0);
    return best;
}

// The decomposition upp_linear_f32 uses when the caller leaves the choice to it (tile <= 0), as a tile code.
int pick_code(int M, int N, int K) {
    if (const int rt = pick_rt(M, N, K)) return rt;
    const int i = pick_config(M, N, K);
    if (i < 0) return 0;
    const LinConfig c = kConfigs[i];
    const int mb = (M + 31) / 32, nb = (N + 31) / 32;
    const long long wgs = (long long)((mb + c.bmb - 1) / c.bmb) * ((nb + c.bnb - 1) / c.bnb);
    return wgs <= 256 ? config_code(c) : pick_multi_round(M, N, K);
}

}  // namespace

__attribute__((visibility("hidden"))) int upp_detail_linear_rt(const void *args, int code, hipStream_t st);

#ifdef UPP_LIN_STAMPS
extern "C" void upp_linear_set_stamps(unsigned long long *p) { g_lin_stamps = p; }
#endif

extern "C" int upp_linear_tile(int M, int N, int K) {
    if (M < 1 || N < 1 || K < 1) return UPP_E_BADARG;
    if (K % 4 != 0) return UPP_E_RANGE;
    const int code = pick_code(M, N, K);
    return code ? code : UPP_E_RANGE;
}

extern "C" int upp_linear_f32(const float *A, long long lda, const float *W, long long ldw, const float *bias, float *C, long long ldc,
                              float *aux, long long ldaux, int M, int N, int K, int epilogue, int tile, void *stream) {
    if (!A || !W || !C || M < 1 || N < 1 || K < 1) return UPP_E_BADARG;
    if (K % 4 != 0 || lda % 4 != 0 || ldw % 4 != 0 || lda < K || ldw < K || ldc < N) return UPP_E_RANGE;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) & 15) return UPP_E_RANGE;
    if (ldc > (1LL << 24) || ldaux > (1LL << 24)) return UPP_E_RANGE;          // 32-bit offsets inside a 32-row block
    if (epilogue < LEPI_NONE || epilogue > LEPI_BIAS_RELU) return UPP_E_RANGE;
    if ((epilogue == LEPI_BIAS || epilogue == LEPI_BIAS_GELU || epilogue == LEPI_BIAS_GELU_D || epilogue == LEPI_BIAS_RELU) && !bias) return UPP_E_BADARG;
    if ((epilogue == LEPI_BIAS_GELU_D || epilogue == LEPI_MUL) && (!aux || ldaux < N)) return UPP_E_BADARG;
    LinArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.bias = bias; g.aux = aux; g.ldaux = ldaux;
    g.M = M; g.N = N; g.K = K; g.epi = epilogue; g.wt = upp_store_policy();
#ifdef UPP_LIN_STAMPS
    g.stamps = g_lin_stamps;
#endif
    hipStream_t st = (hipStream_t)stream;
    if (tile <= 0) tile = pick_code(M, N, K);
    if (tile <= 0) return UPP_E_RANGE;
    if (tile & 0x10000) return upp_detail_linear_rt(&g, tile, st);
    g.ktail = K % (32 * ((tile >> 4) & 15) * (tile & 15)) != 0;
#define UPP_LIN_CASE(a, b, c, d) case a * 4096 + b * 256 + c * 16 + d: return launch_linear<a, b, c, d>(g, st);
    switch (tile) {
        UPP_LIN_CONFIGS(UPP_LIN_CASE)
        default: return UPP_E_RANGE;
    }
#undef UPP_LIN_CASE
}

// C (M,N) = A . W^T + bias[m >> group_shift][:]  -- a bias per GROUP of 2^group_shift consecutive rows (group_shift >= 5), added in the
// epilogue of the register-tiled kernel; only for problems that kernel is chosen for (upp_linear_tile(M, N, K) & 0x10000) with 16-byte
// aligned output rows, UPP_E_RANGE otherwise (the caller then adds the broadcast term itself).
extern "C" int upp_linear_group_bias_f32(const float *A, long long lda, const float *W, long long ldw, const float *bias, int group_shift,
                                         float *C, long long ldc, int M, int N, int K, void *stream) {
    if (!A || !W || !C || !bias || M < 1 || N < 1 || K < 1) return UPP_E_BADARG;
    if (K % 4 != 0 || lda % 4 != 0 || ldw % 4 != 0 || lda < K || ldw < K || ldc < N) return UPP_E_RANGE;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias)) & 15) return UPP_E_RANGE;
    if (group_shift < 5 || group_shift > 30 || N % 4 != 0 || ldc % 4 != 0 || ldc > (1LL << 24)) return UPP_E_RANGE;
    const int tile = pick_code(M, N, K);
    if (tile <= 0 || !(tile & 0x10000)) return UPP_E_RANGE;                    // (not a problem the register-tiled kernel is chosen for)
    LinArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.bias = bias; g.M = M; g.N = N; g.K = K; g.epi = LEPI_BIAS;
    g.bias_shift = group_shift;
#ifdef UPP_LIN_STAMPS
    g.stamps = g_lin_stamps;
#endif
    return upp_detail_linear_rt(&g, tile, (hipStream_t)stream);
}
