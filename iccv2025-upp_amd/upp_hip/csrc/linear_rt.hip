// linear_rt.hip -- register-tiled members of the exact-f32 MFMA Linear family (see linear.hip for the token-matrix kernel):
//
//     C (M,N) = epilogue( A (M,K) . W (N,K)^T )            v_mfma_f32_32x32x2_f32, k-ordered fma chains
//
// for the TALL matrices of the path -- the B*N point rows of the part-segmentation head (reference
// models/Point_MAE_unify_segment.py:420 `propagation_0`, :424-433 `seg_head`: 65,536 rows at B = 32) and of a trainable patch
// embedding -- where the problem is thousands of 128 x 128 tiles, not one round of workgroups.  What differs from linear.hip:
//
//   * A wave owns RM x RN blocks of 32 x 32 (2 x 2: 64 accumulator registers), so every 16-byte operand granule read from
//     the LDS feeds RN (RM) x 4 MFMAs instead of 4: half the LDS bytes per MFMA -- less energy per flop, the lever the chip's
//     clock responds to in an MFMA-dense loop -- and 16 MFMAs (1,024 issue cycles) between two groups of fragment reads.
//   * Workgroups are SMALL (WM x WN = 2 x 2 waves, 64 KB of LDS) so that two of them share a CU at different phases of
//     their lives: one workgroup's prologue (first DMA from HBM) and its store burst run beside the other's MFMAs.  The
//     hardware dispatcher refills a CU as soon as a workgroup retires -- no persistent loop, no tile queue.
//   * Fragment reads for step s + 1 are issued before the MFMAs of step s (two register sets, counted lgkmcnt).
//
// Same operand delivery as linear.hip: LDS-DMA (global_load_lds_dwordx4) of 8 rows x 128 contiguous bytes per
// wave-instruction, the 16-byte granules of a row XOR-swizzled on the SOURCE address, two LDS stages, DMA instructions issued
// between the MFMAs of the running stage.  Same summation order as a KS = 1 decomposition of linear.hip (oracle_linear_f32
// with ks = 1): lower lane half k = 8 i + j, upper 8 i + 4 + j, i, j ascending, 32-wide groups ascending.
#include "common.h"
#include <type_traits>

namespace {

#include "linear_shared.h"

template <int WM, int WN, int RM, int RN, int KC, int NST, bool TAIL, bool FANCY>
__global__ __launch_bounds__(WM *WN * 64) void linear_rt_kernel(LinArgs g) {
    constexpr int NW = WM * WN, BM = WM * RM * 32, BN = WN * RN * 32;
    constexpr int IMG = (BM + BN) * 128;            // one 32-wide k image: [A rows | W rows] x 128 bytes
    constexpr int STAGE = IMG * KC;
    constexpr int T = STAGE / 1024;                 // DMA wave-instructions per stage (8 row images each)
    constexpr int TPW = (T + NW - 1) / NW;
    constexpr int STEPS = 4 * KC;                   // fragment steps per stage: 8 values of k each
    static_assert(NST * STAGE <= 160 * 1024, "stages exceed the 160 KB of LDS");
    static_assert(TPW <= STEPS * 4, "more DMA instructions per wave than slots between the MFMA groups");
    static_assert(NST == 2 || TPW * NW == T, "a counted vmcnt needs the same number of DMA instructions in every wave");
    __shared__ __attribute__((aligned(1024))) char lds[NST * STAGE];     // (the only LDS object of the kernel)

    UPP_STAMP(0)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    const int lin = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    const int by = lin / g.tiles_n, bx = lin - by * g.tiles_n;      // row-panel order: the column tiles of a row panel run together
    const int m0 = by * BM, n0 = bx * BN;
    const int M = g.M, N = g.N;
    const int wm = wave / WN, wn = wave - wm * WN;

    // ---- DMA sources (as linear.hip): instruction t fills row images 8t .. 8t+7; lane -> (image 8t + lane/8, granule lane%8)
    const float *src[TPW];
    int koff[TAIL ? TPW : 1];
    const int nsc = TAIL ? (g.K + 32 * KC - 1) / (32 * KC) : g.K / (32 * KC);
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int t = wave + q * NW;
        const int rho0 = (t < T ? t : 0) * 8;
        const int kc = rho0 / (BM + BN), rr0 = rho0 - kc * (BM + BN);
        const bool isA = rr0 < BM;
        const float *base = isA ? g.A : g.W;
        const long long ld = isA ? g.lda : g.ldw;
        const int first = isA ? m0 + rr0 : n0 + rr0 - BM, last = isA ? M - 1 : N - 1;
        const int row = min(first + (lane >> 3), last);
        const int ko = kc * 32 + 4 * ((lane & 7) ^ (((rho0 >> 1) + (lane >> 4)) & 7));
        if constexpr (TAIL) koff[q] = ko;
        src[q] = base + row * ld + ko;
    }
    auto issue1 = [&](int q, int stage, int c) {
        const int t = wave + q * NW;
        if (TPW * NW == T || t < T) {
            const float *p = src[q] + (long long)c * (32 * KC);
            if constexpr (TAIL) {
                if (c == nsc - 1 && koff[q] + c * (32 * KC) >= g.K) p = g_lin_zeros;
            }
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)p, (lds_ptr_t)(lds + stage * STAGE + t * 1024), 16, 0, 0);
        }
    };

    // ---- fragment addresses (stage 0, image 0, the wave's first block of each operand)
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;
    const int sw = (r >> 1) & 7;
    unsigned adrA[4], adrW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        adrA[i] = lds0 + (wm * RM * 32 + r) * 128 + (((2 * i + h) ^ sw) << 4);
        adrW[i] = lds0 + (BM + wn * RN * 32 + r) * 128 + (((2 * i + h) ^ sw) << 4);
    }

    f32x16 acc[RM][RN];
#pragma unroll
    for (int a = 0; a < RM; ++a)
#pragma unroll
        for (int b = 0; b < RN; ++b)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[a][b][t] = 0.0f;

    struct Frag { f32x4 a[RM], b[RN]; };
    // step s of a stage = (image s / 4, granule pair s % 4); blocks of a wave are 32 row images = 4,096 bytes apart
#define UPP_RT_READ(F, SB, S)                                                                                        \
    {                                                                                                                \
        _Pragma("unroll") for (int a_ = 0; a_ < RM; ++a_)                                                            \
            asm volatile("ds_read_b128 %0, %1" : "=v"(F.a[a_]) : "v"(adrA[(S) & 3] + (SB) + ((S) >> 2) * IMG + a_ * 4096)); \
        _Pragma("unroll") for (int b_ = 0; b_ < RN; ++b_)                                                            \
            asm volatile("ds_read_b128 %0, %1" : "=v"(F.b[b_]) : "v"(adrW[(S) & 3] + (SB) + ((S) >> 2) * IMG + b_ * 4096)); \
    }
#define UPP_RT_LGKM(N_LEFT) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_LEFT) : "memory"); __builtin_amdgcn_sched_barrier(0);
#ifdef UPP_RT_NO_MFMA       // diagnostic build: operands delivered and read, no matrix instructions (the values stay live)
#define UPP_RT_MFMA(AV, BV, ACC) asm volatile("" ::"v"(AV), "v"(BV));
#else
#define UPP_RT_MFMA(AV, BV, ACC) ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, BV, ACC, 0, 0, 0);
#endif

    // NST LDS stages, the DMA runs NST - 1 k-stages ahead.  The barrier that opens k-stage c sits in front of the LAST fragment
    // step of k-stage c - 1: its operands are in registers by then, so a wave arrives having finished every LDS read of stage
    // c - 1, issues the reads of step 0 of stage c right behind the barrier and covers their latency with the 4 RM RN MFMAs it
    // still owes stage c - 1 (a barrier followed by reads followed by the first MFMA leaves the matrix pipe idle for an LDS round
    // trip per stage: measured 7 % of a lone wave's loop).  Behind that barrier the buffer of stage c - 1 is free: the DMA of
    // stage c + NST - 1 goes there, one instruction after each group of RM RN MFMAs.
    Frag f[2];
    auto mfma_step = [&](const Frag &fr, int slot0, int c_fill, auto fill_c) {
        constexpr bool FILL = decltype(fill_c)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int a = 0; a < RM; ++a)
#pragma unroll
                for (int b = 0; b < RN; ++b)
                    UPP_RT_MFMA(fr.a[a][j], fr.b[b][j], acc[a][b])
#ifndef UPP_RT_NO_DMA       // (diagnostic build without the operand stream: the MFMAs run on whatever the LDS holds)
            if constexpr (FILL) {
                if (slot0 + j < TPW) issue1(slot0 + j, c_fill % NST, c_fill);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // one k-stage: [owed last step of stage c - 1] steps 0 .. STEPS - 2 of stage c, the reads of its last step, "stage c + 1 has landed"
    auto stage = [&](int c, auto tail_c, auto fill_c) {
        constexpr bool TAILSTEP = decltype(tail_c)::value;
        const unsigned sb = (unsigned)((c % NST) * STAGE);
        UPP_RT_READ(f[0], sb, 0)
        if constexpr (TAILSTEP) mfma_step(f[(STEPS - 1) & 1], 0, c + NST - 1, fill_c);
#pragma unroll
        for (int s = 0; s + 1 < STEPS; ++s) {
            UPP_RT_READ(f[(s + 1) & 1], sb, s + 1)
            UPP_RT_LGKM(RM + RN)
            mfma_step(f[s & 1], TAILSTEP ? 4 * (s + 1) : 4 * s, c + NST - 1, fill_c);
        }
        UPP_RT_LGKM(0)
    };
    static_assert(TPW <= 4 * (STEPS - 1), "DMA slots of the first k-stage");
#pragma unroll
    for (int p = 0; p < NST - 1; ++p) {
        if (p < nsc) {
#pragma unroll
            for (int q = 0; q < TPW; ++q) issue1(q, p, p);
        }
    }
    const int nfill = nsc - (NST - 1);                  // k-stages behind which another one is still to be fetched
    UPP_STAMP(1)
    if (nfill > 0) wait_vmcnt<TPW *(NST - 2)>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (nfill > 0) stage(0, std::false_type{}, std::true_type{}); else stage(0, std::false_type{}, std::false_type{});
    int c = 1;
    for (; c < nfill; ++c) {
        wait_vmcnt<TPW *(NST - 2)>();
        __builtin_amdgcn_s_barrier();
        stage(c, std::true_type{}, std::true_type{});
    }
    for (; c < nsc; ++c) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        stage(c, std::true_type{}, std::false_type{});
    }
    mfma_step(f[(STEPS - 1) & 1], 0, 0, std::false_type{});
#undef UPP_RT_READ
#undef UPP_RT_LGKM
#undef UPP_RT_MFMA
    UPP_STAMP(2)

    // ---- epilogue.  register t of a block = C[row][col]: col = lane & 31, row = (t & 3) + 8 (t >> 2) + 4 (lane >> 5): a store
    // straight from the accumulators is 64 dword stores per wave (2 x 128 bytes each) -- measured 280-900 cycles per store
    // instruction, 18,000-60,000 cycles per workgroup, and a successor workgroup's first DMA queues behind them.  So the wave
    // turns its tile through the LDS, one row of blocks (32 x 32 RN) at a time -- ds_write_b32 per accumulator register,
    // lane-linear ds_read_b128 -- and stores 16 bytes per lane: 4 x fewer, 4 x wider store instructions, every row segment 128 RN
    // contiguous bytes; bias / activation / GELU' / factor are applied to the turned values (4 consecutive columns per lane).
    const int epi = g.epi;
    const int cw0 = n0 + wn * RN * 32;                                              // (scalar) first column of the wave
    const bool wide = ((g.ldc | N | g.ldaux) & 3) == 0 && ((reinterpret_cast<uintptr_t>(g.C) | reinterpret_cast<uintptr_t>(g.aux)) & 15) == 0;
    if (wide) {
        constexpr bool FREEBUF = STAGE >= NW * RN * 4096;           // the stage buffer nobody reads any more holds every wave's turn-table
        constexpr int CW = RN * 32, PER = CW / 4;                   // columns of the wave; 16-byte pieces per row
        if constexpr (!FREEBUF) __syncthreads();
        float *tt = reinterpret_cast<float *>(lds + (FREEBUF ? (nsc % NST) * STAGE : 0) + wave * (RN * 4096));
        const int piece = lane % PER, prow = lane / PER;            // this lane's piece of a row; 64 / PER rows per read instruction
        const int col = cw0 + 4 * piece;
        const bool col_ok = col < N;
        f32x4 bias4 = {0.0f, 0.0f, 0.0f, 0.0f};
        if (epi != LEPI_NONE && epi != LEPI_MUL && col_ok) bias4 = *reinterpret_cast<const f32x4 *>(g.bias + col);
#pragma unroll
        for (int a = 0; a < RM; ++a) {
            const int rb = m0 + (wm * RM + a) * 32;                                 // (scalar) first row of this row of blocks
            if constexpr (FANCY) {      // a bias per group of 2^bias_shift >= 32 rows: one row of the bias matrix per 32-row block
                if (g.bias_shift > 0 && col_ok && rb < M) bias4 = *reinterpret_cast<const f32x4 *>(g.bias + (size_t)(rb >> g.bias_shift) * N + col);
            }
#pragma unroll
            for (int b = 0; b < RN; ++b)
#pragma unroll
                for (int u = 0; u < 16; ++u) tt[((u & 3) + 8 * (u >> 2) + 4 * h) * CW + 32 * b + r] = acc[a][b][u];
#pragma unroll
            for (int i = 0; i < RN * 4; ++i) {
                const int row = rb + i * (64 / PER) + prow;
                f32x4 v = *reinterpret_cast<const f32x4 *>(tt + (i * 64 + lane) * 4);
                if constexpr (FANCY) epilogue_store4(g, epi, v, bias4, row, col, col_ok && row < M);
                else epilogue_store4(g, epi == LEPI_BIAS_RELU ? LEPI_BIAS_RELU : LEPI_BIAS, v, bias4, row, col, col_ok && row < M);
            }
        }
    } else {
        // rows that are not 16-byte aligned (N = 50: the last layer of the segmentation head): dword stores from the accumulators
        const int ldc = (int)g.ldc, ldx = (int)g.ldaux;
        const int row_l = 4 * h;
#pragma unroll
        for (int a = 0; a < RM; ++a) {
#pragma unroll
            for (int b = 0; b < RN; ++b) {
                const int rb = m0 + (wm * RM + a) * 32, cb = cw0 + b * 32;                 // (scalar) block origin
                if (rb >= M || cb >= N) continue;
                float *cblk = g.C + (long long)rb * g.ldc + cb;
                const int lc = row_l * ldc + r;
                const int rows_left = cb + r < N ? M - rb - row_l : 0;                    // rows rr < rows_left of this lane exist
                const float bias = epi != LEPI_NONE && epi != LEPI_MUL ? g.bias[min(cb + r, N - 1)] : 0.0f;
#define UPP_ROWOF(u) (((u) & 3) + 8 * ((u) >> 2))
                if (!FANCY || epi == LEPI_NONE || epi == LEPI_BIAS || epi == LEPI_BIAS_RELU) {
                    const bool relu = epi == LEPI_BIAS_RELU;
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int rr = UPP_ROWOF(u);
                        float v = acc[a][b][u] + bias;
                        v = relu ? fmaxf(v, 0.0f) : v;
                        if (rr < rows_left) cblk[lc + rr * ldc] = v;
                    }
                } else if constexpr (FANCY) {
                    float *xblk = g.aux + (long long)rb * g.ldaux + cb;
                    const int lx = row_l * ldx + r;
                    if (epi == LEPI_MUL) {
                        float fac[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) { const int rr = UPP_ROWOF(u); fac[u] = rr < rows_left ? xblk[lx + rr * ldx] : 0.0f; }
#pragma unroll
                        for (int u = 0; u < 16; ++u) { const int rr = UPP_ROWOF(u); if (rr < rows_left) cblk[lc + rr * ldc] = acc[a][b][u] * fac[u]; }
                    } else {
                        const bool keep = epi == LEPI_BIAS_GELU_D;
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            const int rr = UPP_ROWOF(u);
                            float gv, dv;
                            gelu_pair(acc[a][b][u] + bias, gv, dv);
                            if (rr < rows_left) { cblk[lc + rr * ldc] = gv; if (keep) xblk[lx + rr * ldx] = dv; }
                        }
                    }
                }
#undef UPP_ROWOF
            }
        }
    }
    UPP_STAMP(3)
}

template <int WM, int WN, int RM, int RN, int KC, int NST>
int launch_rt(const LinArgs &g0, hipStream_t st) {
    LinArgs g = g0;
    constexpr int BM = WM * RM * 32, BN = WN * RN * 32;
    const int tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    const dim3 grid((unsigned)(tiles_m * g.tiles_n)), block(WM * WN * 64);
    const bool tail = g.K % (32 * KC) != 0;
    const bool fancy = g.epi == LEPI_BIAS_GELU || g.epi == LEPI_BIAS_GELU_D || g.epi == LEPI_MUL || g.bias_shift > 0;
    if (fancy) {
#ifdef UPP_RT_NO_FANCY
        return UPP_E_RANGE;
#else
        if (tail) hipLaunchKernelGGL((linear_rt_kernel<WM, WN, RM, RN, KC, NST, true, true>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((linear_rt_kernel<WM, WN, RM, RN, KC, NST, false, true>), grid, block, 0, st, g);
#endif
    } else {
        if (tail) hipLaunchKernelGGL((linear_rt_kernel<WM, WN, RM, RN, KC, NST, true, false>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((linear_rt_kernel<WM, WN, RM, RN, KC, NST, false, false>), grid, block, 0, st, g);
    }
    return upp_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// Grouped weight gradients  dW_p (N_p,K_p) = G_p^T . X_p  of ALL trainable Linear layers of a backward pass in one launch
// (reference: AddmmBackward's second GEMM of every nn.Linear / 1x1 Conv1d under autograd -- models/Point_MAE_cp.py:369-465 in
// pre-training, models/Point_MAE_unify_segment.py:420-433 for the segmentation head).  A weight gradient is read by nothing
// inside the backward pass, so the step driver queues (G, X) pairs and issues them together when the pass is over: 65 GEMMs of
// 864 ... 2,080 rows (27 ... 108 tiles each: a fraction of the chip per launch, 10-25 us apiece) become one grid of a few
// thousand equal work units that fills every CU for several rounds.
//   unit = (problem, 128 x 128 tile of dW, run of `rows` consecutive rows of G / X)  ->  one partial tile; the caller adds the
//   partial dW's of a problem in split order (upp_batched_sum, straight into the flat gradient buffer).
// Same machine as linear_rt_kernel: 2 x 2 waves of 2 x 2 blocks, LDS-DMA into two stages of 32 rows x [128 G columns | 128 X
// columns], the barrier in front of the last fragment group.  Both operands are "k-major" by nature (the contraction runs over
// the rows), so a lane's MFMA operands are dwords of one LDS row: with ds_read_b64 lane r takes columns 2r, 2r+1 -- the wave's
// two blocks are the even and the odd columns of its 64 -- one read per operand and row pair, conflict-free without a swizzle.
// Every output is ONE ascending-row fmaf chain per split: fma(g[m+1], x[m+1], fma(g[m], x[m], acc)).
__global__ __launch_bounds__(256) void wgrad_grouped_kernel(WgGroup g) {
    constexpr int BT = 128;                         // tile edge (N and K direction)
    constexpr int PART = 32 * BT * 4;               // 32 rows x 128 columns of one operand
    constexpr int STAGE = 2 * PART;                 // [G part | X part]
    constexpr int TPW = STAGE / 1024 / 4;           // 8 DMA instructions per wave and stage (2 rows of one part each)
    __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    const int u = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    int p = 0;
    while (u >= g.unit0[p + 1]) ++p;                // (scalar) at most kMaxWgProblems steps
    const int local = u - g.unit0[p], tiles = g.tiles[p];
    const int split = local / tiles, tile = local - split * tiles;
    const int tn = tile / g.tiles_k[p], tk = tile - tn * g.tiles_k[p];
    const int n0 = tn * BT, k0 = tk * BT;
    const int M = g.M[p], N = g.N[p], K = g.K[p];
    const int ms = split * g.rows[p], me = min(M, ms + g.rows[p]);
    const int nst = (me - ms + 31) >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // DMA sources: instruction t = wave + 4 q of a stage covers rows 2 t', 2 t' + 1 of the G part (t < 16) or the X part; lane ->
    // (row 2 t' + lane / 32, 16-byte piece lane % 32).  Columns beyond N / K are clamped (their outputs are never stored), rows
    // beyond M to the last row (their products are masked).
    const float *colptr[TPW];
    int ldq[TPW], row0[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int t = wave + 4 * q;
        const bool isG = t < 16;
        const int tt = isG ? t : t - 16;
        const int col = min((isG ? n0 : k0) + 4 * (lane & 31), (isG ? N : K) - 4);
        colptr[q] = (isG ? g.G[p] : g.X[p]) + col;
        ldq[q] = isG ? g.ldg[p] : g.ldx[p];
        row0[q] = ms + 2 * tt + (lane >> 5);
    }
    auto issue1 = [&](int q, int stage, int c) {
        const int row = min(row0[q] + 32 * c, M - 1);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(colptr[q] + (long long)row * ldq[q]), (lds_ptr_t)(lds + stage * STAGE + (wave + 4 * q) * 1024), 16, 0, 0);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[a][b][t] = 0.0f;

    // fragment group s (4 row pairs: rows 8 s + 2 j + h, j = 0..3): lane (r, h) reads G[row][64 wm + 2r, +1] and X[row][64 wn + 2r, +1]
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    struct Frag { f32x2 a[4], b[4]; };
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;
    const unsigned adrG = lds0 + (h * BT + wm * 64 + 2 * r) * 4, adrX = lds0 + PART + (h * BT + wn * 64 + 2 * r) * 4;
#define UPP_WG_READ(F, SB, S)                                                                                         \
    {                                                                                                                 \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                            \
            asm volatile("ds_read_b64 %0, %1" : "=v"(F.a[j_]) : "v"(adrG + (SB) + (8 * (S) + 2 * j_) * (BT * 4)));    \
            asm volatile("ds_read_b64 %0, %1" : "=v"(F.b[j_]) : "v"(adrX + (SB) + (8 * (S) + 2 * j_) * (BT * 4)));    \
        }                                                                                                             \
    }
#define UPP_WG_LGKM(N_LEFT) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_LEFT) : "memory"); __builtin_amdgcn_sched_barrier(0);
    Frag f[2];
    // rows_left: rows of this split that exist at and after the first row of the group (rows beyond contribute fma(0, x, acc) = acc)
    auto mfma_group = [&](const Frag &fr, int slot0, int c_fill, int rows_left, auto fill_c, auto mask_c) {
        constexpr bool FILL = decltype(fill_c)::value, MASK = decltype(mask_c)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x2 av = fr.a[j];
            if constexpr (MASK) {
                if (2 * j + h >= rows_left) { av[0] = 0.0f; av[1] = 0.0f; }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], fr.b[j][b], acc[a][b], 0, 0, 0);
            if constexpr (FILL) {
                if (slot0 + j < TPW) issue1(slot0 + j, c_fill & 1, c_fill);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto stage = [&](int c, auto tail_c, auto fill_c, auto mask_c) {
        constexpr bool TAILSTEP = decltype(tail_c)::value;
        const unsigned sb = (unsigned)((c & 1) * STAGE);
        const int left = me - ms - 32 * c;                       // rows of stage c that exist (>= 32 unless it is the last one)
        UPP_WG_READ(f[0], sb, 0)
        if constexpr (TAILSTEP) mfma_group(f[1], 0, c + 1, 32, fill_c, std::false_type{});      // (stage c - 1 is never the last: full)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            UPP_WG_READ(f[(s + 1) & 1], sb, s + 1)
            UPP_WG_LGKM(8)
            mfma_group(f[s & 1], TAILSTEP ? 4 * (s + 1) : 4 * s, c + 1, left - 8 * s, fill_c, mask_c);
        }
        UPP_WG_LGKM(0)
    };
    if (nst > 0) {
#pragma unroll
        for (int q = 0; q < TPW; ++q) issue1(q, 0, 0);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (nst > 1) stage(0, std::false_type{}, std::true_type{}, std::false_type{});
        else stage(0, std::false_type{}, std::false_type{}, std::true_type{});
        int c = 1;
        for (; c + 1 < nst; ++c) {
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            stage(c, std::true_type{}, std::true_type{}, std::false_type{});
        }
        if (c < nst) {
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            stage(c, std::true_type{}, std::false_type{}, std::true_type{});
        }
        mfma_group(f[1], 0, 0, me - ms - 32 * (nst - 1) - 24, std::false_type{}, std::true_type{});
    }
#undef UPP_WG_READ
#undef UPP_WG_LGKM

    // ---- partial tile: block (a, b) register t = dW[n][k] with n = n0 + 64 wm + 2 (rowof(t) + 4 h) + a, k = k0 + 64 wn + 2 r + b.
    // Turned through the LDS (all of it: 4 waves x 16 KB, behind a barrier) so that every store instruction writes 16 bytes per lane.
    __syncthreads();
    float *tt = reinterpret_cast<float *>(lds + wave * (64 * 64 * 4));
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            f32x2 v = {acc[a][0][t], acc[a][1][t]};
            *reinterpret_cast<f32x2 *>(tt + (2 * ((t & 3) + 8 * (t >> 2) + 4 * h) + a) * 64 + 2 * r) = v;
        }
    float *out = g.P[p] + (long long)split * N * K;
    const int kcol = k0 + wn * 64 + 4 * (lane & 15);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int n = n0 + wm * 64 + 4 * i + (lane >> 4);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(tt + (i * 64 + lane) * 4);
        if (n < N && kcol < K) *reinterpret_cast<f32x4 *>(out + (long long)n * K + kcol) = v;
    }
}

}  // namespace

// Compiled register-tiled shapes, code = 0x1000000 NST + 0x100000 (4 (RM-1) + (RN-1)) + 0x10000 + 4096 WM + 256 WN + 16 + KC  (the `tile`
// argument of upp_linear_f32: WM x WN waves of RM x RN blocks, 32 KC values of k per LDS stage, NST stages; bits 4-7 = 1 = "KS")
#ifndef UPP_RT_CONFIGS
#define UPP_RT_CONFIGS(X) X(2, 2, 2, 2, 1, 2)
#endif

__attribute__((visibility("hidden"))) int upp_detail_linear_rt(const void *args, int code, hipStream_t st) {
    const LinArgs &g = *static_cast<const LinArgs *>(args);
#define UPP_RT_CASE(wm, wn, rm, rn, kc, nst) \
    case 0x1000000 * nst + 0x100000 * (4 * (rm - 1) + (rn - 1)) + 0x10000 + wm * 4096 + wn * 256 + 16 + kc: return launch_rt<wm, wn, rm, rn, kc, nst>(g, st);
    switch (code) {
        UPP_RT_CONFIGS(UPP_RT_CASE)
        default: return UPP_E_RANGE;
    }
#undef UPP_RT_CASE
}

// Rows per split for every problem of a group (host-side plan; the same rule for a group of one): about six rounds of equal
// work units on the 512 resident workgroups, never fewer than 256 rows per unit (a partial tile is 64 KB of traffic), whole
// multiples of 32 rows.
extern "C" int upp_linear_wgrad_grouped_rows(int count, const int *M, const int *N, const int *K, int *rows) {
    if (count < 1 || !M || !N || !K || !rows) return UPP_E_BADARG;
    double work = 0.0;
    for (int p = 0; p < count; ++p) {
        if (M[p] < 1 || N[p] < 1 || K[p] < 1) return UPP_E_BADARG;
        work += (double)((N[p] + 127) / 128) * ((K[p] + 127) / 128) * M[p];
    }
    long long per_unit = (long long)(work / (512.0 * 6.0));
    per_unit = (per_unit + 31) / 32 * 32;
    if (per_unit < 256) per_unit = 256;
    for (int p = 0; p < count; ++p) {
        const long long m32 = ((long long)M[p] + 31) / 32 * 32;
        long long rws = per_unit < m32 ? per_unit : m32;
        const long long splits = (M[p] + rws - 1) / rws;            // even the splits out
        rws = ((M[p] + splits - 1) / splits + 31) / 32 * 32;
        rows[p] = (int)rws;
    }
    return 0;
}

extern "C" int upp_linear_wgrad_grouped_f32(const float *const *G, const long long *ldg, const float *const *X, const long long *ldx,
                                            float *const *partials, const int *M, const int *N, const int *K, const int *rows, int count,
                                            void *stream) {
    if (count < 1 || !G || !ldg || !X || !ldx || !partials || !M || !N || !K || !rows) return UPP_E_BADARG;
    for (int p = 0; p < count; ++p) {
        if (!G[p] || !X[p] || !partials[p] || M[p] < 1 || N[p] < 1 || K[p] < 1) return UPP_E_BADARG;
        if (N[p] % 4 || K[p] % 4 || ldg[p] % 4 || ldx[p] % 4 || ldg[p] < N[p] || ldx[p] < K[p] || rows[p] < 32 || rows[p] % 32) return UPP_E_RANGE;
        if (ldg[p] > 0x7fffffffLL || ldx[p] > 0x7fffffffLL) return UPP_E_RANGE;
        if ((reinterpret_cast<uintptr_t>(G[p]) | reinterpret_cast<uintptr_t>(X[p]) | reinterpret_cast<uintptr_t>(partials[p])) & 15) return UPP_E_RANGE;
    }
    for (int p0 = 0; p0 < count; p0 += kMaxWgProblems) {
        WgGroup g{};
        const int np = count - p0 < kMaxWgProblems ? count - p0 : kMaxWgProblems;
        long long units = 0;
        for (int q = 0; q < np; ++q) {
            const int p = p0 + q;
            g.G[q] = G[p]; g.X[q] = X[p]; g.P[q] = partials[p]; g.ldg[q] = (int)ldg[p]; g.ldx[q] = (int)ldx[p];
            g.M[q] = M[p]; g.N[q] = N[p]; g.K[q] = K[p]; g.rows[q] = rows[p];
            g.tiles_k[q] = (K[p] + 127) / 128;
            g.tiles[q] = ((N[p] + 127) / 128) * g.tiles_k[q];
            g.unit0[q] = (int)units;
            units += (long long)g.tiles[q] * ((M[p] + rows[p] - 1) / rows[p]);
            if (units > 0x3fffffffLL) return UPP_E_RANGE;
        }
        for (int q = np; q <= kMaxWgProblems; ++q) g.unit0[q] = 0x7fffffff;
        g.unit0[np] = (int)units; 
        for (int q = np + 1; q <= kMaxWgProblems; ++q) g.unit0[q] = 0x7fffffff;
        hipLaunchKernelGGL(wgrad_grouped_kernel, dim3((unsigned)units), dim3(256), 0, (hipStream_t)stream, g);
        const int rc = upp_launch_status();
        if (rc) return rc;
    }
    return 0;
}

// One weight gradient (a group of one): `partials` (upp_linear_wgrad_splits(M,N,K), N, K).
extern "C" int upp_linear_wgrad_splits(int M, int N, int K) {
    int rows = 0;
    const int rc = upp_linear_wgrad_grouped_rows(1, &M, &N, &K, &rows);
    return rc ? rc : (M + rows - 1) / rows;
}

extern "C" int upp_linear_wgrad_f32(const float *G, long long ldg, const float *X, long long ldx, float *partials, int M, int N, int K,
                                    void *stream) {
    int rows = 0;
    const int rc = upp_linear_wgrad_grouped_rows(1, &M, &N, &K, &rows);
    if (rc) return rc;
    return upp_linear_wgrad_grouped_f32(&G, &ldg, &X, &ldx, &partials, &M, &N, &K, &rows, 1, stream);
}
