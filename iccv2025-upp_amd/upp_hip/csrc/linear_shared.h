// linear_shared.h -- types and device helpers shared by the exact-f32 MFMA Linear kernels (linear.hip: one 32x32 block per wave,
// the token matrices of the Transformer blocks; linear_rt.hip: register-tiled waves, the tall point-row matrices and the grouped
// weight gradients).  Included inside each file's anonymous namespace.
#pragma once

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

enum { LEPI_NONE = 0, LEPI_BIAS = 1, LEPI_BIAS_GELU = 2, LEPI_BIAS_GELU_D = 3, LEPI_MUL = 4, LEPI_BIAS_RELU = 5 };

struct LinArgs {
    const float *A; long long lda;
    const float *W; long long ldw;
    float *C; long long ldc;
    const float *bias;            // (N) or null
    float *aux; long long ldaux;  // LEPI_BIAS_GELU_D: out (M,N) GELU'(z); LEPI_MUL: in (M,N) factor
    int M, N, K;
    int tiles_n;                  // workgroup tiles along N
    int epi;
    int ktail;                    // K is not a multiple of the k-stage: the last stage reads zeros beyond K (K % 4 == 0)
    int wt;                       // != 0: the output leaves by write-through (sc1) stores (upp_store_policy(); see store4())
    int bias_shift;               // linear_rt.hip, > 0: bias is a (M >> bias_shift, N) matrix, row m takes row m >> bias_shift of it (a bias
                                  // per GROUP of 2^bias_shift >= 32 rows: the per-sample term of the segmentation head, the per-group
                                  // half of the patch embedding's 512 -> 512 layer)
#ifdef UPP_LIN_STAMPS
    unsigned long long *stamps;   // diagnostic build only (tools/micro/lin_stamps.py): [workgroup][8] clock readings
#endif
};

// Work list of the grouped weight-gradient kernels (linear_rt.hip wgrad_grouped_kernel: exact f32; wgrad_sb.hip: split bf16)
constexpr int kMaxWgProblems = 40;
struct WgGroup {
    const float *G[kMaxWgProblems], *X[kMaxWgProblems];
    float *P[kMaxWgProblems];                     // (splits, N, K) partial gradients
    int ldg[kMaxWgProblems], ldx[kMaxWgProblems];
    int M[kMaxWgProblems], N[kMaxWgProblems], K[kMaxWgProblems];
    int rows[kMaxWgProblems];                     // rows of G / X per split (a multiple of 32)
    int tiles_k[kMaxWgProblems], tiles[kMaxWgProblems];
    int unit0[kMaxWgProblems + 1];                // first work unit of each problem; unit = unit0 + split * tiles + tile
    int tr[kMaxWgProblems];                       // wgrad_sb.hip: G and X were handed over swapped, the tile is stored transposed (P stays (splits, N, K))
};

#ifdef UPP_LIN_STAMPS
unsigned long long *g_lin_stamps = nullptr;
#define UPP_STAMP(slot)                                                                                   \
    if (g.stamps && threadIdx.x == 0) {                                                                   \
        g.stamps[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime();                         \
        if ((slot) == 0 || (slot) == 3) g.stamps[(size_t)blockIdx.x * 8 + 4 + (slot) / 3] = __builtin_amdgcn_s_memrealtime(); \
    }
#else
#define UPP_STAMP(slot)
#endif

// GELU(v) = v Phi(v) and GELU'(v) = Phi(v) + v phi(v) from ONE exponential: with x = |v| / sqrt 2 and t = 1 / (1 + p x),
// erfc(x) = (a1 t + ... + a5 t^5) e^{-x^2} (Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7) and phi(v) = e^{-x^2} / sqrt(2 pi).
// ~20 VALU instructions for both against ~70 for erff + expf: the epilogue of a 16-wave workgroup is VALU time of its SIMDs.
__device__ __forceinline__ void gelu_pair(float v, float &gelu, float &dgelu) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, x, 1.0f));
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * x * x);
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float half_erfc = 0.5f * p * t * e;                     // 0.5 erfc(|v| / sqrt 2) = Phi(-|v|)
    const float cdf = v < 0.0f ? half_erfc : 1.0f - half_erfc;
    gelu = v * cdf;
    dgelu = __builtin_fmaf(v * 0.39894228040143267794f, e, cdf);
}

__device__ __attribute__((aligned(16))) const float g_lin_zeros[4] = {0.0f, 0.0f, 0.0f, 0.0f};

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }


// 16-byte store of an output granule.  wt: agent-scope write-through (`sc1`): the bytes leave the XCD's L2 as they are stored instead of
// staying dirty in it until the end-of-kernel release writes them back -- that write-back is serial time between this kernel and the
// next (MI355X_MICROARCH.md, price list "boundary": + B / 6 TB/s for B dirty bytes; "publish-large": 64 KB per workgroup 8.2 -> 3.0 us
// with write-through stores).  The next kernel reads the tensor from the memory side either way (per-XCD L2s are not coherent).
__device__ __forceinline__ void store4(float *dst, f32x4 v, int wt) {
    if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
    else *reinterpret_cast<f32x4 *>(dst) = v;
}

// the store policy of the Linear kernels' epilogues (option UPP_OPT_STORE_WT, default 1)
static inline int upp_store_policy(void) { return upp_option(UPP_OPT_STORE_WT); }

// Epilogue of four consecutive columns of one output row (a lane of the LDS-turned tile: 16-byte loads and stores): v = the
// accumulated products, bias4 = the four biases (zeros when the epilogue has none).
__device__ __forceinline__ void epilogue_store4(const LinArgs &g, int epi, f32x4 v, f32x4 bias4, int row, int col, bool ok) {
    float *dst = g.C + (long long)row * g.ldc + col;
    if (epi == LEPI_NONE || epi == LEPI_BIAS || epi == LEPI_BIAS_RELU) {
        v += bias4;
        if (epi == LEPI_BIAS_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
        }
        if (ok) store4(dst, v, g.wt);
    } else {
        float *xp = g.aux + (long long)row * g.ldaux + col;
        if (epi == LEPI_MUL) {
            if (ok) store4(dst, v * *reinterpret_cast<const f32x4 *>(xp), g.wt);
        } else {
            f32x4 gv, dv;
#pragma unroll
            for (int e = 0; e < 4; ++e) { float g1, d1; gelu_pair(v[e] + bias4[e], g1, d1); gv[e] = g1; dv[e] = d1; }
            if (ok) {
                store4(dst, gv, g.wt);
                if (epi == LEPI_BIAS_GELU_D) store4(xp, dv, g.wt);
            }
        }
    }
}
