/*
 * upp_hip.h -- C ABI of libupp_hip.so, the MI355X (gfx950) implementation of
 * the UPP / Point-MAE hot-path operators.
 *
 * This is the drop-in boundary: every entry point replaces one native
 * operator the reference binds through a torch C++/CUDA extension (file:line
 * of the replaced interface is cited per function).  The signatures use only
 * plain device pointers, sizes and a HIP stream (passed as void*): no torch
 * types, no allocation inside the library, and no global state other than the
 * four documented options of upp_set_option() below (the *_ex entry points take
 * the kernel variant of a measurement as an ARGUMENT; the library never reads
 * the environment -- `grep getenv csrc/` is empty since ABI 5).  The caller owns
 * every buffer (including scratch) and the library never synchronises; all
 * work is enqueued on `stream` (hipStream_t; NULL = the default stream), and
 * all of it is KERNEL LAUNCHES: every entry point may be captured into a HIP
 * graph (the step drivers do so), and on ROCm 7.2.0 a memset node of a graph
 * works in its first replay only -- the library imports four runtime symbols
 * (hipLaunchKernel, hipFuncSetAttribute, hipGetLastError, hipGetErrorString;
 * tests/test_abi.py) and zeroes buffers with a kernel.
 *
 * Return value: 0 on success; a negative UPP_E_* code for an argument the
 * kernels cannot serve (nothing is launched); a positive value is a
 * hipError_t from the launch.  upp_error_string() renders either.
 *
 * All tensors are dense row-major ("contiguous") f32 unless noted; indices
 * are int32 or int64 exactly as the reference operators return them.
 */
#ifndef UPP_HIP_H
#define UPP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UPP_ABI_VERSION 5   /* 5 (round 6): the measured-slower experiments left the library: upp_ln_adapter_fwd_next, upp_linear_sb_resid_f32,
                               upp_linear_sb_ln_f32, upp_linear_sb_ln_usable (and the PRO = 2 instantiations, the wave-specialised weight-gradient kernel and the
                               nine environment reads behind them); + upp_set_option / upp_get_option, and the fused operators of the secondary recipes'
                               tails: upp_bn_rows_drop_*, upp_logsoftmax_rows_*, upp_nll_mean_*, upp_noise_loss_*, upp_linear_smallk_gelu_d_f32,
                               upp_fps_gather_bwd, upp_ln_adapter_bwd_factors / upp_adapter_wgrad_* (106 prototypes; 4 had 93).
                               4: + upp_argsort_rows, upp_group_max_fwd / _bwd.  3: split-bf16 Linear (upp_linear_sb_*); the k-parts protocol, the attention
                               `variant` entry points (upp_attn_*_ex) and the VALU / 32x32x2 attention kernels behind them are gone.
                               2: grouped weight gradients, register-tiled Linear codes; the round-1 *_set_* toggles are gone */

#define UPP_E_BADARG   (-1)  /* null pointer / non-positive size                   */
#define UPP_E_RANGE    (-2)  /* size outside what the kernels support (see below)  */
#define UPP_E_KGTN     (-3)  /* kNN: k > number of reference points                */

int         upp_abi_version(void);
const char *upp_error_string(int code);

/* ---- options: the library's only process-wide state -------------------------
 * Every knob of the library, each with the product's default (an A/B-measured choice; the other values exist for measurements and
 * tests).  Relaxed atomics: set them before launching work from several threads.  No entry point reads the environment; the Python
 * host (upp_hip/_abi.py) forwards the variables of the same names ONCE at load time so that the A/B scripts under tools/ keep working.
 *   UPP_OPT_SB_TUNED          1 | 0   upp_linear_sb_tile consults the measured tile table (csrc/linear_sb_tuned.h) before its cost
 *                                     model | the model alone                                          [env UPP_SB_TUNED]
 *   UPP_OPT_SB_XCD2D          2 | 0 | 4   column groups of the 2-D XCD tile map of one-round split-bf16 Linear launches with wide
 *                                     outputs (0: row-major XCD ranges); traffic only, same results    [env UPP_SB_XCD2D]
 *   UPP_OPT_STORE_WT          1 | 0   epilogue stores of the Linear kernels are agent-scope write-through (`sc1`) | plain
 *                                     (same results; the end-of-kernel write-back is what differs)     [env UPP_STORE_WT]
 *   UPP_OPT_EMBED_SPLIT_BF16  1 | 0   upp_patch_embed_fwd runs its two large products on the split-bf16 kernel | the exact-f32 chain
 *                                     of rounds 1-3 (read per call)                                    [env UPP_EMBED_SPLIT_BF16]
 * upp_set_option: 0, UPP_E_BADARG (unknown key) or UPP_E_RANGE (value not listed above).  upp_get_option: the value (>= 0) or UPP_E_BADARG. */
enum { UPP_OPT_SB_TUNED = 0, UPP_OPT_SB_XCD2D = 1, UPP_OPT_STORE_WT = 2, UPP_OPT_EMBED_SPLIT_BF16 = 3, UPP_OPT_COUNT = 4 };
int upp_set_option(int key, int value);
int upp_get_option(int key);

/* ---- furthest point sampling ------------------------------------------------
 * Replaces pointnet2_ops._ext.furthest_point_sampling(xyz, npoint)
 * (pointnet2_ops 3.0.0 sampling_gpu.cu furthest_point_sampling_kernel; called
 * from reference utils/misc.py:18, tools/runner_module.py:151,450).
 *   xyz     (B,N,3) f32
 *   idx     (B,M)   int32 out: idx[b,0] = 0, then the farthest-point sequence
 *   centers (B,M,3) f32 out or NULL: xyz[b, idx[b,j]] (fuses the
 *           gather_operation of utils/misc.py:19)
 * Semantics reproduced: running min-distance initialised to 1e10, points with
 * |p|^2 <= 1e-3 never sampled, ties broken exactly as the T-thread block
 * reduction of the CUDA kernel does (T = min(512, 2^floor(log2 N))).
 * Limits: 1 <= M, 1 <= N <= 32768. */
int upp_fps(const float *xyz, int32_t *idx, float *centers,
            int B, int N, int M, void *stream);
/* the same with the launch form forced: identical results.
 *   bits 0-7 of `waves`: wavefronts per cloud (1, 2, 4, 8; 0 = the library's choice = upp_fps) -- for measurements and tests;
 *   bits 8-11: clouds per workgroup (1, 2, 4; 0 = the default, one), bit 12: the launch reserves the whole LDS of the CUs it runs on.
 *       Several clouds per workgroup (4-wave launches of at most 8 points per lane, B divisible by the count) put a 32-cloud call on 16 or
 *       8 CUs instead of 32; with bit 12 no workgroup of another stream shares those CUs.  For callers that run FPS BESIDE another stream
 *       (the pipelined training step): the call itself is 11-39 % slower, the other stream's one-round GEMMs are no longer slowed by the
 *       workgroups they shared CUs with (tools/micro/fps_cpw_ab.sh).  Any other bit: UPP_E_BADARG. */
int upp_fps_ex(const float *xyz, int32_t *idx, float *centers,
               int B, int N, int M, int waves, void *stream);

/* ---- gather_operation ---------------------------------------------------------
 * Replaces pointnet2_ops._ext.gather_points / gather_points_grad
 * (sampling_gpu.cu gather_points_kernel / gather_points_grad_kernel; called
 * from reference utils/misc.py:19).
 *   feat (B,C,N) f32, idx (B,M) int32 -> out (B,C,M) f32
 * Backward: grad_feat (B,C,N) must be zero-filled by the caller; grad_out
 * (B,C,M) is scatter-added into it (f32 atomics, order unspecified -- as in
 * the CUDA kernel). */
int upp_gather_fwd(const float *feat, const int32_t *idx, float *out,
                   int B, int C, int N, int M, void *stream);
int upp_gather_bwd(const float *grad_out, const int32_t *idx, float *grad_feat,
                   int B, int C, int N, int M, void *stream);

/* ---- k nearest neighbours ------------------------------------------------------
 * Replaces knn_cuda.KNN(k, transpose_mode=True).forward(ref, query)
 * (KNN_CUDA 0.2 knn.cu cuComputeDistanceGlobal + cuInsertionSort +
 * cuParallelSqrt, looped over the batch in Python; constructed at reference
 * models/Point_MAE_unify.py:56, called :69).  One launch for the whole batch.
 *   ref   (B,N,3) f32, query (B,Q,3) f32
 *   dist  (B,Q,K) f32 out or NULL: Euclidean distances, ascending
 *   idx   (B,Q,K) int64 out: 0-based, ordered by (distance, index)
 *   neigh (B,Q,K,3) f32 out or NULL: ref[b, idx[b,q,j]] - query[b,q]  (fuses
 *         the gather + centre subtraction of Group.forward,
 *         models/Point_MAE_unify.py:73-88)
 * Limits: 1 <= K <= min(N, 64). */
int upp_knn(const float *ref, const float *query, float *dist, int64_t *idx, float *neigh,
            int B, int N, int Q, int K, void *stream);
/* the same with the k-th-smallest prefilter of the insertion sort switched off
 * (prefilter = 0; 1 = upp_knn): identical results, for measurements and tests. */
int upp_knn_ex(const float *ref, const float *query, float *dist, int64_t *idx, float *neigh,
               int B, int N, int Q, int K, int prefilter, void *stream);

/* ---- grouping gather (stand-alone) and its backward ----------------------------
 * Replaces the torch indexing of Group.forward (reference
 * models/Point_MAE_unify.py:73-88): out[b,g,k] = xyz[b, idx[b,g,k]] - center[b,g].
 *   xyz (B,N,3), center (B,G,3), idx (B,G,K) int64 in [0,N), out (B,G,K,3)
 * Backward: grad_xyz (B,N,3) must be zero-filled by the caller (scatter-add,
 * f32 atomics); grad_center (B,G,3) is overwritten with -sum_k grad_out.
 * Either gradient pointer may be NULL. */
int upp_group_fwd(const float *xyz, const float *center, const int64_t *idx, float *out,
                  int B, int N, int G, int K, void *stream);
int upp_group_bwd(const float *grad_out, const int64_t *idx, float *grad_xyz, float *grad_center,
                  int B, int N, int G, int K, void *stream);
/* upp_fps_gather_bwd (round 6): the backward of the centre gather that upp_fps fuses (reference utils/misc.py:19 gather_operation under
 * autograd): g_xyz (B,N,3) = zeros with g_centers[b][j] added at row idx[b][j] (int32, as upp_fps writes it) -- zero-fill, index
 * conversion and scatter in ONE launch, one workgroup per cloud. */
int upp_fps_gather_bwd(const float *g_centers, const int32_t *idx, float *g_xyz, int B, int N, int M, void *stream);

/* ---- Chamfer distance ----------------------------------------------------------
 * Replaces chamfer.forward / chamfer.backward (reference
 * extensions/chamfer_dist/chamfer_cuda.cpp:36-39, kernels chamfer.cu:15-145 and
 * :173-201).
 *   xyz1 (B,n,3), xyz2 (B,m,3)
 *   dist1 (B,n) f32, idx1 (B,n) int32: squared distance to, and index of, the
 *         nearest xyz2 point (lowest index among equal minima); dist2/idx2 the
 *         other direction.
 * Backward: g1 (B,n,3) and g2 (B,m,3) are OVERWRITTEN (no pre-zeroing: the reference's wrapper allocates zeros and its kernel adds,
 * chamfer.cu:194-199,215-222); sums by f32 atomics in an unspecified order as there -- in the LDS, one workgroup per cloud pair, when
 * both gradient arrays fit (n + m <= 5,461), in global memory otherwise. */
int upp_chamfer_fwd(const float *xyz1, const float *xyz2,
                    float *dist1, float *dist2, int32_t *idx1, int32_t *idx2,
                    int B, int n, int m, void *stream);
int upp_chamfer_bwd(const float *xyz1, const float *xyz2, const int32_t *idx1, const int32_t *idx2,
                    const float *grad_dist1, const float *grad_dist2, float *g1, float *g2,
                    int B, int n, int m, void *stream);
/* upp_chamfer_loss: the reductions of ChamferDistanceL1 / ChamferDistanceL2 (reference extensions/chamfer_dist/__init__.py:44-84) in two
 * launches: loss[0] = (mean sqrt dist1 + mean sqrt dist2) / 2 (l1 != 0) or mean dist1 + mean dist2 (l1 == 0), means over all B n resp.
 * B m entries; fac1 (B,n) / fac2 (B,m) receive d loss / d dist element-wise, so that the module's backward is upp_chamfer_bwd on
 * (upstream scalar x fac).  work: upp_chamfer_loss_work_floats() floats of scratch.  Sums in a fixed order. */
long long upp_chamfer_loss_work_floats(void);
int upp_chamfer_loss(const float *dist1, const float *dist2, int B, int n, int m, int l1, float *loss, float *fac1, float *fac2,
                     float *work, void *stream);

/* ---- approximate earth mover's distance ----------------------------------------
 * Replaces emd_cuda.approxmatch_forward / matchcost_forward / matchcost_backward
 * (reference extensions/emd/cuda/emd.cpp:23-27, kernels emd_kernel.cu:24-157,
 * :199-242, :285-354).
 *   xyz1 (B,n,3), xyz2 (B,m,3)
 *   match (B,m,n) f32 out (overwritten)
 *   work  f32 scratch of upp_emd_work_floats(B,n,m) elements
 *   cost  (B) f32 out: sum_{k,l} |xyz1[k]-xyz2[l]|^2 * match[l,k]
 *   grad1 (B,n,3), grad2 (B,m,3) out (overwritten); match is a constant.
 * Limits: n, m >= 1. */
long long upp_emd_work_floats(int B, int n, int m);
int upp_emd_approxmatch(const float *xyz1, const float *xyz2, float *match, float *work,
                        int B, int n, int m, void *stream);
int upp_emd_matchcost(const float *xyz1, const float *xyz2, const float *match, float *cost,
                      int B, int n, int m, void *stream);
int upp_emd_matchcost_bwd(const float *grad_cost, const float *xyz1, const float *xyz2, const float *match,
                          float *grad1, float *grad2, int B, int n, int m, void *stream);

/* ---- patch embedding (mini-PointNet) forward --------------------------------------
 * Replaces Encoder.forward (reference models/Point_MAE_unify.py:191-222): the chain
 * Conv1d(3,128)-BN-ReLU-Conv1d(128,256) -> group max -> concat -> Conv1d(512,512)-BN-ReLU-
 * Conv1d(512,C) -> group max, as FP32-MFMA GEMMs with BatchNorm statistics, normalisation,
 * ReLU and both max-pools fused into GEMM prologues / epilogues.
 *   pts  (R,3) f32: the centred neighbourhoods, R = B*G*n rows, n = points per group (16 or 32)
 *   w1 (128,3) b1 (128) | w2 (256,128) b2 (256) | w3 (512,512) b3 (512) | w4 (C,512) b4 (C):
 *        the Conv1d weights with their trailing kernel dimension of 1 dropped
 *   bn1_* (128), bn3_* (512): BatchNorm1d weight / bias / running_mean / running_var
 *   training != 0: batch statistics (biased variance) normalise, running stats are updated
 *        in place with `momentum` and the unbiased variance, as nn.BatchNorm1d does;
 *        training == 0: running statistics normalise, nothing is written to them.
 *   work: upp_patch_embed_work_floats(R, n) floats of scratch (16-byte aligned)
 *   out  (R/n, C) f32
 * Forward only (the encoder is frozen in every UPP recipe).
 * Arithmetic (round 4): the 256 -> 512 and 512 -> C products run on the split-bf16 kernel of upp_linear_sb_f32 (both f32 operands as
 * three bf16 terms, six products, f32 accumulation: the error of an f32 GEMM) for C <= 1024, their weights split into `work` by every
 * call; option UPP_OPT_EMBED_SPLIT_BF16 = 0 (read per call) keeps the exact-f32 chain of rounds 1-3.  BatchNorm batch
 * statistics are sums of per-workgroup partial sums in a fixed order (no atomics): the call is deterministic.
 * Limits: n in {16, 32}, R % n == 0, C % 4 == 0. */
long long upp_patch_embed_work_floats(int R, int n);
int upp_patch_embed_fwd(const float *pts, int R, int n,
                        const float *w1, const float *b1, const float *bn1_gamma, const float *bn1_beta,
                        float *bn1_rmean, float *bn1_rvar,
                        const float *w2, const float *b2,
                        const float *w3, const float *b3, const float *bn3_gamma, const float *bn3_beta,
                        float *bn3_rmean, float *bn3_rvar,
                        const float *w4, const float *b4, int C,
                        float momentum, float eps, int training,
                        float *work, float *out, void *stream);

/* ---- Transformer block glue: row gather (+pos) (+scaled residual) -> LayerNorm -----------
 * Fuses the element-wise steps of Block.forward around its GEMMs (reference
 * models/Point_MAE_pretask_dev.py:245-321): `x + pos` (TransformerEncoder :348), prompt insertion
 * (:247-264) or removal (:305-310), `x + drop_path(branch)` (:266,273; timm DropPath: per-sample factor
 * floor(keep + u) / keep), and the following LayerNorm (norm1 / norm2 / Adapter.layer_norm :97).
 *   s = src(t): mode 0 identity | 1 insert P prompts after the cls row | 2 insert P prompts in front |
 *                3 strip the P prompts after the cls row | 4 strip P leading prompts
 *   out row (b,t) = x[b, s] (+ add[b, s])                  if s is a token row
 *                 = prompts[p]                             if s selects prompt p (modes 1, 2)
 *                 (+ floor(keep + u[b]) / keep * (y[b, s] + ybias))  if y (row-aligned with x; u NULL: factor 1; ybias (D)
 *                    NULL or the bias of the Linear whose bias-free GEMM produced y)
 *   xo (B,Lout,D) = those rows (optional);  h = LayerNorm(rows) * gamma + beta, mean / rstd (B,Lout) saved
 *   (gamma NULL: no LayerNorm, only xo).
 * Backward: d = g_xo + LayerNormBackward(g_h); written to g_x[b, s] (every row of g_x / g_y is written: the prompt
 * rows a strip mode drops get zeros), g_prompt (B,P,D) (caller sums over B), g_y[b, s] = factor * d.
 * ln_part (optional, upp_rowln_part_floats floats): per-workgroup partials (workgroups, 2, D) of d_gamma = sum g_h * xhat and
 * d_beta = sum g_h, produced by the same pass; the caller sums them over the workgroups (upp_batched_sum).
 * upp_ln_param_grad: the same partial sums as a stand-alone pass, part (2, chunks, D).
 * Limits: D <= 512. */
int upp_rowln_fwd(const float *x, const float *add, const float *prompts, int mode, int P, const float *y, const float *ybias,
                  const float *u, float keep, const float *gamma, const float *beta, float eps,
                  float *xo, float *h, float *mean, float *rstd, int B, int Lin, int Lout, int D, void *stream);
long long upp_rowln_part_floats(int B, int Lin, int Lout, int D, int mode);
int upp_rowln_bwd(const float *g_xo, const float *g_h, const float *xo, const float *mean, const float *rstd,
                  const float *gamma, int mode, const float *u, float keep,
                  float *g_x, float *g_prompt, float *g_y, float *ln_part,
                  int B, int Lin, int Lout, int D, int P, void *stream);
/* upp_bias_gelu_fwd / bwd: h = GELU(z + bias) (erf form) for the bias-free GEMM output z (rows, C) of Mlp.fc1, and
 * g_z = g_h * GELU'(z + bias).  C % 4 == 0. */
int upp_bias_gelu_fwd(const float *z, const float *bias, float *h, long long rows, int C, void *stream);
int upp_bias_gelu_bwd(const float *g_h, const float *z, const float *bias, float *g_z, long long rows, int C, void *stream);
/* upp_bias_gelu_fwd_d: the forward that also stores d = GELU'(z + bias) (rows, C): the backward of a training step is then
 * g_z = g_h * d, one multiply per element. */
int upp_bias_gelu_fwd_d(const float *z, const float *bias, float *h, float *d, long long rows, int C, void *stream);
int upp_ln_param_grad(const float *g_h, const float *xo, const float *mean, const float *rstd, float *part,
                      int rows, int D, int chunks, void *stream);

/* ---- multi-head attention core ------------------------------------------------------------
 * Replaces the unfused q@k^T -> *scale -> softmax -> @v of Attention.forward (reference
 * models/Point_MAE_pretask_dev.py:186-193) and its autograd.
 *   qkv (B,L,3,H,64): the qkv Linear output as the reference reshapes it (:186)
 *   ctx (B,L,H*64): softmax(q k^T * scale) v, already in the layout of (:193) `.transpose(1,2).reshape(B,N,C)`
 *   lse (B,H,L): log-sum-exp of the scaled scores (saved for backward)
 *   d_qkv (B,L,3,H,64) from d_ctx (B,L,H*64)
 * Limits: head_dim == 64; L <= 192 forward, L <= 160 backward (FP32 MFMA kernels for L <= 96 and L <= 160). */
int upp_attn_fwd(const float *qkv, float *ctx, float *lse, int B, int L, int H, int head_dim, float scale, void *stream);
int upp_attn_bwd(const float *qkv, const float *ctx, const float *d_ctx, const float *lse, float *d_qkv,
                 int B, int L, int H, int head_dim, float scale, void *stream);
/* (L <= 96: the register-resident 16x16x4 MFMA kernels of attn_flash16.hip; L <= 160: attn_long.hip; UPP_E_RANGE beyond) */
/* ---- prompt propagation (Block.forward, reference models/Point_MAE_pretask_dev.py:275-303) ----
 * X (rows, D): the block's token matrix viewed as rows = B*L' rows [cls | prompts | T centre tokens] per sample.
 * Index arguments are ABSOLUTE row numbers of X (the host converts the reference's flat / per-sample index
 * conventions, including its stride-64-into-stride-74 addressing when gather_idx is False).
 *   upp_prop_pool_fwd : pooled[g] = max_k v + mean_k v over v_k = X[i1[g*8+k]] * (1 + s_g), s_g = floor(keep+u[g])/keep
 *                       (u NULL: s = 1, i.e. eval-mode `x + drop_path(x)` = 2x); amax (groups, D) uint8 = arg max k.
 *                       (`pooling` is undefined in the reference: SURVEY D.3 assumption; BatchNorm is applied by the caller.)
 *   upp_prop_pool_bwd : g_X (rows, D), fully written (zeros where no group references a row).
 *   upp_prop_interp_fwd: out[b,t] = X[b,t] (+ 0.3 * sum_k w8[b,i,k] * (lc[b,j] + 0.3 * X[i2[b*G2+j]]), j = idx8[b,i,k])
 *                       for the last T rows (i = t - (L'-T)); `propagate(..., de_neighbors=8)` of models/Point_MAE_unify.py:22-48
 *                       with idx8 / w8 (B,T,8) precomputed once per forward from the centres.
 *   upp_prop_interp_bwd: g_c2 (B*G2, D) = gradient w.r.t. lc; g_X (rows, D) = g_out + 0.3 * g_c2 scattered to rows i2.
 * Limits: D <= 512; 8 neighbours per group / per interpolation. */
int upp_prop_pool_fwd(const float *X, const int32_t *i1, const float *u, float keep, float *pooled, uint8_t *amax,
                      int groups, int D, void *stream);
int upp_prop_pool_bwd(const float *g_pooled, const uint8_t *amax, const int32_t *i1, const float *u, float keep,
                      float *g_X, int rows, int groups, int D, void *stream);
int upp_prop_interp_fwd(const float *X, const float *lc, const int32_t *i2, const int32_t *idx8, const float *w8,
                        float *out, int B, int Lp, int T, int G2, int D, void *stream);
int upp_prop_interp_bwd(const float *g_out, const int32_t *i2, const int32_t *idx8, const float *w8, float *g_c2,
                        float *g_X, int B, int Lp, int T, int G2, int D, void *stream);

/* ---- fused propagation step: pool -> BatchNorm1d(bnorm) -> interpolate, CSR-driven backward ----
 * Same mathematics as upp_prop_pool_fwd -> BatchNorm1d over the B*G2 pooled rows (Block.bnorm, reference
 * models/Point_MAE_pretask_dev.py:275-303 + SURVEY D.3) -> upp_prop_interp_fwd, as 3 launches forward and 3 backward.
 *   upp_csr_build : stable inverse of an index list.  key(q) = (q / seg_len) * seg_rows + keys[q] (seg_rows = 0: absolute
 *                   keys); start (rows+1), perm (n): perm[start[r] .. start[r+1]) = ascending positions q with key(q) == r.
 *                   Keys outside [0, rows) are skipped.  rows <= 15360.  The three lists of a forward are inverted once:
 *                   csr1 = (i1; n = groups*8; rows = B*L'), csr2 = (i2; n = groups; rows = B*L'),
 *                   csr8 = (idx8; n = B*T*8, seg_len = T*8, seg_rows = G2; rows = groups).
 *   upp_prop_fwd  : pooled (groups, D) pre-BatchNorm rows, amax (groups, D) uint8, mean / rstd (D) batch statistics (training)
 *                   or running statistics (training == 0), out (B*L', D).  training != 0 updates running_mean / running_var
 *                   (momentum; unbiased variance) when they are non-NULL.  part: upp_prop_part_floats(groups, D) floats.
 *   upp_prop_bwd  : g_X (B*L', D) complete input gradient (identity + i2 scatter + BatchNorm/pool backward), g_gamma / g_beta (D)
 *                   BatchNorm parameter gradients; g_c2 (groups, D) and part are scratch.
 * Column partials are combined in a fixed order (Chan's update in f64 for the variance): deterministic. */
int upp_csr_build(const int32_t *keys, int n, int seg_len, int seg_rows, int rows, int32_t *start, int32_t *perm, void *stream);
/* upp_prop_index: the four index lists of upp_prop_fwd for one forward in one launch.  c1 (B,T,3) level-1 centres, c2 (B,G2,3)
 * level-2 centres; i1 (B*G2*8) / i2 (B*G2) int64 as Group hands them on (reference models/Point_MAE_unify.py:73-79: flat with
 * batch offsets when gather_idx == 0, per-sample otherwise); L' rows per sample, `off` leading rows (cls) excluded.
 * -> i1a, i2a absolute rows (the stride-T-into-stride-(L'-off) re-interpretation of Point_MAE_pretask_dev.py:291-292 when
 * gather_idx == 0); idx8 / w8 (B,T,8): 8 nearest level-2 centres by the reference's square_distance form, ascending
 * (distance, index), weights (1/(d+eps)) / sum (propagate, models/Point_MAE_unify.py:22-48).  G2 <= 64. */
int upp_prop_index(const float *c1, const float *c2, const int64_t *i1, const int64_t *i2, int gather_idx, int B, int T, int G2,
                   int Lp, int off, float eps, int32_t *i1a, int32_t *i2a, int32_t *idx8, float *w8, void *stream);
long long upp_prop_part_floats(int groups, int D);
int upp_prop_fwd(const float *X, const int32_t *i1, const float *u, float keep, const int32_t *i2, const int32_t *idx8,
                 const float *w8, const float *gamma, const float *beta, float *running_mean, float *running_var,
                 float momentum, float eps, int training, float *pooled, uint8_t *amax, float *part, float *mean,
                 float *rstd, float *out, int B, int Lp, int T, int G2, int D, void *stream);
int upp_prop_bwd(const float *g_out, const float *pooled, const uint8_t *amax, const float *mean, const float *rstd,
                 const float *gamma, const float *u, float keep, const float *w8, const int32_t *start1,
                 const int32_t *perm1, const int32_t *start2, const int32_t *perm2, const int32_t *start8,
                 const int32_t *perm8, int training, float *g_c2, float *part, float *g_gamma, float *g_beta, float *g_X,
                 int B, int Lp, int T, int G2, int D, void *stream);
/* Stage 2 of the recipe (run.sh:16, models/Point_MAE_unify.py:22-48 under autograd): the centres carry a gradient, so the
 * interpolation weights w8 do too.
 *   upp_prop_w8_grad    : g_w8 (B,T,8) = 0.3 * <g_out[token], lc[j] + 0.3 * X[i2[j]]>, j = idx8[token,k]; lc = BatchNorm(pooled) when
 *                         mean / rstd / gamma / beta are given (the tensors upp_prop_fwd saved), `pooled` itself when mean == NULL.
 *   upp_prop_weights_bwd: g_c1 (B,T,3), g_c2 (B,G2,3) from g_w8 through w = (1/(d+eps)) / sum, d = the reference's square_distance
 *                         form (models/modules.py:13-32); idx8 is a constant of the forward.  Fixed summation order, no atomics. */
int upp_prop_w8_grad(const float *g_out, const float *X, const float *pooled, const float *mean, const float *rstd,
                     const float *gamma, const float *beta, const int32_t *i2, const int32_t *idx8, float *g_w8, int B, int Lp,
                     int T, int G2, int D, void *stream);
int upp_prop_weights_bwd(const float *c1, const float *c2, const int32_t *idx8, const float *g_w8, float eps, float *g_c1,
                         float *g_c2, int B, int T, int G2, void *stream);

/* ---- row operators of the prompter branches and the per-point heads ---------------------------------
 * Replace the element-wise / reduction chains of the reference's rectify prompter
 * (models/Point_MAE_pretask_dev.py:22-52 PositionalEmbedding, :386-473 set abstraction / feature propagation):
 *   upp_bn_rows_fwd : y (R,C) = BatchNorm over the R rows of the channels-last matrix x (== BatchNorm1d/2d on the
 *                     reference's (B,C,N[,k]) layout), optional ReLU.  training != 0: batch statistics, running
 *                     statistics updated (momentum, unbiased variance) when non-NULL; training == 0: running
 *                     statistics.  gamma / beta may be NULL (1 / 0).  mean, rstd (C) are outputs; part:
 *                     upp_bn_rows_part_floats(R, C) floats of scratch.  Deterministic (fixed combination order).
 *   upp_bn_rows_bwd : backward of the training-mode form (batch statistics), for the trainable BatchNorm1d layers of the
 *                     per-point heads (models/Point_MAE_unify_segment.py seg_head / propagation_0, the patch embedding in
 *                     pre-training): x, g (R,C), mean / rstd from the forward; with xh = (x-mean)*rstd and
 *                     gm = g * [xh*gamma+beta > 0] (relu != 0; gm = g otherwise):  g_beta = sum gm, g_gamma = sum gm*xh,
 *                     g_x = gamma*rstd*(gm - (g_beta + xh*g_gamma)/R).  g_x may be NULL (parameter gradients only).
 *   upp_bn_rows_drop_fwd / _bwd (round 6): the training-mode pair with nn.Dropout(p) applied to the output in the same passes (the
 *                     segmentation head's `Conv1d, BatchNorm1d, ReLU, Dropout(0.5)`, reference models/Point_MAE_unify_segment.py:424-427):
 *                     y = dropout(relu?(bn(x))), kept values scaled by 1 / (1 - p).  The mask of element i is a counter-based hash of
 *                     (*seed + seed_add, salt, i) recomputed by the backward (nothing is stored; no uniform tensor is read): `seed` is a
 *                     DEVICE int64 the caller changes once per forward -- the layer's own num_batches_tracked -- so steps differ; seed_add
 *                     lets forward and backward of ONE step agree when the counter moves between them (a host that bumps its counters at
 *                     the end of the forward hands the forward 1 and the backward 0).  0 <= p < 1 (p = 0: the plain pair).
 *                     The stream of masks is this library's (not torch's Philox).
 *   upp_sqdist_topk : dist / idx (B,N,k) = the k nearest of the S points src[b] for every query q[b,n], by the reference's
 *                     square_distance form d = |a|^2 + |b|^2 - 2 a.b (models/modules.py:13-32), ascending (d, index):
 *                     `square_distance(q, src).sort(dim=-1)` cut to k columns.  S <= 256, k <= min(S, 64).
 *   upp_interp_fwd  : out[b*N+n][col0 .. col0+C) = sum_{j<k} w_j feat[b][idx[b,n,j]],  w_j = (1/(d_j+eps)) / sum_j(1/(d_j+eps)),
 *                     where (dist, idx) (B*N rows, row stride ld_tab, idx int64) is a neighbour table sorted by distance
 *                     (torch.sort of square_distance, as the reference computes it); feat (B,S,C); k <= 16.
 *   upp_interp_affine_fwd : the same interpolation into a dense (B*N, C) matrix plus a rank-3 term,
 *                     out[row][c] = sum_j w_j feat[..][c] + x3[row][0] wt[0][c] + x3[row][1] wt[1][c] + x3[row][2] wt[2][c],
 *                     x3 (B*N,3), wt (3,C): the first 1x1 convolution of PointNetFeaturePropagation (reference
 *                     models/Point_MAE_pretask_dev.py:463-470) commuted with the interpolation -- feat then holds
 *                     W_x . points2 + b per source row and wt the xyz columns of the weight.  k <= 4, C >= 256, C % 4 == 0.
 *   upp_interp_bwd  : g_feat (B,S,C) = gradient of the above w.r.t. feat for g_out (B*N rows, row stride ld_g, columns
 *                     [col0, col0+C)); the neighbour table is a constant.  Deterministic (no atomics).  N <= 4096.
 *   upp_posenc_fwd  : out[row][col0 ..) = (x, sin(f_0 x), cos(f_0 x), ..., sin(f_{F-1} x), cos(f_{F-1} x)), x (rows,3);
 *                     freqs is a HOST array of F <= 8 frequencies.
 * interp / posenc write a column window of a row-major buffer with row stride ld_out (the reference's torch.cat). */
long long upp_bn_rows_part_floats(int R, int C);
int upp_bn_rows_fwd(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                    float momentum, float eps, int training, int relu, float *part, float *mean, float *rstd, float *y,
                    int R, int C, void *stream);
int upp_bn_rows_bwd(const float *x, const float *g, const float *mean, const float *rstd, const float *gamma, const float *beta,
                    int relu, float *part, float *g_gamma, float *g_beta, float *g_x, int R, int C, void *stream);
int upp_bn_rows_drop_fwd(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                         float momentum, float eps, int relu, float p, const long long *seed, long long seed_add, unsigned salt, float *part,
                         float *mean, float *rstd, float *y, int R, int C, void *stream);
int upp_bn_rows_drop_bwd(const float *x, const float *g, const float *mean, const float *rstd, const float *gamma, const float *beta,
                         int relu, float p, const long long *seed, long long seed_add, unsigned salt, float *part, float *g_gamma, float *g_beta,
                         float *g_x, int R, int C, void *stream);
int upp_sqdist_topk(const float *q, const float *src, float *dist, int64_t *idx, int B, int N, int S, int k, void *stream);
int upp_interp_fwd(const float *dist, const int64_t *idx, int ld_tab, const float *feat, float *out, int ld_out, int col0,
                   int B, int N, int S, int C, int k, float eps, void *stream);
int upp_interp_affine_fwd(const float *dist, const int64_t *idx, int ld_tab, const float *feat, const float *x3, const float *wt,
                          float *out, int B, int N, int S, int C, int k, float eps, void *stream);
int upp_interp_bwd(const float *dist, const int64_t *idx, int ld_tab, const float *g_out, int ld_g, int col0, float *g_feat,
                   int B, int N, int S, int C, int k, float eps, void *stream);
/* upp_interp_geo_bwd: gradient of upp_interp_fwd w.r.t. the GEOMETRY -- the queries xyz1 (B,N,3) and the sources xyz2 (B,S,3) behind the
 * squared distances of the neighbour table (d_j = |x1 - x2_j|^2, reference models/modules.py:13-32; the autograd chain of
 * models/Point_MAE_unify.py:22-48 / models/Point_MAE_pretask_dev.py:446-470 when the prompters train: stage 2, the pre-task recipe).
 * g_xyz1 (B,N,3) and / or g_xyz2 (B,S,3) are written (either may be NULL); contrib: (B*N, k, 3) floats of scratch, needed with g_xyz2
 * (the source-side terms are summed per source point in (row, j) order: deterministic).  k <= 16, k <= S. */
int upp_interp_geo_bwd(const float *dist, const int64_t *idx, int ld_tab, const float *feat, const float *g_out, int ld_g, int col0,
                       const float *xyz1, const float *xyz2, int B, int N, int S, int C, int k, float eps, float *g_xyz1,
                       float *g_xyz2, float *contrib, void *stream);
int upp_posenc_fwd(const float *x, const float *freqs, int F, float *out, int ld_out, int col0, long long rows, void *stream);

/* ---- classification tail (reference models/Point_MAE_unify.py:650-655 and get_loss_acc :499-503) ----------------
 *   upp_cls_pool_fwd : h = LayerNorm(x) (B,L,D) with gamma / beta / eps (self.norm); feat (B,2D) = [h[:,0] | max over rows 1..L-1 of h]
 *                      (first maximum); amax (B,D) int32 = the arg-max row; mean / rstd (B*L) saved.  h itself is not stored.
 *   upp_cls_pool_bwd : g_x (B,L,D) from g_feat (B,2D) (dense: every row is written).  gamma / beta gradients are not
 *                      produced (self.norm is frozen under PEFT; the caller falls back to torch ops otherwise).
 *   upp_ce_acc       : out2[0] = mean cross-entropy of logits (B,C) against labels (B) int64, out2[1] = top-1 accuracy * 100,
 *                      dlogits (B,C) = (softmax - onehot) / B.  C <= 512.  Deterministic. */
int upp_cls_pool_fwd(const float *x, const float *gamma, const float *beta, float eps, float *feat, int32_t *amax, float *mean,
                     float *rstd, int B, int L, int D, void *stream);
int upp_cls_pool_bwd(const float *g_feat, const float *x, const float *mean, const float *rstd, const float *gamma,
                     const int32_t *amax, float *g_x, int B, int L, int D, void *stream);
int upp_ce_acc(const float *logits, const int64_t *labels, float *out2, float *dlogits, int B, int C, void *stream);
/* ---- per-point log-softmax + NLL of the segmentation head (round 6) ---------------------------------------------------
 * Replaces `F.log_softmax(x, dim=-1)` (reference models/Point_MAE_unify_segment.py:433) and `F.nll_loss(pred, target)`
 * (:20-25 get_loss) over R = B * N point rows of C <= 64 classes, and the torch glue around them (gather, mean, neg, div, zero-fill +
 * scatter_add, zero-fill + copy of a column slice's backward):
 *   upp_logsoftmax_rows_fwd : logp (R,C) = log_softmax over c of y[r * ld_y + c] + bias[c] (bias NULL: none).  ld_y >= C: the rows may
 *                             be wider than C (the 50-class layer's GEMM writes a 52-column matrix).
 *   upp_logsoftmax_rows_bwd : g_y[r * ld_gy + c] = g_logp[r][c] - exp(logp[r][c]) * sum_c g_logp[r][c] for c < C and 0 for
 *                             C <= c < Cpad <= ld_gy: the gradient in the producing GEMM's padded layout, every element written once.
 *   upp_nll_mean_fwd        : out[0] = -(1 / R) sum_r logp[r][target[r]] (targets outside [0, C) contribute 0); `part`:
 *                             upp_nll_mean_part_floats(R) floats of scratch; two launches, sums in a fixed order (deterministic).
 *   upp_nll_mean_bwd        : g_logp (R,C) = -g_loss[0] / R at (r, target[r]), 0 elsewhere; g_loss a DEVICE scalar. */
int upp_logsoftmax_rows_fwd(const float *y, long long ld_y, const float *bias, long long R, int C, float *logp, void *stream);
int upp_logsoftmax_rows_bwd(const float *g_logp, const float *logp, long long R, int C, float *g_y, long long ld_gy, int Cpad, void *stream);
long long upp_nll_mean_part_floats(long long R);
int upp_nll_mean_fwd(const float *logp, const int64_t *target, long long R, int C, float *part, float *out, void *stream);
int upp_nll_mean_bwd(const float *g_loss, const int64_t *target, long long R, int C, float *g_logp, void *stream);
/* ---- noise-vector supervision of the pre-task recipe (round 6) -------------------------------------------------------
 * Replaces reference models/Point_MAE_pretask_dev.py:685-692 for a 3-channel rectify prompter:
 *   loss = mean_{b, noise i} |pred_noise - noise_vector|^2 + mean_{b, shape i} |pred_pure|^2,   score = |pred|
 * (`torch.mean(torch.norm(x, 2, dim=-1, keepdim=True) ** 2)` twice and `torch.norm(pred, dim=-1)`: ~30 element-wise launches with the
 * zero guards of norm's backward) where pred (B,P,3) holds pn shape points then P - pn noise points per cloud and noise_vector is
 * (B, P - pn, 3).  upp_noise_loss_fwd: loss[0], score (B,P) or NULL, `part` upp_noise_loss_part_floats(B, P) floats of scratch; two launches,
 * fixed-order sums.  upp_noise_loss_bwd: g_pred (B,P,3) = g_loss[0] * d loss / d pred (g_loss a device scalar), every element written. */
long long upp_noise_loss_part_floats(int B, int P);
int upp_noise_loss_fwd(const float *pred, const float *noise_vector, int B, int P, int pn, float *part, float *loss, float *score, void *stream);
int upp_noise_loss_bwd(const float *g_loss, const float *pred, const float *noise_vector, int B, int P, int pn, float *g_pred, void *stream);
/*   upp_bn_relu_drop_fwd / bwd : BatchNorm1d + ReLU + Dropout of a (R, C) matrix with few rows (the Linear outputs of
 *                      cls_head_finetune, R = batch size), one launch each way.  training != 0: batch statistics (mean / rstd
 *                      outputs, running statistics updated with momentum / unbiased variance when non-NULL); else running
 *                      statistics.  u (R,C) uniforms or NULL (no dropout): kept where u >= p, scaled by 1/(1-p).
 *                      Backward recomputes the ReLU / dropout masks from z and u; g_gamma / g_beta (C) are complete sums. */
int upp_bn_relu_drop_fwd(const float *z, const float *gamma, const float *beta, float *running_mean, float *running_var,
                         float momentum, float eps, int training, const float *u, float p, float *a, float *mean, float *rstd,
                         int R, int C, void *stream);
int upp_bn_relu_drop_bwd(const float *g_a, const float *z, const float *gamma, const float *beta, const float *mean,
                         const float *rstd, int training, const float *u, float p, float *g_z, float *g_gamma, float *g_beta,
                         int R, int C, void *stream);

/* ---- training step tail: gradient clipping + AdamW on flat buffers ------------------------
 * Replaces torch.nn.utils.clip_grad_norm_(params, max_norm) + torch.optim.AdamW.step() of the reference loop
 * (tools/runner_module.py:202-207; parameter groups of tools/builder.py:40-55) for parameters that live in one
 * flat buffer: p, g, m (exp_avg), v (exp_avg_sq) all (n) f32; elements [0, split) have weight decay 0,
 * [split, n) have `weight_decay`.  state (8 floats, zero-initialised once): [0] step count, [1] gradient L2 norm,
 * [2] clip coefficient min(max_norm / (norm + 1e-6), 1), [3] 1 - beta1^step, [4] sqrt(1 - beta2^step).
 * g is overwritten with the clipped gradient (as clip_grad_norm_ does).  max_norm <= 0 disables clipping.
 * lr < 0: the learning rate and the weight decay are read from state[5] and state[6] on the device instead of the
 * by-value arguments -- a launch captured in a HIP graph then follows a learning-rate schedule (the reference steps
 * a CosineLRScheduler per epoch, tools/builder.py:67-75 / tools/runner_module.py:217-221) without being re-captured.
 * scratch: upp_adamw_scratch_floats() floats. */
long long upp_adamw_scratch_floats(void);
/* upp_batched_sum: for `jobs` independent jobs in one launch, dst_j[c] (+)= sum_{i < n_j} src_j[i * ld_j + c], c < len_j
 * (rows added in ascending order; accumulate_j != 0 adds to dst_j, else overwrites; jobs that name the same dst are
 * applied one after the other in submission order by the same workgroups: race-free and deterministic).  src / dst / n / len / ld /
 * accumulate are HOST arrays; the pointers in src / dst are device pointers.  Used for the parameter-gradient partials of
 * a backward pass (upp_adapter_bwd partials per workgroup, upp_ln_param_grad partials per chunk, upp_rowln_bwd per-sample
 * prompt gradients), summed straight into the flat gradient buffer after the pass.
 * dst_width / dst_pitch (ABI 5; HOST arrays, both NULL = every destination flat): dst_width_j > 0 makes destination j a WINDOW of
 * (len_j / dst_width_j) rows of dst_width_j floats, dst_pitch_j floats apart -- element c lives at dst_j[(c / w) * pitch + c % w]: the
 * weight gradient of a Linear layer applied to a column range of a wider weight (the segmentation head's first layer acts on
 * [per-point 1024 | per-sample 2432] columns of one (512, 3456) weight, reference models/Point_MAE_unify_segment.py:424-433) lands in
 * that range of the weight's gradient without a zero-fill, a strided copy and an add.  len_j % dst_width_j == 0. */
int upp_batched_sum(const float *const *src, float *const *dst, const int *n, const int *len, const int *ld,
                    const int *accumulate, const int *dst_width, const int *dst_pitch, int jobs, void *stream);
/* upp_copy_batched: dst_j[0 .. bytes_j) = src_j[0 .. bytes_j) for `count` non-overlapping device buffers in one launch (host arrays of
 * device pointers and byte counts): the hand-over state a pipelined training step passes from its front-end to its back-end. */
int upp_copy_batched(const void *const *src, void *const *dst, const long long *bytes, int count, void *stream);
/* upp_colsum_partials: dst (chunks, len)[ch][c] = sum of src[r][c] over the rows r of chunk ch (ceil(n / chunks) rows each, ascending):
 * first stage of the column sum of one very tall matrix -- the bias gradient of a trainable Linear / 1x1 Conv1d over the 65,536
 * point rows of the patch embedding (reference models/Point_MAE_unify.py:191-222 under autograd) -- the second stage is
 * upp_batched_sum over the `chunks` rows. */
int upp_colsum_partials(const float *src, long long ld, int n, int len, int chunks, float *dst, void *stream);
/* upp_wcolsum_partials: dst (chunks, W, len)[ch][w][c] = sum over the rows r of chunk ch of wts[r][w] * src[r][c] (W <= 4; every fourth
 * row per wavefront, wavefronts combined in order): the weight gradient x^T . g of a rank-W update y += x (rows, W) . wt (W, len) over
 * very many rows -- the xyz columns of the first feature-propagation layer of the segmentation head over its 65,536 label points
 * (reference models/Point_MAE_unify_segment.py:420,605 under autograd).  Second stage: sum over the chunks (upp_batched_sum).
 * Limits: len % 4 == 0, ld % 4 == 0, src and dst 16-byte aligned. */
int upp_wcolsum_partials(const float *src, long long ld, const float *wts, long long ldw, int W, int n, int len, int chunks,
                         float *dst, void *stream);
int upp_adamw_flat(float *p, float *g, float *m, float *v, long long n, long long split, float *state, float *scratch,
                   float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm, void *stream);

/* ---- bottleneck adapter --------------------------------------------------------------------
 * Replaces Adapter.forward after its LayerNorm and the residual that applies it (reference
 * models/Point_MAE_pretask_dev.py:96-104 and :312-320):
 *   out = x + scale * ( W2 . dropout_p(gelu(W1 . ha + b1)) + b2 )        scale = 0.7, exact (erf) GELU
 *   ha, x, out (R, D); W1 (H, D), b1 (H), W2 (D, H), b2 (D); s1 (R, H) = W1 . ha + b1 is saved for backward;
 *   u (R, H) uniforms in [0,1) select the dropout mask (kept iff u >= p, scaled by 1/(1-p)); NULL = no dropout.
 * Backward: g_ha (R, D) and, per workgroup of 32 rows, partial sums [dW1 (H,D) | dW2 (D,H) | db1 (H) | db2 (D)] in
 * `part` (upp_adapter_part_floats(R, D) floats; the caller sums the workgroups).  The gradient w.r.t. x is g_out.
 * Limits: D == 384, H == 32. */
long long upp_adapter_part_floats(int R, int D);
int upp_adapter_fwd(const float *ha, const float *x, const float *W1, const float *b1, const float *W2, const float *b2,
                    const float *u, float p, float scale, float *out, float *s1, int R, int D, int H, void *stream);
int upp_adapter_bwd(const float *g_out, const float *ha, const float *s1, const float *W1, const float *W2, const float *u,
                    float p, float scale, float *g_ha, float *part, int R, int D, int H, void *stream);
/* upp_ln_adapter_fwd: the row operator that closes a block (residual + drop path, prompt strip, the adapter's LayerNorm:
 * upp_rowln_fwd with mode 0 / 3 / 4 and no `add`) and upp_adapter_fwd in ONE launch on 16-row workgroups; the LayerNorm
 * output is never written (reference models/Point_MAE_pretask_dev.py:305-320: `x = x + drop_path(mlp(norm2(x)))`, prompt
 * removal, `x = x + adapter(x)` with Adapter.forward :96-104).
 *   xo (B, Lout, D) = x[src(t)] + dp_scale(u_b) * (y[src(t)] + ybias)     (y, ybias, u may be NULL)
 *   mean, rstd (B, Lout): LayerNorm statistics of xo;  s1 (B*Lout, H);  out (B, Lout, D) = adapter(LayerNorm(xo), xo)
 *   ud (B*Lout, H): dropout uniforms of the adapter or NULL.  Rows and statistics are bit-identical to upp_rowln_fwd's.
 * upp_ln_adapter_bwd = upp_adapter_bwd with the LayerNorm output rebuilt from (xo, mean, rstd, gamma, beta) while the tile is
 * staged; the row operator's own backward stays upp_rowln_bwd (g_xo = g_out, g_h = g_ha). */
int upp_ln_adapter_fwd(const float *x, const float *y, const float *ybias, const float *u, float keep, int mode, int P,
                       const float *gamma, const float *beta, float eps, const float *W1, const float *b1, const float *W2,
                       const float *b2, const float *ud, float p, float scale, float *xo, float *mean, float *rstd, float *s1,
                       float *out, int B, int Lin, int Lout, int D, int H, void *stream);
/* upp_ln_adapter_bwd_fused: the whole backward of upp_ln_adapter_fwd in one launch on 16-row workgroups -- the adapter's backward
 * (g_ha, per-workgroup partials [dW1 (H,D) | dW2 (D,H) | db1 (H) | db2 (D)] in `part`, upp_ln_adapter_part_floats(R, D) floats, or
 * part = NULL), the LayerNorm backward with the residual (g_x / g_y (B, Lin, D): every row is written, zeros for the prompt rows
 * the strip map dropped; either may be NULL), and the LayerNorm parameter-gradient partials (`ln_part`: [workgroup][2][D], or NULL).
 * Same results as upp_ln_adapter_bwd + upp_rowln_bwd up to f32 re-association of the small products. */
long long upp_ln_adapter_part_floats(int R, int D);
int upp_ln_adapter_bwd_fused(const float *g_out, const float *xo, const float *mean, const float *rstd, const float *gamma,
                             const float *beta, const float *s1, const float *W1, const float *W2, const float *ud, float p,
                             float scale, const float *u, float keep, int mode, int P, float *g_x, float *g_y, float *part,
                             float *ln_part, int B, int Lin, int Lout, int D, int H, void *stream);
int upp_ln_adapter_bwd(const float *g_out, const float *xo, const float *mean, const float *rstd, const float *gamma,
                       const float *beta, const float *s1, const float *W1, const float *W2, const float *u, float p,
                       float scale, float *g_ha, float *part, int R, int D, int H, void *stream);
/* The adapter weight gradients from their factors (round 6; reference: autograd through Adapter.forward,
 * models/Point_MAE_pretask_dev.py:96-104, for the trainable `downstream_adapter` of every block).  upp_ln_adapter_bwd_fused writes
 * 24,992 floats of partial results per 16-row workgroup (153 MB per headline step, summed by upp_batched_sum).
 * upp_ln_adapter_bwd_factors is the same launch with `fac` (B*Lout, 2 H) = [ga | d] per row instead of `part`:
 *   ga = gradient at the hidden pre-activation, d = scale * dropout(gelu(s1)).
 * upp_adapter_wgrad_batched then forms, in ONE launch for `jobs` blocks (any number; 16 per kernel launch),
 *   dW1 = sum_rows ga^T . LayerNorm(xo),  dW2 = sum_rows g_out^T . d,  db1 = sum_rows ga,  db2 = scale sum_rows g_out
 * over `splits` row ranges per block: part[j] = (splits, 2 H D + H + D) floats in the per-workgroup layout above, to be summed over
 * its rows (upp_batched_sum).  upp_adapter_wgrad_splits(R): the split count the host side uses (128 rows per split, at most 64).
 * D = 384, H = 32; xo / g_out / fac / gamma / beta 16-byte aligned.  Deterministic; the association of the row sum differs from the
 * per-workgroup form's. */
int upp_ln_adapter_bwd_factors(const float *g_out, const float *xo, const float *mean, const float *rstd, const float *gamma,
                               const float *beta, const float *s1, const float *W1, const float *W2, const float *ud, float p,
                               float scale, const float *u, float keep, int mode, int P, float *g_x, float *g_y, float *fac,
                               float *ln_part, int B, int Lin, int Lout, int D, int H, void *stream);
int upp_adapter_wgrad_splits(int R);
int upp_adapter_wgrad_batched(const float *const *xo, const float *const *mean, const float *const *rstd, const float *const *gamma,
                              const float *const *beta, const float *const *g_out, const float *const *fac, const int *R,
                              const float *scale, float *const *part, int jobs, int splits, int D, int H, void *stream);

/* ---- tail of the denoising prompter ---------------------------------------------------------
 * Replaces RectifyPrompter.score_head (reference models/Point_MAE_pretask_dev.py:491-493,512: Linear(32,64) -> ReLU -> Dropout(0.2)
 * -> Linear(64,3), times score_factor) and the selection that follows it in the model (models/Point_MAE_unify.py:553-559):
 *   pred = head(feature) * factor;  score = ||pred||_2;  order = argsort(score, descending, stable);
 *   moved = pts + nudge * pred;     out = moved[order[N - keep :]]            (reference: nudge 0.2, keep int(0.95 point_num))
 * feature (B*N,32) 16-byte aligned, W0 (64,32), b0 (64), W1 (3,64), b1 (3), u (B*N,64) uniforms in [0,1) or NULL (no dropout;
 * with u: hidden unit kept iff u >= p and scaled by 1 / (1 - p)), pts (B,N,3).
 * Outputs: moved (B,N,3), score (B,N) (both always written: scratch of the second launch), out (B,keep,3), pred (B,N,3) or NULL,
 * order (B,N) int64 or NULL (the full descending order; ties: lower index first).  Limits: N <= 16384. */
int upp_rectify_select(const float *feature, const float *W0, const float *b0, const float *W1, const float *b1, const float *u, float p,
                       float factor, const float *pts, float nudge, int B, int N, int keep, float *pred, float *moved, float *score,
                       float *out, int64_t *order, void *stream);

/* Stable argsort of every row of key (B,N) f32 by rank counting (no library sort): order (B,N) int64 with order[b][r] = the index of the
 * r-th element of row b in ascending (descending != 0: descending) order, equal keys in index order -- what torch.argsort(...,
 * stable=True) returns, and what torch.argsort returns for distinct keys.  NaN ranks ABOVE +inf (last ascending, first descending: torch.sort's order) and -0 equals +0; always a permutation.  Replaces the reference's device sorts of
 * short rows: torch.argsort(score, descending=True) of the pre-task noise recall (models/Point_MAE_pretask_dev.py:702-704), the masking
 * orders of Point-MAE pre-training (models/Point_MAE.py:300-329: argsort of uniform draws / of distances to a random centre, and the
 * visible-first order argsort(mask)).  Limits: N <= 16384 (the row lives in the LDS). */
int upp_argsort_rows(const float *key, int B, int N, int descending, int64_t *order, void *stream);

/* Max-pool over the k rows of every group with its arg-max, and the backward that writes the whole gradient in one pass (replaces
 * `x.max(dim)[0]` under autograd at the max-pool sites whose input carries a gradient -- reference models/Point_MAE_unify.py:205,221 (the
 * patch embedding when it trains: pre-training, stage 2), models/Point_MAE_pretask_dev.py:413 (set abstraction), :294 (`pooling`) -- where
 * torch launches a reduce forward and a zero-fill + scatter backward):
 *   x (R, k, C) -> out (R, C) = max_j x[r][j][c], amax (R, C) uint8 = the first j that attains it;  g_x (R, k, C) = g[r][c] at j = amax[r][c],
 *   0 elsewhere.  k <= 255, C % 4 == 0, x / out / g / g_x 16-byte and amax 4-byte aligned. */
int upp_group_max_fwd(const float *x, int R, int k, int C, float *out, unsigned char *amax, void *stream);
int upp_group_max_bwd(const float *g, const unsigned char *amax, int R, int k, int C, float *g_x, void *stream);

/* ---- token-matrix Linear (exact f32 on the matrix cores) -------------------------------------
 * Replaces the nn.Linear layers of the Transformer blocks and their data gradients: Attention.qkv / .proj
 * (reference models/Point_MAE_pretask_dev.py:178,181 called :186,:194) and Mlp.fc1 / .fc2 (:158,:160 called
 * :164-168), i.e. torch.nn.functional.linear -> cuBLAS in the reference.
 *   C (M,N) = epilogue( A (M,K) . W (N,K)^T )       A, W, C row-major with leading dimensions lda, ldw, ldc (floats)
 * f32 in, f32 accumulate on v_mfma_f32_32x32x2_f32: every output is a k-ordered chain of fused multiply-adds
 * (one rounding per product), the order of k being a fixed permutation inside each group of 32.
 * A data gradient dX = dY . W is the same call with W^T stored row-major: (A = dY, W = W^T (K,N), N <-> K).
 *   epilogue 0: none                      1: + bias[n]
 *            2: GELU(. + bias[n])         3: GELU(. + bias[n]), and aux (M,N; ldaux) receives GELU'(. + bias[n])
 *            5: ReLU(. + bias[n])
 *            4: . * aux[m][n]   (exact erf GELU, as nn.GELU; 3 / 4 make fc1 forward / fc2 data-gradient carry the
 *               activation and its backward, reference :165 `self.act`)
 *   tile: 0 = chosen by the library for (M,N,K) (upp_linear_tile returns that choice); else 4096*BMB + 256*BNB + 16*KS + KC
 *         forces workgroups of BMB x BNB blocks of 32 x 32 (one block per wave), the contraction split KS ways over wave
 *         groups and 32*KS*KC values of k per LDS stage -- one of the compiled shapes (csrc/linear.hip UPP_LIN_CONFIGS),
 *         anything else is UPP_E_RANGE; for measurements.  Codes with bit 16 set name the register-tiled shapes of
 *         csrc/linear_rt.hip (tall matrices: thousands of tiles): 0x1000000 NST + 0x100000 (4 (RM-1) + RN-1) + 0x10000 + 4096 WM +
 *         256 WN + 16 + KC = WM x WN waves of RM x RN blocks each, NST LDS stages of 32 KC values of k, no split of the
 *         contraction (the low byte reads KS = 1, KC: the summation order).
 * Limits: K % 4 == 0 (a k-stage holds 32 KS KC values of k; the last one reads zeros beyond K: exact), lda % 4 == 0, ldw % 4 == 0,
 * A and W 16-byte aligned. */
int upp_linear_tile(int M, int N, int K);
/* Weight gradients of trainable Linear layers (AddmmBackward's second GEMM in the reference: every nn.Linear / 1x1 Conv1d under
 * autograd -- models/Point_MAE_cp.py:369-465 in pre-training, models/Point_MAE_unify_segment.py:420-433 for the segmentation head):
 * dW_p (N_p,K_p) = G_p^T . X_p with G_p = dY (M_p,N_p) and X_p (M_p,K_p) row-major (leading dimensions ldg, ldx), for `count`
 * layers in ONE launch (a weight gradient is read by nothing inside the backward pass, so a step driver issues them together when
 * the pass is over).  The rows of problem p are cut into runs of rows[p] (a multiple of 32; upp_linear_wgrad_grouped_rows plans them
 * for a group: equal work units, about six rounds of the resident workgroups); partials[p] (ceil(M_p / rows[p]), N_p, K_p) receives
 * one partial dW per run -- every entry ONE ascending-row chain of fused multiply-adds -- and the caller sums the runs in order
 * (upp_batched_sum).  All array arguments are HOST arrays; G / X / partials hold device pointers.
 * upp_linear_wgrad_f32 / upp_linear_wgrad_splits: a group of one.
 * Limits: N % 4 == 0, K % 4 == 0, ldg % 4 == 0, ldx % 4 == 0, G, X and partials 16-byte aligned. */
int upp_linear_wgrad_grouped_rows(int count, const int *M, const int *N, const int *K, int *rows);
int upp_linear_wgrad_grouped_f32(const float *const *G, const long long *ldg, const float *const *X, const long long *ldx,
                                 float *const *partials, const int *M, const int *N, const int *K, const int *rows,
                                 int count, void *stream);
/* upp_linear_wgrad_grouped_sb (round 4): the same work list, limits and partial layout on the BF16 matrix pipe -- both operands split
 * into three bf16 terms inside the kernel, six products per pair as upp_linear_sb_f32 (error of an f32 GEMM against float64; each entry a
 * fixed-order sum, not bit-comparable with the f32 chain).  csrc/wgrad_sb.hip. */
int upp_linear_wgrad_grouped_sb_rows(int count, const int *M, const int *N, const int *K, int *rows);   /* its plan of rows[] (tile classes) */
int upp_linear_wgrad_grouped_sb(const float *const *G, const long long *ldg, const float *const *X, const long long *ldx,
                                float *const *partials, const int *M, const int *N, const int *K, const int *rows,
                                int count, void *stream);
int upp_linear_wgrad_splits(int M, int N, int K);
int upp_linear_wgrad_f32(const float *G, long long ldg, const float *X, long long ldx, float *partials,
                         int M, int N, int K, void *stream);
int upp_linear_f32(const float *A, long long lda, const float *W, long long ldw, const float *bias,
                   float *C, long long ldc, float *aux, long long ldaux,
                   int M, int N, int K, int epilogue, int tile, void *stream);
/* upp_linear_group_bias_f32: C = A . W^T + bias[m >> group_shift][:], bias a (ceil(M / 2^group_shift), N) matrix -- a bias per group of
 * 2^group_shift >= 32 consecutive rows in the GEMM's epilogue: `h + per_sample.unsqueeze(1)` of the segmentation head
 * (models/Point_MAE_unify_segment.py:424-433 with the concat's broadcast half multiplied once per sample) and the per-group half of the
 * patch embedding's Conv1d(512, 512) (models/Point_MAE_unify.py:213-216).  Tall problems only (the register-tiled kernel:
 * upp_linear_tile(M, N, K) & 0x10000), UPP_E_RANGE otherwise. */
int upp_linear_group_bias_f32(const float *A, long long lda, const float *W, long long ldw, const float *bias, int group_shift,
                              float *C, long long ldc, int M, int N, int K, void *stream);
/* ---- the same Linear layers at f32 accuracy on the BF16 matrix pipe (csrc/linear_sb.hip) ------------------------------
 * Replaces the same reference calls as upp_linear_f32 (models/Point_MAE_pretask_dev.py:153-196, nn.Linear -> cuBLAS) for FROZEN
 * weights: C (M,N) = epilogue( A (M,K) . W (N,K)^T ) with every f32 operand split exactly into three bf16 terms (x = x1 + x2 + x3,
 * round-to-nearest, both residuals exact) and the six products of weight >= 2^-16 accumulated in f32 by v_mfma_f32_32x32x16_bf16;
 * the dropped terms are below 2^-25 |a w|: the error against an exact product is that of an f32 GEMM (tests/test_gpu_linear_sb.py
 * measures both against float64), at 6/16 of the matrix-pipe time of v_mfma_f32_32x32x2_f32.  Not bit-pinned (the summation order
 * inside a bf16 MFMA is not documented): the exact-f32 kernel above stays the bit-pinned one.
 *   upp_linear_sb_planes_bytes(N, K): size of the plane image of a (N,K) weight (6 bytes per element, N and K rounded up to 32).
 *   upp_linear_sb_prep(W, ldw, N, K, transposed, planes): split the (N,K) weight operand once per weight version into the kernel's LDS image
 *       [32-row block][32-wide k-stage][plane 0..2][16-byte granule of 8 k][row] (zero beyond N, K); planes 16-byte aligned.  transposed != 0:
 *       the operand is the TRANSPOSE of the row-major (K,N) matrix at W (leading dimension ldw >= N) -- the B operand of a data gradient
 *       dX = dY . W read straight from W, no f32 copy of W^T.  upp_linear_sb_prep_batched: `count` weights in one launch (HOST arrays of
 *       device pointers / sizes): the trainable weights of a step driver, re-split at the start of every step.
 *   upp_linear_sb_tile(M, N, K): the tile code the library would take (hex digits 0x4 BMB BNB RN KS NST: workgroups of
 *       BMB x BNB blocks of 32 x 32, a wave owns 1 x RN of them, the contraction cut over KS wave groups, NST LDS stages of 32 KS values
 *       of k), or 0 when the problem is not one for this kernel (K % 32, fewer k-stages than LDS stages, no tile fits).  The choice is that
 *       of a cost model fitted to a measured sweep of every tile shape over the Linear problems of the six recipes (csrc/linear_sb_model.h,
 *       tools/micro/sb_model_fit.py, profiles/r05_sb_sweep.json: within 3 % of the measured best on 95 % of the problems, also of problems
 *       held out of the fit), behind a nine-row table of the swept problems where the model is still behind (csrc/linear_sb_tuned.h, exact
 *       (M, N, K) matches); option UPP_OPT_SB_TUNED = 0 leaves the model alone.  Any
 *       compiled tile code is valid for any problem whose K its wave groups' k-stages divide (a forced code: upp_linear_sb_f32's `tile`).
 *   upp_linear_sb_f32: epilogues and aux as upp_linear_f32; tile 0 = the library's choice.  Limits: K % 32 == 0 (% 64, % 128 for KS = 2, 4 tiles),
 *       N % 4 == 0, lda % 4 == 0, ldc % 4 == 0, ldaux % 4 == 0, A / C / bias / aux 16-byte aligned; UPP_E_RANGE otherwise (the caller
 *       then takes upp_linear_f32).
 *       Value range: the split is exact for finite operands up to the largest bf16 (|x| <= 0x7F7F0000 = 3.39e38); a finite weight beyond
 *       it counts as infinite (upp_linear_sb_prep zeroes its residual terms instead of forming inf - inf).
 *       Non-finite operands: x = +-inf splits into (inf, NaN, ...) in the loop and into (inf, 0, 0) in upp_linear_sb_prep; either way the
 *       six-product sum mixes +inf and -inf terms (the residual terms of the OTHER operand carry both signs), so every output a non-finite
 *       operand reaches is non-finite -- NaN where upp_linear_f32 yields +-inf.  Overflow still surfaces, its sign does not.
 *   upp_linear_sb_group_bias_f32: C = A . W^T + bias[m >> group_shift][:] (a bias per group of 2^group_shift >= 32 rows), the split-bf16
 *       form of upp_linear_group_bias_f32 (reference models/Point_MAE_unify_segment.py:424-433, models/Point_MAE_unify.py:213-216). */
int upp_linear_sb_tile(int M, int N, int K);
long long upp_linear_sb_planes_bytes(int N, int K);
int upp_linear_sb_prep(const float *W, long long ldw, int N, int K, int transposed, void *planes, void *stream);
int upp_linear_sb_prep_batched(const float *const *W, const long long *ldw, const int *N, const int *K, const int *transposed,
                               void *const *planes, int count, void *stream);
int upp_linear_sb_f32(const float *A, long long lda, const void *planes, const float *bias, float *C, long long ldc, float *aux,
                      long long ldaux, int M, int N, int K, int epilogue, int tile, void *stream);
int upp_linear_sb_group_bias_f32(const float *A, long long lda, const void *planes, const float *bias, int group_shift, float *C,
                                 long long ldc, int M, int N, int K, void *stream);

/* upp_linear_smallk_f32: y (M,N) = act(x (M,K) . W (N,K)^T + bias) for the Linear layers upp_linear_f32 does not take (K not a
 * multiple of 4, unaligned rows): the first layer of every position MLP (K = 3; reference models/Point_MAE_unify.py pos_embed /
 * models/Point_MAE_pretask_dev.py:395-399 `nn.Linear(3, 128), nn.GELU(), nn.Linear(128, dim)`) and the first point-wise layer of
 * the rectify prompter's feature propagation (K = 59, :475-517).  act: 0 none, 1 ReLU, 2 GELU (erf).  VALU kernel, sums over k in
 * ascending order.  Limits: K <= 64, N <= 256.
 * upp_transpose_f32: dst (cols, rows; ld_dst) = src (rows, cols; ld_src)^T -- the W^T a data-gradient GEMM needs of a TRAINABLE
 * weight (frozen ones are transposed once and cached by the caller). */
int upp_linear_smallk_f32(const float *x, long long ldx, const float *W, long long ldw, const float *bias, float *y, long long ldy,
                          int M, int N, int K, int act, void *stream);
/* upp_linear_smallk_gelu_d_f32 (round 6): y = GELU(x . W^T + bias) and d = GELU'(x . W^T + bias) in the same pass (erf form, as
 * upp_linear_f32's epilogue 3): the first layer of a TRAINABLE position MLP keeps the derivative for its backward. */
int upp_linear_smallk_gelu_d_f32(const float *x, long long ldx, const float *W, long long ldw, const float *bias, float *y, long long ldy,
                                 float *d, long long ldd, int M, int N, int K, void *stream);
/* upp_linear_smallk_wgrad_f32: partials (chunks, N, K)[c][n][k] = sum over the rows m of chunk c of G[m][n] * X[m][k] -- the weight gradient of
 * a small Linear (N K <= 2048, any alignment) over many rows, rows added in ascending order; the caller sums the chunks (upp_batched_sum).
 * (The rectify prompter's point-wise layers in the pre-task recipe: 32 x 27, 64 x 3, ... over 34,432 rows.) */
int upp_linear_smallk_wgrad_f32(const float *G, long long ldg, const float *X, long long ldx, float *partials, int M, int N, int K,
                                int chunks, void *stream);
int upp_transpose_f32(const float *src, long long ld_src, float *dst, long long ld_dst, int rows, int cols, void *stream);
/* upp_transpose_batched_f32: dst_j (cols_j, rows_j) = src_j (rows_j, cols_j)^T for `count` contiguous matrices in one launch (host arrays of
 * device pointers and sizes): the W^T copies of ALL trainable weights of a recipe, refreshed once per training step. */
int upp_transpose_batched_f32(const float *const *src, float *const *dst, const int *rows, const int *cols, int count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* UPP_HIP_H */
