/* ppf_hip.h -- C ABI of the MI355X (gfx950) ProtoPFormer hot-path library (protopformer_amd/lib/libppf_hip.so).
 *
 * The reference (zju-vipa/ProtoPFormer) has no native/FFI layer: its hot path is Python calling ATen.  This header is
 * the boundary a maintainer binds instead (ctypes stub in INTEGRATION.md): one entry point per fused operation, each
 * citing the reference lines it replaces (paths relative to the reference repo; deit = tools/deit_models_attn.py,
 * cait = tools/cait_models_attn.py).
 *
 * Conventions
 *   - plain C: raw device pointers + explicit sizes/strides; no torch types.  "bf16" buffers are uint16_t storage.
 *   - every function enqueues work on `stream` and returns immediately (asynchronous w.r.t. the host).
 *   - the caller owns all memory; the library keeps no pointers after return and allocates nothing persistent.
 *   - return 0 on success, <0 for invalid shape/alignment/argument (PPF_ERR_*), >0 = hipError_t of a failed launch;
 *     ppf_last_error() returns a thread-local message.  No exceptions cross the boundary.
 *   - one host thread per process drives one GPU (data parallel = one process per GPU); re-entrant across streams.
 */
#ifndef PPF_HIP_H
#define PPF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* ppf_stream_t; /* == hipStream_t */

#define PPF_ERR_SHAPE (-1)
#define PPF_ERR_ALIGN (-2)
#define PPF_ERR_ARG (-3)

/* Bumped whenever an entry point changes its parameter list or meaning.  ppf_abi_version() returns the value the library was built
 * with; the binding (protopformer_amd/_lib.py EXPECTED_ABI) refuses a library of another version instead of shifting arguments. */
#define PPF_ABI_VERSION 10

/* ---- runtime ------------------------------------------------------------------------------------------------- */
const char* ppf_last_error(void);
int ppf_abi_version(void);
int ppf_device_info(int* cu_count, int* clock_mhz, char* name, int name_len);
/* stream ordering for the two-lane schedule of the host mirror (weight gradients / head-mean maps / prototype gradients on a side
 * stream): pooled, timing-disabled events inside the library; one call per dependency.
 * wait_stream: everything on src so far happens-before later work on dst.  mark / wait_mark: a ticket for "src so far".
 * arm (round 6): while a stream is armed (on = 1) every kernel the library launches on it carries its own completion event
 * (hipExtLaunchKernel stop event), and the next wait_stream(dst, stream) waits for the last of them instead of recording an event --
 * no packet enters the producer's queue.  Armed by the replay loop around the one library call that precedes a wait. */
int ppf_stream_arm(ppf_stream_t stream, int on);
int ppf_stream_wait_stream(ppf_stream_t dst, ppf_stream_t src);
int64_t ppf_stream_mark(ppf_stream_t stream);
int ppf_stream_wait_mark(ppf_stream_t stream, int64_t ticket);

/* ---- dense bf16 MFMA GEMM family: every nn.Linear / 1x1 conv / PatchEmbed conv of the path ----------------------
 * C[m][n] (+)= epi( sum_kc A(m,kc) * B(n,kc) ), fp32 accumulate.
 *   trans_a = 0: A[m*lda + kc]      trans_a = 1: A[kc*lda + m]        (same for B with ldb / n)
 *   forward  y = x W^T + b                (0,0)  deit:47,58 (qkv, proj), timm Mlp fc1/fc2 (deit:80), PatchEmbed (deit:174),
 *                                                add_on_layers 1x1 conv (protopformer.py:111-114,171-172)
 *   dgrad    dx = dy W                    (0,1)  autograd of the above
 *   wgrad    dW += dy^T x (+ db)          (1,1)  split over kc, fp32 atomics into C, colsum[m] += sum_kc A(m,kc)
 * epi: 0 bf16 out | 1 f32 out | 2 bias+GELU(erf): C=gelu(pre) bf16, aux_out=gelu'(pre) as ONE BYTE per element (saved for backward:
 *        code q = rint(196 d + 26), d' = (q - 26) / 196: 0 and 1 are code points, |error| <= 2.55e-3) | 3 sigmoid f32 out |
 *      4 residual: C f32 = res + rowscale[m/rows_per_group]*colscale[n]*(acc+bias)  (DropPath deit:79-80, LayerScale
 *        cait:156-157), optional aux_out = raw branch output bf16 | 5 dGELU: C bf16 = acc * aux_in (aux_in = the gelu' codes of epi 2) | 6 atomic f32.
 * ldaux counts ELEMENTS of the aux tensor (bytes for epi 2 / 5, bf16 for epi 4). */
int ppf_gemm_bf16(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int trans_a,
                  int trans_b, int epi, const float* bias, const float* res, int ldres, const float* rowscale,
                  int rows_per_group, const float* colscale, const void* aux_in, void* aux_out, int ldaux, float* colsum,
                  float alpha, void* workspace, size_t workspace_bytes, ppf_stream_t stream);
/* split-K scratch an accumulating (epi 6) GEMM of this shape wants (deterministic ordered reduction instead of atomics).  Contract: the
 * workspace is handed to ppf_gemm_bf16 calls of ONE stream only (every weight gradient recycles it: the 28 MB of partial tiles live and die
 * in the L2 / memory-side cache); its first 16 KiB are a reserved prefix (the arrival counters of round 3's in-kernel reduce, deleted). */
size_t ppf_gemm_workspace_bytes(int M, int N, int K);

/* Test hook (tests/test_gpu_gemm.py): force = 1 sends every shape the 224 x 128 NT kernel can legally take to it, 0 = the cost model. */
int ppf_gemm_test_force_g224(int force);

/* Roofline probe of the split-K weight-gradient kernel (epi 6 with a workspace): HIP events (from a reused pool) on the launch
 * stream around the kernel itself.  ppf_gemm_probe(1) clears and starts, (0) stops, (2) stops and destroys the pool;
 * ppf_gemm_probe_read synchronises the events and returns the summed kernel time, the launch count and the algorithmic
 * flops / bytes of those launches (bench.py's `roofline`).  Enabled while a step is being captured into a HIP graph, the records
 * become event-record nodes: a read after replays returns the last replay's durations. */
int ppf_gemm_probe(int enable);
int ppf_gemm_probe_read(double* ms_total, int64_t* launches, double* flops, double* bytes);

/* Path probe (round 5): the same mechanism around the kernels BASELINE.json's north_star names -- tag 0 attention forward
 * (ppf_attn_fwd / ppf_attn_fwd_hm), 1 attention backward (ppf_attn_bwd), 2 prototype forward (ppf_proto_fwd) -- for bench.py's
 * roofline.named_path.  ppf_path_probe(1) clears and starts, (0) stops, (2) stops and destroys the pools; ppf_path_probe_read
 * returns the summed in-step kernel time, the launch count and the algorithmic flops / bytes of the recorded launches.
 * Single-threaded and eager-only: launches into a capturing stream are not recorded, and at most 4096 launches per tag are kept. */
int ppf_path_probe(int enable);
int ppf_path_probe_read(int tag, double* ms_total, int64_t* launches, double* flops, double* bytes);

/* nbatch = batch_outer*batch_inner plain GEMMs; problem (o,i) uses A + o*sa_o + i*sa_i (elements), same for B / C.
 * kpad = 1: contraction-contiguous operands may read the (zero) padding up to the next multiple of 8 beyond K.
 * CaiT talking-heads attention per-(sample, head) products: A.V, dO.V^T, dS.K, dS^T.Q, A^T.dO (cait:128-130 and autograd). */
int ppf_gemm_bf16_batched(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int trans_a,
                          int trans_b, int out_f32, float alpha, int batch_outer, int batch_inner, int64_t sa_o, int64_t sa_i,
                          int64_t sb_o, int64_t sb_i, int64_t sc_o, int64_t sc_i, int kpad, ppf_stream_t stream);

/* ---- fp32 verification mode, backward (csrc/precise.hip, round 5; DeiT): never on the measured path -----------------------------
 * LayerNorm backward with the statistics recomputed from x (rows of dy <-> rows row_map[r] of x, dx_out written at the x rows -- dx_out may
 * alias dres_in; dw / db by fp32 atomics, either may be NULL for a frozen LayerNorm); elementwise pieces (kind 0 x gelu'(pre), 1 sigmoid', 2 DropPath row scale, 3 product, 4 row scale x LayerScale column, 5 ReLU' from its output); column sums; Attention backward with the
 * policy softmax of deit:29-43 (scratch: B*H*2*N*N floats). */
int ppf_layernorm_bwd_f32(const float* dy, const float* x, const int* row_map, const float* w, const float* dres_in, float* dx_out, float* dw, float* db,
                          int rows, int D, float eps, ppf_stream_t stream);
int ppf_ew_bwd_f32(int kind, const float* a, const float* b, float* out, const float* rowscale, int rows_per_group, int M, int N, ppf_stream_t stream);
int ppf_colsum_f32(const float* in, float* out, int M, int N, ppf_stream_t stream);
int ppf_attn_bwd_f32(const float* qkv, const float* dout, const float* policy, float* dqkv, float* scratch, int B, int H, int N, int D, int self_keep,
                     int eps_n, ppf_stream_t stream);
/* CaiT: TalkingHeadAttn backward (cait:115-132; dqkv zero-filled by the caller, mixer gradients accumulate) and ClassAttn backward (cait:50-90) */
int ppf_th_attn_bwd_f32(const float* qkv, const float* dout, const float* wl, const float* bl, const float* ww, const float* bw, float* dqkv,
                        float* dwl, float* dbl, float* dww, float* dbw, int B, int H, int N, int D, ppf_stream_t stream);
int ppf_class_attn_bwd_f32(const float* q, const float* k, const float* v, const float* policy, const float* dout, float* dq, float* dk, float* dv,
                           int B, int H, int N1, int D, ppf_stream_t stream);

/* ---- full-row GEMM with row-wise fused epilogues (csrc/rowgemm.hip) ------------------------------------------------------------
 * acc = A[M][K] . B[D][K]^T with D = the model width (192 or 384): a workgroup owns complete output rows (tiles of rows_per_tile <=
 * 208 rows, the host passes the tokens of one sample), so what the reference does next to those rows runs in the epilogue and the
 * GEMM output / the separate LayerNorm pass never touch HBM.  Both operands contraction-contiguous, bf16; K % 32 == 0.
 *   ppf_rowgemm_bf16      out bf16 = acc + bias                                   input gradient of attn.proj (autograd of deit:58)
 *   ppf_rowgemm_resid_ln  xout = res + rowscale[m / rows_per_group] * colscale[n] * (acc + bias) (fp32, may alias res):
 *                         `x = x + drop_path(attn(...))` / `x = x + drop_path(mlp(...))` (deit:79-80, timm DropPath); colscale = CaiT's
 *                         LayerScale gamma_1 / gamma_2 (cait:153-155) or NULL, aux_out (optional) = bf16(acc + bias), the unscaled
 *                         branch its gradient needs; ln_out = bf16(LN(xout)), mean, rstd = the
 *                         LayerNorm that follows (deit:79 norm2 / the next block's norm1, eps 1e-6); ln_out == NULL: residual only
 *   ppf_rowgemm_lnbwd     dn = acc: gradient w.r.t. a LayerNorm output (input of qkv / fc1, deit:47 / timm Mlp fc1);
 *                         dx_out = dres_in + LN'(dn; x, mean, rstd, w) (fp32, may alias dres_in; NULL = 0);
 *                         cast_out = bf16(rowscale[m / rows_per_group] * dx_out) = the gradient entering the residual branch below
 *                         (optional); partial[tiles][2][D] = per-tile column sums (d ln weight, d ln bias), summed in a fixed order
 *                         by ppf_rowgemm_colsum (may run on another stream).  colscale != NULL = CaiT's LayerScale on the branch
 *                         below (cait:153-155): cast_out = bf16(rowscale * colscale[n] * dx_out), branch = the bf16 unscaled branch
 *                         output saved by ppf_rowgemm_resid_ln's aux_out, partial[tiles][3][D] with d gamma's sums third. */
int ppf_rowgemm_supported(int D, int K, int rows_per_tile);
int ppf_rowgemm_bf16(const void* A, const void* B, int M, int D, int K, int lda, int ldb, int rows_per_tile, const float* bias, void* out,
                     ppf_stream_t stream);
int ppf_rowgemm_resid_ln(const void* A, const void* B, int M, int D, int K, int lda, int ldb, int rows_per_tile, const float* bias,
                         const float* res, float* xout, const float* rowscale, int rows_per_group, const float* colscale, void* aux_out,
                         const float* ln_w, const float* ln_b, void* ln_out, float* ln_mean, float* ln_rstd, float eps, ppf_stream_t stream);
int ppf_rowgemm_lnbwd(const void* A, const void* B, int M, int D, int K, int lda, int ldb, int rows_per_tile, const float* x, const float* mean,
                      const float* rstd, const float* w, const float* dres_in, float* dx_out, void* cast_out, const float* rowscale,
                      int rows_per_group, const float* colscale, const void* branch, float* partial, size_t partial_bytes, ppf_stream_t stream);
int ppf_rowgemm_colsum(const float* partial, int tiles, int D, int nparts, float* dw, float* db, float* dg, ppf_stream_t stream);
/* Transposed bf16 copies of n weight matrices in one launch (the input-gradient products read W^T contraction-contiguous):
 * desc (device, int64 [n][4]) = (source element offset, destination element offset, rows, cols) into src / dst; dst[c][r] = src[r][c].
 * total_tiles = sum over the matrices of ceil(rows / 64) * ceil(cols / 64). */
int ppf_transpose_bf16_batched(const void* src, void* dst, const void* desc_i64, int n, int total_tiles, ppf_stream_t stream);

/* ---- LayerNorm (deit:67,72,238 norm1/norm2/norm, eps 1e-6) ----------------------------------------------------
 * fwd: y bf16 [rows][D] = LN(x fp32 [row_map ? row_map[r] : r][D]); saves mean / rstd per output row.
 * bwd: dx_out[src] = dres_in[src] + LN'(dy); dw/db accumulate.  Optional fused pass for the residual branch below:
 *      cast_out = bf16(rowscale*colscale*dx_out), dbias_next += its column sums, dcolscale += sum(rowscale*dx_out*branch).
 *      dy == NULL: only the fused pass over dres_in. */
int ppf_layernorm_fwd(const float* x, const int* row_map, const float* w, const float* b, void* y, float* mean, float* rstd,
                      int rows, int D, float eps, ppf_stream_t stream);
int ppf_layernorm_bwd(const void* dy, const float* x, const int* row_map, const float* w, const float* mean, const float* rstd,
                      const float* dres_in, float* dx_out, float* dw, float* db, void* cast_out, const float* rowscale,
                      int rows_per_group, const float* colscale, float* dbias_next, const void* branch, float* dcolscale,
                      int rows, int D, float* partial, size_t partial_bytes, ppf_stream_t stream);
/* partial != NULL: the column sums (dw, db, dbias_next, dcolscale) are NOT accumulated by ppf_layernorm_bwd; each workgroup
 * writes its partial sums into partial[ppf_layernorm_bwd_blocks(rows)][4][D] (the last rows are second-level scratch) and ppf_layernorm_bwd_reduce adds them up in a
 * fixed order (deterministic, no float atomics; may run on another stream).  partial == NULL: fp32 atomics. */
int ppf_layernorm_bwd_blocks(int rows);
int ppf_layernorm_bwd_reduce(const float* partial, int rows, int D, float* dw, float* db, float* dbias_next, float* dcolscale,
                             ppf_stream_t stream);

/* ---- attention with the policy softmax (deit:29-60; class attention cait:50-90 with self_keep = 0) -------------
 * qkv bf16 [B*N][3D] packed q|k|v, head h at columns h*hd.  policy [B][N] in {0,1} or NULL.
 * fwd saves rowmax and 1/(sum+eps) [B][H][N]; headmean = mean over heads of the probabilities, [B][N][NP] fp32
 * (the rollout input, deit:104 `attn.mean(axis=1)`), recomputed from those statistics.
 * eps_n: the N of the softmax's +eps/N term (deit:42); 0 = N.  Blocks that run on the reserved tokens only (compacted after the
 * rollout) pass the original token count so that the kept entries equal the masked full-length computation. */
int ppf_attn_fwd(const void* qkv, void* out, const float* policy, float* rowmax, float* zinv, int B, int H, int N, int D,
                 int self_keep, int eps_n, ppf_stream_t stream);
/* forward pass + head-mean map in ONE launch (16-row tiles: Q.K^T is computed once, N = 197 runs as 13 x 16 = 208 rows): headmean may be
 * NULL (no map).  ppf_attn_fwd_hm_supported: 1 for head_dim 64 and N <= 208, else use ppf_attn_fwd + ppf_attn_headmean. */
int ppf_attn_fwd_hm_supported(int H, int N, int D);
int ppf_attn_fwd_hm(const void* qkv, void* out, const float* policy, float* rowmax, float* zinv, float* headmean, int NP, int B, int H, int N,
                    int D, int self_keep, int eps_n, ppf_stream_t stream);
int ppf_attn_headmean(const void* qkv, const float* policy, const float* rowmax, const float* zinv, float* headmean, int NP,
                      int B, int H, int N, int D, int self_keep, int eps_n, ppf_stream_t stream);
int ppf_attn_bwd(const void* qkv, const void* out, const void* dout, void* dqkv, const float* policy, const float* rowmax,
                 const float* zinv, float* delta, int B, int H, int N, int D, int self_keep, int eps_n, ppf_stream_t stream);

/* ---- CaiT: talking-heads self-attention (cait:93-132) and class attention (cait:34-90) --------------------------------
 * th_scores: sp[b][g][q][key] = sum_h Wl[g][h]*(scale q_h.k_h) + bl[g];  th_softmax_mix (in place): sp <- softmax(sp),
 * a16 = bf16(proj_w(P)) [B][H][N][NPK], hm = mean over heads of proj_w(P) [B][N][NP] (rollout input, cait:228).
 * th_softmax_bwd: da (in dA, out dS'), ds16 = bf16(Wl^T dS'), dww/dbw/dbl +=;  th_dwl: dWl += sum dS' * raw scores. */
int ppf_th_scores(const void* qkv, const float* wl, const float* bl, float* sp, int B, int H, int N, int D, int NP, ppf_stream_t stream);
int ppf_th_dwl(const void* qkv, const float* ds_prime, float* dwl, int B, int H, int N, int D, int NP, ppf_stream_t stream);
int ppf_th_softmax_mix(float* sp, void* a16, float* hm, const float* ww, const float* bw, int B, int H, int N, int NP, int NPK,
                       ppf_stream_t stream);
int ppf_th_softmax_bwd(const float* prob, float* da, void* ds16, const float* ww, const float* wl, float* dww, float* dbw, float* dbl,
                       int B, int H, int N, int NP, int NPK, ppf_stream_t stream);
/* Fused talking-heads attention (cait:119-126; replaces th_scores + th_softmax_mix and, in backward, the dA product + th_softmax_bwd +
 * th_dwl: no (B,H,N,N) fp32 tensor is materialised).  ppf_th_fused_supported: 1 when (heads, tokens, width) is covered (K and V images
 * of all heads must fit the 160 KiB LDS: head_dim in {32,48,64}, heads in {2,4}, N <= 224), else 0 -- callers then use the kernels above.
 * ppf_th_fwd: a16 = bf16 proj_w(softmax(proj_l(scale q k^T))) [B][H][N][NPK], hm = its head mean [B][N][NP], rowmax / zinv [B][H][N] =
 * softmax statistics of the mixed logits (saved for backward); out (optional) = the attention output A V, bf16 [B*N][D] (cait:128).
 * ppf_th_bwd: ds16 = bf16 dS [B][H][N][NPK] from dout = dO [B*N][D]; partial (>= ppf_th_bwd_partial_floats floats) receives one row of
 * parameter-gradient sums per workgroup; ppf_th_param_reduce adds their fixed-order (bit-reproducible) totals to dww, dbw, dbl, dwl. */
int ppf_th_fused_supported(int H, int N, int D);
int ppf_th_fwd(const void* qkv, const float* wl, const float* bl, const float* ww, const float* bw, void* a16, float* hm, float* rowmax,
               float* zinv, void* out, int B, int H, int N, int D, int NP, int NPK, ppf_stream_t stream);
size_t ppf_th_bwd_partial_floats(int B, int H, int N);
int ppf_th_bwd(const void* qkv, const void* dout, const float* wl, const float* bl, const float* ww, const float* rowmax, const float* zinv,
               void* ds16, float* partial, int B, int H, int N, int D, int NPK, ppf_stream_t stream);
int ppf_th_param_reduce(const float* partial, int B, int H, int N, float* dww, float* dbw, float* dbl, float* dwl, ppf_stream_t stream);
/* the three per-head products behind ppf_th_bwd in one launch: dqkv [B*N][3D] bf16 <- dQ_h = scale dS_h K_h | dK_h = scale dS_h^T Q_h |
 * dV_h = A_h^T dO_h from ds16 / a16 [B][H][N][NPK], packed qkv and dout [B*N][D] (ppf_th_grads_supported: N <= 208, head_dim 32/48/64;
 * otherwise three ppf_gemm_bf16_batched calls) */
int ppf_th_grads_supported(int H, int N, int D);
int ppf_th_grads(const void* qkv, const void* dout, const void* ds16, const void* a16, void* dqkv, int B, int H, int N, int D, int NPK,
                 ppf_stream_t stream);
/* class attention: q [B][D] (cls rows, unscaled), k/v [B*N1][D] bf16; policy softmax WITHOUT the identity term (cait:58-59) */
int ppf_class_attn_fwd(const void* q, const void* k, const void* v, const float* policy, float* attn, float* zinv, float* rowmean,
                       void* out, int B, int H, int N1, int D, ppf_stream_t stream);
int ppf_class_attn_bwd(const void* q, const void* k, const void* v, const float* attn, const float* zinv, const void* dout, void* dq,
                       void* dk, void* dv, int B, int H, int N1, int D, ppf_stream_t stream);
/* out bf16 [rows][D] = a + b (+ cq[row / N1] where row % N1 == 0): input gradient of the class-attention q/k/v projections */
int ppf_merge3_cast(const float* a, const float* b, const float* cq, void* out, int rows, int D, int N1, ppf_stream_t stream);

/* ---- attention rollout + token reservation (deit:99-124, 223-234; cait:223-261, 328-339) ------------------------
 * hm [L][B][N][NP] head-mean attention of the rollout layers; kdrop = int(N*N*0.9), kdrop_init = int((N+1)*0.9).
 * lead = 1 (DeiT): row 0 of a_{L-1}..a_0, outputs skip the cls column; lead = 0 (CaiT): init_rows [n_init][B][N+1] are the
 * class-attention rows.  Outputs: cls_attn [B][N-lead], idx [B][k] int32 ascending (topk + sort), policy [B][N-lead+1].
 * thr_u32 (optional) [L][B]: the per (layer, sample) discard thresholds from ppf_rollout_threshold -- the exact order statistic
 * "int(N*N*0.9)-th smallest entry" (deit:108-112) does not depend on the chain, so it can be taken per layer as soon as that layer's
 * head-mean map exists (off the critical path); NULL: selected inside ppf_rollout. */
int ppf_rollout_threshold(const float* hm_layer, int B, int N, int NP, int kdrop, void* thr_out_u32, ppf_stream_t stream);
int ppf_rollout(const float* hm, int64_t layer_stride, int L, int B, int N, int NP, const float* init_rows, int n_init, int lead,
                int kdrop, int kdrop_init, float identity, int k, const void* thr_u32, float* cls_attn, int* idx, float* policy,
                ppf_stream_t stream);
/* topk(k) + ascending sort of indices on given scores [B][n] (protopformer.py:157-158, 273-274) */
int ppf_topk_sorted(const float* scores, int B, int n, int k, int* idx, ppf_stream_t stream);

/* ---- prototype layer (protopformer.py:201-247): squared L2 distances, log similarity, max-pool -------------------
 * sample b's token i is at tok + b*stride_b + (t0+i)*Dp (fp32); T tokens per sample (T == 1: global / cls branch).
 * act_kind 0: log((d+1)/(d+eps)), 1: -d.  Outputs act_max [B][P], argmax [B][P], optional dist_full/act_full [B][P][T]. */
int ppf_proto_fwd(const float* tok, int64_t stride_b, int t0, int T, const float* protos, int B, int P, int Dp, int act_kind,
                  float eps, float* act_max, int* argmax, float* dist_full, float* act_full, void* workspace, size_t workspace_bytes,
                  ppf_stream_t stream);
/* workspace: optional scratch of ppf_proto_fwd_workspace(P, Dp) bytes (any content, 16-byte aligned) -- the prototypes split ONCE per launch
 * into the bf16 piece planes the pooled kernel reads (NULL: every workgroup splits its own rows; same results to fp32 summation order). */
size_t ppf_proto_fwd_workspace(int P, int Dp);
int ppf_proto_bwd(const float* tok, int64_t stride_b, int t0, int T, const float* protos, int B, int P, int Dp, int act_kind,
                  float eps, const float* dist_full, int map_is_act, const float* g_full, const float* g_max, const int* argmax, float* dtok,
                  int64_t dstride_b, float* dprotos, void* zeroed_workspace, size_t workspace_bytes, ppf_stream_t stream);
/* map_is_act != 0: the [B][P][T] map passed as dist_full holds the ACTIVATIONS ppf_proto_fwd wrote (act_full), not the distances; the
 * derivative is then taken from them (a = log((d+1)/(d+eps)) => da/dd = -(e^a - 1)^2 / ((1 - eps) e^a), 0 at the clipped d == 0), so a
 * training forward writes one (B,P,T) map instead of two. */
/* workspace: ppf_proto_bwd_workspace() bytes = [bitmap of the non-zero dL/dd entries, B*T*ceil(P/32)*4 bytes rounded up to 256, present
 * and ZERO-FILLED by the caller when dtok != NULL][scratch of the prototype gradients when dprotos != NULL, any content].  A workspace
 * that only holds the bitmap is accepted: the prototype gradients then take the per-prototype gather kernel (same results to rounding). */
size_t ppf_proto_bwd_workspace(int B, int T, int P, int Dp, int want_dtok, int want_dprotos);
/* The same backward with the activation-map gradient in the block form get_PPC_loss produces (protopformer.py:259-288: only the ppc
 * prototypes of the sample's own class get a gradient): g_rows [B][ppc][T] = dL/d act_full[b][label[b]*ppc + k][t]; every other entry
 * of the (B,P,T) gradient is zero and is never materialised.  label_i64: int64 labels [B] (0 <= label*ppc <= P - ppc). */
int ppf_proto_bwd_rows(const float* tok, int64_t stride_b, int t0, int T, const float* protos, int B, int P, int Dp, int act_kind,
                       float eps, const float* dist_full, int map_is_act, const float* g_rows, const void* label_i64, int ppc, const float* g_max,
                       const int* argmax, float* dtok, int64_t dstride_b, float* dprotos, void* workspace, size_t workspace_bytes,
                       ppf_stream_t stream);
/* T == 1 (global branch, protopformer.py:295,311 and its autograd): dense form of the same backward -- two fp32 products instead of the
 * gather.  g [B][P] = upstream gradient of the activations, dist [B][P]; dtok rows are overwritten (when non-null), dprotos accumulated. */
size_t ppf_proto_bwd_single_workspace(int B, int P, int Dp);
int ppf_proto_bwd_single(const float* tok, int64_t stride_b, int t0, const float* protos, int B, int P, int Dp, int act_kind, float eps,
                         const float* dist, const float* g, float* dtok, int64_t dstride_b, float* dprotos, void* workspace,
                         size_t workspace_bytes, ppf_stream_t stream);

/* ---- losses and the frozen class-connection head ------------------------------------------------------------------
 * PPC loss (protopformer.py:249-288): loss[0] = cov term, loss[1] = mean term; gcov/gmean [B][ppc][T] analytic grads. */
int ppf_ppc_loss(const float* act, const int* idx, const void* label_i64, int B, int P, int T, int ppc, int side,
                 float cov_thresh, float mean_thresh, float* partial, float* gcov, float* gmean, float* loss, ppf_stream_t stream);
int ppf_ppc_loss_bwd(const float* gcov, const float* gmean, const float* up_cov, const float* up_mean, const void* label_i64,
                     float* g_full, int B, int P, int T, int ppc, ppf_stream_t stream);
/* nn.CrossEntropyLoss (main.py:390): mean loss + d/dlogits */
int ppf_cross_entropy(const float* logits, const void* label_i64, float* per_sample, float* dlogits, float* loss, int B, int C,
                      ppf_stream_t stream);
/* last_layer / last_layer_global (protopformer.py:126-131, 314-316): C = alpha * A B^T + beta * C, arbitrary strides */
int ppf_sgemm(const float* A, const float* Bm, float* C, int M, int N, int K, int64_t sam, int64_t sak, int64_t sbn, int64_t sbk,
              int ldc, float alpha, float beta, float* workspace, int64_t workspace_floats, ppf_stream_t stream);
/* Two such products with the same M in one launch + one ordered reduction: out_i = alpha_i A_i B_i^T (may be NULL) and
 * sum = c0 A0 B0^T + c1 A1 B1^T (may be NULL; N0 == N1): logits = g * logits_global + (1 - g) * logits_local (protopformer.py:314-316)
 * and the two input gradients of that layer. */
int64_t ppf_sgemm_pair_workspace(int M, int N0, int K0, int N1, int K1);
int ppf_sgemm_pair(const float* A0, const float* B0, float* out0, int N0, int K0, int64_t sam0, int64_t sak0, int64_t sbn0, int64_t sbk0,
                   int ldo0, float alpha0, const float* A1, const float* B1, float* out1, int N1, int K1, int64_t sam1, int64_t sak1,
                   int64_t sbn1, int64_t sbk1, int ldo1, float alpha1, float* sum, int ldsum, float c0, float c1, int M, float* workspace,
                   int64_t workspace_floats, ppf_stream_t stream);
int ppf_axpby(const float* x, const float* y, float* out, float a, float b, int64_t n, ppf_stream_t stream);
/* out = a x + b y + c z: loss = CE + ppc_cov_coe * cov + ppc_mean_coe * mean in one launch (tools/engine_proto.py:61-64) */
int ppf_axpbypcz(const float* x, const float* y, const float* z, float* out, float a, float b, float c, int64_t n, ppf_stream_t stream);

/* ---- streaming kernels -------------------------------------------------------------------------------------------- */
int ppf_cast_f32_bf16(const float* in, void* out, int64_t n, ppf_stream_t stream);
/* bf16 -> fp32 (exact; n % 8 == 0): the optional bf16 wire format of the gradient all-reduce (engine.GradSync payload='bf16';
 * replaces nothing in the reference -- main.py:369-371 exchanges fp32 through DistributedDataParallel) */
int ppf_cast_bf16_f32(const void* in, float* out, int64_t n, ppf_stream_t stream);
/* PatchEmbed unfold (timm PatchEmbed conv k=s=patch, deit:174): img [B][C][H][W] -> cols bf16 [B*gh*gw][C*p*p] */
int ppf_im2col_patch(const float* img, void* cols, int B, int C, int H, int W, int patch, ppf_stream_t stream);
/* cls token + position embedding (deit:176-178 lead=1; cait:307-309 lead=0) and its backward */
int ppf_assemble_tokens(const float* tok, const float* cls, const float* pos, float* x, int B, int Np, int D, int lead,
                        ppf_stream_t stream);
int ppf_assemble_tokens_bwd(const float* dx, void* dtok, float* dpos, float* dcls, int B, int Np, int D, int lead,
                            ppf_stream_t stream);
/* backward of the add-on Sigmoid (protopformer.py:113): dz = bf16(df*f*(1-f)), dbias += column sums.  partial (>=
 * ppf_sigmoid_bwd_blocks(rows)*cols floats) makes the column sums a fixed-order two-pass reduction; NULL = fp32 atomics. */
int ppf_sigmoid_bwd_blocks(int rows);
int ppf_sigmoid_bwd(const float* df, const float* f, void* dz, float* dbias, int rows, int cols, float* partial, size_t partial_bytes,
                    ppf_stream_t stream);
/* fused AdamW (+EMA, +bf16 re-cast) over flat buffers (tools/create_optimizer.py:92, engine_proto.py:80-81) */
int ppf_adamw_step(float* p, const float* g, float* m, float* v, float* ema, void* p16, int64_t n, int nseg,
                   const int64_t* seg_bounds, const float* seg_lr, const float* seg_wd, float beta1, float beta2, float eps,
                   int step, float ema_decay, float grad_scale, ppf_stream_t stream);
/* The same step with every step-dependent scalar in DEVICE memory: hyper[0..7] lr per segment, [8..15] weight decay per segment,
 * [16] 1-beta1^t, [17] sqrt(1-beta2^t), [18] gradient scale (1/world), [19] clip coefficient (ppf_clip_grad_scale; 1 = off).
 * Nothing step-dependent is baked into the launch, so a captured HIP graph of the train step can be replayed
 * (lr schedule tools/create_scheduler.py:20-32 keeps working: the host refreshes `hyper` before each replay). */
int ppf_adamw_step_dev(float* p, const float* g, float* m, float* v, float* ema, void* p16, int64_t n, int nseg,
                       const int64_t* seg_bounds, const float* hyper, float beta1, float beta2, float eps, float ema_decay,
                       ppf_stream_t stream);
/* ppf_adamw_step_dev with the reference's per-step non-finite-loss check (tools/engine_proto.py:66-70: print + sys.exit(1) in front of
 * optimizer.step()) evaluated on the device: `loss` = the step's loss scalar (device); if it is NaN / inf nothing is updated and
 * *nonfinite_flag = 1 (read by the host at its logging cadence: engine.train_one_epoch).  loss == NULL && flag == NULL: no check. */
int ppf_adamw_step_guarded(float* p, const float* g, float* m, float* v, float* ema, void* p16, int64_t n, int nseg,
                           const int64_t* seg_bounds, const float* hyper, float beta1, float beta2, float eps, float ema_decay,
                           const float* loss, int* nonfinite_flag, ppf_stream_t stream);
/* hyper[0..n) <- host_vals[0..n) (n <= 20), passed by value at enqueue time (safe when the host runs ahead of the device) */
int ppf_hyper_set(float* hyper, const float* host_vals, int n, ppf_stream_t stream);
/* clip_grad_norm_ (timm dispatch_clip_grad 'norm' via NativeScaler, engine_proto.py:74-76, main.py --clip-grad): hyper[19] =
 * min(1, max_norm / (pre_scale*||g||_2 + 1e-6)); partial = ppf_clip_grad_blocks() floats of scratch; norm_out optional. */
int ppf_clip_grad_blocks(void);
int ppf_clip_grad_scale(const float* g, int64_t n, float max_norm, float pre_scale, float* partial, float* hyper, float* norm_out,
                        ppf_stream_t stream);
/* timm DropPath factors floor(keep+U)/keep for all (slot, sample) pairs of one step (deit:71,79-80): out [nslot][B], keep [nslot]
 * (device).  Philox4x32-10 keyed by seed, counter = (state_u64[0], element); the launch advances state_u64[0] (device) by one. */
int ppf_droppath_scales(float* out, const float* keep, int nslot, int B, uint64_t seed, void* state_u64, ppf_stream_t stream);
/* token reservation plumbing (protopformer.py:156-162 gather of [cls, 1+idx...]): flat source rows of the reserved tokens,
 * row gather, and zero-filled row scatter (its backward). row_bytes % 16 == 0. */
int ppf_reserved_rows_map(const int* idx, int* rows, int B, int k, int N, ppf_stream_t stream);
int ppf_gather_rows(const void* src, const int* rows, void* dst, int nrows, int row_bytes, ppf_stream_t stream);
int ppf_scatter_rows(const void* src, const int* rows, void* dst, int nrows_src, int nrows_dst, int row_bytes, ppf_stream_t stream);
int ppf_memset_zero(void* ptr, size_t bytes, ppf_stream_t stream);
/* strided sub-matrix copy (bytes): dst[r][0..width) = src[r][0..width), rows rows, independent row pitches (torch.cat / slicing of the
 * class-attention stage, cait:314-316, kept on the library's launch path) */
int ppf_copy_2d(void* dst, int64_t dst_pitch, const void* src, int64_t src_pitch, int64_t width, int64_t rows, ppf_stream_t stream);
/* input pipeline finisher (tools/datasets.py:280-336: ToTensor + Normalize; timm RandomErasing 'pixel'): uint8 [B][H][W][3] frames
 * -> fp32 [B][3][H][W] = (x/255 - mean)/std; rects [B][4] = (y, x, h, w) erase rectangles filled with N(0,1) noise (h == 0: none;
 * rects == NULL: no erasing); mean3 / std3 are HOST pointers; state_u64 (device, optional) = step counter mixed into the noise key. */
int ppf_image_finish_u8(const void* in_u8_hwc, float* out_nchw, int B, int H, int W, const float* mean3, const float* std3,
                        const int* rects, uint64_t seed, const void* state_u64, ppf_stream_t stream);
/* out = x * (*scalar_dev): chain rule with a device-resident upstream scalar (autograd of nn.CrossEntropyLoss, main.py:390) */
int ppf_scale_by_scalar(const float* x, const float* scalar_dev, float* out, int64_t n, ppf_stream_t stream);

/* ---- fp32 verification path (forward only; PPNet.precise / PPF_PRECISE=1) ------------------------------------------------
 * Every operand and intermediate fp32, contractions through ppf_sgemm, exact-erf GELU: used by the parity tests to hold the whole
 * forward / loss to 1e-3 rel against the reference fixtures.  Same reference lines as the bf16 kernels above. */
int ppf_im2col_patch_f32(const float* img, float* cols, int B, int C, int H, int W, int patch, ppf_stream_t stream);
int ppf_layernorm_fwd_f32(const float* x, const int* row_map, const float* w, const float* b, float* y, int rows, int D, float eps,
                          ppf_stream_t stream);
/* in place on C[M][N]: kind 0 +bias | 1 gelu_erf(+bias) | 2 sigmoid(+bias) | 3 res + rowscale[m/rows_per_group]*colscale[n]*(C+bias) |
 * 4 relu(+bias) (the bottleneck add-on head, protopformer.py:90-107) */
int ppf_epilogue_f32(float* C, const float* bias, int kind, const float* res, const float* rowscale, int rows_per_group,
                     const float* colscale, int M, int N, ppf_stream_t stream);
/* deit:29-60 on fp32 qkv [B*N][3D]; headmean (optional) [B][N][NP] = mean over heads of the probabilities (deit:104) */
int ppf_attn_fwd_f32(const float* qkv, float* out, const float* policy, float* headmean, int NP, int B, int H, int N, int D,
                     int self_keep, int eps_n, ppf_stream_t stream);
/* cait:115-132 on fp32 qkv; headmean = mean over heads of the returned (post proj_w) attention (cait:328) */
int ppf_th_attn_fwd_f32(const float* qkv, const float* wl, const float* bl, const float* ww, const float* bw, float* out,
                        float* headmean, int NP, int B, int H, int N, int D, ppf_stream_t stream);
/* cait:50-90: q [B][D] cls rows, k / v [B*N1][D]; attn_mean [B][N1] = mean over heads of the probabilities */
int ppf_class_attn_fwd_f32(const float* q, const float* k, const float* v, const float* policy, float* attn_mean, float* out, int B,
                           int H, int N1, int D, ppf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PPF_HIP_H */
