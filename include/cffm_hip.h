/* cffm_hip.h -- C ABI of libcffm_hip.so: the MI355X-native (gfx950) CFFM hot path.
 *
 * The reference (GuoleiSun/VSS-CFFM) is pure Python; it has no FFI for this path.  The entry points
 * below are what a binding of the reference's hot-path *modules* needs, one per module / stage, with
 * plain device pointers and sizes (no torch types):
 *
 *   cffm_layer_forward / _backward   <->  BasicLayer3d3.forward            cffm_transformer.py:917-927
 *   cffm_block_forward / _backward   <->  CffmTransformerBlock3d3.forward  cffm_transformer.py:709-832
 *   cffm_ln_pool_fwd / _bwd          <->  CFFA: norm1 + pad + pool_layers / pool_layers_clips
 *                                                                           cffm_transformer.py:716-805
 *   cffm_bias_assemble / _scatter    <->  relative-position bias gathers    cffm_transformer.py:536-587
 *   cffm_attn_fwd / _bwd             <->  WindowAttention3d3.forward (CFM)  cffm_transformer.py:364-606
 *   cffm_linear_*                    <->  nn.Linear (qkv :374, proj :602, Mlp :10-26)
 *   cffm_residual_ln, cffm_bias_gelu, cffm_residual_out  <->  residual/norm2/Mlp glue  :823-824
 *   cffm_gtc_*                       <->  BasicLayer_cluster / WindowAttention_cluster (CFFM++)
 *                                                         pvt/swin_transformer_2d.py:1103-1148, :208-262
 *
 * Conventions: every pointer is a DEVICE pointer (fp32 unless stated) borrowed for the duration of
 * the call; `stream` is a hipStream_t (NULL = default stream); work is enqueued asynchronously on it;
 * functions return 0 on success, otherwise a negative code and cffm_last_error() describes it
 * (mirrors the reference's Python asserts: wrong T / shape is an error, not UB).  Single-threaded per
 * process (one process per GPU).  C = 256, 8 heads, window 7 are fixed as in cffm_head.py:74-95.
 */
#ifndef CFFM_HIP_H
#define CFFM_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define CFFM_ABI_VERSION 10

typedef struct cffm_geom {
    int B, H0, W0;      /* clips, unpadded 1/8-scale grid                                   */
    int Hp, Wp, gy, gx; /* padded to multiples of 7; windows per side                        */
    int nW, HW, RC;     /* windows per clip; H0*W0; token rows per clip = 64*nW (49 + 15)    */
} cffm_geom;

/* parameters of one CffmTransformerBlock3d3 (state_dict names in SURVEY.md Appendix C) */
typedef struct cffm_block_params {
    const float *norm1_w, *norm1_b;
    const float *pool_w[4], *pool_b[4]; /* pool_layers.0, pool_layers_clips.{0,1,2}: [49],[49],[9],[4] / [1] */
    const float *rpb_own;               /* attn.relative_position_bias_table [169,8]                         */
    const float *rpb_ring;              /* attn.relative_position_bias_table_to_neighbors [1,8,49,132]      */
    const float *rpb_pool[4];           /* ..._to_windows.0 [8,121], ..._to_windows_clips.{0,1,2} [8,169|121|81] */
    const float *qkv_w, *qkv_b;         /* [768,256], [768] */
    const float *proj_w, *proj_b;       /* [256,256], [256] */
    const float *norm2_w, *norm2_b;
    const float *fc1_w, *fc1_b;         /* [1024,256], [1024] */
    const float *fc2_w, *fc2_b;         /* [256,1024], [256]  */
} cffm_block_params;

/* same fields, writable: gradients (every field is fully written by cffm_block_backward) */
typedef struct cffm_block_grads {
    float *norm1_w, *norm1_b;
    float *pool_w[4], *pool_b[4];
    float *rpb_own, *rpb_ring, *rpb_pool[4];
    float *qkv_w, *qkv_b, *proj_w, *proj_b, *norm2_w, *norm2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} cffm_block_grads;

/* float offsets of the activations one block saves for its backward (inside its slice of `saved`) */
typedef struct cffm_block_ws {
    long mean1, rstd1, M, zall, qkv, bias, biasT, lse, ao, x1, mean2, rstd2, z2, hraw, act, x2, w_split, w_frag, ao_t, zall_t /* ABI 9 */, total;
} cffm_block_ws;

int cffm_abi_version(void);
const char* cffm_last_error(void);
int cffm_geom_init(cffm_geom* g, int B, int H0, int W0);
int cffm_block_ws_layout(const cffm_geom* g, cffm_block_ws* out);
long cffm_layer_saved_floats(const cffm_geom* g, int depth);  /* NHWC stack + depth * block_ws.total */
long cffm_layer_scratch_floats(const cffm_geom* g);

/* ---- optional per-stage HIP-event timing on the caller's stream (bench.py's live roofline numbers) ---- */
int cffm_profile_enable(long long stage_mask); /* bit i = stage i; 0 off; -1 all (perturbs: two event records per launch) */
int cffm_profile_sample_every(int period); /* time every period-th launch of an enabled stage only (1 = all; counted from the next cffm_profile_enable) */
int cffm_side_streams(int on); /* 0: parameter-gradient work stays on the caller's stream (per-kernel timing); returns the previous setting */
/* ABI 10: branches for the caller.  cffm_branch_begin(stream, i), i = 0..3: a library-owned stream ordered behind everything issued on `stream`
 * so far (or `stream` itself when side streams are off / unavailable); pass it as the `stream` argument of independent stage calls (the head's
 * per-scale embedding chains of cffm_head.py:102-119, weight vs input gradients of a classifier); cffm_branch_join(stream): `stream` continues
 * behind every branch.  Tensors a branch touches must outlive the join.  Stage calls that use library-owned scratch (cffm_linear_bwd_weight,
 * cffm_linear_bwd_weight_group, cffm_colsum) take it from a pool of the branch they are issued on, so they may run on different branches at
 * the same time; every other entry point with scratch of its own (the layer / block / gtc calls) belongs on the caller's stream. */
void* cffm_branch_begin(void* stream, int i);
int cffm_branch_join(void* stream);
/* the same in two steps, for callers that launch their own (longest) chain FIRST: under stream capture the first-launched dependant of a node keeps
 * the node's hardware queue.  cffm_branch_mark(stream): remember this point of `stream` (returns 1 when branches are available);
 * cffm_branch_take(stream, i): branch i ordered behind the marked point (or `stream` itself). */
int cffm_branch_mark(void* stream);
void* cffm_branch_take(void* stream, int i);
/* a DEFERRED branch: a library-owned stream of its own for work whose results `stream` needs only much later (the frame classifier's backward
 * of the CFFM heads, cffm_head.py:121, beside the whole CFFM layer's backward).  cffm_defer_begin(stream): that stream, ordered behind `stream`
 * (or `stream` itself); cffm_defer_join(stream): `stream` continues behind it (no-op when nothing is pending).  Nothing on `stream` may read the
 * deferred results before the join.  cffm_add_inplace: a += b (n floats, n % 4 == 0) on a stream. */
void* cffm_defer_begin(void* stream);
int cffm_defer_join(void* stream);
int cffm_add_inplace(float* a, const float* b, long n, void* stream);
int cffm_profile_stage_count(void);
int cffm_profile_null_pair(void* stream); /* stage "event_pair_null": two event records with nothing between (the interval's own cost) */
const char* cffm_profile_stage_name(int i);
int cffm_profile_collect(float* ms /*[stage_count]*/, int* calls /*[stage_count]*/); /* synchronises, sums, clears */
/* Stages enabled while the caller's stream is being CAPTURED into a HIP graph put their event pairs into the graph as
 * event-record nodes (re-recorded by every replay).  This reads, per stage, the intervals of the most recent replay;
 * the pairs stay valid for the life of the graph; reset != 0 forgets them (before capturing another graph). */
int cffm_profile_collect_graph(float* ms /*[stage_count]*/, int* calls /*[stage_count]*/, int reset);

/* ---- stage level (each is one kernel; used by the parity tests and by the block functions) ---- */
/* dst[n][c][r] = src[n][r][c] for n < batch (element strides src_bs / dst_bs between batches) */
int cffm_transpose(const float* src, float* dst, int batch, int rows, int cols, long src_bs, long dst_bs, void* stream);
int cffm_pool_matrix(const float* const pool_w[4], float* M /*[15*49]*/, void* stream);
int cffm_pool_matrix_bwd(const float* dM, float* const dpool_w[4], void* stream);
/* Process-wide: tells the library that EVERY gradient tensor handed to the block / layer backward sits in a slice of a flat buffer
 * padded to a multiple of 4 floats (vss_cffm_amd.ops lays them out so; a data-parallel exchange all-reduces that buffer whole), and
 * asks it to zero the padding itself: the backward of the pooling Linears -- owners of the only tensors whose length is not a multiple
 * of 4 (49 / 49 / 9 weights, four scalar biases) -- writes 3 zeros behind each of them.  Off by default (tensors of their own);
 * vss_cffm_amd.ops switches it on around its own layer-backward calls only.  The setting is PER CALLING THREAD (thread_local). */
void cffm_grad_slices_padded(int yes);
int cffm_ln_pool_fwd(const cffm_geom* g, const float* x_ref, long ref_bs, const float* x_tgt, long tgt_bs,
                     const float* gamma, const float* beta, const float* M, const float* const pool_b[4],
                     float* zall, float* mean, float* rstd, void* stream);
int cffm_ln_pool_bwd(const cffm_geom* g, const float* x_ref, long ref_bs, const float* x_tgt, long tgt_bs,
                     const float* gamma, const float* beta, const float* M, const float* mean, const float* rstd,
                     const float* dzall, const float* dres /* [B*HW,256] added to the target-frame grad, may be NULL */,
                     float* dx_ref, long dref_bs, int accum_ref, float* dx_tgt, long dtgt_bs,
                     float* dgamma, float* dbeta, float* dM, float* const dpool_b[4], void* stream);
/* the dense additive bias [8 heads][64 queries][304 keys] (cffm_transformer.py:536-587 gathers it from six tables) in two
 * layouts, each may be NULL: bias = query-major fp32 [8,64,304] (checks); biasH = what cffm_attn_fwd / cffm_attn_bwd read: f16
 * B-operand fragments [8 heads][4 waves][10 key-tile pairs][64 lanes][8], entry (h, wave, p, lane = 16 g + j, e) = bias(h, query
 * 16 wave + j, key 16 (2 p + (g >> 1)) + 8 (g & 1) + e), keys 304..319 = 0 -- the kernels add the bias on the matrix pipe
 * (S^T tile t = K Q^T + Sel_(t & 1) * B) from one contiguous 1 KiB load per (wave, tile pair).  (8*4*10*512 halfs = 320 KB; ABI 6:
 * tile pairs, ABI <= 5 kept one tile per fragment.) */
int cffm_bias_assemble(const float* own, const float* ring, const float* const pool[4], float* bias, void* biasH, void* stream);
int cffm_bias_scatter(const float* dbiasT, float* down, float* dring, float* const dpool[4], void* stream);
/* qkv16 [B*RC,768] f16 = zall w^T + b with the q third times 32^-0.5 (cffm_linear_qkv_fwd) */
int cffm_linear_qkv_fwd(const float* zall, const float* w /*[768,256]*/, const float* b /*[768]*/, void* qkv16, long M,
                        void* stream);
int cffm_attn_fwd(const cffm_geom* g, const void* qkv16, const int* key_src /*[nW,304]*/, const int* q_dst /*[nW,49]*/,
                  const void* biasH /* f16 fragments, see cffm_bias_assemble */, float* ao /*[B*HW,256]*/, float* lse /*[B*nW*8,64]*/,
                  void* stream);
/* inv_ptr [RC+1] / inv_idx: CSR inverse of key_src (token row -> the window*304+slot pairs reading it);
 * dkv_part: scratch of B*nW*(304*256 + 8) floats: the per-window dK/dV rows the gather pass sums, kept as f16 [B*nW*304][512]
 * (a row = 8 heads x (K 32 | V 32) since ABI 6) in units of a per-(window, head) power-of-two scale, followed by those scales.
 * Replaces the backward of WindowAttention3d3.forward (cffm_transformer.py:364-606: autograd through the gather / roll / unfold /
 * softmax chain); one fused key-split kernel + the bias-tile sum + the dK / dV gather (csrc/cfm_attn_kernels.h). */
int cffm_attn_bwd(const cffm_geom* g, const void* qkv16, const int* key_src, const int* q_dst,
                  const int* inv_ptr, const int* inv_idx, const void* biasH, const float* ao,
                  const float* dao, const float* lse, float* dqkv /*[B*RC,768] fp32: d(zall w^T), overwritten*/,
                  float* dbiasT /*[8,304,64], overwritten*/, float* dkv_part, void* stream);
/* y[M,N] = x[M,K] w[N,K]^T ;  dx[M,K] = dy[M,N] w[N,K] ;  dw[N,K] = dy[M,N]^T x[M,K]   (row-major, no bias) */
int cffm_linear_fwd(const float* x, const float* w, float* y, long M, int N, int K, void* stream);
/* y[M,N] = x[M,K] w[N,K]^T + b[N]   (a 1x1 convolution on channels-last token rows: the head's classifiers, cffm_head.py:121,147) */
int cffm_linear_bias_fwd(const float* x, const float* w, const float* b, float* y, long M, int N, int K, void* stream);
int cffm_linear_bwd_input(const float* dy, const float* w, float* dx, long M, int N, int K, void* stream);
int cffm_linear_bwd_weight(const float* dy, const float* x, float* dw, long M, int N, int K, void* stream);
/* n <= 4 independent weight gradients (dw_i[N_i,K_i] = dy_i[M_i,N_i]^T x_i[M_i,K_i]) in one launch: the four Linear layers of a
 * block (qkv / proj / fc1 / fc2 .weight.grad, cffm_transformer.py:374, :381, :18-19), none of which feeds the backward chain */
typedef struct { const float* dy; const float* x; float* dw; long M; int N; int K; } cffm_wgrad;
/* Weight gradient with both operands in split-4 storage (16 bytes = {bf16 hi x4 | bf16 lo x4} of four consecutive floats of a row: what the
 * block's producers write for GEMM-only tensors): tiles go global -> LDS by DMA, no staging arithmetic (csrc/dw_kernels.h).  N, K multiples
 * of 128.  cffm_split4 makes such a copy of a plain fp32 array (n floats, n % 4 == 0). */
int cffm_split4(const float* src, float* dst, long n, void* stream);
int cffm_linear_bwd_weight_split(const float* dy_s, const float* x_s, float* dw, long M, int N, int K, void* stream);
int cffm_linear_bwd_weight_split_group(const cffm_wgrad* problems, int n /* <= 3 */, void* stream);   /* one launch, one common k-slice length */
int cffm_linear_bwd_weight_group(const cffm_wgrad* problems /* host */, int n, void* stream);
/* ABI 9: weight gradient with both operands in "T-frag" storage (csrc/dws_kernels.h): MFMA fragments along the contraction -- unit (ks, jt, h)
 * = 64 lanes x 16 B at 16-byte word ((ks * C/16 + jt) * 2 + h) * 64, lane (l15, g) holding x[32 ks + 8 g + e][16 jt + l15], e = 0..7, as bf16
 * hi (h = 0) / lo (h = 1), rows past M zero -- streamed straight into registers: no LDS staging, no barrier in the contraction loop.  The
 * block's row-panel kernels leave z2 / act / dh / dx1 / ao / dout (and the q|k|v panel GEMMs zall / dqkv) in this order for the block's four
 * weight gradients (qkv / proj / fc1 / fc2 .weight.grad, cffm_transformer.py:374, :381, :18-19); cffm_tfrag_pack makes such a copy of a plain
 * row-major fp32 array (dst: cffm_tfrag_floats(R, C) floats).  N a multiple of 64, K of 128. */
long cffm_tfrag_floats(long R, int C);
/* Which form the block's four weight gradients take (process-wide; returns the previous setting, on < 0 only asks): 1 = the streaming kernel on
 * T-frag operands the row-panel kernels leave behind, 0 = the LDS-staged grouped kernel on split-4 / fp32 operands.  A block forward stores its
 * operands in the form the backward will read, so the setting must not change between a forward and its backward. */
int cffm_dw_stream(int on);
int cffm_tfrag_pack(const float* x, float* dst, long R, int C /* % 16 == 0 */, void* stream);
int cffm_linear_bwd_weight_tfrag(const float* dy_t, const float* x_t, float* dw, long M, int N, int K, void* stream);
int cffm_linear_bwd_weight_tfrag_group(const cffm_wgrad* problems /* host; dy / x in T-frag storage */, int n /* <= 4 */, void* stream);
/* fused Mlp halves: hraw = x w^T (raw, kept for backward), act = gelu(hraw + b)  |  out = res + x w^T + b */
int cffm_linear_gelu_fwd(const float* x, const float* w, const float* b, float* hraw, float* act, long M, int N, int K,
                         void* stream);
int cffm_linear_residual_fwd(const float* x, const float* w, const float* b, const float* res, float* out, long M, int N,
                             int K, void* stream);
/* ---- fused row-panel stages (round 3): one launch per direction for everything between the attention output and the block
 * output -- proj (cffm_transformer.py:602), residual (:823), norm2 + Mlp (fc1, exact GELU, fc2: :10-26) + residual (:824).
 * Weights are passed in MFMA-fragment order (cffm_panel_pack_weight: form 0 = forward y = x W^T, form 1 = input gradient
 * dx = dy W; N*K floats each; inside a block k_param_prep makes these copies).  z2s / acts / dhs are written in "split-4"
 * storage ({bf16 hi x4, bf16 lo x4} per 4 floats: the operand format of the weight-gradient GEMMs); acts may be NULL (not
 * stored: inside a block the fc2 weight gradient re-applies bias + GELU to hraw while it stages its tiles).
 *   forward : x1 = xt + ao Wp^T + bp; z2 = LN(x1; g2, be2); hraw = z2 W1^T; act = gelu(hraw + b1); x2 = x1 + act W2^T + b2
 *   backward: dh = (dout W2) * gelu'(hraw + b1); dx1 = dout + LN'(dh W1); dao = dx1 Wp; dg2, dbe2, db1, db2 = colsum(dout),
 *             dbp = colsum(dx1) (every one fully written)                                                                   */
int cffm_panel_pack_weight(const float* w, int N, int K, int form, float* w_frag, void* stream);
long cffm_mlp_records(long NP);
int cffm_mlp_fwd(const float* ao, const float* xt, long xt_bs, int rows_per_batch, const float* wp_f, const float* w1_f,
                 const float* w2_f, const float* bp, const float* b1, const float* b2, const float* g2, const float* be2,
                 float* x1, float* z2s, float* mean2, float* rstd2, float* hraw, float* acts, float* x2, long NP, void* stream);
int cffm_mlp_bwd(const float* dout, const float* hraw, const float* b1, const float* x1, const float* mean2,
                 const float* rstd2, const float* g2, const float* w2_n, const float* w1_n, const float* wp_n, float* dhs,
                 float* dx1, float* dao, float* dg2, float* dbe2, float* db1, float* db2, float* dbp, long NP, void* stream);
/* ABI 9: the same two kernels also leaving T-frag copies (cffm_tfrag_floats(NP, 256 | 1024) floats; see cffm_linear_bwd_weight_tfrag) of the
 * operands of the block's weight gradients: ao, z2, act (forward), dout, dh, dx1 (backward).  Every split-4 / T-frag output may be NULL. */
int cffm_mlp_fwd_tfrag(const float* ao, const float* xt, long xt_bs, int rows_per_batch, const float* wp_f, const float* w1_f,
                       const float* w2_f, const float* bp, const float* b1, const float* b2, const float* g2, const float* be2,
                       float* x1, float* z2s, float* mean2, float* rstd2, float* hraw, float* acts, float* x2, float* ao_t, float* z2_t,
                       float* act_t, long NP, void* stream);
int cffm_mlp_bwd_tfrag(const float* dout, const float* hraw, const float* b1, const float* x1, const float* mean2,
                       const float* rstd2, const float* g2, const float* w2_n, const float* w1_n, const float* wp_n, float* dhs,
                       float* dx1, float* dao, float* dg2, float* dbe2, float* db1, float* db2, float* dbp, float* dout_t, float* dh_t,
                       float* dx1_t, long NP, void* stream);
int cffm_colsum(const float* a, long rows, int cols /* multiple of 4 */, float* out /* overwritten */, void* stream);
int cffm_residual_ln(const float* xt, long xt_bs, int rows_per_batch, const float* yraw, const float* bproj,
                     const float* gamma, const float* beta, float* x1, float* z2, float* mean, float* rstd,
                     long nrows, void* stream);
/* dx1 = (dres or 0) + LNbwd(dz2); dgamma/dbeta are overwritten (zero_grads != 0) or accumulated into; the column
 * sums of dres and of dx1 (the bias gradients of the Linear layers on either side) come for free: pass NULL to skip */
int cffm_ln_bwd_residual(const float* x1, const float* mean, const float* rstd, const float* gamma, const float* dz2,
                         const float* dres /* may be NULL */, float* dx1, float* dgamma, float* dbeta, long nrows,
                         int zero_grads, float* dres_colsum /* [256] or NULL */, float* dx1_colsum /* [256] or NULL */,
                         void* stream);
int cffm_bias_gelu(const float* hraw, const float* b1, float* act, long rows, int cols, void* stream);
int cffm_gelu_bwd(const float* hraw, const float* b1, float* dact_inout, long rows, int cols /* 1024 */,
                  float* db1 /* [1024] column sums of the result, or NULL */, void* stream);
int cffm_residual_out(const float* x1, const float* oraw, const float* b2, float* out, long rows, void* stream);

/* ---- CFFM++ global temporal context (WindowAttention_cluster, pvt/swin_transformer_2d.py:208-262) ---- */
int cffm_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* z, float* mean, float* rstd,
                       long nrows, void* stream);
/* q_raw [B*T,256] = LN(x) Wq^T without bias, kv_raw [B*K,512] = LN(centers) Wkv^T without bias; softmax over the K
 * prototypes per (token, head); o [B*T,256]; lse [B*T,8].  K <= 128: on the matrix pipe (three-pass bf16 split of every product: the
 * 1e-4 tolerance of the fp32 form is kept), the head's Kc / Vc packed once per call as MFMA fragments; above: fp32 on the VALU.  The
 * backward's `o` is not read by the matrix-pipe form (D = sum_keys p dp comes out of the same products as dp). */
int cffm_gtc_attn_fwd(const float* q_raw, const float* q_b, const float* kv_raw, const float* kv_b, float* o, float* lse,
                      int B, int T, int K, void* stream);
int cffm_gtc_attn_bwd(const float* q_raw, const float* q_b, const float* kv_raw, const float* kv_b, const float* o,
                      const float* dout, const float* lse, float* dq_raw, float* dkv /* overwritten */, int B, int T, int K,
                      void* stream);

/* The whole block of the CFFM++ prototype layer in one call per direction (round 5, ABI 7; replaces the ~33 stage launches a caller
 * had to sequence): SwinTransformerBlock_cluster.forward, pvt/swin_transformer_2d.py:605-665, shift 0, with
 * WindowAttention_cluster.forward :208-262 inside (only_use_cluster_center_as_context, :216: no in-window keys, no position bias, no mask
 * -- window partition / padding are numerically no-ops, SURVEY.md A "GTC is window-independent").
 *   x [B,T,256] tokens, centers [B,K,256] prototypes (1 <= K <= 256), out [B,T,256]; `ws`: cffm_gtc_ws_floats(B, T, K) floats, written
 *   by the forward and read (and extended) by the backward of the same call pair.
 * Parameter order = the reference's state_dict of `decoder_swin.blocks.0`: norm1, attn.qkv (only its first 256 rows / entries -- the q
 * third -- are used and receive a gradient; the rest of the gradient is written as zeros), attn.qkv_cluster, attn.proj_cluster, norm2,
 * mlp.fc1, mlp.fc2.  (attn.proj and attn.relative_position_bias_table exist in the reference module but receive no gradient: not here.) */
typedef struct {
    const float *norm1_w, *norm1_b, *qkv_w /* [768,256] */, *qkv_b /* [768] */, *kv_w /* [512,256] */, *kv_b /* [512] */, *proj_w, *proj_b,
        *norm2_w, *norm2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} cffm_gtc_params;
typedef struct {
    float *norm1_w, *norm1_b, *qkv_w, *qkv_b, *kv_w, *kv_b, *proj_w, *proj_b, *norm2_w, *norm2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} cffm_gtc_grads;
long cffm_gtc_ws_floats(int B, int T, int K);
int cffm_gtc_block_forward(const cffm_gtc_params* p, const float* x, const float* centers, float* out, float* ws, int B, int T, int K,
                           void* stream);
int cffm_gtc_block_backward(const cffm_gtc_params* p, const cffm_gtc_grads* g, const float* x, const float* centers, const float* dout,
                            float* dx, float* dcenters, float* ws, int B, int T, int K, void* stream);

/* ABI 10: the composed embedding weights of ops.segformer_fuse (cffm_head.py:102-119 without the 1024-channel concat): mats[i] [e][C_i] =
 * Wf_i W_i (Wf_i = input-channel block k - 1 - i of linear_fuse.conv.weight [e][k e], W_i = linear_c{i+1}.proj.weight), d [e] = sum_i Wf_i b_i;
 * and the gradients of the nine tensors from dmats / dd.  The k products run on branches of their own (cffm_branch_begin). */
int cffm_fuse_compose_fwd(const float* fuse_w, const float* const* lin_w /* host [k] */, const float* const* lin_b /* host [k] */,
                          const int* C_in /* host [k] */, int k /* <= 4 */, int e /* 256 */, float* const* mats /* host [k] */, float* d, void* stream);
int cffm_fuse_compose_bwd(const float* fuse_w, const float* const* lin_w, const float* const* lin_b, const int* C_in, int k, int e,
                          const float* const* dmats /* host [k] */, const float* dd, float* dfuse_w /* [e][k e] */, float* const* dlin_w /* host [k] */,
                          float* const* dlin_b /* host [k] */, void* stream);

/* ---- SegFormer embedding in front of the hot path, without the 1024-channel concat ----
 * Replaces cffm_head.py:102-119 (4 x `MLP` embed, 3 x bilinear resize to the 1/4 map, torch.cat, 1x1 `linear_fuse.conv`):
 * conv(cat_i up_i(W_i c_i + b_i)) = sum_i up_i((Wf_i W_i) c_i) + sum_i Wf_i b_i.  The host embeds every scale once at its own
 * resolution with the composed matrix (cffm_linear_fwd on token rows) and calls this for the full-resolution pass:
 *   y [N*H*W,256] (the 1/4-scale embedding on entry) += d[256] + sum_m bilinear(z_m [N*h_m*w_m,256] -> H x W)
 * with F.interpolate(mode='bilinear', align_corners=False) taps; nmaps <= 3, resize factors <= 16 per dimension. */
int cffm_segfuse_fwd(float* y, const float* d, const float* const z[3], const int h[3], const int w[3], int nmaps, int N,
                     int H, int W, void* stream);
/* the adjoint of the three resizes: dz_m [N*h_m*w_m,256] = up_m^T g, g [N*H*W,256] (gather form, deterministic) */
int cffm_segfuse_bwd(const float* g, float* const dz[3], const int h[3], const int w[3], int nmaps, int N, int H, int W,
                     void* stream);

/* ---- the head's training loss without full-resolution logits ----
 * Replaces decode_head.py:744-835's resize(seg_logit, size=label size, 'bilinear', align_corners=False) followed by
 * F.cross_entropy(reduction='none', ignore_index) (losses/cross_entropy_loss.py:9-40) and `accuracy` (losses/accuracy.py:4):
 * logits [M,K,h,w] fp32, labels [M,H,W] int64 (ignore_index, and anything outside [0,K), contributes 0), H <= 8h, W <= 8w, K <= 256.
 * fwd: lse [M,H,W] (log-sum-exp of the interpolated logits, kept for bwd); part [cffm_upce_blocks(...)][2]: per-workgroup
 *      sums of the per-pixel losses and of the pixels whose arg-max is the label (the caller adds them up and divides).
 * bwd: dlogits [M,K,h,w] = scale * (*gscale, device scalar, or 1 when NULL) * d(sum of per-pixel losses)/dlogits. */
long cffm_upce_blocks(int M, int H, int W);
int cffm_upce_fwd(const float* logits, const long long* labels, float* lse, float* part, int M, int K, int h, int w, int H, int W,
                  int ignore_index, void* stream);
int cffm_upce_bwd(const float* logits, const long long* labels, const float* lse, const float* gscale, float scale,
                  float* dlogits, int M, int K, int h, int w, int H, int W, int ignore_index, void* stream);
/* The same with the caller's layout of the logits / their gradient (the heads keep them as token rows straight out of the classifier
 * GEMMs, the T frame maps and the clip-level map of a clip in one buffer): element (map m, class k, cell (r, c)) at
 *   (m / inner) * ms_outer + (m % inner) * ms_inner + k * ks + (r * w + c) * ps      (floats; one of ks, ps is 1)
 * label_idx[M] (device int32, NULL = identity): the label map a logits map is judged on (the clip-level map: the last frame's);
 * map_scale[M] (device fp32, NULL = 1): per-map factor on the gradient (0.5 / pixels for frame maps, 1 / pixels for clip maps:
 * decode_head.py:805-835); part records are ordered map-major (cffm_upce_blocks(M,H,W) / M per map), so the caller weights them. */
int cffm_upce_maps_fwd(const float* logits, const long long* labels, const int* label_idx, float* lse, float* part, int M, int K, int h,
                       int w, int H, int W, int ignore_index, int inner, long ms_outer, long ms_inner, int ks, int ps, void* stream);
/* ABI 10: the two loss scalars from those records on the device: out[0] = sum_m wl[m] * sum(loss records of map m), out[1] = sum_m wh[m] *
 * sum(hit records of map m), accumulated in double in a fixed order (wl, wh: M doubles each, device) -- decode_head.py:805-835's weighting. */
int cffm_upce_maps_finalize(const float* part, int M, long per, const double* wl, const double* wh, float* out /* [2] */, void* stream);
int cffm_upce_maps_bwd(const float* logits, const long long* labels, const int* label_idx, const float* lse, const float* gscale,
                       const float* map_scale, float scale, float* dlogits, int M, int K, int h, int w, int H, int W, int ignore_index,
                       int inner, long ms_outer, long ms_inner, int ks, int ps, void* stream);

/* ---- evaluation counts ----
 * mmseg/core/evaluation/metrics.py:62-119 `intersect_and_union` for one prediction / label map pair of n pixels (both int64 on
 * the device): counts [3][num_classes] int64 = per-class pixels of (prediction == label) | prediction | label, over the pixels
 * whose label is not ignore_index, ACCUMULATED into what is there (zero it first; `total_intersect_and_union` = call per image);
 * union = prediction + label - intersect.  Exact integers; num_classes <= 1024; at most 2^32 pixels per call. */
int cffm_seg_counts(const long long* pred, const long long* label, long n, int num_classes, int ignore_index,
                    int reduce_zero_label, long long* counts, void* stream);

/* video consistency VC_n, VC_perclip.py:62-78 `get_common`: gt / pred [F, npix] int64 label maps of one video; for every start
 * frame i < F - n: counts[i][0] += pixels whose label is constant over frames i..i+n-1 in BOTH gt and pred, counts[i][1] += pixels
 * constant in gt (acc_i = counts[i][0] / counts[i][1]); counts int64 [F - n][2], zero it first. */
int cffm_vc_counts(const long long* gt, const long long* pred, int F, long npix, int n, long long* counts, void* stream);

/* ---- block / layer level ---- */
/* x_ref: NHWC frames 0..2 [B,3,HW,256] (batch stride ref_bs), x_tgt NHWC target [B,HW,256] (stride tgt_bs);
 * writes the block's saved activations into `ws` (layout: cffm_block_ws_layout; ws[x2] is the output). */
int cffm_block_forward(const cffm_geom* g, const cffm_block_params* p, const float* x_ref, long ref_bs,
                       const float* x_tgt, long tgt_bs, const int* key_src, const int* q_dst, float* ws,
                       float* scratch, void* stream);
int cffm_block_backward(const cffm_geom* g, const cffm_block_params* p, const cffm_block_grads* gr,
                        const float* x_ref, long ref_bs, const float* x_tgt, long tgt_bs, const int* key_src,
                        const int* q_dst, const int* inv_ptr, const int* inv_idx, const float* ws,
                        const float* dout /*[B*HW,256]*/,
                        float* dx_ref, long dref_bs, int accum_ref, float* dx_tgt, long dtgt_bs,
                        float* scratch, void* stream);
/* x [B,4,256,H0,W0] -> y_tgt [B,256,H0,W0] (frames 0..2 of the reference's output equal the input) */
int cffm_layer_forward(const cffm_geom* g, int depth, const cffm_block_params* params, const float* x_nchw,
                       float* y_tgt_nchw, const int* key_src, const int* q_dst, float* saved, float* scratch,
                       void* stream);
/* dy_tgt [B,256,H0,W0] -> dx [B,4,256,H0,W0] (gradient through the hot path only; the pass-through of
 * frames 0..2 is the caller's torch.cat) and every parameter gradient of every block */
int cffm_layer_backward(const cffm_geom* g, int depth, const cffm_block_params* params, const cffm_block_grads* grads,
                        const float* dy_tgt_nchw, long dy_bs /* elements between clips: 256*H0*W0 when dy is dense, 4x that
                        when it is the last-frame slice of a [B,4,256,H0,W0] gradient */, float* dx_nchw, const int* key_src, const int* q_dst,
                        const int* inv_ptr, const int* inv_idx, const float* saved, float* scratch, void* stream);
/* The same in pieces: blocks first_block, first_block - 1, ..., last_block (depth - 1 >= first >= last >= 0), issued in order on
 * one stream; the piece with first == depth - 1 starts from dy, the piece with last == 0 writes dx.  Lets a data-parallel
 * trainer start the gradient all-reduce of block i while block i - 1 is still running (mmseg/apis/train.py:57-65). */
int cffm_layer_backward_range(const cffm_geom* g, int depth, const cffm_block_params* params, const cffm_block_grads* grads,
                        const float* dy_tgt_nchw, long dy_bs /* elements between clips: 256*H0*W0 when dy is dense, 4x that
                        when it is the last-frame slice of a [B,4,256,H0,W0] gradient */, float* dx_nchw, const int* key_src, const int* q_dst,
                        const int* inv_ptr, const int* inv_idx, const float* saved, float* scratch, int first_block, int last_block, void* stream);
/* The same pair on the reference's whole output (BasicLayer3d3.forward returns [B,4,C,H,W] whose frames 0..2 ARE the input frames,
 * cffm_transformer.py:826,917-927): forward writes y_full = [x[:, :3] | new target frame] (the copy runs on the library's side
 * stream under the blocks), backward takes the upstream gradient of that whole tensor and adds its pass-through frames into dx in
 * the final layout pass -- no torch.cat, no zero-fill / add of the pass-through gradient around the call. */
int cffm_layer_forward_full(const cffm_geom* g, int depth, const cffm_block_params* params, const float* x_nchw,
                            float* y_full_nchw, const int* key_src, const int* q_dst, float* saved, float* scratch,
                            void* stream);
int cffm_layer_backward_full(const cffm_geom* g, int depth, const cffm_block_params* params, const cffm_block_grads* grads,
                             const float* dy_full_nchw, float* dx_nchw, const int* key_src, const int* q_dst, const int* inv_ptr,
                             const int* inv_idx, const float* saved, float* scratch, int first_block, int last_block, void* stream);

/* The layer on token rows (channels-last) on both sides -- what the heads call (their neighbours work on rows too):
 * x_rows [B,4,HW,256] -> y_rows [B,HW,256]; no layout transposes, `x_rows` itself is the stack the blocks read and must be
 * handed to the backward again; dy_rows [B,HW,256] -> dx_rows [B,4,HW,256]. */
int cffm_layer_forward_rows(const cffm_geom* g, int depth, const cffm_block_params* params, const float* x_rows, float* y_rows,
                            const int* key_src, const int* q_dst, float* saved, float* scratch, void* stream);
int cffm_layer_backward_rows(const cffm_geom* g, int depth, const cffm_block_params* params, const cffm_block_grads* grads,
                             const float* x_rows, const float* dy_rows, float* dx_rows, const int* key_src, const int* q_dst,
                             const int* inv_ptr, const int* inv_idx, const float* saved, float* scratch, void* stream);

/* ---- `linear_fuse`'s BatchNorm + ReLU and the 1/4 -> 1/8 resize that builds the clip stack (cffm_head.py:119, :131-135), on token
 * rows [pixels,256].  With even H, W the bilinear 1/2 resize (align_corners=False) is exactly a 2x2 average.
 *   cffm_colstats:          part[cffm_colstats_records(rows)][512] = per-workgroup column sums | sums of squares of y [rows,256]
 *                           (the caller adds the records -- in fp64 -- and, under SyncBN, all-reduces them)
 *   cffm_bn_relu_pool_fwd:  fused = max(y*scale + shift, 0) [N*H*W,256]; stack = 2x2 average of fused [N*(H/2)*(W/2),256] (or NULL)
 *   cffm_bn_relu_pool_bwd1: g = [y*scale+shift > 0] * (dfused + dstack(parent)/4) (g may alias dfused; either gradient may be
 *                           NULL); part[cffm_bn_relu_pool_records(N,H,W)][512] = sums of g | g*xhat, xhat = y*xs + xo
 *   cffm_bn_bwd2:           g <- c1 * (g - mg - xhat*mgx)   (c1 = gamma*rstd, mg / mgx = per-channel means of g / g*xhat)
 *   mask (fwd / bwd1, may be NULL): Dropout2d in front of `linear_pred` (cffm_head.py:120) folded in: [N,256] factors 0 or 1/(1-p) per
 *                           (frame, channel), applied to `fused` (not to `stack`) and to dfused
 *   cffm_bn_finalize_fwd:   the per-channel arithmetic between the passes in one launch: part (NULL = eval mode, running statistics)
 *                           -> coef[4][256] = scale | shift | rstd | -mean*rstd (fp64 inside); running buffers updated as torch does
 *   cffm_bn_finalize_bwd:   part -> out[5][256] = dbias | dweight | mean g | mean g*xhat (0 in eval mode) | gamma*rstd */
long cffm_colstats_records(long rows);
int cffm_colstats(const float* y, long rows, float* part, void* stream);
long cffm_bn_relu_pool_records(int N, int H, int W);
int cffm_bn_relu_pool_fwd(const float* y, const float* scale, const float* shift, const float* mask, float* fused, float* stack, int N,
                          int H, int W, void* stream);
int cffm_bn_relu_pool_bwd1(const float* y, const float* scale, const float* shift, const float* xs, const float* xo, const float* mask,
                           const float* dfused, const float* dstack, float* g, float* part, int N, int H, int W, void* stream);
int cffm_bn_bwd2(float* g, const float* y, const float* xs, const float* xo, const float* c1, const float* mg, const float* mgx, long rows,
                 void* stream);
int cffm_bn_finalize_fwd(const float* part, long nrec, double count, const float* weight, const float* bias, float* running_mean,
                         float* running_var, float momentum, float eps, float* coef, void* stream);
int cffm_bn_finalize_bwd(const float* part, long nrec, double count, const float* weight, const float* xs, int training, float* out,
                         void* stream);
/* Bilinear resize (align_corners = False, ATen's tap rule and nesting) of token rows [N][h*w][C] -> [N][H*W][C], maps `*_map_stride`
 * floats apart (so a map may sit inside a larger buffer: the clip-level logits inside the [B, T+1, h, w, K] logits rows,
 * cffm_head.py:149); C % 4 == 0.  bwd = the adjoint in gather form (deterministic), reading the gradient where it lies. */
int cffm_rows_resize_fwd(const float* src, long src_map_stride, float* dst, long dst_map_stride, int N, int h, int w, int H, int W, int C,
                         void* stream);
int cffm_rows_resize_bwd(const float* ddst, long ddst_map_stride, float* dsrc, long dsrc_map_stride, int N, int h, int w, int H, int W, int C,
                         void* stream);

/* ---- clip data path after decoding: the `*_clips` transforms of local_configs/_base_/datasets/vspw_repeat2.py:8-19
 * -- LoadAnnotations(reduce_zero_label), RandomCrop_clips (transforms.py:1524), RandomFlip_clips (:852), Normalize_clips (:1260),
 * Pad_clips (:990), DefaultFormatBundle_clips (formating.py:261) -- applied to a whole clip in one pass.
 * frames [T,H,W,3] uint8 (BGR as decoded), labels [T,H,W] uint8 or NULL (then out_lab must be NULL);
 * crop box rows y1..y1+ch-1, columns x1..x1+cw-1 (one box for all frames), then an optional horizontal flip, BGR->RGB (to_rgb),
 * (v - mean) * (1/std) in float32, padding of the bottom / right to Ho x Wo with pad_val (image, after normalisation) and
 * seg_pad_val (labels); out_img [T,3,Ho,Wo] float32, out_lab [T,1,Ho,Wo] int64.  The random draws stay on the host
 * (vss_cffm_amd/data.py draws them in the reference's order). */
int cffm_clip_format(const unsigned char* frames, const unsigned char* labels, float* out_img, long long* out_lab, int T, int H, int W,
                     int y1, int x1, int ch, int cw, int flip, int Ho, int Wo, const float mean[3], const float std[3], int to_rgb,
                     float pad_val, int seg_pad_val, int reduce_zero_label, void* stream);
/* the same with PhotoMetricDistortion_clips' brightness / contrast (mmseg/datasets/pipelines/transforms.py:2028-2150, convert() :2057)
 * between flip and normalisation, per frame: brightness_beta[t] / contrast_alpha[t] (HOST arrays of T floats, or NULL) -- NaN = branch
 * not taken for that frame; v = u8(clip(v + beta)), then v = u8(clip(v * alpha)) in float32 as numpy does.  (No saturation / hue:
 * cffm_clip_format_hsv.) */
int cffm_clip_format_photo(const unsigned char* frames, const unsigned char* labels, float* out_img, long long* out_lab, int T, int H,
                           int W, int y1, int x1, int ch, int cw, int flip, int Ho, int Wo, const float mean[3], const float std[3],
                           int to_rgb, float pad_val, int seg_pad_val, int reduce_zero_label, const float* brightness_beta,
                           const float* contrast_alpha, void* stream);
/* ABI 8: the whole of PhotoMetricDistortion_clips.__call__ (transforms.py:2112-2139) per frame: brightness, contrast when
 * contrast_first[t] (the reference's mode == 1), saturation (:2082-2091: S of mmcv.bgr2hsv = cv2 COLOR_BGR2HSV scaled by saturation[t],
 * convert()-style, back through COLOR_HSV2BGR), hue (:2093-2102: H + hue_shift[t] mod 180, its own round trip), contrast when
 * !contrast_first[t].  HOST arrays of T entries or NULL; NaN = branch not taken; hue_shift holds integers.  The two colour conversions
 * restate OpenCV's 8-bit arithmetic (color_hsv.cpp RGB2HSV_b: integer with the 12-bit division tables, hue range 180; HSV2RGB_b: float32
 * through HSV2RGB_native, * 255, cvRound) -- OpenCV is not available where this library is built or tested, so that restatement is checked
 * against oracle/cv_oracle.py only (parity unpinned). */
int cffm_clip_format_hsv(const unsigned char* frames, const unsigned char* labels, float* out_img, long long* out_lab, int T, int H, int W,
                         int y1, int x1, int ch, int cw, int flip, int Ho, int Wo, const float mean[3], const float std[3], int to_rgb,
                         float pad_val, int seg_pad_val, int reduce_zero_label, const float* brightness_beta, const float* contrast_alpha,
                         const int* contrast_first, const float* saturation, const float* hue_shift, void* stream);
/* ABI 8: cv2.resize of a clip, as mmcv.imrescale / imresize call it from Resize (transforms.py:475, config vspw_repeat2.py:10) and
 * AlignedResize_clips (:236, config :27): frames [T,H,W,3] uint8 -> out_frames [T,Ho,Wo,3] with INTER_LINEAR (OpenCV's 8-bit path:
 * 11-bit fixed-point weights, horizontal then vertical pass, the 2x2 -> 1 case as INTER_AREA's box mean), labels [T,H,W] uint8 ->
 * out_labels [T,Ho,Wo] with INTER_NEAREST; either pair may be NULL.  The target size is the caller's (vss_cffm_amd/data.py applies
 * mmcv.rescale_size and the size_divisor alignment).  Restated from resize.cpp; parity unpinned like cffm_clip_format_hsv. */
int cffm_clip_resize(const unsigned char* frames, const unsigned char* labels, int T, int H, int W, unsigned char* out_frames,
                     unsigned char* out_labels, int Ho, int Wo, void* stream);

/* ---- parameter update of the training step (the reference trains the head with AdamW, lr 6e-5, betas (0.9, 0.999),
 * weight decay 0.01: local_configs/cffm/B1/cffm.b1.480x480.vspw2.160k.py:35) ----
 * One launch over every parameter tensor of the hot path: `chunks` is a device table, one entry per <= CFFM_ADAMW_CHUNK
 * consecutive elements of one tensor.  Decoupled weight decay, bias-corrected moments, exactly torch.optim.AdamW
 * (amsgrad off): p *= 1 - lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps). */
#define CFFM_ADAMW_CHUNK 2048
typedef struct {
    float* p;           /* parameter values (updated in place) */
    const float* g;     /* gradient */
    float* m;           /* first moment (updated) */
    float* v;           /* second moment (updated) */
    long n;             /* 1..CFFM_ADAMW_CHUNK elements */
} cffm_adamw_chunk;
int cffm_adamw_step(const cffm_adamw_chunk* chunks /* device */, int nchunks, double lr, double beta1, double beta2, double eps,
                    double weight_decay, int step /* t >= 1 */, void* stream);
/* The same update with the step count kept ON THE DEVICE (state: 4 floats, zero-initialised by the caller: [0] = t, advanced
 * by the call, [1], [2] = the bias-correction factors derived from it), so a training step captured in a HIP graph replays
 * with the right t.  grad_base != NULL: every chunk's `g` is a byte offset from grad_base instead of a pointer (all the
 * gradients of the layer live in one buffer; its address may change between steps, the table does not). */
int cffm_adamw_step_dev(const cffm_adamw_chunk* chunks /* device */, int nchunks, const float* grad_base, double lr, double beta1,
                        double beta2, double eps, double weight_decay, float* state /* device [4] */, void* stream);
/* Every parameter group of the optimizer in ONE launch (the reference's paramwise_cfg -- `head` lr_mult 10, `norm` /
 * `pos_block` decay_mult 0: local_configs/cffm/B1/cffm.b1.480x480.vspw2.160k.py:35-39 -- makes several): a chunk names the
 * ROW of the hyper-parameter tables it is updated with.  Device tables, `nrows` rows each:
 *   consts [nrows][12] double: beta1, beta2, eps, schedule kind, max_iters, power, min_lr, warmup_iters, warmup_ratio, first
 *                              iteration, GLOBAL iteration (slot 10: how many steps the optimizer has taken -- the call ADVANCES it
 *                              for every row, whether or not the row owns a chunk), (1 unused).  kind 0: lr_t = sched's lr.  kind 1:
 *                              the reference's schedule evaluated ON THE DEVICE (mmcv poly decay + linear warm-up,
 *                              cffm.b1...160k.py:41-45) from the global iteration, as mmcv derives every group's rate from the
 *                              runner's iteration: it = global - first; lr_t = (lr - min_lr) (1 - it/max_iters)^power + min_lr,
 *                              times 1 - (1 - it/warmup_iters)(1 - warmup_ratio) while it < warmup_iters; 0 for it < 0
 *   sched  [nrows][2] float : (base) lr, weight_decay                -- refreshed by the caller; vss_cffm_amd.optim copies it
 *                              from a pinned host mirror INSIDE the captured step, so graph replays see the current values
 *                              (the caller must order mirror writes against replays in flight; kind 1 needs no writes)
 *   state  [nrows][4] float : t, lr_t/(1-b1^t), 1/sqrt(1-b2^t), 1-lr_t*wd -- t advanced by the call (zero-initialise, or seed
 *                              with the step count of a resumed run)
 * grad_base as in cffm_adamw_step_dev. */
#define CFFM_ADAMW_TICKETS 65
typedef struct {
    float* p;
    const float* g;
    float* m;
    float* v;
    int n;              /* 1..CFFM_ADAMW_CHUNK elements */
    int row;            /* row of consts / sched / state */
} cffm_adamw_chunk2;
/* active_rows (device, [nrows] ints, or NULL = every row): only rows that own a chunk of this call advance their step count.
 * ticket (device int[CFFM_ADAMW_TICKETS], zero before the first call, or NULL): with tickets the step count is advanced INSIDE the update
 * launch (one kernel; the workgroup that finishes last stores the new state) -- same results as the two-launch form used without.
 * sched is read by the device when the launch runs: it may be device memory or pinned, device-visible host memory. */
int cffm_adamw_step_rows(const cffm_adamw_chunk2* chunks /* device */, int nchunks, const float* grad_base, float* state,
                         const float* sched, double* consts, int nrows, const int* active_rows, int* ticket, void* stream);

#ifdef __cplusplus
}
#endif
#endif
