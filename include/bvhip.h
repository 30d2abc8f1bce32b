/* libbvhip — C ABI of the MI355X (gfx950) SigLIP/ViT training-step kernels.
 *
 * The reference (google-research/big_vision) has NO native boundary: every op
 * on the hot path is a jax.numpy / flax.linen call lowered by XLA
 * (SURVEY.md §1, §8b).  This header is therefore the boundary a maintainer
 * would bind (ctypes / jax.ffi custom-call) to replace those lowered ops; each
 * entry point cites the reference call site(s) it replaces, paths relative to
 * big_vision/ in the reference tree.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (no hidden
 *     allocation); tensors are row-major, densely packed unless an ld* argument
 *     says otherwise; `bf16` tensors are raw 16-bit bfloat16.
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*) and
 *     returns immediately: 0 = ok, <0 = error (BV_ERR_*); the message is
 *     available from bv_last_error().  No call synchronises.
 *   - thread-safe per stream.  The library keeps NO process-global state besides the (thread-local)
 *     last-error string (what else is static is immutable after its first use: the compute-unit count of
 *     the device, the kernel attributes the HIP runtime caches, the RCCL entry points bv_comm_* binds): every option
 *     that selects a kernel variant, the split-K workspace and the launch counters live in an opaque
 *     `bv_ctx` the caller creates and passes to the entry points that consult it (GEMM, attention,
 *     fp32 GEMM).  ctx = NULL means "all defaults, no workspace".  Two callers in one process that
 *     hold their own contexts cannot change each other's kernels.
 */
#ifndef BVHIP_H_
#define BVHIP_H_

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): the GEMM / attention / fp32-GEMM entry points take a trailing `const bv_ctx*`, the process-global option
 * setters and the workspace registry are gone, bv_adafactor_step gained block_rms_clip / block_usq. */
#define BVHIP_VERSION 2

/* error codes */
#define BV_OK 0
#define BV_ERR_INVALID_ARG (-1)
#define BV_ERR_UNSUPPORTED (-2)
#define BV_ERR_HIP (-3)

const char* bv_last_error(void);
int bv_version(void);

/* --------------------------------------------------------------- Context ----
 * Options, split-K workspace and launch counters of ONE caller ("no global state").  A
 * context is plain host memory: creating one allocates nothing on the device and enqueues nothing.
 * Entry points take it as `const bv_ctx*` (they only read options; the counters are atomics); a context
 * may be shared by threads as long as nobody calls bv_ctx_set / bv_ctx_set_workspace on it concurrently
 * with a launch.  Its workspace serves ONE stream at a time: callers that enqueue weight-gradient GEMMs
 * on several streams concurrently give each stream its own context. */
typedef struct bv_ctx bv_ctx;
bv_ctx* bv_ctx_create(void);
void bv_ctx_destroy(bv_ctx* ctx);
/* Options (diagnostics / A-B benchmarking; defaults in parentheses; results never depend on them
 * beyond the summation order of a different kernel). */
#define BV_OPT_FAST_PATH 0         /* (1) GEMMs with M, N multiples of 256, K of 64 and both operands in the same
                                      layout run on the 256x256x64 direct-to-LDS kernel; 0: everything on the
                                      general 128x128x64 kernel */
#define BV_OPT_GEMM_NT 1           /* (0) bit 0 = streaming (nontemporal) stores of C / C2, bit 1 = streaming aux loads */
#define BV_OPT_GEMM_SKEW_MODE 2    /* (1) start phase per XCD (0) / per workgroup (1), see SKEW_PCT */
#define BV_OPT_GEMM_SKEW_PCT 3     /* (0) spread of the persistent workgroups' start, % of one tile period */
#define BV_OPT_GEMM_PRE_ISSUE 4    /* (0) issue the next tile's K-tile 1 / 2 loads ahead of the epilogue's stores */
#define BV_OPT_GEMM_ROLL 5         /* (1) which epilogues of k-major GEMMs with K >= 128 run on the rolling-epilogue
                                      kernel (no separate epilogue phase; the residual is loaded straight into the
                                      accumulators): bit mask 1 = RESIDUAL fp32 (alpha = 1), 2 = NONE (bf16 out),
                                      4 = GELU, 8 = stores inside the MFMA segments; 1 is the only one measured faster */
#define BV_OPT_GEMM_GROUP_N 6      /* (0) tile order of the k-major 256x256 kernels: column tiles are walked in groups
                                      of g (column tile fastest inside a group, then row tile): the g weight panels an
                                      XCD works on stay in its L2.  0 or >= N/256: plain order */
#define BV_OPT_GEMM_RESERVE_CUS 7  /* (0) CUs the persistent 256x256 grid leaves free (<= 128).  Its workgroups fill
                                      a CU, so kernels that must run BESIDE it (RCCL collectives overlapping the
                                      backward) need CUs of their own; the data-parallel trainer reserves one per
                                      RCCL channel for the duration of the overlapped backward */
#define BV_OPT_ATTN_CFG 8          /* (0) attention A/B switches: 8 = forward of the L <= 208 kernels as 8 waves x 2
                                      workgroups; +16 = two-sweep dQ kernel; +32 / +64 = 32-key dK/dV kernels;
                                      +128 = two-launch backward where the one-launch kernel applies; +256 = the
                                      one-launch kernel reduces the bias gradients by DPP column sums; +1024 = the
                                      16-key dK/dV kernel also at 28+ key fragments (L > 272), where the 32-key
                                      7-wave kernel is the default since round 6 */
#define BV_OPT_SGEMM_MFMA 9        /* (1) bv_sgemm_strided on the fp32 matrix pipe wherever a 64 x 64 tile is filled;
                                      0 = always the VALU kernel (both are k-ordered fmaf chains: identical results) */
#define BV_OPT_COUNT 10
/* Read-only statistics of a context (bv_ctx_get): launches bv_gemm_bf16[_colsum] put on the 256x256 kernels
 * through this context: all of them / those in which a persistent workgroup walks more than one tile / those of
 * the latter with a fused epilogue (anything but NONE / ATOMIC) or fused column sums.  The parity suite asserts
 * with them that a case really ran where it claims to. */
#define BV_STAT_GEMM256_CALLS 100
#define BV_STAT_GEMM256_MULTI 101
#define BV_STAT_GEMM256_FUSED 102
/* Sets an option and returns its previous value (>= 0); value < 0 only queries.  BV_ERR_INVALID_ARG for an
 * unknown option or a NULL context. */
long bv_ctx_set(bv_ctx* ctx, int opt, long value);
long bv_ctx_get(const bv_ctx* ctx, int opt);
/* Caller-provided scratch (device memory, >= 64 MiB recommended) for the split-K partial tiles of the
 * weight-gradient GEMMs (EPI_ATOMIC) launched through this context: with a workspace the partials are written
 * with plain coalesced stores and combined by a second small kernel (deterministic); without one (no context, or
 * ptr = NULL) fp32 atomics are used.  bv_gemm_workspace_bytes(M, N, K) = bytes the dW GEMM C[M,N] = A[K,M]^T B[K,N]
 * takes with the automatic split choice (0: the shape needs none) -- size the slab for the largest. */
int bv_ctx_set_workspace(bv_ctx* ctx, void* ptr, long bytes);
long bv_gemm_workspace_bytes(int M, int N, int K);

/* ---------------------------------------------------------------- GEMM ----
 * C[M,N] = alpha * A(MxK) * B(KxN) (+ epilogue), bf16 inputs, fp32 MFMA
 * accumulate.  Replaces nn.Dense / nn.DenseGeneral / nn.Conv(stem) matmuls:
 * models/vit.py:72,77 (MlpBlock), :93-98 (q/k/v/out projections), :212-214
 * (patch embedding as a matmul over patchified pixels), :261,:272 (heads),
 * models/proj/image_text/text_transformer.py:98, and their dX / dW transposes
 * produced by jax.value_and_grad (trainers/proj/image_text/siglip.py:311).
 *
 * a_kmajor: 1 = A stored [M][K] (K contiguous, row stride lda);
 *           0 = A stored [K][M] (M contiguous, row stride lda).
 * b_kmajor: 1 = B stored [N][K] (K contiguous, row stride ldb)  ("B^T");
 *           0 = B stored [K][N] (N contiguous, row stride ldb).
 *   forward  Y = X W      : a_kmajor=1, b_kmajor=0 (W is Flax (in,out))
 *   dX = dY W^T           : a_kmajor=1, b_kmajor=1
 *   dW = X^T dY           : a_kmajor=0, b_kmajor=0, epilogue ATOMIC (split-K)
 * out_f32: C is float (1) or bf16 (0).
 * epilogue (applied after alpha, then + bias[N] if bias != NULL):
 */
#define BV_EPI_NONE 0
#define BV_EPI_RESIDUAL 1 /* C += aux[m,n], aux has C's dtype: fp32 (out_f32 = 1) or bf16 (out_f32 = 0,
                             bf16 residual stream)               (x + f(x), vit.py:101,110)   */
#define BV_EPI_POS 2      /* C(f32) += aux(f32)[m % aux_rows,n] (posemb add, vit.py:220-221) */
#define BV_EPI_GELU 3     /* C(bf16)=pre-activation h, C2(bf16)=gelu_tanh(h) (vit.py:75); the
                             activation is applied to the bf16-rounded h that is stored     */
#define BV_EPI_GELU_BWD 4 /* C *= gelu_tanh'(aux(bf16)[m,n])   (backward of vit.py:75)      */
#define BV_EPI_ATOMIC 5   /* C(f32) += result via fp32 atomics; split-K over K              */
#define BV_EPI_GELU_BWD_EMIT 6 /* GELU_BWD, and C2(bf16) = gelu_tanh(aux): the activation is
                             recomputed by the backward instead of being kept (bit-identical
                             to what BV_EPI_GELU wrote)                                      */
#define BV_EPI_GELU_GD 7  /* C(bf16) = gelu_tanh(h), C2(bf16) = gelu_tanh'(h), h = the bf16-rounded result
                             (vit.py:75 and its derivative in ONE forward epilogue: the pre-activation
                             is not stored; the backward multiplies by C2 with BV_EPI_MUL)        */
#define BV_EPI_MUL 8      /* C(bf16) *= aux(bf16)[m,n]  (dH = dG o gelu'(h), gelu' kept by GELU_GD) */
#define BV_EPI_GELU_G 9   /* C(bf16) = gelu_tanh(h), h = the bf16-rounded result; nothing else is written (C2 ignored):
                             the MLP of a forward that saves no context - a frozen tower, inference (vit.py:75).
                             Bit-identical to the C2 of BV_EPI_GELU / the C of BV_EPI_GELU_GD                      */
int bv_gemm_bf16(int a_kmajor, int b_kmajor, const void* A, long lda, const void* B, long ldb,
                 void* C, long ldc, int out_f32, int M, int N, int K, int epilogue,
                 const float* bias, const void* aux, long ldaux, int aux_rows, void* C2,
                 float alpha, int split_k /*0 = auto*/, void* stream, const bv_ctx* ctx);

/* bv_gemm_bf16 with a fused column reduction: colsum[n] += sum_m C[m][n], taken from the fp32
 * results before the output rounding (fp32 atomics).  Supported with the GELU_BWD / MUL epilogues:
 * the column sums of dH are the gradient of the MlpBlock Dense_0 bias (vit.py:72), which
 * saves a separate pass over the [tokens, mlp_dim] tensor.  colsum = NULL: plain bv_gemm_bf16. */
int bv_gemm_bf16_colsum(int a_kmajor, int b_kmajor, const void* A, long lda, const void* B, long ldb,
                        void* C, long ldc, int out_f32, int M, int N, int K, int epilogue,
                        const float* bias, const void* aux, long ldaux, int aux_rows, void* C2,
                        float alpha, int split_k, float* colsum, void* stream, const bv_ctx* ctx);

/* fp32 GEMM with arbitrary element strides (small, numerically sensitive
 * products: the B x B logits of the sigmoid loss and its gradients,
 * trainers/proj/image_text/siglip.py:291).  C[m,n] = alpha * sum_k A(m,k) B(k,n)
 * + beta * C[m,n]; element (m,k) of A is A[m*sam + k*sak] etc. */
int bv_sgemm_strided(const float* A, long sam, long sak, const float* B, long sbk, long sbn,
                     float* C, long ldc, int M, int N, int K, float alpha, float beta,
                     const float* log_alpha /*device, optional: alpha *= exp(*log_alpha)*/,
                     void* stream, const bv_ctx* ctx);
/* ------------------------------------------------------------ LayerNorm ----
 * flax nn.LayerNorm(): eps=1e-6, fp32 statistics (models/vit.py:92,103,160,181).
 * Input row r is read at x + (r*row_stride + row_offset)*D (row_stride>=1 lets
 * the final encoder_norm run only on the pooled token, text_transformer.py:82-84).
 * Outputs: y_bf16 and/or y_f32 (either may be NULL), mean[rows], rstd[rows]. */
int bv_layernorm_fwd(const float* x, const float* scale, const float* bias, void* y_bf16,
                     float* y_f32, float* mean, float* rstd, int rows, int D, long row_stride,
                     long row_offset, float eps, void* stream);
/* dx_out[row] = (dres ? dres[row] : 0) + LN_bwd(dy[row]); optional bf16 copy of
 * dx_out; dscale/dbias are ACCUMULATED (+=) with fp32 atomics.  dy is bf16 or
 * fp32 (dy_is_f32).  With row_stride>1 only the selected rows of dx are
 * written (caller zero-fills the rest).  dx_colsum (optional, [D], accumulated):
 * column sums of dx_out = the bias gradient of the Dense layer that produced
 * this LayerNorm's input row (out-projection / MLP Dense_1, vit.py:101,110),
 * fused here so that gradient needs no extra pass over dx. */
int bv_layernorm_bwd(const void* dy, int dy_is_f32, const float* x, const float* scale,
                     const float* mean, const float* rstd, const float* dres, float* dx,
                     void* dx_bf16, float* dscale, float* dbias, float* dx_colsum, int rows, int D,
                     long row_stride, long row_offset, void* stream);
/* bv_layernorm_bwd that ALSO re-emits the forward's bf16 output y = LN(x) (y_bf16 [rows][D], needs the LayerNorm
 * bias; row_stride must be 1): the "light" activation contexts of the micro-batched trainer do not keep the
 * LayerNorm outputs (vit.py:92,103 feed them to the QKV / Dense_0 projections, whose weight gradients need them
 * again), and this kernel reads x anyway - re-deriving y here costs 2 bytes per element instead of the 6 of a
 * second bv_layernorm_fwd pass.  Same expression, same bits as bv_layernorm_fwd. */
int bv_layernorm_bwd_y(const void* dy, int dy_is_f32, const float* x, const float* scale,
                       const float* mean, const float* rstd, const float* dres, float* dx,
                       void* dx_bf16, float* dscale, float* dbias, float* dx_colsum, int rows, int D,
                       long row_stride, long row_offset, const float* bias, void* y_bf16, void* stream);
/* The same on a bf16 RESIDUAL STREAM (trainer option config.residual_stream = "bfloat16": the activations
 * between the blocks, their gradients and the saved block inputs are bf16; statistics, scale / bias
 * gradients, dx_colsum and all arithmetic stay fp32).  x, dres and dx are bf16 here; there is no separate
 * fp32 dx (the bf16 dx IS the gradient stream and the GEMM operand).  D % 8 == 0, D <= 2048. */
int bv_layernorm_fwd_bf16x(const void* x_bf16, const float* scale, const float* bias, void* y_bf16, float* y_f32,
                           float* mean, float* rstd, int rows, int D, long row_stride, long row_offset, float eps,
                           void* stream);
int bv_layernorm_bwd_bf16x(const void* dy, int dy_is_f32, const void* x_bf16, const float* scale, const float* mean,
                           const float* rstd, const void* dres_bf16, void* dx_bf16, float* dscale, float* dbias,
                           float* dx_colsum, int rows, int D, long row_stride, long row_offset, void* stream);

/* ------------------------------------------------------------ Attention ----
 * Self-attention core of nn.MultiHeadDotProductAttention (models/vit.py:93-98):
 * S = (q/sqrt(Dh)) k^T, P = softmax(S), O = P v per (sample, head); no mask, no
 * dropout.  qkv is the packed projection output [n*L][3][H][64] bf16 (row
 * stride 3*H*64); o is [n*L][H][64] bf16; lse is [n][H][L] fp32 (row
 * log-sum-exp, saved for the backward).  Dh must be 64; L <= 576. */
int bv_attn_fwd(const void* qkv, void* o, float* lse, int n, int L, int H, void* stream, const bv_ctx* ctx);
/* dqkv (same layout as qkv) from do ([n*L][H][64] bf16); delta is fp32 scratch
 * [n][H][L] (rowsum(dO*O), computed here).  dbias_rows (optional, fp32 [n][3][H][64]) receives
 * the per-sample column sums of dqkv, reduced inside the kernels from the fp32 results; summed
 * over n (bv_colsum) they are the gradient of the query/key/value projection biases
 * (vit.py:93-98) - no separate pass over dqkv.  Precision of dbias_rows on the one-launch path (unmasked
 * L <= 64 and 193..208, attention5.hip): value bias = fp32 column sums of the dO tile (exact identity: rows
 * of P sum to 1); key bias = 0 - the exact value (sum_j dS_ij = 0 for every query: a key bias shifts a whole
 * score row), where autodiff of the reference leaves rounding noise of ~1e-9 relative; query bias =
 * scale * sum_j cs_j K_j with the per-key sums cs_j = sum_i dS_ij taken by the matrix pipe in fp32 and ROUNDED
 * TO bf16 (2^-9 relative per key) when L % 16 != 0 - the same rounding every dS element gets before it enters
 * dQ, so the bias gradient carries the dQ rows' error, not more (tests bound both at 2e-2 of the tensor norm;
 * measured ~3e-3).  The two-launch path (masked / other L) sums the fp32 results by DPP. */
int bv_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta,
                void* dqkv, float* dbias_rows, int n, int L, int H, void* stream, const bv_ctx* ctx);

/* The same with a key-padding length per sample: keys >= kv_len[i] (1 <= kv_len[i] <= L, int32
 * device array, NULL = no mask) get zero probability and zero dK / dV rows; all L query rows are
 * computed.  Replaces nn.MultiHeadDotProductAttention(mask=...) of the NaFlex tower
 * (models/proj/image_text/naflex_vit.py:84-293: the mask marks the valid patches, padding sits
 * at the end of the sequence).  The backward computes delta = rowsum(P o dP) itself (fp32) and
 * does not read o. */
int bv_attn_fwd_masked(const void* qkv, void* o, float* lse, const int* kv_len, int n, int L, int H,
                       void* stream, const bv_ctx* ctx);
int bv_attn_bwd_masked(const void* qkv, const void* d_o, const float* lse, const int* kv_len, float* delta,
                       void* dqkv, float* dbias_rows, int n, int L, int H, void* stream, const bv_ctx* ctx);

/* Single-query attention of the MAP head (models/vit.py:176-178): q [n][H][64]
 * bf16, kv packed [n*L][2][H][64] bf16 -> o [n][H][64] bf16, probabilities p
 * [n][H][L] fp32 (saved for the backward). */
int bv_map_attn_fwd(const void* q, const void* kv, void* o, float* p, int n, int L, int H,
                    void* stream);
int bv_map_attn_bwd(const void* q, const void* kv, const float* p, const void* d_o, void* dq,
                    void* dkv, int n, int L, int H, void* stream);
/* The same with the NaFlex pool mask (naflex_vit.py:183-199, :262-263): keys >= kv_len[i] (int32 [n])
 * get probability 0; bv_map_attn_bwd needs no mask (it works from the saved probabilities). */
int bv_map_attn_fwd_masked(const void* q, const void* kv, void* o, float* p, const int* kv_len, int n, int L,
                           int H, void* stream);
/* Head dims other than 64 (Dh % 8 == 0, Dh <= 128) and any sequence length: So400m (1152 / 16 = 72),
 * `mu` (32 / 2 = 16, the variant of the reference's own tests), Ti / S at other widths
 * (models/vit.py:284-303 decode_variant; nn.MultiHeadDotProductAttention models/vit.py:93-98, MAP head
 * :176-178).  Same tensors as bv_attn_fwd / bv_attn_bwd_masked / bv_map_attn_* with 64 -> Dh:
 * qkv [n*L][3][H][Dh], o [n*L][H][Dh], lse / delta [n][H][L], dbias_rows [n][3][H][Dh] (optional),
 * kv_len optional (int32 [n], >= 1).  Flash-style kernels (attention_dh.hip); the Dh = 64 entry points
 * above stay the fast path of every BASELINE model. */
int bv_attn_fwd_dh(const void* qkv, void* o, float* lse, const int* kv_len, int n, int L, int H, int Dh,
                   void* stream);
int bv_attn_bwd_dh(const void* qkv, const void* d_o, const float* lse, const int* kv_len, float* delta,
                   void* dqkv, float* dbias_rows, int n, int L, int H, int Dh, void* stream);
int bv_map_attn_fwd_dh(const void* q, const void* kv, void* o, float* p, const int* kv_len, int n, int L, int H,
                       int Dh, void* stream);
int bv_map_attn_bwd_dh(const void* q, const void* kv, const float* p, const void* d_o, void* dq, void* dkv, int n,
                       int L, int H, int Dh, void* stream);
/* Masked global average pooling of the NaFlex tower (naflex_vit.py:264-266): mean over the first
 * len[b] tokens of [n][L][D] fp32; the backward writes dy / len[b] to those rows, 0 to the padding. */
int bv_pool_gap_masked_fwd(const float* x, float* y, const int* len, int n, int L, int D, void* stream);
int bv_pool_gap_masked_bwd(const float* dy, float* dx, const int* len, int n, int L, int D, void* stream);
/* NaFlex position embedding (naflex_vit.py:38-83): per token t of example e the weights of
 * jax.image.scale_and_translate(bilinear, antialias) from the learned [P][P] grid to the example's
 * own patch grid (max coordinate + 1 per axis), gathered at (yabs, xabs):
 * W[e*N + t][i*P + j] = Wy_e[yabs[t]][i] * Wx_e[xabs[t]][j] (bf16).  The embedding of a token is
 * W[t] . pos[P*P][D] and d pos = W^T d tok: both run on bv_gemm_bf16.  P <= 64. */
int bv_naflex_posemb_weights(const int* yabs, const int* xabs, void* W, int n, int N, int P, void* stream);

/* ---------------------------------------------------------- Patch / embed --
 * NHWC fp32 image [n][Hi][Wi][3] -> bf16 patch matrix [n*h*w][P*P*3] in HWIO
 * flattening order (row, col, channel), the im2col of the stride-P VALID conv
 * of models/vit.py:212-217. */
int bv_patchify(const float* image, void* patches, int n, int Hi, int Wi, int P, void* stream);
/* The same with a row pitch ldo >= P*P*3 (elements); columns [P*P*3, ldo) are written as zeros.  Patch
 * sizes whose P*P*3 is not a multiple of 8 (So400m/14: 588) pad their rows to the next multiple for the
 * 16-byte operand loads of bv_gemm_bf16. */
int bv_patchify_ld(const float* image, void* patches, int n, int Hi, int Wi, int P, int ldo, void* stream);
/* x[(i*L+l)] = table[ids[i*L+l]] + pos[l]  (nn.Embed + pos_embedding,
 * models/proj/image_text/text_transformer.py:63-70).  fp32 in/out. */
int bv_embed_fwd(const int* ids, const float* table, const float* pos, float* x, int n, int L,
                 int D, int vocab, void* stream);
/* dtable[ids[r]] += dx[r] (fp32 atomics). */
int bv_embed_bwd(const int* ids, const float* dx, float* dtable, int rows, int D, int vocab,
                 void* stream);

/* ------------------------------------------------------------- Reductions --
 * out[c] += sum_r x[r][c] (bias gradients db = sum_rows dY).  x bf16 or fp32. */
int bv_colsum(const void* x, int x_is_f32, long ldx, float* out, int rows, int cols, void* stream);
/* out[l][c] += sum_i x[i][l][c]  (dpos = sum over the batch, fp32). */
int bv_batchsum(const float* x, float* out, int n, int L, int D, void* stream);
/* fp32 -> bf16 cast of a flat buffer (bf16 weight shadows). */
int bv_cast_bf16(const float* x, void* y, long count, void* stream);
/* bf16 -> fp32 (the boundary of a bf16 residual stream). */
int bv_cast_f32(const void* x_bf16, float* y, long count, void* stream);
/* dst[cols][rows] = src[rows][cols]^T, bf16 (row strides in elements): the
 * [out][in] image of a Flax (in,out) kernel, consumed by the forward
 * projections (models/vit.py:72,77,93-98) on the k-major GEMM path. */
int bv_transpose_bf16(const void* src, void* dst, int rows, int cols, long lds, long ldd,
                      void* stream);
/* The same for a table of matrices in ONE launch (all projection kernels of the towers after an optimizer
 * step).  `leaves`: device array; tile0 = number of 64x64 tiles of the entries before this one (ascending),
 * tiles_x = ceil(cols / 64); total_tiles = sum over entries of ceil(rows / 64) * tiles_x. */
typedef struct bv_tr_leaf {
  const void* src;
  void* dst;
  long lds, ldd;
  int rows, cols;
  int tile0, tiles_x;
} bv_tr_leaf;
int bv_transpose_bf16_batched(const bv_tr_leaf* leaves, int nleaves, int total_tiles, void* stream);
/* y[i][0] = cls, y[i][1+l] = x[i][l] (cls-token concat, models/vit.py:223-225), fp32. */
int bv_concat_cls(const float* cls, const float* x, float* y, int n, int L, int D, void* stream);

/* pooled[i][c] = mean_l x[i][l][c] (pool_type="gap", models/vit.py:246); fp32. */
int bv_pool_gap_fwd(const float* x, float* y, int n, int L, int D, void* stream);
int bv_pool_gap_bwd(const float* dy, float* dx, int n, int L, int D, void* stream);
/* pooled[i][c] = max_l x[i][l][c] (pool_type "max" / "gmp" of the text tower,
 * models/proj/image_text/text_transformer.py:89-90); argmax [n][D] int32 keeps the position of the (first) maximum,
 * the backward writes dy[i][c] there and 0 to the other L - 1 positions; fp32. */
int bv_pool_max_fwd(const float* x, float* y, int* argmax, int n, int L, int D, void* stream);
int bv_pool_max_bwd(const float* dy, const int* argmax, float* dx, int n, int L, int D, void* stream);
/* The same over the first len[i] positions of sample i only (NaFlex pool_type "max", naflex_vit.py:267-271: padded tokens
 * enter the maximum as finfo.min, i.e. never win); the backward is bv_pool_max_bwd (argmax lies inside the valid range). */
int bv_pool_max_masked_fwd(const float* x, float* y, int* argmax, const int* len, int n, int L, int D, void* stream);

/* ------------------------------------------------------------ L2 normalise --
 * zn = z / (||z||_2 + eps), eps = 1e-8 (models/proj/image_text/two_towers.py:60-61,
 * :73-74).  fp32. */
int bv_l2norm_fwd(const float* z, float* zn, float* norm, int rows, int D, float eps, void* stream);
int bv_l2norm_bwd(const float* z, const float* norm, const float* dzn, float* dz, int rows, int D,
                  float eps, void* stream);

/* ---------------------------------------------------------- Sigmoid loss ----
 * Pairwise sigmoid loss of trainers/proj/image_text/siglip.py:291-306 (==
 * _deprecated_contrastive.py:117-141 summed over devices).  `raw` holds
 * zimg_local . ztxt_all^T [n][B] fp32 on entry; on exit it holds
 * G = dL/dS = -(1/B_global) * m * sigmoid(-m*S), S = t*raw + b, m = +1 on the
 * positive diagonal (column row_offset + i) else -1.
 * stats is double[3] (device), ACCUMULATED:
 * stats[0] += sum_ij -log_sigmoid(m*S) / B_global   (this rank's loss share)
 * stats[1] += sum_ij G*(S-b)   (= dL/dt', t = exp(t'))
 * stats[2] += sum_ij G         (= dL/db)
 * t and b are read from device memory (t_param holds t' = log t). */
int bv_siglip_loss(float* raw, const float* t_param, const float* b_param, double* stats, int n,
                   int B, int row_offset, int B_global, void* stream);

/* Logit statistics the pmap contrastive trainer logs with the sigmoid loss
 * (trainers/proj/image_text/_deprecated_contrastive.py:143-160): for logits = exp(t') raw + b of this
 * rank's n images against all B texts (raw as passed to bv_siglip_loss, BEFORE it is overwritten),
 * out9 = {pos_min, pos_max, pos_avg, local_neg_min, local_neg_max, local_neg_avg, neg_min, neg_max,
 * neg_avg}; "local" = the n x n block of this rank's own texts (columns row_offset ..).  part is
 * scratch of BV_LOGIT_STATS_BLOCKS * 9 floats. */
#define BV_LOGIT_STATS_BLOCKS 512
int bv_logit_stats(const float* raw, const float* t_param, const float* b_param, float* part, float* out9,
                   int n, int B, int row_offset, void* stream);
/* out[0] += sum_i a[i] b[i] (fp64 accumulator): dL/dt' = sum G o logits of the softmax contrastive
 * loss (_deprecated_contrastive.py:80-101). */
int bv_dot_f32(const float* a, const float* b, long count, double* out, void* stream);

/* Classification losses of big_vision/train.py:295-300 (BASELINE config 1), soft labels
 * [n, C] fp32.  n = rows of this call, n_global = rows of the whole (data-parallel) batch: the
 * mean is over n_global, so per-rank results are partial sums (all-reduce SUM).
 * softmax_xent (utils.py:276-281): loss_sum[0] += sum_i -sum_c y log_softmax(l) / n_global,
 *   dlogits = (softmax * sum_c(y) - y) / n_global.
 * sigmoid_xent (utils.py:236-243): loss_sum[0] += sum_i -sum_c [y log sig(l) + (1-y) log sig(-l)]
 *   / n_global, dlogits = (sigmoid(l) - y) / n_global.  dlogits may be NULL (forward only). */
int bv_softmax_xent(const float* logits, const float* labels, double* loss_sum, float* dlogits,
                    int n, int C, int n_global, void* stream);
int bv_sigmoid_xent(const float* logits, const float* labels, double* loss_sum, float* dlogits,
                    int n, int C, int n_global, void* stream);

/* pre_logits = tanh(Dense(x)) of the classification head (models/vit.py:259-262) and its
 * backward dx = dy (1 - y^2); fp32, elementwise. */
int bv_tanh_fwd(const float* x, float* y, long count, void* stream);
int bv_tanh_bwd(const float* y, const float* dy, float* dx, long count, void* stream);

/* Dropout of the encoder (models/vit.py:76 after the GELU, :100 / :109 on the two residual branches, :228 behind the
 * position embedding; `nn.Dropout(rate)(x, deterministic)`): y = keep x / (1 - rate), keep ~ Bernoulli(1 - rate).
 * The keep bits are a pure function of (key, element index): Philox-4x32-10 keyed by the caller's 64-bit site key,
 * counter = index of a group of four consecutive elements, element j of the group keeps iff output word j <
 * floor((1 - rate) 2^32).  Nothing is stored: the backward (and a re-run forward) passes the same key.
 *   bv_dropout_f32:  out = addend (NULL: 0) + keep x / (1 - rate), fp32 in; writes out_f32 and / or out_bf16 (either
 *                    may be NULL; out_f32 may alias x or addend).  Forward of a residual branch: x = branch output,
 *                    addend = the stream; backward: x = the stream's gradient, out_bf16 = the branch's GEMM operand.
 *   bv_dropout_bf16: a (and b, may be NULL) *= keep / (1 - rate) in place, bf16, the SAME bits on both (gelu(h) and the
 *                    stored gelu'(h): the backward's product then carries the mask).
 *   bv_dropout_mask: keep_u8[i] = the bit of element i (parity tests hand the masks to the oracle).
 * count: elements, a multiple of 4. */
int bv_dropout_f32(const float* x, const float* addend, float* out_f32, void* out_bf16, long count,
                   unsigned long long key, float rate, void* stream);
int bv_dropout_bf16(void* a, void* b, long count, unsigned long long key, float rate, void* stream);
int bv_dropout_mask(void* keep_u8, long count, unsigned long long key, float rate, void* stream);

/* Mixup (utils.py:1146-1154): out[i] = a x[i] + (1 - a) x[(i - 1) mod n] over n rows of
 * row_elems floats (jnp.roll(x, shift=1, axis=0)); used for images and soft labels alike. */
int bv_mixup(const float* x, float* out, float a, int n, long row_elems, void* stream);

/* -------------------------------------------------------------- Optimizer --
 * sqnorm_out[0] += sum x^2 (double accumulator) — global grad norm for
 * optax.clip_by_global_norm (optax.py:100-105) and the l2_* measurements
 * (trainers/proj/image_text/siglip.py:315-321). */
int bv_sqnorm(const float* x, long count, double* sqnorm_out, void* stream);

/* Fused bv_optax chain (optax.py:143-149): clip -> scale_by_adam -> scale(lr)
 * -> lr_mult -> add_decayed_weights -> scale_by_schedule -> scale(-1) ->
 * apply_updates (trainers/proj/image_text/siglip.py:312-313), over a flat fp32
 * parameter buffer of `count` elements (multiple of 1024).  Chunk c (1024
 * elements) uses hyper-parameters segs[chunk_seg[c]] (both device arrays):
 * lr_eff = lr*lr_mult, wd_eff = wd*wd_mult, sched_idx selects one of the (at
 * most BV_MAX_SCHED) per-step schedule values passed by value from the host.  mu may be bf16 (mu_bf16=1, optax mu_dtype) or fp32.  gsq points at
 * sum(g^2) over the not-frozen grads (device double); clip_norm <= 0 disables
 * clipping.  bc1 = 1-b1^k, bc2 = 1-b2^k.  Also refreshes the bf16 shadow of the
 * params (may be NULL) and accumulates stats[0] += sum p_new^2,
 * stats[1] += sum update^2 (device doubles; stats may be NULL). */
#define BV_MAX_SCHED 8
typedef struct {
  float lr_eff;
  float wd_eff;
  int sched_idx; /* index into the per-step schedule values `sched` (host array) */
  int pad_;
} bv_adam_seg;
int bv_adam_step(float* params, const float* grads, void* mu, int mu_bf16, float* nu,
                 void* shadow_bf16, const bv_adam_seg* segs, const int* chunk_seg, long count,
                 const float* sched /*host, nsched values*/, int nsched, const double* gsq,
                 float clip_norm, float b1, float b2, float eps, float bc1, float bc2,
                 double* stats, void* stream);

/* One parameter LEAF of the BigVision Adafactor step (big_vision/optax.py:187-216:
 * scale_by_factored_rms(decay 1 - t^-0.8 capped at 0.999, min_dim_size_to_factor 32, eps 1e-30) ->
 * ema(0.9, debias=False, bf16 accumulator)) fused with the rest of the bv_optax chain
 * (optax.py:100-149: global-norm clip factor from gsq, lr * lr_mult, decoupled weight decay,
 * schedule, sign, apply + bf16 shadow refresh).  The leaf is the strided 4-D view
 * view = {offset, B1, B2, R, C, strideB1, strideB2, strideR, strideC} (HOST array of 9 longs,
 * element units) of params / grads / momentum / shadow: C = its largest axis, R = the second
 * largest (optax's factored dims), B1, B2 = the remaining axes (1 if absent).  state (device fp32):
 * factored != 0: v_row[B*R], v_col[B*C], rcm[B] consecutively; factored == 0: v per element in the
 * view's [B1][B2][R][C] order.  decay / sched / lr_eff / wd are this step's host scalars; mom = 0
 * disables the momentum EMA; stats[0] += sum p_new^2, stats[1] += sum update^2 (may be NULL). */
int bv_adafactor_leaf(float* params, const float* grads, void* momentum, int mom_bf16, void* shadow_bf16,
                      const long* view, float* state, int factored, const double* gsq, float clip_norm,
                      float decay, float eps, float mom, float lr_eff, float wd, float sched,
                      double* stats, void* stream);

/* The same for ALL leaves of a model in four launches (a B/16 + text-B model has ~440 leaves: one launch group per
 * leaf is ~1.5 k tiny launches per step).  `leaves`: DEVICE table, one entry per leaf = the view of
 * bv_adafactor_leaf + soff (offset of the leaf's statistics in `state`) + its hyper-parameters; sched: HOST array of
 * this step's schedule values, indexed by sched_idx; max_*: grid extents (max over the factored leaves of B*R, B*C,
 * B; max over all leaves of B*R*C). */
typedef struct {
  long off;               /* element offset of the view origin in params / grads / momentum / shadow */
  long sB1, sB2, sR, sC;  /* strides of the view (elements) */
  long soff;              /* element offset of this leaf's statistics in `state` */
  int B1, B2, R, C;
  int factored, sched_idx, r_fast, pad_;
  float lr_eff, wd;
} bv_af_leaf;
int bv_adafactor_step(float* params, const float* grads, void* momentum, int mom_bf16, void* shadow_bf16,
                      const bv_af_leaf* leaves, int nleaves, long max_rows, long max_cols, long max_b, long max_total,
                      float* state, const double* gsq, float clip_norm, float decay, float eps, float mom,
                      const float* sched /*host*/, int nsched, double* stats,
                      float block_rms_clip /* scale_by_adafactor(clipping_threshold=...) = optax.clip_by_block_rms per leaf
                                              (optax.py:190,208); <= 0: off */,
                      double* block_usq /* device scratch [nleaves] when block_rms_clip > 0, else NULL */, void* stream);

/* ------------------------------------------------------- Collectives (RCCL) ----
 * The exchange steps of the data-parallel step for hosts that do not go through torch.distributed (the
 * Python host does: big_vision_amd/dp.py issues the same collectives through ProcessGroupNCCL = RCCL):
 *   all_gather of the local text embeddings for the global-batch loss
 *       (trainers/proj/image_text/_deprecated_contrastive.py:67-77,122; siglip.py:287-306 under a device mesh)
 *   reduce_scatter of their gradients (the transpose of that gather)
 *   bucketed all_reduce(SUM) of the flat fp32 gradient buffer (pmean of grads :343-344; sharding.py:83-101)
 *   reduce_scatter / all_gather of the flat buffer for the "fsdp" placement (sharding.py:104-139)
 * RCCL is bound at run time (dlopen): an RCCL already in the process is reused, libbvhip.so loads without one.
 * One communicator per process = per GPU (the calling thread's current HIP device).  id: BV_COMM_ID_BYTES
 * bytes created by rank 0 and carried to the other ranks by the host.  Counts are ELEMENTS of `dtype`.
 * The calls are asynchronous on `stream`. */
#define BV_COMM_ID_BYTES 128
#define BV_COMM_F32 0
#define BV_COMM_BF16 1
#define BV_COMM_F64 2
int bv_comm_version(int* version);                       /* RCCL version code, e.g. 22606 */
int bv_comm_unique_id(void* id_out /* BV_COMM_ID_BYTES */);
int bv_comm_init(const void* id, int rank, int world, void** comm_out);
int bv_comm_destroy(void* comm);
int bv_comm_all_gather(void* comm, const void* send, void* recv /* world * count_per_rank */, long count_per_rank,
                       int dtype, void* stream);
int bv_comm_reduce_scatter(void* comm, const void* send /* world * count_per_rank */, void* recv, long count_per_rank,
                           int dtype, void* stream);
/* in-place SUM of buf[0 .. count) in buckets of bucket_elems elements (0 = one message) */
int bv_comm_all_reduce_bucket(void* comm, void* buf, long count, long bucket_elems, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BVHIP_H_ */
