/* libvla_native.so - C ABI of the MI355X-native VLA-Adapter fine-tune hot path.
 *
 * The reference (liruiluo/VLA-Adapter) is pure Python and has NO FFI/plugin boundary: every
 * device op is reached through torch / timm / transformers / flash-attn.  This header is therefore the boundary
 * a maintainer binds from Python (ctypes stub in INTEGRATION.md); each entry cites the reference call site whose
 * arithmetic it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no torch types.  All pointers are DEVICE pointers unless marked host.
 *  - `stream` is a hipStream_t (NULL = default stream).  All work is enqueued on it; no hidden syncs, no
 *    allocation: the caller (PyTorch) owns every buffer.  One process per GPU.
 *  - bf16 tensors are raw uint16 storage; "f32" = float.  Leading dimensions / strides are in ELEMENTS.
 *  - Return 0 on success, negative on error (VLA_ERR_*); vla_last_error() gives a message.  No exceptions cross
 *    the ABI.  Shapes are validated on the host BEFORE launch (a faulting kernel can take the node down).
 *  - Rounding points follow the reference under bf16 autocast (one bf16 rounding after every Linear(+bias),
 *    activation, residual add, norm); accumulation / softmax / statistics are fp32.
 */
#ifndef VLA_NATIVE_H
#define VLA_NATIVE_H
#ifdef __cplusplus
extern "C" {
#endif

#define VLA_ERR_ARG (-1)
#define VLA_ERR_LAUNCH (-2)
#define VLA_ERR_UNSUPPORTED (-3)

/* ABI version.  It changes whenever an entry point's signature or a descriptor struct's layout changes:
 *   1  rounds 1-2 (vla_gemm_desc ended at `bias_post_round`, later at `b_scale`);
 *   2  round 3: vla_gemm_desc gained the K extension (A2 .. ldb2), new entry points vla_gemm_bf16_tn (+ _grouped), vla_gemm256_extent_ok,
 *      vla_copy_rows3d, vla_layerscale_fwd / _bwd, vla_token_ce_bwd, vla_desc_size.
 *   3  round 3: vla_head_attn_desc gained the optional backward workspace (ws, ws_floats).
 *   4  round 3: vla_gemm_desc gained the RMSNorm fields (ssq_out .. rstd_out); new entry point vla_gemm_uses_256.
 *   5  round 4: the RMSNorm fields are gone again (the fold measured slower and left the tree: tools/diag/gemm256_pruned_paths.patch);
 *      fp8 = 1 may be combined with the K extension (and, in that form, with the SwiGLU-backward epilogue).
 *   6  round 4: new entry points vla_gemm_latency_hint, vla_dropout_bf16, vla_dropout_bwd_add_bf16, vla_inc_i32 (no layout change).
 * A binder checks vla_version() AND vla_desc_size() against its own struct definitions before the first call. */
#define VLA_ABI_VERSION 6
int vla_version(void);
/* sizeof() of the descriptor structs as this library was compiled: which = 0 vla_gemm_desc, 1 vla_attn_desc, 2 vla_head_attn_desc,
 * 3 vla_gemm_tn_desc; -1 for an unknown index.  A caller whose struct is shorter would make the library read past its end. */
int vla_desc_size(int which);
const char* vla_last_error(void);      /* thread-local message of the last failing call */

/* ---------------------------------------------------------------- GEMM */
enum { VLA_ACT_NONE = 0, VLA_ACT_GELU = 1, VLA_ACT_RELU = 2, VLA_ACT_GELU_TANH = 3, VLA_ACT_SWIGLU = 4, VLA_ACT_SWIGLU_BWD = 5 };

typedef struct vla_gemm_desc {
  const void* A;    /* [batch][M, K] bf16, row stride lda */
  const void* B;    /* [batch][N, K] bf16, row stride ldb (nn.Linear weight layout [out, in]) */
  void* C;          /* [batch][M, N] bf16, row stride ldc (may be NULL for VLA_ACT_SWIGLU) */
  const void* bias; /* [N] bf16 or NULL */
  const void* R;    /* residual [M or res_mod, N] bf16 or NULL: C = bf16(bf16(act(..)) + R) */
  void* C2;         /* VLA_ACT_SWIGLU only: h[M, N/2] = silu(gate)*up */
  int M, N, K, lda, ldb, ldc, ldr, ldc2;
  int res_mod;      /* >0: residual row = m % res_mod (ViT pos_embed broadcast over the batch) */
  int act, batch;
  long long sA, sB, sC, sR, sC2, sBias; /* batch strides (elements) */
  float alpha;      /* accumulator scale (0 -> 1) */
  /* optional row-group addressing (0 = plain): row r of A lives at (r / a_group) * a_group_stride + (r % a_group) * lda
   * (likewise C): lets a GEMM read/write the [B, first Np of S, D] window of a [B, S, D] tensor as one [B*Np, D] matrix */
  int a_group, c_group;
  long long a_group_stride, c_group_stride;
  /* optional fused rotary embedding on output columns [0, rope_cols) (after bias, before residual): position =
   * row % rope_T.  rope_mode 1: HF rotate_half, head dim 64 or 128, tables f32 [rope_T, rope_dh/2] (Qwen2 q/k, see vla_rope_half);
   * rope_mode 2: action-head interleaved pairs, tables f32 [rope_T, rope_dh] (see vla_rope_interleaved). */
  int rope_mode, rope_T, rope_dh, rope_cols;
  const float* rope_cos; const float* rope_sin;
  /* row-group addressing for R (0 = plain), same rule as a_group: the residual / the SwiGLU-backward pre-activations
   * may be the [B, rows r0.. of S, N] window of a larger tensor (live-row LLM backward) */
  int r_group; long long r_group_stride;
  /* optional: rows m of C with (m % c_live_mod) < c_live_from are NOT stored (c_live_mod 0 = store all).  Used for
   * tensors kept only for a live-row backward: the SwiGLU pre-activations of the rows the backward never visits. */
  int c_live_mod, c_live_from;
  /* optional split-K (0/1 = off): the K range is cut into split_k slices that run as blockIdx.z, each parking its fp32
   * accumulators in its own plane of `ws` (device, fp32 [split_k, M, N], no initialisation needed); a second small kernel
   * sums the planes and applies bias / activation (none, GELU, ReLU) / residual.  For few-tile long-K problems (batch-1
   * inference); batch == 1 and N % 4 == 0 only.  A slice is ceil(K / 64 / split_k) K-tiles of 64; when that does not divide K / 64 the
   * last slice is shorter (and fewer than split_k slices may run): any split_k >= 2 is accepted. */
  int split_k; float* ws;
  /* optional: 1 = round the product to bf16 BEFORE the bias is added, C = bf16(bf16(alpha A.B^T) + bias).  This is what
   * torch's CPU nn.Linear computes for a non-contiguous bf16 input (matmul, then add_) - the reference's k_task / v_task
   * projections of the strided h_t slice at batch size > 1 (action_heads.py:57, 118, 366-367); pinned by the reference-run
   * fixtures tests/golden/head_bf16_*.npz.  0 = the fused single rounding bf16(alpha A.B^T + bias). */
  int bias_post_round;
  /* optional (round 2; BASELINE configs[4]'s "fp8 MFMA weight path" - the reference has no fp8 code, parity unpinned):
   * fp8 = 1: A and B point at OCP e4m3 bytes (lda / ldb in elements, multiples of 16; K % 128 == 0), a_scale[M] / b_scale[N]
   * are the per-row dequantisation factors (vla_quant_fp8_rows): C = epilogue(a_scale[m] b_scale[n] (A . B^T)[m, n]).
   * Plain / activation / residual / SwiGLU-forward / rotate_half epilogues; no split-K, no batch.  With a K extension: see A2. */
  int fp8; const float* a_scale; const float* b_scale;
  /* optional K extension (ABI 2; K2 = 0: off): the contraction continues over a second operand pair,
   * C = epilogue(A . B^T + A2 . B2^T) with A2 bf16 [M, K2] (row stride lda2) and B2 bf16 [N, K2] (ldb2), K2 % 64 == 0, in ONE fp32
   * accumulator.  A LoRA-wrapped Linear (peft, vla-scripts/finetune.py:832-844) as one product: y = x W^T + (2 x A^T) B^T with the
   * base GEMM's epilogue intact; its backward dx = dy W + dt A likewise.  batch 1, no split-K / interleaved RoPE.  With fp8 = 1 (ABI 5)
   * the base operands A / B are e4m3 bytes and A2 / B2 stay bf16: C = epilogue(a_scale[m] b_scale[n] (A . B^T)[m, n] + (A2 . B2^T)[m, n]) -
   * the frozen base weights of a LoRA fine-tune on the fp8 MFMA path (BASELINE configs[4]); SwiGLU backward is allowed in this form. */
  const void* A2; const void* B2; int K2, lda2, ldb2;
} vla_gemm_desc;

/* 1 when every operand row a 256-row tile of this problem can touch lies below 4 GiB from its base (the 256 x 256 kernel keeps
 * 32-bit per-lane byte offsets; larger operands are routed to the 128-row kernel, which uses 64-bit pointers).  Host arithmetic. */
/* 1 when vla_gemm_bf16_nt would run this descriptor on the 256 x 256 kernel (the routing is shape- and device-dependent). */
int vla_gemm_uses_256(const vla_gemm_desc* desc);
/* Process-wide hint for vla_gemm_bf16_nt, read when a product is launched (= when a hipGraph is captured): on = 1 says the following
 * products run on an otherwise idle chip and are bound by the latency of a launch, not by throughput - the batch-1 predict_action of
 * modeling_prismatic.py:892-972 / openvla_utils.py:737-825 (every product there is at most one workgroup per CU and took 17-22 us
 * whatever its size: one K-tile in flight per workgroup).  Under the hint: launches of at most one workgroup per CU use a deeper operand
 * ring (bit-identical results); products of at most 512 rows run on gemm_skinny.hip's small tiles with the contraction split over a
 * workgroup's four waves, and vla_attn_fwd launches of fewer workgroups than half the CUs split the KEYS over the waves - both the same
 * arithmetic in another fp32 association (results agree to rounding; in a training step a product's bits must not depend on the batch
 * size, hence the hint).  on = 0 clears it, on < 0 only queries.  Returns the previous value. */
int vla_gemm_latency_hint(int on);
int vla_gemm256_extent_ok(const vla_gemm_desc* desc /* host */);

/* ---------------------------------------------------------------- TN GEMM (weight gradients) */
typedef struct vla_gemm_tn_desc {
  const void* A;    /* [batch][M, N1] bf16, row stride lda: dY (rows = tokens, columns = output features) */
  const void* B;    /* [batch][M, N2] bf16, row stride ldb: X  (rows = tokens, columns = input features) */
  void* C;          /* [batch][N1, N2] bf16, row stride ldc: C = bf16(alpha sum_m A[m, :]^T B[m, :])  (+ R: bf16(bf16(..) + R)) */
  const void* R;    /* optional addend [batch][N1, N2] (ldr); may alias C (gradient accumulation) */
  int M, N1, N2, lda, ldb, ldc, ldr, batch;
  long long sA, sB, sC, sR;            /* batch strides (elements) */
  float alpha;                         /* 0 -> 1 */
  /* contraction-row groups (0 = plain): row m of A lives at (m / a_group) * a_group_stride + (m % a_group) * lda, a_group % 64 == 0
   * (likewise B): the first Kt rows of every sequence of a [B, S, D] hidden state, read in place */
  int a_group, b_group; long long a_group_stride, b_group_stride;
  /* column groups on A (0 = plain): column c of the product's N1 axis is column (c / g) * stride + c % g of A (g % 8 == 0): the
   * gate (or up) columns of a gate/up-interleaved dY */
  int a_col_group, a_col_group_stride;
  /* split of the contraction (0/1 = off): `split` slices of M as extra blocks, fp32 planes in ws [batch, split, N1, N2] (device,
   * no initialisation needed), summed by a second kernel - for few-tile long-M products (LoRA pairs) */
  int split; float* ws;
} vla_gemm_tn_desc;
/* dW = dY^T . X for every trainable nn.Linear - torch.autograd's weight gradient behind loss.backward() (vla-scripts/finetune.py:
 * 1039-1042): the action head's Linears (action_heads.py:111-121, 337-410), the LoRA pairs (finetune.py:832-844), every VLM
 * Linear in the full fine-tune (:846-849) - on dY and X as the backward / forward left them (no operand transposes).
 * N1 % 8 == 0, N2 % 8 == 0, lda / ldb % 8 == 0; M arbitrary. */
int vla_gemm_bf16_tn(void* stream, const vla_gemm_tn_desc* desc /* host */);
/* GROUPED form: `count` (1 .. 48) independent TN products as ONE launch over the concatenated tile list - the weight gradients of
 * several Linears / layers, which torch.autograd computes one GEMM at a time (vla-scripts/finetune.py:1039-1042).  A single dW
 * product is 49 ... 532 tiles on a chip with 512 workgroup slots and leaves up to half of it idle in its tail round; a whole
 * backward piece's products together run at the granularity of their common tile list.  Plain problems only (batch 1, no split,
 * no addend, no row groups; column groups on A allowed).  descs: host array; the problem table travels in the kernel arguments. */
int vla_gemm_bf16_tn_grouped(void* stream, const vla_gemm_tn_desc* descs /* host array */, int count);

/* Row-wise dynamic fp8 quantisation: q[r, :] = e4m3(x[r, :] * 448 / amax_r) (round to nearest even, saturating), scale[r] =
 * amax_r / 448 (1 for an all-zero row).  x bf16 [rows, cols] (ldx), q bytes [rows, cols] (ldq, % 16 == 0), cols % 8 == 0. */
int vla_quant_fp8_rows(void* stream, const void* x, void* q, float* scale, int rows, int cols, int ldx, int ldq);
/* The norms with that quantisation fused behind them (the row is still in registers): q8 / qscale as vla_quant_fp8_rows of the
 * bf16 output; y may be null when only the fp8 form is consumed (frozen Linears: the backward needs x and the statistics). */
int vla_rmsnorm_fwd_q8(void* stream, const void* x, const void* w, void* y, float* rstd, void* q8, float* qscale, int rows, int cols, int ldq,
                       float eps);
int vla_layernorm_fwd_q8(void* stream, const void* x, const void* w, const void* b, void* y, float* stats, void* q8, float* qscale, int rows,
                         int cols, int ldx, int ldy, int ldq, float eps);

/* C = epilogue(A . B^T).  Replaces nn.Linear forward and, with pre-transposed operands, its dX / dW products:
 * timm ViT qkv/proj/mlp (modeling_prismatic.py:120-144), PrismaticProjector (:261-273), Qwen2 q/k/v/o/gate/up/down
 * (:644-655), ProprioProjector (projectors.py:19-24), MLPResNet / MLPResNetBlock(_Pro) Linears
 * (action_heads.py:111-121, 337-410).  K % 64 == 0; lda, ldb % 8 == 0; M/N edges are handled.
 * VLA_ACT_SWIGLU: B rows interleaved in groups of 16 (rows 32t..32t+15 = gate[16t..], 32t+16.. = up[16t..]);
 * C (optional) receives the interleaved pre-activations, C2 the product.
 * VLA_ACT_SWIGLU_BWD: the product A.B^T is dH [M, N]; R = the forward's interleaved pre-activations GU [M, 2N];
 * C receives dGU [M, 2N] (same interleave) - the SwiGLU backward fused into the dH GEMM, dH itself is never stored. */
int vla_gemm_bf16_nt(void* stream, const vla_gemm_desc* desc /* host */);

/* out[b][c, r] = in[b][r, c]; rows/cols of `in`; out has leading dim ldo >= rows (tail NOT touched: pre-zero it).
 * Used to build the K-contiguous operands of dX (W^T) and dW (dY^T, X^T) products for vla_gemm_bf16_nt. */
int vla_transpose_bf16(void* stream, const void* in, void* out, int rows, int cols, int ldi, int ldo, int batch,
                       long long s_in, long long s_out);

/* ---------------------------------------------------------------- norms */
/* nn.LayerNorm forward over the last dim (timm Block norm1/norm2 eps 1e-6; action_heads.py:96,108,306 eps 1e-5).
 * x,y bf16 [rows, cols]; w,b bf16 [cols]; stats (optional) f32 [rows,2] = (mean, rstd) for the backward. */
int vla_layernorm_fwd(void* stream, const void* x, const void* w, const void* b, void* y, float* stats, int rows,
                      int cols, int ldx, int ldy, float eps);
/* dx (bf16, may be NULL) and dw/db (f32 [cols], accumulated with +=, may be NULL) of the above. */
int vla_layernorm_bwd(void* stream, const void* dy, const void* x, const void* w, const float* stats, void* dx,
                      float* dw, float* db, int rows, int cols, int ldx, int lddy, int lddx);
/* Qwen2RMSNorm: y = bf16(w * bf16(x * rsqrt(mean(x^2)+eps))) (transformers Qwen2RMSNorm.forward; call site
 * modeling_prismatic.py:644).  rstd (optional) f32 [rows]. */
int vla_rmsnorm_fwd(void* stream, const void* x, const void* w, void* y, float* rstd, int rows, int cols, float eps);
/* dx = rmsnorm backward (+ optional residual-stream gradient add: dx += dres).  No dw (frozen LLM).
 * dy/dres/dx are compact [rows, cols]; x_group > 0: row r of x / rstd is row (r / x_group) * x_group_rows + x_row0 +
 * r % x_group of the forward's tensors (the live-row window [x_row0, x_row0 + x_group) of every sequence). */
int vla_rmsnorm_bwd(void* stream, const void* dy, const void* x, const void* w, const float* rstd, const void* dres,
                    void* dx, int rows, int cols, int x_group, int x_group_rows, int x_row0);

/* Qwen2RMSNorm weight gradient: dw[c] += sum_r dy[r, c] * bf16(x[r, c] * rstd[r]) (f32 accumulator).  Needed once the LLM
 * trains (full fine-tune, vla-scripts/finetune.py:846-849; LoRA leaves the norms frozen). */
int vla_rmsnorm_dw(void* stream, const void* dy, const void* x, const float* rstd, float* dw, int rows, int cols);

/* ---------------------------------------------------------------- attention (MFMA, flash-style) */
typedef struct vla_attn_desc {
  const void* q; const void* k; const void* v;   /* bf16; element (b, s, h, d) at b*sb + s*ss + h*dh + d */
  void* o;                                       /* bf16 [B, Sq, Hq, dh] same addressing with o strides */
  float* lse;                                    /* f32 [B, Hq, Sq] log-sum-exp (natural log) or NULL */
  const unsigned char* kmask;                    /* [B, Sk] 1 = key allowed, or NULL */
  long long q_sb, k_sb, v_sb, o_sb;              /* batch strides */
  int q_ss, k_ss, v_ss, o_ss;                    /* sequence strides */
  int B, Sq, Sk, Hq, Hkv, dh, causal;
  float scale;
  /* backward only */
  const void* dout; void* dq; void* dk; void* dv; float* delta; /* delta f32 [B,Hq,Sq] workspace */
  long long do_sb, dq_sb, dk_sb, dv_sb; int do_ss, dq_ss, dk_ss, dv_ss;
  /* backward, optional: q/k were produced by rotate_half RoPE (tables f32 [S, dh/2], dh 64 or 128, position = sequence index):
   * dq/dk are returned already mapped through its transpose, i.e. as gradients of the PRE-rotation projections */
  const float* rope_cos; const float* rope_sin;
  /* optional window (all 0 = plain): query i sits at sequence position q_off + i (causal masking and RoPE use that
   * position; q/o/dout/dq point at the first live query row); lse is f32 [B, Hq, lse_hs] (0 -> Sq) indexed by i;
   * backward: dK/dV are produced for keys >= dkv_k0 only, stored at row key - dkv_k0 of dk/dv.  With q_off = dkv_k0 = r0
   * the backward costs only what the rows >= r0 of a causal sequence need (live-row backward of a frozen LLM). */
  int q_off, dkv_k0, lse_hs;
} vla_attn_desc;

/* softmax(scale * Q K^T + mask) V, GQA (Hq % Hkv == 0), causal and/or key-padding mask, dh in {64,72,112,128}.
 * Replaces F.scaled_dot_product_attention in timm Attention and flash-attn / eager attention in Qwen2Attention. */
int vla_attn_fwd(void* stream, const vla_attn_desc* desc /* host */);
/* dQ, dK, dV of the above (recompute from q,k,v,o,lse).  Replaces flash-attn backward / autograd of eager attention. */
int vla_attn_bwd(void* stream, const vla_attn_desc* desc /* host */);

/* ---------------------------------------------------------------- rotary embeddings */
/* HF rotate_half RoPE in place on x[rows = B*S, nheads*dh] (row stride ldx): position = row % S;
 * cos/sin tables f32 [S, dh/2] holding bf16-rounded values.  sign=+1 forward, -1 backward (inverse rotation).
 * transformers apply_rotary_pos_emb; Qwen2 theta=1e6 (config.json text_config). */
int vla_rope_half(void* stream, void* x, const float* cos_t, const float* sin_t, int rows, int S, int nheads, int dh,
                  int ldx, int sign);
/* action_heads.py:125-146 apply_rope: pairs (2i,2i+1) with cos/sin = cat([f,f]) tables (f32 [T, dh]).
 * x[b, t, head, dh] rows = B*T, position = row % T.  mode 0 forward, 1 backward (transpose of the linear map). */
int vla_rope_interleaved(void* stream, void* x, const float* cos_t, const float* sin_t, int rows, int T, int nheads,
                         int dh, int ldx, int mode);

/* ---------------------------------------------------------------- input stage */
/* ToTensor + Normalize of PrismaticImageProcessor.apply_transform (processing_prismatic.py:128-145) for images already at
 * the model's input size: img u8 [B, H, W, 3] -> out[b, c0 + c, y, x] = ((img / 255) - mean[c]) / std[c]  (f32 math in
 * torchvision's order; stored bf16, or f32 if out_f32) inside a channel-stacked [B, Ctot, H, W] tensor.  mean3 / std3: host. */
int vla_image_normalize_u8(void* stream, const void* img, void* out, int B, int H, int W, int Ctot, int c0,
                           const float* mean3 /* host */, const float* std3 /* host */, int out_f32);
/* ActionTokenizer.__call__ (prismatic/vla/action_tokenizer.py:60-74, use_minivlm): ids[i] = tokenizer_len -
 * digitize(clip(actions[i], lo, hi), bins) with numpy semantics (bins: device f64 [nbins], increasing). */
int vla_action_tokenize(void* stream, const float* actions, const double* bins, long long* ids, long long n, int nbins,
                        float lo, float hi, long long tokenizer_len);

/* ---------------------------------------------------------------- glue */
/* timm PatchEmbed conv PxP/P as im2col: pixels [B, Ctot, H, W] (f32 if px_f32 else bf16), channels c0..c0+2 ->
 * cols bf16 [B*(H/P)*(W/P), ldo] with (c, py, px) ordering, zero-filled up to ldo.  (modeling_prismatic.py:229-230) */
int vla_im2col_patch(void* stream, const void* pixels, void* cols, int B, int Ctot, int c0, int H, int W, int P,
                     int ldo, int px_f32);
/* get_current_action_mask | get_next_actions_mask (train_utils.py:8-41) on int64 labels [B, L] shifted by
 * `shift` (0: labels, 1: labels[:,1:]).  qidx int32 [B, L-shift]: k for the k-th selected position, else -1;
 * pos int32 [B, 64]: selected positions (-1 padded); count int32 [B]. */
int vla_action_mask(void* stream, const long long* labels, int* qidx, int* pos, int* count, int B, int L, int shift);
/* Embedding gather + action-query splice + multimodal layout (modeling_prismatic.py:601, 418-454, 486-510):
 * out[b, 0] = tok 0, out[b, 1..Np] untouched (projector GEMM writes there), out[b, Np+j] = tok j (j>=1);
 * tok j = action_queries[qidx[b,j]] if qidx>=0 else table[ids[b,j]].  mm_mask u8 [B, S]. */
int vla_embed_splice(void* stream, const long long* ids, const unsigned char* attn_mask, const int* qidx,
                     const void* table, const void* action_queries, void* out, unsigned char* mm_mask, int B, int L,
                     int Np, int D, int vocab);
/* d action_queries[k] = sum_b dX[b, Np + pos[b,k]] (f32 [64, D]); backward of the splice. pos from shift=0 mask.
 * dx holds the rows >= row0 of every sequence: bf16 [B, S, D] with S = (sequence length - row0). */
int vla_action_query_grad(void* stream, const void* dx, const int* pos, float* dq, int B, int S, int Np, int D, int row0);
/* out[i, :] = in[idx[i], :] (idx == -2 -> row i is left untouched, any other idx < 0 -> zeros); bf16 rows of D elements. */
int vla_gather_rows(void* stream, const void* in, const int* idx, void* out, int n, int D, int ldi, int ldo);
/* out[idx[i], :] += in[i, :] (idx unique, idx<0 skipped). */
int vla_scatter_add_rows(void* stream, const void* in, const int* idx, void* out, int n, int D, int ldi, int ldo);
/* y = bf16(a + b) elementwise over n bf16 (n % 8 == 0) */
int vla_add_bf16(void* stream, const void* a, const void* b, void* y, long long n);
/* GELU(erf) forward y=gelu(x) / backward dx = dy*gelu'(x), bf16, n elements */
int vla_gelu_fwd(void* stream, const void* x, void* y, long long n);
int vla_gelu_bwd(void* stream, const void* dy, const void* x, void* dx, long long n);
/* relu backward through the OUTPUT: dx = dy * (y > 0) */
int vla_relu_bwd(void* stream, const void* dy, const void* y, void* dx, long long n);
/* SwiGLU backward: dH [M, I] and interleaved pre-activations GU [M, 2I] -> dGU [M, 2I] (same interleave). */
int vla_swiglu_bwd(void* stream, const void* dh, const void* gu, void* dgu, int M, int I);
/* h[M, I] = bf16(bf16(silu(gate)) * up) from the interleaved pre-activations gu[M, 2I] (same layout as VLA_ACT_SWIGLU): the
 * stand-alone form of the fused epilogue, used when LoRA deltas are added to the pre-activations first (finetune.py:832-844). */
int vla_swiglu_fwd(void* stream, const void* gu, void* h, int M, int I);
/* column sums of bf16 matrices [batch][rows, cols] into f32 out[batch][cols] (+=): bias gradients. */
int vla_colsum_bf16(void* stream, const void* x, float* out, int rows, int cols, int ldx, int batch, long long s_x,
                    long long s_out);
/* f32 -> bf16 / bf16 -> f32 casts */
int vla_cast_f32_bf16(void* stream, const float* x, void* y, long long n);
int vla_cast_bf16_f32(void* stream, const void* x, float* y, long long n);

/* Gradient of the embedding table (nn.Embedding backward, full fine-tune): grad_table[id] = sum of dx rows of the token
 * positions holding id (positions overwritten by an action query excluded: qidx >= 0); rows of ids that do not occur are NOT
 * touched (zero grad_table first).  dx [B, L + Np, D] = gradient w.r.t. inputs_embeds of the full sequence. */
int vla_embed_grad(void* stream, const void* dx, const long long* ids, const int* qidx, void* grad_table, int B, int L,
                   int Np, int D, int vocab);

/* One pass of Pillow's 8-bit resampler along the middle axis of a uint8 [outer, in_len, inner] array (horizontal: outer = B*H,
 * inner = 3; vertical: outer = B, inner = out_w*3): out = clip8((2^21 + sum_k in[lo+k] * coef[k]) >> 22).  bounds int32
 * [out_len, 2] = (lo, count), coefs int32 [out_len, ksize] - Pillow's bicubic taps in 22-bit fixed point.  Two calls = the resize
 * of PrismaticImageProcessor.apply_transform (processing_prismatic.py:128-145: TVF.resize(img, (224, 224), BICUBIC, antialias)). */
int vla_resample_u8(void* stream, const void* src, void* dst, long long outer, int in_len, int out_len, int inner,
                    const int* bounds, const int* coefs, int ksize);

/* Token cross-entropy of the native VLM path (prismatic/models/vlms/prismatic.py:469-481 -> HF shifted causal-LM loss):
 * logits bf16 [rows, V] (row stride ld_logits), shifted_labels int64 [rows] (the target of each row, -100 = ignore);
 * out[0] += sum over valid rows of (logsumexp(float(logits[row])) - logits[row, label]), out[1] += number of valid rows
 * (zero `out` first; loss = out[0] / out[1]). */
int vla_token_ce(void* stream, const void* logits, long long ld_logits, const long long* shifted_labels, int rows, int V,
                 float* loss_sum_and_count);

/* Backward of vla_token_ce w.r.t. the logits (autograd of the HF loss on ``logits.float()``; the native trainer of
 * prismatic/training/strategies/base_strategy.py:257-417 calls loss.backward()): valid rows get
 * bf16((softmax(float(logits[row])) - onehot(label)) * gscale / count), ignored rows zeros; count = loss_sum_and_count[1] of the
 * forward, read on the device.  dlogits may alias logits. */
int vla_token_ce_bwd(void* stream, const void* logits, long long ld_logits, const long long* shifted_labels, int rows, int V,
                     const float* loss_sum_and_count, float gscale, void* dlogits, long long ld_dlogits);

/* ---------------------------------------------------------------- host-glue replacements
 * The reference's training step strings its ops together with dozens of small ATen index / cast / copy kernels
 * (finetune.py:331-343, 396-409; action_heads.py:53-72; modeling_prismatic.py:499-508); these entry points do the same
 * glue for the native step so that no framework kernel sits between the hand-written ones. */
/* Strided 2-D copy with optional cast: dst[r, 0:cols] = src[r % src_mod (if src_mod > 0, else r), 0:cols]; dst row r lives at
 * (r / d_group) * d_group_stride + (r % d_group) * ld_dst when d_group > 0.  dtype codes: 0 = bf16, 1 = f32. */
int vla_copy2d(void* stream, const void* src, void* dst, long long rows, int cols, long long ld_src, long long ld_dst,
               int src_dtype, int dst_dtype, int src_mod, int d_group, long long d_group_stride);
/* dst[g][r][0:cols] = src[g][r][0:cols] for g < groups, r < rows (bf16; strides in elements): moves between the fused feature
 * buffer [B, n_img * 256, sum of backbone widths] and the per-backbone [n_img * B, tokens, d] layouts (modeling_prismatic.py:
 * 226-237 cat / split), drops or inserts the DINOv2 prefix tokens, compacts the patch rows of d inputs_embeds. */
int vla_copy_rows3d(void* stream, const void* src, void* dst, int groups, int rows, int cols, long long src_group_stride,
                    long long src_row_stride, long long dst_group_stride, long long dst_row_stride);
/* LayerScale as a parameter (timm LayerScale patched at modeling_prismatic.py:58-66: `x * self.scale_factor`) fused with the
 * block's residual add: out = bf16(x + bf16(a * ls)); a, x, out bf16 [rows, cols] contiguous (out may alias x), ls bf16 [cols]. */
int vla_layerscale_fwd(void* stream, const void* a, const void* ls, const void* x, void* out, long long rows, int cols);
/* its backward: da = bf16(dy * ls); dls (f32 [cols], +=, may be NULL when the scale is frozen - LoRA) += sum_r dy[r, :] * a[r, :]. */
int vla_layerscale_bwd(void* stream, const void* dy, const void* a, const void* ls, void* da, float* dls, int rows, int cols);
/* Zero nbytes at ptr (16-B aligned) with a store kernel on the given stream (gradient accumulators, dHS); graph-capturable. */
int vla_fill_zero(void* stream, void* ptr, long long nbytes);
/* Dropout on the input of a LoRA branch - peft's Linear.forward `lora_B(lora_A(lora_dropout(x))) * scaling` with `--lora_dropout > 0`
 * (vla-scripts/finetune.py:110, 832-840).  y[r, c] = keep(r, c) ? bf16(x[r, c] / (1 - p)) : 0 on bf16 [rows, cols] tensors (row strides
 * ldx / ldy in elements, cols and strides multiples of 8).  keep() is a counter-based hash of (seed, *step, r * cols + c): `step` points
 * at a DEVICE int (may be NULL = 0) so that a captured step draws a fresh mask on every replay (vla_inc_i32 bumps it inside the graph);
 * the backward regenerates the same mask from the same (seed, step): nothing is stored.  torch's Philox stream is not reproduced
 * (parity with a peft run is statistical; the tests hand the generated mask to the oracle). */
int vla_dropout_bf16(void* stream, const void* x, void* y, long long rows, int cols, long long ldx, long long ldy, float p,
                     unsigned long long seed, const int* step);
/* its backward, accumulated: dx[r, c] = bf16(dx[r, c] + (keep(r, c) ? bf16(u[r, c] / (1 - p)) : 0)) - the LoRA branch's share of a
 * wrapped Linear's input gradient (u = dt A) added to the base product's. */
int vla_dropout_bwd_add_bf16(void* stream, const void* u, void* dx, long long rows, int cols, long long ldu, long long lddx, float p,
                             unsigned long long seed, const int* step);
int vla_inc_i32(void* stream, int* ptr);      /* *ptr += 1 on the stream (the dropout step counter; graph-capturable) */
/* Gather / scatter row indices of the 64 action-query hidden states (+ the proprio slot) for engine.Head, and the NaN guard of a
 * frozen live-row window: see the kernel comment in elementwise.hip (finetune.py:396-409 regroup, done by index). */
int vla_head_index_prep(void* stream, const int* pos1, const int* pos0, const int* cnt0, int* gather, int* scatter,
                        float* guard, int B, int S, int Np, int row0);
/* x[0:n] += *s (device scalar). */
int vla_add_scalar_f32(void* stream, float* x, const float* s, int n);

/* ---------------------------------------------------------------- action head attention (action_heads.py:337-410) */
typedef struct vla_head_attn_desc {
  const void* q;      /* [B, T, H*dh] bf16 (RoPE applied) */
  const void* k_self; const void* v_self;     /* [B, T, H*dh] */
  const void* k_adp;  const void* v_adp;      /* [B, Ka, H*dh] */
  const void* k_task; const void* v_task;     /* [B, Kt, H*dh] */
  const void* gate;   /* bf16 scalar gating_factor (device) */
  void* out;          /* [B, T, H*dh] */
  float* probs;       /* f32 [B, H, T, T+Ka+Kt] saved softmax (workspace for backward) */
  int B, T, Ka, Kt, H, dh;
  int ld_q, ld_self, ld_adp, ld_task, ld_out; /* row strides */
  int gate_on_adapter;                        /* 0: Pro (tanh(g) on task segment); 1: original block (on 3rd segment too) */
  /* backward */
  const void* dout; void* dq; void* dk_self; void* dv_self; void* dk_adp; void* dv_adp; void* dk_task; void* dv_task;
  float* dgate;       /* f32 scalar, += */
  /* backward, optional: tables f32 [>= max(T,Ka,Kt), dh] of vla_rope_interleaved; dq and the three dk are then
   * returned through the transpose of that RoPE map (positions restart per segment, action_heads.py:383-388) */
  const float* rope_cos; const float* rope_sin;
  /* backward, optional (ABI 3): f32 workspace of >= B*H*ceil((T+Ka+Kt)/32)*(T*dh + 1) floats.  With it the MFMA backward runs
   * in its tile-uniform form (one wave per (sample, head, 32-key tile); dq and dgate summed from per-tile partials in tile
   * order); without it (NULL / too small) the combined kernel of ABI 2 runs.  Contents are scratch. */
  float* ws; long long ws_floats;
  /* forward, optional (ABI 5): 1 = the softmax weights are rounded to bf16 AFTER normalisation, P = bf16(exp(s - max) / sum), as ATen's
   * bf16 softmax emits them (action_heads.py:397: attn_weights = softmax(attn_scores) on a bf16 module) - two passes over the keys
   * (statistics, then P.V).  0 = one flash-style pass (un-normalised bf16 weights, one division at the end): faster, and further from
   * the reference's bf16 run by the rounding of 8 x 585 weights. */
  int ref_softmax;
} vla_head_attn_desc;
int vla_head_attn_fwd(void* stream, const vla_head_attn_desc* desc /* host */);
int vla_head_attn_bwd(void* stream, const vla_head_attn_desc* desc /* host */);

/* ---------------------------------------------------------------- loss + optimiser */
/* torch.nn.L1Loss (finetune.py:418): loss[0] = mean|pred - target|, loss[1] = curr (chunk 0), loss[2] = next;
 * dpred = sign(pred-target) * gscale / n (bf16).  pred/target bf16 [B, C, Da]. */
int vla_l1_loss(void* stream, const void* pred, const void* target, float* loss3, void* dpred, int B, int C, int Da,
                float gscale);
/* torch.optim.AdamW step on bf16 params/grads/states with bf16 rounding after every elementwise op, as the
 * foreach implementation does (finetune.py:910, 1079).  g may be f32 (g_f32=1: rounded to bf16 first). */
int vla_adamw_bf16(void* stream, void* p, const void* g, void* m, void* v, long long n, double lr, double beta1,
                   double beta2, double eps, double wd, int step, int g_f32, float gscale);

#ifdef __cplusplus
}
#endif
#endif /* VLA_NATIVE_H */
