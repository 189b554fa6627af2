/* libhalva_hip.so - C ABI of the hand-written gfx950 (MI355X / CDNA4) kernels behind the HALVA DPA step.
 *
 * The reference (pritamqu/HALVA) is pure Python and owns no FFI; its only optimisation seam is the
 * attention monkey-patch (reference llava/train/llama_flash_attn_monkey_patch.py:16-115).  This header is
 * therefore the drop-in boundary defined by this build: each entry point names the reference call it
 * replaces (file:line, relative to the reference tree).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain C: raw device pointers + explicit sizes; no torch types.  The caller owns every buffer; no entry
 *     point allocates, frees or synchronises.  All work is enqueued on `stream` (a hipStream_t, NULL = default).
 *   - return value: 0 on success, a negative HALVA_ERR_* otherwise; halva_last_error() gives the text
 *     (thread-local).  Entry points are re-entrant and hold no mutable global state.
 *   - dtype arguments use halva_dtype.  "rows" are tokens (S*T); matrices are row-major.
 *   - sequences are described by seq_start[S], seq_len[S] (int32): the valid tokens of sequence s are
 *     [seq_start[s], seq_start[s]+seq_len[s]) on its T-long padded row (right padding: start 0).
 */
#ifndef HALVA_HIP_H
#define HALVA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HALVA_ABI_VERSION 1

typedef enum { HALVA_BF16 = 0, HALVA_F32 = 1 } halva_dtype;

#define HALVA_OK 0
#define HALVA_ERR_INVALID_ARG (-1)
#define HALVA_ERR_LAUNCH (-2)
#define HALVA_ERR_UNSUPPORTED (-3)

int halva_abi_version(void);
const char* halva_last_error(void);

/* ---- Llama RMSNorm.  replaces LlamaRMSNorm.forward (llava/model/language_model/modelling_llama.py:65-70)
 * y = bf16(w * x_f32 * rsqrt(mean(x_f32^2) + eps)) - the module's fp32 value, rounded once (its intermediate cast to the input
 * dtype is a no-op on the fp32 CPU path that parity is defined against); rstd[rows] (f32) is saved for the backward.
 * With HALVA_RMSNORM_MODULE_ROUNDING=1 in the environment (read once per process) the forward instead rounds x * rstd to bf16 before
 * the multiplication with w - bit for bit what the module computes on a bf16 device - for comparisons with the reference's GPU numerics.
 * bwd gives dx only (norm weights are frozen on the LoRA DPA path; llava/train/train_halva.py:1085-1101). */
int halva_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int d, float eps, void* stream);
int halva_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, void* dx, int64_t rows, int d,
                      void* stream);
/* same with an explicit row stride (elements) for y / dy: lets the normalised rows land directly in the left columns of the
 * [rows, d + G*r] LoRA operand buffer of the next projection (no concat copy). */
int halva_rmsnorm_fwd_ld(const void* x, const void* w, void* y, int64_t ldy, float* rstd, int64_t rows, int d, float eps,
                         void* stream);
int halva_rmsnorm_bwd_ld(const void* dy, int64_t lddy, const void* x, const void* w, const float* rstd, void* dx, int64_t rows,
                         int d, void* stream);
/* the forward of the decoder layer's residual fork: as halva_rmsnorm_fwd_ld, and the row is also written to x_copy [rows, d]
 * (may be NULL) - the buffer the residual connection then accumulates the block's output onto IN PLACE
 * (`hidden_states = residual + ...`, modelling_llama.py:395-417, as a beta = 1 GEMM).  Saves the pass that would otherwise copy the
 * residual into the GEMM's output buffer (hipMemcpy D2D of [rows, d] per residual add: 0.8 % of the 7B step). */
int halva_rmsnorm_fwd_fork_ld(const void* x, const void* w, void* y, int64_t ldy, float* rstd, void* x_copy, int64_t rows, int d,
                              float eps, void* stream);
/* the decoder layer's residual fork in one pass: dx = rmsnorm_bwd(dy) + dres, where dres [rows, d] is the gradient that reaches
 * the same hidden state through the residual connection (modelling_llama.py:395-417: `hidden_states = residual + ...`).  Saves
 * the separate read-read-write pass of autograd's accumulation.  dres == NULL: plain halva_rmsnorm_bwd_ld.  dx may alias dres. */
int halva_rmsnorm_bwd_res_ld(const void* dy, int64_t lddy, const void* x, const void* w, const float* rstd, const void* dres,
                             void* dx, int64_t rows, int d, void* stream);

/* ---- RoPE, in place on the q and k thirds of a packed qkv buffer [rows, 3, H, D] (bf16).
 * replaces apply_rotary_pos_emb (modelling_llama.py:154-169) as called from
 * llava/train/llama_flash_attn_monkey_patch.py:51-55.  cos/sin: [max_pos, D/2] bf16 (the table halves are equal,
 * modelling_llama.py:98-100).  pos: int32 [rows] or NULL for pos = row % T.  inverse != 0 applies the transpose
 * rotation (the backward). */
int halva_rope_qk(void* qkv, const void* cos, const void* sin, const int32_t* pos, int64_t rows, int T, int H, int D,
                  int max_pos, int inverse, void* stream);
/* The same with the positions of branch-packed rows [prefix | A | pad | B] implied by the branch points instead of a table: row t of sequence s
 * (rows = S * T) sits at position t, and at br_a[s] + (t - br_b[s]) once t >= br_b[s] (branch B continues from the prefix: what
 * halva_amd/splice.py:pack_pairs writes into `pos` for every row that carries a token).  br_a / br_b: int32 [S] or both NULL (= pos NULL above). */
int halva_rope_qk_branch(void* qkv, const void* cos, const void* sin, const int32_t* br_a, const int32_t* br_b, int64_t rows, int T, int H,
                         int D, int max_pos, int inverse, void* stream);

/* ---- CLIP's activation.  replaces `input * torch.sigmoid(1.702 * input)` (transformers QuickGELUActivation, reached through
 * llava/model/multimodal_encoder/clip_encoder.py:46 -> CLIPMLP) of the frozen tower: one pass instead of three element-wise kernels, with the
 * same three bf16 roundings.  n elements (multiple of 8), out may be x. */
int halva_quick_gelu(const void* x, void* out, int64_t n, void* stream);

/* ---- SwiGLU.  replaces act_fn(gate_proj(x)) * up_proj(x) (modelling_llama.py:197).  gu = [rows, 2F] (gate | up). */
int halva_swiglu_fwd(const void* gu, void* out, int64_t rows, int F, void* stream);
int halva_swiglu_bwd(const void* dout, const void* gu, void* dgu, int64_t rows, int F, void* stream);
int halva_swiglu_fwd_ld(const void* gu, void* out, int64_t ldo, int64_t rows, int F, void* stream);
int halva_swiglu_bwd_ld(const void* dout, int64_t lddo, const void* gu, void* dgu, int64_t rows, int F, void* stream);

/* ---- causal self-attention on right/left padded rows, bf16, head_dim 128 (Llama) - THE headline kernel.
 * replaces flash_attn_varlen_qkvpacked_func(qkv, cu_q_lens, max_s, 0.0, softmax_scale=None, causal=True)
 * together with unpad_input / pad_input (llava/train/llama_flash_attn_monkey_patch.py:71-91).
 * qkv: [S, T, 3, H, D] packed (RoPE already applied); out: [S, T, H, D]; lse: [S, H, T] f32 (natural log).
 * Padded query rows get zeros (pad_input semantics).  scale = 1/sqrt(D) when <= 0.
 * bwd: dqkv [S, T, 3, H, D] bf16 is fully written (zeros on padded rows); delta_ws: [S, H, T] f32 scratch owned by the
 * caller (written by the dQ launch, read by the dK/dV launch).  dq_ws is IGNORED (may be NULL): the argument survives from an
 * atomics-based dQ design that was never shipped; dQ has its own kernel and needs no scratch. */
int halva_sdpa_causal_fwd(const void* qkv, void* out, float* lse, const int32_t* seq_start, const int32_t* seq_len,
                          int S, int T, int H, int D, float scale, void* stream);
int halva_sdpa_causal_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                          float* delta_ws, float* dq_ws, const int32_t* seq_start, const int32_t* seq_len, int S, int T,
                          int H, int D, float scale, void* stream);
/* same with explicit row strides (elements) of out / dout */
int halva_sdpa_causal_fwd_ld(const void* qkv, void* out, int64_t ld_out, float* lse, const int32_t* seq_start,
                             const int32_t* seq_len, int S, int T, int H, int D, float scale, void* stream);
int halva_sdpa_causal_bwd_ld(const void* qkv, const void* out, int64_t ld_out, const void* dout, int64_t ld_dout,
                             const float* lse, void* dqkv, float* delta_ws, float* dq_ws, const int32_t* seq_start,
                             const int32_t* seq_len, int S, int T, int H, int D, float scale, void* stream);

/* Branched causal attention: one packed sequence [prefix | A | pad | B] per row; rows >= br_b[s] (branch B) do not attend to
 * rows [br_a[s], br_b[s]) (branch A and the padding); everything else is causal.  Lets the correct and the hallucinated
 * response of a pair share ONE forward/backward over their common prefix (image + prompt + identical start of the response) -
 * the reference runs the prefix twice, once per row of the concatenated batch (llava/train/halva_trainer.py:434-470).
 * br_a / br_b: int32 [S] row indices, or both NULL = plain causal (the entries above).  Contract: br_b[s] is a multiple of 64
 * (no 32-row strip / 64-key tile straddles it; rows in [br_a + len(A), br_b) are padding whose output is unspecified) and
 * seq_start[s] == 0 for a branched sequence; br_a[s] = br_b[s] >= seq_len[s] marks a sequence without a branch.  RoPE
 * positions of branch B restart at br_a (halva_rope_qk's `pos` argument). */
int halva_sdpa_branch_fwd(const void* qkv, void* out, int64_t ld_out, float* lse, const int32_t* seq_start, const int32_t* seq_len,
                          const int32_t* br_a, const int32_t* br_b, int S, int T, int H, int D, float scale, void* stream);
int halva_sdpa_branch_bwd(const void* qkv, const void* out, int64_t ld_out, const void* dout, int64_t ld_dout, const float* lse,
                          void* dqkv, float* delta_ws, const int32_t* seq_start, const int32_t* seq_len, const int32_t* br_a,
                          const int32_t* br_b, int S, int T, int H, int D, float scale, void* stream);
/* The same backward with a caller-owned workspace for dS = P o (dP - delta) (bf16, halva_sdpa_bwd_ws_bytes(S, T, H, D) bytes; 0 for
 * head dims the workspace path does not serve).  With it the backward forms dS once - three launches: delta (+ zeros into dq of padded
 * rows), dK/dV (which stores dS in its register layout), dQ = scale * dS K (which reads it back, HBM-bound) - five matrix products per
 * (query, key) tile pair instead of the seven of the split backward above (S and dP are otherwise formed in both kernels).  Same
 * results to bf16 rounding of dS (which the split backward applies as well before its dQ/dK products); bitwise reproducible.
 * ds_ws == NULL: identical to halva_sdpa_branch_bwd.  The workspace holds no state between calls.
 * Behind the dS region the workspace also carries the row statistics of the dK/dV kernel of this path (sdpa_bwd_dkv3, one wave per SIMD, its
 * steps in generated inline-asm blocks - halva_amd/csrc/sdpa_dkv3.h), written by the delta pass: per (sequence, head) ceil(T / 64) records
 * of 512 bytes, one per 64-row step in sequence coordinates, [lse * log2(e) x 64][-delta x 64] f32 (round 4; round 3 kept two [S, H, T]
 * arrays and wrote -delta to delta_ws, which this path now leaves untouched) - one LDS-DMA request per step fetches both, the dP chain
 * starts from -delta; and 1 KiB of work-queue counters (zeroed by the delta pass of each call): that kernel runs as one persistent workgroup per
 * CU drawing (sequence, head, key block) items from a queue per XCD.  The workspace still holds no state between calls, but one workspace
 * serves ONE call at a time.  HALVA_SDPA_DKV3=0 selects the two-role kernel of rounds 1-2 instead (same results up to bf16 rounding of P
 * before dZ). */
int64_t halva_sdpa_bwd_ws_bytes(int S, int T, int H, int D);
int halva_sdpa_branch_bwd_ws(const void* qkv, const void* out, int64_t ld_out, const void* dout, int64_t ld_dout, const float* lse,
                             void* dqkv, float* delta_ws, void* ds_ws, int64_t ds_ws_bytes, const int32_t* seq_start,
                             const int32_t* seq_len, const int32_t* br_a, const int32_t* br_b, int S, int T, int H, int D, float scale,
                             void* stream);
/* The same backward with the INVERSE RoPE of dq and dk applied on the way out: autograd of apply_rotary_pos_emb on the query / key gradients
 * (reference llava/model/language_model/modelling_llama.py:154-169 differentiated; llava/train/llama_flash_attn_monkey_patch.py:58-66 applies the
 * rotation in front of the attention) = halva_sdpa_branch_bwd_ws followed by halva_rope_qk_branch(dqkv, cos, sin, br_a, br_b, ..., inverse = 1),
 * with the same roundings (row rounded to bf16, rotated in fp32 with the bf16 table entries, rounded again) - but inside the store epilogues of
 * the dQ and dK/dV kernels where that kernel combination runs (head_dim 128 with the workspace: sdpa_bwd_dq2 + sdpa_bwd_dkv3), saving one launch
 * and a read-modify-write pass over dq and dk per call; any other combination ends with the rotation as its own launch.  Positions: row t of a
 * sequence sits at position t; rows of branch B (t >= br_b) at br_a + (t - br_b) - what halva_amd/splice.py:pack_pairs feeds the forward's
 * halva_rope_qk.  rope_cos / rope_sin: the [max_pos, D / 2] bf16 tables of halva_rope_qk; both NULL = halva_sdpa_branch_bwd_ws.
 * HALVA_ROPE_FUSED_BWD=0 (A/B switch): always the separate launch. */
int halva_sdpa_branch_bwd_rope(const void* qkv, const void* out, int64_t ld_out, const void* dout, int64_t ld_dout, const float* lse,
                               void* dqkv, float* delta_ws, void* ds_ws, int64_t ds_ws_bytes, const int32_t* seq_start,
                               const int32_t* seq_len, const int32_t* br_a, const int32_t* br_b, const void* rope_cos, const void* rope_sin,
                               int max_pos, int S, int T, int H, int D, float scale, void* stream);
/* ---- non-causal self-attention, bf16, head_dim 64, forward only (the CLIP tower runs under no_grad:
 * llava/model/multimodal_encoder/clip_encoder.py:37-49; replaces HF CLIPAttention's softmax(QK^T*scale)V).
 * qkv: [N, S, 3, H, D] packed; out [N, S, H, D]. */
int halva_sdpa_full_fwd(const void* qkv, void* out, int N, int S, int H, int D, float scale, void* stream);

/* ---- bf16 GEMM on MFMA with fused epilogue: C[M,N] = epi(A[M,K] @ B[N,K]^T + bias[N]).
 * replaces the mm_projector Linear/GELU/Linear (llava/model/multimodal_projector/builder.py:39-46) and,
 * with im2col addressing, CLIP's patch-embed Conv2d(3, d, k=14, s=14, bias=False) reached through
 * clip_encoder.py:46.  epilogue: 0 none, 1 GELU(erf).  trans_a / trans_b select the backward forms
 * (A given as [K,M] / B given as [K,N]).  bias may be NULL.  pre_act (optional) receives A B^T + bias before the
 * epilogue (needed by the GELU backward); accumulate != 0 adds into C (gradient accumulation, f32 or bf16). */
int halva_gemm_bf16(const void* A, const void* B, const void* bias, void* C, void* pre_act, int M, int N, int K,
                    int trans_a, int trans_b, int epilogue, halva_dtype out_dtype, int accumulate, void* stream);

/* ---- LoRA weight gradients: C[M, N] (f32) += alpha * A^T B with A [rows, M] and B [rows, N] bf16 column windows of wider row-major
 * buffers (row strides lda / ldb), i.e. what peft's lora_A / lora_B receive from autograd on the DPA path
 * (llava/train/train_halva.py:1085-1101 makes them the only trainable decoder tensors): dB_g = scale * dy_g^T (x A_g^T),
 * dA = (scale * dy B)^T x.  One side of the product is 128..384 wide, the contraction runs over all token rows, so the library
 * GEMM has 32-86 tiles for 256 CUs; here the rows are split into k-slabs (partials in ws, summed in a fixed order - no atomics).
 * ws: scratch of ws_floats >= M * N floats; the number of slabs is what fits, up to one resident
 * round of workgroups (512). */
int halva_wgrad_accumulate(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int M, int N, int64_t rows, float alpha,
                           float* ws, int64_t ws_floats, void* stream);
/* The same for the two to four products of ONE LoRA group at once (its A factor and its B factors: they become available together in the group's
 * backward - halva_amd/llama.py:_LoraGroupLinear, i.e. peft's LoRA linear backward as used by llava/train/train_halva.py:1085-1101): one launch of
 * the tile kernel and one of the reduction instead of two per product.  Results are BITWISE those of halva_wgrad_accumulate called once per item
 * with the same workspace (same k-slabs, same summation order); items that the batched kernel does not take (more than 4, shapes that are no
 * multiples of 128, partials that do not fit the workspace side by side) simply run one by one.  items: HOST array of n descriptors. */
typedef struct halva_wgrad_item {
    const void* A;      /* [rows, M] bf16 window, row stride lda */
    int64_t lda;
    const void* B;      /* [rows, N] bf16 window, row stride ldb */
    int64_t ldb;
    float* C;           /* [M, N] f32, += alpha * A^T B */
    int32_t M, N;
    int64_t rows;
    float alpha;
    int32_t reserved;
} halva_wgrad_item;
int halva_wgrad_accumulate_batch(int n, const halva_wgrad_item* items, float* ws, int64_t ws_floats, void* stream);
/* images [n, 3, hw, hw] bf16; weight_kp [d, Kp] bf16 = the conv weight flattened to [d, 3*p*p] and zero padded to
 * Kp (multiple of 8); col_ws: caller scratch [n * (hw/p)^2, Kp] bf16 -> out [n, (hw/p)^2, d] bf16 */
int halva_clip_patch_embed(const void* images, const void* weight_kp, void* col_ws, void* out, int n, int hw, int p, int d,
                           int Kp, void* stream);
/* Same with a bias and a 'valid' grid of floor(hw/p)^2 patches: SigLIP's Conv2d(3, d, k=14, s=14, padding="valid") of
 * the VILA path (vila/model/multimodal_encoder/siglip/modeling_siglip.py SiglipVisionEmbeddings, reached through
 * vila/model/multimodal_encoder/vision_encoder.py:136-140).  bias [d] bf16 or NULL. */
int halva_vit_patch_embed(const void* images, const void* weight_kp, const void* bias, void* col_ws, void* out, int n, int hw,
                          int p, int d, int Kp, void* stream);
/* dh[M,N] = dy[M,N] * gelu'(h[M,N]) (h = pre-activation), and column sums for the bias grads. */
int halva_gelu_bwd(const void* dy, const void* h, void* dh, int64_t M, int N, void* stream);
int halva_colsum(const void* x, float* out, int64_t M, int N, void* stream);
/* dst[c][r] = src[r][c], bf16, [rows x cols] with row strides ld_src / ld_dst in elements (rows, cols, strides multiples of 8; 16-byte
 * aligned pointers).  Serves the once-per-optimizer-step refresh of the transposed, LoRA-merged weight copy the decoder's dgrad GEMMs read
 * (halva_amd/llama.py:LoraGroup.refresh_tail; the reference has no counterpart - peft keeps W, A, B apart and autograd forms dx from each):
 * `torch.addmm(W.t(), ...)` spent 0.45 ms per group copying the strided W^T with the framework's element-wise copy kernel. */
int halva_transpose_bf16(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int rows, int cols, void* stream);

/* ---- vision-side row kernels (VILA path).
 * LayerNorm over rows of [rows, d] bf16 (d % 8 == 0, d <= 8192): the nn.LayerNorm of mlp_downsample
 * (vila/model/multimodal_projector/base_projector.py:78) and of the CLIP / SigLIP encoder blocks.
 * stats (optional) receives (mean, rstd) per row as f32 [rows, 2] for the parameter gradients:
 * dw[j] += sum_r dy[r,j] * xhat[r,j], db[j] += sum_r dy[r,j] (f32 accumulators; the input carries no gradient
 * because the vision tower is frozen, src_vila/halva_vila_13b.sh:44).
 * downsample2x2: DownSampleBlock.flat_square (base_projector.py:33-54): x [n, g*g, c] -> out [n, G*G, 4c], G = ceil(g/2),
 * odd grids zero padded, token order and channel order exactly as the reference's view/permute chain. */
int halva_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* stats, int64_t rows, int d, float eps,
                        void* stream);
int halva_layernorm_bwd_params(const void* dy, const void* x, const float* stats, float* dw, float* db, int64_t rows, int d,
                               void* stream);
int halva_downsample2x2(const void* x, void* out, int n, int g, int c, void* stream);

/* ---- image preprocessing on the GPU.  replaces, for a batch of DECODED uint8 RGB images, the per-sample
 * expand2square + processor.preprocess of HallDataset.__getitem__ (llava/train/train_halva.py:735-751: PIL paste on a
 * mean-coloured square, CLIPImageProcessor resize(shortest_edge, BICUBIC) / center_crop / rescale / normalize) and the
 * VILA twin's `image.resize((S, S))` + SiglipImageProcessor (vila/mm_utils.py:150-193).  Pillow's 8-bit resample arithmetic
 * is reproduced exactly (uint8 results bit-exact; the float output is a per-channel 256-entry table built by the host from
 * rescale_factor / mean / std, so it is bit-exact too).  One descriptor per image; coefficient / bounds tables are the
 * host's precompute_coeffs (22-bit fixed point), shared between images of equal geometry.
 * tmp: uint8 scratch for the horizontal pass, out: [n_images, 3, crop_h, crop_w] bf16 or f32. */
typedef struct {
    int64_t src_off;              /* byte offset of the [src_h, src_w, 3] image in src_pack */
    int64_t tmp_off;              /* byte offset of the [tmp_rows, out_w, 3] horizontal-pass result in tmp */
    int32_t src_h, src_w;
    int32_t pad_x, pad_y;         /* position of the image on the (virtual) expand2square canvas */
    int32_t bg[3];                /* canvas fill colour */
    int32_t out_h, out_w;         /* size after the resize, before the crop */
    int32_t row0, tmp_rows;       /* canvas rows [row0, row0 + tmp_rows) feed the vertical pass */
    int32_t kh_off, ksize_h, bh_off;
    int32_t kv_off, ksize_v, bv_off;   /* vertical bounds are relative to row0 */
    int32_t crop_y, crop_x;
    int32_t need_h, need_v;
} HalvaImageDesc;
int halva_image_preprocess(const void* src_pack, const HalvaImageDesc* descs, const int32_t* coef, const int32_t* bounds,
                           const float* lut, void* tmp, void* out, int n_images, int max_tmp_pixels, int crop_h, int crop_w,
                           halva_dtype out_dtype, void* stream);

/* ---- splice gather.  replaces the per-sample python loop of prepare_inputs_labels_for_multimodal[_signed]
 * (llava/model/llava_arch.py:285-374): out[r] = embed[src[r]] if src[r] >= 0, feats[-src[r]-2] if src[r] <= -2,
 * zeros if src[r] == -1 (padding).  src is the host-computed index plan, rows = S*T. */
int halva_splice_rows(const void* embed, const void* feats, const int32_t* src, void* out, int64_t rows, int d,
                      void* stream);

/* ---- token log-prob.  replaces logits.log_softmax(-1) + gather (llava/train/halva_trainer.py:406-407).
 * logits [R, V] (row stride ld elements), target int32 [R] (already shifted; IGNORE_INDEX mapped to 0 by the
 * caller exactly as halva_trainer.py:406).  logp[R], lse[R] f32.
 * bwd: dlogits[r, v] = g[r] * (1[v == target[r]] - exp(logits[r,v] - lse[r])); may alias logits. */
int halva_token_logp_fwd(const void* logits, halva_dtype dt, int64_t ld, const int32_t* target, float* logp, float* lse,
                         int64_t R, int V, void* stream);
int halva_token_logp_bwd(const void* logits, halva_dtype dt, int64_t ld, const int32_t* target, const float* lse,
                         const float* g, void* dlogits, int64_t R, int V, void* stream);

/* ---- KL(ref || policy) per row.  replaces the softmax/log/product/mask/sum chain (halva_trainer.py:583-588).
 * kl[r] = w[r] * sum_v p_ref (log p_ref - log p_pol), computed with log-softmax algebra (identical where the
 * reference's softmax().log() is finite).  If dpol != NULL it also receives d kl[r] / d pol_logits[r, :]
 * = w[r] * (p_pol - p_ref) * gscale (may alias pol). */
int halva_kl_rows(const void* pol, const void* ref, halva_dtype dt, int64_t ld, const float* w, float* kl, void* dpol,
                  float gscale, int64_t R, int V, void* stream);

/* ---- phrase accumulation.  replaces accumulate_logps (halva_trainer.py:411-419) incl. the loss-mask multiply
 * (:556-557): acc[b, p] = sum_t logp[b,t] * (labels[b,t] != -100) * (signs[b,t] == slot_ids[p]).
 * slot_ids: sorted unique non-zero sign ids of the WHOLE half batch (host computed).  bwd scatters dacc back. */
int halva_phrase_sum_fwd(const float* logp, const int64_t* labels, const int64_t* signs, const int64_t* slot_ids, int P,
                         float* acc, int B, int T1, void* stream);
int halva_phrase_sum_bwd(const float* dacc, const int64_t* labels, const int64_t* signs, const int64_t* slot_ids, int P,
                         float* dlogp, int B, int T1, void* stream);

/* ---- hardware-layout probes used by the GPU tests (MFMA fragment maps, ds_read_b64_tr_b16). */
int halva_probe_layouts(int32_t* out, int n, void* stream);
/* ---- measurement aid (bench.py's clock trace; no reference counterpart): `blocks` one-wave workgroups each spin for `spin_ticks` ticks of
 * the constant 100 MHz counter and report out[4 b + {0,1,2,3}] = {shader cycles elapsed (s_memtime), 100 MHz ticks elapsed (s_memrealtime),
 * start tick, XCC id}: shader MHz = 100 * out[4b] / out[4b+1], the clock the chip holds while the step's kernels run beside the probe. */
int halva_clock_probe(uint64_t* out, int blocks, int spin_ticks, void* stream);
/* Host-side mirror (same inline function the kernels call) of how the forward and the dQ kernel pair the 256-row blocks of a sequence under
 * the causal mask: out[2k], out[2k+1] = the heavier and the lighter block of workgroup k, k < (nblk + 1) / 2 (equal for the middle block of
 * an odd count).  br_a / br_b as in halva_sdpa_branch_fwd (0x7fffffff for both: plain causal, i.e. nblk-1-k with k).  No GPU work: for
 * tests of the index arithmetic. */
int halva_sdpa_block_pairs(int nblk, int start, int br_a, int br_b, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* HALVA_HIP_H */
