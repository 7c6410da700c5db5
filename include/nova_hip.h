/*
 * nova_hip.h — C ABI of libnova_hip.so, the MI355X (gfx950) implementation of NOVA's
 * autoregressive point-set generation hot path.
 *
 * The reference (zailaiyiwan123/NOVA_pointcloud, `diffnext`) is pure Python and has no FFI of its
 * own; its kernel boundary is "whatever torch op a module's forward() calls". Each entry point
 * below therefore replaces a group of torch calls, cited as reference file:line (paths relative
 * to the reference's repository root). The drop-in `diffnext` package shipped in
 * nova_pointcloud_amd/diffnext binds these through ctypes (see INTEGRATION.md for the stub a
 * maintainer of the reference would add).
 *
 * Conventions
 *   - plain C: raw device pointers + sizes; no torch / HIP types in signatures. `stream` is a
 *     hipStream_t passed as void* (NULL = the legacy default stream). All work is enqueued on that
 *     stream; nothing synchronises, allocates or frees; buffers are caller-owned.
 *   - every function returns 0 on success or a negative nova_status; nova_last_error() returns a
 *     thread-local message for the last failure. Nothing throws.
 *   - dtype: storage type of activations and GEMM weights (NOVA_F32 = parity mode on exact-f32
 *     MFMA, NOVA_BF16 / NOVA_F16 = throughput mode on the bf16 / f16 MFMA forms - same kernels, same rate; f16 is the
 *     default precision of the reference's callers, scripts/app_nova_t2i.py:36,87-89. The fp8 GEMM mode and the attention backward take bf16 only;
 *     nova_row_norm_bwd and nova_act_fwd / nova_act_bwd take all three). Biases, LayerNorm affine parameters, RoPE
 *     tables, point coordinates, timesteps and sigmas are always float32; token ids are int64
 *     (torch.long, as the reference's pred_ids / prev_ids).
 *   - row-major everywhere; a "row" is one token's feature vector.
 */
#ifndef NOVA_HIP_H_
#define NOVA_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define NOVA_HIP_VERSION 401 /* 0.4.1: nova_decoder_denoise_echo (guidance renorm with any sampler step); 0.4.0 (round 4): the loader checks this number against its own; nova_attn_fwd_lse / nova_attn_bwd carry a key_limit pointer before the stream, nova_row_norm_bwd, nova_act_fwd, nova_act_bwd, nova_debug_drop_graphs (all added after 0.3.0 without a bump), nova_prof slots 7-9; 0.3.0: NOVA_F16 storage mode through every dtype-taking entry, nova_row_norm_chain takes a dtype, nova_debug_set_attn_variant; 0.2.2: nova_attn_fwd_lse, nova_attn_bwd; 0.2.1: nova_adaln_fc1, nova_row_norm_chain (0.2.0: 3-pass guidance fields in nova_sampler_step, KV-cached block stack, nova_modulate_rows) */

typedef enum { NOVA_F32 = 0, NOVA_BF16 = 1, NOVA_F16 = 2 } nova_dtype;
typedef enum { NOVA_ACT_NONE = 0, NOVA_ACT_GELU_ERF = 1, NOVA_ACT_SILU = 2 } nova_act;
typedef enum {
  NOVA_OK = 0,
  NOVA_ERR_ARG = -1,    /* bad enum / null pointer */
  NOVA_ERR_SHAPE = -2,  /* shape not supported by the built kernels */
  NOVA_ERR_LAUNCH = -3, /* HIP launch error */
  NOVA_ERR_DEVICE = -4  /* no gfx950 device */
} nova_status;

int nova_version(void);
const char* nova_last_error(void);
/* 0 when a gfx950 device is current, NOVA_ERR_DEVICE otherwise (checked once by the loader). */
int nova_check_device(void);

/* ---- measurement hooks (bench.py): HIP-event timing of the GEMM / attention / row-norm launches on
 * the stream they are launched on. nova_prof_enable(1) starts recording; nova_prof_collect waits
 * for the recorded events and returns, per slot, summed milliseconds, summed algorithmic work
 * (FLOPs for slots 0-4 and 6, bytes for slot 5) and launch counts, then clears the record.
 * Slots 0-3: the large-M (256x256 tile) GEMM kernel by epilogue - 0 bias with K <= N (the out-projection), 1 bias+GELU, 2 bias+SiLU,
 * 3 qkv+RoPE; 4 attention, 5 row_norm, 6 the small-M (128x128 tile / whole-K) GEMM kernels with any epilogue (decoder, embeddings),
 * 7 the large-M GEMM, bias only, K > N (the MLP's second projection), 8 token plumbing (canvas embedding, sequence build / scatter,
 * RoPE table, KV append, frame mixer, fp8 row quantisation; work = 0), 9 the diffusion MLP's glue kernels (work = 0). */
#define NOVA_PROF_SLOTS 10
int nova_prof_enable(int on);
int nova_prof_collect(double* ms, double* work, long long* launches, int slots);

/* Test hook: 0 = automatic choice between the GEMM structures by shape, 16 = the small-M whole-K kernel (bf16, K 768 /
 * 1024; an error for other shapes; 161 / 162 / 164 = the same with 16 / 32 / 64 rows per workgroup), 128 / 256 = force the 128x128 or the 256x256 (large-M, persistent) structure,
 * 257 = the 256 structure in its one-tile-per-workgroup form (the fallback of the persistent kernel); all compute
 * bit-identical results (tests/test_gpu_kernels.py compares them). Per calling thread. */
int nova_debug_force_gemm_tile(int tile);

/* Test / A-B hook: which structure the bf16, head_dim 64 attention launches for the calling thread: 0 = 32x32x16 MFMA,
 * 32 query rows per wave; 1 = 16x16x32 MFMA, 32 rows per wave; 2 = 16x16x32 MFMA, 64 rows per wave; 3 / 4 = 1 / 2 with the softmax row sums taken on the matrix pipe (an
 * all-ones V^T block) instead of vector adds; -1 = the shipped default. Same algorithm in all three (tests/test_gpu_kernels.py compares each with the f32 reference and with each other). */
int nova_debug_set_attn_variant(int variant);

/* nova_decoder_denoise replays its launch sequence as a hipGraph, captured once per distinct argument set (on by
 * default; environment NOVA_GRAPHS=0 or on = 0 here switches to direct launches for every thread and drops the cached graphs -
 * the calling thread's at once, every other thread's at its next nova_decoder_denoise). Results are identical either way. Stats: graphs captured / replayed by the calling thread. */
int nova_debug_set_graphs(int on);
int nova_debug_graph_stats(long* captures, long* replays);
/* Drops the calling thread's cached graphs (a caller that re-allocates the workspaces it passes to nova_decoder_denoise: graphs keyed
 * by the old addresses can never be replayed and would only pile up until the 2048-entry cap). */
int nova_debug_drop_graphs(void);

/* ---- projection GEMM -----------------------------------------------------------------------
 * out[M,N] = act(A[M,K] * W[N,K]^T + bias[N])        W in nn.Linear layout.
 * Replaces nn.Linear (+ nn.GELU() / nn.SiLU()) at vision_transformer.py:33-38 (MLP.fc1/fc2),
 * :64 (Attention.proj), diffusion_mlp.py:31-36 (Projector), normalization.py:28,35
 * (AdaLayerNormZero.proj), embeddings.py:174,203-206 (TextEmbed.proj).
 * Needs N % 128 == 0 and K % (128 / sizeof(dtype)) == 0; any M >= 0. */
int nova_gemm_bias_act(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int act,
                       int dtype, void* stream);

/* ---- MX-fp8 projection GEMM (BASELINE configs[4]: d48w1536 "fp8 MFMA path"; the reference has no fp8 path, so this is
 * compared with the bf16 result of the same layer) ---------------------------------------------
 * nova_quantize_rows_fp8: per-row dynamic quantisation of bf16 rows to OCP e4m3, scale[r] = max|x[r]| / 448.
 * nova_gemm_fp8_bias_act: out_bf16[M,N] = act((A8[M,K] . W8[N,K]^T) * a_scale[m] * w_scale[n] + bias[n]) on
 * v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales); N % 256 == 0, K % 128 == 0. Replaces the same nn.Linear
 * calls as nova_gemm_bias_act (vision_transformer.py:33-38,64) with quantised operands. */
int nova_quantize_rows_fp8(const void* x_bf16, void* out_fp8, float* scale, long long rows, int D, void* stream);
int nova_gemm_fp8_bias_act(const void* A8, const float* a_scale, const void* W8, const float* w_scale, const float* bias,
                           void* out_bf16, int M, int N, int K, int act, void* stream);

/* ---- fused QKV projection + 3-D RoPE -------------------------------------------------------
 * qkv[S*L, 3D] = x[S*L, D] * Wqkv^T + b, then q and k (columns [0, 2D)) rotated pairwise with
 * rope[(s % rope_batch), l, pair] = (cos, sin). rope == NULL: no rotation (abs-PE models).
 * Replaces vision_transformer.py:52-54 (qkv Linear, view/permute/unbind, pe_func on q and k) and
 * embeddings.py:36-43 (RotaryEmbed3D.ApplyFunc). The head split is never materialised: the
 * attention kernel reads q/k/v in place from this buffer. */
int nova_qkv_rope(const void* x, const void* Wqkv, const float* bias, const float* rope, void* qkv, int S, int L, int D,
                  int heads, int rope_batch, int dtype, void* stream);

/* Generalisation used for the last encoder block, whose output is only consumed at the rows predicted in the
 * current AR step (transformer_3d.py:129-132 -> diffusion_mlp.py:93): out[M,N] = x[M,K] W[N,K]^T + bias with columns
 * [0, rope_cols) rotated by rope[(m / L) % rope_batch, m % L] (heads of width head_dim). The engine calls it once with
 * the K|V rows of attn.qkv.weight over all tokens (rope_cols = D) and once with the Q rows over the n gathered rows. */
int nova_qkv_rope_cols(const void* x, const void* W, const float* bias, const float* rope, void* out, int M, int N, int K,
                       int L, int rope_batch, int head_dim, int rope_cols, int dtype, void* stream);

/* rope[nb, pad + n_tok, hd/2, 2] from integer grid positions (embeddings.py:59-67 get_func):
 * pos [n_pos,3] (t,h,w); ids [nb,n_tok] int64 gathers token -> position (NULL: identity);
 * the first `pad` rows (condition prefix) sit at position 0; inv_freq [hd/2] = 1 / theta^scale
 * laid out axis t, h, w (embeddings.py:48-50,63-64). */
int nova_rope_table(const float* pos, const long long* ids, float* rope, int nb, int pad, int n_tok, int n_pos, int hd,
                    const float* inv_freq, void* stream);

/* ---- attention ------------------------------------------------------------------------------
 * o[s, i, h, :] = softmax_j(q[s,i,h,:] . k[s,j,h,:] * scale) v[s,j,h,:], non-causal, no mask.
 * Element (s, l, head, c) of q lives at q + (s*Lq + l)*q_row_stride + head*head_dim + c (same
 * for k/v with Lk / kv_row_stride, o with o_row_stride); strides in elements, 16-byte multiples.
 * Replaces F.scaled_dot_product_attention at vision_transformer.py:63 and the
 * transpose(1,2).flatten(2) merge at :64. head_dim 64 (d48w768, d48w1024) and 96 (d48w1536) are built.
 * Non-finite inputs: the 16-bit kernels are compiled without NaN-honouring max / compare instructions. A query row whose scores
 * contain a NaN or overflow to +inf yields a non-finite output row (as SDPA's softmax would); all other output rows are
 * bit for bit those of the same call without the offending values (tests/test_gpu_kernels.py::
 * test_attention_non_finite_rows_stay_in_their_row). -inf scores do not occur (no mask input). */
int nova_attn_fwd(const void* q, const void* k, const void* v, void* o, int S, int heads, int Lq, int Lk, int head_dim,
                  long q_row_stride, long kv_row_stride, long o_row_stride, float scale, int dtype, void* stream);

/* Training path of the same attention (bf16, head_dim 64 or 96; Lq = Lk = L). q must already be multiplied by
 * scale * log2(e) (what the fused QKV epilogue does for the generation path); the forward also writes
 * lse[s, head, l] = log2 sum_j 2^(q~_l . k_j), the backward rebuilds P from it (flash-style, nothing of size L x L is
 * stored) and returns the gradients w.r.t. the UNSCALED q, k and v; it needs the forward's o for delta[s, head, l] =
 * sum_c dO * O, which it writes into delta_scratch [S, heads, L] f32 first. All matrices token-major with row strides
 * as above. key_limit (NULL = no mask): int32 [L], non-decreasing, >= 1 - query l attends to keys [0, key_limit[l]); this is the
 * reference's block-causal frame mask of multi-frame training (`MaskEmbed.get_attn_mask`, embeddings.py:247-260, set on the
 * video encoder's blocks at transformer_3d.py:176-177: a token sees the tokens of frames <= its own, the prefix counting as
 * frame 0), passed as the end of each token's visible range instead of an L x L matrix. Replaces the autograd of
 * F.scaled_dot_product_attention at vision_transformer.py:63 inside the training forward (transformer_3d.py:79-100). */
int nova_attn_fwd_lse(const void* q_scaled, const void* k, const void* v, void* o, float* lse, int S, int heads, int L,
                      int head_dim, long qkv_row_stride, long o_row_stride, const int* key_limit, void* stream);
int nova_attn_bwd(const void* q_scaled, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                  float* delta_scratch, void* dq, void* dk, void* dv, int S, int heads, int L, int head_dim,
                  long qkv_row_stride, long o_row_stride, long do_row_stride, long dqkv_row_stride, float scale,
                  const int* key_limit, void* stream);

/* ---- training side, pointwise -------------------------------------------
 * The activation between the two projections of an MLP, forward and backward, n contiguous elements of `dtype`
 * (n a multiple of 16 bytes' worth: 4 floats / 8 16-bit values), arithmetic in f32:
 *   kind NOVA_ACT_GELU_ERF: y = 0.5 x (1 + erf(x / sqrt 2)), the exact form of nn.GELU() the reference trains with
 *                           (vision_transformer.py:33-38); dx = dy (Phi(x) + x phi(x))
 *   kind NOVA_ACT_SILU:     y = x / (1 + e^-x) (the decoder MLP, diffusion_mlp.py:33,36; SiLU(z) of AdaLayerNormZero,
 *                           normalization.py:32,35); dx = dy s (1 + x (1 - s)), s = 1 / (1 + e^-x)
 * The forward keeps nothing but x; the backward recomputes the slope from it. */
int nova_act_fwd(const void* x, void* y, long long n, int kind, int dtype, void* stream);
int nova_act_bwd(const void* x, const void* dy, void* dx, long long n, int kind, int dtype, void* stream);

/* ---- training: backward of the LayerNorm family --------------------------------------------------------------------
 * For y = LN(x; eps) [* gamma + beta] [* (1 + scale) + shift] [* gate] [+ res] (nova_row_norm's forward) and the output
 * gradient dy [rows, D]: dx [rows, D]; d_scale / d_shift / d_gate written into dmod [rows, mod_ld] at the forward's offsets
 * (dmod may be NULL when mod is); the parameter gradients as `parts` partial rows d_gamma_part / d_beta_part [parts, D] f32
 * that the caller sums over dim 0 (one row per wave of the launch: parts % 4 == 0, parts / 4 workgroups; no atomics, so
 * the result is bitwise reproducible); d_res = dy is the caller's. Statistics are recomputed from x. Replaces the
 * autograd of nn.LayerNorm + residual in Block.forward (vision_transformer.py:78-82,91-92), of AdaLayerNormZero.forward
 * (normalization.py:34-36) and of DiffusionBlock's `norm2(h) * gate + x` (diffusion_mlp.py:52-53) in the training
 * forward/backward (transformer_3d.py:79-100,166-190). */
int nova_row_norm_bwd(const void* x, const void* dy, const float* gamma, const float* beta, const void* mod, long mod_ld,
                      int scale_off, int shift_off, int gate_off, void* dx, void* dmod, float* dgamma_part, float* dbeta_part,
                      int parts, long rows, int D, float eps, int dtype, void* stream);

/* ---- LayerNorm family -----------------------------------------------------------------------
 * y = LN(in[gather ? gather[r] : r]; eps) [* gamma + beta] [* (1 + mod[r, scale_off..]) +
 * mod[r, shift_off..]] [* mod[r, gate_off..]] [+ res[r]] -> out[r]. Offsets < 0 disable a term.
 * Replaces: Block post-norm residual `norm(attn(x)).add_(x)` vision_transformer.py:78-82,91-92;
 * AdaLayerNormZero.forward normalization.py:34-36; DiffusionBlock `norm2(proj(h)).mul(gate)
 * .add_(x)` diffusion_mlp.py:52-53; final encoder norm vision_transformer.py:146 fused with the
 * pred_ids row gather of diffusion_mlp.py:93; TextEmbed.norm embeddings.py:203-206. */
int nova_row_norm(const void* in, void* out, const float* gamma, const float* beta, const void* mod, long mod_ld,
                  int scale_off, int shift_off, int gate_off, const void* res, const int* gather, long rows, int D,
                  float eps, int dtype, void* stream);

/* Two chained row norms of the diffusion MLP in one pass over bf16 rows: x_new = (LN(g) gamma + beta) * mod[:, gate_off:+D]
 * + x (eps_first; DiffusionBlock's `norm2(proj(h)).mul(gate).add_(x)`, diffusion_mlp.py:52-53), then h = LN(x_new)(1 +
 * mod[:, scale_off:+D]) + mod[:, shift_off:+D] (eps_second, no affine; the next block's or the final layer's modulate,
 * diffusion_mlp.py:41-43,96-97 / normalization.py:34-36). The second norm reads x_new as stored (rounded to bf16), so the
 * result equals nova_row_norm twice, bit for bit. x_new_out may be NULL (x not needed afterwards: the last block). */
int nova_row_norm_chain(const void* g, const void* x, const float* gamma, const float* beta, const void* mod, long mod_ld,
                        int gate_off, int scale_off, int shift_off, float eps_first, float eps_second, void* x_new_out,
                        void* h_out, long rows, int D, int dtype, void* stream);

/* out = act(h W^T + bias) with h = LN(x)(1 + mod[:, scale_off:+D]) + mod[:, shift_off:+D], LN without affine: the first
 * half of DiffusionBlock.forward, `self.proj(self.norm1(x, z)...)` up to the activation (diffusion_mlp.py:41-47 with
 * AdaLayerNormZero normalization.py:34-36 and the Projector's fc1 + SiLU diffusion_mlp.py:31-36). For bf16 rows of width
 * 768 / 1024 and a few hundred rows (the per-step launches of the denoising loop at small batch) this is ONE launch with
 * the modulate as the GEMM's prologue; otherwise it runs as nova_row_norm into `h` followed by nova_gemm_bias_act, and
 * both forms give bit-identical `out`. `h` [rows, D] is scratch (written only by the two-launch form). */
int nova_adaln_fc1(const void* x, const void* mod, long mod_ld, int scale_off, int shift_off, float eps, const void* w,
                   const float* bias, void* h, void* out, long rows, int N, int D, int act, int dtype, void* stream);

/* ---- token plumbing of the masked-autoregressive encoder ------------------------------------
 * z0 = MaskEmbed(PatchEmbed(canvas)) (+ abs-PE): embeddings.py:160-166,272-274,90-91.
 * canvas [B,N,P] f32 patchified points, mask [B,N] f32 (1 = unknown), w [D,P] patchified order. */
int nova_embed_canvas(const float* canvas, const float* mask, const void* w, const float* bias, const void* mask_token,
                      const void* pos_embed, void* z0, int B, int N, int P, int D, int dtype, void* stream);
/* x[s] = cat(prefix[s] (Lp rows), tokens[s % B][ids]) : vision_transformer.py:133-136 (gather by
 * prev_ids + torch.cat). ids NULL = all n_sel tokens in order. tok_batch_rows 0 = shared tokens. */
int nova_build_sequence(const void* prefix, long prefix_seq_rows, const void* tokens, long tok_batch_rows,
                        const long long* ids, void* x, int S, int B, int Lp, int n_sel, int D, int dtype, void* stream);
/* x2[s][Lp + ids[s % B][j]] = x1[s][Lp + j] : x_masked.scatter(1, prev_ids, x) vision_transformer.py:141-143 */
int nova_scatter_tokens(const void* x1, const long long* ids, void* x2, int S, int B, int Lp, int N, int n_prev, int D,
                        int dtype, void* stream);

/* ---- diffusion-MLP glue ----------------------------------------------------------------------
 * silu(a + rowvec): the SiLU in front of every AdaLN projection with z = cond + time
 * (normalization.py:35, diffusion_mlp.py:73-75). rowvec may be NULL. */
int nova_silu_add_rows(const void* a, const void* rowvec, void* out, long rows, int D, int dtype, void* stream);
/* sinusoidal timestep features [n, freq_dim] = [cos(t f) ; sin(t f)] : diffusion_mlp.py:65-71 */
int nova_timestep_freq(const float* t, const float* freq, void* out, int n, int freq_dim, int dtype, void* stream);
/* u[s, j] = W x[s % B, j] + b for the n tokens being denoised: diffusion_mlp.py:89-92 */
int nova_patch_embed_rows(const float* x, const void* w, const float* bias, void* out, int S, int B, int n, int P, int D,
                          int dtype, void* stream);
/* head Linear + CFG combine + flow-matching Euler step on x [B,n,P] f32 in place:
 * diffusion_mlp.py:98, guidance_scaler.py:86-87, scheduling_cfm.py:134-136. h = [2B*n, D]
 * (cond rows then uncond rows) when cfg != 0, [B*n, D] otherwise. */
int nova_head_cfg_euler(const void* h, const void* w, const float* bias, float* x, int B, int n, int P, int D,
                        float guidance, int cfg, float dt, int dtype, void* stream);

/* ---- point-set metrics (the step after generation) ------------------------------------------
 * x [B, N, 3], y [B, M, 3] float32 point sets (NOVAPipeline's latent output flattened to points). Coordinates are clamped to
 * [clamp_lo, clamp_hi] first, as the reference does before torch.cdist (test_optimize.py:357-358,388-389: +-5;
 * train_newloss.py:321-322,357-358: +-1 / +-2).
 *   nova_pointset_nn_dist        d[b, i] = min_j ||x[b, i] - y[b, j]||  (Euclidean, not squared): the `dist.min(dim=2)`
 *                                of compute_chamfer_distance (test_optimize.py:367-369); call it twice, operands swapped, for
 *                                the two directions. unit_norm != 0 scales every point to unit length after the clamp
 *                                (distChamfer, train_newloss.py:325-337).
 *   nova_pointset_pairwise_dist  D[b, i, j] = ||x[b, i] - y[b, j]||: the cost matrix `torch.cdist(pred[i], target[i])`
 *                                handed to scipy's linear_sum_assignment by compute_emd_distance (test_optimize.py:399-404)
 *                                and emd_approx (train_newloss.py:360-370). */
int nova_pointset_nn_dist(const float* x, const float* y, float* d, int B, int N, int M, float clamp_lo, float clamp_hi,
                          int unit_norm, void* stream);
int nova_pointset_pairwise_dist(const float* x, const float* y, float* D, int B, int N, int M, float clamp_lo, float clamp_hi,
                                void* stream);

/* ---- composite entry points (what the AR loop actually calls) --------------------------------
 * One ViT block's parameters (reference state_dict names in comments). GEMM weights in `dtype`,
 * everything else f32. */
typedef struct {
  const void* qkv_w;    /* attn.qkv.weight [3D, D] */
  const float* qkv_b;   /* attn.qkv.bias   [3D]    */
  const void* proj_w;   /* attn.proj.weight [D, D] */
  const float* proj_b;
  const float* norm1_w; /* norm1.weight [D] */
  const float* norm1_b;
  const void* fc1_w;    /* mlp.fc1.weight [4D, D] */
  const float* fc1_b;
  const void* fc2_w;    /* mlp.fc2.weight [D, 4D] */
  const float* fc2_b;
  const float* norm2_w;
  const float* norm2_b;
} nova_vit_block;

/* x[S*L, D] <- blocks[nblocks-1](...blocks[0](x)) in place, post-norm residual blocks with fused
 * QKV+RoPE, flash attention, GELU MLP. Replaces the `for blk in self.blocks[...]` loops of
 * VisionTransformer.forward (vision_transformer.py:137-138,144-145) and Block.forward (:89-92).
 * Workspaces: ws_qkv [S*L, 3D], ws_a [S*L, D], ws_b [S*L, D], ws_h [S*L, hidden] in `dtype`. */
int nova_vit_blocks_forward(const nova_vit_block* blocks, int nblocks, void* x, int S, int L, int D, int heads,
                            int hidden, const float* rope, int rope_batch, void* ws_qkv, void* ws_a, void* ws_b,
                            void* ws_h, int dtype, void* stream);

/* MX-fp8 weights of one block's three large GEMMs (BASELINE configs[4]): OCP e4m3 bytes in nn.Linear's [N][K] layout
 * + one float32 scale per output row (w = w8 * ws[n]), made once with nova_quantize_rows_fp8 on the bf16 weights. */
typedef struct {
  const void* qkv_w8;  const float* qkv_ws;   /* [3D, D], [3D] */
  const void* fc1_w8;  const float* fc1_ws;   /* [4D, D], [4D] */
  const void* fc2_w8;  const float* fc2_ws;   /* [D, 4D], [D]  */
} nova_vit_block_fp8;

/* Building blocks of the fp8 stack, exposed for tests: the post-norm residual LayerNorm (vision_transformer.py:78-82,91-92)
 * that also emits its bf16 output row as e4m3 + per-row scale, bit-compatible with nova_quantize_rows_fp8(out); and the
 * fused QKV projection on fp8 operands with the RoPE / q-scale epilogue of nova_qkv_rope (bf16 result). */
int nova_row_norm_fp8(const void* in, void* out, const float* gamma, const float* beta, const void* res, void* out8, float* out8_scale,
                      long rows, int D, float eps, void* stream);
/* out8[M,N] (e4m3 bytes) = saturate(GELU((A8 . W8^T) * a_scale[m] * w_scale[n] + bias) / *out_scale), and *out_amax raised
 * (atomic max on the float's bits) to the largest |GELU(.)|: the fc1 of the fp8 stack under delayed scaling. */
int nova_gemm_fp8_gelu_q8(const void* A8, const float* a_scale, const void* W8, const float* w_scale, const float* bias, void* out8,
                          int M, int N, int K, const float* out_scale, unsigned* out_amax, void* stream);
int nova_qkv_rope_fp8(const void* x8, const float* x_scale, const void* w8, const float* w_scale, const float* bias, const float* rope,
                      void* qkv, int S, int L, int D, int heads, int rope_batch, float q_scale, void* stream);

/* nova_vit_blocks_forward with the fused-QKV, fc1 and fc2 GEMMs of every block on the block-scaled fp8 MFMA
 * (v_mfma_scale_f32_16x16x128_f8f6f4, 2x the bf16 rate): activations are quantised per row on the fly (the residual
 * stream by the LayerNorm kernel that produces it, the MLP hidden rows by one pass), f32 accumulation, bf16 results;
 * attention, its out-projection, the LayerNorms and the residual stream are as in the bf16 stack. `blocks` supplies the
 * biases, norms and the bf16 out-projection, `q` the fp8 weights. bf16 activations only. The reference has no fp8 path:
 * results are compared with the bf16 stack (tests), not with the reference. Extra workspaces: ws_x8 [S*L, D] bytes,
 * ws_xs [S*L] f32, ws_h8 [S*L, hidden] bytes, ws_hs [S*L] f32. Needs D, hidden % 256 == 0 and L >= 16.
 * h_scale / h_amax: both NULL (the MLP hidden rows are quantised per row by a pass between fc1 and fc2), or device arrays of
 * nblocks floats / unsigned ("delayed scaling"): fc1's epilogue writes GELU(.) / h_scale[i] as e4m3 itself, saturated at +-448,
 * and raises h_amax[i] (float bits, atomic max) to the largest |GELU(.)| it saw; fc2 uses h_scale[i] for every row. The
 * caller zeroes h_amax and sets h_scale[i] = margin * amax_i / 448 from the PREVIOUS call of the same stack (first call: a
 * guess) - NovaEngine does, per lane. */
int nova_vit_blocks_forward_fp8(const nova_vit_block* blocks, const nova_vit_block_fp8* q, int nblocks, void* x, int S, int L, int D,
                                int heads, int hidden, const float* rope, int rope_batch, void* ws_qkv, void* ws_a, void* ws_b,
                                void* ws_h, void* ws_x8, float* ws_xs, void* ws_h8, float* ws_hs, float* h_scale, unsigned* h_amax,
                                void* stream);

/* The same block stack for the conditioning encoder of multi-frame generation (max_latent_length > 1): the k | v rows
 * each block's fused QKV projection produces for the L rows of x are appended to that block's cache and attention runs
 * over cache_len + L keys (vision_transformer.py:55-60: `torch.cat([cache_kv, k], dim=2)`; enable_kvcache :125-126).
 *   kv_cache [nblocks][S][cache_cap][2D] in `dtype` (k then v per row, k already rotated), caller-owned; rows
 *   [0, cache_len) of every sequence are valid on entry, [cache_len, cache_len + L) on return. The caller advances
 *   cache_len by L after the call. rope covers the L new rows only ([rope_batch, L, hd/2, 2]). */
int nova_vit_blocks_forward_kv(const nova_vit_block* blocks, int nblocks, void* x, int S, int L, int D, int heads,
                               int hidden, const float* rope, int rope_batch, void* kv_cache, long cache_cap,
                               long cache_len, void* ws_qkv, void* ws_a, void* ws_b, void* ws_h, int dtype, void* stream);

/* out[r, :] = x[r, :] * (1 + mod[r, 0:D]) + mod[r, D:2D]: AdaLayerNorm with eps=None (no normalisation), the frame mixer
 * `video_encoder.mixer` of transformer_nova.py:87-89 / normalization.py:39-46 applied at transformer_3d.py:156-158. */
int nova_modulate_rows(const void* x, const void* mod, void* out, long rows, int D, int dtype, void* stream);

typedef struct {
  const void* fc1_w;  /* blocks.i.proj.fc1.weight [D, D] */
  const float* fc1_b;
  const void* fc2_w;  /* blocks.i.proj.fc2.weight [D, D] */
  const float* fc2_b;
  const float* norm2_w; /* blocks.i.norm2 */
  const float* norm2_b;
} nova_mlp_block;

typedef struct {
  int depth;              /* number of DiffusionBlocks (6) */
  const nova_mlp_block* blocks;
  const void* adaln_w;    /* cat(blocks.i.norm1.proj.weight (3D each), norm.proj.weight (2D)) [(3*depth+2)D, D] */
  const float* adaln_b;   /* same concatenation of the biases */
  const void* patch_w;    /* patch_embed.proj.weight as [D, P] in patchified (i, j, c) order */
  const float* patch_b;
  const void* head_w;     /* head.weight [P, D] */
  const float* head_b;
} nova_decoder;

/* One sampler step, prepared on the host (float32): with v = CFG-combined head output,
 *   x0 = clamp(kx*x + kv*v, +-clip) (clip <= 0: none);   x <- c0*x0 + cx*x + sigma*noise
 * Flow-matching Euler (scheduling_cfm.py:134-136): kx=0, kv=1, clip=0, c0=sigma_{i+1}-sigma_i, cx=1, sigma=0.
 * DDPM (scheduling_ddpm.py:236-316): kx, kv from prediction_type (:271-280), clip = clip_sample_range (:283-289),
 * c0 / cx the posterior-mean coefficients (:293-298), sigma the posterior std (:303-312).
 * guidance <= 1 disables CFG for that step (guidance_trunc, guidance_scaler.py:59-65).
 * 3-pass guidance (guidance_scaler.py:46-57,78-85; rows [cond ; uncond ; third], S = 3B): extra_kind 1 = image guidance
 * (third = uncond text with the image condition): v = u + g (c - t) [renorm] + extra_scale (t - u); extra_kind 2 =
 * spatiotemporal guidance (third = cond text): v = u + g (c - u) [renorm] + extra_scale (c - t); 0 = 2-pass. */
typedef struct {
  float guidance, kx, kv, clip, c0, cx, sigma;
  float extra_scale;
  int extra_kind;
} nova_sampler_step;

/* All `steps` sampler steps of Transformer3DModel.denoise (transformer_3d.py:102-113) for the n tokens predicted in
 * this AR step, launched back to back on `stream`:
 *   zc    [S*n, D]   condition rows: condition_proj(LN(z)[pred_ids]) (time term NOT yet added)
 *   temb  [steps, D] timestep_proj(freq(t_i)) per step (diffusion_mlp.py:73)
 *   x     [B, n, P]  f32, in: noise rows, out: denoised patch vectors
 *   sched [steps]    host array of nova_sampler_step
 *   noise [steps, B, n, P] f32 or NULL: the per-step gaussian rows of an ancestral sampler (used where sigma != 0)
 *   renorm           guidance_renorm (>= 1: off). When < 1 (flow-matching Euler only): guidance_scaler.py:67-72 with
 *                    echo_energy [B] f32 in/out = squared norm of each sample's rows that are NOT predicted in this AR
 *                    step (they echo x_t in the reference and enter both norms), ws_v [2*B*n*P] f32 scratch
 *                    ([3*B*n*P] with 3-pass guidance).
 * S = 2B when any sched[i].guidance > 1 (cond rows then uncond rows; 3B with a third guidance pass), else S = B.
 * Workspaces in `dtype`: ws_u, ws_h, ws_f, ws_g [S*n, D]; ws_a [mod_steps*S*n, D]; ws_mod [mod_steps*S*n, (3*depth+2)*D].
 * mod_steps = 1: the AdaLN projection of SiLU(zc + temb[i]) (normalization.py:35) is computed inside every step;
 * mod_steps = steps: for all steps in ONE GEMM ahead of the loop (the condition rows do not change during the loop,
 * diffusion_mlp.py:93-95), which needs the steps-times larger ws_a / ws_mod. Same results either way. */
int nova_decoder_denoise(const nova_decoder* dec, const void* zc, const void* temb, float* x, const nova_sampler_step* sched,
                         const float* noise, float renorm, float* echo_energy, int steps, int S, int B, int n, int P, int D,
                         void* ws_a, void* ws_u, void* ws_h, void* ws_f, void* ws_g, void* ws_mod, float* ws_v, int mod_steps,
                         int dtype, void* stream);

/* The same loop with guidance_renorm < 1 for ANY sampler step (the DDPM ancestral step of scheduling_ddpm.py:236-316: noise and a
 * clamp act on the rows that merely echo x_t, so their squared norm is not a scalar that evolves by itself): the echo rows are
 * carried explicitly and take every step beside the predicted ones (guidance_scaler.py:67-72 norms over all N rows; for an echo
 * row every guidance pass returns x_t).
 *   echo_rows  [B, Ne, P] f32 in/out: the Ne = N - n rows of each sample that are not predicted in this AR step (in: their x_T)
 *   echo_noise [steps, B, Ne, P] f32 or NULL: their per-step gaussian rows (used where sigma != 0)
 * Everything else as nova_decoder_denoise (ws_v as for renorm there). Launched directly (no graph replay: per-call noise). */
int nova_decoder_denoise_echo(const nova_decoder* dec, const void* zc, const void* temb, float* x, const nova_sampler_step* sched,
                              const float* noise, float renorm, float* echo_rows, const float* echo_noise, int Ne, int steps, int S, int B,
                              int n, int P, int D, void* ws_a, void* ws_u, void* ws_h, void* ws_f, void* ws_g, void* ws_mod, float* ws_v,
                              int mod_steps, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NOVA_HIP_H_ */
