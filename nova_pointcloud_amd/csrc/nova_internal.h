// Internal declarations shared by the kernel translation units and the C ABI (capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/nova_hip.h"

#include "common.h"

namespace nova {

// storage dtype code <-> element type. Every dtype-generic launcher goes through dispatch_dtype, so a new storage type is
// one more branch here plus the traits in common.h.
template <typename T> constexpr int dtype_of() { return sizeof(T) == 4 ? NOVA_F32 : NOVA_BF16; }
template <> constexpr int dtype_of<f16_t>() { return NOVA_F16; }
inline bool dtype_is16(int dtype) { return dtype == NOVA_BF16 || dtype == NOVA_F16; }
template <typename F> inline auto dispatch_dtype(int dtype, F&& f) {
  return dtype == NOVA_BF16 ? f(bf16_t{}) : dtype == NOVA_F16 ? f(f16_t{}) : f(float{});
}
template <typename F> inline auto dispatch_half(int dtype, F&& f) { return dtype == NOVA_F16 ? f(f16_t{}) : f(bf16_t{}); }

// thread-local last-error record; returns `code` so callers can `return set_error(...)`.
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// hipGetLastError() -> NOVA_ERR_LAUNCH (+ message) or 0. Never synchronises.
int check_launch(const char* what);

// ---- optional per-kernel timing with HIP events on the launch stream (capi.hip). Slots: one per
// kernel family; `work` = algorithmic FLOPs (or bytes) of the launch. No-ops unless enabled.
// PROF_GEMM_* : the 256x256 (large-M, encoder) GEMM kernels by epilogue (PROF_GEMM_NONE: bias-only launches with K <= N, the
// out-projection; PROF_GEMM_WIDEK: bias-only launches with K > N, the MLP's second projection); PROF_GEMM_SMALL: the 128x128 and
// small-M kernels, any epilogue; PROF_TOKENS: the token plumbing of the split encoder (canvas embedding, sequence build / scatter,
// RoPE table, KV append, frame mixer, fp8 row quantisation); PROF_DECODER: the diffusion MLP's glue kernels (timestep features,
// SiLU-add, patch embedding of the predicted rows, head + guidance + sampler step)
enum { PROF_GEMM_NONE = 0, PROF_GEMM_GELU, PROF_GEMM_SILU, PROF_GEMM_ROPE, PROF_ATTN, PROF_ROWNORM, PROF_GEMM_SMALL, PROF_GEMM_WIDEK,
       PROF_TOKENS, PROF_DECODER, PROF_SLOTS };
inline int prof_gemm_slot(int epi, int N, int K) { return epi == 0 && K > N ? PROF_GEMM_WIDEK : PROF_GEMM_NONE + epi; }
struct ProfScope {
  ProfScope(int slot, double work, hipStream_t st);
  ~ProfScope();
  int idx;
  hipStream_t st;
};

// ---- gemm.hip
int gemm_bias_act(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int act,
                  int dtype, hipStream_t st);
int gemm_qkv_rope(const void* x, const void* Wqkv, const float* bias, const float* rope, void* qkv, int S, int L,
                  int D, int heads, int rope_batch, int dtype, hipStream_t st, float q_scale = 1.0f);

int gemm_rope_cols(const void* x, const void* W, const float* bias, const float* rope, void* out, int M, int N, int K,
                   int L, int rope_batch, int hd, int rope_cols, int dtype, hipStream_t st);
int gemm_force_tile(int tile);  // 0 on success, -1 for a value this build does not know
int gemm_forced_tile();  // the calling thread's current setting (part of the decoder graph key)
// traversal direction of the NEXT launches of the row-tiled kernels (persistent GEMM, attention, row_norm); see
// xcd_remap_dir in common.h. Set by the block composites, false for direct calls of the single-kernel entry points.
void walk_reverse(bool on);
bool walk_is_reverse();

// ---- attn.hip
// kv_seq_stride: elements between the first key rows of consecutive sequences (0 = Lk * kv_row_stride, i.e. packed);
// a KV cache of capacity cap rows per sequence passes cap * kv_row_stride.
int attn_fwd(const void* q, const void* k, const void* v, void* o, int S, int heads, int Lq, int Lk, int hd,
             long q_row_stride, long kv_row_stride, long o_row_stride, float scale, int dtype, hipStream_t st,
             bool q_prescaled = false, long kv_seq_stride = 0, float* lse = nullptr, const int* klim = nullptr);
// attn16.hip: the same attention on the 16x16x32 MFMA shape (bf16, head_dim 64), 32 or 64 query rows per wave
int attn_fwd_m16(const void* q, const void* k, const void* v, void* o, int S, int heads, int Lq, int Lk, int hd, long q_rs, long kv_rs,
                 long o_rs, float cl, int dtype, hipStream_t st, long kv_ss, float* lse, int rows_per_wave, bool sum_on_mfma, bool pipelined = false,
                 const int* klim = nullptr);  // klim [Lq]: optional per-query key limit (training forward, block-causal frame mask)
// 16x16x32, 32 rows per wave, row sums on the matrix pipe: +8 .. 11 % over variant 0 at every L measured (profiles/r03_attn_variants_ab.txt)
constexpr int NOVA_ATTN_DEFAULT_VARIANT = 3;
int attn_set_variant(int v);  // -1 default, 0 .. 5 (attn.hip); -1 returned for other values
int attn_variant();
// attn_bwd.hip (bf16, head_dim 64 / 96): gradients of the attention above from q (pre-scaled by scale * log2 e), k, v, the
// forward's output o and log2-domain row log-sum-exp, and dO; delta [S, heads, L] is scratch (filled with sum_c dO * O first).
// All matrices token-major [S, L, heads * hd] with row strides.
int attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, float* delta, void* dq,
             void* dk, void* dv, int S, int heads, int L, int hd, long qkv_rs, long o_rs, long do_rs, long dqkv_rs, float scale,
             hipStream_t st, const int* klim = nullptr);  // klim [L]: optional per-query key limit (block-causal frame mask)

// ---- rowops.hip
struct RowNormArgs {
  const void* in;        // [rows, D]
  void* out;             // [rows, D]
  const float* gamma;    // [D] or null
  const float* beta;     // [D] or null
  const void* mod;       // modulation rows [rows, mod_ld] or null
  long mod_ld;
  int scale_off, shift_off, gate_off;   // column offsets inside a mod row; -1 = absent
  const void* res;       // residual [rows, D] or null
  const int* gather;     // optional source row index per output row (rows of `in`), or null
  long rows;
  int D;
  float eps;
  int rev = 0;           // set by row_norm() from walk_is_reverse(); callers leave it
  void* out8 = nullptr;  // bf16 only, optional: the output row once more as OCP e4m3 bytes [rows, D] with the per-row
  float* out8_scale = nullptr;  // scale amax / 448 [rows] (the fp8 A operand of the next GEMM; same rule as quantize_rows_fp8)
};
int row_norm(const RowNormArgs& a, int dtype, hipStream_t st);

// ---- skinny.hip: small-M GEMM (bf16 / f16, K in {768, 1024}), bit-identical to the tile kernels; optional AdaLN-modulate prologue
bool skinny_gemm_fits(int M, int N, int K, bool modulate);
bool gemm_modulate_fused(int M, int N, int K, int dtype);  // gemm.hip: modulate + GEMM taken as one launch
int skinny_forced_row_blocks();  // the calling thread's current setting (part of the decoder graph key)
void skinny_force_row_blocks(int rb);  // calling thread: 1 / 2 / 4 = rows per workgroup 16 / 32 / 64 for plain launches, 0 = by rule
int skinny_gemm(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int act,
                const RowNormArgs* pro, int dtype, hipStream_t st);
int row_norm_chain(const RowNormArgs& a2, const RowNormArgs& a1, void* x_new_out, int dtype, hipStream_t st);  // rowops.hip, bf16 / f16: two chained norms, one pass
// out = act(modulate(pro) W^T + bias): one launch where skinny.hip applies, else row_norm into pro.out followed by the GEMM
int gemm_modulate_act(const RowNormArgs& pro, const void* W, const float* bias, void* out, int M, int N, int K, int act,
                      int dtype, hipStream_t st);

int rope_table(const float* pos, const long long* ids, float* table, int nb, int pad, int n_tok, int n_pos,
               int hd, const float* inv_freq, hipStream_t st);
int embed_canvas(const float* canvas, const float* mask, const void* w, const float* bias, const void* mask_token,
                 const void* pos_embed, void* z0, int B, int N, int P, int D, int dtype, hipStream_t st);
int build_sequence(const void* prefix, long prefix_seq_rows, const void* tokens, long tok_batch_rows,
                   const long long* ids, void* x, int S, int B, int Lp, int n_sel, int D, int dtype, hipStream_t st);
int scatter_tokens(const void* x1, const long long* ids, void* x2, int S, int B, int Lp, int N, int n_prev, int D,
                   int dtype, hipStream_t st);
int quantize_rows_fp8(const void* x, void* out, float* scale, long rows, int D, hipStream_t st);
int gemm256_fp8_launch(const void* A8, const float* sa, const void* W8, const float* sw, const float* bias, void* C, int M,
                       int N, int K, int epi, hipStream_t st, const float* rope = nullptr, int L = 1, int rope_batch = 1,
                       int hd = 2, int rope_cols = 0, float q_scale = 1.0f, int q_cols = 0, int sa_scalar = 0,
                       const float* q8_scale = nullptr, unsigned* q8_amax = nullptr);
constexpr int NOVA_EPI_GELU_Q8 = 5;  // gemm256.hip E_GELU_Q8: GELU + e4m3 output with a static scale (fp8 operands only)
int silu_add_rows(const void* a, const void* rowvec, void* out, long rows, int D, int dtype, hipStream_t st);
int silu_add_steps(const void* a, const void* vecs, void* out, long rows, int nvec, int D, int dtype, hipStream_t st);
int timestep_freq(const float* t, const float* freq, void* out, int n, int freq_dim, int dtype, hipStream_t st);
int patch_embed_rows(const float* x, const void* w, const float* bias, void* out, int S, int B, int n, int P, int D,
                     int dtype, hipStream_t st);
struct SamplerStep {  // == nova_sampler_step (include/nova_hip.h)
  float guidance, kx, kv, clip, c0, cx, sigma;
  float extra_scale;  // 3-pass guidance: image / spatiotemporal guidance scale
  int extra_kind;     // 0 = 2-pass, 1 = image guidance, 2 = spatiotemporal guidance (third row block of h)
};
int head_cfg_step(const void* h, const void* w, const float* bias, float* x, const float* noise, float* vhat, float* cond,
                  float* extra, int B, int n, int P, int D, const SamplerStep& sp, int defer, int dtype, hipStream_t st);
int scale_vector(float* v, int n, float f, hipStream_t st);
int renorm_euler(float* x, const float* vhat, const float* cond, const float* extra, float* echo, int B, int n, int P, float dt,
                 float renorm, hipStream_t st);
int renorm_step(float* x, const float* vhat, const float* cond, const float* extra, const float* noise, float* echo, const float* echo_noise,
                int B, int nP, int eP, const SamplerStep& sp, float renorm, int echo_only, hipStream_t st);
int kv_append(const void* qkv, void* cache, int S, int Lq, int D, long cap, long base, int dtype, hipStream_t st);
int modulate_rows(const void* x, const void* mod, void* out, long rows, int D, int dtype, hipStream_t st);


// ---- pointset.hip (Chamfer / EMD distance work)
int pointset_nn_dist(const float* x, const float* y, float* d, int B, int N, int M, float lo, float hi, int unit, hipStream_t st);
int pointset_pairwise_dist(const float* x, const float* y, float* D, int B, int N, int M, float lo, float hi, hipStream_t st);

}  // namespace nova
