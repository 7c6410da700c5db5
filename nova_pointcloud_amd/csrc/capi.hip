// C ABI of libnova_hip.so (declared in include/nova_hip.h): argument checks, error record,
// and the composite launch sequences (ViT block stack, diffusion-MLP denoise loop).
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>
#include "common.h"
#include "nova_internal.h"

namespace nova {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  return set_error(NOVA_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
}

// ---- profiling: event pairs recorded around selected launches, summed on nova_prof_collect()
struct ProfRec { hipEvent_t a, b; int slot; double work; };
static std::atomic<bool> g_prof_on{false};
static std::mutex g_prof_mu;  // the record list is shared by all calling threads
static std::vector<ProfRec> g_prof;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;

ProfScope::ProfScope(int slot, double work, hipStream_t s) : idx(-1), st(s) {
  if (!g_prof_on.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r;
  if (!g_prof_pool.empty()) {
    r.a = g_prof_pool.back().first;
    r.b = g_prof_pool.back().second;
    g_prof_pool.pop_back();
  } else if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) {
    return;
  }
  r.slot = slot;
  r.work = work;
  (void)hipEventRecord(r.a, st);
  g_prof.push_back(r);
  idx = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope() {
  if (idx < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (idx < (int)g_prof.size()) (void)hipEventRecord(g_prof[idx].b, st);
}

static inline bool bad_dtype(int dtype) { return dtype != NOVA_F32 && dtype != NOVA_BF16 && dtype != NOVA_F16; }
static inline size_t esize(int dtype) { return dtype_is16(dtype) ? 2 : 4; }

}  // namespace nova

using namespace nova;

#define NOVA_REQUIRE(cond, code, ...) \
  do {                                \
    if (!(cond)) return set_error(code, __VA_ARGS__); \
  } while (0)
#define NOVA_TRY(expr)        \
  do {                        \
    const int rc_ = (expr);   \
    if (rc_ != 0) return rc_; \
  } while (0)

namespace nova {
// Host-side launch state is per calling thread (ctypes releases the GIL during a call): the walk direction is set by a
// composite for the launches it issues itself and restored when it returns, on every path.
static thread_local bool g_walk_rev = false;
void walk_reverse(bool on) { g_walk_rev = on; }
bool walk_is_reverse() { return g_walk_rev; }
#ifdef NOVA_EXPERIMENTS
static int g_walk_alternate = 1;  // nova_debug_force_gemm_tile(50000 / 50001) switches the alternation off / on
void walk_set_alternate(int on) { g_walk_alternate = on; }
#else
constexpr int g_walk_alternate = 1;
#endif
}  // namespace nova

extern "C" {

int nova_version(void) { return NOVA_HIP_VERSION; }
const char* nova_last_error(void) { return g_err; }

int nova_check_device(void) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return set_error(NOVA_ERR_DEVICE, "no HIP device");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return set_error(NOVA_ERR_DEVICE, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return set_error(NOVA_ERR_DEVICE, "device %d is %s; libnova_hip is built for gfx950 only", dev, prop.gcnArchName);
  return 0;
}

int nova_debug_force_gemm_tile(int tile) {
  NOVA_REQUIRE(gemm_force_tile(tile) == 0, NOVA_ERR_ARG,
               "force_gemm_tile: this build knows 0 (auto), 16 (small-M kernel; 161 / 162 / 164 with 16 / 32 / 64 rows per workgroup), 64 / 128 (the small-M tile kernels), 256 (persistent), 257 (one tile per workgroup) and 258 (persistent, prologue form); experiment codes need the NOVA_EXPERIMENTS build");
  return 0;
}

int nova_prof_enable(int on) {
  g_prof_on.store(on != 0);
  return 0;
}

int nova_prof_collect(double* ms, double* work, long long* launches, int slots) {
  NOVA_REQUIRE(ms && work && launches && slots >= PROF_SLOTS, NOVA_ERR_ARG, "prof_collect: need %d slots", PROF_SLOTS);
  for (int i = 0; i < slots; ++i) { ms[i] = 0; work[i] = 0; launches[i] = 0; }
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
      ms[r.slot] += t;
      work[r.slot] += r.work;
      launches[r.slot] += 1;
    }
    g_prof_pool.emplace_back(r.a, r.b);
  }
  g_prof.clear();
  return 0;
}

int nova_gemm_bias_act(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int act,
                       int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "gemm: bad dtype %d", dtype);
  NOVA_REQUIRE(M == 0 || (A && W && out), NOVA_ERR_ARG, "gemm: null pointer");
  return gemm_bias_act(A, W, bias, out, M, N, K, act, dtype, (hipStream_t)stream);
}

int nova_quantize_rows_fp8(const void* x, void* out, float* scale, long long rows, int D, void* stream) {
  NOVA_REQUIRE(rows <= 0 || (x && out && scale), NOVA_ERR_ARG, "quantize_rows_fp8: null pointer");
  return quantize_rows_fp8(x, out, scale, (long)rows, D, (hipStream_t)stream);
}

int nova_gemm_fp8_bias_act(const void* A8, const float* a_scale, const void* W8, const float* w_scale, const float* bias,
                           void* out, int M, int N, int K, int act, void* stream) {
  NOVA_REQUIRE(M <= 0 || (A8 && a_scale && W8 && w_scale && out), NOVA_ERR_ARG, "gemm_fp8: null pointer");
  NOVA_REQUIRE(act >= NOVA_ACT_NONE && act <= NOVA_ACT_SILU, NOVA_ERR_ARG, "gemm_fp8: bad activation %d", act);
  return gemm256_fp8_launch(A8, a_scale, W8, w_scale, bias, out, M, N, K, act, (hipStream_t)stream);
}

int nova_qkv_rope(const void* x, const void* Wqkv, const float* bias, const float* rope, void* qkv, int S, int L, int D,
                  int heads, int rope_batch, int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "qkv_rope: bad dtype %d", dtype);
  NOVA_REQUIRE(S * L == 0 || (x && Wqkv && qkv), NOVA_ERR_ARG, "qkv_rope: null pointer");
  return gemm_qkv_rope(x, Wqkv, bias, rope, qkv, S, L, D, heads, rope_batch, dtype, (hipStream_t)stream);
}

int nova_qkv_rope_cols(const void* x, const void* W, const float* bias, const float* rope, void* out, int M, int N, int K,
                       int L, int rope_batch, int head_dim, int rope_cols, int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "qkv_rope_cols: bad dtype %d", dtype);
  NOVA_REQUIRE(M == 0 || (x && W && out), NOVA_ERR_ARG, "qkv_rope_cols: null pointer");
  NOVA_REQUIRE(L > 0 && head_dim > 0 && head_dim % 4 == 0 && rope_cols % 128 == 0 && rope_cols <= N, NOVA_ERR_SHAPE,
               "qkv_rope_cols: bad L/head_dim/rope_cols");
  return gemm_rope_cols(x, W, bias, rope, out, M, N, K, L, rope_batch, head_dim, rope_cols, dtype, (hipStream_t)stream);
}

int nova_rope_table(const float* pos, const long long* ids, float* rope, int nb, int pad, int n_tok, int n_pos, int hd,
                    const float* inv_freq, void* stream) {
  NOVA_REQUIRE(pos && rope && inv_freq, NOVA_ERR_ARG, "rope_table: null pointer");
  NOVA_REQUIRE(hd > 0 && hd % 8 == 0, NOVA_ERR_SHAPE, "rope_table: head_dim %d", hd);
  return rope_table(pos, ids, rope, nb, pad, n_tok, n_pos, hd, inv_freq, (hipStream_t)stream);
}

int nova_attn_fwd(const void* q, const void* k, const void* v, void* o, int S, int heads, int Lq, int Lk, int head_dim,
                  long q_row_stride, long kv_row_stride, long o_row_stride, float scale, int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "attn_fwd: bad dtype %d", dtype);
  NOVA_REQUIRE(q && k && v && o, NOVA_ERR_ARG, "attn_fwd: null pointer");
  return attn_fwd(q, k, v, o, S, heads, Lq, Lk, head_dim, q_row_stride, kv_row_stride, o_row_stride, scale, dtype,
                  (hipStream_t)stream);
}

int nova_attn_fwd_lse(const void* q_scaled, const void* k, const void* v, void* o, float* lse, int S, int heads, int L,
                      int head_dim, long qkv_row_stride, long o_row_stride, const int* key_limit, void* stream) {
  NOVA_REQUIRE(q_scaled && k && v && o && lse, NOVA_ERR_ARG, "attn_fwd_lse: null pointer");
  return attn_fwd(q_scaled, k, v, o, S, heads, L, L, head_dim, qkv_row_stride, qkv_row_stride, o_row_stride, 1.0f, NOVA_BF16,
                  (hipStream_t)stream, true, 0, lse, key_limit);
}

int nova_attn_bwd(const void* q_scaled, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                  float* delta_scratch, void* dq, void* dk, void* dv, int S, int heads, int L, int head_dim,
                  long qkv_row_stride, long o_row_stride, long do_row_stride, long dqkv_row_stride, float scale,
                  const int* key_limit, void* stream) {
  NOVA_REQUIRE(q_scaled && k && v && o && d_o && lse && delta_scratch && dq && dk && dv, NOVA_ERR_ARG, "attn_bwd: null pointer");
  return attn_bwd(q_scaled, k, v, o, d_o, lse, delta_scratch, dq, dk, dv, S, heads, L, head_dim, qkv_row_stride, o_row_stride,
                  do_row_stride, dqkv_row_stride, scale, (hipStream_t)stream, key_limit);
}

int nova_row_norm(const void* in, void* out, const float* gamma, const float* beta, const void* mod, long mod_ld,
                  int scale_off, int shift_off, int gate_off, const void* res, const int* gather, long rows, int D,
                  float eps, int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "row_norm: bad dtype %d", dtype);
  NOVA_REQUIRE(rows == 0 || (in && out), NOVA_ERR_ARG, "row_norm: null pointer");
  RowNormArgs a{in, out, gamma, beta, mod, mod_ld, scale_off, shift_off, gate_off, res, gather, rows, D, eps};
  return row_norm(a, dtype, (hipStream_t)stream);
}

int nova_row_norm_chain(const void* g, const void* x, const float* gamma, const float* beta, const void* mod, long mod_ld,
                        int gate_off, int scale_off, int shift_off, float eps_first, float eps_second, void* x_new_out,
                        void* h_out, long rows, int D, int dtype, void* stream) {
  NOVA_REQUIRE(dtype_is16(dtype), NOVA_ERR_ARG, "row_norm_chain: dtype %d is not a 16-bit storage type", dtype);
  NOVA_REQUIRE(rows == 0 || (g && x && mod && h_out), NOVA_ERR_ARG, "row_norm_chain: null pointer");
  RowNormArgs a2{g, nullptr, gamma, beta, mod, mod_ld, -1, -1, gate_off, x, nullptr, rows, D, eps_first};
  RowNormArgs a1{nullptr, h_out, nullptr, nullptr, mod, mod_ld, scale_off, shift_off, -1, nullptr, nullptr, rows, D, eps_second};
  return row_norm_chain(a2, a1, x_new_out, dtype, (hipStream_t)stream);
}

int nova_adaln_fc1(const void* x, const void* mod, long mod_ld, int scale_off, int shift_off, float eps, const void* w,
                   const float* bias, void* h, void* out, long rows, int N, int D, int act, int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "adaln_fc1: bad dtype %d", dtype);
  NOVA_REQUIRE(rows == 0 || (x && mod && w && h && out), NOVA_ERR_ARG, "adaln_fc1: null pointer");
  NOVA_REQUIRE(rows >= 0 && rows <= 0x7fffffffL && scale_off >= 0 && shift_off >= 0, NOVA_ERR_ARG, "adaln_fc1: bad rows / offsets");
  RowNormArgs m{x, h, nullptr, nullptr, mod, mod_ld, scale_off, shift_off, -1, nullptr, nullptr, rows, D, eps};
  return gemm_modulate_act(m, w, bias, out, (int)rows, N, D, act, dtype, (hipStream_t)stream);
}

int nova_embed_canvas(const float* canvas, const float* mask, const void* w, const float* bias, const void* mask_token,
                      const void* pos_embed, void* z0, int B, int N, int P, int D, int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "embed_canvas: bad dtype %d", dtype);
  NOVA_REQUIRE(canvas && mask && w && bias && mask_token && z0, NOVA_ERR_ARG, "embed_canvas: null pointer");
  return embed_canvas(canvas, mask, w, bias, mask_token, pos_embed, z0, B, N, P, D, dtype, (hipStream_t)stream);
}

int nova_build_sequence(const void* prefix, long prefix_seq_rows, const void* tokens, long tok_batch_rows,
                        const long long* ids, void* x, int S, int B, int Lp, int n_sel, int D, int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "build_sequence: bad dtype %d", dtype);
  NOVA_REQUIRE(x && (Lp == 0 || prefix) && (n_sel == 0 || tokens) && B > 0, NOVA_ERR_ARG, "build_sequence: null pointer");
  return build_sequence(prefix, prefix_seq_rows, tokens, tok_batch_rows, ids, x, S, B, Lp, n_sel, D, dtype, (hipStream_t)stream);
}

int nova_scatter_tokens(const void* x1, const long long* ids, void* x2, int S, int B, int Lp, int N, int n_prev, int D,
                        int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "scatter_tokens: bad dtype %d", dtype);
  NOVA_REQUIRE(n_prev == 0 || (x1 && ids && x2 && B > 0), NOVA_ERR_ARG, "scatter_tokens: null pointer");
  return scatter_tokens(x1, ids, x2, S, B, Lp, N, n_prev, D, dtype, (hipStream_t)stream);
}

int nova_silu_add_rows(const void* a, const void* rowvec, void* out, long rows, int D, int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "silu_add_rows: bad dtype %d", dtype);
  NOVA_REQUIRE(rows == 0 || (a && out), NOVA_ERR_ARG, "silu_add_rows: null pointer");
  return silu_add_rows(a, rowvec, out, rows, D, dtype, (hipStream_t)stream);
}

int nova_timestep_freq(const float* t, const float* freq, void* out, int n, int freq_dim, int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "timestep_freq: bad dtype %d", dtype);
  NOVA_REQUIRE(t && freq && out && freq_dim % 2 == 0, NOVA_ERR_ARG, "timestep_freq: bad argument");
  return timestep_freq(t, freq, out, n, freq_dim, dtype, (hipStream_t)stream);
}

int nova_patch_embed_rows(const float* x, const void* w, const float* bias, void* out, int S, int B, int n, int P, int D,
                          int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "patch_embed_rows: bad dtype %d", dtype);
  NOVA_REQUIRE(x && w && bias && out && B > 0, NOVA_ERR_ARG, "patch_embed_rows: null pointer");
  return patch_embed_rows(x, w, bias, out, S, B, n, P, D, dtype, (hipStream_t)stream);
}

int nova_head_cfg_euler(const void* h, const void* w, const float* bias, float* x, int B, int n, int P, int D,
                        float guidance, int cfg, float dt, int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "head_cfg_euler: bad dtype %d", dtype);
  NOVA_REQUIRE(h && w && bias && x, NOVA_ERR_ARG, "head_cfg_euler: null pointer");
  // the kernel takes "guidance > 1" as the CFG switch; cfg != 0 with g <= 1 still combines (u + g (c - u))
  NOVA_REQUIRE(!cfg || guidance > 1.0f, NOVA_ERR_ARG, "head_cfg_euler: cfg needs guidance > 1");
  SamplerStep sp{cfg ? guidance : 1.0f, 0.f, 1.f, 0.f, dt, 1.f, 0.f, 0.f, 0};
  return head_cfg_step(h, w, bias, x, nullptr, nullptr, nullptr, nullptr, B, n, P, D, sp, 0, dtype, (hipStream_t)stream);
}


// Block stack of VisionTransformer.forward. With `cache` the k | v rows of every block are appended to that block's
// cache ([S][cap][2D], `cache_len` rows already valid) and attention runs over cache_len + L keys
// (vision_transformer.py:55-60, the conditioning encoder of multi-frame generation); without it over the L rows of x.
static int vit_blocks(const nova_vit_block* blocks, int nblocks, void* x, int S, int L, int D, int heads, int hidden,
                      const float* rope, int rope_batch, void* ws_qkv, void* ws_a, void* ws_b, void* ws_h, void* cache,
                      long cap, long cache_len, int dtype, hipStream_t st) {
  const int M = S * L, hd = D / heads;
  if (M == 0) return 0;
  const size_t es = esize(dtype);
  const float scale = 1.0f / sqrtf((float)hd);
  const char* qkv = static_cast<const char*>(ws_qkv);
  // every launch walks its row tiles in the opposite direction of the previous one (fresh-first reads, common.h)
  struct Walk {
    int k = 0;
    void next() { walk_reverse(g_walk_alternate && (k++ & 1)); }
    ~Walk() { walk_reverse(false); }
  } walk;
  for (int i = 0; i < nblocks; ++i) {
    const nova_vit_block& b = blocks[i];
    // bf16 / f16: the softmax scale (in the exp2 domain) is folded into q by the QKV epilogue, before the 16-bit rounding
    const bool pre = dtype_is16(dtype);
    walk.next();
    NOVA_TRY(gemm_qkv_rope(x, b.qkv_w, b.qkv_b, rope, ws_qkv, S, L, D, heads, rope_batch, dtype, st,
                           pre ? scale * 1.4426950408889634f : 1.0f));
    walk.next();
    if (cache) {
      char* cb = static_cast<char*>(cache) + (size_t)i * S * cap * 2 * D * es;
      NOVA_TRY(kv_append(ws_qkv, cb, S, L, D, cap, cache_len, dtype, st));
      NOVA_TRY(attn_fwd(qkv, cb, cb + (size_t)D * es, ws_a, S, heads, L, (int)(cache_len + L), hd, 3L * D, 2L * D, D, scale, dtype,
                        st, pre, cap * 2L * D));
    } else {
      NOVA_TRY(attn_fwd(qkv, qkv + (size_t)D * es, qkv + (size_t)2 * D * es, ws_a, S, heads, L, L, hd, 3L * D, 3L * D, D,
                        scale, dtype, st, pre));
    }
    walk.next();
    NOVA_TRY(gemm_bias_act(ws_a, b.proj_w, b.proj_b, ws_b, M, D, D, NOVA_ACT_NONE, dtype, st));
    RowNormArgs n1{ws_b, x, b.norm1_w, b.norm1_b, nullptr, 0, -1, -1, -1, x, nullptr, M, D, 1e-5f};
    walk.next();
    NOVA_TRY(row_norm(n1, dtype, st));
    walk.next();
    NOVA_TRY(gemm_bias_act(x, b.fc1_w, b.fc1_b, ws_h, M, hidden, D, NOVA_ACT_GELU_ERF, dtype, st));
    walk.next();
    NOVA_TRY(gemm_bias_act(ws_h, b.fc2_w, b.fc2_b, ws_b, M, D, hidden, NOVA_ACT_NONE, dtype, st));
    RowNormArgs n2{ws_b, x, b.norm2_w, b.norm2_b, nullptr, 0, -1, -1, -1, x, nullptr, M, D, 1e-5f};
    walk.next();
    NOVA_TRY(row_norm(n2, dtype, st));
  }
  return 0;
}

int nova_vit_blocks_forward(const nova_vit_block* blocks, int nblocks, void* x, int S, int L, int D, int heads,
                            int hidden, const float* rope, int rope_batch, void* ws_qkv, void* ws_a, void* ws_b,
                            void* ws_h, int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "vit_blocks: bad dtype %d", dtype);
  NOVA_REQUIRE(nblocks == 0 || (blocks && x && ws_qkv && ws_a && ws_b && ws_h), NOVA_ERR_ARG, "vit_blocks: null pointer");
  NOVA_REQUIRE(heads > 0 && D % heads == 0, NOVA_ERR_SHAPE, "vit_blocks: D %% heads != 0");
  return vit_blocks(blocks, nblocks, x, S, L, D, heads, hidden, rope, rope_batch, ws_qkv, ws_a, ws_b, ws_h, nullptr, 0, 0, dtype,
                    (hipStream_t)stream);
}

int nova_row_norm_fp8(const void* in, void* out, const float* gamma, const float* beta, const void* res, void* out8, float* out8_scale,
                      long rows, int D, float eps, void* stream) {
  NOVA_REQUIRE(rows == 0 || (in && out && out8 && out8_scale), NOVA_ERR_ARG, "row_norm_fp8: null pointer");
  RowNormArgs a{in, out, gamma, beta, nullptr, 0, -1, -1, -1, res, nullptr, rows, D, eps};
  a.out8 = out8;
  a.out8_scale = out8_scale;
  return row_norm(a, NOVA_BF16, (hipStream_t)stream);
}

int nova_gemm_fp8_gelu_q8(const void* A8, const float* a_scale, const void* W8, const float* w_scale, const float* bias, void* out8,
                          int M, int N, int K, const float* out_scale, unsigned* out_amax, void* stream) {
  NOVA_REQUIRE(M <= 0 || (A8 && a_scale && W8 && w_scale && out8 && out_scale && out_amax), NOVA_ERR_ARG, "gemm_fp8_gelu_q8: null pointer");
  return gemm256_fp8_launch(A8, a_scale, W8, w_scale, bias, out8, M, N, K, NOVA_EPI_GELU_Q8, (hipStream_t)stream, nullptr, 1, 1, 2, 0, 1.0f,
                            0, 0, out_scale, out_amax);
}

int nova_qkv_rope_fp8(const void* x8, const float* x_scale, const void* w8, const float* w_scale, const float* bias, const float* rope,
                      void* qkv, int S, int L, int D, int heads, int rope_batch, float q_scale, void* stream) {
  NOVA_REQUIRE(S * L == 0 || (x8 && x_scale && w8 && w_scale && qkv), NOVA_ERR_ARG, "qkv_rope_fp8: null pointer");
  NOVA_REQUIRE(heads > 0 && D % heads == 0, NOVA_ERR_SHAPE, "qkv_rope_fp8: D %% heads != 0");
  return gemm256_fp8_launch(x8, x_scale, w8, w_scale, bias, qkv, S * L, 3 * D, D, 3, (hipStream_t)stream, rope, L, rope_batch, D / heads,
                            rope ? 2 * D : 0, q_scale, q_scale != 1.0f ? D : 0);
}

// The block stack with the three large GEMMs of every block (fused QKV, fc1, fc2 = 11/12 of a block's GEMM FLOPs) on the
// MX-fp8 MFMA path (BASELINE configs[4]): weights quantised once per output row at pack time, activations quantised per
// row on the fly - the residual stream by the LayerNorm kernel that produces it (fp8 side output), the MLP hidden rows by
// one quantisation pass. Attention, its out-projection, LayerNorm statistics and the residual stream stay bf16 / f32.
int nova_vit_blocks_forward_fp8(const nova_vit_block* blocks, const nova_vit_block_fp8* q, int nblocks, void* x, int S, int L, int D,
                                int heads, int hidden, const float* rope, int rope_batch, void* ws_qkv, void* ws_a, void* ws_b,
                                void* ws_h, void* ws_x8, float* ws_xs, void* ws_h8, float* ws_hs, float* h_scale, unsigned* h_amax,
                                void* stream) {
  NOVA_REQUIRE(nblocks == 0 || (blocks && q && x && ws_qkv && ws_a && ws_b && ws_h && ws_x8 && ws_xs && ws_h8 && ws_hs), NOVA_ERR_ARG,
               "vit_blocks_fp8: null pointer");
  NOVA_REQUIRE((h_scale == nullptr) == (h_amax == nullptr), NOVA_ERR_ARG, "vit_blocks_fp8: h_scale and h_amax come together");
  NOVA_REQUIRE(heads > 0 && D % heads == 0, NOVA_ERR_SHAPE, "vit_blocks_fp8: D %% heads != 0");
  NOVA_REQUIRE(D % 256 == 0 && hidden % 256 == 0, NOVA_ERR_SHAPE, "vit_blocks_fp8: D and hidden must be multiples of 256");
  NOVA_REQUIRE(!rope || L >= 16, NOVA_ERR_SHAPE, "vit_blocks_fp8: L >= 16 with RoPE");
  hipStream_t st = (hipStream_t)stream;
  const int M = S * L, hd = D / heads;
  if (M == 0 || nblocks == 0) return 0;
  const float scale = 1.0f / sqrtf((float)hd);
  const char* qkv = static_cast<const char*>(ws_qkv);
  struct Walk {
    int k = 0;
    void next() { walk_reverse(g_walk_alternate && (k++ & 1)); }
    ~Walk() { walk_reverse(false); }
  } walk;
  NOVA_TRY(quantize_rows_fp8(x, ws_x8, ws_xs, M, D, st));  // the stack's input rows; later ones come from the LN kernels
  for (int i = 0; i < nblocks; ++i) {
    const nova_vit_block& b = blocks[i];
    const nova_vit_block_fp8& w = q[i];
    walk.next();
    NOVA_TRY(gemm256_fp8_launch(ws_x8, ws_xs, w.qkv_w8, w.qkv_ws, b.qkv_b, ws_qkv, M, 3 * D, D, 3 /* RoPE + q-scale */, st, rope, L,
                                rope_batch, hd, rope ? 2 * D : 0, scale * 1.4426950408889634f, D));
    walk.next();
    NOVA_TRY(attn_fwd(qkv, qkv + (size_t)D * 2, qkv + (size_t)2 * D * 2, ws_a, S, heads, L, L, hd, 3L * D, 3L * D, D, scale, NOVA_BF16,
                      st, true));
    walk.next();
    NOVA_TRY(gemm_bias_act(ws_a, b.proj_w, b.proj_b, ws_b, M, D, D, NOVA_ACT_NONE, NOVA_BF16, st));
    RowNormArgs n1{ws_b, x, b.norm1_w, b.norm1_b, nullptr, 0, -1, -1, -1, x, nullptr, M, D, 1e-5f};
    n1.out8 = ws_x8;
    n1.out8_scale = ws_xs;
    walk.next();
    NOVA_TRY(row_norm(n1, NOVA_BF16, st));
    walk.next();
    if (h_scale) {
      // delayed scaling: fc1 writes GELU(.) / h_scale[i] as e4m3 itself and records max |GELU(.)| in h_amax[i], from which the
      // caller sets the scale of the next call; fc2 reads one scale for all rows
      NOVA_TRY(gemm256_fp8_launch(ws_x8, ws_xs, w.fc1_w8, w.fc1_ws, b.fc1_b, ws_h8, M, hidden, D, NOVA_EPI_GELU_Q8, st, nullptr, 1, 1, 2, 0,
                                  1.0f, 0, 0, h_scale + i, h_amax + i));
      walk.next();
      NOVA_TRY(gemm256_fp8_launch(ws_h8, h_scale + i, w.fc2_w8, w.fc2_ws, b.fc2_b, ws_b, M, D, hidden, NOVA_ACT_NONE, st, nullptr, 1, 1, 2, 0,
                                  1.0f, 0, 1));
    } else {
      NOVA_TRY(gemm256_fp8_launch(ws_x8, ws_xs, w.fc1_w8, w.fc1_ws, b.fc1_b, ws_h, M, hidden, D, NOVA_ACT_GELU_ERF, st));
      NOVA_TRY(quantize_rows_fp8(ws_h, ws_h8, ws_hs, M, hidden, st));
      walk.next();
      NOVA_TRY(gemm256_fp8_launch(ws_h8, ws_hs, w.fc2_w8, w.fc2_ws, b.fc2_b, ws_b, M, D, hidden, NOVA_ACT_NONE, st));
    }
    RowNormArgs n2{ws_b, x, b.norm2_w, b.norm2_b, nullptr, 0, -1, -1, -1, x, nullptr, M, D, 1e-5f};
    if (i + 1 < nblocks) {  // the next block's QKV input
      n2.out8 = ws_x8;
      n2.out8_scale = ws_xs;
    }
    walk.next();
    NOVA_TRY(row_norm(n2, NOVA_BF16, st));
  }
  return 0;
}

int nova_vit_blocks_forward_kv(const nova_vit_block* blocks, int nblocks, void* x, int S, int L, int D, int heads,
                               int hidden, const float* rope, int rope_batch, void* kv_cache, long cache_cap,
                               long cache_len, void* ws_qkv, void* ws_a, void* ws_b, void* ws_h, int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "vit_blocks_kv: bad dtype %d", dtype);
  NOVA_REQUIRE(nblocks == 0 || (blocks && x && kv_cache && ws_qkv && ws_a && ws_b && ws_h), NOVA_ERR_ARG, "vit_blocks_kv: null pointer");
  NOVA_REQUIRE(heads > 0 && D % heads == 0, NOVA_ERR_SHAPE, "vit_blocks_kv: D %% heads != 0");
  NOVA_REQUIRE(cache_len >= 0 && cache_len + L <= cache_cap, NOVA_ERR_SHAPE, "vit_blocks_kv: %ld cached + %d new rows exceed the capacity %ld",
               cache_len, L, cache_cap);
  return vit_blocks(blocks, nblocks, x, S, L, D, heads, hidden, rope, rope_batch, ws_qkv, ws_a, ws_b, ws_h, kv_cache, cache_cap,
                    cache_len, dtype, (hipStream_t)stream);
}

int nova_pointset_nn_dist(const float* x, const float* y, float* d, int B, int N, int M, float clamp_lo, float clamp_hi,
                          int unit_norm, void* stream) {
  NOVA_REQUIRE(B * (long)N == 0 || (x && y && d), NOVA_ERR_ARG, "pointset_nn_dist: null pointer");
  NOVA_REQUIRE(clamp_lo <= clamp_hi, NOVA_ERR_ARG, "pointset_nn_dist: empty clamp range");
  return pointset_nn_dist(x, y, d, B, N, M, clamp_lo, clamp_hi, unit_norm, (hipStream_t)stream);
}

int nova_pointset_pairwise_dist(const float* x, const float* y, float* D, int B, int N, int M, float clamp_lo, float clamp_hi,
                                void* stream) {
  NOVA_REQUIRE(B * (long)N * M == 0 || (x && y && D), NOVA_ERR_ARG, "pointset_pairwise_dist: null pointer");
  NOVA_REQUIRE(clamp_lo <= clamp_hi, NOVA_ERR_ARG, "pointset_pairwise_dist: empty clamp range");
  return pointset_pairwise_dist(x, y, D, B, N, M, clamp_lo, clamp_hi, (hipStream_t)stream);
}

int nova_modulate_rows(const void* x, const void* mod, void* out, long rows, int D, int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "modulate_rows: bad dtype %d", dtype);
  NOVA_REQUIRE(rows == 0 || (x && mod && out), NOVA_ERR_ARG, "modulate_rows: null pointer");
  return modulate_rows(x, mod, out, rows, D, dtype, (hipStream_t)stream);
}

}  // extern "C"

namespace nova {

// The launch sequence of one AR step's denoising loop (called directly, or once under stream capture: see below).
static int decoder_denoise_launches(const nova_decoder* dec, const void* zc, const void* temb, float* x, const nova_sampler_step* sched,
                                    const float* noise, float renorm, float* echo_energy, int steps, int S, int B, int n, int P, int D,
                                    void* ws_a, void* ws_u, void* ws_h, void* ws_f, void* ws_g, void* ws_mod, float* ws_v, int mod_steps,
                                    int dtype, hipStream_t st, float* echo_rows = nullptr, const float* echo_noise = nullptr, int Ne = 0) {
  const bool do_renorm = renorm < 1.0f;
  const size_t es = esize(dtype);
  const int depth = dec->depth;
  const long mod_ld = (long)(3 * depth + 2) * D;
  const size_t nP = (size_t)B * n * P;
  NOVA_REQUIRE(mod_steps == 1 || mod_steps == steps, NOVA_ERR_ARG, "decoder_denoise: mod_steps must be 1 or steps");
  // mod_steps == steps: the AdaLN projections of ALL steps in one GEMM ahead of the loop (M = steps * S * n rows: the large-M
  // kernel at ~2.5x the rate of the per-step small-M launches, and two launches fewer per step). Rows are laid out per
  // step for the full S sequences; a step that runs fewer guidance passes (guidance_trunc) reads its leading rows.
  const bool hoist = mod_steps == steps && steps > 1;
  const long rows_all = (long)S * n;
  if (hoist) {
    NOVA_TRY(silu_add_steps(zc, temb, ws_a, rows_all, steps, D, dtype, st));
    NOVA_TRY(gemm_bias_act(ws_a, dec->adaln_w, dec->adaln_b, ws_mod, (int)(rows_all * steps), (int)mod_ld, D, NOVA_ACT_NONE, dtype, st));
  }
  void* const ws_mod_base = ws_mod;
  for (int i = 0; i < steps; ++i) {
    const SamplerStep sp{sched[i].guidance, sched[i].kx,    sched[i].kv,          sched[i].clip,      sched[i].c0,
                         sched[i].cx,       sched[i].sigma, sched[i].extra_scale, sched[i].extra_kind};
    const int cfg = sp.guidance > 1.0f ? 1 : 0;
    NOVA_REQUIRE(sp.extra_kind >= 0 && sp.extra_kind <= 2, NOVA_ERR_ARG, "decoder_denoise: extra_kind must be 0, 1 or 2");
    const int passes = cfg ? (sp.extra_kind ? 3 : 2) : 1;
    NOVA_REQUIRE(S >= passes * B, NOVA_ERR_SHAPE, "decoder_denoise: step %d needs %d guidance passes but S = %d, B = %d", i, passes, S, B);
    NOVA_REQUIRE(!(do_renorm && cfg) || echo_rows || (sp.kx == 0.f && sp.kv == 1.f && sp.cx == 1.f && sp.sigma == 0.f && sp.clip <= 0.f),
                 NOVA_ERR_ARG, "decoder_denoise: guidance renorm with a scalar echo energy holds for the flow-matching Euler step only "
                 "(other samplers: nova_decoder_denoise_echo with the echo rows)");
    const int Se = passes * B;
    const long rows = (long)Se * n;
    if (hoist) {
      ws_mod = static_cast<char*>(ws_mod_base) + (size_t)i * rows_all * mod_ld * es;
    } else {
      NOVA_TRY(silu_add_rows(zc, static_cast<const char*>(temb) + (size_t)i * D * es, ws_a, rows, D, dtype, st));
      NOVA_TRY(gemm_bias_act(ws_a, dec->adaln_w, dec->adaln_b, ws_mod, (int)rows, (int)mod_ld, D, NOVA_ACT_NONE, dtype, st));
    }
    NOVA_TRY(patch_embed_rows(x, dec->patch_w, dec->patch_b, ws_u, Se, B, n, P, D, dtype, st));
    // Per block: m1 (modulate -> h), fc1 + SiLU, fc2, m2 (gated norm + residual -> u). Where the small-M kernel does not take m1 + fc1 as one
    // launch (more than ~320 rows: every step of the batch-32 workload but the first few), m2 of block b and m1 of block b + 1 are ONE row pass
    // (row_norm_chain: u is written, h comes out of the same registers; bit-identical to the two launches), so a block is 3 launches instead of 4.
    const bool m1_fused = gemm_modulate_fused((int)rows, D, D, dtype);  // m1 rides inside the fc1 launch (skinny.hip)
    const bool chain = !m1_fused && dtype_is16(dtype);
    for (int b = 0; b < depth; ++b) {
      const nova_mlp_block& blk = dec->blocks[b];
      RowNormArgs m1{ws_u, ws_h, nullptr, nullptr, ws_mod, mod_ld, b * 3 * D, b * 3 * D + D, -1, nullptr, nullptr, rows, D, 1e-6f};
      if (chain && b > 0) {
        NOVA_TRY(gemm_bias_act(ws_h, blk.fc1_w, blk.fc1_b, ws_f, (int)rows, D, D, NOVA_ACT_SILU, dtype, st));  // h: the previous block's chain launch
      } else {
        NOVA_TRY(gemm_modulate_act(m1, blk.fc1_w, blk.fc1_b, ws_f, (int)rows, D, D, NOVA_ACT_SILU, dtype, st));  // one launch at small M
      }
      NOVA_TRY(gemm_bias_act(ws_f, blk.fc2_w, blk.fc2_b, ws_g, (int)rows, D, D, NOVA_ACT_NONE, dtype, st));
      RowNormArgs m2{ws_g, ws_u, blk.norm2_w, blk.norm2_b, ws_mod, mod_ld, -1, -1, b * 3 * D + 2 * D, ws_u, nullptr, rows, D, 1e-5f};
      RowNormArgs mf{ws_u, ws_h, nullptr, nullptr, ws_mod, mod_ld, depth * 3 * D, depth * 3 * D + D, -1, nullptr, nullptr, rows, D, 1e-6f};
      if (b == depth - 1 && dtype_is16(dtype)) {
        // last block: its gated norm + residual and the final layer's modulate in ONE row pass (x itself is not needed
        // any more, so it is neither written nor read back); bit-identical to the two launches (rownorm.h)
        NOVA_TRY(row_norm_chain(m2, mf, nullptr, dtype, st));
      } else if (chain) {
        RowNormArgs m1n{ws_u, ws_h, nullptr, nullptr, ws_mod, mod_ld, (b + 1) * 3 * D, (b + 1) * 3 * D + D, -1, nullptr, nullptr, rows, D, 1e-6f};
        NOVA_TRY(row_norm_chain(m2, m1n, ws_u, dtype, st));  // u <- m2 in place (row-local), h <- m1 of the next block
      } else {
        NOVA_TRY(row_norm(m2, dtype, st));
        if (b == depth - 1) NOVA_TRY(row_norm(mf, dtype, st));
      }
    }
    const void* head_in = ws_h;
    const float* nz = (noise && sp.sigma != 0.f) ? noise + (size_t)i * nP : nullptr;
    if (do_renorm && echo_rows) {  // any sampler step: the echo rows are carried explicitly (rowops.hip: renorm_step_kernel)
      const float* enz = (echo_noise && sp.sigma != 0.f) ? echo_noise + (size_t)i * B * Ne * P : nullptr;
      if (cfg) {
        float* extra = passes == 3 ? ws_v + 2 * nP : nullptr;
        NOVA_TRY(head_cfg_step(head_in, dec->head_w, dec->head_b, x, nullptr, ws_v, ws_v + nP, extra, B, n, P, D, sp, 1, dtype, st));
        NOVA_TRY(renorm_step(x, ws_v, ws_v + nP, extra, nz, echo_rows, enz, B, n * P, Ne * P, sp, renorm, 0, st));
      } else {
        NOVA_TRY(head_cfg_step(head_in, dec->head_w, dec->head_b, x, nz, nullptr, nullptr, nullptr, B, n, P, D, sp, 0, dtype, st));
        NOVA_TRY(renorm_step(nullptr, nullptr, nullptr, nullptr, nullptr, echo_rows, enz, B, n * P, Ne * P, sp, renorm, 1, st));
      }
    } else if (do_renorm && cfg) {  // guidance_scaler.py:67-72: two small launches, norms over the whole sample
      float* extra = passes == 3 ? ws_v + 2 * nP : nullptr;
      NOVA_TRY(head_cfg_step(head_in, dec->head_w, dec->head_b, x, nullptr, ws_v, ws_v + nP, extra, B, n, P, D, sp, 1, dtype, st));
      NOVA_TRY(renorm_euler(x, ws_v, ws_v + nP, extra, echo_energy, B, n, P, sp.c0, renorm, st));
    } else {
      NOVA_TRY(head_cfg_step(head_in, dec->head_w, dec->head_b, x, nz, nullptr, nullptr, nullptr, B, n, P, D, sp, 0, dtype, st));
      // guidance switched off for this step (guidance_trunc): no renorm, but the echo rows still take the Euler step
      if (do_renorm && echo_energy) NOVA_TRY(scale_vector(echo_energy, B, (1.0f + sp.c0) * (1.0f + sp.c0), st));
    }
  }
  return 0;
}


// ---- hipGraph replay of the denoising loop.
// One AR step issues 25 x (15 to 27) launches of 4-15 us kernels from one host thread. The sequence is a pure function
// of the call's arguments, so it is captured
// once per distinct argument set - pointers, shapes, the per-step sampler plan and the decoder's weight pointers all
// go into the key - and replayed with one hipGraphLaunch afterwards. The engine keeps every buffer of the call at a
// stable address (workspace slots), so from the second generation on every AR step of a fixed schedule is a replay.
// Measured (tools/host_vs_gpu.py): the host thread's time in a batch-8 generation call drops 907 -> 795 ms; wall time is
// unchanged at batch 8 and 32 (the device, not the launch rate, bounds both), the first call of a schedule pays ~3 ms
// per captured graph.
// Calls that read per-call tensors the engine allocates afresh (ancestral noise, guidance-renorm scratch) and profiled
// runs (HIP-event brackets around launches) stay on the direct path.
static std::atomic<long> g_graph_epoch{0};  // bumped by nova_debug_set_graphs(0): every thread drops its cache at its next call
struct DecoderGraphs {
  std::unordered_map<std::string, hipGraphExec_t> execs;
  long captures = 0, replays = 0, epoch = 0;
  ~DecoderGraphs() { clear(false); }  // thread exit / process teardown: the runtime may already be going down, no device call but the destroys
  // A launch of one of these execs may still be queued or running on some stream (the host runs ahead of the device, and the caller
  // drops the cache exactly when it is about to re-plan: a workspace that changed shape, graphs switched off): wait for the device
  // before the execs and their kernel-argument storage go. Rare by construction (shape changes), so the wait is not a cost.
  void clear(bool wait = true) {
    if (wait && !execs.empty()) (void)hipDeviceSynchronize();
    for (auto& kv : execs) (void)hipGraphExecDestroy(kv.second);
    execs.clear();
  }
};
static thread_local DecoderGraphs g_dec_graphs;
static std::atomic<int> g_graphs_on{-1};  // -1: not decided yet (NOVA_GRAPHS env, default on)

static bool graphs_enabled() {
  int v = g_graphs_on.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("NOVA_GRAPHS");
    v = (e && e[0] == '0') ? 0 : 1;
    g_graphs_on.store(v);
  }
  return v != 0;
}

template <typename T> static void key_put(std::string& k, const T& v) { k.append(reinterpret_cast<const char*>(&v), sizeof(T)); }

}  // namespace nova

extern "C" {

int nova_debug_set_attn_variant(int variant) {
  NOVA_REQUIRE(attn_set_variant(variant) == 0, NOVA_ERR_ARG, "nova_debug_set_attn_variant: %d is not one of -1 .. 5", variant);
  return 0;
}

int nova_debug_set_graphs(int on) {
  g_graphs_on.store(on ? 1 : 0);
  if (!on) {  // the calling thread's graphs go now, every other thread's at its next nova_decoder_denoise (the caches are thread-local)
    g_graph_epoch.fetch_add(1);
    g_dec_graphs.clear();
  }
  return 0;
}

int nova_debug_drop_graphs(void) {
  g_dec_graphs.clear();
  return 0;
}

int nova_debug_graph_stats(long* captures, long* replays) {
  if (captures) *captures = g_dec_graphs.captures;
  if (replays) *replays = g_dec_graphs.replays;
  return 0;
}

int nova_decoder_denoise(const nova_decoder* dec, const void* zc, const void* temb, float* x, const nova_sampler_step* sched,
                         const float* noise, float renorm, float* echo_energy, int steps, int S, int B, int n, int P, int D,
                         void* ws_a, void* ws_u, void* ws_h, void* ws_f, void* ws_g, void* ws_mod, float* ws_v, int mod_steps,
                         int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "decoder_denoise: bad dtype %d", dtype);
  NOVA_REQUIRE(dec && zc && temb && x && sched && ws_a && ws_u && ws_h && ws_f && ws_g && ws_mod, NOVA_ERR_ARG,
               "decoder_denoise: null pointer");
  NOVA_REQUIRE(S == B || S == 2 * B || S == 3 * B, NOVA_ERR_SHAPE, "decoder_denoise: S must be B, 2B or 3B");
  NOVA_REQUIRE(dec->depth >= 1 && dec->blocks, NOVA_ERR_ARG, "decoder_denoise: the decoder needs at least one block");
  const bool do_renorm = renorm < 1.0f;
  NOVA_REQUIRE(!do_renorm || (echo_energy && ws_v), NOVA_ERR_ARG, "decoder_denoise: renorm needs echo_energy and ws_v");
  if (n == 0 || B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const bool direct = !graphs_enabled() || g_prof_on.load(std::memory_order_relaxed) || noise != nullptr || do_renorm || steps < 2;
  if (direct)
    return decoder_denoise_launches(dec, zc, temb, x, sched, noise, renorm, echo_energy, steps, S, B, n, P, D, ws_a, ws_u, ws_h, ws_f,
                                    ws_g, ws_mod, ws_v, mod_steps, dtype, st);
  if (const long e = g_graph_epoch.load(std::memory_order_relaxed); e != g_dec_graphs.epoch) {
    g_dec_graphs.clear();
    g_dec_graphs.epoch = e;
  }
  std::string key;
  key.reserve(512 + sizeof(nova_sampler_step) * steps + sizeof(nova_mlp_block) * dec->depth);
  key_put(key, *dec);
  for (int b = 0; b < dec->depth; ++b) key_put(key, dec->blocks[b]);
  for (int i = 0; i < steps; ++i) key_put(key, sched[i]);
  const void* ptrs[] = {zc, temb, x, ws_a, ws_u, ws_h, ws_f, ws_g, ws_mod, (const void*)st};
  key_put(key, ptrs);
  const int ints[] = {steps, S, B, n, P, D, mod_steps, dtype, walk_is_reverse() ? 1 : 0, gemm_forced_tile(), skinny_forced_row_blocks()};
  key_put(key, ints);
  auto it = g_dec_graphs.execs.find(key);
  if (it == g_dec_graphs.execs.end()) {
    if (g_dec_graphs.execs.size() >= 2048) g_dec_graphs.clear();  // schedules come and go: start over rather than grow
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      (void)hipGetLastError();
      return decoder_denoise_launches(dec, zc, temb, x, sched, noise, renorm, echo_energy, steps, S, B, n, P, D, ws_a, ws_u, ws_h, ws_f,
                                      ws_g, ws_mod, ws_v, mod_steps, dtype, st);
    }
    const int rc = decoder_denoise_launches(dec, zc, temb, x, sched, noise, renorm, echo_energy, steps, S, B, n, P, D, ws_a, ws_u, ws_h,
                                            ws_f, ws_g, ws_mod, ws_v, mod_steps, dtype, st);
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(st, &graph);
    if (rc != 0) {
      if (graph) (void)hipGraphDestroy(graph);
      return rc;
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = (ec == hipSuccess && graph) ? hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) : ec;
    if (graph) (void)hipGraphDestroy(graph);
    if (ei != hipSuccess || !exec) {  // nothing has executed yet (the launches above were only recorded): run them directly
      (void)hipGetLastError();
      return decoder_denoise_launches(dec, zc, temb, x, sched, noise, renorm, echo_energy, steps, S, B, n, P, D, ws_a, ws_u, ws_h, ws_f,
                                      ws_g, ws_mod, ws_v, mod_steps, dtype, st);
    }
    it = g_dec_graphs.execs.emplace(std::move(key), exec).first;
    ++g_dec_graphs.captures;
  } else {
    ++g_dec_graphs.replays;
  }
  const hipError_t el = hipGraphLaunch(it->second, st);
  NOVA_REQUIRE(el == hipSuccess, NOVA_ERR_LAUNCH, "decoder_denoise: hipGraphLaunch failed: %s", hipGetErrorString(el));
  return 0;
}

int nova_decoder_denoise_echo(const nova_decoder* dec, const void* zc, const void* temb, float* x, const nova_sampler_step* sched,
                              const float* noise, float renorm, float* echo_rows, const float* echo_noise, int Ne, int steps, int S, int B,
                              int n, int P, int D, void* ws_a, void* ws_u, void* ws_h, void* ws_f, void* ws_g, void* ws_mod, float* ws_v,
                              int mod_steps, int dtype, void* stream) {
  NOVA_REQUIRE(!bad_dtype(dtype), NOVA_ERR_ARG, "decoder_denoise_echo: bad dtype %d", dtype);
  NOVA_REQUIRE(dec && zc && temb && x && sched && ws_a && ws_u && ws_h && ws_f && ws_g && ws_mod, NOVA_ERR_ARG,
               "decoder_denoise_echo: null pointer");
  NOVA_REQUIRE(S == B || S == 2 * B || S == 3 * B, NOVA_ERR_SHAPE, "decoder_denoise_echo: S must be B, 2B or 3B");
  NOVA_REQUIRE(dec->depth >= 1 && dec->blocks, NOVA_ERR_ARG, "decoder_denoise_echo: the decoder needs at least one block");
  NOVA_REQUIRE(renorm < 1.0f && echo_rows && ws_v && Ne >= 0, NOVA_ERR_ARG,
               "decoder_denoise_echo: for guidance_renorm < 1 with the echo rows given (else: nova_decoder_denoise)");
  if (n == 0 || B == 0) return 0;
  return decoder_denoise_launches(dec, zc, temb, x, sched, noise, renorm, nullptr, steps, S, B, n, P, D, ws_a, ws_u, ws_h, ws_f, ws_g, ws_mod,
                                  ws_v, mod_steps, dtype, (hipStream_t)stream, echo_rows, echo_noise, Ne);
}

}  // extern "C"
