// NOVA hot path: dense projection GEMM  out[M,N] = epi(A[M,K] · W[N,K]^T + bias[N])
//
// Replaces, on the device, every nn.Linear of the reference's ViT blocks and diffusion MLP
// (reference diffnext/models/vision_transformer.py:33-38,47-48,52,64; diffusion_mlp.py:31-36;
// normalization.py:28,35). Weights keep the reference's nn.Linear layout [N][K] (K contiguous),
// so both MFMA operands are K-contiguous and no transpose is ever materialised.
//
// Structure (MI355X-first, not a port): 128x128 output tile per 256-thread workgroup (4 waves as
// 2x2, 64x64 per wave = 4x4 MFMA 16x16 fragments), K-tile of 128 BYTES per row (64 bf16 / 32 f32),
// both operand tiles streamed global->LDS with 16-byte LDS-DMA (global_load_lds_dwordx4), double
// buffered, one barrier per K-tile. The LDS image is lane-linear (DMA requirement), so the
// bank-conflict XOR swizzle is applied to the per-lane SOURCE address and to the ds_read_b128
// address (cdna_hip_programming.md rule 21). The MFMA is issued with the WEIGHT fragment as the
// A operand so each lane ends up holding 4 consecutive output columns of one row: bias / GELU /
// SiLU / RoPE pairs are lane-local and the store is 8 B (bf16) or 16 B (f32) per lane.
// bf16: v_mfma_f32_16x16x32_bf16; f32 (parity mode): v_mfma_f32_16x16x4_f32 (exact f32).
#include <cstdlib>
#include "common.h"
#include "nova_internal.h"

namespace nova {

constexpr int ROWB = 128;       // bytes of K per tile row

struct GemmEpi {
  const float* bias;     // [N] or nullptr
  const float* rope;     // [rope_batch, L, hd/2, 2] (cos, sin) or nullptr
  int L;                 // tokens per sequence (rows m -> (s = m / L, l = m % L))
  int rope_batch;        // table batch count; sequence s uses table s % rope_batch
  int hd;                // head dim
  int rope_cols;         // columns [0, rope_cols) are rotated (q and k thirds of the fused QKV)
  float q_scale;         // columns [0, q_cols) are multiplied by this after rotation (softmax scale folded into q)
  int q_cols;
};

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_SILU = 2, EPI_ROPE = 3 };

template <typename T> struct Frag;   // one 16-byte LDS read = the per-lane K slice of a fragment
template <> struct Frag<bf16_t> { u4v v; };
template <> struct Frag<f16_t> { u4v v; };
template <> struct Frag<float> { f4v v; };

__device__ __forceinline__ f4v mma(const Frag<bf16_t>& w, const Frag<bf16_t>& a, f4v c) { return Half16<bf16_t>::mfma16(w.v, a.v, c); }
__device__ __forceinline__ f4v mma(const Frag<f16_t>& w, const Frag<f16_t>& a, f4v c) { return Half16<f16_t>::mfma16(w.v, a.v, c); }
__device__ __forceinline__ f4v mma(const Frag<float>& w, const Frag<float>& a, f4v c) {
  // 4 exact-f32 MFMAs; lane group g = lane>>4 supplies k-slot g of each, so the logical k of
  // (chunk, j) is 4*chunk + j on BOTH operands (any consistent k order is a valid contraction).
#pragma unroll
  for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v[j], a.v[j], c, 0, 0, 0);
  return c;
}

template <typename T>
__device__ __forceinline__ Frag<T> lds_frag(const char* tile, int row, int chunk) {
  const int phys = chunk ^ ((row >> 1) & 7);
  Frag<T> f;
  f.v = *reinterpret_cast<const decltype(f.v)*>(tile + row * ROWB + phys * 16);
  return f;
}

// F = MFMA fragments per wave and dimension: 4 -> the 128x128 tile described above; 2 -> a 64x64 tile (round 4) for the denoising
// loop's launches at 1100 .. 4000 rows, where the 128 tile gives fewer workgroups than CUs and a workgroup's LDS-DMA rate, not the chip's,
// sets the time: a quarter of the operand bytes per workgroup, four times the workgroups. Same MFMA, same k order: bit-identical.
#ifndef NOVA_GEMM64_STAGES
#define NOVA_GEMM64_STAGES 2  // K-tile ring depth of the 64 x 64 tile: 3, 4 and 5 measured SLOWER than 2 (tools/gemm64_stages_ab.py, profiles/r04_gemm64_stages_ab.txt)
#endif
template <typename T, int EPI, int F = 4>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                       T* __restrict__ C, int M, int N, int K, int ntm,
                                                       int ntn, GemmEpi e) {
  constexpr int BM = 32 * F, BN = 32 * F, TILE_BYTES = BM * ROWB, WT = 16 * F;  // tile, bytes of one operand tile, rows / columns per wave
  // K-tile buffers: 2 (a 4-deep ring measured 5 % slower for the 128 tile; for the 64 tile, whose K-tile is only 8 MFMAs per wave, rings of 3 - 5 were
  // 4 - 11 % slower as well: what hides a K-tile's load latency there is the number of workgroups a CU holds, and a deeper ring takes that away)
  constexpr int NST = F == 2 ? NOVA_GEMM64_STAGES : 2;
  __shared__ __attribute__((aligned(16))) char smem[NST * 2 * TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  // XCD-aware + grouped tile order: 8 row panels x all column panels per group, row fastest.
  const int nwg = ntm * ntn;
  const int t = xcd_remap(blockIdx.x, nwg);
  constexpr int GM = 8;
  const int per_group = GM * ntn;
  const int group = t / per_group, first_m = group * GM;
  const int gsz = min(ntm - first_m, GM);
  const int tm = first_m + (t % per_group) % gsz;
  const int tn = (t % per_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging addresses: wave w owns LDS-DMA pieces F w .. F w + F - 1 (8 rows x 128 B each) of A and W
  const int rr = lane >> 3, cp = lane & 7;
  const char* a_src[4];  // (F used; sized by the constant: an array sized by the template parameter and captured by the lambda
  const char* w_src[4];  //  below makes hipcc drop the kernel's host stub - no diagnostic, an undefined symbol at load time)
  const size_t rowbytes = (size_t)K * sizeof(T);
#pragma unroll
  for (int i = 0; i < F; ++i) {
    const int row = (wid * F + i) * 8 + rr;                 // row inside the tile
    const int c = cp ^ ((row >> 1) & 7);                    // logical 16-B chunk this lane fetches
    const int am = min(m0 + row, M - 1);                    // clamp: rows past M are never stored
    a_src[i] = reinterpret_cast<const char*>(A) + (size_t)am * rowbytes + c * 16;
    w_src[i] = reinterpret_cast<const char*>(W) + (size_t)(n0 + row) * rowbytes + c * 16;
  }
  auto stage = [&](int buf, int kt) {
    char* la = smem + buf * 2 * TILE_BYTES + wid * F * 1024;
    char* lw = la + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < F; ++i) {
      __builtin_amdgcn_global_load_lds(a_src[i] + (size_t)kt * ROWB, NOVA_LDS_PTR(la + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(w_src[i] + (size_t)kt * ROWB, NOVA_LDS_PTR(lw + i * 1024), 16, 0, 0);
    }
  };

  const int nkt = K / (ROWB / (int)sizeof(T));
  const int fr = lane & 15, fg = lane >> 4;

  // The accumulators START at the bias (all three GEMM kernels do, so that they stay bit-identical): no add in the
  // epilogue, where every VALU issue is paid with the matrix pipe idle.
  f4v bv[F];
#pragma unroll
  for (int nf = 0; nf < F; ++nf) bv[nf] = f4v{0.f, 0.f, 0.f, 0.f};
  if (e.bias) {
#pragma unroll
    for (int nf = 0; nf < F; ++nf) bv[nf] = *reinterpret_cast<const f4v*>(e.bias + n0 + wn * WT + nf * 16 + fg * 4);
  }
  f4v acc[F][F];  // [nf][mf]
#pragma unroll
  for (int i = 0; i < F; ++i)
#pragma unroll
    for (int j = 0; j < F; ++j) acc[i][j] = bv[i];

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nkt) stage(s, s);
  for (int kt = 0; kt < nkt; ++kt) {
    // tile kt landed for this wave, then for all of them; the buffer of tile kt - 1 is free. The wait is explicit: whether the
    // compiler drains vmcnt for LDS-DMA before a barrier depends on what else it has in flight (with the bias loads
    // ahead of the loop it stopped doing so). With NST > 2 it is counted: the NST - 2 younger tiles (2 F requests each) stay in flight.
    if constexpr (NST == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      const int young = min(NST - 2, nkt - 1 - kt);
      if (young >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * 2 * F) : "memory");
      else if (young == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 2 * F) : "memory");
      else if (young == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * 2 * F) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // NST > 2: the bare barrier - __syncthreads() carries a fence that drains every outstanding request (vmcnt(0)), i.e. the ring
    if constexpr (NST == 2) __syncthreads();
    else asm volatile("s_barrier" ::: "memory");
    if (kt + NST - 1 < nkt) stage((kt + NST - 1) % NST, kt + NST - 1);
    const char* ta = smem + (kt % NST) * 2 * TILE_BYTES;
    const char* tw = ta + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      Frag<T> af[F], wf[F];
#pragma unroll
      for (int f = 0; f < F; ++f) {
        af[f] = lds_frag<T>(ta, wm * WT + f * 16 + fr, fg + 4 * kk);
        wf[f] = lds_frag<T>(tw, wn * WT + f * 16 + fr, fg + 4 * kk);
      }
#pragma unroll
      for (int nf = 0; nf < F; ++nf)
#pragma unroll
        for (int mf = 0; mf < F; ++mf) acc[nf][mf] = mma(wf[nf], af[mf], acc[nf][mf]);
    }
  }

  // ---- epilogue: lane holds out[m][n..n+3], m = .. + (lane & 15), n = .. + 4 * (lane >> 4).
  // Loads are batched (all bias vectors once, the 4 RoPE vectors of a row together) so their latencies
  // overlap instead of serialising behind per-fragment branches.
  const bool rot = EPI == EPI_ROPE && n0 < e.rope_cols;  // tile-uniform: rope_cols is a multiple of the tile width
#pragma unroll
  for (int mf = 0; mf < F; ++mf) {
    const int m = m0 + wm * WT + mf * 16 + fr;
    if (m >= M) continue;
    f4v cs[F];
    if (rot) {
      const int s = m / e.L, l = m - s * e.L;
      const float* ropem = e.rope + ((size_t)(s % e.rope_batch) * e.L + l) * e.hd;  // hd/2 pairs x (cos, sin)
#pragma unroll
      for (int nf = 0; nf < F; ++nf) cs[nf] = *reinterpret_cast<const f4v*>(ropem + (n0 + wn * WT + nf * 16 + fg * 4) % e.hd);
    }
    T* dst = C + (size_t)m * N + n0 + wn * WT + fg * 4;
#pragma unroll
    for (int nf = 0; nf < F; ++nf) {
      f4v v = acc[nf][mf];
      if (EPI == EPI_GELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sizeof(T) == 2 ? v[j] : gelu_erf(v[j]);
        if (sizeof(T) == 2) v = gelu_erf_fast4(v);
      } else if (EPI == EPI_SILU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = silu(v[j]);
      } else if (EPI == EPI_ROPE) {
        if (rot) {  // pairs (n, n+1), (n+2, n+3); cs = cos0, sin0, cos1, sin1
          v = rope_rotate4(v, cs[nf]);
        }
        if (n0 < e.q_cols) v = v * e.q_scale;  // tile-uniform like `rot`
      }
      if constexpr (sizeof(T) == 2) {
        u2v o = {Half16<T>::pack(v[0], v[1]), Half16<T>::pack(v[2], v[3])};
        *reinterpret_cast<u2v*>(dst + nf * 16) = o;
      } else {
        *reinterpret_cast<f4v*>(dst + nf * 16) = v;
      }
    }
  }
}

// gemm256.hip: 256x256 ping-pong kernel for large M
int gemm256_launch(const void* A, const void* W, void* C, int M, int N, int K, int epi, const float* bias,
                   const float* rope, int L, int rope_batch, int hd, int rope_cols, float q_scale, int q_cols, int dtype,
                   hipStream_t st, int form = 0);  // 0 shipped choice, 1 one tile per workgroup, 2 persistent prologue form
int gemm256_cu_count();  // CUs of the device = the persistent kernel's grid

// 0 = auto, 128 / 256 = force that tile structure (tests compare the two structures bit for bit). Thread-local like
// the walk direction: a debugging knob of the calling host thread, not shared state.
static thread_local int g_force_tile = 0;
#ifdef NOVA_EXPERIMENTS
void gemm256_set_variant(int v);
void gemm256_set_gm(int g);
void gemm256_set_stagger(int cycles);
void gemm256_set_grid(int n);
void walk_set_alternate(int on);
#endif
int gemm_forced_tile() { return g_force_tile; }
int gemm_force_tile(int tile) {
#ifdef NOVA_EXPERIMENTS  // A/B knobs of tools/ (make exp): never compiled into the shipped library
  if (tile == 50000 || tile == 50001) {  // alternate the walk direction between launches of a block (off / on)
    walk_set_alternate(tile - 50000);
    return 0;
  }
  if (tile >= 40000 && tile <= 40512) {  // 40000 + n: persistent grid of n workgroups (0 = one per CU)
    gemm256_set_grid(tile - 40000);
    return 0;
  }
  if (tile >= 30000 && tile < 31000) {  // 30000 + x: start stagger of the persistent 256 kernel, x * 256 cycles
    gemm256_set_stagger((tile - 30000) * 256);
    return 0;
  }
  if (tile >= 7001 && tile <= 7064) {  // 7000 + g: row panels per tile group of the 256 kernel
    gemm256_set_gm(tile - 7000);
    return 0;
  }
  if (tile >= 2560 && tile <= 2580) {  // 2560 + v: force the 256 tile with schedule variant v (>= 10: timing-only, wrong results)
    gemm256_set_variant(tile - 2560);
    tile = 256;
  }
#endif
  if (tile == 161 || tile == 162 || tile == 164) {  // the small-M kernel with 16 / 32 / 64 rows per workgroup
    skinny_force_row_blocks(tile - 160);
    tile = 16;
  } else {
    skinny_force_row_blocks(0);
  }
  if (tile != 0 && tile != 16 && tile != 64 && tile != 128 && tile != 256 && tile != 257 && tile != 258) return -1;
  g_force_tile = tile;  // 257: the 256 tile in its one-tile-per-workgroup form; 258: its persistent prologue form (256: the shipped choice);
  return 0;             // 16: the small-M kernel of skinny.hip wherever its shapes allow (an error elsewhere)
}

// fewest 256 x 256 tiles the persistent kernel is given a launch for: half the CUs, or NOVA_GEMM_MIN_TILES (read once; 0 = no lower
// bound, the rule of rounds 1-2)
static long min_tiles256() {
  static const long v = [] {
    const char* e = getenv("NOVA_GEMM_MIN_TILES");
    return e && *e ? atol(e) : (long)gemm256_cu_count() / 2;
  }();
  return v;
}

template <typename T>
static int launch_gemm(const void* A, const void* W, void* C, int M, int N, int K, int epi, const GemmEpi& e,
                       hipStream_t st) {
  const int kelems = ROWB / (int)sizeof(T);
  if (M <= 0) return 0;
  if (N % 128 != 0 || K % kelems != 0 || K <= 0)
    return set_error(NOVA_ERR_SHAPE, "gemm: need N %% 128 == 0 and K %% %d == 0 (got M=%d N=%d K=%d)", kelems, M, N, K);
  const bool can256 = N % 256 == 0 && (epi != EPI_ROPE || (e.rope_cols % 256 == 0 && e.q_cols % 256 == 0));  // rotation is decided per tile
  if (g_force_tile >= 256 && !can256) return set_error(NOVA_ERR_SHAPE, "gemm: 256-tile kernel needs N %% 256 == 0");
  // The persistent 256 kernel needs tiles to spread over the chip: with fewer 256 x 256 tiles than half the CUs (batch 1: 20 row
  // panels x 4 column tiles for the out-projection and fc2) the 128 tile takes the launch. End to end at batch 1: 702 against
  // 722 ms per sample; a threshold of one tile per CU, which also moves the 160-tile launches of batch 2, loses 2.4 % there
  // (profiles/r03_gemm_min_tiles_ab.txt). Shapes of batch >= 4 have 320 tiles or more and are not touched.
  const long tiles256 = (long)((M + 255) / 256) * (N / 256);
  if (can256 && (g_force_tile >= 256 || (g_force_tile == 0 && M >= 4096 && tiles256 >= min_tiles256())))
    return gemm256_launch(A, W, C, M, N, K, epi, e.bias, e.rope, e.L, e.rope_batch, e.hd, e.rope_cols, e.q_scale, e.q_cols,
                          dtype_of<T>(), st, g_force_tile == 257 ? 1 : g_force_tile == 258 ? 2 : 0);
  constexpr int BM = 128, BN = 128;
  int ntm = (M + BM - 1) / BM, ntn = N / BN;
  ProfScope prof(PROF_GEMM_SMALL, 2.0 * M * N * K, st);
  const T* a = static_cast<const T*>(A);
  const T* w = static_cast<const T*>(W);
  T* c = static_cast<T*>(C);
  // fewer 128 x 128 tiles than CUs (the denoising loop at 1100 .. 4000 rows): 64 x 64 tiles - the launch is bound by what ONE workgroup
  // can pull through its CU, so four times the workgroups at a quarter of the operand bytes each (rotated tiles stay on the 128 tile:
  // the 64 tile's columns would split a head's table row differently only in cost, not in result, but it is not needed there).
  // NOVA_GEMM_TILE64=0 keeps the 128 tile (A/B: tools/gemm_tile64_ab.py)
  static const bool tile64_ok = [] { const char* v = getenv("NOVA_GEMM_TILE64"); return !(v && v[0] == '0'); }();
  // (3/4 of the CUs: at 208 tiles of 128 and more the 64 tile is 3-4 % behind, at 192 it is 1.1-1.2 x ahead, at 128 tiles 1.4 x:
  // profiles/r04_gemm_tile64_ab.txt, r04_gemm_small_m_tiles.txt)
  if (g_force_tile == 64 || (tile64_ok && g_force_tile == 0 && epi != EPI_ROPE && (long)ntm * ntn * 4 <= (long)gemm256_cu_count() * 3)) {
    if (epi == EPI_ROPE) return set_error(NOVA_ERR_SHAPE, "gemm: the 64 tile has no RoPE epilogue");
    ntm = (M + 63) / 64;
    ntn = N / 64;
    dim3 grid64(ntm * ntn), block(256);
    switch (epi) {
      case EPI_NONE: hipLaunchKernelGGL((gemm_kernel<T, EPI_NONE, 2>), grid64, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
      case EPI_GELU: hipLaunchKernelGGL((gemm_kernel<T, EPI_GELU, 2>), grid64, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
      case EPI_SILU: hipLaunchKernelGGL((gemm_kernel<T, EPI_SILU, 2>), grid64, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
      default: return set_error(NOVA_ERR_ARG, "gemm: unknown epilogue %d", epi);
    }
    return check_launch("gemm (64 tile)");
  }
  dim3 grid(ntm * ntn), block(256);
  switch (epi) {
    case EPI_NONE: hipLaunchKernelGGL((gemm_kernel<T, EPI_NONE>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    case EPI_GELU: hipLaunchKernelGGL((gemm_kernel<T, EPI_GELU>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    case EPI_SILU: hipLaunchKernelGGL((gemm_kernel<T, EPI_SILU>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    case EPI_ROPE: hipLaunchKernelGGL((gemm_kernel<T, EPI_ROPE>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    default: return set_error(NOVA_ERR_ARG, "gemm: unknown epilogue %d", epi);
  }
  return check_launch("gemm");
}

int gemm_bias_act(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int act,
                  int dtype, hipStream_t st) {
  GemmEpi e{bias, nullptr, 1, 1, 2, 0, 1.0f, 0};
  if (act < 0 || act > 2) return set_error(NOVA_ERR_ARG, "gemm: unknown activation %d", act);
  if (dtype_is16(dtype) && (g_force_tile == 16 || (g_force_tile == 0 && skinny_gemm_fits(M, N, K, false))))
    return skinny_gemm(A, W, bias, out, M, N, K, act, nullptr, dtype, st);
  return dispatch_dtype(dtype, [&](auto tag) { return launch_gemm<decltype(tag)>(A, W, out, M, N, K, act, e, st); });
}

// whether gemm_modulate_act takes this shape as ONE launch (the AdaLN modulate as the small-M kernel's prologue)
bool gemm_modulate_fused(int M, int N, int K, int dtype) {
  return dtype_is16(dtype) && (g_force_tile == 16 || (g_force_tile == 0 && skinny_gemm_fits(M, N, K, true)));
}

int gemm_modulate_act(const RowNormArgs& pro, const void* W, const float* bias, void* out, int M, int N, int K, int act,
                      int dtype, hipStream_t st) {
  if (gemm_modulate_fused(M, N, K, dtype)) return skinny_gemm(nullptr, W, bias, out, M, N, K, act, &pro, dtype, st);
  if (int rc = row_norm(pro, dtype, st)) return rc;
  return gemm_bias_act(pro.out, W, bias, out, M, N, K, act, dtype, st);
}

int gemm_qkv_rope(const void* x, const void* Wqkv, const float* bias, const float* rope, void* qkv, int S, int L,
                  int D, int heads, int rope_batch, int dtype, hipStream_t st, float q_scale) {
  const int hd = D / heads;
  if (heads <= 0 || D % heads != 0 || hd % 4 != 0) return set_error(NOVA_ERR_SHAPE, "qkv_rope: bad heads/D");
  const int epi = (rope || q_scale != 1.0f) ? EPI_ROPE : EPI_NONE;
  if (rope && rope_batch <= 0) return set_error(NOVA_ERR_ARG, "qkv_rope: rope_batch must be > 0");
  GemmEpi e{bias, rope, L, rope ? rope_batch : 1, hd, rope ? 2 * D : 0, q_scale, q_scale != 1.0f ? D : 0};
  return dispatch_dtype(dtype, [&](auto tag) { return launch_gemm<decltype(tag)>(x, Wqkv, qkv, S * L, 3 * D, D, epi, e, st); });
}

int gemm_rope_cols(const void* x, const void* W, const float* bias, const float* rope, void* out, int M, int N, int K,
                   int L, int rope_batch, int hd, int rope_cols, int dtype, hipStream_t st) {
  if (rope && rope_batch <= 0) return set_error(NOVA_ERR_ARG, "rope_cols: rope_batch must be > 0");
  GemmEpi e{bias, rope, L, rope ? rope_batch : 1, hd, rope_cols, 1.0f, 0};
  const int epi = rope ? EPI_ROPE : EPI_NONE;
  return dispatch_dtype(dtype, [&](auto tag) { return launch_gemm<decltype(tag)>(x, W, out, M, N, K, epi, e, st); });
}

}  // namespace nova
