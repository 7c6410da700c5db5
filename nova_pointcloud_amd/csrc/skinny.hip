// NOVA hot path: small-M projection GEMM for the diffusion MLP's per-step launches
//   out[M,N] = act(pro(A)[M,K] · W[N,K]^T + bias[N]),   M = (guidance passes x batch) x tokens predicted in this AR step
// (reference diffnext/models/diffusion_mlp.py:31-36,41-47: DiffusionBlock = AdaLN modulate -> fc1 -> SiLU -> fc2 -> gated
// norm; 25 denoising steps x 6 blocks per AR step, transformer_3d.py:102-113). At batch 8 these GEMMs have a few hundred
// rows: the 128x128-tile kernel then runs 12-48 workgroups through a 12-16 deep chain of barrier-separated K-tiles
// (11-14 us per launch) and the AdaLN modulate ahead of fc1 is a launch of its own (4-6 us for 256 rows).
//
// Structure: a workgroup owns 16 x RB rows (RB = 1, 2, 4 as M grows: a weight fragment then serves RB row blocks) x 64
// columns and the WHOLE K extent (K in {768, 1024}: d48w768 / d48w1024).
//   * the weight fragments of a wave's 16 columns go global -> registers directly, all K/32 of them requested before
//     anything else (each element is used by exactly one wave: staging it through LDS would buy nothing);
//   * the activation rows are written to LDS once, either copied or - PRO, 16 rows - produced by the AdaLN modulate
//     LN(x)(1 + scale) + shift itself (rownorm.h: the arithmetic of row_norm_kernel, one wave per row), so the
//     modulate launch disappears; every column tile recomputes its 16 rows, which is cheap next to a launch;
//   * one barrier, then K/32 MFMAs per wave and row block in K order on an accumulator that starts at the bias.
// Same MFMA (v_mfma_f32_16x16x32_bf16, weight fragment as A operand), same K order, same lane <-> k-slice placement,
// same epilogue functions as gemm_kernel / gemm256: results are BIT-IDENTICAL to the large-tile kernels, so the
// choice of kernel by M never shows in the output (batch and lane splits stay exact; tests compare bit for bit).
#include "common.h"
#include "nova_internal.h"
#include "rownorm.h"

namespace nova {

constexpr int SK_R = 16, SK_C = 64;

template <typename T, int K, int EPI, bool PRO, int RB>  // T: bf16_t / f16_t; RB: 16-row blocks per workgroup (1, 2 or 4; the prologue form is RB = 1 only)
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                          T* __restrict__ C, int M, int N, const float* __restrict__ bias,
                                                          RowNormArgs pro) {
  constexpr int NS = K / 32;            // MFMA steps
  constexpr int LROW = K * 2 + 16;      // LDS row pitch: +16 B so the 16 rows of a fragment read start in different banks
  constexpr int NIT = (K / 8 + 63) / 64;
  constexpr int R = SK_R * RB;          // rows per workgroup
  __shared__ __attribute__((aligned(16))) char smem[R * LROW];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int ntn = N / SK_C;
  const int tn = blockIdx.x % ntn, tm = blockIdx.x / ntn;
  const int m0 = tm * R, n0 = tn * SK_C + wid * 16;

  // ---- weights: lane (fr, fg) holds W[n0 + fr][32 s + 8 fg .. + 8] for every step s
  u4v wf[NS];
  {
    const T* wp = W + (size_t)(n0 + fr) * K + 8 * fg;
#pragma unroll
    for (int s = 0; s < NS; ++s) wf[s] = *reinterpret_cast<const u4v*>(wp + 32 * s);
  }
  f4v acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb] = bias ? *reinterpret_cast<const f4v*>(bias + n0 + 4 * fg) : f4v{0.f, 0.f, 0.f, 0.f};

  // ---- activation rows -> LDS; wave w owns rows 4w .. 4w+3 (rows past M repeat row M-1 and are never stored); the
  // loads of all four rows are requested before the first row's reductions
  if (PRO) {
    RowRegs<T, NIT, false, true> g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) row_norm_load<T, NIT, false, true>(pro, min(m0 + wid * 4 + i, M - 1), lane, g[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Chunk<T> y[NIT];
      row_norm_finish<T, NIT, false, true>(pro, lane, g[i], y);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int d = (it * 64 + lane) * 8;
        if (d < K) {
          const u4v u = {Half16<T>::pack(y[it].v[0][0], y[it].v[0][1]), Half16<T>::pack(y[it].v[0][2], y[it].v[0][3]),
                         Half16<T>::pack(y[it].v[1][0], y[it].v[1][1]), Half16<T>::pack(y[it].v[1][2], y[it].v[1][3])};
          *reinterpret_cast<u4v*>(smem + (wid * 4 + i) * LROW + d * 2) = u;
        }
      }
    }
  } else {  // plain copy: wave w owns rows 4 RB w .. 4 RB (w + 1) - 1, four rows' loads in flight at a time
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      u4v raw[4][NIT];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int d = (it * 64 + lane) * 8;
          if (d < K) raw[i][it] = *reinterpret_cast<const u4v*>(A + (size_t)min(m0 + (wid * RB + b) * 4 + i, M - 1) * K + d);
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int d = (it * 64 + lane) * 8;
          if (d < K) *reinterpret_cast<u4v*>(smem + ((wid * RB + b) * 4 + i) * LROW + d * 2) = raw[i][it];
        }
    }
  }
  __syncthreads();

  // ---- K/32 MFMAs per row block, K order; a weight fragment serves all RB row blocks
  const char* ap = smem + fr * LROW + 16 * fg;
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const u4v af = *reinterpret_cast<const u4v*>(ap + rb * 16 * LROW + 64 * s);
      acc[rb] = Half16<T>::mfma16(wf[s], af, acc[rb]);
    }

  // ---- epilogue: lane holds out[m0 + 16 rb + fr][n0 + 4 fg .. + 4]
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int m = m0 + rb * 16 + fr;
    if (m < M) {
      f4v v = acc[rb];
      if (EPI == 1) {
        v = gelu_erf_fast4(v);
      } else if (EPI == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = silu(v[j]);
      }
      const u2v o = {Half16<T>::pack(v[0], v[1]), Half16<T>::pack(v[2], v[3])};
      *reinterpret_cast<u2v*>(C + (size_t)m * N + n0 + 4 * fg) = o;
    }
  }
}

template <typename T, int K, bool PRO, int RB>
static void launch_skinny(const T* A, const T* W, T* C, int M, int N, const float* bias, int act,
                          const RowNormArgs& pro, hipStream_t st) {
  const dim3 grid((unsigned)(((M + SK_R * RB - 1) / (SK_R * RB)) * (N / SK_C))), block(256);
  switch (act) {
    case 0: hipLaunchKernelGGL((skinny_gemm_kernel<T, K, 0, PRO, RB>), grid, block, 0, st, A, W, C, M, N, bias, pro); break;
    case 1: hipLaunchKernelGGL((skinny_gemm_kernel<T, K, 1, PRO, RB>), grid, block, 0, st, A, W, C, M, N, bias, pro); break;
    default: hipLaunchKernelGGL((skinny_gemm_kernel<T, K, 2, PRO, RB>), grid, block, 0, st, A, W, C, M, N, bias, pro); break;
  }
}

template <typename T, int K>
static void launch_skinny_k(const T* a, const T* w, T* c, int M, int N, const float* bias, int act,
                            const RowNormArgs* pro, int rb, hipStream_t st) {
  const RowNormArgs none{};
  if (pro) launch_skinny<T, K, true, 1>(a, w, c, M, N, bias, act, *pro, st);
  else if (rb == 1) launch_skinny<T, K, false, 1>(a, w, c, M, N, bias, act, none, st);
  else if (rb == 2) launch_skinny<T, K, false, 2>(a, w, c, M, N, bias, act, none, st);
  else launch_skinny<T, K, false, 4>(a, w, c, M, N, bias, act, none, st);
}

// Shapes this kernel is built for and worth using on. Every row tile re-reads the whole weight matrix and every column
// tile its activation rows, all from L2, so the right rows-per-workgroup grows with M (a weight fragment then serves 1, 2
// or 4 row blocks) until the 128-tile kernel's LDS-shared operands win. Kernel durations under rocprofv3, N = K = D
// (tools/skinny_bench.py; us at D = 768 | 1024, 128-tile kernel 11-12 | 13-15 throughout):
//   M <= 256: 16 rows 5.7-5.9 | 6.3-7.3      M 400-512: 32 rows 7.0-7.2 | 8.8      M 800-1024: 64 rows 9.2 | 12.2
//   M >= 1600: none (14-37 | 21-46).   [round 4: superseded for the plain GEMM by the rule in skinny_row_blocks]   Modulate + fc1 in one launch (16 rows only): 9.4-9.7 | 10.4-11.7 up to M = 256
//   against 14.4-15.1 | 18.1-18.3 for row_norm + GEMM; from M = 400 on the two launches are faster.
// skinny_row_blocks returns 1, 2, 4 (x 16 rows) or 0 = use the tile kernels.
static thread_local int g_skinny_rb = 0;  // tools / tests: force a row-block count (0 = by the rule above)
void skinny_force_row_blocks(int rb) { g_skinny_rb = rb; }
int skinny_forced_row_blocks() { return g_skinny_rb; }

int gemm256_cu_count();  // gemm256.hip: CUs of the device

int skinny_row_blocks(int M, int N, int K, bool modulate) {
  if (M <= 0 || (K != 768 && K != 1024) || N % SK_C != 0) return 0;
  if (g_skinny_rb == 1 || g_skinny_rb == 2 || g_skinny_rb == 4) return modulate ? 1 : g_skinny_rb;
  const bool tiles_can = N % 128 == 0;  // the 64 x 64 tile kernel of gemm.hip takes the shape
  if (modulate || !tiles_can) {
    const int rb = M <= 320 ? 1 : (modulate ? 0 : M <= 640 ? 2 : M <= 1100 ? 4 : 0);
    if (rb == 0) return 0;
    // wide outputs (N >> K: the AdaLN projection, fc1 of a ViT block): the weight re-reads of all row tiles must stay a few
    // tens of MB of L2 traffic, or the tile kernels' once-per-tile-row weight reads win
    const double weight_bytes = (double)((M + SK_R * rb - 1) / (SK_R * rb)) * N * K * 2.0;
    return weight_bytes <= 64.0e6 ? rb : 0;
  }
  // Plain GEMM, round 4: the 64 x 64 tile kernel is level with the 16-row form while this kernel's workgroups all fit the chip at once
  // and ahead of it beyond (N = K = 1024: 9.3 against 10.8 us at M = 512, 10.3 against 13.7 at 1024; N = 3072: 9.4 against 12.2 us at
  // M = 96, where 288 workgroups need a second round: profiles/r04_gemm_small_m_tiles.txt), so the 32- and 64-row forms are only
  // reached through the force codes now.
  const long wgs = (long)((M + SK_R - 1) / SK_R) * (N / SK_C);
  return wgs <= gemm256_cu_count() ? 1 : 0;
}

bool skinny_gemm_fits(int M, int N, int K, bool modulate) { return skinny_row_blocks(M, N, K, modulate) != 0; }

// `pro` null: plain GEMM on A. Otherwise pro->in / mod / scale_off / shift_off / eps describe the AdaLN modulate whose
// result is the A operand (pro->out, gamma, res, gate and gather are not used: the m1 form of diffusion_mlp.py:41-43).
int skinny_gemm(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int act,
                const RowNormArgs* pro, int dtype, hipStream_t st) {
  if (M <= 0) return 0;
  if (!dtype_is16(dtype)) return set_error(NOVA_ERR_ARG, "skinny_gemm: 16-bit storage types only");
  if ((K != 768 && K != 1024) || N % SK_C != 0) return set_error(NOVA_ERR_SHAPE, "skinny_gemm: need K in {768, 1024} and N %% 64 == 0 (got N=%d K=%d)", N, K);
  if (act < 0 || act > 2) return set_error(NOVA_ERR_ARG, "skinny_gemm: unknown activation %d", act);
  if (pro && (pro->D != K || pro->rows < M || !pro->in || !pro->mod || pro->scale_off < 0 || pro->shift_off < 0 || pro->gate_off >= 0 ||
              pro->gamma || pro->res || pro->gather || pro->mod_ld % 8 || pro->scale_off % 8 || pro->shift_off % 8))
    return set_error(NOVA_ERR_ARG, "skinny_gemm: the prologue is the scale/shift modulate of %d-wide rows only", K);
  ProfScope prof(PROF_GEMM_SMALL, 2.0 * M * N * K, st);
  int rb = pro ? 1 : skinny_row_blocks(M, N, K, false);
  if (rb == 0) rb = 1;  // forced onto a shape the traffic rule would not pick (nova_debug_force_gemm_tile(16))
  dispatch_half(dtype, [&](auto tag) {
    using T = decltype(tag);
    const T* a = static_cast<const T*>(A);
    const T* w = static_cast<const T*>(W);
    T* c = static_cast<T*>(out);
    if (K == 768) launch_skinny_k<T, 768>(a, w, c, M, N, bias, act, pro, rb, st);
    else launch_skinny_k<T, 1024>(a, w, c, M, N, bias, act, pro, rb, st);
    return 0;
  });
  return check_launch("skinny_gemm");
}

}  // namespace nova
