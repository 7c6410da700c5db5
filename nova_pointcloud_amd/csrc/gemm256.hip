// NOVA hot path: large-M projection GEMM, 256x256 tile, ping-pong schedule.
//
// Same contract and epilogues as gemm.hip (out[M,N] = epi(A[M,K] W[N,K]^T + bias)); used for the encoder
// GEMMs where M = S*L is in the tens of thousands and N is a multiple of 256.
//
// Why a second structure: the 128x128 kernel spends most of a K-step waiting on "LDS-DMA landed -> barrier"
// (its MFMA pipe is ~35 % busy). Here one 512-thread workgroup per CU owns a 256x256 tile; its 8 waves form
// two groups (wr = 0/1, one wave of each group on every SIMD) that run ONE BARRIER APART: while a group issues
// its 16 MFMAs of a phase, the other group issues its LDS fragment reads and the next LDS-DMA prefetch, then they
// swap. LDS-DMA stays in flight across barriers under a counted s_waitcnt vmcnt (never 0 in the loop).
//
// Operand roles (round 4). The tile is computed as P[x][y] = sum_k X[x][k] Y[y][k] with X = 256 rows of W (output COLUMNS,
// split over the two wave groups wr) and Y = 256 rows of A (output ROWS, split over wc = 0..3); Y fragments are the MFMA's
// first operand, so a lane (j = lane & 15, fg = lane >> 4) holds, per accumulator, output rows 4 fg + {0,1,2,3} of ONE
// column j. Which W row sits at fragment row j is free (it is only the LDS-DMA source address), two maps are used:
//   COLS8 (every epilogue but RoPE): X fragment f of half xi takes W rows 8 j + 4 xi + f - a lane's eight X fragments are 8
//     CONSECUTIVE columns, one 16-byte store per output row (16-bit types), 4 rows x 256 contiguous bytes per store instruction;
//   COLS4 (RoPE): X fragment f of half xi takes W rows 64 xi + 4 j + f - 4 consecutive columns per half, so the 16 lanes j read
//     one table row's 64 floats as 256 contiguous bytes, ONE 16-byte table load per output row serves both halves when the
//     head width divides 64, and stores are 8 bytes per lane (4 rows x 128 contiguous bytes per instruction). Rounds 1-3 had the roles the other way round (a lane = one
// row, 4 columns per fragment): adjacent lanes then sat in different rows, the memory pipeline took such a store or load
// one LANE at a time (tools/ta_probe.hip: 67 against 31 cycles per 16-byte store instruction, 22 per 8-byte one), and the epilogue's 16 stores + 32 table
// loads per lane were most of the K = 1024 tiles' seam. Same MFMA, same k order: results are bit-identical to gemm.hip.
//
// Phase plan. A K-tile (64 bf16 / 32 f32 per row) is staged as 4 units of 128 rows x 128 B:
//   X(xi): W rows {wr*128 + xi*64 + 4 j + f}  as unit row wr*64 + f*16 + j, for both wr       (read in phase 0 / 2)
//   Y(yi): A rows {wc*64  + yi*32 + [0,32)}  as unit row wc*32 + [0,32), for all four wc     (read in phase 0 / 1)
// A wave (wr, wc) owns out[wc*64 + 64 rows][wr*128 + 128 columns] = 4 (Y) x 8 (X) MFMA 16x16 fragments. Per K-tile, 4 phases:
//   p0: read X(0), Y(0) -> quadrant (0,0)   p1: read Y(1) -> (0,1)   p2: read X(1) -> (1,1)
//   p3: no LDS reads, quadrant (1,0) on the Y(0) fragments kept in registers since p0
// 16 MFMAs each (8 independent accumulators per k-half). Staging order is X0, Y1, X1, Y0 per tile (flat index
// q = 4*tile + unit); phase p of tile t issues q = 4t + p + 6, i.e. always the unit whose LDS slot (2 K-tile
// buffers x 4 units = 128 KiB) had its last read at least TWO phases earlier - the distance that is safe for
// groups running one barrier apart with the fragment-read wait placed after the phase's first barrier.
// One wait per K-tile: a counted s_waitcnt vmcnt before the first barrier of phase 3 retires everything but the
// youngest unit(s) (= all of tile t+1); the first read of tile t+1 happens in the next phase, one barrier later.
// Epilogue: bias (the accumulators start at it), GELU / SiLU / RoPE pair rotation / q-scale lane-local on the lane's 4
// consecutive columns, as in gemm.hip. Results are bit-identical to gemm.hip: tests/test_gpu_kernels.py compares them.
#include <type_traits>
#ifndef NOVA_GEMM_ACT_ROWS
#define NOVA_GEMM_ACT_ROWS 1  // 0: all activations of a tile, then all its stores (A/B build)
#endif
#include "common.h"
#include "nova_internal.h"

namespace nova {

constexpr int P_UNIT = 128 * 128;   // bytes of one staged unit (128 rows x 128 B)
constexpr int P_BUF = 4 * P_UNIT;   // X(0) X(1) Y(0) Y(1)
constexpr int P_KLDS = 2 * P_BUF;   // 128 KiB: two K-tile buffers
constexpr int P_BIAS = P_KLDS;      // persistent form: the tile's 256 bias values (1 KiB), brought in by LDS-DMA like the operands
constexpr int P_LDS = P_KLDS + 1024;

struct GemmEpi;  // same POD as gemm.hip (redeclared below to keep the translation units independent)
struct GemmEpi256 {
  const float* bias;
  const float* rope;
  int L, rope_batch, hd, rope_cols;
  float q_scale;
  int q_cols;
  int gm;  // row panels per tile group (L2 reuse shape)
  int rev;      // persistent form: walk each XCD's chunk of the tile list back to front (xcd_remap_dir, common.h)
  const float* sa;  // fp8 path: per-row dequantisation scale of A [M]
  const float* sw;  // fp8 path: per-row (output column) dequantisation scale of W [N]
  int stagger;  // persistent form: start delay of the last workgroup in cycles (0 = none), see gemm256p_kernel
  int sa_scalar = 0;            // fp8 path: sa points at ONE scale shared by all rows of A (a tensor quantised with a static scale)
  const float* q8_scale = nullptr;  // E_GELU_Q8: the scale the e4m3 output is quantised with (one float, read at kernel entry)
  unsigned* q8_amax = nullptr;      // E_GELU_Q8: running max |value| of the un-quantised outputs, as float bits (atomic max)
};

// E_GELU_Q8 (fp8 operands only): GELU, then the result is written as OCP e4m3 bytes, value / *q8_scale saturated at +-448,
// instead of bf16 - the A operand of the next fp8 GEMM without a quantisation pass ("delayed scaling": the caller derives
// the next call's scale from q8_amax).
enum { E_NONE = 0, E_GELU = 1, E_SILU = 2, E_ROPE = 3, E_GELU_Q8 = 5 };

template <typename T> struct PFrag;
template <> struct PFrag<bf16_t> { u4v v; };
template <> struct PFrag<f16_t> { u4v v; };
template <> struct PFrag<float> { f4v v; };
typedef uint8_t fp8_t;  // OCP e4m3 storage (MX-fp8 path: 128 elements per 128-byte K-tile row)
template <> struct PFrag<fp8_t> { u4v v; };
typedef __attribute__((ext_vector_type(8))) int i8v;
template <typename T> struct OutOf { typedef T type; };
template <> struct OutOf<fp8_t> { typedef bf16_t type; };  // fp8 operands produce bf16 results

// first operand: the Y (output row) fragment, second: the X (output column) fragment
__device__ __forceinline__ f4v pmma(const PFrag<bf16_t>& y, const PFrag<bf16_t>& x, f4v c) { return Half16<bf16_t>::mfma16(y.v, x.v, c); }
__device__ __forceinline__ f4v pmma(const PFrag<f16_t>& y, const PFrag<f16_t>& x, f4v c) { return Half16<f16_t>::mfma16(y.v, x.v, c); }
__device__ __forceinline__ f4v pmma(const PFrag<float>& y, const PFrag<float>& x, f4v c) {
#pragma unroll
  for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(y.v[j], x.v[j], c, 0, 0, 0);
  return c;
}

template <typename T>
__device__ __forceinline__ PFrag<T> punit_frag(const char* unit, int row, int chunk) {
  PFrag<T> f;
  f.v = *reinterpret_cast<const decltype(f.v)*>(unit + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
  return f;
}

// Per-lane byte offsets (from the tile's W / A base) of the two LDS-DMA pieces a wave moves for each staged unit, in staging
// order u: 0 = X(0), 1 = Y(1), 2 = X(1), 3 = Y(0). Piece i of wave wid fills unit rows (wid*2 + i)*8 + [0, 8), 8 lanes (128 B) per
// row, the 16-byte chunks in source-swizzled order (punit_frag reads with the same xor). `rmax` = last valid A row of the tile
// (rows past M re-read row M-1; their results are stored onto row M-1 again, identical bytes).
template <bool COLS8>
__device__ __forceinline__ void dma_offsets(int lane, int wid, int rmax, uint32_t rowbytes, uint32_t (&soff)[4][2]) {
  const int cp = lane & 7;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wid * 2 + i) * 8 + (lane >> 3);              // row inside the unit, 0..127
    const uint32_t c = (uint32_t)((cp ^ ((r >> 1) & 7)) << 4);  // source chunk (swizzle on the source side)
    // unit row (wr, f, j) -> W row wr*128 + 8 j + f (+ 4 xi)   or   wr*128 + 4 j + f (+ 64 xi)
    const int x_lo = (r >> 6) * 128 + (r & 15) * (COLS8 ? 8 : 4) + ((r >> 4) & 3);
    const int y_lo = (r >> 5) * 64 + (r & 31);  // unit row (wc, i) -> A row wc*64 + i (+ yi*32)
    soff[0][i] = (uint32_t)x_lo * rowbytes + c;
    soff[2][i] = (uint32_t)(x_lo + (COLS8 ? 4 : 64)) * rowbytes + c;
    soff[3][i] = (uint32_t)min(y_lo, rmax) * rowbytes + c;
    soff[1][i] = (uint32_t)min(y_lo + 32, rmax) * rowbytes + c;
  }
}

// first column of X fragment (xi, f = 0) of a lane (fr = lane & 15) inside the wave's 128 columns
template <bool COLS8> __device__ __forceinline__ int col_of(int fr, int xi) { return COLS8 ? fr * 8 + xi * 4 : fr * 4 + xi * 64; }

#define NOVA_BARRIER() asm volatile("s_barrier" ::: "memory")
#define NOVA_LOOP_BARRIER() do { if (VAR < 12) NOVA_BARRIER(); } while (0)

// The lane-local part of every epilogue: the 4 consecutive output columns a lane holds of one output row.
// ROT: pair rotation with the row's table entry t = (cos0, sin0, cos1, sin1) (the two pairs of these 4 columns).
template <typename OT, int EPI, bool ROT>
__device__ __forceinline__ f4v epi_apply(f4v v, f4v t, float qmul) {
  if (EPI == E_GELU) {
    if (sizeof(OT) == 2) {
      v = gelu_erf_fast4(v);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = gelu_erf(v[q]);
    }
  } else if (EPI == E_SILU) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = silu(v[q]);
  } else if (EPI == E_ROPE) {
    if (ROT) v = rope_rotate4(v, t);
    v = v * qmul;  // tile-uniform; x * 1.0f is exact
  }
  return v;
}
template <typename OT>
__device__ __forceinline__ void store4(OT* dst, f4v v) {  // 4 consecutive columns of one row
  if constexpr (sizeof(OT) == 2) {
    *reinterpret_cast<u2v*>(dst) = u2v{Half16<OT>::pack(v[0], v[1]), Half16<OT>::pack(v[2], v[3])};
  } else {
    *reinterpret_cast<f4v*>(dst) = v;
  }
}
template <typename OT>
__device__ __forceinline__ void store8(OT* dst, f4v lo, f4v hi) {  // 8 consecutive columns of one row
  if constexpr (sizeof(OT) == 2) {
    const u4v pk = u4v{Half16<OT>::pack(lo[0], lo[1]), Half16<OT>::pack(lo[2], lo[3]), Half16<OT>::pack(hi[0], hi[1]), Half16<OT>::pack(hi[2], hi[3])};
    // Streaming (`nt`) stores: 1.4-2.3 % on the K = 1024 launches, level at K = 4096, and the LayerNorm pass that reads the result next is
    // not slower (profiles/r04_gemm_nt_stores_ab.txt; -DNOVA_PLAIN_STORES: the A/B build). The counters do not show why: L2 hit rate and
    // fabric fetch per launch are the same with either policy (profiles/r04_pmc_table.md) - the gain is on the write path.
#ifdef NOVA_PLAIN_STORES
    *reinterpret_cast<u4v*>(dst) = pk;
#else
    __builtin_nontemporal_store(pk, reinterpret_cast<u4v*>(dst));
#endif
  } else {
    *reinterpret_cast<f4v*>(dst) = lo;
    *reinterpret_cast<f4v*>(dst + 4) = hi;
  }
}

// VAR selects where the two LDS-DMA instructions of a phase are issued (A/B-tested on the GPU, tools/microbench.py):
//   0: both between the two k-halves of the MFMA segment   1: one in the load segment, one in the MFMA segment
//   2: both in the load segment after the fragment reads   3: both in the load segment before the fragment reads
// The shipped library instantiates VAR 2 only (the fallback of the persistent kernel). VAR 0/1/3 and the timing-only
// ablations VAR 10-12 (no prefetch / no fragment reads / no barriers: WRONG results by design) exist only in the
// -DNOVA_EXPERIMENTS build that tools/ loads (make -C nova_pointcloud_amd/csrc exp).
template <typename T, int EPI, int VAR>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                         T* __restrict__ C, int M, int N, int K, int ntm, int ntn,
                                                         GemmEpi256 e) {
  __shared__ __attribute__((aligned(16))) char smem[P_KLDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;

  const int nwg = ntm * ntn;
  const int t = xcd_remap(blockIdx.x, nwg);
  const int GM = e.gm;
  const int per_group = GM * ntn;
  const int group = t / per_group, first_m = group * GM;
  const int gsz = min(ntm - first_m, GM);
  const int tm = first_m + (t % per_group) % gsz;
  const int tn = (t % per_group) / gsz;
  const int m0 = tm * 256, n0 = tn * 256;

  // wave-uniform tile bases + per-lane 32-bit byte offsets: the LDS-DMA instructions then take an SGPR base and a
  // 32-bit VGPR offset, and advancing along K is scalar arithmetic (no 64-bit vector add per issue)
  const size_t rowbytes = (size_t)K * sizeof(T);
  const char* a_base = reinterpret_cast<const char*>(A) + (size_t)m0 * rowbytes;
  const char* w_base = reinterpret_cast<const char*>(W) + (size_t)n0 * rowbytes;
  constexpr bool COLS8 = EPI != E_ROPE;  // column map of the X fragments (header comment)
  uint32_t soff[4][2];
  dma_offsets<COLS8>(lane, wid, M - 1 - m0, (uint32_t)rowbytes, soff);
  const int nkt = K / (128 / (int)sizeof(T));
  // LDS offset of unit u inside a buffer: X(0) X(1) Y(0) Y(1)
  auto unit_off = [](int u) { return (u == 0 ? 0 : u == 2 ? 1 : u == 3 ? 2 : 3) * P_UNIT; };
  auto stage_piece = [&](int u, int kt, int i) {
    if (kt < nkt) {
      char* dst = smem + (kt & 1) * P_BUF + unit_off(u) + wid * 2048 + i * 1024;
      const char* base = ((u == 0 || u == 2) ? w_base : a_base) + (size_t)kt * 128;
      __builtin_amdgcn_global_load_lds(base + soff[u][i], NOVA_LDS_PTR(dst), 16, 0, 0);
    }
  };
  auto stage = [&](int u, int kt) {  // all 8 waves: 2 LDS-DMA instructions each (wave-uniform condition)
    if (VAR >= 10 && kt >= 2) return;  // ablation builds (timing only, wrong results): no prefetch inside the loop
    stage_piece(u, kt, 0);
    stage_piece(u, kt, 1);
  };

  const int fr = lane & 15, fg = lane >> 4;
  // the accumulators start at the bias (see gemm.hip): every accumulator of X fragment (xi, f) holds column col_of(fr, xi) + f
  f4v bcol[2] = {f4v{0.f, 0.f, 0.f, 0.f}, f4v{0.f, 0.f, 0.f, 0.f}};
  if (e.bias) {
#pragma unroll
    for (int xi = 0; xi < 2; ++xi) bcol[xi] = *reinterpret_cast<const f4v*>(e.bias + n0 + wr * 128 + col_of<COLS8>(fr, xi));
  }
  f4v acc[4][8];  // [y fragment][x fragment]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float b = bcol[j >> 2][j & 3];
      acc[i][j] = f4v{b, b, b, b};
    }

  // ---- prologue: q = 0..5 = tile 0 (X0 Y1 X1 Y0) + tile 1 (X0 Y1); wait for tile 0
  stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
  stage(0, 1); stage(1, 1);
  if (nkt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  NOVA_BARRIER();
  if (wr == 1) NOVA_BARRIER();  // group 1 runs one barrier behind group 0 from here on

  PFrag<T> xf[4][2], yf0[2][2], yf1[2][2];
  auto read_x = [&](const char* buf, int xi) {
    if (VAR >= 11 && buf != smem) return;
    const char* u = buf + xi * P_UNIT;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) xf[f][kk] = punit_frag<T>(u, wr * 64 + f * 16 + fr, fg + 4 * kk);
  };
  auto read_y = [&](const char* buf, int yi, PFrag<T> (&yf)[2][2]) {
    if (VAR >= 11 && buf != smem) return;
    const char* u = buf + (2 + yi) * P_UNIT;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) yf[f][kk] = punit_frag<T>(u, wc * 32 + f * 16 + fr, fg + 4 * kk);
  };
  // 16 MFMAs of one quadrant; the LDS-DMA prefetch of this phase is issued between the two k halves, where
  // its issue cost hides under the matrix pipe instead of lengthening the other group's critical load segment.
  auto mma_quadrant = [&](int xi, int yi, PFrag<T> (&yf)[2][2], int su, int skt) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int x = 0; x < 4; ++x) acc[yi * 2 + f][xi * 4 + x] = pmma(yf[f][kk], xf[x][kk], acc[yi * 2 + f][xi * 4 + x]);
      if (kk == 0) {
        if (VAR == 0) stage(su, skt);
        if (VAR == 1) stage_piece(su, skt, 1);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  auto lstage_pre = [&](int u, int kt) { if (VAR == 3) stage(u, kt); };
  auto lstage_post = [&](int u, int kt) {
    if (VAR == 2 || VAR >= 10) stage(u, kt);
    if (VAR == 1) stage_piece(u, kt, 0);
  };
  for (int kt = 0; kt < nkt; ++kt) {
    const char* buf = smem + (kt & 1) * P_BUF;
    // phase 0: quadrant (0,0); stages q = 4kt+6 = (kt+1, X1)
    lstage_pre(2, kt + 1);
    read_x(buf, 0);
    read_y(buf, 0, yf0);
    lstage_post(2, kt + 1);
    NOVA_LOOP_BARRIER();
    mma_quadrant(0, 0, yf0, 2, kt + 1);
    NOVA_LOOP_BARRIER();
    // phase 1: quadrant (0,1); stages (kt+1, Y0)
    lstage_pre(3, kt + 1);
    read_y(buf, 1, yf1);
    lstage_post(3, kt + 1);
    NOVA_LOOP_BARRIER();
    mma_quadrant(0, 1, yf1, 3, kt + 1);
    NOVA_LOOP_BARRIER();
    // phase 2: quadrant (1,1); stages (kt+2, X0)
    lstage_pre(0, kt + 2);
    read_x(buf, 1);
    lstage_post(0, kt + 2);
    NOVA_LOOP_BARRIER();
    mma_quadrant(1, 1, yf1, 0, kt + 2);
    NOVA_LOOP_BARRIER();
    // phase 3: quadrant (1,0) on the Y(0) fragments still held from phase 0 (no LDS reads); stages (kt+2, Y1).
    // Before the first barrier: retire all of tile kt+1 = everything but the youngest unit issued so far
    // ((kt+2, X0) of phase 2; this phase's unit is issued after the barrier).
    // (VAR 1: one more piece of this phase's unit is already issued -> 3; VAR 2/3: the whole unit -> 4)
    lstage_pre(1, kt + 2);
    lstage_post(1, kt + 2);
    if (kt + 2 < nkt) {
      if (VAR == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      if (VAR == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      if (VAR >= 2 && VAR < 10) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      if (VAR >= 10) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    NOVA_LOOP_BARRIER();
    mma_quadrant(1, 0, yf0, 1, kt + 2);
    NOVA_LOOP_BARRIER();
  }
  if (wr == 0) NOVA_BARRIER();  // re-align the groups

  // ---- epilogue (lane-local): the lane holds out[rows 4 fg + r of Y fragment y][columns col_of(fr, xi) + {0..3}]
  const bool rot = EPI == E_ROPE && n0 < e.rope_cols;
  const float qmul = (EPI == E_ROPE && n0 < e.q_cols) ? e.q_scale : 1.0f;
  const int nw = n0 + wr * 128;
#pragma unroll
  for (int y = 0; y < 4; ++y) {
    const int mb = m0 + wc * 64 + y * 16;
    if (__builtin_amdgcn_readfirstlane(mb) >= M) continue;  // whole fragment row block past M (wave-uniform)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = min(mb + fg * 4 + r, M - 1);  // lanes past M recompute row M-1 and store the identical bytes (benign)
      const float* ropem = nullptr;
      if (rot) {
        const int s = m / e.L, l = m - s * e.L;
        ropem = e.rope + ((size_t)(s % e.rope_batch) * e.L + l) * e.hd;
      }
      f4v v[2];
#pragma unroll
      for (int xi = 0; xi < 2; ++xi) {
        v[xi] = f4v{acc[y][xi * 4][r], acc[y][xi * 4 + 1][r], acc[y][xi * 4 + 2][r], acc[y][xi * 4 + 3][r]};
        if (rot) v[xi] = epi_apply<T, EPI, true>(v[xi], *reinterpret_cast<const f4v*>(ropem + (nw + col_of<COLS8>(fr, xi)) % e.hd), qmul);
        else v[xi] = epi_apply<T, EPI, false>(v[xi], v[xi], qmul);
      }
      if constexpr (COLS8) {
        store8<T>(C + (size_t)m * N + nw + fr * 8, v[0], v[1]);
      } else {
        store4<T>(C + (size_t)m * N + nw + col_of<false>(fr, 0), v[0]);
        store4<T>(C + (size_t)m * N + nw + col_of<false>(fr, 1), v[1]);
      }
    }
  }
}

#ifdef NOVA_STAMPS
// Diagnostic build only (tools/gemm_stamps.py; never in the shipped library): workgroup 0 records the shader clock at eight points
// of every tile, for one wave of each group (waves 0 and 4), into LDS (an LDS store: nothing is added to the vector-memory queue
// whose counts the kernel's waits rely on) and copies the record to g_gemm_stamps when it exits.
constexpr int STAMP_TILES = 64, STAMP_PTS = 8;
__device__ uint32_t g_gemm_stamps[2 * STAMP_TILES * STAMP_PTS];
#define NOVA_STAMP(k)                                                                                                            \
  do {                                                                                                                           \
    if (blockIdx.x == 0 && (wid & 3) == 0 && lane == 0 && stamp_tile < STAMP_TILES)                                               \
      reinterpret_cast<uint32_t*>(smem + P_LDS)[((wid >> 2) * STAMP_TILES + stamp_tile) * STAMP_PTS + (k)] = (uint32_t)__builtin_amdgcn_s_memtime(); \
  } while (0)
constexpr int P_STAMPS = 2 * STAMP_TILES * STAMP_PTS * 4;
#else
#define NOVA_STAMP(k) do {} while (0)
constexpr int P_STAMPS = 0;
#endif

#ifdef NOVA_CLOCK
// Diagnostic build only (tools/kernel_clock.py): every workgroup of the continuous form reads the shader clock (s_memtime) and the
// constant 100 MHz clock (s_memrealtime) once when it starts and once when it ends: the quotient is the clock the chip held
// under this kernel's load (MI355X_MICROARCH.md, 'DVFS give-back' item 6). No stamp inside the loop.
__device__ long long g_gemm_clock[2 * 1024];
#endif

// ------------------------------------------------------------------------------------------
// Persistent form: one workgroup per CU walks its XCD's chunk of the tile list (same tile order as the
// one-tile-per-workgroup launch above). What it buys on the K = 1024 shapes, where a tile's main loop is only
// 16 K-tiles long: no workgroup dispatch between tiles, and the first six LDS-DMA units of the NEXT tile are
// issued at the start of the epilogue (after the re-align barrier every LDS slot is dead), so their HBM/L2
// latency hides under the epilogue's VALU work and stores. vmcnt retires in issue order and counts stores, so the
// wait at the top of the next tile is vmcnt(4 + stores of the epilogue).
// Nothing the compiler has to wait for may be requested while those DMAs are in flight: beside pending LDS-DMA it waits
// with vmcnt(0) at the first use of an ordinary load (the whole queue, stores included). So the RoPE table rows are
// loaded and consumed BEFORE the DMAs are issued, and the next tile's bias takes the operands' own route: one 1-KiB
// LDS-DMA (wave 0) ahead of the prologue units, read from LDS after the wait + barrier at the tile top. (Rounds 1-3
// loaded the bias into registers in the epilogue: the first MFMA of every tile then sat behind an s_waitcnt vmcnt(0)
// that drained the just-issued prefetch of K-tile 1 and all of the previous tile's stores.)
template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                          typename OutOf<T>::type* __restrict__ C, int M, int N, int K,
                                                          int ntm, int ntn, GemmEpi256 e) {
  typedef typename OutOf<T>::type OT;
  constexpr bool FP8 = sizeof(T) == 1;
  constexpr bool Q8 = EPI == E_GELU_Q8;
  constexpr bool COLS8 = EPI != E_ROPE;  // column map of the X fragments (header comment)
  // vector-memory operations a wave issues per tile epilogue behind the next tile's prologue DMAs (exact: the wait at the next
  // tile top counts them): one store per output row and 16 bytes = 16 for 16-bit results with the 8-column map, 32 with the
  // 4-column map and for f32 results; e4m3 results: 16 8-byte stores + 1 atomic max
  constexpr int STORES_TILE = Q8 ? 17 : (COLS8 && sizeof(OT) == 2) ? 16 : 32;
  __shared__ __attribute__((aligned(16))) char smem[P_LDS + P_STAMPS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  [[maybe_unused]] int stamp_tile = 0;

  // this workgroup's share of the tile list: XCD x = blockIdx % 8 owns one contiguous chunk (xcd_remap's
  // split), its gridDim/8 workgroups take that chunk's tiles round-robin
  const int nwg = ntm * ntn;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int cq = nwg >> 3, cr = nwg & 7;
  const int cbase = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
  const int csize = cq + (xcd < cr ? 1 : 0);
  if (slot >= csize) return;  // workgroup-uniform, before any barrier

  const int GM = e.gm;
  const int per_group = GM * ntn;
  auto tile_origin = [&](int t, int& m0, int& n0) {
    const int group = t / per_group, first_m = group * GM;
    const int gsz = min(ntm - first_m, GM);
    m0 = (first_m + (t % per_group) % gsz) * 256;
    n0 = ((t % per_group) / gsz) * 256;
  };

  // Lane-derived values (fragment rows, DMA source offsets) are re-derived from an opaque copy of the lane id once
  // per tile and once per epilogue: kept as kernel-lifetime invariants they occupy ~45 VGPRs across the epilogue,
  // which then spills its RoPE table rows.
  auto fresh_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
  const size_t rowbytes = (size_t)K * sizeof(T);
  const char *a_base, *w_base;
  uint32_t soff[4][2];
  auto set_tile = [&](int m0, int n0) {
    a_base = reinterpret_cast<const char*>(A) + (size_t)m0 * rowbytes;
    w_base = reinterpret_cast<const char*>(W) + (size_t)n0 * rowbytes;
    dma_offsets<COLS8>(fresh_lane(), wid, M - 1 - m0, (uint32_t)rowbytes, soff);
  };
  const int nkt = K / (128 / (int)sizeof(T));
  auto unit_off = [](int u) { return (u == 0 ? 0 : u == 2 ? 1 : u == 3 ? 2 : 3) * P_UNIT; };
  auto stage = [&](int u, int kt) {
    if (kt < nkt) {
      char* dst = smem + (kt & 1) * P_BUF + unit_off(u) + wid * 2048;
      const char* base = ((u == 0 || u == 2) ? w_base : a_base) + (size_t)kt * 128;
      __builtin_amdgcn_global_load_lds(base + soff[u][0], NOVA_LDS_PTR(dst), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(base + soff[u][1], NOVA_LDS_PTR(dst + 1024), 16, 0, 0);
    }
  };
  const bool lds_bias = !FP8 && e.bias != nullptr;  // fp8: accumulators start at zero, the scales multiply the raw sums (bias added in the epilogue)
  // the tile's 256 bias values -> LDS, by wave 0, ahead of (= older than) the prologue units: landed when the wait at the tile top
  // returns (in-order retirement) and visible to all waves after that wait's barrier
  auto stage_prologue = [&](int n0) {
    if (lds_bias && wid == 0)
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(e.bias + n0) + lane * 16, NOVA_LDS_PTR(smem + P_BIAS), 16, 0, 0);
    stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
    stage(0, 1); stage(1, 1);
  };

  int fr, fg;
  f4v acc[4][8];  // [y fragment][x fragment]
  PFrag<T> xf[4][2], yf0[2][2], yf1[2][2];
  auto read_x = [&](const char* buf, int xi) {
    const char* u = buf + xi * P_UNIT;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) xf[f][kk] = punit_frag<T>(u, wr * 64 + f * 16 + fr, fg + 4 * kk);
  };
  auto read_y = [&](const char* buf, int yi, PFrag<T> (&yf)[2][2]) {
    const char* u = buf + (2 + yi) * P_UNIT;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) yf[f][kk] = punit_frag<T>(u, wc * 32 + f * 16 + fr, fg + 4 * kk);
  };
  // The accumulators start at the bias: the first MFMA of every accumulator (k half 0 of the tile's first K-tile) takes
  // the bias of its column, splat over the lane's 4 rows, as its C operand, so there is neither a zeroing pass nor a bias
  // add in the epilogue.
  f4v bcol[2];
  auto mma_quadrant = [&](int xi, int yi, PFrag<T> (&yf)[2][2], auto first_ktile) {
    constexpr bool FIRST = decltype(first_ktile)::value;
    f4v c0[4];
    if constexpr (FIRST) {
#pragma unroll
      for (int x = 0; x < 4; ++x) c0[x] = f4v{bcol[xi][x], bcol[xi][x], bcol[xi][x], bcol[xi][x]};
    }
    __builtin_amdgcn_s_setprio(1);
    if constexpr (FP8) {
      // MX-fp8: ONE v_mfma_scale_f32_16x16x128_f8f6f4 per accumulator and K-tile (twice the bf16 rate). A lane's 32 operand
      // bytes are the two 16-byte chunks (fg, fg + 4) it reads anyway - the same (lane, byte) -> k map on both operands,
      // which is all a contraction needs. Block scales are 1 (e8m0 127); the per-row scales are applied in the epilogue.
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const i8v yv = __builtin_bit_cast(i8v, __builtin_shufflevector(yf[f][0].v, yf[f][1].v, 0, 1, 2, 3, 4, 5, 6, 7));
          const i8v xv = __builtin_bit_cast(i8v, __builtin_shufflevector(xf[x][0].v, xf[x][1].v, 0, 1, 2, 3, 4, 5, 6, 7));
          acc[yi * 2 + f][xi * 4 + x] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
              yv, xv, FIRST ? c0[x] : acc[yi * 2 + f][xi * 4 + x], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int x = 0; x < 4; ++x)
            acc[yi * 2 + f][xi * 4 + x] = pmma(yf[f][kk], xf[x][kk], (FIRST && kk == 0) ? c0[x] : acc[yi * 2 + f][xi * 4 + x]);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto read_bias = [&]() {  // after the tile-top wait + barrier
#pragma unroll
    for (int xi = 0; xi < 2; ++xi)
      bcol[xi] = lds_bias ? *reinterpret_cast<const f4v*>(smem + P_BIAS + (wr * 128 + col_of<COLS8>(fr, xi)) * 4) : f4v{0.f, 0.f, 0.f, 0.f};
  };
  // one K-tile: 4 phases (see the header comment)
  auto ktile = [&](int kt, auto first_ktile) {
    const char* buf = smem + (kt & 1) * P_BUF;
    read_x(buf, 0);
    read_y(buf, 0, yf0);
    stage(2, kt + 1);
    NOVA_BARRIER();
    mma_quadrant(0, 0, yf0, first_ktile);
    NOVA_BARRIER();
    read_y(buf, 1, yf1);
    stage(3, kt + 1);
    NOVA_BARRIER();
    mma_quadrant(0, 1, yf1, first_ktile);
    NOVA_BARRIER();
    read_x(buf, 1);
    stage(0, kt + 2);
    NOVA_BARRIER();
    mma_quadrant(1, 1, yf1, first_ktile);
    NOVA_BARRIER();
    stage(1, kt + 2);
    if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NOVA_BARRIER();
    mma_quadrant(1, 0, yf0, first_ktile);
    NOVA_BARRIER();
  };

  if (e.stagger > 0) {  // phase-shift the workgroups so that their epilogue store bursts do not coincide chip-wide
    const long long until = clock64() + (long long)e.stagger * (int)blockIdx.x / (int)gridDim.x;
    while (clock64() < until) __builtin_amdgcn_s_sleep(8);
  }
  int m0, n0;
  int it = slot;
  auto tile_at = [&](int i) { return cbase + (e.rev ? csize - 1 - i : i); };
  tile_origin(tile_at(it), m0, n0);
  set_tile(m0, n0);
  stage_prologue(n0);
  bool first = true;
  for (;;) {
    { const int l = fresh_lane(); fr = l & 15; fg = l >> 4; }
    NOVA_STAMP(0);
    // K-tile 0 of this tile (and its bias) landed: all but (tile 1: X0, Y1) and, after the first tile, everything the previous
    // epilogue issued after the prologue DMAs (its STORES_TILE stores)
    if (nkt == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (first) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + STORES_TILE) : "memory");
    NOVA_BARRIER();
    if (wr == 1) NOVA_BARRIER();  // group 1 runs one barrier behind group 0 inside the K loop
    NOVA_STAMP(1);
    read_bias();
    ktile(0, std::true_type{});
    NOVA_STAMP(2);
    for (int kt = 1; kt < nkt - 1; ++kt) ktile(kt, std::false_type{});
    NOVA_STAMP(3);
    if (nkt > 1) ktile(nkt - 1, std::false_type{});
    NOVA_STAMP(4);
    if (wr == 0) NOVA_BARRIER();  // re-align the groups: every LDS slot is dead from here on
    NOVA_STAMP(5);

    // ---- epilogue
    // (opaque copies: otherwise the row/table address arithmetic below is hoisted above the K loop and its
    // results are spilled around it)
    int cm0 = m0, cn0 = n0;
    asm volatile("" : "+s"(cm0), "+s"(cn0));
    { const int l = fresh_lane(); fr = l & 15; fg = l >> 4; }
    const bool rot = EPI == E_ROPE && cn0 < e.rope_cols;
    const float qmul = (EPI == E_ROPE && cn0 < e.q_cols) ? e.q_scale : 1.0f;  // tile-uniform; x * 1.0f is exact
    const int nw = cn0 + wr * 128;  // the wave's first column; the lane's 4 columns of half xi start at nw + col_of(fr, xi)
    // fp8: out = acc * sa[row] * sw[col] + bias[col]; the three vectors are requested first and consumed before the DMAs
    if constexpr (FP8) {
      f4v swv[2], bfv[2], sav[4];
#pragma unroll
      for (int xi = 0; xi < 2; ++xi) {
        swv[xi] = *reinterpret_cast<const f4v*>(e.sw + nw + col_of<COLS8>(fr, xi));
        bfv[xi] = e.bias ? *reinterpret_cast<const f4v*>(e.bias + nw + col_of<COLS8>(fr, xi)) : f4v{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int r = 0; r < 4; ++r) sav[y][r] = e.sa[e.sa_scalar ? 0 : min(cm0 + wc * 64 + y * 16 + fg * 4 + r, M - 1)];
      asm volatile("" ::"v"(sav[3][3]));  // youngest of them: the compiler's wait sits here (loads retire in order)
      // dequantise in place, ahead of everything else: the scale / bias vectors are dead before the epilogue proper starts
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 8; ++x) acc[y][x] = (acc[y][x] * sav[y]) * swv[x >> 2][x & 3] + bfv[x >> 2][x & 3];
    }
    // RoPE table rows of the lane's 16 output rows: (cos0, sin0, cos1, sin1) of the pairs in its 4 columns. When the head
    // width divides 64 both column halves of the wave see the same table columns (the common case: head_dim 64); otherwise
    // (head_dim 96) the table rows of half 1 are loaded after half 0 is done.
    f4v cs[4][4];  // [y fragment][r]
    const bool same_cols = 64 % e.hd == 0;
    auto load_cs = [&](int xi) {
      const int tcol = (nw + col_of<COLS8>(fr, xi)) % e.hd;
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        // (sequence, position) of the fragment's first row on wave-uniform values, the lane's rows by increment
        const int mb = min(__builtin_amdgcn_readfirstlane(cm0 + wc * 64 + y * 16), M - 1);
        const int s0 = mb / e.L, l0 = mb - s0 * e.L;
        const int b0 = s0 % e.rope_batch, b1 = b0 + 1 == e.rope_batch ? 0 : b0 + 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int l = l0 + min(fg * 4 + r, M - 1 - mb);  // rows past M reuse row M-1 (never stored differently)
          int sb = b0;
          if (l >= e.L) { l -= e.L; sb = b1; }  // a 16-row block crosses at most one sequence boundary (L >= 16)
          cs[y][r] = *reinterpret_cast<const f4v*>(e.rope + ((size_t)sb * e.L + l) * e.hd + tcol);
        }
      }
    };
    float q8_inv = 1.0f, q8_max = 0.f;
    if constexpr (Q8) q8_inv = 1.0f / *e.q8_scale;
    // rows past M were staged as copies of row M-1, so their lanes hold row M-1's results and store the identical
    // bytes there again: no branch, and every tile issues the same number of stores (the vmcnt count above)
    auto row_of = [&](int y, int r) { return min(cm0 + wc * 64 + y * 16 + fg * 4 + r, M - 1); };
    auto value4 = [&](int y, int r, int xi) { return f4v{acc[y][xi * 4][r], acc[y][xi * 4 + 1][r], acc[y][xi * 4 + 2][r], acc[y][xi * 4 + 3][r]}; };
    auto finish_half = [&](int xi, auto rotated) {  // COLS4: the lane's 4 columns of half xi, all 16 rows
      constexpr bool ROT = decltype(rotated)::value;
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          store4<OT>(C + (size_t)row_of(y, r) * N + nw + col_of<false>(fr, xi), epi_apply<OT, EPI, ROT>(value4(y, r, xi), cs[y][r], qmul));
    };
    auto finish_rows = [&]() {  // COLS8: the lane's 8 consecutive columns, all 16 rows
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f4v lo = value4(y, r, 0), hi = value4(y, r, 1);
          if constexpr (Q8) {
            lo = gelu_erf_fast4(lo);
            hi = gelu_erf_fast4(hi);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              q8_max = fmaxf(q8_max, fmaxf(fabsf(lo[q]), fabsf(hi[q])));
              lo[q] = __builtin_amdgcn_fmed3f(lo[q] * q8_inv, -448.0f, 448.0f);
              hi[q] = __builtin_amdgcn_fmed3f(hi[q] * q8_inv, -448.0f, 448.0f);
            }
            int w0 = 0, w1 = 0;
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0], lo[1], w0, false);
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2], lo[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(hi[0], hi[1], w1, false);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(hi[2], hi[3], w1, true);
            // 8 columns = 8 bytes per lane; the 16 lanes of a row group write 128 contiguous bytes
            *reinterpret_cast<u2v*>(reinterpret_cast<uint8_t*>(C) + (size_t)row_of(y, r) * N + nw + fr * 8) = u2v{(uint32_t)w0, (uint32_t)w1};
          } else {
            store8<OT>(C + (size_t)row_of(y, r) * N + nw + fr * 8, epi_apply<OT, EPI, false>(lo, lo, qmul), epi_apply<OT, EPI, false>(hi, hi, qmul));
          }
        }
    };
    // the next tile's origin, then its prologue (bias + six units) ahead of this tile's stores in the memory pipeline
    // (issued behind them the DMAs queue for thousands of cycles)
    const int nxt = it + nslot;
    const bool more = nxt < csize;
    if (more) tile_origin(tile_at(nxt), m0, n0);
    auto next_prologue = [&]() {
      if (more) {
        set_tile(m0, n0);
        stage_prologue(n0);
      }
      NOVA_STAMP(6);
    };
    if constexpr (COLS8) {
      next_prologue();
      finish_rows();
    } else if (rot) {  // the rotation is tile-uniform (decided per 256-column tile): two straight-line epilogues behind one scalar branch
      load_cs(0);
      asm volatile("" ::"v"(cs[3][3]));  // the compiler's wait for the table rows sits here, ahead of the DMAs (loads retire in order)
      next_prologue();
      finish_half(0, std::true_type{});
      if (!same_cols) load_cs(1);  // (waits for everything in flight: head widths that do not divide 64 only)
      finish_half(1, std::true_type{});
    } else {
      next_prologue();
      finish_half(0, std::false_type{});
      finish_half(1, std::false_type{});
    }
    if constexpr (Q8) {  // one atomic max per wave and tile (values are >= 0: float order == unsigned order of the bits)
      q8_max = wave_max(q8_max);
      if (fresh_lane() == 0) __hip_atomic_fetch_max(e.q8_amax, __float_as_uint(q8_max), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    NOVA_STAMP(7);
#ifdef NOVA_STAMPS
    ++stamp_tile;
#endif
    if (!more) break;
    first = false;
    it = nxt;
  }
#ifdef NOVA_STAMPS
  if (blockIdx.x == 0) {
    __syncthreads();
    for (int i = tid; i < 2 * STAMP_TILES * STAMP_PTS; i += 512) g_gemm_stamps[i] = reinterpret_cast<uint32_t*>(smem + P_LDS)[i];
  }
#endif
}

// ------------------------------------------------------------------------------------------
// Continuous form (round 4): the K-stream does not stop at a tile boundary. In-kernel stamps of the prologue form above
// (tools/gemm_stamps.py, profiles/r04_gemm_stamps_prologue_form.txt) put 14-17k of a K = 1024 tile's 55-57k cycles into the seam,
// whatever the epilogue computes, and show why: the next tile's first K-tiles are requested at the START of the epilogue, side by
// side with its 16-32 stores per wave in the CU's one memory pipeline - the tile top waits for K-tile 0 queued behind stores
// (2-6k cycles), K-tile 0 then runs at half speed because its own wait needs every older store retired (in-order vmcnt) while
// K-tile 1 is still arriving, and re-ordering inside the seam only moves the time around (a barrier between the DMAs and the
// stores: top wait 4.2k -> 1.8k, K loop + 2k).
// Here the staging of the K loop simply carries on into the next tile: phase p of K-tile kt requests the unit that is 6 phases
// ahead in the FLAT sequence of (tile, K-tile) pairs, so the last two K-tiles of a tile request K-tiles 0 and 1 of the next one
// (same LDS slots, same distances, the same counted wait: nothing about the K loop's LDS protocol changes), wave 0 requests the
// next tile's bias with them, and when a tile's last MFMA retires the next tile's first K-tile is already in LDS. The epilogue is
// then VALU work and stores with no DMA beside them; there is no tile-top wait, no prologue, and the two wave groups stay one
// barrier apart across tiles (no re-align / re-skew pair): each group's epilogue runs under the other group's MFMA phase.
// The stores are retired by the first K-tile's ordinary wait (they are older than the units it covers).
// Preconditions (launcher): an even number of K-tiles (buffer parity carries over) and at least 4 of them.
template <typename T, int EPI, bool TDMA = false>
__global__ __launch_bounds__(512, 2) void gemm256c_kernel(const T* __restrict__ A, const T* __restrict__ W, T* __restrict__ C, int M,
                                                          int N, int K, int ntm, int ntn, GemmEpi256 e) {
  typedef T OT;
  static_assert(!TDMA || EPI == E_ROPE, "the table-through-LDS form belongs to the RoPE epilogue");
  // TDMA (RoPE, head width 64): the cos / sin rows come through LDS - each wave brings the 16 table rows of one Y fragment
  // (16 x 256 B) into a window of its own by LDS-DMA, one fragment ahead of the one it is rotating, and reads them back with
  // ds_read_b128. Through registers the rows are 16-32 global loads per lane and tile at ~75 cycles of the CU's memory pipeline
  // each (tools/ta_probe.hip) - with the stores, 15k cycles of pipeline time per tile, which IS the epilogue of the fused-QKV
  // GEMM once the prologue is out of it; an LDS-DMA instruction moves the same kilobyte in 21-33. The windows take the 32 KiB
  // beside the K-tile buffers (so the bias travels in registers in this form) and, since the table is read from LDS, the lane
  // can hold 8 consecutive columns (16-byte stores) like every other epilogue.
  constexpr bool COLS8 = EPI != E_ROPE || TDMA;  // column map of the X fragments (header comment)
  constexpr int P_TAB = P_KLDS;                   // TDMA: 8 wave windows of 4 KiB
  __shared__ __attribute__((aligned(16))) char smem[TDMA ? P_KLDS + 8 * 4096 : P_LDS + P_STAMPS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  [[maybe_unused]] int stamp_tile = TDMA ? 1 << 20 : 0;  // (no stamp record in the TDMA form: its LDS is full)

  const int nwg = ntm * ntn;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int cq = nwg >> 3, cr = nwg & 7;
  const int cbase = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
  const int csize = cq + (xcd < cr ? 1 : 0);
  if (slot >= csize) return;  // workgroup-uniform, before any barrier

#ifdef NOVA_CLOCK
  const long long ck_t0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int GM = e.gm;
  const int per_group = GM * ntn;
  auto tile_origin = [&](int t, int& m0, int& n0) {
    const int group = t / per_group, first_m = group * GM;
    const int gsz = min(ntm - first_m, GM);
    m0 = (first_m + (t % per_group) % gsz) * 256;
    n0 = ((t % per_group) / gsz) * 256;
  };
  auto fresh_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
  const size_t rowbytes = (size_t)K * sizeof(T);
  // the tile being STAGED (the tile being computed until the switch in the second-to-last K-tile, the next one after it)
  const char *a_base, *w_base;
  uint32_t soff[4][2];
  auto set_tile = [&](int m0, int n0) {
    a_base = reinterpret_cast<const char*>(A) + (size_t)m0 * rowbytes;
    w_base = reinterpret_cast<const char*>(W) + (size_t)n0 * rowbytes;
    dma_offsets<COLS8>(fresh_lane(), wid, M - 1 - m0, (uint32_t)rowbytes, soff);
  };
  const int nkt = K / (128 / (int)sizeof(T));
  auto unit_off = [](int u) { return (u == 0 ? 0 : u == 2 ? 1 : u == 3 ? 2 : 3) * P_UNIT; };
  auto stage = [&](int u, int kt) {  // unit u of K-tile kt of the staged tile
    char* dst = smem + (kt & 1) * P_BUF + unit_off(u) + wid * 2048;
    const char* base = ((u == 0 || u == 2) ? w_base : a_base) + (size_t)kt * 128;
    // (the `nt` policy on either operand's staging loads loses 3-10 %: profiles/r04_gemm_nt_stores_ab.txt)
    __builtin_amdgcn_global_load_lds(base + soff[u][0], NOVA_LDS_PTR(dst), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(base + soff[u][1], NOVA_LDS_PTR(dst + 1024), 16, 0, 0);
  };
  const bool lds_bias = !TDMA && e.bias != nullptr;
  auto stage_bias = [&](int n0) {  // the tile's 256 bias values -> LDS, by wave 0 (older than the units requested after it)
    if (lds_bias && wid == 0)
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(e.bias + n0) + lane * 16, NOVA_LDS_PTR(smem + P_BIAS), 16, 0, 0);
  };

  int fr, fg;
  f4v acc[4][8];  // [y fragment][x fragment]
  PFrag<T> xf[4][2], yf0[2][2], yf1[2][2];
  auto read_x = [&](const char* buf, int xi) {
    const char* u = buf + xi * P_UNIT;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) xf[f][kk] = punit_frag<T>(u, wr * 64 + f * 16 + fr, fg + 4 * kk);
  };
  auto read_y = [&](const char* buf, int yi, PFrag<T> (&yf)[2][2]) {
    const char* u = buf + (2 + yi) * P_UNIT;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) yf[f][kk] = punit_frag<T>(u, wc * 32 + f * 16 + fr, fg + 4 * kk);
  };
  f4v bcol[2];
  // TDMA: the table rows of Y fragment y of tile rows m0 + wc*64 + 16 y + [0, 16) -> this wave's window (4 instructions of 4 rows)
  auto stage_table = [&](int m0, int y) {
    const int mb = min(__builtin_amdgcn_readfirstlane(m0 + wc * 64 + y * 16), M - 1);
    const int s0 = mb / e.L, l0 = mb - s0 * e.L;
    const int b0 = s0 % e.rope_batch, b1 = b0 + 1 == e.rope_batch ? 0 : b0 + 1;
    const int ln = fresh_lane();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int l = l0 + min(i * 4 + (ln >> 4), M - 1 - mb);  // rows past M reuse row M-1
      int sb = b0;
      if (l >= e.L) { l -= e.L; sb = b1; }  // a 16-row block crosses at most one sequence boundary (L >= 16)
      __builtin_amdgcn_global_load_lds(e.rope + ((size_t)sb * e.L + l) * 64 + (ln & 15) * 4, NOVA_LDS_PTR(smem + P_TAB + wid * 4096 + i * 1024), 16, 0, 0);
    }
  };
  auto mma_quadrant = [&](int xi, int yi, PFrag<T> (&yf)[2][2], auto first_ktile) {
    constexpr bool FIRST = decltype(first_ktile)::value;
    f4v c0[4];
    if constexpr (FIRST) {
#pragma unroll
      for (int x = 0; x < 4; ++x) c0[x] = f4v{bcol[xi][x], bcol[xi][x], bcol[xi][x], bcol[xi][x]};
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int x = 0; x < 4; ++x)
          acc[yi * 2 + f][xi * 4 + x] = pmma(yf[f][kk], xf[x][kk], (FIRST && kk == 0) ? c0[x] : acc[yi * 2 + f][xi * 4 + x]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto read_bias = [&]() {
    if constexpr (TDMA) return;  // in registers since the previous epilogue (or the kernel entry)
#pragma unroll
    for (int xi = 0; xi < 2; ++xi)
      bcol[xi] = lds_bias ? *reinterpret_cast<const f4v*>(smem + P_BIAS + (wr * 128 + col_of<COLS8>(fr, xi)) * 4) : f4v{0.f, 0.f, 0.f, 0.f};
  };

  int m0, n0, nm0 = 0, nn0 = 0;
  bool more = false;
  // One K-tile = 4 phases (header comment). MODE 0: inside a tile (K-tiles 0 .. nkt-3): requests K-tiles kt+1, kt+2 of the same
  // tile. MODE 1: the second-to-last K-tile: the staged tile switches to the next one between phases 1 and 2. MODE 2: the last
  // K-tile: all four requests belong to the next tile. Without a next tile modes 1 / 2 request nothing and drain.
  auto ktile = [&](int kt, auto first_ktile, auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    const char* buf = smem + (kt & 1) * P_BUF;
    read_x(buf, 0);
    read_y(buf, 0, yf0);
    if (MODE == 0 || MODE == 1) stage(2, kt + 1);
    else if (more) stage(2, 0);
    NOVA_BARRIER();
    mma_quadrant(0, 0, yf0, first_ktile);
    NOVA_BARRIER();
    read_y(buf, 1, yf1);
    if (MODE == 0 || MODE == 1) stage(3, kt + 1);
    else if (more) stage(3, 0);
    NOVA_BARRIER();
    mma_quadrant(0, 1, yf1, first_ktile);
    NOVA_BARRIER();
    read_x(buf, 1);
    if (MODE == 0) {
      stage(0, kt + 2);
    } else if (MODE == 1) {
      if (more) {
        set_tile(nm0, nn0);  // from here on the next tile is the staged one
        stage_bias(nn0);
        stage(0, 0);
      }
    } else if (more) {
      stage(0, 1);
    }
    NOVA_BARRIER();
    mma_quadrant(1, 1, yf1, first_ktile);
    NOVA_BARRIER();
    if (MODE == 0) {
      stage(1, kt + 2);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else if (MODE == 1) {
      if (more) {
        stage(1, 0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // the last K-tile of this tile has landed (and, wave 0, the next bias)
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else if (more) {
      stage(1, 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // K-tile 0 of the next tile has landed
    }
    if constexpr (TDMA && MODE == 2) {
      if (n0 < e.rope_cols) stage_table(m0, 0);  // behind the wait: the first fragment's table rows arrive under this tile's last MFMAs
    }
    NOVA_BARRIER();
    mma_quadrant(1, 0, yf0, first_ktile);
    NOVA_BARRIER();
  };

  if (e.stagger > 0) {
    const long long until = clock64() + (long long)e.stagger * (int)blockIdx.x / (int)gridDim.x;
    while (clock64() < until) __builtin_amdgcn_s_sleep(8);
  }
  int it = slot;
  auto tile_at = [&](int i) { return cbase + (e.rev ? csize - 1 - i : i); };
  tile_origin(tile_at(it), m0, n0);
  set_tile(m0, n0);
  auto load_bias_regs = [&](int n0) {  // TDMA: the lane's 8 bias values by ordinary loads, waited for on the spot (nothing else in flight)
    const int f = fresh_lane() & 15;
#pragma unroll
    for (int xi = 0; xi < 2; ++xi)
      bcol[xi] = e.bias ? *reinterpret_cast<const f4v*>(e.bias + n0 + wr * 128 + col_of<COLS8>(f, xi)) : f4v{0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(bcol[0]), "+v"(bcol[1]));
  };
  if constexpr (TDMA) load_bias_regs(n0);
  stage_bias(n0);
  stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
  stage(0, 1); stage(1, 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  NOVA_BARRIER();
  if (wr == 1) NOVA_BARRIER();  // group 1 runs one barrier behind group 0 inside the K loop
  for (;;) {
    { const int l = fresh_lane(); fr = l & 15; fg = l >> 4; }
    NOVA_STAMP(0);
    const int nxt = it + nslot;
    more = nxt < csize;
    if (more) tile_origin(tile_at(nxt), nm0, nn0);
    read_bias();
    NOVA_STAMP(1);
    ktile(0, std::true_type{}, std::integral_constant<int, 0>{});
    NOVA_STAMP(2);
    for (int kt = 1; kt < nkt - 2; ++kt) ktile(kt, std::false_type{}, std::integral_constant<int, 0>{});
    NOVA_STAMP(3);
    ktile(nkt - 2, std::false_type{}, std::integral_constant<int, 1>{});
    ktile(nkt - 1, std::false_type{}, std::integral_constant<int, 2>{});
    NOVA_STAMP(4);
    // Both groups run their epilogue at the same time (group 0 waits one barrier for group 1's last MFMA phase, group 1 spends
    // one barrier after the epilogue to fall behind again). Left one barrier apart across the tile boundary the groups take
    // turns - a group's epilogue holds the other one at its next barrier - and the seam is the SUM of the two epilogues
    // (stamps: K-tile 0 of group 0 10.1k cycles, the last K-tile of group 1 13.5k, against 2.5k for an ordinary K-tile).
    if (wr == 0) NOVA_BARRIER();
    NOVA_STAMP(5);

    // ---- epilogue of tile (m0, n0): no LDS traffic, no DMA issued beside it
    int cm0 = m0, cn0 = n0;
    asm volatile("" : "+s"(cm0), "+s"(cn0));
    { const int l = fresh_lane(); fr = l & 15; fg = l >> 4; }
    const bool rot = EPI == E_ROPE && cn0 < e.rope_cols;
    const int nw = cn0 + wr * 128;  // the wave's first column; the lane's 4 columns of half xi start at nw + col_of(fr, xi)
    f4v cs[4][4];  // RoPE table rows of the lane's 16 output rows: (cos0, sin0, cos1, sin1) of the pairs in its 4 columns
    const bool same_cols = 64 % e.hd == 0;  // both 64-column halves of the wave see the same table columns (head_dim 64)
    auto load_cs = [&](int xi) {
      const int tcol = (nw + col_of<COLS8>(fr, xi)) % e.hd;
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        const int mb = min(__builtin_amdgcn_readfirstlane(cm0 + wc * 64 + y * 16), M - 1);
        const int s0 = mb / e.L, l0 = mb - s0 * e.L;
        const int b0 = s0 % e.rope_batch, b1 = b0 + 1 == e.rope_batch ? 0 : b0 + 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int l = l0 + min(fg * 4 + r, M - 1 - mb);  // rows past M reuse row M-1 (never stored differently)
          int sb = b0;
          if (l >= e.L) { l -= e.L; sb = b1; }  // a 16-row block crosses at most one sequence boundary (L >= 16)
          cs[y][r] = *reinterpret_cast<const f4v*>(e.rope + ((size_t)sb * e.L + l) * e.hd + tcol);
        }
      }
    };
    // Output addressing: a wave-uniform 64-bit tile base + a 32-bit lane offset per row (rows past M are copies of row M-1 and store
    // the identical bytes there again, so every tile issues the same number of stores)
    OT* const ctile = C + (size_t)cm0 * N + nw;
    const int rlim = M - 1 - cm0;
    auto row_off = [&](int y, int r) { return (uint32_t)min(wc * 64 + y * 16 + fg * 4 + r, rlim) * (uint32_t)N; };
    // 4 / 8 consecutive columns of output row (y, r): component r of 4 / 8 accumulators. The packing instruction takes any two
    // registers, so nothing has to be moved together first.
    auto put4 = [&](int y, int r, int xi, f4v v) { store4<OT>(ctile + row_off(y, r) + col_of<false>(fr, xi), v); };
    auto value4 = [&](int y, int r, int xi) { return f4v{acc[y][xi * 4][r], acc[y][xi * 4 + 1][r], acc[y][xi * 4 + 2][r], acc[y][xi * 4 + 3][r]}; };
    auto put8 = [&](int y, int r, f4v lo, f4v hi) { store8<OT>(ctile + row_off(y, r) + fr * 8, lo, hi); };
    // The elementwise epilogues run IN PLACE on the accumulators, one register quad (4 rows of one column) at a time: the packed
    // f32 forms (v_pk_mul / v_pk_fma, gelu_erf_fast4) take aligned register pairs, and applied to the per-row gathers of the
    // stores below they cost one v_mov per operand pair on top (194 moves in 1096 vector instructions of the fc1 epilogue).
    // (activations: applied row block by row block beside the stores, see act_rows below)
    constexpr bool ACT_ROWS = NOVA_GEMM_ACT_ROWS && (EPI == E_GELU || EPI == E_SILU) && COLS8 && !TDMA;
    if constexpr ((EPI == E_GELU || EPI == E_SILU) && !ACT_ROWS) {
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 8; ++x) acc[y][x] = epi_apply<OT, EPI, false>(acc[y][x], acc[y][x], 1.0f);
    }
    const bool qs = EPI == E_ROPE && cn0 < e.q_cols;  // a q tile: results times the softmax scale (after the rotation)
    if (EPI == E_ROPE && qs && !rot) {  // (not a combination the model produces: q columns are rotated)
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 8; ++x) acc[y][x] = acc[y][x] * e.q_scale;
    }
    auto rotated = [&](f4v v, f4v t, auto scaled) {  // pair rotation, then (q tiles) the scale: the order every GEMM structure rounds in
      v = rope_rotate4(v, t);
      if constexpr (decltype(scaled)::value) v = v * e.q_scale;
      return v;
    };
    const float qmul = qs ? e.q_scale : 1.0f;  // x * 1.0f is exact
    auto finish_half = [&](int xi) {  // 4-column map, rotated tile (one code path: this form holds 64 table registers)
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int r = 0; r < 4; ++r) put4(y, r, xi, rope_rotate4(value4(y, r, xi), cs[y][r]) * qmul);
    };
    auto plain_rows = [&]() {  // no rotation: the accumulators as they stand
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (COLS8) {
            put8(y, r, value4(y, r, 0), value4(y, r, 1));
          } else {
            put4(y, r, 0, value4(y, r, 0));
            put4(y, r, 1, value4(y, r, 1));
          }
        }
    };
    NOVA_STAMP(6);
    if constexpr (TDMA) {
      // everything in flight lands here: the two units of the next tile's K-tile 1 and, youngest, table fragment 0; then the
      // next tile's bias (requested and consumed on the spot, so that no later use of it waits for this epilogue's stores)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (more) load_bias_regs(nn0);
      if (rot) {
        constexpr int ST = sizeof(OT) == 2 ? 4 : 8;  // stores of one fragment pass
        const char* win = smem + P_TAB + wid * 4096 + (fr & 7) * 32;  // both heads of the wave's 128 columns use the same 64 table floats
        auto passes = [&](auto scaled) {
#pragma unroll
          for (int y = 0; y < 4; ++y) {
            if (y > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ST) : "memory");  // fragment y's rows landed (the previous pass's stores may be in flight)
            f4v tlo[4], thi[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              tlo[r] = *reinterpret_cast<const f4v*>(win + (fg * 4 + r) * 256);
              thi[r] = *reinterpret_cast<const f4v*>(win + (fg * 4 + r) * 256 + 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the window is read: the next fragment's rows may overwrite it
            __builtin_amdgcn_sched_barrier(0);
            if (y < 3) stage_table(cm0, y + 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) put8(y, r, rotated(value4(y, r, 0), tlo[r], scaled), rotated(value4(y, r, 1), thi[r], scaled));
          }
        };
        if (qs) passes(std::true_type{});
        else passes(std::false_type{});
      } else {
        plain_rows();
      }
    } else if constexpr (ACT_ROWS) {
      // GELU / SiLU one 16-row block at a time, its four stores right behind it: the memory pipeline works the stores off while the vector
      // pipe is on the next block's activation (hipcc's own order: all 128 exponentials and reciprocals, then all 16 stores)
#pragma unroll
      for (int y = 0; y < 4; ++y) {
#pragma unroll
        for (int x = 0; x < 8; ++x) acc[y][x] = epi_apply<OT, EPI, false>(acc[y][x], acc[y][x], 1.0f);
#pragma unroll
        for (int r = 0; r < 4; ++r) put8(y, r, value4(y, r, 0), value4(y, r, 1));
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (COLS8) {
      plain_rows();
    } else if (rot) {
      load_cs(0);  // (the compiler waits for everything in flight at their first use: the two units of the next tile's K-tile 1)
      finish_half(0);
      if (!same_cols) load_cs(1);
      finish_half(1);
    } else {
      plain_rows();
    }
    NOVA_STAMP(7);
#ifdef NOVA_STAMPS
    ++stamp_tile;
#endif
    if (!more) break;
    if (wr == 1) NOVA_BARRIER();  // one barrier behind group 0 again
    it = nxt;
    m0 = nm0;
    n0 = nn0;
  }
#ifdef NOVA_CLOCK
  if (tid == 0 && blockIdx.x < 1024) {
    g_gemm_clock[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - ck_t0;
    g_gemm_clock[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - ck_r0;
  }
#endif
#ifdef NOVA_STAMPS
  if (!TDMA && blockIdx.x == 0) {
    __syncthreads();
    for (int i = tid; i < 2 * STAMP_TILES * STAMP_PTS; i += 512) g_gemm_stamps[i] = reinterpret_cast<uint32_t*>(smem + P_LDS)[i];
  }
#endif
}

#ifdef NOVA_CLOCK
}  // namespace nova
extern "C" int nova_debug_gemm_clock(long long* out, int n) {  // diagnostic build only: (shader cycles, 100 MHz ticks) per workgroup of the last launch
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(nova::g_gemm_clock), (size_t)(n < 2048 ? n : 2048) * 8, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
namespace nova {
#endif
#ifdef NOVA_STAMPS
}  // namespace nova
extern "C" int nova_debug_gemm_stamps(unsigned* out, int n) {  // diagnostic build only: the record of the last persistent launch
  const int m = n < 2 * nova::STAMP_TILES * nova::STAMP_PTS ? n : 2 * nova::STAMP_TILES * nova::STAMP_PTS;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(nova::g_gemm_stamps), (size_t)m * 4, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
namespace nova {
#endif
#ifdef NOVA_EXPERIMENTS
static int g_gm256 = 8;
static int g_stagger256 = 0;
void gemm256_set_stagger(int cycles) { g_stagger256 = cycles; }
void gemm256_set_gm(int g) { g_gm256 = g; }
static int g_var256 = 20;  // measured best (tools/gemm_variants.py): the persistent form; 0-3 = one tile per workgroup, LDS-DMA placement variants
void gemm256_set_variant(int v) { g_var256 = v; }
#else
constexpr int g_gm256 = 8;       // row panels per tile group
constexpr int g_stagger256 = 0;  // no start stagger
constexpr int g_var256 = 20;     // the persistent form
#endif

template <typename T, int VAR>
static int launch256v(const void* A, const void* W, void* C, int M, int N, int K, int epi, const GemmEpi256& e,
                     hipStream_t st) {
  const int ntm = (M + 255) / 256, ntn = N / 256;
  dim3 grid(ntm * ntn), block(512);
  ProfScope prof(prof_gemm_slot(epi, N, K), 2.0 * M * N * K, st);
  const T* a = static_cast<const T*>(A);
  const T* w = static_cast<const T*>(W);
  T* c = static_cast<T*>(C);
  switch (epi) {
    case E_NONE: hipLaunchKernelGGL((gemm256_kernel<T, E_NONE, VAR>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    case E_GELU: hipLaunchKernelGGL((gemm256_kernel<T, E_GELU, VAR>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    case E_SILU: hipLaunchKernelGGL((gemm256_kernel<T, E_SILU, VAR>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    case E_ROPE: hipLaunchKernelGGL((gemm256_kernel<T, E_ROPE, VAR>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    default: return set_error(NOVA_ERR_ARG, "gemm256: unknown epilogue %d", epi);
  }
  return check_launch("gemm256");
}

#ifdef NOVA_EXPERIMENTS
static int g_grid256 = 0;  // 0 = one workgroup per CU; otherwise the persistent grid size (a multiple of 8)
void gemm256_set_grid(int n) { g_grid256 = n & ~7; }
#else
constexpr int g_grid256 = 0;
#endif
static int cu_slots() {  // persistent grid: one workgroup per CU, a multiple of the 8 XCDs
  if (g_grid256 > 0) return g_grid256;
  static int n = 0;
  if (n == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
    n = cus & ~7;
  }
  return n;
}

int gemm256_cu_count() { return cu_slots(); }  // the persistent grid = CUs of the device (gemm.hip's kernel choice reads it)

template <typename T>
static int launch256p(const void* A, const void* W, void* C, int M, int N, int K, int epi, const GemmEpi256& e,
                      hipStream_t st) {
  const int ntm = (M + 255) / 256, ntn = N / 256;
  dim3 grid(cu_slots()), block(512);
  ProfScope prof(prof_gemm_slot(epi, N, K), 2.0 * M * N * K, st);
  const T* a = static_cast<const T*>(A);
  const T* w = static_cast<const T*>(W);
  T* c = static_cast<T*>(C);
  switch (epi) {
    case E_NONE: hipLaunchKernelGGL((gemm256p_kernel<T, E_NONE>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    case E_GELU: hipLaunchKernelGGL((gemm256p_kernel<T, E_GELU>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    case E_SILU: hipLaunchKernelGGL((gemm256p_kernel<T, E_SILU>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    case E_ROPE: hipLaunchKernelGGL((gemm256p_kernel<T, E_ROPE>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    default: return set_error(NOVA_ERR_ARG, "gemm256: unknown epilogue %d", epi);
  }
  return check_launch("gemm256p");
}

// MX-fp8 operands (OCP e4m3 bytes, K % 128 == 0), bf16 result; persistent kernel only. Internal entry for capi.hip.
// epi E_ROPE: the fused-QKV epilogue (rotation of the first rope_cols columns with the [rope_batch, L, hd/2, 2] table,
// q_scale on the first q_cols columns), applied after the dequantisation scales and the bias.
int gemm256_fp8_launch(const void* A8, const float* sa, const void* W8, const float* sw, const float* bias, void* C, int M,
                       int N, int K, int epi, hipStream_t st, const float* rope, int L, int rope_batch, int hd, int rope_cols,
                       float q_scale, int q_cols, int sa_scalar, const float* q8_scale, unsigned* q8_amax) {
  if (M <= 0) return 0;
  if (N % 256 != 0 || K % 128 != 0 || K <= 0) return set_error(NOVA_ERR_SHAPE, "gemm_fp8: need N %% 256 == 0 and K %% 128 == 0 (got N=%d K=%d)", N, K);
  if (epi == E_ROPE && (rope_cols % 256 || q_cols % 256 || (rope && (L < 16 || rope_batch <= 0 || hd <= 0))))
    return set_error(NOVA_ERR_SHAPE, "gemm_fp8: RoPE epilogue needs rope_cols, q_cols %% 256 == 0 and L >= 16");
  GemmEpi256 e{bias, rope, rope ? L : 1, rope ? rope_batch : 1, rope ? hd : 2, rope ? rope_cols : 0, q_scale, q_cols, g_gm256,
               walk_is_reverse() ? 1 : 0, sa, sw, 0};
  e.sa_scalar = sa_scalar;
  e.q8_scale = q8_scale;
  e.q8_amax = q8_amax;
  if (epi == E_GELU_Q8 && (!q8_scale || !q8_amax)) return set_error(NOVA_ERR_ARG, "gemm_fp8: the e4m3-output epilogue needs a scale and an amax word");
  const int ntm = (M + 255) / 256, ntn = N / 256;
  dim3 grid(cu_slots()), block(512);
  ProfScope prof(prof_gemm_slot(epi == E_GELU_Q8 ? E_GELU : epi, N, K), 2.0 * M * N * K, st);
  const fp8_t* a = static_cast<const fp8_t*>(A8);
  const fp8_t* w = static_cast<const fp8_t*>(W8);
  bf16_t* c = static_cast<bf16_t*>(C);
  switch (epi) {
    case E_NONE: hipLaunchKernelGGL((gemm256p_kernel<fp8_t, E_NONE>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    case E_GELU: hipLaunchKernelGGL((gemm256p_kernel<fp8_t, E_GELU>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    case E_SILU: hipLaunchKernelGGL((gemm256p_kernel<fp8_t, E_SILU>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    case E_ROPE: hipLaunchKernelGGL((gemm256p_kernel<fp8_t, E_ROPE>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    case E_GELU_Q8: hipLaunchKernelGGL((gemm256p_kernel<fp8_t, E_GELU_Q8>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    default: return set_error(NOVA_ERR_ARG, "gemm_fp8: unknown epilogue %d", epi);
  }
  return check_launch("gemm256p fp8");
}

template <typename T>
static int launch256c(const void* A, const void* W, void* C, int M, int N, int K, int epi, const GemmEpi256& e,
                      hipStream_t st) {
  const int ntm = (M + 255) / 256, ntn = N / 256;
  dim3 grid(cu_slots()), block(512);
  ProfScope prof(prof_gemm_slot(epi, N, K), 2.0 * M * N * K, st);
  const T* a = static_cast<const T*>(A);
  const T* w = static_cast<const T*>(W);
  T* c = static_cast<T*>(C);
  switch (epi) {
    case E_NONE: hipLaunchKernelGGL((gemm256c_kernel<T, E_NONE>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    case E_GELU: hipLaunchKernelGGL((gemm256c_kernel<T, E_GELU>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    case E_SILU: hipLaunchKernelGGL((gemm256c_kernel<T, E_SILU>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e); break;
    case E_ROPE:
      // head width 64 with a table: the table rows through LDS and 8 consecutive columns per lane; other head widths (96: d48w1536)
      // and the table-less q-scale-only use: table rows through registers, 4 consecutive columns per lane and column half
      if (e.rope && e.hd == 64) hipLaunchKernelGGL((gemm256c_kernel<T, E_ROPE, true>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e);
      else hipLaunchKernelGGL((gemm256c_kernel<T, E_ROPE, false>), grid, block, 0, st, a, w, c, M, N, K, ntm, ntn, e);
      break;
    default: return set_error(NOVA_ERR_ARG, "gemm256: unknown epilogue %d", epi);
  }
  return check_launch("gemm256c");
}

// form: 0 = the shipped choice (continuous form where its preconditions hold, else the prologue form), 1 = one tile per
// workgroup, 2 = the persistent prologue form (nova_debug_force_gemm_tile 257 / 258: A/B and the bitwise tests)
template <typename T>
static int launch256(const void* A, const void* W, void* C, int M, int N, int K, int epi, const GemmEpi256& e,
                     hipStream_t st, int form) {
  // (the persistent RoPE epilogues step through a 16-row block assuming it crosses at most one sequence boundary)
  if (form != 1 && g_var256 == 20 && !(epi == E_ROPE && e.rope && e.L < 16)) {
    const int nkt = K / (128 / (int)sizeof(T));
    if (form == 0 && nkt >= 4 && (nkt & 1) == 0) return launch256c<T>(A, W, C, M, N, K, epi, e, st);
    return launch256p<T>(A, W, C, M, N, K, epi, e, st);
  }
#ifdef NOVA_EXPERIMENTS
  switch (g_var256) {
    case 1: return launch256v<T, 1>(A, W, C, M, N, K, epi, e, st);
    case 3: return launch256v<T, 3>(A, W, C, M, N, K, epi, e, st);
    case 0: return launch256v<T, 0>(A, W, C, M, N, K, epi, e, st);
    case 10: return launch256v<T, 10>(A, W, C, M, N, K, epi, e, st);
    case 11: return launch256v<T, 11>(A, W, C, M, N, K, epi, e, st);
    case 12: return launch256v<T, 12>(A, W, C, M, N, K, epi, e, st);
    default: break;
  }
#endif
  return launch256v<T, 2>(A, W, C, M, N, K, epi, e, st);
}

// Entry used by gemm.hip's dispatcher. Preconditions (checked by the caller): N % 256 == 0,
// K % (128 / sizeof(T)) == 0, K > 0, M > 0.
int gemm256_launch(const void* A, const void* W, void* C, int M, int N, int K, int epi, const float* bias,
                   const float* rope, int L, int rope_batch, int hd, int rope_cols, float q_scale, int q_cols, int dtype,
                   hipStream_t st, int form) {
#ifdef NOVA_EXPERIMENTS
  const int gm = g_gm256;
#else
  // row panels per tile group: 8 (the L2-hit maximum of round 3's sweep), 16 where a row of tiles is at most 4 wide and K is short
  // (the out-projection: 0.282 against 0.294 ms at 163840 x 1024 x 1024; K = 4096 loses 6 % with 16: profiles/r04_gemm_group_height_sweep.txt)
  const int gm = (N <= 1024 && K * (int)(dtype_is16(dtype) ? 2 : 4) <= 2048) ? 16 : g_gm256;
#endif
  GemmEpi256 e{bias, rope, L, rope_batch, hd, rope_cols, q_scale, q_cols, gm, walk_is_reverse() ? 1 : 0, nullptr, nullptr, g_stagger256};
  return dispatch_dtype(dtype, [&](auto tag) { return launch256<decltype(tag)>(A, W, C, M, N, K, epi, e, st, form); });
}

}  // namespace nova
