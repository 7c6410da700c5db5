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
// Phase plan. A K-tile (64 bf16 / 32 f32 per row) is staged as 4 units of 128 rows x 128 B:
//   A(mi): rows {wr*128 + mi*64 + [0,64)}  of the A tile, for both wr      (read in phase 0 / 2)
//   W(ni): rows {wc*64  + ni*32 + [0,32)}  of the W tile, for all four wc  (read in phase 0 / 1)
// A wave (wr, wc) owns out[wr*128 + 128][wc*64 + 64] = 8 x 4 MFMA 16x16 fragments. Per K-tile, 4 phases:
//   p0: read A(0), W(0) -> quadrant (0,0)   p1: read W(1) -> (0,1)   p2: read A(1) -> (1,1)
//   p3: no LDS reads, quadrant (1,0) on the W(0) fragments kept in registers since p0
// 16 MFMAs each (8 independent accumulators per k-half). Staging order is A0, W1, A1, W0 per tile (flat index
// q = 4*tile + unit); phase p of tile t issues q = 4t + p + 6, i.e. always the unit whose LDS slot (2 K-tile
// buffers x 4 units = 128 KiB) had its last read at least TWO phases earlier - the distance that is safe for
// groups running one barrier apart with the fragment-read wait placed after the phase's first barrier.
// One wait per K-tile: a counted s_waitcnt vmcnt before the first barrier of phase 3 retires everything but the
// youngest unit(s) (= all of tile t+1); the first read of tile t+1 happens in the next phase, one barrier later.
// Epilogue: bias/GELU/SiLU/RoPE lane-local as in gemm.hip; bf16 results of two fragments are exchanged with
// v_permlane16_swap so every lane stores 16 contiguous bytes (half the store instructions of the 8-byte form).
// Results are bit-identical to gemm.hip (same MFMA, same k order): tests/test_gpu_kernels.py compares them.
#include <type_traits>
#include "common.h"
#include "nova_internal.h"

namespace nova {

constexpr int P_UNIT = 128 * 128;   // bytes of one staged unit (128 rows x 128 B)
constexpr int P_BUF = 4 * P_UNIT;   // A(0) A(1) W(0) W(1)
constexpr int P_LDS = 2 * P_BUF;    // 128 KiB

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

__device__ __forceinline__ f4v pmma(const PFrag<bf16_t>& w, const PFrag<bf16_t>& a, f4v c) { return Half16<bf16_t>::mfma16(w.v, a.v, c); }
__device__ __forceinline__ f4v pmma(const PFrag<f16_t>& w, const PFrag<f16_t>& a, f4v c) { return Half16<f16_t>::mfma16(w.v, a.v, c); }
__device__ __forceinline__ f4v pmma(const PFrag<float>& w, const PFrag<float>& a, f4v c) {
#pragma unroll
  for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v[j], a.v[j], c, 0, 0, 0);
  return c;
}

template <typename T>
__device__ __forceinline__ PFrag<T> punit_frag(const char* unit, int row, int chunk) {
  PFrag<T> f;
  f.v = *reinterpret_cast<const decltype(f.v)*>(unit + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
  return f;
}

#define NOVA_BARRIER() asm volatile("s_barrier" ::: "memory")
#define NOVA_LOOP_BARRIER() do { if (VAR < 12) NOVA_BARRIER(); } while (0)

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
  __shared__ __attribute__((aligned(16))) char smem[P_LDS];
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

  // ---- per-lane source pointers of the two LDS-DMA pieces this wave moves for each of the 4 units
  // unit order u: 0 = A(0), 1 = W(1), 2 = A(1), 3 = W(0)   (the staging order)
  const int cp = lane & 7;
  const size_t rowbytes = (size_t)K * sizeof(T);
  // wave-uniform tile bases + per-lane 32-bit byte offsets: the LDS-DMA instructions then take an SGPR base and a
  // 32-bit VGPR offset, and advancing along K is scalar arithmetic (no 64-bit vector add per issue)
  const char* a_base = reinterpret_cast<const char*>(A) + (size_t)m0 * rowbytes;
  const char* w_base = reinterpret_cast<const char*>(W) + (size_t)n0 * rowbytes;
  uint32_t soff[4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wid * 2 + i) * 8 + (lane >> 3);          // row inside the unit, 0..127
    const uint32_t c = (uint32_t)((cp ^ ((r >> 1) & 7)) << 4);  // source chunk (swizzle on the source side)
    const int a_lo = (r >> 6) * 128 + (r & 63);             // + mi*64
    const int w_lo = (r >> 5) * 64 + (r & 31);              // + ni*32
    const int rmax = M - 1 - m0;                            // rows past M re-read row M-1 (never stored)
    soff[0][i] = (uint32_t)min(a_lo, rmax) * (uint32_t)rowbytes + c;
    soff[2][i] = (uint32_t)min(a_lo + 64, rmax) * (uint32_t)rowbytes + c;
    soff[3][i] = (uint32_t)w_lo * (uint32_t)rowbytes + c;
    soff[1][i] = (uint32_t)(w_lo + 32) * (uint32_t)rowbytes + c;
  }
  const int nkt = K / (128 / (int)sizeof(T));
  // LDS offset of unit u inside a buffer: A(0) A(1) W(0) W(1)
  auto unit_off = [](int u) { return (u == 0 ? 0 : u == 2 ? 1 : u == 3 ? 2 : 3) * P_UNIT; };
  auto stage_piece = [&](int u, int kt, int i) {
    if (kt < nkt) {
      char* dst = smem + (kt & 1) * P_BUF + unit_off(u) + wid * 2048 + i * 1024;
      const char* base = ((u == 0 || u == 2) ? a_base : w_base) + (size_t)kt * 128;
      __builtin_amdgcn_global_load_lds(base + soff[u][i], NOVA_LDS_PTR(dst), 16, 0, 0);
    }
  };
  auto stage = [&](int u, int kt) {  // all 8 waves: 2 LDS-DMA instructions each (wave-uniform condition)
    if (VAR >= 10 && kt >= 2) return;  // ablation builds (timing only, wrong results): no prefetch inside the loop
    stage_piece(u, kt, 0);
    stage_piece(u, kt, 1);
  };

  const int fr = lane & 15, fg = lane >> 4;
  f4v bv[4];  // the accumulators start at the bias (see gemm.hip)
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) bv[nf] = f4v{0.f, 0.f, 0.f, 0.f};
  if (e.bias) {
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) bv[nf] = *reinterpret_cast<const f4v*>(e.bias + n0 + wc * 64 + nf * 16 + fg * 4);
  }
  f4v acc[4][8];  // [nf][mf]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = bv[i];

  // ---- prologue: q = 0..5 = tile 0 (A0 W1 A1 W0) + tile 1 (A0 W1); wait for tile 0
  stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
  stage(0, 1); stage(1, 1);
  if (nkt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  NOVA_BARRIER();
  if (wr == 1) NOVA_BARRIER();  // group 1 runs one barrier behind group 0 from here on

  PFrag<T> af[4][2], wf0[2][2], wf1[2][2];
  auto read_a = [&](const char* buf, int mi) {
    if (VAR >= 11 && buf != smem) return;
    const char* u = buf + mi * P_UNIT;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) af[f][kk] = punit_frag<T>(u, wr * 64 + f * 16 + fr, fg + 4 * kk);
  };
  auto read_w = [&](const char* buf, int ni, PFrag<T> (&wf)[2][2]) {
    if (VAR >= 11 && buf != smem) return;
    const char* u = buf + (2 + ni) * P_UNIT;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) wf[f][kk] = punit_frag<T>(u, wc * 32 + f * 16 + fr, fg + 4 * kk);
  };
  // 16 MFMAs of one quadrant; the LDS-DMA prefetch of this phase is issued between the two k halves, where
  // its issue cost hides under the matrix pipe instead of lengthening the other group's critical load segment.
  auto mma_quadrant = [&](int mi, int ni, PFrag<T> (&wf)[2][2], int su, int skt) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) acc[ni * 2 + nf][mi * 4 + mf] = pmma(wf[nf][kk], af[mf][kk], acc[ni * 2 + nf][mi * 4 + mf]);
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
    // phase 0: quadrant (0,0); stages q = 4kt+6 = (kt+1, A1)
    lstage_pre(2, kt + 1);
    read_a(buf, 0);
    read_w(buf, 0, wf0);
    lstage_post(2, kt + 1);
    NOVA_LOOP_BARRIER();
    mma_quadrant(0, 0, wf0, 2, kt + 1);
    NOVA_LOOP_BARRIER();
    // phase 1: quadrant (0,1); stages (kt+1, W0)
    lstage_pre(3, kt + 1);
    read_w(buf, 1, wf1);
    lstage_post(3, kt + 1);
    NOVA_LOOP_BARRIER();
    mma_quadrant(0, 1, wf1, 3, kt + 1);
    NOVA_LOOP_BARRIER();
    // phase 2: quadrant (1,1); stages (kt+2, A0)
    lstage_pre(0, kt + 2);
    read_a(buf, 1);
    lstage_post(0, kt + 2);
    NOVA_LOOP_BARRIER();
    mma_quadrant(1, 1, wf1, 0, kt + 2);
    NOVA_LOOP_BARRIER();
    // phase 3: quadrant (1,0) on the W(0) fragments still held from phase 0 (no LDS reads); stages (kt+2, W1).
    // Before the first barrier: retire all of tile kt+1 = everything but the youngest unit issued so far
    // ((kt+2, A0) of phase 2; this phase's unit is issued after the barrier).
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
    mma_quadrant(1, 0, wf0, 1, kt + 2);
    NOVA_LOOP_BARRIER();
  }
  if (wr == 0) NOVA_BARRIER();  // re-align the groups

  // ---- epilogue (same lane-local form as gemm.hip): lane holds out[m][n..n+3]
  const bool rot = EPI == E_ROPE && n0 < e.rope_cols;
#pragma unroll
  for (int mf = 0; mf < 8; ++mf) {
    const int m_raw = m0 + wr * 128 + mf * 16 + fr;
    if (__builtin_amdgcn_readfirstlane(m0 + wr * 128 + mf * 16) >= M) continue;  // whole fragment row block past M (wave-uniform)
    const int m = min(m_raw, M - 1);  // lanes past M recompute row M-1 and store the identical bytes (benign)
    f4v cs[4];
    if (rot) {
      const int s = m / e.L, l = m - s * e.L;
      const float* ropem = e.rope + ((size_t)(s % e.rope_batch) * e.L + l) * e.hd;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) cs[nf] = *reinterpret_cast<const f4v*>(ropem + (n0 + wc * 64 + nf * 16 + fg * 4) % e.hd);
    }
    T* dst = C + (size_t)m * N + n0 + wc * 64 + fg * 4;
    u2v pk[4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      f4v v = acc[nf][mf];
      if (EPI == E_GELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sizeof(T) == 2 ? v[j] : gelu_erf(v[j]);
        if (sizeof(T) == 2) v = gelu_erf_fast4(v);
      } else if (EPI == E_SILU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = silu(v[j]);
      } else if (EPI == E_ROPE) {
        if (rot) {
          v = rope_rotate4(v, cs[nf]);
        }
        if (n0 < e.q_cols) v = v * e.q_scale;
      }
      if constexpr (sizeof(T) == 2) {
        pk[nf] = u2v{Half16<T>::pack(v[0], v[1]), Half16<T>::pack(v[2], v[3])};
      } else {
        *reinterpret_cast<f4v*>(dst + nf * 16) = v;
      }
    }
    if constexpr (sizeof(T) == 2) {
      // widen the store: v_permlane16_swap exchanges the odd 16-lane rows of fragment a with the even rows of
      // fragment b, after which a lane holds 8 consecutive bf16 columns -> one 16-byte store per fragment pair
      // (lanes with fg even: fragment a, fg odd: fragment b; columns 8*(fg>>1) .. +7 of that fragment).
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const auto lo = __builtin_amdgcn_permlane16_swap(pk[2 * pr][0], pk[2 * pr + 1][0], false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(pk[2 * pr][1], pk[2 * pr + 1][1], false, false);
        u4v o = {lo[0], hi[0], lo[1], hi[1]};
        *reinterpret_cast<u4v*>(C + (size_t)m * N + n0 + wc * 64 + (2 * pr + (fg & 1)) * 16 + (fg >> 1) * 8) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Persistent form: one workgroup per CU walks its XCD's chunk of the tile list (same tile order as the
// one-tile-per-workgroup launch above). What it buys on the K = 1024 shapes, where a tile's main loop is only
// 16 K-tiles long: no workgroup dispatch between tiles, and the first six LDS-DMA units of the NEXT tile are
// issued in the middle of the epilogue (after the re-align barrier every LDS slot is dead), so their HBM/L2
// latency hides under the second half of the epilogue's VALU work and stores. vmcnt retires in issue order
// and counts stores, so the wait at the top of the next tile is vmcnt(4 + stores of the second half).
// Everything the second half needs from memory (RoPE table rows) is loaded BEFORE those DMAs are issued: a
// load issued after them could only be consumed once they have all landed.
template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                          typename OutOf<T>::type* __restrict__ C, int M, int N, int K,
                                                          int ntm, int ntn, GemmEpi256 e) {
  typedef typename OutOf<T>::type OT;
  constexpr bool FP8 = sizeof(T) == 1;
  constexpr bool Q8 = EPI == E_GELU_Q8;
  // vector-memory operations a wave issues per tile epilogue behind the next tile's prologue DMAs (exact: the wait at the next
  // tile top counts them): 16-byte stores of bf16 / f32 results; e4m3 results: 8 stores + 1 atomic max
  constexpr int STORES_TILE = Q8 ? 9 : (sizeof(OT) == 2 ? 16 : 32);
  __shared__ __attribute__((aligned(16))) char smem[P_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;

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
    const int lane = fresh_lane(), cp = lane & 7;
    a_base = reinterpret_cast<const char*>(A) + (size_t)m0 * rowbytes;
    w_base = reinterpret_cast<const char*>(W) + (size_t)n0 * rowbytes;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wid * 2 + i) * 8 + (lane >> 3);
      const uint32_t c = (uint32_t)((cp ^ ((r >> 1) & 7)) << 4);
      const int a_lo = (r >> 6) * 128 + (r & 63);
      const int w_lo = (r >> 5) * 64 + (r & 31);
      const int rmax = M - 1 - m0;
      soff[0][i] = (uint32_t)min(a_lo, rmax) * (uint32_t)rowbytes + c;
      soff[2][i] = (uint32_t)min(a_lo + 64, rmax) * (uint32_t)rowbytes + c;
      soff[3][i] = (uint32_t)w_lo * (uint32_t)rowbytes + c;
      soff[1][i] = (uint32_t)(w_lo + 32) * (uint32_t)rowbytes + c;
    }
  };
  const int nkt = K / (128 / (int)sizeof(T));
  auto unit_off = [](int u) { return (u == 0 ? 0 : u == 2 ? 1 : u == 3 ? 2 : 3) * P_UNIT; };
  auto stage = [&](int u, int kt) {
    if (kt < nkt) {
      char* dst = smem + (kt & 1) * P_BUF + unit_off(u) + wid * 2048;
      const char* base = ((u == 0 || u == 2) ? a_base : w_base) + (size_t)kt * 128;
      __builtin_amdgcn_global_load_lds(base + soff[u][0], NOVA_LDS_PTR(dst), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(base + soff[u][1], NOVA_LDS_PTR(dst + 1024), 16, 0, 0);
    }
  };
  auto stage_prologue = [&]() {
    stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
    stage(0, 1); stage(1, 1);
  };

  int fr, fg;
  f4v acc[4][8];  // [nf][mf]
  PFrag<T> af[4][2], wf0[2][2], wf1[2][2];
  auto read_a = [&](const char* buf, int mi) {
    const char* u = buf + mi * P_UNIT;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) af[f][kk] = punit_frag<T>(u, wr * 64 + f * 16 + fr, fg + 4 * kk);
  };
  auto read_w = [&](const char* buf, int ni, PFrag<T> (&wf)[2][2]) {
    const char* u = buf + (2 + ni) * P_UNIT;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) wf[f][kk] = punit_frag<T>(u, wc * 32 + f * 16 + fr, fg + 4 * kk);
  };
  // The accumulators start at the bias: the first MFMA of every accumulator (k half 0 of the tile's first K-tile) takes
  // the bias registers as its C operand, so there is neither a zeroing pass nor a bias add in the epilogue.
  f4v bv[4];
  auto mma_quadrant = [&](int mi, int ni, PFrag<T> (&wf)[2][2], auto first_ktile) {
    constexpr bool FIRST = decltype(first_ktile)::value;
    __builtin_amdgcn_s_setprio(1);
    if constexpr (FP8) {
      // MX-fp8: ONE v_mfma_scale_f32_16x16x128_f8f6f4 per accumulator and K-tile (twice the bf16 rate). A lane's 32 operand
      // bytes are the two 16-byte chunks (fg, fg + 4) it reads anyway - the same (lane, byte) -> k map on both operands,
      // which is all a contraction needs. Block scales are 1 (e8m0 127); the per-row scales are applied in the epilogue.
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
          const i8v wv = __builtin_bit_cast(i8v, __builtin_shufflevector(wf[nf][0].v, wf[nf][1].v, 0, 1, 2, 3, 4, 5, 6, 7));
          const i8v av = __builtin_bit_cast(i8v, __builtin_shufflevector(af[mf][0].v, af[mf][1].v, 0, 1, 2, 3, 4, 5, 6, 7));
          acc[ni * 2 + nf][mi * 4 + mf] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
              wv, av, FIRST ? bv[ni * 2 + nf] : acc[ni * 2 + nf][mi * 4 + mf], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
#pragma unroll
          for (int mf = 0; mf < 4; ++mf)
            acc[ni * 2 + nf][mi * 4 + mf] = pmma(wf[nf][kk], af[mf][kk], (FIRST && kk == 0) ? bv[ni * 2 + nf] : acc[ni * 2 + nf][mi * 4 + mf]);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto load_bias = [&](int n0) {
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) bv[nf] = f4v{0.f, 0.f, 0.f, 0.f};
    if (!FP8 && e.bias) {  // fp8: accumulators start at zero, the scales multiply the raw sums (bias added below)
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) bv[nf] = *reinterpret_cast<const f4v*>(e.bias + n0 + wc * 64 + nf * 16 + fg * 4);
    }
  };
  // one K-tile: 4 phases (see the header comment)
  auto ktile = [&](int kt, auto first_ktile) {
    const char* buf = smem + (kt & 1) * P_BUF;
    read_a(buf, 0);
    read_w(buf, 0, wf0);
    stage(2, kt + 1);
    NOVA_BARRIER();
    mma_quadrant(0, 0, wf0, first_ktile);
    NOVA_BARRIER();
    read_w(buf, 1, wf1);
    stage(3, kt + 1);
    NOVA_BARRIER();
    mma_quadrant(0, 1, wf1, first_ktile);
    NOVA_BARRIER();
    read_a(buf, 1);
    stage(0, kt + 2);
    NOVA_BARRIER();
    mma_quadrant(1, 1, wf1, first_ktile);
    NOVA_BARRIER();
    stage(1, kt + 2);
    if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NOVA_BARRIER();
    mma_quadrant(1, 0, wf0, first_ktile);
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
  { const int l = fresh_lane(); fr = l & 15; fg = l >> 4; }
  load_bias(n0);  // older than the DMAs: landed when the wait below returns (in-order retirement)
  stage_prologue();
  bool first = true;
  for (;;) {
    { const int l = fresh_lane(); fr = l & 15; fg = l >> 4; }
    // K-tile 0 of this tile landed: all but (tile 1: A0, W1) and, after the first tile, everything the previous
    // epilogue issued after the prologue DMAs (at least its STORES_TILE stores)
    if (nkt == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (first) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + STORES_TILE) : "memory");
    NOVA_BARRIER();
    if (wr == 1) NOVA_BARRIER();  // group 1 runs one barrier behind group 0 inside the K loop
    ktile(0, std::true_type{});
    for (int kt = 1; kt < nkt; ++kt) ktile(kt, std::false_type{});
    if (wr == 0) NOVA_BARRIER();  // re-align the groups: every LDS slot is dead from here on

    // ---- epilogue
    // (opaque copies: otherwise the row/table address arithmetic below is hoisted above the K loop and its
    // results are spilled around it)
    int cm0 = m0, cn0 = n0;
    asm volatile("" : "+s"(cm0), "+s"(cn0));
    { const int l = fresh_lane(); fr = l & 15; fg = l >> 4; }
    const bool rot = EPI == E_ROPE && cn0 < e.rope_cols;
    const float qmul = (EPI == E_ROPE && cn0 < e.q_cols) ? e.q_scale : 1.0f;  // tile-uniform; x * 1.0f is exact
    // four groups of 2 fragment row blocks. The next tile's prologue DMAs go out FIRST, ahead of this tile's stores in
    // the memory pipeline (issued behind them they queue for ~7k cycles). What the epilogue needs from memory before
    // it can start (the RoPE rows of groups 0-1) is therefore consumed before the DMAs are issued; the RoPE rows
    // of groups 2-3 are requested after them and arrive behind them (in-order retirement), by which time they are done.
    // fp8: out = acc * sa[row] * sw[col] + bias[col]; the three vectors are requested first and consumed before the DMAs
    f4v swv[4], bfv[4];
    float sav[8];
    if constexpr (FP8) {
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        swv[nf] = *reinterpret_cast<const f4v*>(e.sw + cn0 + wc * 64 + nf * 16 + fg * 4);
        bfv[nf] = e.bias ? *reinterpret_cast<const f4v*>(e.bias + cn0 + wc * 64 + nf * 16 + fg * 4) : f4v{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int mf = 0; mf < 8; ++mf) sav[mf] = e.sa[e.sa_scalar ? 0 : min(cm0 + wr * 128 + mf * 16 + fr, M - 1)];
      asm volatile("" ::"v"(sav[7]));  // youngest of them: the compiler's wait sits here (loads retire in order)
      // dequantise in place, ahead of everything else: the three scale / bias vectors (40 registers) are dead before the
      // epilogue proper starts to load its RoPE rows (kept live through it they cost the RoPE variant 97 spilled VGPRs)
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int mf = 0; mf < 8; ++mf) acc[nf][mf] = (acc[nf][mf] * sav[mf]) * swv[nf] + bfv[nf];
    }
#ifdef NOVA_ROPE_HALF_TABLE  // timing-only experiment (tools/ab_lib.py, never shipped): the cos / sin table read as f16 - half the bytes, half the registers
    typedef u2v CsT;
#else
    typedef f4v CsT;
#endif
    CsT cs[4][2][4];  // [group][row block][nf]
    int coff[4];      // column of this lane's 4 floats inside a table row (head-relative), per nf
    if (rot) {
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) coff[nf] = (cn0 + wc * 64 + nf * 16 + fg * 4) % e.hd;
    }
    const int nxt = it + nslot;
    const bool more = nxt < csize;
    auto load_cs = [&](int g) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // (sequence, position) of the row block's first row on wave-uniform values, the lane's row by increment
        const int mb = min(__builtin_amdgcn_readfirstlane(cm0 + wr * 128 + (2 * g + j) * 16), M - 1);
        const int s0 = mb / e.L, l0 = mb - s0 * e.L;
        const int b0 = s0 % e.rope_batch, b1 = b0 + 1 == e.rope_batch ? 0 : b0 + 1;
        int l = l0 + min(fr, M - 1 - mb);  // rows past M reuse row M-1 (never stored differently)
        int sb = b0;
        if (l >= e.L) { l -= e.L; sb = b1; }  // a 16-row block crosses at most one sequence boundary (L >= 16)
#ifdef NOVA_ROPE_HALF_TABLE
        const uint16_t* roph = reinterpret_cast<const uint16_t*>(e.rope) + ((size_t)sb * e.L + l) * e.hd;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) cs[g][j][nf] = *reinterpret_cast<const u2v*>(roph + coff[nf]);
#else
        const float* ropem = e.rope + ((size_t)sb * e.L + l) * e.hd;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) cs[g][j][nf] = *reinterpret_cast<const f4v*>(ropem + coff[nf]);
#endif
      }
    };
    float q8_inv = 1.0f, q8_max = 0.f;
    if constexpr (Q8) q8_inv = 1.0f / *e.q8_scale;
    auto finish_group = [&](int g, auto rotated) {
      constexpr bool ROT = decltype(rotated)::value;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int mf = 2 * g + j;
        // rows past M were staged as copies of row M-1, so their lanes hold row M-1's results and store the identical
        // bytes there again: no branch, and every tile issues the same number of stores (the vmcnt count above)
        const int m = min(cm0 + wr * 128 + mf * 16 + fr, M - 1);
        if constexpr (Q8) {
          uint32_t d[4];  // this lane's 4 columns of every 16-column fragment as 4 e4m3 bytes
#pragma unroll
          for (int nf = 0; nf < 4; ++nf) {
            f4v v = gelu_erf_fast4(acc[nf][mf]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              q8_max = fmaxf(q8_max, fabsf(v[q]));
              v[q] = __builtin_amdgcn_fmed3f(v[q] * q8_inv, -448.0f, 448.0f);
            }
            int w = 0;
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
            d[nf] = (uint32_t)w;
          }
          // 4 x 4 transpose of dwords between the wave's four 16-lane rows (fg) and the four fragments (nf): afterwards the
          // lane in row fg holds fragment nf = fg's columns 0..15 of its matrix row = 16 contiguous bytes, one 16-byte store
          const auto s01 = __builtin_amdgcn_permlane16_swap(d[0], d[1], false, false);
          const auto s23 = __builtin_amdgcn_permlane16_swap(d[2], d[3], false, false);
          const auto sac = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);
          const auto sbd = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);
          u4v o = {sac[0], sbd[0], sac[1], sbd[1]};
          *reinterpret_cast<u4v*>(reinterpret_cast<uint8_t*>(C) + (size_t)m * N + cn0 + wc * 64 + fg * 16) = o;
          continue;
        }
        u2v pk[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
          f4v v = acc[nf][mf];
          if (EPI == E_GELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = sizeof(OT) == 2 ? v[q] : gelu_erf(v[q]);
            if (sizeof(OT) == 2) v = gelu_erf_fast4(v);
          } else if (EPI == E_SILU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = silu(v[q]);
          } else if (EPI == E_ROPE) {
            if constexpr (ROT) {
#ifdef NOVA_ROPE_HALF_TABLE
              const f2v ta = Half16<f16_t>::unpack(cs[g][j][nf][0]), tb = Half16<f16_t>::unpack(cs[g][j][nf][1]);
              v = rope_rotate4(v, f4v{ta[0], ta[1], tb[0], tb[1]});
#else
              v = rope_rotate4(v, cs[g][j][nf]);
#endif
            }
            v = v * qmul;
          }
          if constexpr (sizeof(OT) == 2) {
            pk[nf] = u2v{Half16<OT>::pack(v[0], v[1]), Half16<OT>::pack(v[2], v[3])};
          } else {
            *reinterpret_cast<f4v*>(C + (size_t)m * N + cn0 + wc * 64 + fg * 4 + nf * 16) = v;
          }
        }
        if constexpr (sizeof(OT) == 2) {  // 16-byte stores through v_permlane16_swap, as in the kernel above
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const auto lo = __builtin_amdgcn_permlane16_swap(pk[2 * pr][0], pk[2 * pr + 1][0], false, false);
            const auto hi = __builtin_amdgcn_permlane16_swap(pk[2 * pr][1], pk[2 * pr + 1][1], false, false);
            u4v o = {lo[0], hi[0], lo[1], hi[1]};
            *reinterpret_cast<u4v*>(C + (size_t)m * N + cn0 + wc * 64 + (2 * pr + (fg & 1)) * 16 + (fg >> 1) * 8) = o;
          }
        }
      }
    };
    auto next_prologue = [&]() {
      if (more) {
        set_tile(m0, n0);
        stage_prologue();
      }
    };
    // the next tile's origin and its bias first: the bias loads are older than everything else this epilogue issues
    // and have landed when the wait at the next tile top returns
    if (more) {
      tile_origin(tile_at(nxt), m0, n0);
      if (!rot) load_bias(n0);  // (rotating tiles: after groups 0-1, when 64 accumulator registers are free)
    }
    // the rotation is tile-uniform (decided per 256-column tile): two straight-line epilogues behind one scalar branch
    if (rot) {
      load_cs(0);
      load_cs(1);
      asm volatile("" ::"v"(cs[1][1][3]));  // same for the table rows of groups 0-1 (loads retire in order)
      next_prologue();
      finish_group(0, std::true_type{});
      finish_group(1, std::true_type{});
      __builtin_amdgcn_sched_barrier(0);  // keep the later groups' table loads below groups 0-1 (register budget)
      if (more) load_bias(n0);
      load_cs(2);
      load_cs(3);
      finish_group(2, std::true_type{});
      finish_group(3, std::true_type{});
    } else {
      next_prologue();
      finish_group(0, std::false_type{});
      finish_group(1, std::false_type{});
      finish_group(2, std::false_type{});
      finish_group(3, std::false_type{});
    }
    if constexpr (Q8) {  // one atomic max per wave and tile (values are >= 0: float order == unsigned order of the bits)
      q8_max = wave_max(q8_max);
      if (fresh_lane() == 0) __hip_atomic_fetch_max(e.q8_amax, __float_as_uint(q8_max), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!more) break;
    first = false;
    it = nxt;
  }
}

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
  ProfScope prof(PROF_GEMM_NONE + epi, 2.0 * M * N * K, st);
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
  ProfScope prof(PROF_GEMM_NONE + epi, 2.0 * M * N * K, st);
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
  ProfScope prof(PROF_GEMM_NONE + (epi == E_GELU_Q8 ? E_GELU : epi), 2.0 * M * N * K, st);
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
static int launch256(const void* A, const void* W, void* C, int M, int N, int K, int epi, const GemmEpi256& e,
                     hipStream_t st, bool single) {
  // (the persistent RoPE epilogue steps through a 16-row block assuming it crosses at most one sequence boundary)
  if (!single && g_var256 == 20 && !(epi == E_ROPE && e.rope && e.L < 16)) return launch256p<T>(A, W, C, M, N, K, epi, e, st);
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
                   hipStream_t st, bool one_tile_per_workgroup) {
  GemmEpi256 e{bias, rope, L, rope_batch, hd, rope_cols, q_scale, q_cols, g_gm256, walk_is_reverse() ? 1 : 0, nullptr, nullptr, g_stagger256};
  return dispatch_dtype(dtype, [&](auto tag) { return launch256<decltype(tag)>(A, W, C, M, N, K, epi, e, st, one_tile_per_workgroup); });
}

}  // namespace nova
