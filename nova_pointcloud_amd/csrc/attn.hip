// NOVA hot path: non-causal multi-head self-attention over [prefix ; point tokens]
//   o = softmax(q k^T * scale) v        (reference diffnext/models/vision_transformer.py:63,
//   F.scaled_dot_product_attention with attn_mask=None; L <= Nv + N = 2560 at 2048 points)
//
// q/k/v are read IN PLACE from the fused-QKV GEMM output [S*L, 3D] (row strides passed in), so the
// reference's view/permute/unbind (vision_transformer.py:52-53) never materialises; o is written
// merged-head [S*L, D], which is what the out-projection GEMM consumes (vision_transformer.py:64).
//
// bf16 kernel (throughput mode): flash-style, one 256-thread workgroup = 128 query rows of one
// (sequence, head); each wave owns 32 query rows with Q resident in registers. K/V tiles of 64
// keys stream global->LDS by 16-byte LDS-DMA, double buffered, one barrier per tile.
//   S^T = K Q^T  with v_mfma_f32_32x32x16_bf16 (K fragment = A operand): every lane then holds
//   one query COLUMN, so the online-softmax max/sum are lane-local plus one xor-32 shuffle.
//   O^T = V^T P^T: the S^T accumulator, converted pairwise to bf16, IS the B operand of the next
//   MFMA (no LDS round trip); V^T fragments come from the row-major V tile through the hardware
//   transposing read ds_read_b64_tr_b16. Rescale factors and 1/l are lane-local as well.
// f32 kernel (parity mode): same skeleton on v_mfma_f32_32x32x2_f32 (exact f32), 32-key tiles.
// head_dim 64 (d48w768 / d48w1024) and 96 (d48w1536: the K/V tile is staged as a 64-wide image plus a 32-wide
// image so every LDS-DMA piece stays row-aligned and both images keep conflict-free read patterns).
#include "common.h"
#include "nova_internal.h"

namespace nova {

constexpr float NEG_INF = -__builtin_huge_valf();

// ------------------------------------------------------------------------------------------
// bf16, head_dim HD in {64, 96}
// ------------------------------------------------------------------------------------------
constexpr int A_KV = 64;                 // keys per tile
constexpr int A_T64 = A_KV * 128;        // 64-wide image: 128-byte rows, 8 KiB
constexpr int A_T32 = A_KV * 64;         // 32-wide image (HD = 96 only): 64-byte rows, 4 KiB

// q arrives scaled by (softmax scale * log2 e) - folded into the fused QKV GEMM epilogue by the block composite, or
// applied at load (c != 1) for generic callers - and the running max is carried as the C operand of the first
// QK^T MFMA (a 16-register block holding -m), so the softmax needs no multiply-subtract per score: p = exp2(acc).
// The block is rewritten only when the deferred-rescale branch fires.
template <typename E, int HD, bool LSE>  // E: bf16_t / f16_t; LSE: the training forward, which also writes the row log-sum-exp (attn_bwd.hip)
__global__ __launch_bounds__(256, HD == 64 ? 3 : 2) void attn_bf16(const E* __restrict__ q, const E* __restrict__ k,
                                                                    const E* __restrict__ v, E* __restrict__ o,
                                                                    int Lq, int Lk, long q_rs, long kv_rs, long o_rs, float c,
                                                                    int heads, int nq, int rev, long kv_ss, float* __restrict__ lse) {
  constexpr int NKS = HD / 16, NDV = HD / 32;
  constexpr int BUF = 2 * A_T64 + (HD == 96 ? 2 * A_T32 : 0);  // [K64 | V64 | K32 | V32]
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: LDS-DMA bases stay scalar
  const int r = lane & 31, hh = lane >> 5;
  // XCD-aware order: all query tiles of one (sequence, head) are consecutive in the remapped list, so they
  // run on one XCD and its K/V is served from that XCD's L2 after the first tile.
  const int t = xcd_remap_dir(blockIdx.x, gridDim.x, rev != 0);
  const int sh = t / nq, qt = t - sh * nq;
  const int head = sh % heads, s = sh / heads;
  const int q0 = qt * 128 + wid * 32;

  const E* qb = q + (size_t)s * Lq * q_rs + head * HD;
  const E* kb_ = k + (size_t)s * kv_ss + head * HD;
  const E* vb_ = v + (size_t)s * kv_ss + head * HD;

  // Q fragments: B operand of S^T = K Q^T; lane (r, hh) holds Q[q0 + r][16 ks + 8 hh + 0..7]
  u4v qf[NKS];
  {
    const int qrow = min(q0 + r, Lq - 1);
    const E* qp = qb + (size_t)qrow * q_rs + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const u4v*>(qp + 16 * ks);
    if (c != 1.0f) {  // q not pre-scaled by the producer (generic nova_attn_fwd callers)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f2v t = Half16<E>::unpack(qf[ks][j]);
          qf[ks][j] = Half16<E>::pack(t[0] * c, t[1] * c);
        }
    }
  }

  // staging: wave w moves LDS-DMA pieces 2w, 2w+1 (8 rows x 128 B) of the 64-wide K and V images and, for
  // HD = 96, piece w (16 rows x 64 B) of the 32-wide images. Per-lane byte offsets inside a tile are loop
  // invariant (32-bit); the tile base advances as a wave-uniform scalar, so a full tile costs no vector address
  // arithmetic. Only the ragged last tile recomputes clamped rows.
  const uint32_t rowB = (uint32_t)kv_rs * 2u;
  const int srow0 = (wid * 2) * 8 + (lane >> 3), srow1 = srow0 + 8, scp = lane & 7;
  const int srow32 = wid * 16 + (lane >> 2), scp32 = lane & 3;
  const uint32_t ck0 = (uint32_t)((scp ^ ((srow0 >> 1) & 7)) << 4), ck1 = (uint32_t)((scp ^ ((srow1 >> 1) & 7)) << 4);
  const uint32_t cv0 = (uint32_t)((scp ^ (((srow0 >> 1) & 1) << 2)) << 4), cv1 = (uint32_t)((scp ^ (((srow1 >> 1) & 1) << 2)) << 4);
  const uint32_t ck32 = 128u + (uint32_t)((scp32 ^ ((srow32 >> 2) & 3)) << 4), cv32 = 128u + (uint32_t)(scp32 << 4);
  const uint32_t ko0 = srow0 * rowB + ck0, ko1 = srow1 * rowB + ck1, vo0 = srow0 * rowB + cv0, vo1 = srow1 * rowB + cv1;
  const uint32_t ko32 = srow32 * rowB + ck32, vo32 = srow32 * rowB + cv32;
  auto stage = [&](int buf, int kt) {
    char* lk = smem + buf * BUF;
    char* lv = lk + A_T64;
    const char* kbase = reinterpret_cast<const char*>(kb_) + (size_t)kt * A_KV * rowB;  // wave-uniform
    const char* vbase = reinterpret_cast<const char*>(vb_) + (size_t)kt * A_KV * rowB;
    const int lim = Lk - 1 - kt * A_KV;  // last valid row of this tile
    if (lim >= A_KV - 1) {
      glds16(kbase, ko0, lk + wid * 2048);
      glds16(vbase, vo0, lv + wid * 2048);
      glds16(kbase, ko1, lk + wid * 2048 + 1024);
      glds16(vbase, vo1, lv + wid * 2048 + 1024);
      if constexpr (HD == 96) {
        glds16(kbase, ko32, lk + 2 * A_T64 + wid * 1024);
        glds16(vbase, vo32, lk + 2 * A_T64 + A_T32 + wid * 1024);
      }
    } else {  // ragged tile: rows past Lk re-read the last valid row (their scores are masked to -inf)
      const uint32_t r0 = (uint32_t)min(srow0, lim) * rowB, r1 = (uint32_t)min(srow1, lim) * rowB;
      glds16(kbase, r0 + ck0, lk + wid * 2048);
      glds16(vbase, r0 + cv0, lv + wid * 2048);
      glds16(kbase, r1 + ck1, lk + wid * 2048 + 1024);
      glds16(vbase, r1 + cv1, lv + wid * 2048 + 1024);
      if constexpr (HD == 96) {
        const uint32_t r32 = (uint32_t)min(srow32, lim) * rowB;
        glds16(kbase, r32 + ck32, lk + 2 * A_T64 + wid * 1024);
        glds16(vbase, r32 + cv32, lk + 2 * A_T64 + A_T32 + wid * 1024);
      }
    }
  };

  f16v ot[NDV];
#pragma unroll
  for (int d = 0; d < NDV; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) ot[d][i] = 0.f;
  float m_run = 0.f, l_run = 0.f;
  f16v negm;  // -m_run replicated: the accumulator input of every tile's first MFMA
#pragma unroll
  for (int i = 0; i < 16; ++i) negm[i] = 0.f;

  // per-lane constants of the transposing V read: lane 4*qr + p of each 16-lane group supplies
  // the address of row key0 + qr, columns dv0 + 4p .. 4p+3
  const int t_qr = (lane & 15) >> 2, t_p = lane & 3, t_gp = (lane >> 4) & 1;

  const int nkt = (Lk + A_KV - 1) / A_KV;
  stage(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile kt (asm LDS-DMA: not counted by the compiler)
    __syncthreads();
    if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
    const char* tk = smem + (kt & 1) * BUF;
    const char* tv = tk + A_T64;
    const char* tk32 = tk + 2 * A_T64;
    const char* tv32 = tk32 + A_T32;

    // ---- S^T[key][q] - m for the two 32-key blocks
    // (A/B'd in round 2, bit-identical and time-neutral: chain heads as inline asm with an early-clobber result, which
    // removes the 8 v_mov_b64 copies of the -m block the register allocator inserts for one chain, and a written-out
    // v_max3 chain without the canonicalising v_max - the loop is not bound by the count of such cheap instructions.)
    f16v st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int row = kb * 32 + r;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        u4v kf;
        if (ks < 4) kf = *reinterpret_cast<const u4v*>(tk + row * 128 + (((2 * ks + hh) ^ ((row >> 1) & 7)) << 4));
        else kf = *reinterpret_cast<const u4v*>(tk32 + row * 64 + (((2 * (ks - 4) + hh) ^ ((row >> 2) & 3)) << 4));
        st[kb] = Half16<E>::mfma32(kf, qf[ks], ks == 0 ? negm : st[kb]);
      }
    }
    if (kt == nkt - 1 && (Lk & (A_KV - 1)) != 0) {  // ragged last tile: keys >= Lk contribute nothing
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = kt * A_KV + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (key >= Lk) st[kb][i] = NEG_INF;
        }
    }

    // ---- online softmax, lane-local per query column (partner lane^32 holds the other keys)
    float mx = st[0][0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, st[0][i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, st[1][i]);
    mx = max_xor32(mx);
    // st already holds s*c - m_run (c folded into q, -m_run carried in by the first MFMA): mx is the growth of the
    // running max in the exp2 domain. Deferred rescale: O, l and the carried max only move when some query of the
    // wave grew by more than 2^8 (always on the first tile); until then p <= 2^8, exact in the f32 accumulators and
    // with unchanged relative precision in bf16. The decision is wave-uniform; both lanes of a query agree on m.
    if (kt == 0 || __any(mx > 8.0f)) {
      const float delta = kt == 0 ? mx : fmaxf(mx, 0.f);
      if (kt != 0) {
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < NDV; ++d)
#pragma unroll
          for (int i = 0; i < 16; ++i) ot[d][i] *= alpha;
      }
      m_run += delta;
#pragma unroll
      for (int i = 0; i < 16; ++i) { st[0][i] -= delta; st[1][i] -= delta; negm[i] = -m_run; }
    }
    float psum = 0.f;
    u4v pb[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        u4v packed;
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // two probabilities -> one v_cvt_pk_bf16_f32
          const float p0 = __builtin_amdgcn_exp2f(st[kb][8 * s2 + 2 * j]);
          const float p1 = __builtin_amdgcn_exp2f(st[kb][8 * s2 + 2 * j + 1]);
          psum += p0 + p1;
          packed[j] = Half16<E>::pack(p0, p1);
        }
        pb[kb][s2] = packed;
      }
    l_run += psum;

    // ---- O^T[dv][q] += V^T[dv][key] P^T[key][q]
#pragma unroll
    for (int dvb = 0; dvb < NDV; ++dvb) {
      const int col = (dvb & 1) * 32 + 16 * t_gp + 4 * t_p;  // column inside its image (64-wide: dvb 0,1; 32-wide: dvb 2)
      const int chunk = col >> 3, within = (t_p & 1) * 8;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int row0 = kb * 32 + 16 * s2 + 4 * hh + t_qr;
          const int row1 = row0 + 8;
          const char *a0, *a1;
          if (dvb < 2) {
            a0 = tv + row0 * 128 + ((chunk ^ (((row0 >> 1) & 1) << 2)) << 4) + within;
            a1 = tv + row1 * 128 + ((chunk ^ (((row1 >> 1) & 1) << 2)) << 4) + within;
          } else {  // 4 consecutive 64-byte rows per 32-lane half = one 256-byte bank row: conflict-free as is
            a0 = tv32 + row0 * 64 + (chunk << 4) + within;
            a1 = tv32 + row1 * 64 + (chunk << 4) + within;
          }
          const bf4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf4v*)a0);
          const bf4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf4v*)a1);
          const u4v vf = __builtin_bit_cast(u4v, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          ot[dvb] = Half16<E>::mfma32(vf, pb[kb][s2], ot[dvb]);
        }
    }
  }

  // ---- finalize: lane (r, hh) holds O[q0 + r][dvb*32 + (i&3) + 8(i>>2) + 4hh]
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + r;
  if (qrow < Lq) {
    // training: log2-domain log-sum-exp of the row's scaled scores, what the backward kernels rebuild P from (attn_bwd.hip)
    if (LSE && hh == 0) lse[((size_t)s * heads + head) * Lq + qrow] = m_run + __log2f(l_tot);
    E* op = o + ((size_t)s * Lq + qrow) * o_rs + head * HD;
#pragma unroll
    for (int dvb = 0; dvb < NDV; ++dvb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dv = dvb * 32 + 8 * g + 4 * hh;
        u2v pk = {Half16<E>::pack(ot[dvb][4 * g] * inv, ot[dvb][4 * g + 1] * inv),
                  Half16<E>::pack(ot[dvb][4 * g + 2] * inv, ot[dvb][4 * g + 3] * inv)};
        *reinterpret_cast<u2v*>(op + dv) = pk;
      }
  }
}

// ------------------------------------------------------------------------------------------
// f32 (parity mode), head_dim HD in {64, 96}, 32-key tiles, exact-f32 MFMA
// ------------------------------------------------------------------------------------------
constexpr int F_KV = 32;

template <int HD>
__global__ __launch_bounds__(256) void attn_f32(const float* __restrict__ q, const float* __restrict__ k,
                                                const float* __restrict__ v, float* __restrict__ o, int Lq, int Lk,
                                                long q_rs, long kv_rs, long o_rs, float c, long kv_ss) {
  constexpr int ROWB = HD * 4, TILE = F_KV * ROWB, NCS = HD / 8, NDV = HD / 32, PPW = TILE / 4096;
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int head = blockIdx.y, s = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wid * 32;

  const float* qb = q + (size_t)s * Lq * q_rs + head * HD;
  const float* kb_ = k + (size_t)s * kv_ss + head * HD;
  const float* vb_ = v + (size_t)s * kv_ss + head * HD;

  // lane (r, hh) holds Q[q0 + r][4 (2 cs + hh) + j]; MFMA step (cs, j) contracts the k pair
  // {4(2cs)+j, 4(2cs+1)+j} (any consistent order is a valid contraction)
  f4v qf[NCS];
  {
    const int qrow = min(q0 + r, Lq - 1);
    const float* qp = qb + (size_t)qrow * q_rs + 4 * hh;
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs) qf[cs] = *reinterpret_cast<const f4v*>(qp + 8 * cs);
  }

  // K image: 16-byte chunk XOR-swizzled by row for HD = 64 (256-byte rows = one bank row); HD = 96 is left
  // unswizzled (bank conflicts only cost time, and this is the parity path)
  auto stage = [&](int buf, int kt) {
    char* lk = smem + buf * 2 * TILE;
    char* lv = lk + TILE;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wid * PPW + i;
      const int lin = piece * 64 + lane;           // 16-byte chunk index inside the tile (lane-linear LDS image)
      const int row = lin / (ROWB / 16), cp = lin % (ROWB / 16);
      const int key = min(kt * F_KV + row, Lk - 1);
      const int ck = HD == 64 ? (cp ^ (row & 15)) : cp;
      __builtin_amdgcn_global_load_lds(kb_ + (size_t)key * kv_rs + ck * 4, NOVA_LDS_PTR(lk + piece * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(vb_ + (size_t)key * kv_rs + cp * 4, NOVA_LDS_PTR(lv + piece * 1024), 16, 0, 0);
    }
  };

  f16v ot[NDV];
#pragma unroll
  for (int d = 0; d < NDV; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) ot[d][i] = 0.f;
  float m_run = NEG_INF, l_run = 0.f;

  const int nkt = (Lk + F_KV - 1) / F_KV;
  stage(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // explicit: do not rely on the compiler draining LDS-DMA before a barrier
    __syncthreads();
    if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
    const char* tk = smem + (kt & 1) * 2 * TILE;
    const float* tv = reinterpret_cast<const float*>(tk + TILE);

    f16v st;
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs) {
      const int phys = HD == 64 ? ((2 * cs + hh) ^ (r & 15)) : (2 * cs + hh);
      const f4v kf = *reinterpret_cast<const f4v*>(tk + r * ROWB + phys * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[cs][j], st, 0, 0, 0);
    }
    if (kt == nkt - 1 && (Lk & (F_KV - 1)) != 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = kt * F_KV + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (key >= Lk) st[i] = NEG_INF;
      }
    }

    float mx = st[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, st[i]);
    mx = max_xor32(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f((m_run - m_new) * c);
    const float mc = m_new * c;
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      st[i] = exp2f(st[i] * c - mc);
      psum += st[i];
    }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int d = 0; d < NDV; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) ot[d][i] *= alpha;

#pragma unroll
    for (int dvb = 0; dvb < NDV; ++dvb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = (i & 3) + 8 * (i >> 2) + 4 * hh;
        const float vv = tv[key * HD + dvb * 32 + r];
        ot[dvb] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, st[i], ot[dvb], 0, 0, 0);
      }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + r;
  if (qrow < Lq) {
    float* op = o + ((size_t)s * Lq + qrow) * o_rs + head * HD;
#pragma unroll
    for (int dvb = 0; dvb < NDV; ++dvb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dv = dvb * 32 + 8 * g + 4 * hh;
        f4v ov = {ot[dvb][4 * g] * inv, ot[dvb][4 * g + 1] * inv, ot[dvb][4 * g + 2] * inv, ot[dvb][4 * g + 3] * inv};
        *reinterpret_cast<f4v*>(op + dv) = ov;
      }
  }
}

// which bf16 / head_dim 64 structure nova_attn_fwd launches (per calling thread): 0 = 32x32x16 (attn_bf16 above),
// 1 / 2 = 16x16x32 with 32 / 64 query rows per wave (attn16.hip), 3 / 4 = the same with the row sums on the matrix pipe, 5 = 4 software-pipelined (P V of tile t-1 beside the exponentials of tile t); -1 = the shipped choice
static thread_local int g_attn_variant = -1;
int attn_set_variant(int v) {
  if (v < -1 || v > 5) return -1;
  g_attn_variant = v;
  return 0;
}
int attn_variant() { return g_attn_variant < 0 ? NOVA_ATTN_DEFAULT_VARIANT : g_attn_variant; }

int attn_fwd(const void* q, const void* k, const void* v, void* o, int S, int heads, int Lq, int Lk, int hd,
             long q_rs, long kv_rs, long o_rs, float scale, int dtype, hipStream_t st, bool q_prescaled, long kv_ss, float* lse,
             const int* klim) {
  if (S <= 0 || Lq <= 0) return 0;
  if (hd != 64 && hd != 96) return set_error(NOVA_ERR_SHAPE, "attn_fwd: head_dim %d not built (have 64 and 96)", hd);
  if (Lk <= 0 || heads <= 0) return set_error(NOVA_ERR_SHAPE, "attn_fwd: bad Lk/heads");
  if (klim && (!lse || !dtype_is16(dtype) || Lq != Lk)) return set_error(NOVA_ERR_ARG, "attn_fwd: a key-limit mask comes with the 16-bit training forward (Lq == Lk, log-sum-exp output)");
  if (lse && !dtype_is16(dtype)) return set_error(NOVA_ERR_ARG, "attn_fwd: the log-sum-exp output is built for the 16-bit kernels");
  const int align = dtype_is16(dtype) ? 8 : 4;  // 16-byte row alignment for the vector loads
  if (q_rs % align || kv_rs % align || o_rs % align) return set_error(NOVA_ERR_SHAPE, "attn_fwd: row strides must be 16-byte multiples");
  if (kv_ss == 0) kv_ss = (long)Lk * kv_rs;
  if (kv_ss % align || kv_ss < (long)Lk * kv_rs) return set_error(NOVA_ERR_SHAPE, "attn_fwd: kv sequence stride must cover Lk rows and be a 16-byte multiple");
  const int nq = (Lq + 127) / 128;
  if (S > 65535 || heads > 65535 || (long)nq * heads * S > 0x7fffffffL) return set_error(NOVA_ERR_SHAPE, "attn_fwd: grid too large");
  const float c = scale * 1.4426950408889634f;
  dim3 grid(nq, heads, S), block(256), grid1((unsigned)((long)nq * heads * S));
  ProfScope prof(PROF_ATTN, 4.0 * S * heads * (double)Lq * Lk * hd, st);
  if (dtype_is16(dtype)) {
    const float cl = q_prescaled ? 1.0f : c;
    const int rev = walk_is_reverse() ? 1 : 0;
    if (klim) return attn_fwd_m16(q, k, v, o, S, heads, Lq, Lk, hd, q_rs, kv_rs, o_rs, cl, dtype, st, kv_ss, lse, 32, true, false, klim);
    if (attn_variant() != 0) {
      // the shipped choice: 16x16x32, 32 query rows per wave, row sums on the matrix pipe, at every length. (64 rows per wave where
      // 256-row workgroups tile the queries exactly is +1.3 .. 2 % standalone at L = 2560 - profiles/r03_attn_variants_ab.txt - but
      // measured ~5 % SLOWER inside the generation step, where two lanes share the chip: attention / plain-GEMM rate 0.78 against 0.82
      // over four bench runs; it also re-fetches 7 % more K / V, L2 hit 83 % against 91 %.)
      const int av = attn_variant();
      return attn_fwd_m16(q, k, v, o, S, heads, Lq, Lk, hd, q_rs, kv_rs, o_rs, cl, dtype, st, kv_ss, lse, (av & 1) && av != 5 ? 32 : 64, av >= 3,
                          av == 5);
    }
    dispatch_half(dtype, [&](auto tag) {
      using E = decltype(tag);
      const E *qq = (const E*)q, *kk = (const E*)k, *vv = (const E*)v;
      if (lse && hd == 64) hipLaunchKernelGGL((attn_bf16<E, 64, true>), grid1, block, 0, st, qq, kk, vv, (E*)o, Lq, Lk, q_rs, kv_rs, o_rs, cl, heads, nq, rev, kv_ss, lse);
      else if (lse) hipLaunchKernelGGL((attn_bf16<E, 96, true>), grid1, block, 0, st, qq, kk, vv, (E*)o, Lq, Lk, q_rs, kv_rs, o_rs, cl, heads, nq, rev, kv_ss, lse);
      else if (hd == 64) hipLaunchKernelGGL((attn_bf16<E, 64, false>), grid1, block, 0, st, qq, kk, vv, (E*)o, Lq, Lk, q_rs, kv_rs, o_rs, cl, heads, nq, rev, kv_ss, lse);
      else hipLaunchKernelGGL((attn_bf16<E, 96, false>), grid1, block, 0, st, qq, kk, vv, (E*)o, Lq, Lk, q_rs, kv_rs, o_rs, cl, heads, nq, rev, kv_ss, lse);
      return 0;
    });
  } else {
    const float *qq = (const float*)q, *kk = (const float*)k, *vv = (const float*)v;
    if (hd == 64) hipLaunchKernelGGL(attn_f32<64>, grid, block, 0, st, qq, kk, vv, (float*)o, Lq, Lk, q_rs, kv_rs, o_rs, c, kv_ss);
    else hipLaunchKernelGGL(attn_f32<96>, grid, block, 0, st, qq, kk, vv, (float*)o, Lq, Lk, q_rs, kv_rs, o_rs, c, kv_ss);
  }
  return check_launch("attn_fwd");
}

}  // namespace nova
