// NOVA hot path: the bf16 / head_dim 64 self-attention of attn.hip rebuilt on v_mfma_f32_16x16x32_bf16
//   o = softmax(q k^T * scale) v        (reference diffnext/models/vision_transformer.py:63)
//
// Why a second shape: the four MFMA kernels of the generation path run power-limited; on random
// data the chip holds a higher clock on the 16x16x32 shape than on 32x32x16 at equal cycles per FLOP
// (MI355X_MICROARCH 'DVFS give-back' item 7; tools/mfma_issue_probe.hip at this loop's VALU density: +6 % FLOP/s).
// The 16x16 output block also makes the work per wave a free parameter: NQB 16-query blocks per wave share every K and
// V fragment read from LDS, so NQB = 4 (64 query rows per wave, 256 per workgroup) halves the LDS read bytes per FLOP
// of the NQB = 2 form (32 rows per wave, 128 per workgroup - the geometry of attn.hip).
//
// Same algorithm and semantics as attn_bf16 (attn.hip): S^T = K Q^T so that a query's scores sit on lanes {i, i+16,
// i+32, i+48} x 4 registers per 16-key block; q arrives pre-scaled by scale * log2 e and the running max rides in as the
// C operand of the first MFMA of a score chain (p = exp2(acc)); deferred rescale (only when some query of the wave grew
// by more than 2^8 - and only then is the max reduced across the four lanes of a query: the decision itself needs no
// cross-lane step); the bf16-packed S^T block pair IS the B operand of O^T = V^T P^T (k order inside the 32-key
// contraction: element j < 4 from block 2kp, j >= 4 from block 2kp + 1, and the transposing V read fetches exactly
// those rows); K/V 64-key tiles by LDS-DMA, double buffered, one barrier per tile.
// LDS images (128-byte rows, 16-byte chunks): K chunk ^ ((row >> 1) & 7) - conflict-free for the 16-row ds_read_b128
// fragment read; V chunk ^ (((row >> 1) & 3) << 1) - conflict-free for ds_read_b64_tr_b16 over 8 consecutive rows x 32 B.
#include <type_traits>
#include "common.h"
#include "nova_internal.h"

namespace nova {

namespace {
constexpr float NEG_INF16 = -__builtin_huge_valf();
constexpr int B_KV = 64;           // keys per tile
constexpr int B_T = B_KV * 128;    // one image: 8 KiB

// This file is compiled with -fno-honor-nans (Makefile): in IEEE mode hipcc canonicalises every MFMA result
// before an fmaxf (one extra v_max per score, 16 of the loop's ~90 vector instructions per tile); without it plain
// fmaxf(fmaxf(a, b), c) becomes one v_max3_f32 and the scheduler is free to interleave the chains. The loop produces no
// NaN (masked scores are -inf, and -inf only ever meets finite values).
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// max of the 16 scores a lane holds for one 16-query block: a depth-3 tree of 8 instructions
__device__ __forceinline__ float max16(const f4v (&s)[4]) {
  const float t0 = max3(s[0][0], s[0][1], s[0][2]), t1 = max3(s[0][3], s[1][0], s[1][1]), t2 = max3(s[1][2], s[1][3], s[2][0]);
  const float t3 = max3(s[2][1], s[2][2], s[2][3]), t4 = max3(s[3][0], s[3][1], s[3][2]);
  return fmaxf(max3(t0, t1, t2), max3(t3, t4, s[3][3]));
}
__device__ __forceinline__ float max_over_g(float x) {  // over lanes {i, i+16, i+32, i+48}
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float y = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const uint32_t w = __float_as_uint(y);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float sum_over_g(float x) {
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const uint32_t w = __float_as_uint(y);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
}  // namespace

// HD = 96 (d48w1536): a 64-wide image plus a 32-wide image (64-byte rows) per K / V tile, as attn.hip, so every LDS-DMA piece
// stays row aligned; their swizzles: K32 chunk ^ s((row >> 2) & 3), s = (0, 2, 3, 1); V32 chunk ^ (((row >> 2) & 1) << 1).
// MASK (the training forward of multi-frame models): a per-query key limit klim[Lq] - query i attends to keys [0, klim[i]), klim
// non-decreasing and >= 1 - which is what the reference's block-causal frame mask is (embeddings.py:247-260: token i sees the tokens
// of frames <= its own; the prefix counts as frame 0). Tiles are then walked in natural order (tile 0 holds an allowed key for
// every row, so the first step's max is finite), every tile applies the limit, and tiles past the workgroup's largest limit are
// never staged.
#ifdef NOVA_CLOCK
// Diagnostic build only (tools/kernel_clock.py): shader cycles and 100 MHz ticks between a workgroup's start and end, kept for the
// last workgroup that used each of 1024 slots (MI355X_MICROARCH.md, 'DVFS give-back' item 6)
__device__ long long g_attn_clock[2 * 1024];
#endif
template <typename E, int HD, int NQB, bool LSE, bool SUMM, bool MASK = false>  // E: bf16_t / f16_t (same loads, LDS images and stores; MFMA form and pair packing differ)
__global__ __launch_bounds__(256, (NQB == 2 && HD == 64) ? 3 : 2) void attn_bf16_m16(
    const E* __restrict__ q, const E* __restrict__ k,
                                                                         const E* __restrict__ v, E* __restrict__ o,
                                                                         int Lq, int Lk, long q_rs, long kv_rs, long o_rs, float c,
                                                                         int heads, int nq, int rev, long kv_ss, float* __restrict__ lse,
                                                                         const int* __restrict__ klim = nullptr) {
  constexpr int NDS = HD / 32, NDVB = HD / 16, RW = 16 * NQB;
  constexpr int B_T32 = B_KV * 64;                                  // 32-wide image (HD = 96): 4 KiB
  constexpr int BUF = 2 * B_T + (HD == 96 ? 2 * B_T32 : 0);         // [K64 | V64 | K32 | V32]
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef NOVA_CLOCK
  const long long ck_t0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int t = xcd_remap_dir(blockIdx.x, gridDim.x, rev != 0);
  const int sh = t / nq, qt = t - sh * nq;
  const int head = sh % heads, s = sh / heads;
  const int q0 = qt * (4 * RW) + wid * RW;

  const E* qb_ = q + (size_t)s * Lq * q_rs + head * HD;
  const E* kb_ = k + (size_t)s * kv_ss + head * HD;
  const E* vb_ = v + (size_t)s * kv_ss + head * HD;

  // Q fragments, B operand of S^T = K Q^T: lane (i, g) holds Q[q0 + 16 qb + i][32 ds + 8 g + 0..7]
  u4v qf[NQB][NDS];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const int qrow = min(q0 + 16 * qb + i, Lq - 1);
    const E* qp = qb_ + (size_t)qrow * q_rs + 8 * g;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) qf[qb][ds] = *reinterpret_cast<const u4v*>(qp + 32 * ds);
    if (c != 1.0f) {
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f2v t = Half16<E>::unpack(qf[qb][ds][j]);
          qf[qb][ds][j] = Half16<E>::pack(t[0] * c, t[1] * c);
        }
    }
  }

  // staging (as attn.hip): wave w moves LDS-DMA pieces 2w, 2w+1 (8 rows x 128 B each) of the K and the V image
  const uint32_t rowB = (uint32_t)kv_rs * 2u;
  const int srow0 = (wid * 2) * 8 + (lane >> 3), srow1 = srow0 + 8, scp = lane & 7;
  const uint32_t ck0 = (uint32_t)((scp ^ ((srow0 >> 1) & 7)) << 4), ck1 = (uint32_t)((scp ^ ((srow1 >> 1) & 7)) << 4);
  const uint32_t cv0 = (uint32_t)((scp ^ (((srow0 >> 1) & 3) << 1)) << 4), cv1 = (uint32_t)((scp ^ (((srow1 >> 1) & 3) << 1)) << 4);
  const uint32_t ko0 = srow0 * rowB + ck0, ko1 = srow1 * rowB + ck1, vo0 = srow0 * rowB + cv0, vo1 = srow1 * rowB + cv1;
  // 32-wide images: wave w moves piece w (16 rows x 64 B) of each; source columns 64 .. 95 = byte 128 + 16 * chunk
  const int srow32 = wid * 16 + (lane >> 2), scp32 = lane & 3;
  const uint32_t ck32 = 128u + (uint32_t)((scp32 ^ ((0x78 >> (2 * ((srow32 >> 2) & 3))) & 3)) << 4);
  const uint32_t cv32 = 128u + (uint32_t)((scp32 ^ (((srow32 >> 2) & 1) << 1)) << 4);
  auto stage = [&](int buf, int kt) {
    char* lk = smem + buf * BUF;
    char* lv = lk + B_T;
    const char* kbase = reinterpret_cast<const char*>(kb_) + (size_t)kt * B_KV * rowB;  // wave-uniform
    const char* vbase = reinterpret_cast<const char*>(vb_) + (size_t)kt * B_KV * rowB;
    const int lim = Lk - 1 - kt * B_KV;
    if (lim >= B_KV - 1) {
      glds16(kbase, ko0, lk + wid * 2048);
      glds16(vbase, vo0, lv + wid * 2048);
      glds16(kbase, ko1, lk + wid * 2048 + 1024);
      glds16(vbase, vo1, lv + wid * 2048 + 1024);
      if constexpr (HD == 96) {
        glds16(kbase, srow32 * rowB + ck32, lk + 2 * B_T + wid * 1024);
        glds16(vbase, srow32 * rowB + cv32, lk + 2 * B_T + B_T32 + wid * 1024);
      }
    } else {  // ragged tile: rows past Lk re-read the last valid row (their scores are masked to -inf)
      const uint32_t r0 = (uint32_t)min(srow0, lim) * rowB, r1 = (uint32_t)min(srow1, lim) * rowB;
      glds16(kbase, r0 + ck0, lk + wid * 2048);
      glds16(vbase, r0 + cv0, lv + wid * 2048);
      glds16(kbase, r1 + ck1, lk + wid * 2048 + 1024);
      glds16(vbase, r1 + cv1, lv + wid * 2048 + 1024);
      if constexpr (HD == 96) {
        const uint32_t r32 = (uint32_t)min(srow32, lim) * rowB;
        glds16(kbase, r32 + ck32, lk + 2 * B_T + wid * 1024);
        glds16(vbase, r32 + cv32, lk + 2 * B_T + B_T32 + wid * 1024);
      }
    }
  };

  f4v ot[NQB][NDVB], negm[NQB], lacc[NQB];  // lacc (SUMM): row sums of the bf16-rounded P from an all-ones V^T block
  float m_run[NQB], l_run[NQB];
  const u4v ones = {Half16<E>::ONE2, Half16<E>::ONE2, Half16<E>::ONE2, Half16<E>::ONE2};
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    m_run[qb] = 0.f;
    l_run[qb] = 0.f;
    negm[qb] = f4v{0.f, 0.f, 0.f, 0.f};
    lacc[qb] = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < NDVB; ++d) ot[qb][d] = f4v{0.f, 0.f, 0.f, 0.f};
  }

  // lane constants of the two read kinds
  const uint32_t kx = (uint32_t)((i >> 1) & 7);                                    // K swizzle of rows 16 kb + i (independent of kb)
  const uint32_t koff0 = (uint32_t)i * 128u + (((uint32_t)g ^ kx) << 4);           // d-step 0; d-step 1 flips chunk bit 2
  const uint32_t koff1 = (uint32_t)i * 128u + ((((uint32_t)g + 4u) ^ kx) << 4);
  const int t_q = (lane & 15) >> 2, t_p = lane & 3;
  const uint32_t vrow = (uint32_t)(4 * g + t_q);                                   // + 32 kp (+ 16 for the high half): swizzle unchanged
  const uint32_t vx = ((vrow >> 1) & 3u) << 1;
  uint32_t voff[NDVB];  // dvb < 4: inside the 64-wide V image; dvb 4, 5: inside the 32-wide one (64-byte rows)
#pragma unroll
  for (int dvb = 0; dvb < NDVB; ++dvb)
    voff[dvb] = dvb < 4 ? vrow * 128u + ((((uint32_t)(2 * dvb + (t_p >> 1))) ^ vx) << 4) + 8u * (t_p & 1)
                        : vrow * 64u + ((((uint32_t)(2 * (dvb - 4) + (t_p >> 1))) ^ (((vrow >> 2) & 1u) << 1)) << 4) + 8u * (t_p & 1);
  const uint32_t koff2 = (uint32_t)i * 64u + (((uint32_t)g ^ (uint32_t)((0x78 >> (2 * ((i >> 2) & 3))) & 3)) << 4);  // d-step 2: the 32-wide K image

  int kl[NQB];  // MASK: this lane's key limit per query block
  int k_end = Lk;
  if constexpr (MASK) {
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) kl[qb] = min(klim[min(q0 + 16 * qb + i, Lq - 1)], Lk);
    k_end = min(klim[min(qt * (4 * RW) + 4 * RW - 1, Lq - 1)], Lk);  // the workgroup's last row has its largest limit
  }
  const int nkt = (k_end + B_KV - 1) / B_KV;
  // Tile order: the (possibly ragged) LAST tile goes first, as a peeled step (softmax does not care about key order), so the
  // steady-state loop carries neither masking code nor the first-tile case, and the compiler is free to run the lane-local
  // max chains under the score MFMAs: step 0 <-> tile nkt - 1, step j >= 1 <-> tile j - 1; K/V of step j in buffer j & 1.
  auto step = [&](auto first, int buf, int tile) {
    constexpr bool FIRST = decltype(first)::value;
    const char* tk = smem + buf * BUF;
    const char* tv = tk + B_T;
    const char* tk32 = tk + 2 * B_T;
    const char* tv32 = tk32 + B_T32;
    // ---- S^T[key][q] - m: 4 key blocks x NQB query blocks, every K fragment feeds NQB MFMAs
    f4v st[NQB][4];
    u4v pb[NQB][2];
    float psum[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) psum[qb] = 0.f;
    // p = exp2(S - m) of the key-block pair kp, packed to 16 bits: the B operand of the P V product
    auto exps = [&](int kp) {
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        u4v packed;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float p0 = __builtin_amdgcn_exp2f(st[qb][2 * kp + h2][2 * j]);
            const float p1 = __builtin_amdgcn_exp2f(st[qb][2 * kp + h2][2 * j + 1]);
            if (!SUMM) psum[qb] += p0 + p1;
            packed[2 * h2 + j] = Half16<E>::pack(p0, p1);
          }
        pb[qb][kp] = packed;
      }
    };
    // O^T[dv][q] += V^T[dv][key] P^T[key][q] for the key-block pair kp (every V fragment feeds NQB MFMAs); SUMM: l += 1 P^T on the same pipe
    auto pv = [&](int kp) {
#pragma unroll
      for (int dvb = 0; dvb < NDVB; ++dvb) {
        const char* a0 = dvb < 4 ? tv + kp * 4096 + voff[dvb] : tv32 + kp * 2048 + voff[dvb];
        const bf4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf4v*)a0);
        const bf4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf4v*)(a0 + (dvb < 4 ? 2048 : 1024)));
        const u4v vf = __builtin_bit_cast(u4v, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) ot[qb][dvb] = Half16<E>::mfma16(vf, pb[qb][kp], ot[qb][dvb]);
      }
      if (SUMM) {
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) lacc[qb] = Half16<E>::mfma16(ones, pb[qb][kp], lacc[qb]);
      }
    };
    // The d-steps of TWO key blocks are interleaved (and the order prescribed to the scheduler), so that an MFMA never directly follows the one whose
    // result it accumulates on (hipcc's own order put the two d-steps of a score block back to back, with wait states between); K fragment reads
    // run two ahead of their MFMAs. + 0.4-0.6 % (profiles/r04_attn_weave_ab.txt, second part); same operations in the same order per score: bit-identical.
#pragma unroll
    for (int kb2 = 0; kb2 < 4; kb2 += 2) {
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int kb = kb2 + kk;
          const u4v kf = ds < 2 ? *reinterpret_cast<const u4v*>(tk + kb * 2048 + (ds == 0 ? koff0 : koff1))
                                : *reinterpret_cast<const u4v*>(tk32 + kb * 1024 + koff2);
#pragma unroll
          for (int qb = 0; qb < NQB; ++qb)
            st[qb][kb] = Half16<E>::mfma16(kf, qf[qb][ds], ds == 0 ? negm[qb] : st[qb][kb]);
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // K fragment reads run two ahead of the MFMAs that use them
#pragma unroll
      for (int i = 0; i < 2 * NDS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, NQB, 0);  // the NQB MFMAs of one K fragment
        if (i < 2 * NDS - 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one more read
      }
    }
    if constexpr (MASK) {  // every tile: keys at or past the query's limit (which is <= Lk) contribute nothing
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = tile * B_KV + 16 * kb + 4 * g + r;
#pragma unroll
          for (int qb = 0; qb < NQB; ++qb) st[qb][kb][r] = key < kl[qb] ? st[qb][kb][r] : NEG_INF16;
        }
    } else if constexpr (FIRST) {
      if ((Lk & (B_KV - 1)) != 0) {  // ragged tile: keys >= Lk contribute nothing
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = tile * B_KV + 16 * kb + 4 * g + r;
            if (key >= Lk) {
#pragma unroll
              for (int qb = 0; qb < NQB; ++qb) st[qb][kb][r] = NEG_INF16;
            }
          }
      }
    }
    // ---- online softmax. Lane-local maxima decide (wave-uniformly) whether anything moves; the reduction across the four
    // lanes of a query only runs inside the branch.
    float mx[NQB], mall = NEG_INF16;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      mx[qb] = max16(st[qb]);
      mall = qb == 0 ? mx[qb] : fmaxf(mall, mx[qb]);
    }
    if (FIRST || __any(mall > 8.0f)) {
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        const float mr = max_over_g(mx[qb]);
        const float delta = FIRST ? mr : fmaxf(mr, 0.f);
        if constexpr (!FIRST) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          if (SUMM) lacc[qb] *= alpha;
          else l_run[qb] *= alpha;
#pragma unroll
          for (int d = 0; d < NDVB; ++d) ot[qb][d] *= alpha;
        }
        m_run[qb] += delta;
        negm[qb] = f4v{-m_run[qb], -m_run[qb], -m_run[qb], -m_run[qb]};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) st[qb][kb] -= delta;
      }
    }
#ifndef NOVA_ATTN_NO_WEAVE
    // The exponentials of the second key-block pair are woven into the P V MFMAs of the first (one MFMA, two exponentials, one pack,
    // prescribed to the scheduler): left alone hipcc emits all 32 exponentials, then all 20 MFMAs, and a wave's vector and matrix work
    // overlap only through the SIMD's other waves. Same operations on the same values: bit-identical. Measured (tools/ab_attn.py, one process,
    // -DNOVA_ATTN_NO_WEAVE for the plain order): + 1-2 % at head_dim 64 (L = 2049 .. 2560), + 2.5-3.7 % at head_dim 96, level at L <= 1537;
    // in the benchmark's pass 979-983 -> 999-1000 TFLOP/s on one box (profiles/r04_attn_weave_ab.txt). 137 instead of 122 registers at head_dim 64.
    exps(0);
    __builtin_amdgcn_sched_barrier(0);
    exps(1);
    pv(0);
#pragma unroll
    for (int i = 0; i < NQB * NDVB; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);  // two exponentials
      __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);  // one pack
    }
    __builtin_amdgcn_sched_barrier(0);
    pv(1);
#else
    exps(0);
    exps(1);
    pv(0);
    pv(1);
#endif
    if (!SUMM) {
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) l_run[qb] += psum[qb];
    }
  };

  auto tile_of = [&](int j) { return MASK ? j : (j == 0 ? nkt - 1 : j - 1); };
  stage(0, tile_of(0));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA pieces (asm: not counted by the compiler)
  __syncthreads();
  if (nkt > 1) stage(1, tile_of(1));
  step(std::true_type{}, 0, tile_of(0));
  for (int j = 1; j < nkt; ++j) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + 1 < nkt) stage((j + 1) & 1, tile_of(j + 1));
    step(std::false_type{}, j & 1, tile_of(j));
  }
  // (Skipping the steps of a wave whose query rows all lie past Lq - `if (q0 < Lq) step(...)`, wave-uniform - was measured and lost at
  // every length, 2 % at L = 2560 where no wave is idle and 24 % at L = 768: the guarded call changes the loop hipcc builds.
  // profiles/r04_attn_dead_wave_skip_rejected.txt)

  // ---- finalize: lane (i, g) holds O[q0 + 16 qb + i][16 dvb + 4 g + 0..3]
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const float l_tot = SUMM ? lacc[qb][0] : sum_over_g(l_run[qb]);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + 16 * qb + i;
    if (qrow < Lq) {
      if (LSE && g == 0) lse[((size_t)s * heads + head) * Lq + qrow] = m_run[qb] + __log2f(l_tot);
      E* op = o + ((size_t)s * Lq + qrow) * o_rs + head * HD + 4 * g;
#pragma unroll
      for (int dvb = 0; dvb < NDVB; ++dvb) {
        u2v pk = {Half16<E>::pack(ot[qb][dvb][0] * inv, ot[qb][dvb][1] * inv), Half16<E>::pack(ot[qb][dvb][2] * inv, ot[qb][dvb][3] * inv)};
        *reinterpret_cast<u2v*>(op + 16 * dvb) = pk;
      }
    }
  }
#ifdef NOVA_CLOCK
  if (tid == 0) {
    g_attn_clock[2 * (blockIdx.x & 1023)] = __builtin_amdgcn_s_memtime() - ck_t0;
    g_attn_clock[2 * (blockIdx.x & 1023) + 1] = __builtin_amdgcn_s_memrealtime() - ck_r0;
  }
#endif
}


// ------------------------------------------------------------------------------------------------------------------
// Software-pipelined form, 64 query rows per wave: the P V product of tile t-1 shares a basic block with the
// exponentials of tile t, so one wave's matrix work and vector work overlap in its own instruction stream instead of
// relying on the SIMD's other waves (2 per SIMD at this register count). Schedule of iteration t:
//   wait + barrier | stage K(t+1), V(t) | S(t) = K(t) Q^T - m | lane-local max, rare rescale (also of the pending packed
//   P(t-1)) | { p(t) = exp2(S(t)), pack  ||  O += V(t-1) P(t-1), l += 1 P(t-1) }
// V lags K by one tile in the LDS ring (V(t) is staged with K(t+1) and read in iteration t+1), so two buffers suffice.
// Row sums ride on the matrix pipe (all-ones block). Results equal attn_bf16_m16<4, LSE, true> bit for bit.
template <typename E, bool LSE>
__global__ __launch_bounds__(256, 2) void attn_bf16_m16p(const E* __restrict__ q, const E* __restrict__ k,
                                                          const E* __restrict__ v, E* __restrict__ o, int Lq, int Lk,
                                                          long q_rs, long kv_rs, long o_rs, float c, int heads, int nq, int rev,
                                                          long kv_ss, float* __restrict__ lse) {
  constexpr int NQB = 4, HD = 64, NDS = 2, NDVB = 4, RW = 64;
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * B_T + 4 * 64 * 128];  // [buffer][K | V], then the waves' Q rows
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef NOVA_CLOCK
  const long long ck_t0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int t = xcd_remap_dir(blockIdx.x, gridDim.x, rev != 0);
  const int sh = t / nq, qt = t - sh * nq;
  const int head = sh % heads, s = sh / heads;
  const int q0 = qt * (4 * RW) + wid * RW;
  const E* qb_ = q + (size_t)s * Lq * q_rs + head * HD;
  const E* kb_ = k + (size_t)s * kv_ss + head * HD;
  const E* vb_ = v + (size_t)s * kv_ss + head * HD;

  // The wave's 64 Q rows live in its own 8 KiB of LDS (K's image and swizzle; written once, read back as B fragments at the
  // top of every tile): 32 registers that the exp / P V block of the pipeline needs more than the score block does.
  char* qs = smem + 2 * 2 * B_T + wid * (64 * 128);
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const int row = 16 * qb + i;
    const int qrow = min(q0 + row, Lq - 1);
    const E* qp = qb_ + (size_t)qrow * q_rs + 8 * g;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
      u4v f = *reinterpret_cast<const u4v*>(qp + 32 * ds);
      if (c != 1.0f) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f2v t = Half16<E>::unpack(f[j]);
          f[j] = Half16<E>::pack(t[0] * c, t[1] * c);
        }
      }
      *reinterpret_cast<u4v*>(qs + row * 128 + (((4 * ds + g) ^ ((row >> 1) & 7)) << 4)) = f;
    }
  }

  const uint32_t rowB = (uint32_t)kv_rs * 2u;
  const int srow0 = (wid * 2) * 8 + (lane >> 3), srow1 = srow0 + 8, scp = lane & 7;
  const uint32_t ck0 = (uint32_t)((scp ^ ((srow0 >> 1) & 7)) << 4), ck1 = (uint32_t)((scp ^ ((srow1 >> 1) & 7)) << 4);
  const uint32_t cv0 = (uint32_t)((scp ^ (((srow0 >> 1) & 3) << 1)) << 4), cv1 = (uint32_t)((scp ^ (((srow1 >> 1) & 3) << 1)) << 4);
  // one image (K or V, chosen by the wave-uniform base / chunk offsets) of tile kt into LDS at dst
  auto stage1 = [&](const E* src, char* dst, int kt, uint32_t c0, uint32_t c1) {
    const char* base = reinterpret_cast<const char*>(src) + (size_t)kt * B_KV * rowB;
    const int lim = Lk - 1 - kt * B_KV;
    const uint32_t r0 = (uint32_t)min(srow0, lim) * rowB, r1 = (uint32_t)min(srow1, lim) * rowB;  // rows past Lk re-read the last valid row
    glds16(base, r0 + c0, dst + wid * 2048);
    glds16(base, r1 + c1, dst + wid * 2048 + 1024);
  };

  f4v ot[NQB][NDVB], lacc[NQB];
  float m_run[NQB];
  const u4v ones = {Half16<E>::ONE2, Half16<E>::ONE2, Half16<E>::ONE2, Half16<E>::ONE2};
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    m_run[qb] = 0.f;
    lacc[qb] = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < NDVB; ++d) ot[qb][d] = f4v{0.f, 0.f, 0.f, 0.f};
  }
  const uint32_t kx = (uint32_t)((i >> 1) & 7);
  const uint32_t koff0 = (uint32_t)i * 128u + (((uint32_t)g ^ kx) << 4);
  const uint32_t koff1 = (uint32_t)i * 128u + ((((uint32_t)g + 4u) ^ kx) << 4);
  const int t_q = (lane & 15) >> 2, t_p = lane & 3;
  const uint32_t vrow = (uint32_t)(4 * g + t_q);
  const uint32_t vx = ((vrow >> 1) & 3u) << 1;
  uint32_t voff[NDVB];
#pragma unroll
  for (int dvb = 0; dvb < NDVB; ++dvb) voff[dvb] = vrow * 128u + ((((uint32_t)(2 * dvb + (t_p >> 1))) ^ vx) << 4) + 8u * (t_p & 1);

  const int nkt = (Lk + B_KV - 1) / B_KV;

  // Tile order: the (possibly ragged) LAST tile is processed first, in the prologue, so the steady-state loop carries no
  // masking code at all (softmax does not care about key order): step 0 <-> tile nkt - 1, step j >= 1 <-> tile j - 1.
  // Scores of a step minus the carried max, lane-local max, rescale (FIRST: always, sets m; else rare, and O, l and the
  // pending packed P of the previous step move too). `buf` = LDS buffer of the step's K image.
  auto score_block = [&](auto first, int buf, f4v (&st)[NQB][4], u4v (&pend)[NQB][2]) {
    constexpr bool FIRST = decltype(first)::value;
    const char* tk = smem + buf * 2 * B_T;
    u4v qf[NQB][NDS];
    f4v negm[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds) qf[qb][ds] = *reinterpret_cast<const u4v*>(qs + qb * 2048 + (ds == 0 ? koff0 : koff1));
      float nm = -m_run[qb];
      asm volatile("" : "+v"(nm));  // rebuilt per tile: the 4-register -m block is not carried through the exp / P V block
      negm[qb] = f4v{nm, nm, nm, nm};
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds) {
        const u4v kf = *reinterpret_cast<const u4v*>(tk + kb * 2048 + (ds == 0 ? koff0 : koff1));
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
          st[qb][kb] = Half16<E>::mfma16(kf, qf[qb][ds], ds == 0 ? negm[qb] : st[qb][kb]);
      }
    }
    if constexpr (FIRST) {
      if ((Lk & (B_KV - 1)) != 0) {  // ragged tile: keys >= Lk contribute nothing
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = (nkt - 1) * B_KV + 16 * kb + 4 * g + r;
            if (key >= Lk) {
#pragma unroll
              for (int qb = 0; qb < NQB; ++qb) st[qb][kb][r] = NEG_INF16;
            }
          }
      }
    }
    float mx[NQB], mall = 0.f;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      mx[qb] = max16(st[qb]);
      mall = qb == 0 ? mx[qb] : fmaxf(mall, mx[qb]);
    }
    if (FIRST || __any(mall > 8.0f)) {
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        const float mr = max_over_g(mx[qb]);
        const float delta = FIRST ? mr : fmaxf(mr, 0.f);
        if constexpr (!FIRST) {  // everything still at the old max moves exactly once: O, l and the pending P of the previous step
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          lacc[qb] *= alpha;
#pragma unroll
          for (int d = 0; d < NDVB; ++d) ot[qb][d] *= alpha;
#pragma unroll
          for (int kp = 0; kp < 2; ++kp) {
            u4v w = pend[qb][kp];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const f2v t = Half16<E>::unpack(w[e]);
              w[e] = Half16<E>::pack(t[0] * alpha, t[1] * alpha);
            }
            pend[qb][kp] = w;
          }
        }
        m_run[qb] += delta;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) st[qb][kb] -= delta;
      }
    }
  };
  // p = exp2(s), two per v_cvt_pk: the 8 probabilities of (qb, kp) in the B-operand order of the P V product
  auto exp_pack = [&](const f4v (&st)[NQB][4], int qb, int kp) -> u4v {
    u4v packed;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        packed[2 * h2 + j] = Half16<E>::pack(__builtin_amdgcn_exp2f(st[qb][2 * kp + h2][2 * j]), __builtin_amdgcn_exp2f(st[qb][2 * kp + h2][2 * j + 1]));
    return packed;
  };
  // O^T += V^T(tile) P^T for one (kp, dvb): one transposed fragment, NQB MFMAs
  auto pv_step = [&](const char* tv, const u4v (&pend)[NQB][2], int kp, int dvb) {
    const char* a0 = tv + kp * 4096 + voff[dvb];
    const bf4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf4v*)a0);
    const bf4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf4v*)(a0 + 2048));
    const u4v vf = __builtin_bit_cast(u4v, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) ot[qb][dvb] = Half16<E>::mfma16(vf, pend[qb][kp], ot[qb][dvb]);
  };
  auto sum_step = [&](const u4v (&pend)[NQB][2], int kp) {
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) lacc[qb] = Half16<E>::mfma16(ones, pend[qb][kp], lacc[qb]);
  };

  u4v pbp[NQB][2];  // packed P of the previous step, pending its P V product
  auto tile_of = [&](int step) { return step == 0 ? nkt - 1 : step - 1; };
  // ---- prologue: step 0 (the last tile) up to its packed P. Ring: K(step) in buffer step & 1, V(step) likewise; V lags K
  // by one step (staged with K(step + 1), read during step + 1).
  stage1(kb_, smem, tile_of(0), ck0, ck1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (nkt > 1) stage1(kb_, smem + 2 * B_T, tile_of(1), ck0, ck1);
  stage1(vb_, smem + B_T, tile_of(0), cv0, cv1);
  {
    f4v st[NQB][4];
    score_block(std::true_type{}, 0, st, pbp);
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
      for (int kp = 0; kp < 2; ++kp) pbp[qb][kp] = exp_pack(st, qb, kp);
  }
  // ---- steady state: this step's scores, then { P V of the previous step || exp2 / pack of this one } as ONE basic block,
  // written interleaved: after each (kp, dvb) fragment's MFMAs comes one (qb, kp) slice of exponentials
  for (int j = 1; j < nkt; ++j) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + 1 < nkt) stage1(kb_, smem + ((j + 1) & 1) * 2 * B_T, j, ck0, ck1);  // tile_of(j + 1) = j
    stage1(vb_, smem + (j & 1) * 2 * B_T + B_T, j - 1, cv0, cv1);               // tile_of(j) = j - 1
    const char* tv = smem + ((j + 1) & 1) * 2 * B_T + B_T;  // V(step j - 1)
    f4v st[NQB][4];
    score_block(std::false_type{}, j & 1, st, pbp);
    u4v pbn[NQB][2];
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
#pragma unroll
      for (int dvb = 0; dvb < NDVB; ++dvb) {
        pv_step(tv, pbp, kp, dvb);
        const int u = kp * NDVB + dvb;  // 0..7 -> (qb, kp') = (u >> 1, u & 1)
        pbn[u >> 1][u & 1] = exp_pack(st, u >> 1, u & 1);
      }
      sum_step(pbp, kp);
    }
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) { pbp[qb][0] = pbn[qb][0]; pbp[qb][1] = pbn[qb][1]; }
  }
  // ---- epilogue: the last step's P V
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    const char* tv = smem + ((nkt + 1) & 1) * 2 * B_T + B_T;  // V(step nkt - 1)
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
#pragma unroll
      for (int dvb = 0; dvb < NDVB; ++dvb) pv_step(tv, pbp, kp, dvb);
      sum_step(pbp, kp);
    }
  }

#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const float l_tot = lacc[qb][0];
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + 16 * qb + i;
    if (qrow < Lq) {
      if (LSE && g == 0) lse[((size_t)s * heads + head) * Lq + qrow] = m_run[qb] + __log2f(l_tot);
      E* op = o + ((size_t)s * Lq + qrow) * o_rs + head * HD + 4 * g;
#pragma unroll
      for (int dvb = 0; dvb < NDVB; ++dvb) {
        u2v pk = {Half16<E>::pack(ot[qb][dvb][0] * inv, ot[qb][dvb][1] * inv), Half16<E>::pack(ot[qb][dvb][2] * inv, ot[qb][dvb][3] * inv)};
        *reinterpret_cast<u2v*>(op + 16 * dvb) = pk;
      }
    }
  }
#ifdef NOVA_CLOCK
  if (tid == 0) {
    g_attn_clock[2 * (blockIdx.x & 1023)] = __builtin_amdgcn_s_memtime() - ck_t0;
    g_attn_clock[2 * (blockIdx.x & 1023) + 1] = __builtin_amdgcn_s_memrealtime() - ck_r0;
  }
#endif
}

#ifdef NOVA_CLOCK
}  // namespace nova
extern "C" int nova_debug_attn_clock(long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(nova::g_attn_clock), (size_t)(n < 2048 ? n : 2048) * 8, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
namespace nova {
#endif

// rows_per_wave 32 or 64 (head_dim 64; head_dim 96 runs the 32-row form with MFMA row sums whatever is asked); dtype NOVA_BF16 or NOVA_F16 (attn_fwd in attn.hip checks shapes and strides before it dispatches here)
int attn_fwd_m16(const void* q, const void* k, const void* v, void* o, int S, int heads, int Lq, int Lk, int hd, long q_rs, long kv_rs,
                 long o_rs, float cl, int dtype, hipStream_t st, long kv_ss, float* lse, int rows_per_wave, bool sum_on_mfma, bool pipelined,
                 const int* klim) {
  if (klim) {  // the masked training forward: built in the shipped form with the log-sum-exp output
    if (!lse) return set_error(NOVA_ERR_ARG, "attn_fwd: a key-limit mask comes with the training forward (log-sum-exp output)");
    const int nqm = (Lq + 127) / 128;
    dim3 blk(256), grd((unsigned)((long)nqm * heads * S));
    const int rv = walk_is_reverse() ? 1 : 0;
    dispatch_half(dtype, [&](auto tag) {
      using E = decltype(tag);
      if (hd == 64) hipLaunchKernelGGL((attn_bf16_m16<E, 64, 2, true, true, true>), grd, blk, 0, st, (const E*)q, (const E*)k, (const E*)v, (E*)o, Lq, Lk, q_rs, kv_rs, o_rs, cl, heads, nqm, rv, kv_ss, lse, klim);
      else hipLaunchKernelGGL((attn_bf16_m16<E, 96, 2, true, true, true>), grd, blk, 0, st, (const E*)q, (const E*)k, (const E*)v, (E*)o, Lq, Lk, q_rs, kv_rs, o_rs, cl, heads, nqm, rv, kv_ss, lse, klim);
      return 0;
    });
    return check_launch("attn_fwd_m16 (masked)");
  }
  const int rw = (rows_per_wave == 64 && hd == 64) ? 64 : 32;
  const int nq = (Lq + 4 * rw - 1) / (4 * rw);
  if ((long)nq * heads * S > 0x7fffffffL) return set_error(NOVA_ERR_SHAPE, "attn_fwd: grid too large");
  dim3 block(256), grid((unsigned)((long)nq * heads * S));
  const int rev = walk_is_reverse() ? 1 : 0;
  dispatch_half(dtype, [&](auto tag) {
    using E = decltype(tag);
    const E *qq = (const E*)q, *kk = (const E*)k, *vv = (const E*)v;
    E* oo = (E*)o;
#define NOVA_A16(NQB_, SUMM_)                                                                                                       \
  do {                                                                                                                             \
    if (lse) hipLaunchKernelGGL((attn_bf16_m16<E, 64, NQB_, true, SUMM_>), grid, block, 0, st, qq, kk, vv, oo, Lq, Lk, q_rs, kv_rs, o_rs, cl, heads, nq, rev, kv_ss, lse, nullptr); \
    else hipLaunchKernelGGL((attn_bf16_m16<E, 64, NQB_, false, SUMM_>), grid, block, 0, st, qq, kk, vv, oo, Lq, Lk, q_rs, kv_rs, o_rs, cl, heads, nq, rev, kv_ss, lse);    \
  } while (0)
    if (hd == 96) {  // built in the shipped form only: 32 rows per wave, row sums on the matrix pipe
      if (lse) hipLaunchKernelGGL((attn_bf16_m16<E, 96, 2, true, true>), grid, block, 0, st, qq, kk, vv, oo, Lq, Lk, q_rs, kv_rs, o_rs, cl, heads, nq, rev, kv_ss, lse);
      else hipLaunchKernelGGL((attn_bf16_m16<E, 96, 2, false, true>), grid, block, 0, st, qq, kk, vv, oo, Lq, Lk, q_rs, kv_rs, o_rs, cl, heads, nq, rev, kv_ss, lse);
    } else if (pipelined) {
      if (lse) hipLaunchKernelGGL((attn_bf16_m16p<E, true>), grid, block, 0, st, qq, kk, vv, oo, Lq, Lk, q_rs, kv_rs, o_rs, cl, heads, nq, rev, kv_ss, lse);
      else hipLaunchKernelGGL((attn_bf16_m16p<E, false>), grid, block, 0, st, qq, kk, vv, oo, Lq, Lk, q_rs, kv_rs, o_rs, cl, heads, nq, rev, kv_ss, lse);
    } else if (rw == 32 && !sum_on_mfma) NOVA_A16(2, false);
    else if (rw == 32) NOVA_A16(2, true);
    else if (!sum_on_mfma) NOVA_A16(4, false);
    else NOVA_A16(4, true);
#undef NOVA_A16
    return 0;
  });
  return check_launch("attn_fwd_m16");
}

}  // namespace nova
