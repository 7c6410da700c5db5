// NOVA training path: backward of the non-causal self-attention of attn.hip (bf16, head_dim 64 and 96)
//   o = softmax(q k^T * scale) v   ->   dq, dk, dv from dO        (reference vision_transformer.py:63: the autograd of
//   F.scaled_dot_product_attention inside Attention.forward, reached from train_video transformer_3d.py:79-100)
// Flash-style: P is rebuilt from q, k and the forward's per-row log-sum-exp (attn_bf16 writes it in the log2 domain),
// never stored. With s2 = q~ . k (q~ = q * scale * log2 e, the pre-scaled q the forward consumed), P = 2^(s2 - lse),
// delta = rowsum(dO * O):   dV = P^T dO,   dP = dO V^T,   dS = P * (dP - delta),   dq = scale * dS K,   dk = ln 2 * dS^T q~.
//
// Two kernels, both the forward's skeleton (4 waves x 32 rows resident in registers, the other side streamed through
// 64-row LDS tiles by 16-byte LDS-DMA, double buffered, one barrier per tile, v_mfma_f32_32x32x16_bf16), no atomics:
//   attn_bwd_dq:  a workgroup owns 128 QUERY rows and streams K / V. S^T = K Q~^T puts a query on every lane, so lse and
//                 delta are lane-local; dS^T (cast to bf16) is the B operand of dQ^T += K^T dS^T, K^T by the transposing
//                 LDS read - exactly the forward with P replaced by dS and V^T by K^T.
//   attn_bwd_dkv: a workgroup owns 128 KEY rows (K and V fragments in registers) and streams Q~ / dO. S = Q~ K^T puts a key
//                 on every lane and 16 queries in a lane's registers (their lse / delta come from the tile's LDS copy);
//                 P and dS (cast) are the B operands of dV^T += dO^T P and dK^T += Q~^T dS, dO^T / Q~^T by transposing
//                 reads of the same LDS image the row fragments come from.
// S and dP are computed in both kernels (7 MFMA products instead of the minimal 5): the price of having no cross-workgroup
// sum and of running every product in the orientation whose accumulator is directly the next product's operand.
#include "common.h"
#include "nova_internal.h"

namespace nova {

constexpr int B_T = 64;            // streamed rows per tile
constexpr int B_IMG = B_T * 128;   // one 64 x 64 bf16 image, 8 KiB
constexpr int B_IMG32 = B_T * 64;  // head_dim 96: the columns 64..95 as a second image with 64-byte rows, 4 KiB
template <int HD> constexpr int tile_bytes() { return B_IMG + (HD == 96 ? B_IMG32 : 0); }
constexpr float LN2 = 0.6931471805599453f;

// LDS image of a streamed 64 x 64 tile: chunk c (16 bytes) of row r sits at chunk c ^ swz(r), swz = the three bits of r >> 1
// in REVERSED order. Two read kinds hit this image and both must be conflict-free:
//   ds_read_b128 of row fragments: 16 consecutive rows x one chunk per pass - swz takes all 8 values over the 8 row pairs
//     (any bijection of (r >> 1) & 7 does that; rows 2j and 2j + 1 are 32 banks apart by themselves);
//   ds_read_b64_tr_b16 of transposed fragments: rows r .. r + 3 (r % 4 == 0) x 64 contiguous bytes per pass - rows r and
//     r + 2 must land 64 bytes apart, i.e. swz(r) ^ swz(r + 2) = 4: the LOW bit of r >> 1 has to become bit 2. With the
//     forward's K swizzle ((r >> 1) & 7 unreversed) those reads lost 25 % of the LDS-active cycles to bank conflicts
//     (SQ_LDS_BANK_CONFLICT); staging a second image in the forward's V swizzle cost more than it saved.
__device__ __forceinline__ int swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 1) & 2) | ((row >> 3) & 1); }

// Stage rows [t0, t0 + 64) of a token-major matrix (row stride rowB bytes, this head's first 128 bytes) into that image.
// Wave w moves pieces 2w, 2w+1 (8 rows each). Rows past `last` re-read row `last` (callers mask them).
__device__ __forceinline__ void stage_tile(const char* base, uint32_t rowB, int t0, int last, char* img, int wid, int lane) {
  const int scp = lane & 7;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wid * 2 + i) * 8 + (lane >> 3);
    glds16(base, (uint32_t)min(t0 + row, last) * rowB + (uint32_t)((scp ^ swz(row)) << 4), img + (wid * 2 + i) * 1024);
  }
}

// head_dim 96: the 32-wide image of the same rows (chunk c of row r at c ^ ((r >> 2) & 3), the forward's K32 image); wave w
// moves piece w (16 rows x 64 B). `img` points at the 64-wide image, the 32-wide one follows it.
template <int HD>
__device__ __forceinline__ void stage_rows(const char* base, uint32_t rowB, int t0, int last, char* img, int wid, int lane) {
  stage_tile(base, rowB, t0, last, img, wid, lane);
  if constexpr (HD == 96) {
    const int row = wid * 16 + (lane >> 2), scp = lane & 3;
    glds16(base, (uint32_t)min(t0 + row, last) * rowB + 128u + (uint32_t)((scp ^ ((row >> 2) & 3)) << 4), img + B_IMG + wid * 1024);
  }
}

// row fragment (A operand, rows = tile rows): lane (r, hh) of block rb reads 8 bf16 at k = 16 ks + 8 hh (ks >= 4: the 32-wide image)
__device__ __forceinline__ bf8v frag_plain(const char* img, int row, int ks, int hh) {
  if (ks < 4) return *reinterpret_cast<const bf8v*>(img + row * 128 + (((2 * ks + hh) ^ swz(row)) << 4));
  return *reinterpret_cast<const bf8v*>(img + B_IMG + row * 64 + (((2 * (ks - 4) + hh) ^ ((row >> 2) & 3)) << 4));
}

// transposed fragment (A operand, rows = the 32 columns [32 cb, 32 cb + 32) of the tile, k = tile rows 16 j .. 16 j + 15 of
// block rb): the forward's V^T read
__device__ __forceinline__ bf8v frag_tr(const char* img, int cb, int rb, int j, int lane) {
  const int hh = lane >> 5, t_qr = (lane & 15) >> 2, t_p = lane & 3, t_gp = (lane >> 4) & 1;
  const int col = cb * 32 + 16 * t_gp + 4 * t_p;
  const int chunk = col >> 3, within = (t_p & 1) * 8;
  const int row0 = rb * 32 + 16 * j + 4 * hh + t_qr, row1 = row0 + 8;
  const char *a0, *a1;
  if (cb < 2) {
    a0 = img + row0 * 128 + ((chunk ^ swz(row0)) << 4) + within;
    a1 = img + row1 * 128 + ((chunk ^ swz(row1)) << 4) + within;
  } else {  // columns 64..95: the 32-wide image
    a0 = img + B_IMG + row0 * 64 + (((chunk - 8) ^ ((row0 >> 2) & 3)) << 4) + within;
    a1 = img + B_IMG + row1 * 64 + (((chunk - 8) ^ ((row1 >> 2) & 3)) << 4) + within;
  }
  const bf4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf4v*)a0);
  const bf4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf4v*)a1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// accumulator block [32 x 32] -> two bf16 B-operand fragments (k = the block's rows 16 j .. 16 j + 15 in MFMA order)
__device__ __forceinline__ void pack_acc(const f16v& a, bf8v (&out)[2]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    u4v u;
#pragma unroll
    for (int t = 0; t < 4; ++t) u[t] = pack_bf2(a[8 * j + 2 * t], a[8 * j + 2 * t + 1]);
    out[j] = __builtin_bit_cast(bf8v, u);
  }
}

__device__ __forceinline__ f16v zero16() {
  f16v z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// ------------------------------------------------------------------------------------------
// dq: workgroup = 128 query rows of one (sequence, head)
// ------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                      const bf16_t* __restrict__ v, const bf16_t* __restrict__ d_o,
                                                      const float* __restrict__ lse, const float* __restrict__ delta,
                                                      bf16_t* __restrict__ dq, int L, long qkv_rs, long do_rs, long dq_rs,
                                                      float scale, int heads, int nq, const int* __restrict__ klim) {
  constexpr int NKS = HD / 16, NDB = HD / 32, TILE = tile_bytes<HD>();
  constexpr int BUF = 2 * TILE;  // [K | V]
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int sh = t / nq, qt = t - sh * nq;
  const int head = sh % heads, s = sh / heads;
  const int q0 = qt * 128 + wid * 32;
  const int qrow = min(q0 + r, L - 1);

  // resident: Q~ and dO fragments of the lane's query (B operands), its lse and delta
  bf8v qf[NKS], dof[NKS];
  {
    const bf16_t* qp = q + ((size_t)s * L + qrow) * qkv_rs + head * HD + 8 * hh;
    const bf16_t* dp = d_o + ((size_t)s * L + qrow) * do_rs + head * HD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      qf[ks] = *reinterpret_cast<const bf8v*>(qp + 16 * ks);
      dof[ks] = *reinterpret_cast<const bf8v*>(dp + 16 * ks);
    }
  }
  const float lse_q = lse[((size_t)s * heads + head) * L + qrow];
  const float del_q = delta[((size_t)s * heads + head) * L + qrow];
  // key limit of the lane's query (block-causal frame mask of multi-frame training: keys [0, klim[q]) are visible, klim
  // non-decreasing - attn16.hip MASK); without a mask every key below L
  const int kl_q = klim ? min(klim[qrow], L) : L;
  const int k_end = klim ? min(klim[min(qt * 128 + 127, L - 1)], L) : L;  // the workgroup's last row sees the most keys

  const char* kb_ = reinterpret_cast<const char*>(k + (size_t)s * L * qkv_rs + head * HD);
  const char* vb_ = reinterpret_cast<const char*>(v + (size_t)s * L * qkv_rs + head * HD);
  const uint32_t rowB = (uint32_t)qkv_rs * 2u;
  auto stage = [&](int buf, int kt) {
    char* b = smem + buf * BUF;
    stage_rows<HD>(kb_, rowB, kt * B_T, L - 1, b, wid, lane);
    stage_rows<HD>(vb_, rowB, kt * B_T, L - 1, b + TILE, wid, lane);
  };

  f16v dqt[NDB];  // dQ^T [d x q] in 32-row d blocks
#pragma unroll
  for (int db = 0; db < NDB; ++db) dqt[db] = zero16();
  const int nkt = (k_end + B_T - 1) / B_T;
  stage(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
    const char* tk = smem + (kt & 1) * BUF;
    const char* tv = tk + TILE;
    bf8v dsb[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int row = kb * 32 + r;
      f16v st = zero16(), dpt = zero16();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(tk, row, ks, hh), qf[ks], st, 0, 0, 0);    // S^T = K Q~^T
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(tv, row, ks, hh), dof[ks], dpt, 0, 0, 0); // dP^T = V dO^T
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = kt * B_T + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        const float p = key < kl_q ? __builtin_amdgcn_exp2f(st[i] - lse_q) : 0.f;
        st[i] = p * (dpt[i] - del_q);  // dS^T
      }
      pack_acc(st, dsb[kb]);
    }
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          dqt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(tk, db, kb, j, lane), dsb[kb][j], dqt[db], 0, 0, 0);  // dQ^T += K^T dS^T
  }

  if (q0 + r < L) {
    bf16_t* op = dq + ((size_t)s * L + q0 + r) * dq_rs + head * HD;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * hh;
        u2v pk = {pack_bf2(dqt[db][4 * g] * scale, dqt[db][4 * g + 1] * scale), pack_bf2(dqt[db][4 * g + 2] * scale, dqt[db][4 * g + 3] * scale)};
        *reinterpret_cast<u2v*>(op + d) = pk;
      }
  }
}

// ------------------------------------------------------------------------------------------
// dk, dv: workgroup = 128 key rows of one (sequence, head)
// ------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256, HD == 64 ? 2 : 1) void attn_bwd_dkv(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                       const bf16_t* __restrict__ v, const bf16_t* __restrict__ d_o,
                                                       const float* __restrict__ lse, const float* __restrict__ delta,
                                                       bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int L, long qkv_rs,
                                                       long do_rs, long dkv_rs, int heads, int nk, const int* __restrict__ klim) {
  constexpr int NKS = HD / 16, NDB = HD / 32, TILE = tile_bytes<HD>();
  constexpr int BUF = 2 * TILE + 3 * B_T * 4;  // [Q | dO | lse | delta | key limit]
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int sh = t / nk, kt0 = t - sh * nk;
  const int head = sh % heads, s = sh / heads;
  const int k0 = kt0 * 128 + wid * 32;
  const int krow = min(k0 + r, L - 1);

  bf8v kf[NKS], vf[NKS];  // B operands: the lane's key, 8 features at 16 ks + 8 hh
  {
    const bf16_t* kp = k + ((size_t)s * L + krow) * qkv_rs + head * HD + 8 * hh;
    const bf16_t* vp = v + ((size_t)s * L + krow) * qkv_rs + head * HD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      kf[ks] = *reinterpret_cast<const bf8v*>(kp + 16 * ks);
      vf[ks] = *reinterpret_cast<const bf8v*>(vp + 16 * ks);
    }
  }
  const char* qb_ = reinterpret_cast<const char*>(q + (size_t)s * L * qkv_rs + head * HD);
  const char* ob_ = reinterpret_cast<const char*>(d_o + (size_t)s * L * do_rs + head * HD);
  const float* lse_b = lse + ((size_t)s * heads + head) * L;
  const float* del_b = delta + ((size_t)s * heads + head) * L;
  const uint32_t qB = (uint32_t)qkv_rs * 2u, oB = (uint32_t)do_rs * 2u;
  auto stage = [&](int buf, int qt) {
    char* b = smem + buf * BUF;
    stage_rows<HD>(qb_, qB, qt * B_T, L - 1, b, wid, lane);
    stage_rows<HD>(ob_, oB, qt * B_T, L - 1, b + TILE, wid, lane);
    if (tid < 3 * B_T) {  // the tile's lse | delta | key limit (plain stores: visible after the barrier that opens the tile)
      const int i = tid & (B_T - 1);
      const int row = min(qt * B_T + i, L - 1);
      if (tid < 2 * B_T) reinterpret_cast<float*>(b + 2 * TILE)[tid] = tid < B_T ? lse_b[row] : del_b[row];
      else reinterpret_cast<int*>(b + 2 * TILE)[tid] = (qt * B_T + i) >= L ? 0 : (klim ? klim[row] : L);  // queries past L see no key
    }
  };

  f16v dkt[NDB], dvt[NDB];  // dK^T, dV^T [d x key] in 32-row d blocks
#pragma unroll
  for (int db = 0; db < NDB; ++db) dkt[db] = dvt[db] = zero16();
  const int nqt = (L + B_T - 1) / B_T;
  // masked: query tiles whose largest limit does not reach this workgroup's first key contribute nothing (klim is non-decreasing,
  // so they form a prefix of the tile list)
  int qt_first = 0;
  if (klim) {
    const int key_first = kt0 * 128;
    while (qt_first < nqt - 1 && klim[min(qt_first * B_T + B_T - 1, L - 1)] <= key_first) ++qt_first;
  }
  const int key_lane = k0 + r;  // the lane's key
  stage(qt_first & 1, qt_first);
  for (int qt = qt_first; qt < nqt; ++qt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (qt + 1 < nqt) stage((qt + 1) & 1, qt + 1);
    const char* tq = smem + (qt & 1) * BUF;
    const char* to = tq + TILE;
    const float* tl = reinterpret_cast<const float*>(tq + 2 * TILE);
    const int* tkl = reinterpret_cast<const int*>(tq + 2 * TILE) + 2 * B_T;
    bf8v pb[2][2], dsb[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int row = qb * 32 + r;
      f16v sa = zero16(), dpa = zero16();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(tq, row, ks, hh), kf[ks], sa, 0, 0, 0);    // S = Q~ K^T
        dpa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain(to, row, ks, hh), vf[ks], dpa, 0, 0, 0);  // dP = dO V^T
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // accumulator rows 4 g .. 4 g + 3 are queries 32 qb + 8 g + 4 hh + 0..3
        const int qi = qb * 32 + 8 * g + 4 * hh;
        const f4v l4 = *reinterpret_cast<const f4v*>(tl + qi), d4 = *reinterpret_cast<const f4v*>(tl + B_T + qi);
        const auto k4 = *reinterpret_cast<const __attribute__((ext_vector_type(4))) int*>(tkl + qi);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = key_lane < k4[e] ? __builtin_amdgcn_exp2f(sa[4 * g + e] - l4[e]) : 0.f;
          sa[4 * g + e] = p;
          dpa[4 * g + e] = p * (dpa[4 * g + e] - d4[e]);  // dS
        }
      }
      pack_acc(sa, pb[qb]);
      pack_acc(dpa, dsb[qb]);
    }
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          dvt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(to, db, qb, j, lane), pb[qb][j], dvt[db], 0, 0, 0);   // dV^T += dO^T P
          dkt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(tq, db, qb, j, lane), dsb[qb][j], dkt[db], 0, 0, 0);  // dK^T += Q~^T dS
        }
  }

  if (k0 + r < L) {
    bf16_t* kp = dk + ((size_t)s * L + k0 + r) * dkv_rs + head * HD;
    bf16_t* vp = dv + ((size_t)s * L + k0 + r) * dkv_rs + head * HD;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * hh;
        u2v pk = {pack_bf2(dkt[db][4 * g] * LN2, dkt[db][4 * g + 1] * LN2), pack_bf2(dkt[db][4 * g + 2] * LN2, dkt[db][4 * g + 3] * LN2)};
        u2v pv = {pack_bf2(dvt[db][4 * g], dvt[db][4 * g + 1]), pack_bf2(dvt[db][4 * g + 2], dvt[db][4 * g + 3])};
        *reinterpret_cast<u2v*>(kp + d) = pk;
        *reinterpret_cast<u2v*>(vp + d) = pv;
      }
  }
}

// delta[s, head, l] = sum_c dO[s, l, head, c] * O[s, l, head, c]: 16 lanes per (token, head), 16 bytes each (HD / 8 of them active)
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ o,
                                                         float* __restrict__ delta, int L, int heads, int hd, long do_rs, long o_rs,
                                                         long total) {
  const long i = (long)blockIdx.x * 16 + (threadIdx.x >> 4);  // (s, l, head) index, head fastest
  const int part = threadIdx.x & 15;
  float acc = 0.f;
  long s = 0;
  int l = 0, head = 0;
  if (i < total) {
    head = (int)(i % heads);
    const long sl = i / heads;
    l = (int)(sl % L);
    s = sl / L;
    if (part * 8 < hd) {
      const u4v a = *reinterpret_cast<const u4v*>(d_o + sl * do_rs + head * hd + part * 8);
      const u4v b = *reinterpret_cast<const u4v*>(o + sl * o_rs + head * hd + part * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc += __uint_as_float(a[j] << 16) * __uint_as_float(b[j] << 16) + __uint_as_float(a[j] & 0xffff0000u) * __uint_as_float(b[j] & 0xffff0000u);
    }
  }
  acc += dpp_move<0xb1>(acc);   // xor 1
  acc += dpp_move<0x4e>(acc);   // xor 2
  acc += dpp_move<0x141>(acc);  // i <-> 7 - i
  acc += dpp_move<0x140>(acc);  // i <-> 15 - i: completes the 16-lane sum
  if (i < total && part == 0) delta[(s * heads + head) * L + l] = acc;
}

int attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, float* delta, void* dq,
             void* dk, void* dv, int S, int heads, int L, int hd, long qkv_rs, long o_rs, long do_rs, long dqkv_rs, float scale,
             hipStream_t st, const int* klim) {
  if (S <= 0 || L <= 0) return 0;
  if (heads <= 0) return set_error(NOVA_ERR_SHAPE, "attn_bwd: bad heads");
  if (hd != 64 && hd != 96) return set_error(NOVA_ERR_SHAPE, "attn_bwd: head_dim %d not built (have 64 and 96)", hd);
  if (qkv_rs % 8 || do_rs % 8 || dqkv_rs % 8 || o_rs % 8) return set_error(NOVA_ERR_SHAPE, "attn_bwd: row strides must be 16-byte multiples");
  if ((long)L * qkv_rs * 2 > 0x7fffffffL || (long)L * do_rs * 2 > 0x7fffffffL) return set_error(NOVA_ERR_SHAPE, "attn_bwd: a sequence's rows must span < 2 GiB");
  const int nt = (L + 127) / 128;
  const long blocks = (long)nt * heads * S;
  if (blocks > 0x7fffffffL) return set_error(NOVA_ERR_SHAPE, "attn_bwd: grid too large");
  const bf16_t *qq = (const bf16_t*)q, *kk = (const bf16_t*)k, *vv = (const bf16_t*)v, *oo = (const bf16_t*)d_o;
  const long rows = (long)S * L * heads;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, oo, (const bf16_t*)o, delta, L, heads, hd,
                     do_rs, o_rs, rows);
  const dim3 grid((unsigned)blocks), block(256);
  if (hd == 64) {
    hipLaunchKernelGGL(attn_bwd_dq<64>, grid, block, 0, st, qq, kk, vv, oo, lse, delta, (bf16_t*)dq, L, qkv_rs, do_rs, dqkv_rs, scale, heads, nt, klim);
    hipLaunchKernelGGL(attn_bwd_dkv<64>, grid, block, 0, st, qq, kk, vv, oo, lse, delta, (bf16_t*)dk, (bf16_t*)dv, L, qkv_rs, do_rs, dqkv_rs,
                       heads, nt, klim);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq<96>, grid, block, 0, st, qq, kk, vv, oo, lse, delta, (bf16_t*)dq, L, qkv_rs, do_rs, dqkv_rs, scale, heads, nt, klim);
    hipLaunchKernelGGL(attn_bwd_dkv<96>, grid, block, 0, st, qq, kk, vv, oo, lse, delta, (bf16_t*)dk, (bf16_t*)dv, L, qkv_rs, do_rs, dqkv_rs,
                       heads, nt, klim);
  }
  return check_launch("attn_bwd");
}

}  // namespace nova
