// Row LayerNorm arithmetic shared by row_norm_kernel (rowops.hip) and the LN prologue of the small-M GEMM (skinny.hip):
// one wave per row, the same lane <-> element mapping and the same operation order in both, so the fused and the
// two-kernel forms give bit-identical rows (tests/test_gpu_kernels.py compares them).
#pragma once
#include "common.h"
#include "nova_internal.h"

namespace nova {

// One wave per row, 16-byte accesses (8 bf16 / 4 f32 per lane per chunk), the whole row (D <= 2048) held in
// registers; the residual / modulation rows are requested together with the input row so that all of a row's
// HBM traffic is in flight before the two wave reductions.
template <typename T> struct Chunk;  // 16 bytes of a row as float values
template <> struct Chunk<float> {
  static constexpr int N = 4;
  typedef f4v Raw;  // the 16 bytes as loaded
  f4v v[1];
  static __device__ __forceinline__ Raw load_raw(const float* p) { return *reinterpret_cast<const f4v*>(p); }
  static __device__ __forceinline__ Chunk from_raw(Raw u) { Chunk c; c.v[0] = u; return c; }
  static __device__ __forceinline__ Chunk load(const float* p) { return from_raw(load_raw(p)); }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<f4v*>(p) = v[0]; }
  __device__ __forceinline__ Chunk rounded() const { return *this; }
};
// 16-bit storage types (bf16_t / f16_t): 8 values per 16 bytes
template <typename T> struct Chunk16 {
  static constexpr int N = 8;
  typedef u4v Raw;
  f4v v[2];
  static __device__ __forceinline__ Raw load_raw(const T* p) { return *reinterpret_cast<const u4v*>(p); }
  static __device__ __forceinline__ Chunk<T> load(const T* p) { return from_raw(load_raw(p)); }
  static __device__ __forceinline__ Chunk<T> from_raw(Raw u) {
    Chunk<T> c;
    const f2v a = Half16<T>::unpack(u[0]), b = Half16<T>::unpack(u[1]), d = Half16<T>::unpack(u[2]), e = Half16<T>::unpack(u[3]);
    c.v[0] = f4v{a[0], a[1], b[0], b[1]};
    c.v[1] = f4v{d[0], d[1], e[0], e[1]};
    return c;
  }
  __device__ __forceinline__ Raw packed() const {
    return u4v{Half16<T>::pack(v[0][0], v[0][1]), Half16<T>::pack(v[0][2], v[0][3]), Half16<T>::pack(v[1][0], v[1][1]), Half16<T>::pack(v[1][2], v[1][3])};
  }
  __device__ __forceinline__ void store(T* p) const { *reinterpret_cast<u4v*>(p) = packed(); }  // (streaming stores here: - 1.6 % on the big LayerNorm pass alone, nothing on its pair with the next GEMM: not taken, profiles/r04_gemm_nt_stores_ab.txt)
  __device__ __forceinline__ Chunk<T> rounded() const { return from_raw(packed()); }  // the values as they read back after store()
};
template <> struct Chunk<bf16_t> : Chunk16<bf16_t> {};
template <> struct Chunk<f16_t> : Chunk16<f16_t> {};

// y = LN(in[src])(*gamma + beta)(*(1 + scale) + shift)(*gate)(+ res) for one row held by one wave (reference
// vision_transformer.py:78-82,91-92 post-norm residual; diffusion_mlp.py:31-36,41-47 AdaLN-Zero modulate / gate).
// Two phases so that a caller with several rows per wave (skinny.hip) can have all their loads in flight first:
// row_norm_load requests the row's 16-byte pieces as loaded, row_norm_finish does the arithmetic.
template <typename T, int NIT, bool HAS_RES, bool HAS_MOD>
struct RowRegs {
  typename Chunk<T>::Raw x[NIT], r[HAS_RES ? NIT : 1], ms[HAS_MOD ? NIT : 1], mb[HAS_MOD ? NIT : 1], mg[HAS_MOD ? NIT : 1];
};

template <typename T, int NIT, bool HAS_RES, bool HAS_MOD>
__device__ __forceinline__ void row_norm_load(const RowNormArgs& a, long row, int lane, RowRegs<T, NIT, HAS_RES, HAS_MOD>& g,
                                              bool with_input = true) {
  using C = Chunk<T>;
  const long src = a.gather ? (long)a.gather[row] : row;
  const T* in = static_cast<const T*>(a.in) + src * a.D;
  const T* mod = HAS_MOD ? static_cast<const T*>(a.mod) + row * a.mod_ld : nullptr;
  const T* res = HAS_RES ? static_cast<const T*>(a.res) + row * a.D : nullptr;
  const bool has_ss = HAS_MOD && a.scale_off >= 0, has_gate = HAS_MOD && a.gate_off >= 0;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int d = (it * 64 + lane) * C::N;
    if (d < a.D) {
      if (with_input) g.x[it] = C::load_raw(in + d);
      if (HAS_RES) g.r[it] = C::load_raw(res + d);
      if (has_ss) { g.ms[it] = C::load_raw(mod + a.scale_off + d); g.mb[it] = C::load_raw(mod + a.shift_off + d); }
      if (has_gate) g.mg[it] = C::load_raw(mod + a.gate_off + d);
    }
  }
}

template <typename T, int NIT, bool HAS_RES, bool HAS_MOD>
__device__ __forceinline__ void row_norm_finish(const RowNormArgs& a, int lane, const RowRegs<T, NIT, HAS_RES, HAS_MOD>& g,
                                                Chunk<T> (&y)[NIT]) {
  using C = Chunk<T>;
  constexpr int NV = C::N / 4;
  const bool has_ss = HAS_MOD && a.scale_off >= 0, has_gate = HAS_MOD && a.gate_off >= 0;
  C x[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it)
    if ((it * 64 + lane) * C::N < a.D) x[it] = C::from_raw(g.x[it]);
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it)
    if ((it * 64 + lane) * C::N < a.D)
#pragma unroll
      for (int k = 0; k < NV; ++k) sum += (x[it].v[k][0] + x[it].v[k][1]) + (x[it].v[k][2] + x[it].v[k][3]);
  const float mean = wave_sum(sum) / (float)a.D;
  float sq = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it)
    if ((it * 64 + lane) * C::N < a.D)
#pragma unroll
      for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float c = x[it].v[k][j] - mean;
          sq += c * c;
        }
  const float rstd = rsqrtf(wave_sum(sq) / (float)a.D + a.eps);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int d = (it * 64 + lane) * C::N;
    if (d < a.D) {
      C ms, mb, mg, r;
      if (has_ss) { ms = C::from_raw(g.ms[it]); mb = C::from_raw(g.mb[it]); }
      if (has_gate) mg = C::from_raw(g.mg[it]);
      if (HAS_RES) r = C::from_raw(g.r[it]);
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        f4v v = (x[it].v[k] - mean) * rstd;
        if (a.gamma) v = v * *reinterpret_cast<const f4v*>(a.gamma + d + 4 * k) + *reinterpret_cast<const f4v*>(a.beta + d + 4 * k);
        if (has_ss) v = v * (1.0f + ms.v[k]) + mb.v[k];
        if (has_gate) v = v * mg.v[k];
        if (HAS_RES) v = v + r.v[k];
        y[it].v[k] = v;
      }
    }
  }
}

template <typename T, int NIT, bool HAS_RES, bool HAS_MOD>
__device__ __forceinline__ void row_norm_compute(const RowNormArgs& a, long row, int lane, Chunk<T> (&y)[NIT]) {
  RowRegs<T, NIT, HAS_RES, HAS_MOD> g;
  row_norm_load<T, NIT, HAS_RES, HAS_MOD>(a, row, lane, g);
  row_norm_finish<T, NIT, HAS_RES, HAS_MOD>(a, lane, g, y);
}

// The block boundary of the diffusion MLP as ONE row pass (diffusion_mlp.py:52-53 then :41-43 of the next block, or the
// final layer's modulate :96-97): x_new = LN(g) * gamma + beta) * gate + x, stored as bf16, then h = LN(x_new)(1 + scale)
// + shift. a2 describes the first norm (in = g, res = x, gate_off, gamma / beta), a1 the second (scale_off / shift_off;
// its `in` is not read). The second norm sees x_new exactly as it would read it back from memory (rounded to bf16), so
// the chain equals row_norm(a2) followed by row_norm(a1) bit for bit. 16-bit rows only.
template <typename T, int NIT>
struct RowChainRegs {
  RowRegs<T, NIT, true, true> first;
  RowRegs<T, NIT, false, true> second;
};

template <typename T, int NIT>
__device__ __forceinline__ void row_chain_load(const RowNormArgs& a2, const RowNormArgs& a1, long row, int lane, RowChainRegs<T, NIT>& g) {
  row_norm_load<T, NIT, true, true>(a2, row, lane, g.first);
  row_norm_load<T, NIT, false, true>(a1, row, lane, g.second, false);
}

template <typename T, int NIT>
__device__ __forceinline__ void row_chain_finish(const RowNormArgs& a2, const RowNormArgs& a1, int lane, RowChainRegs<T, NIT>& g,
                                                 u4v (&x_new)[NIT], Chunk<T> (&y)[NIT]) {
  Chunk<T> y2[NIT];
  row_norm_finish<T, NIT, true, true>(a2, lane, g.first, y2);
#pragma unroll
  for (int it = 0; it < NIT; ++it)
    if ((it * 64 + lane) * 8 < a2.D) {
      x_new[it] = y2[it].packed();
      g.second.x[it] = x_new[it];
    }
  row_norm_finish<T, NIT, false, true>(a1, lane, g.second, y);
}

}  // namespace nova
