// Shared device helpers for the NOVA gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nova {

typedef __attribute__((ext_vector_type(8))) short s8v;   // 8 x bf16 bit patterns (one MFMA A/B fragment)
typedef __attribute__((ext_vector_type(8))) __bf16 bf8v;
typedef __attribute__((ext_vector_type(4))) __bf16 bf4v;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2v;
typedef __attribute__((ext_vector_type(2))) float f2v;
typedef __attribute__((ext_vector_type(4))) short s4v;
typedef __attribute__((ext_vector_type(4))) float f4v;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) uint32_t u4v;
typedef __attribute__((ext_vector_type(2))) uint32_t u2v;

typedef uint16_t bf16_t;  // storage type for bf16 tensors

#define NOVA_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even via the hardware cvt (keeps NaN a NaN, MI355X_MICROARCH correctness table)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }

template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

// exact-erf GELU to ~1.5e-7 absolute (Abramowitz-Stegun 7.1.26): 1 rcp + 1 exp2 + 7 FMA-class ops instead
// of libm erff's ~30; used where the result is rounded to bf16 anyway (bf16 eps = 3.9e-3).
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);  // 1 - erf(z)
  const float half_x = 0.5f * x;
  return x >= 0.f ? fmaf(-half_x, e, x) : half_x * e;  // 0.5x(1 + erf) = x - 0.5x e ; 0.5x(1 - erf(|.|)) = 0.5x e
}

// XCD-aware, bijective remap of a 1-D block id: blocks that share an XCD (id % 8) get a
// contiguous chunk of the logical tile list, so neighbouring tiles hit the same per-XCD L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

// Same split, walked back to front inside each XCD's chunk when `rev` is set. The block composites alternate the
// direction from one launch to the next: a consumer then starts on the rows its producer wrote last, which are the
// ones still held by the 256 MB memory-side cache (measured: 1.8x the read rate of the evicted head of a buffer).
__device__ __forceinline__ int xcd_remap_dir(int bid, int nwg, bool rev) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int size = q + (xcd < r ? 1 : 0);
  return base + (rev ? size - 1 - k : k);
}

}  // namespace nova
