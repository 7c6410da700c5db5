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
// storage type for IEEE half tensors (NOVA_F16, the default precision of the reference's callers: scripts/app_nova_t2i.py:36):
// a distinct 2-byte type so that every kernel template instantiates for it beside bf16_t
struct f16_t { uint16_t bits; };
typedef __attribute__((ext_vector_type(8))) _Float16 h8v;
typedef __attribute__((ext_vector_type(4))) _Float16 h4v;
typedef __attribute__((ext_vector_type(2))) _Float16 h2v;

#define NOVA_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// 16-byte LDS-DMA with a wave-uniform 64-bit base in SGPRs and a per-lane 32-bit byte offset (the instruction's saddr
// form). Through the builtin the compiler materialises a 64-bit VGPR address per issue (v_lshl_add_u64, plus one more
// per loop-carried pointer), which in the VALU-bound attention loop is ~8 two-pass instructions per tile.
// M0 carries the LDS destination and is compiler-reserved: saved and restored inside the statement. The load is
// invisible to the compiler's s_waitcnt bookkeeping: callers wait with an explicit s_waitcnt vmcnt.
__device__ __forceinline__ void glds16(const void* base_uniform, uint32_t byte_off, const void* lds_dst_uniform) {
  uint32_t keep;
  const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds_dst_uniform;  // LDS byte offset
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(byte_off), "s"(base_uniform), "s"(dst)
               : "memory");
}

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even via the hardware cvt (keeps NaN a NaN, MI355X_MICROARCH correctness table)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

// two floats -> one dword of two bf16 with ONE v_cvt_pk_bf16_f32 (the scalar-convert-shift-or form costs 3-4 VALU
// issues per pair, which in the un-overlapped GEMM epilogues is paid 64 times per lane per tile)
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  const bf2v h = __builtin_convertvector(f2v{lo, hi}, bf2v);
  return __builtin_bit_cast(uint32_t, h);
}

// two floats -> one dword of two halves, round-to-nearest-even (v_cvt_pk_f16_f32 on gfx950; out-of-range values become
// infinities exactly as torch's .half() makes them)
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
  const h2v h = __builtin_convertvector(f2v{lo, hi}, h2v);
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float h2f(uint16_t bits) { return (float)__builtin_bit_cast(_Float16, bits); }
// the two halves of a dword as floats
__device__ __forceinline__ f2v unpack_h2(uint32_t u) { return __builtin_convertvector(__builtin_bit_cast(h2v, u), f2v); }

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <> __device__ __forceinline__ float to_f<f16_t>(f16_t v) { return h2f(v.bits); }

template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }
template <> __device__ __forceinline__ f16_t from_f<f16_t>(float v) { return f16_t{__builtin_bit_cast(uint16_t, (_Float16)v)}; }

// 16-bit storage types: pair packing / unpacking and the MFMA that consumes 8-element fragments of them (held as u4v: the
// two types share every load, LDS image and store; only these three operations differ)
template <typename T> struct Half16;
template <> struct Half16<bf16_t> {
  static __device__ __forceinline__ uint32_t pack(float lo, float hi) { return pack_bf2(lo, hi); }
  static __device__ __forceinline__ f2v unpack(uint32_t u) { return f2v{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }
  static __device__ __forceinline__ f4v mfma16(u4v a, u4v b, f4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f16v mfma32(u4v a, u4v b, f16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c, 0, 0, 0);
  }
  static constexpr uint32_t ONE2 = 0x3f803f80u;  // two ones
};
template <> struct Half16<f16_t> {
  static __device__ __forceinline__ uint32_t pack(float lo, float hi) { return pack_h2(lo, hi); }
  static __device__ __forceinline__ f2v unpack(uint32_t u) { return unpack_h2(u); }
  static __device__ __forceinline__ f4v mfma16(u4v a, u4v b, f4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f16v mfma32(u4v a, u4v b, f16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
  }
  static constexpr uint32_t ONE2 = 0x3c003c00u;
};

// Wave-wide sum / max, every lane ending with the same value, without the LDS crossbar: __shfl_xor compiles to
// ds_bpermute_b32 (an LDS-pipe round trip per step, six steps); here four steps are DPP operands of the add itself
// and the last two are the gfx950 row swaps. Each step pairs lane groups symmetrically (both partners compute
// a + b), so all lanes agree bit for bit. Pairing: i <-> 7 - i (row_half_mirror), xor 1, xor 2 (quad_perm),
// i <-> 15 - i (row_mirror), rows 0|1 and 2|3 (v_permlane16_swap), halves (v_permlane32_swap).
template <int CTRL> __device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <typename F> __device__ __forceinline__ float wave_combine(float v, F f) {
  v = f(v, dpp_move<0x141>(v));
  v = f(v, dpp_move<0xb1>(v));
  v = f(v, dpp_move<0x4e>(v));
  v = f(v, dpp_move<0x140>(v));
  const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = f(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
  const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return f(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
}
__device__ __forceinline__ float wave_sum(float v) { return wave_combine(v, [](float a, float b) { return a + b; }); }
__device__ __forceinline__ float wave_max(float v) { return wave_combine(v, [](float a, float b) { return fmaxf(a, b); }); }

// max over a lane and its partner lane ^ 32 without the LDS round trip of ds_bpermute: v_permlane32_swap of x with
// itself leaves (x.lo, x.lo) and (x.hi, x.hi) in the two results
__device__ __forceinline__ float max_xor32(float x) {
  const uint32_t u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

// erf-GELU for results that are rounded to bf16 anyway: x * sigmoid(x * q(x^2)), q an even quintic fitted (minimax,
// tools/fit_gelu.py) to 0.5 x (1 + erf(x / sqrt 2)): max abs error 2.5e-5 over all x, i.e. <= 1.5 % of a bf16 ulp
// wherever |gelu| >= 0.25 and far below the rounding of the bf16 value it is stored as. 9 VALU issues (2 of them the
// 8-cycle exp2 / rcp) against ~25 for an erfc-polynomial form: in the fc1 epilogue this is 128 values per lane per
// 256x256 tile with no matrix work to hide under (measured: the old form cost 37 % of the fc1 tile period).
// Coefficients carry the -log2(e) of the exp2 domain. x^2 is clamped at 64: beyond |x| = 8 the sigmoid is 0 or 1 to
// 1e-11 and the quintic would turn over at |x| ~ 10.7. The f32 parity path keeps libm's erff (gelu_erf).
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float x2 = fminf(x * x, 64.0f);
  float q = fmaf(1.0142630601e-3f, x2, -1.0677572399e-1f);
  q = fmaf(q, x2, -2.3011213396e+0f);
  const float e = __builtin_amdgcn_exp2f(q * x);  // exp(-x q(x^2))
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// The same arithmetic on the 4 values a lane holds of one accumulator fragment, written on float pairs so that the
// polynomial, the 1 + e and the final product become v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 (two values per issue;
// every element goes through the same single-rounded operations in the same order as gelu_erf_fast, so the results are
// bit-identical). The epilogue runs with the matrix pipe idle, where every vector issue is paid in full.
__device__ __forceinline__ f4v gelu_erf_fast4(f4v x) {
  f4v out;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f2v xx = {x[2 * h], x[2 * h + 1]};
    f2v x2 = xx * xx;
    x2 = f2v{fminf(x2[0], 64.0f), fminf(x2[1], 64.0f)};
    f2v q = __builtin_elementwise_fma(f2v{1.0142630601e-3f, 1.0142630601e-3f}, x2, f2v{-1.0677572399e-1f, -1.0677572399e-1f});
    q = __builtin_elementwise_fma(q, x2, f2v{-2.3011213396e+0f, -2.3011213396e+0f});
    const f2v z = q * xx;
    const f2v s = f2v{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])} + f2v{1.0f, 1.0f};
    const f2v r = xx * f2v{__builtin_amdgcn_rcpf(s[0]), __builtin_amdgcn_rcpf(s[1])};
    out[2 * h] = r[0];
    out[2 * h + 1] = r[1];
  }
  return out;
}

// RoPE rotation of the two adjacent pairs a lane holds of one accumulator fragment (embeddings.py:36-43): (x0, x1) -> (c0 x0 - s0 x1,
// s0 x0 + c0 x1), likewise (x2, x3) with (c1, s1); t = (c0, s0, c1, s1). One function for every GEMM epilogue, so the tile
// structures stay bit-identical - and with the roundings WRITTEN OUT: each output is one product rounded to f32 and one fused
// multiply-add on top of it. Left to the compiler (fp-contract is on by default), which of the two products of `a b - c d` is fused
// depends on the code around the call, and two kernels inlining the same source line can round differently (round 4: the rewritten
// 256-tile epilogue no longer matched the 128-tile kernel bit for bit).
// (A v_pk_mul_f32 + v_pk_fma_f32 form with op_sel / neg_lo modifiers - half the issues - measured SLOWER in the persistent
// kernel: rotate / no-rotate time ratio 1.146 against 1.107 for the scalar form, round 3.)
__device__ __forceinline__ f4v rope_rotate4(f4v x, f4v t) {
  return f4v{__builtin_fmaf(t[0], x[0], -__fmul_rn(t[1], x[1])), __builtin_fmaf(t[1], x[0], __fmul_rn(t[0], x[1])),
             __builtin_fmaf(t[2], x[2], -__fmul_rn(t[3], x[3])), __builtin_fmaf(t[3], x[2], __fmul_rn(t[2], x[3]))};
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
