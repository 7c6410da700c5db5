// Training side: backward of the LayerNorm family of rowops.hip / rownorm.h
//   y = LN(x; eps) [* gamma + beta] [* (1 + scale) + shift] [* gate] [+ res]
// i.e. the post-norm residual of a ViT block (reference vision_transformer.py:78-82,91-92), AdaLayerNormZero's modulate
// (normalization.py:34-36) and the gated norm of a DiffusionBlock (diffusion_mlp.py:52-53), in one row pass:
//   n = (x - mean) rstd;  u1 = n gamma + beta;  u2 = u1 (1 + scale) + shift;  y = u2 gate + res
//   d_gate = dy u2;  du2 = dy gate;  d_scale = du2 u1;  d_shift = du2;  du1 = du2 (1 + scale)
//   d_gamma += du1 n;  d_beta += du1;  dn = du1 gamma;  dx = rstd (dn - mean(dn) - n mean(dn n));  d_res = dy (the caller's)
// One wave per row, the whole row in registers, 16-byte accesses, statistics recomputed from x (nothing but x is kept by the
// forward); a wave walks rows w, w + W, ... and keeps its lanes' d_gamma / d_beta sums in registers, written once per wave
// as a partial row ([waves, D] f32, summed by the caller): no atomics, bitwise reproducible. HBM-bound like the forward:
// reads x, dy (+ the modulation rows it needs), writes dx (+ d_scale / d_shift / d_gate).
#include "common.h"
#include "nova_internal.h"
#include "rownorm.h"

namespace nova {

struct RowNormBwdArgs {
  const void* x;        // [rows, D] the forward's LN input
  const void* dy;       // [rows, D]
  const float* gamma;   // [D] or null
  const float* beta;    // [D] with gamma (u1 = n gamma + beta enters d_scale and d_gate)
  const void* mod;      // modulation rows [rows, mod_ld] or null
  long mod_ld;
  int scale_off, shift_off, gate_off;  // as in the forward; -1 = absent (shift is read only when a gate needs u2)
  void* dx;             // [rows, D]
  void* dmod;           // [rows, mod_ld] or null: d_scale / d_shift / d_gate at the forward's offsets
  float* dgamma_part;   // [parts, D] or null
  float* dbeta_part;    // [parts, D] or null
  long rows;
  int D;
  float eps;
};

template <typename T, int NIT, bool HAS_MOD>
__global__ __launch_bounds__(256) void row_norm_bwd_kernel(RowNormBwdArgs a) {
  using C = Chunk<T>;
  constexpr int NV = C::N / 4;
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  const bool has_ss = HAS_MOD && a.scale_off >= 0, has_gate = HAS_MOD && a.gate_off >= 0;
  const bool affine = a.gamma != nullptr;
  f4v gam[NIT][NV], bet[NIT][NV], dgam[NIT][NV], dbet[NIT][NV];
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int d = (it * 64 + lane) * C::N + 4 * k;
      gam[it][k] = (affine && d < a.D) ? *reinterpret_cast<const f4v*>(a.gamma + d) : f4v{1.f, 1.f, 1.f, 1.f};
      bet[it][k] = (affine && d < a.D) ? *reinterpret_cast<const f4v*>(a.beta + d) : f4v{0.f, 0.f, 0.f, 0.f};
      dgam[it][k] = f4v{0.f, 0.f, 0.f, 0.f};
      dbet[it][k] = f4v{0.f, 0.f, 0.f, 0.f};
    }
  const float inv_d = 1.0f / (float)a.D;
  for (long row = wave; row < a.rows; row += nwaves) {
    const T* xp = static_cast<const T*>(a.x) + row * a.D;
    const T* dyp = static_cast<const T*>(a.dy) + row * a.D;
    const T* mp = HAS_MOD ? static_cast<const T*>(a.mod) + row * a.mod_ld : nullptr;
    C x[NIT], dy[NIT], ms[NIT], mb[NIT], mg[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {  // all of the row's traffic in flight before the first reduction
      const int d = (it * 64 + lane) * C::N;
      if (d < a.D) {
        x[it] = C::load(xp + d);
        dy[it] = C::load(dyp + d);
        if (has_ss) ms[it] = C::load(mp + a.scale_off + d);
        if (has_gate) mg[it] = C::load(mp + a.gate_off + d);
        if (has_gate && has_ss) mb[it] = C::load(mp + a.shift_off + d);
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if ((it * 64 + lane) * C::N < a.D)
#pragma unroll
        for (int k = 0; k < NV; ++k) sum += (x[it].v[k][0] + x[it].v[k][1]) + (x[it].v[k][2] + x[it].v[k][3]);
    const float mean = wave_sum(sum) * inv_d;
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
    const float rstd = rsqrtf(wave_sum(sq) * inv_d + a.eps);
    // n ends up in x, dn in dy; the modulation gradients are stored as soon as they are formed
    T* dmp = (HAS_MOD && a.dmod) ? static_cast<T*>(a.dmod) + row * a.mod_ld : nullptr;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int d = (it * 64 + lane) * C::N;
      if (d < a.D) {
        C dsc, dsh, dgt;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          const f4v n = (x[it].v[k] - mean) * rstd;
          const f4v u1 = n * gam[it][k] + bet[it][k];
          f4v g = dy[it].v[k];
          if (has_gate) {
            const f4v u2 = has_ss ? u1 * (1.0f + ms[it].v[k]) + mb[it].v[k] : u1;
            dgt.v[k] = g * u2;
            g = g * mg[it].v[k];
          }
          if (has_ss) {
            dsh.v[k] = g;
            dsc.v[k] = g * u1;
            g = g * (1.0f + ms[it].v[k]);
          }
          dgam[it][k] += g * n;
          dbet[it][k] += g;
          const f4v dn = g * gam[it][k];
          x[it].v[k] = n;
          dy[it].v[k] = dn;
          s1 += (dn[0] + dn[1]) + (dn[2] + dn[3]);
          s2 += (dn[0] * n[0] + dn[1] * n[1]) + (dn[2] * n[2] + dn[3] * n[3]);
        }
        if (dmp) {
          if (has_gate) dgt.store(dmp + a.gate_off + d);
          if (has_ss) { dsc.store(dmp + a.scale_off + d); dsh.store(dmp + a.shift_off + d); }
        }
      }
    }
    const float m1 = wave_sum(s1) * inv_d, m2 = wave_sum(s2) * inv_d;
    T* dxp = static_cast<T*>(a.dx) + row * a.D;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int d = (it * 64 + lane) * C::N;
      if (d < a.D) {
        C out;
#pragma unroll
        for (int k = 0; k < NV; ++k) out.v[k] = (dy[it].v[k] - m1 - x[it].v[k] * m2) * rstd;
        out.store(dxp + d);
      }
    }
  }
  if (a.dgamma_part) {
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int d = (it * 64 + lane) * C::N + 4 * k;
        if (d < a.D) {
          *reinterpret_cast<f4v*>(a.dgamma_part + wave * a.D + d) = dgam[it][k];
          *reinterpret_cast<f4v*>(a.dbeta_part + wave * a.D + d) = dbet[it][k];
        }
      }
  }
}

template <typename T, int NIT>
static void launch_bwd(const RowNormBwdArgs& a, dim3 grid, hipStream_t st) {
  if (a.mod) hipLaunchKernelGGL((row_norm_bwd_kernel<T, NIT, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((row_norm_bwd_kernel<T, NIT, false>), grid, dim3(256), 0, st, a);
}

// parts: number of partial rows of d_gamma / d_beta the caller allocated; the launch uses parts / 4 workgroups (parts % 4 == 0)
int row_norm_bwd(const RowNormBwdArgs& a, int parts, int dtype, hipStream_t st) {
  if (a.rows <= 0) return 0;
  const int vec = dtype_is16(dtype) ? 8 : 4;
  if (a.D % vec != 0 || a.D > 2048) return set_error(NOVA_ERR_SHAPE, "row_norm_bwd: D=%d unsupported (need D %% %d == 0, D <= 2048)", a.D, vec);
  if (a.mod && (a.mod_ld % vec || (a.scale_off >= 0 && (a.scale_off % vec || a.shift_off % vec || a.shift_off < 0)) ||
                (a.gate_off >= 0 && a.gate_off % vec)))
    return set_error(NOVA_ERR_SHAPE, "row_norm_bwd: modulation offsets must be multiples of %d", vec);
  if (a.mod) {  // every term's D columns inside a modulation row, and somewhere to put their gradients
    for (int off : {a.scale_off, a.shift_off, a.gate_off})
      if (off >= 0 && (long)off + a.D > a.mod_ld) return set_error(NOVA_ERR_SHAPE, "row_norm_bwd: modulation offset %d + D %d exceeds the row stride %ld", off, a.D, (long)a.mod_ld);
    if (!a.dmod) return set_error(NOVA_ERR_ARG, "row_norm_bwd: modulation terms given without a gradient buffer (dmod)");
  }
  if ((a.gamma == nullptr) != (a.beta == nullptr)) return set_error(NOVA_ERR_ARG, "row_norm_bwd: gamma/beta must come together");
  if ((a.dgamma_part == nullptr) != (a.dbeta_part == nullptr) || (a.gamma && !a.dgamma_part))
    return set_error(NOVA_ERR_ARG, "row_norm_bwd: an affine norm needs both partial-sum buffers");
  if (parts < 4 || parts % 4) return set_error(NOVA_ERR_ARG, "row_norm_bwd: parts must be a positive multiple of 4");
  dim3 grid((unsigned)(parts / 4));
  const int chunks = (a.D / vec + 63) / 64;
  if (dtype_is16(dtype)) {
    dispatch_half(dtype, [&](auto tag) {
      if (chunks <= 2) launch_bwd<decltype(tag), 2>(a, grid, st);
      else launch_bwd<decltype(tag), 4>(a, grid, st);
      return 0;
    });
  } else {
    if (chunks <= 4) launch_bwd<float, 4>(a, grid, st);
    else launch_bwd<float, 8>(a, grid, st);
  }
  return check_launch("row_norm_bwd");
}

// ---------------------------------------------------------------------------------------------------------------------------
// Training side, pointwise: the activations between the two projections of an MLP, forward and backward in f32 registers on the
// storage type's values. GELU is the exact erf form the reference trains with (nn.GELU(), vision_transformer.py:33-38; the fitted
// form of common.h is the generation path's), SiLU the DiffusionBlock's (diffusion_mlp.py:31-36):
//   gelu(x) = x Phi(x),  gelu'(x) = Phi(x) + x phi(x);     silu(x) = x s(x),  silu'(x) = s (1 + x (1 - s)),  s = 1 / (1 + e^-x)
// HBM-bound: forward reads x and writes y, backward reads x and dy and writes dx (nothing but x is kept by the forward).
template <int KIND> __device__ __forceinline__ float act_value(float x) { return KIND == NOVA_ACT_GELU_ERF ? gelu_erf(x) : silu(x); }
template <int KIND> __device__ __forceinline__ float act_slope(float x) {
  if (KIND == NOVA_ACT_GELU_ERF) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    return cdf + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
  }
  const float s = 1.0f / (1.0f + __expf(-x));
  return s * (1.0f + x * (1.0f - s));
}

template <typename T, int KIND, bool BWD>
__global__ __launch_bounds__(256) void act_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, long n) {
  using C = Chunk<T>;  // 16 bytes per lane: 8 values of a 16-bit type, 4 floats
  const long nchunks = n / C::N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    C v = C::load(x + i * C::N), g, o;
    if (BWD) g = C::load(dy + i * C::N);
#pragma unroll
    for (int k = 0; k < C::N / 4; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) o.v[k][j] = BWD ? g.v[k][j] * act_slope<KIND>(v.v[k][j]) : act_value<KIND>(v.v[k][j]);
    o.store(out + i * C::N);
  }
}

// forward: dy == nullptr, out = act(x); backward: out = dy * act'(x). n % (16 / element size) == 0 (whole 16-byte chunks).
int act_pointwise(const void* x, const void* dy, void* out, long n, int kind, int dtype, hipStream_t st) {
  if (n <= 0) return 0;
  if (kind != NOVA_ACT_GELU_ERF && kind != NOVA_ACT_SILU) return set_error(NOVA_ERR_ARG, "act: kind must be NOVA_ACT_GELU_ERF or NOVA_ACT_SILU (got %d)", kind);
  const int vec = dtype_is16(dtype) ? 8 : 4;
  if (n % vec) return set_error(NOVA_ERR_SHAPE, "act: element count %ld is not a multiple of %d", n, vec);
  const long nchunks = n / vec;
  const int blocks = (int)((nchunks + 255) / 256 > 16384 ? 16384 : (nchunks + 255) / 256);
  dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    const T* xp = static_cast<const T*>(x);
    const T* gp = static_cast<const T*>(dy);
    T* op = static_cast<T*>(out);
    if (kind == NOVA_ACT_GELU_ERF) {
      if (dy) hipLaunchKernelGGL((act_kernel<T, NOVA_ACT_GELU_ERF, true>), dim3(blocks), dim3(256), 0, st, xp, gp, op, n);
      else hipLaunchKernelGGL((act_kernel<T, NOVA_ACT_GELU_ERF, false>), dim3(blocks), dim3(256), 0, st, xp, gp, op, n);
    } else {
      if (dy) hipLaunchKernelGGL((act_kernel<T, NOVA_ACT_SILU, true>), dim3(blocks), dim3(256), 0, st, xp, gp, op, n);
      else hipLaunchKernelGGL((act_kernel<T, NOVA_ACT_SILU, false>), dim3(blocks), dim3(256), 0, st, xp, gp, op, n);
    }
    return 0;
  });
  return check_launch("act");
}

}  // namespace nova

extern "C" int nova_act_fwd(const void* x, void* y, long long n, int kind, int dtype, void* stream) {
  using namespace nova;
  if (dtype != NOVA_F32 && dtype != NOVA_BF16 && dtype != NOVA_F16) return set_error(NOVA_ERR_ARG, "act_fwd: bad dtype %d", dtype);
  if (n > 0 && (!x || !y)) return set_error(NOVA_ERR_ARG, "act_fwd: null pointer");
  return act_pointwise(x, nullptr, y, (long)n, kind, dtype, (hipStream_t)stream);
}

extern "C" int nova_act_bwd(const void* x, const void* dy, void* dx, long long n, int kind, int dtype, void* stream) {
  using namespace nova;
  if (dtype != NOVA_F32 && dtype != NOVA_BF16 && dtype != NOVA_F16) return set_error(NOVA_ERR_ARG, "act_bwd: bad dtype %d", dtype);
  if (n > 0 && (!x || !dy || !dx)) return set_error(NOVA_ERR_ARG, "act_bwd: null pointer");
  return act_pointwise(x, dy, dx, (long)n, kind, dtype, (hipStream_t)stream);
}

extern "C" int nova_row_norm_bwd(const void* x, const void* dy, const float* gamma, const float* beta, const void* mod, long mod_ld,
                                 int scale_off, int shift_off, int gate_off, void* dx, void* dmod, float* dgamma_part,
                                 float* dbeta_part, int parts, long rows, int D, float eps, int dtype, void* stream) {
  using namespace nova;
  if (dtype != NOVA_F32 && dtype != NOVA_BF16 && dtype != NOVA_F16) return set_error(NOVA_ERR_ARG, "row_norm_bwd: bad dtype %d", dtype);
  if (rows > 0 && (!x || !dy || !dx)) return set_error(NOVA_ERR_ARG, "row_norm_bwd: null pointer");
  RowNormBwdArgs a{x, dy, gamma, beta, mod, mod_ld, mod ? scale_off : -1, mod ? shift_off : -1, mod ? gate_off : -1, dx, dmod,
                   dgamma_part, dbeta_part, rows, D, eps};
  return row_norm_bwd(a, parts, dtype, (hipStream_t)stream);
}
