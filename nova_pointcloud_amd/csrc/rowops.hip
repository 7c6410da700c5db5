// NOVA hot path: row-wise (HBM-bound) kernels between the GEMMs and around the AR loop.
// One wave (64 lanes) owns one row; reductions are wave shuffles; all loads/stores are 16-byte
// (f32) or 8-byte (bf16) vectors; statistics and arithmetic are always f32.
//
//   row_norm         LayerNorm family: post-norm residual of the ViT block
//                    (reference vision_transformer.py:78-82,91-92), AdaLN-Zero modulate
//                    (normalization.py:34-36), gate*LN+residual of the diffusion block
//                    (diffusion_mlp.py:52-53), final encoder LN with row gather
//                    (vision_transformer.py:146 + diffusion_mlp.py:93)
//   rope_table       cos/sin table of RotaryEmbed3D.get_func (embeddings.py:59-67)
//   embed_canvas     PatchEmbed conv (k = s = p) + MaskEmbed blend (+ abs-PE)
//                    (embeddings.py:160-166, 272-274, 90-91)
//   build_sequence   cat([c ; gather(x, prev_ids)]) (vision_transformer.py:133-136)
//   scatter_tokens   x_masked.scatter(1, prev_ids, x) (vision_transformer.py:141-143)
//   silu_add_rows, timestep_freq, patch_embed_rows, head_cfg_step, renorm_euler: diffusion-MLP glue and samplers
//                    (diffusion_mlp.py:65-75,89-99; guidance_scaler.py:86-87; scheduling_cfm.py:134-136)
#include "common.h"
#include "nova_internal.h"
#include "rownorm.h"

namespace nova {

template <typename T> struct Vec4;
template <> struct Vec4<float> {
  static __device__ __forceinline__ f4v load(const float* p) { return *reinterpret_cast<const f4v*>(p); }
  static __device__ __forceinline__ void store(float* p, f4v v) { *reinterpret_cast<f4v*>(p) = v; }
};
template <typename T> struct Vec4Half {  // bf16_t / f16_t
  static __device__ __forceinline__ f4v load(const T* p) {
    const u2v u = *reinterpret_cast<const u2v*>(p);
    const f2v a = Half16<T>::unpack(u[0]), b = Half16<T>::unpack(u[1]);
    return f4v{a[0], a[1], b[0], b[1]};
  }
  static __device__ __forceinline__ void store(T* p, f4v v) {
    u2v u = {Half16<T>::pack(v[0], v[1]), Half16<T>::pack(v[2], v[3])};
    *reinterpret_cast<u2v*>(p) = u;
  }
};
template <> struct Vec4<bf16_t> : Vec4Half<bf16_t> {};
template <> struct Vec4<f16_t> : Vec4Half<f16_t> {};

template <typename T, int NIT, bool HAS_RES, bool HAS_MOD>
__global__ __launch_bounds__(256) void row_norm_kernel(RowNormArgs a) {
  using C = Chunk<T>;
  constexpr int NV = C::N / 4;
  const int lane = threadIdx.x & 63;
  // XCD x walks one contiguous eighth of the rows, front to back or back to front (xcd_remap_dir, common.h)
  const long row = (long)xcd_remap_dir(blockIdx.x, gridDim.x, a.rev != 0) * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  C x[NIT];  // the finished row (rownorm.h: all of a row's HBM traffic is in flight before the two wave reductions)
  row_norm_compute<T, NIT, HAS_RES, HAS_MOD>(a, row, lane, x);
  T* out = static_cast<T*>(a.out) + row * a.D;
  const bool q8 = sizeof(T) == 2 && a.out8 != nullptr;  // also emit the row as e4m3 + scale (what the next fp8 GEMM reads)
  float amax = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int d = (it * 64 + lane) * C::N;
    if (d < a.D) {
      x[it].store(out + d);
      if (q8) {  // quantise what was STORED (the bf16-rounded row), so the fp8 copy equals quantize_rows_fp8(out)
        x[it] = x[it].rounded();
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fabsf(x[it].v[k][j]));
      }
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (q8) {
      amax = wave_max(amax);
      const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
      const float inv = 1.0f / sc;
      if (lane == 0) a.out8_scale[row] = sc;
      uint8_t* o8 = static_cast<uint8_t*>(a.out8) + row * a.D;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int d = (it * 64 + lane) * C::N;
        if (d < a.D) {
          u2v o;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            int w = 0;
            w = __builtin_amdgcn_cvt_pk_fp8_f32(x[it].v[k % NV][0] * inv, x[it].v[k % NV][1] * inv, w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(x[it].v[k % NV][2] * inv, x[it].v[k % NV][3] * inv, w, true);
            o[k] = (uint32_t)w;
          }
          *reinterpret_cast<u2v*>(o8 + d) = o;
        }
      }
    }
  }
}

template <typename T, int NIT>
static void launch_row_norm(const RowNormArgs& a, dim3 grid, hipStream_t st) {
  const bool r = a.res != nullptr, m = a.mod != nullptr;
  if (r && m) hipLaunchKernelGGL((row_norm_kernel<T, NIT, true, true>), grid, dim3(256), 0, st, a);
  else if (r) hipLaunchKernelGGL((row_norm_kernel<T, NIT, true, false>), grid, dim3(256), 0, st, a);
  else if (m) hipLaunchKernelGGL((row_norm_kernel<T, NIT, false, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((row_norm_kernel<T, NIT, false, false>), grid, dim3(256), 0, st, a);
}

int row_norm(const RowNormArgs& a, int dtype, hipStream_t st) {
  if (a.rows <= 0) return 0;
  const int vec = dtype_is16(dtype) ? 8 : 4;
  if (a.D % vec != 0 || a.D > 2048) return set_error(NOVA_ERR_SHAPE, "row_norm: D=%d unsupported (need D %% %d == 0, D <= 2048)", a.D, vec);
  if (a.mod && (a.mod_ld % vec || (a.scale_off >= 0 && (a.scale_off % vec || a.shift_off % vec || a.shift_off < 0)) ||
                (a.gate_off >= 0 && a.gate_off % vec)))
    return set_error(NOVA_ERR_SHAPE, "row_norm: modulation offsets must be multiples of %d", vec);
  if ((a.gamma == nullptr) != (a.beta == nullptr)) return set_error(NOVA_ERR_ARG, "row_norm: gamma/beta must come together");
  if (a.out8 && (dtype != NOVA_BF16 || !a.out8_scale)) return set_error(NOVA_ERR_ARG, "row_norm: the fp8 side output needs bf16 rows and a scale buffer");
  dim3 grid((unsigned)((a.rows + 3) / 4));
  RowNormArgs ar = a;
  ar.rev = walk_is_reverse() ? 1 : 0;
  // algorithmic bytes: read in (+res, +mod terms) and write out once
  const double esz = dtype_is16(dtype) ? 2.0 : 4.0;
  const int nmod = a.mod ? ((a.scale_off >= 0 ? 2 : 0) + (a.gate_off >= 0 ? 1 : 0)) : 0;
  ProfScope prof(PROF_ROWNORM, esz * a.rows * a.D * (2.0 + (a.res ? 1 : 0) + nmod), st);
  const int chunks = (a.D / vec + 63) / 64;  // 16-byte chunks per lane
  if (dtype_is16(dtype)) {
    dispatch_half(dtype, [&](auto tag) {
      if (chunks <= 2) launch_row_norm<decltype(tag), 2>(ar, grid, st);
      else launch_row_norm<decltype(tag), 4>(ar, grid, st);
      return 0;
    });
  } else {
    if (chunks <= 4) launch_row_norm<float, 4>(ar, grid, st);
    else launch_row_norm<float, 8>(ar, grid, st);
  }
  return check_launch("row_norm");
}

template <typename T, int NIT>
__global__ __launch_bounds__(256) void row_norm_chain_kernel(RowNormArgs a2, RowNormArgs a1, T* __restrict__ x_new_out) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a2.rows) return;
  RowChainRegs<T, NIT> g;
  row_chain_load<T, NIT>(a2, a1, row, lane, g);
  u4v xn[NIT];
  Chunk<T> y[NIT];
  row_chain_finish<T, NIT>(a2, a1, lane, g, xn, y);
  T* out = static_cast<T*>(a1.out) + row * a1.D;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int d = (it * 64 + lane) * 8;
    if (d < a1.D) {
      y[it].store(out + d);
      if (x_new_out) *reinterpret_cast<u4v*>(x_new_out + row * a1.D + d) = xn[it];
    }
  }
}

// 16-bit rows; a2 = gated affine norm with residual (its `out` is not used: x_new goes to x_new_out when that is given),
// a1 = scale / shift modulate writing a1.out. Equals row_norm(a2) then row_norm(a1) bit for bit (rownorm.h).
int row_norm_chain(const RowNormArgs& a2, const RowNormArgs& a1, void* x_new_out, int dtype, hipStream_t st) {
  if (a2.rows <= 0) return 0;
  if (!dtype_is16(dtype)) return set_error(NOVA_ERR_ARG, "row_norm_chain: 16-bit rows only");
  if (a2.D != a1.D || a2.rows != a1.rows || a2.D % 8 || a2.D > 2048) return set_error(NOVA_ERR_SHAPE, "row_norm_chain: bad shapes");
  if (!a2.in || !a2.res || !a2.mod || a2.gate_off < 0 || a2.scale_off >= 0 || a2.gather || !a1.mod || a1.scale_off < 0 || a1.shift_off < 0 ||
      a1.gate_off >= 0 || a1.gamma || a1.res || a1.gather || !a1.out || (a2.gamma == nullptr) != (a2.beta == nullptr))
    return set_error(NOVA_ERR_ARG, "row_norm_chain: first norm = gate + residual, second = scale / shift modulate");
  if (a2.mod_ld % 8 || a2.gate_off % 8 || a1.mod_ld % 8 || a1.scale_off % 8 || a1.shift_off % 8)
    return set_error(NOVA_ERR_SHAPE, "row_norm_chain: modulation offsets must be multiples of 8");
  const dim3 grid((unsigned)((a2.rows + 3) / 4));
  ProfScope prof(PROF_ROWNORM, 2.0 * a2.rows * a2.D * 6.0, st);
  dispatch_half(dtype, [&](auto tag) {
    using T = decltype(tag);
    if ((a2.D / 8 + 63) / 64 <= 2) hipLaunchKernelGGL((row_norm_chain_kernel<T, 2>), grid, dim3(256), 0, st, a2, a1, static_cast<T*>(x_new_out));
    else hipLaunchKernelGGL((row_norm_chain_kernel<T, 4>), grid, dim3(256), 0, st, a2, a1, static_cast<T*>(x_new_out));
    return 0;
  });
  return check_launch("row_norm_chain");
}

// ------------------------------------------------------------------------------------------
// RoPE table: out[b][l][p] = (cos, sin)(pos_axis(p) * inv_freq[p]); rows l < pad are position 0.
// pos [n_pos, 3] (t,h,w) f32 is batch independent (the reference expands one grid over the batch);
// ids [nb, n_tok] int64 selects token -> position (gathered first-half sequence) or null (identity).
__global__ void rope_table_kernel(const float* __restrict__ pos, const long long* __restrict__ ids,
                                  float* __restrict__ table, int nb, int pad, int n_tok, int n_pos, int hd,
                                  const float* __restrict__ inv_freq) {
  const int half = hd / 2;
  const long total = (long)nb * (pad + n_tok) * half;
  const int n0 = (hd / 8) / 2, n1 = ((hd - hd / 8) / 2) / 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int p = (int)(i % half);
    const long bl = i / half;
    const int l = (int)(bl % (pad + n_tok)), b = (int)(bl / (pad + n_tok));
    float cs = 1.f, sn = 0.f;
    if (l >= pad) {
      const int tok = l - pad;
      long idx = ids ? ids[(long)b * n_tok + tok] : tok;
      idx = idx < 0 ? 0 : (idx >= n_pos ? n_pos - 1 : idx);
      const int axis = p < n0 ? 0 : (p < n0 + n1 ? 1 : 2);
      const float ang = pos[idx * 3 + axis] * inv_freq[p];
      cs = cosf(ang);
      sn = sinf(ang);
    }
    table[2 * i] = cs;
    table[2 * i + 1] = sn;
  }
}

int rope_table(const float* pos, const long long* ids, float* table, int nb, int pad, int n_tok, int n_pos, int hd,
               const float* inv_freq, hipStream_t st) {
  ProfScope prof(PROF_TOKENS, 0.0, st);
  const long total = (long)nb * (pad + n_tok) * (hd / 2);
  if (total <= 0) return 0;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(rope_table_kernel, dim3(blocks), dim3(256), 0, st, pos, ids, table, nb, pad, n_tok, n_pos, hd, inv_freq);
  return check_launch("rope_table");
}

// ------------------------------------------------------------------------------------------
// z0[b][n][:] = (W x[b][n] + bias) * (1 - m) + mask_token * m   (+ pos_embed[n][:])
// x = patchified canvas [B, N, P] f32, W [D, P] in patchified order, m = mask[b][n] in {0, 1}.
template <typename T>
__global__ __launch_bounds__(256) void embed_canvas_kernel(const float* __restrict__ canvas, const float* __restrict__ mask,
                                                           const T* __restrict__ w, const float* __restrict__ bias,
                                                           const T* __restrict__ mask_token, const T* __restrict__ pos_embed,
                                                           T* __restrict__ z0, int N, int P, int D) {
  __shared__ float xs[64];
  const long tok = blockIdx.x;
  if (threadIdx.x < P) xs[threadIdx.x] = canvas[tok * P + threadIdx.x];
  __syncthreads();
  const float m = mask[tok];
  const int n = (int)(tok % N);
  for (int d = threadIdx.x; d < D; d += 256) {
    float e = bias[d];
    for (int p = 0; p < P; ++p) e += xs[p] * to_f<T>(w[(long)d * P + p]);
    float z = e * (1.0f - m) + to_f<T>(mask_token[d]) * m;
    if (pos_embed) z += to_f<T>(pos_embed[(long)n * D + d]);
    z0[tok * D + d] = from_f<T>(z);
  }
}

int embed_canvas(const float* canvas, const float* mask, const void* w, const float* bias, const void* mask_token,
                 const void* pos_embed, void* z0, int B, int N, int P, int D, int dtype, hipStream_t st) {
  ProfScope prof(PROF_TOKENS, 0.0, st);
  if ((long)B * N <= 0) return 0;
  if (P > 64 || P <= 0) return set_error(NOVA_ERR_SHAPE, "embed_canvas: patch vector length %d unsupported (1..64)", P);
  dim3 grid((unsigned)((long)B * N)), block(256);
  dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(embed_canvas_kernel<T>, grid, block, 0, st, canvas, mask, (const T*)w, bias, (const T*)mask_token, (const T*)pos_embed, (T*)z0, N, P, D);
    return 0;
  });
  return check_launch("embed_canvas");
}

// ------------------------------------------------------------------------------------------
// x[s][l] = l < Lp ? prefix[s][l] : tokens[s % B][ids ? ids[s % B][l - Lp] : l - Lp]
// prefix rows of sequence s start at prefix + s * prefix_seq_rows * D; tokens of batch b at
// tokens + b * tok_batch_rows * D (0 = shared by the whole batch).
template <typename T>
__global__ __launch_bounds__(256) void build_sequence_kernel(const T* __restrict__ prefix, long prefix_seq_rows,
                                                             const T* __restrict__ tokens, long tok_batch_rows,
                                                             const long long* __restrict__ ids, T* __restrict__ x,
                                                             long rows, int B, int Lp, int n_sel, int D) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int L = Lp + n_sel;
  const long s = row / L;
  const int l = (int)(row - s * L);
  const T* src;
  if (l < Lp) {
    src = prefix + (s * prefix_seq_rows + l) * D;
  } else {
    const int b = (int)(s % B);
    const long j = ids ? (long)ids[(long)b * n_sel + (l - Lp)] : (long)(l - Lp);
    src = tokens + ((long)b * tok_batch_rows + j) * D;
  }
  T* dst = x + row * D;
  constexpr int V = 16 / sizeof(T);
  for (int d = lane * V; d < D; d += 64 * V) *reinterpret_cast<u4v*>(dst + d) = *reinterpret_cast<const u4v*>(src + d);
}

int build_sequence(const void* prefix, long prefix_seq_rows, const void* tokens, long tok_batch_rows,
                   const long long* ids, void* x, int S, int B, int Lp, int n_sel, int D, int dtype, hipStream_t st) {
  ProfScope prof(PROF_TOKENS, 0.0, st);
  const long rows = (long)S * (Lp + n_sel);
  if (rows <= 0) return 0;
  const int V = dtype_is16(dtype) ? 8 : 4;
  if (D % V) return set_error(NOVA_ERR_SHAPE, "build_sequence: D must be a multiple of %d", V);
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(build_sequence_kernel<T>, grid, block, 0, st, (const T*)prefix, prefix_seq_rows, (const T*)tokens, tok_batch_rows, ids, (T*)x, rows, B, Lp, n_sel, D);
    return 0;
  });
  return check_launch("build_sequence");
}

// x2[s][Lp + ids[s % B][j]] = x1[s][Lp + j]   (x1 rows per sequence: Lp + n_prev, x2: Lp + N)
template <typename T>
__global__ __launch_bounds__(256) void scatter_tokens_kernel(const T* __restrict__ x1, const long long* __restrict__ ids,
                                                             T* __restrict__ x2, long rows, int B, int Lp, int N,
                                                             int n_prev, int D) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long s = row / n_prev;
  const int j = (int)(row - s * n_prev);
  const int b = (int)(s % B);
  const long tgt = (long)ids[(long)b * n_prev + j];
  const T* src = x1 + (s * (Lp + n_prev) + Lp + j) * D;
  T* dst = x2 + (s * (Lp + N) + Lp + tgt) * D;
  constexpr int V = 16 / sizeof(T);
  for (int d = lane * V; d < D; d += 64 * V) *reinterpret_cast<u4v*>(dst + d) = *reinterpret_cast<const u4v*>(src + d);
}

int scatter_tokens(const void* x1, const long long* ids, void* x2, int S, int B, int Lp, int N, int n_prev, int D,
                   int dtype, hipStream_t st) {
  ProfScope prof(PROF_TOKENS, 0.0, st);
  const long rows = (long)S * n_prev;
  if (rows <= 0) return 0;
  const int V = dtype_is16(dtype) ? 8 : 4;
  if (D % V) return set_error(NOVA_ERR_SHAPE, "scatter_tokens: D must be a multiple of %d", V);
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(scatter_tokens_kernel<T>, grid, block, 0, st, (const T*)x1, ids, (T*)x2, rows, B, Lp, N, n_prev, D);
    return 0;
  });
  return check_launch("scatter_tokens");
}

// ------------------------------------------------------------------------------------------
// out[r][:] = silu(a[r][:] + vec[:])     (SiLU(z) in front of every AdaLN projection, z = cond + time)
template <typename T>
__global__ __launch_bounds__(256) void silu_add_rows_kernel(const T* __restrict__ a, const T* __restrict__ vec,
                                                            T* __restrict__ out, long total4, int D) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 4;
    const int d = (int)(e % D);
    f4v x = Vec4<T>::load(a + e);
    if (vec) x = x + Vec4<T>::load(vec + d);
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = silu(x[j]);
    Vec4<T>::store(out + e, x);
  }
}

// ------------------------------------------------------------------------------------------
// MX-fp8 path (BASELINE configs[4]): per-row dynamic quantisation of bf16 activations to OCP e4m3.
//   scale[r] = max|x[r][:]| / 448 (1 for an all-zero row), out[r][d] = e4m3(x[r][d] / scale[r])
// One wave per row, 16-byte loads, 8-byte stores; D % 8 == 0, D <= 8192.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const bf16_t* __restrict__ x, uint8_t* __restrict__ out,
                                                                float* __restrict__ scale, long rows, int D) {
  using C = Chunk<bf16_t>;
  constexpr int NIT = 16;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* in = x + row * D;
  C v[NIT];
  float amax = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int d = (it * 64 + lane) * 8;
    if (d < D) {
      v[it] = C::load(in + d);
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fabsf(v[it].v[k][j]));
    }
  }
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int d = (it * 64 + lane) * 8;
    if (d < D) {
      u2v o;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[it].v[k][0] * inv, v[it].v[k][1] * inv, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[it].v[k][2] * inv, v[it].v[k][3] * inv, w, true);
        o[k] = (uint32_t)w;
      }
      *reinterpret_cast<u2v*>(out + row * D + d) = o;
    }
  }
}

int quantize_rows_fp8(const void* x, void* out, float* scale, long rows, int D, hipStream_t st) {
  ProfScope prof(PROF_TOKENS, 0.0, st);
  if (rows <= 0) return 0;
  if (D % 8 != 0 || D > 8192) return set_error(NOVA_ERR_SHAPE, "quantize_rows_fp8: need D %% 8 == 0 and D <= 8192 (got %d)", D);
  hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const bf16_t*)x, (uint8_t*)out, scale,
                     rows, D);
  return check_launch("quantize_rows_fp8");
}

// out[i * rows + r][:] = silu(a[r][:] + vecs[i][:]) for i < nvec: the SiLU(z + t_i) rows of ALL diffusion steps at once, so that
// their AdaLN projections become one large GEMM (M = steps * rows) instead of one small GEMM per step.
template <typename T>
__global__ __launch_bounds__(256) void silu_add_steps_kernel(const T* __restrict__ a, const T* __restrict__ vecs, T* __restrict__ out,
                                                             long per_step4, long total4, int D) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long step = i / per_step4, e = (i - step * per_step4) * 4;
    const int d = (int)(e % D);
    f4v x = Vec4<T>::load(a + e) + Vec4<T>::load(vecs + step * D + d);
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = silu(x[j]);
    Vec4<T>::store(out + i * 4, x);
  }
}

int silu_add_steps(const void* a, const void* vecs, void* out, long rows, int nvec, int D, int dtype, hipStream_t st) {
  ProfScope prof(PROF_DECODER, 0.0, st);
  if (rows <= 0 || nvec <= 0) return 0;
  if (D % 4) return set_error(NOVA_ERR_SHAPE, "silu_add_steps: D %% 4 != 0");
  const long per4 = rows * D / 4, total4 = per4 * nvec;
  const int blocks = (int)((total4 + 255) / 256 > 16384 ? 16384 : (total4 + 255) / 256);
  dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(silu_add_steps_kernel<T>, dim3(blocks), dim3(256), 0, st, (const T*)a, (const T*)vecs, (T*)out, per4, total4, D);
    return 0;
  });
  return check_launch("silu_add_steps");
}

int silu_add_rows(const void* a, const void* rowvec, void* out, long rows, int D, int dtype, hipStream_t st) {
  ProfScope prof(PROF_DECODER, 0.0, st);
  if (rows <= 0) return 0;
  if (D % 4) return set_error(NOVA_ERR_SHAPE, "silu_add_rows: D %% 4 != 0");
  const long total4 = rows * D / 4;
  const int blocks = (int)((total4 + 255) / 256 > 8192 ? 8192 : (total4 + 255) / 256);
  dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(silu_add_rows_kernel<T>, dim3(blocks), dim3(256), 0, st, (const T*)a, (const T*)rowvec, (T*)out, total4, D);
    return 0;
  });
  return check_launch("silu_add_rows");
}

// out[i][:] = [cos(t_i f_k) (k < F/2) ; sin(t_i f_k)]   f = time_freq table of the reference
template <typename T>
__global__ void timestep_freq_kernel(const float* __restrict__ t, const float* __restrict__ freq, T* __restrict__ out,
                                     int n, int F) {
  const int half = F / 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * half; i += gridDim.x * blockDim.x) {
    const int row = i / half, kf = i % half;
    const float ang = t[row] * freq[kf];
    out[(long)row * F + kf] = from_f<T>(cosf(ang));
    out[(long)row * F + half + kf] = from_f<T>(sinf(ang));
  }
}

int timestep_freq(const float* t, const float* freq, void* out, int n, int freq_dim, int dtype, hipStream_t st) {
  ProfScope prof(PROF_DECODER, 0.0, st);
  if (n <= 0) return 0;
  const int total = n * (freq_dim / 2);
  dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(timestep_freq_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, st, t, freq, (T*)out, n, freq_dim);
    return 0;
  });
  return check_launch("timestep_freq");
}

// u[s][j][:] = W x[s % B][j] + bias   (decoder patch embed of the noisy tokens being predicted)
template <typename T>
__global__ __launch_bounds__(256) void patch_embed_rows_kernel(const float* __restrict__ x, const T* __restrict__ w,
                                                               const float* __restrict__ bias, T* __restrict__ out,
                                                               int B, int n, int P, int D) {
  __shared__ float xs[64];
  const long row = blockIdx.x;  // s * n + j
  const long s = row / n;
  const int j = (int)(row - s * n), b = (int)(s % B);
  if (threadIdx.x < P) xs[threadIdx.x] = x[((long)b * n + j) * P + threadIdx.x];
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) {
    float e = bias[d];
    for (int p = 0; p < P; ++p) e += xs[p] * to_f<T>(w[(long)d * P + p]);
    out[row * D + d] = from_f<T>(e);
  }
}

int patch_embed_rows(const float* x, const void* w, const float* bias, void* out, int S, int B, int n, int P, int D,
                     int dtype, hipStream_t st) {
  ProfScope prof(PROF_DECODER, 0.0, st);
  if ((long)S * n <= 0) return 0;
  if (P > 64 || P <= 0) return set_error(NOVA_ERR_SHAPE, "patch_embed_rows: patch vector length %d unsupported", P);
  dim3 grid((unsigned)((long)S * n)), block(256);
  dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(patch_embed_rows_kernel<T>, grid, block, 0, st, x, (const T*)w, bias, (T*)out, B, n, P, D);
    return 0;
  });
  return check_launch("patch_embed_rows");
}

// Head projection + guidance combine + one sampler step for the n tokens of this AR step:
//   pc = Wh h[b][j] + bh, pu = Wh h[B + b][j] + bh, (3-pass: p3 = Wh h[2B + b][j] + bh)
//   2-pass (guidance_scaler.py:86-87):            v = pu + g (pc - pu)
//   3-pass, image guidance (:78-81):              v = pu + g (pc - p3) + e (p3 - pu)
//   3-pass, spatiotemporal guidance (:82-85):     v = pu + g (pc - pu) + e (pc - p3)
//   x0 = clamp(kx x + kv v, +-clip);  x <- c0 x0 + cx x + sigma noise
// Flow-matching Euler (scheduling_cfm.py:134-136): kx = 0, kv = 1, no clip, c0 = dt, cx = 1, sigma = 0.
// DDPM ancestral step (scheduling_ddpm.py:236-316): kx, kv from the prediction type, c0 / cx the posterior-mean
// coefficients, sigma the posterior std, noise the fresh gaussian of that step.
// defer != 0 (guidance renorm): do not touch x; write the renormalised part of v (everything but the `e` term) to
// vhat[b][j][:], pc to cond[b][j][:] and the `e` term to extra[b][j][:] (3-pass only) instead.
template <typename T>
__global__ __launch_bounds__(256) void head_cfg_step_kernel(const T* __restrict__ h, const T* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ x,
                                                            const float* __restrict__ noise, float* __restrict__ vhat,
                                                            float* __restrict__ cond, float* __restrict__ extra, long rows, int B,
                                                            int n, int P, int D, SamplerStep sp, int defer) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // b * n + j
  if (row >= rows) return;
  const int cfg = sp.guidance > 1.0f;
  const int kind = cfg ? sp.extra_kind : 0;
  const T* hc = h + row * D;
  const T* hu = h + ((long)B * n + row) * D;
  const T* h3 = h + ((long)2 * B * n + row) * D;
  for (int p = 0; p < P; ++p) {
    float ac = 0.f, au = 0.f, a3 = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
      const f4v wv = Vec4<T>::load(w + (long)p * D + d);
      const f4v c4 = Vec4<T>::load(hc + d);
      ac += (wv[0] * c4[0] + wv[1] * c4[1]) + (wv[2] * c4[2] + wv[3] * c4[3]);
      if (cfg) {
        const f4v u4 = Vec4<T>::load(hu + d);
        au += (wv[0] * u4[0] + wv[1] * u4[1]) + (wv[2] * u4[2] + wv[3] * u4[3]);
      }
      if (kind) {
        const f4v t4 = Vec4<T>::load(h3 + d);
        a3 += (wv[0] * t4[0] + wv[1] * t4[1]) + (wv[2] * t4[2] + wv[3] * t4[3]);
      }
    }
    ac = wave_sum(ac) + bias[p];
    float vv = ac, ex = 0.f;
    if (cfg) {
      au = wave_sum(au) + bias[p];
      if (kind) a3 = wave_sum(a3) + bias[p];
      if (kind == 1) {
        vv = au + (ac - a3) * sp.guidance;
        ex = (a3 - au) * sp.extra_scale;
      } else {
        vv = au + (ac - au) * sp.guidance;
        if (kind == 2) ex = (ac - a3) * sp.extra_scale;
      }
    }
    if (lane == 0) {
      const long e = row * P + p;
      if (defer) {
        vhat[e] = vv;
        cond[e] = ac;
        if (extra) extra[e] = ex;
      } else {
        vv += ex;
        const float xo = x[e];
        float x0 = sp.kx * xo + sp.kv * vv;
        if (sp.clip > 0.f) x0 = fminf(fmaxf(x0, -sp.clip), sp.clip);
        float xn = sp.c0 * x0 + sp.cx * xo;
        if (noise) xn += sp.sigma * noise[e];
        x[e] = xn;
      }
    }
  }
}

int head_cfg_step(const void* h, const void* w, const float* bias, float* x, const float* noise, float* vhat, float* cond,
                  float* extra, int B, int n, int P, int D, const SamplerStep& sp, int defer, int dtype, hipStream_t st) {
  ProfScope prof(PROF_DECODER, 0.0, st);
  const long rows = (long)B * n;
  if (rows <= 0) return 0;
  if (D % 4) return set_error(NOVA_ERR_SHAPE, "head_cfg_step: D %% 4 != 0");
  if (defer && (!vhat || !cond)) return set_error(NOVA_ERR_ARG, "head_cfg_step: deferred mode needs vhat and cond buffers");
  if (sp.extra_kind < 0 || sp.extra_kind > 2) return set_error(NOVA_ERR_ARG, "head_cfg_step: extra_kind must be 0, 1 or 2");
  if (defer && sp.extra_kind && sp.guidance > 1.0f && !extra) return set_error(NOVA_ERR_ARG, "head_cfg_step: deferred 3-pass mode needs the extra buffer");
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(head_cfg_step_kernel<T>, grid, block, 0, st, (const T*)h, (const T*)w, bias, x, noise, vhat, cond, extra, rows, B, n, P, D, sp, defer);
    return 0;
  });
  return check_launch("head_cfg_step");
}

// Guidance renormalisation (guidance_scaler.py:67-72) for the flow-matching Euler step. The reference takes the
// norms over ALL N rows of a sample: the n predicted rows plus the rows that merely echo the current x_t. The
// echo rows never leave this kernel's view: their squared norm E_b is a scalar that evolves with the same step
// (x_echo <- x_echo (1 + dt ratio)), so one block per sample reduces its n*P predicted values deterministically:
//   ratio = clamp(sqrt((sum c^2 + E) / (sum v^2 + E)), renorm, 1);  x += dt (ratio v + extra);  E *= (1 + dt ratio)^2
// `extra` (3-pass guidance only, else null) is the term the reference adds AFTER the renormalisation
// (guidance_scaler.py:80-81,84-85); it vanishes on the echo rows, where all passes return x_t.
__global__ __launch_bounds__(256) void renorm_euler_kernel(float* __restrict__ x, const float* __restrict__ vhat,
                                                          const float* __restrict__ cond, const float* __restrict__ extra,
                                                          float* __restrict__ echo, int nP, float dt, float renorm) {
  __shared__ float red[2][4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* vb = vhat + (long)b * nP;
  const float* cb = cond + (long)b * nP;
  float sv = 0.f, sc = 0.f;
  for (int i = threadIdx.x; i < nP; i += 256) {
    sv += vb[i] * vb[i];
    sc += cb[i] * cb[i];
  }
  sv = wave_sum(sv);
  sc = wave_sum(sc);
  if (lane == 0) { red[0][wv] = sv; red[1][wv] = sc; }
  __syncthreads();
  const float E = echo[b];
  const float nx = sqrtf((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]) + E);
  const float nc = sqrtf((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]) + E);
  const float ratio = fminf(fmaxf(nc / nx, renorm), 1.0f);
  for (int i = threadIdx.x; i < nP; i += 256) x[(long)b * nP + i] += dt * (ratio * vb[i] + (extra ? extra[(long)b * nP + i] : 0.f));
  __syncthreads();
  if (threadIdx.x == 0) {
    const float f = 1.0f + dt * ratio;
    echo[b] = E * f * f;
  }
}

// Guidance renormalisation for ANY sampler step (the ancestral DDPM step: the echo rows take noise and a clamp, so their squared
// norm is no longer a scalar that evolves by itself): the rows that merely echo x_t are carried explicitly - echo[b][0:eP], updated
// here with the same step as the predicted rows. For an echo row every guidance pass returns x_t, so v-hat = cond = x_t and the
// `extra` term vanishes (guidance_scaler.py:74-87 on rows where the decoder scatters nothing: diffusion_mlp.py).
//   ratio = clamp(sqrt((sum cond^2 + sum echo^2) / (sum vhat^2 + sum echo^2)), renorm, 1)         (guidance_scaler.py:67-72)
//   predicted rows: v = ratio vhat + extra;  echo rows: v = ratio x_t;  both: x0 = clamp(kx x + kv v), x <- c0 x0 + cx x + sigma noise
// echo_only != 0 (guidance switched off for this step, guidance_trunc): the predicted rows were stepped by head_cfg_step already, the
// echo rows take the step with ratio 1. One block per sample, fixed summation order.
__global__ __launch_bounds__(256) void renorm_step_kernel(float* __restrict__ x, const float* __restrict__ vhat, const float* __restrict__ cond,
                                                         const float* __restrict__ extra, const float* __restrict__ noise,
                                                         float* __restrict__ echo, const float* __restrict__ echo_noise, int nP, int eP,
                                                         SamplerStep sp, float renorm, int echo_only) {
  __shared__ float red[3][4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* eb = echo + (long)b * eP;
  float ratio = 1.0f;
  if (!echo_only) {
    const float* vb = vhat + (long)b * nP;
    const float* cb = cond + (long)b * nP;
    float sv = 0.f, sc = 0.f, se = 0.f;
    for (int i = threadIdx.x; i < nP; i += 256) {
      sv += vb[i] * vb[i];
      sc += cb[i] * cb[i];
    }
    for (int i = threadIdx.x; i < eP; i += 256) se += eb[i] * eb[i];
    sv = wave_sum(sv);
    sc = wave_sum(sc);
    se = wave_sum(se);
    if (lane == 0) { red[0][wv] = sv; red[1][wv] = sc; red[2][wv] = se; }
    __syncthreads();
    const float E = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    const float nx = sqrtf((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]) + E);
    const float nc = sqrtf((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]) + E);
    ratio = fminf(fmaxf(nc / nx, renorm), 1.0f);
    __syncthreads();  // every thread has read the echo rows before any of them is overwritten below
  }
  auto step = [&](float xo, float v, const float* nz, long e) {
    float x0 = sp.kx * xo + sp.kv * v;
    if (sp.clip > 0.f) x0 = fminf(fmaxf(x0, -sp.clip), sp.clip);
    float xn = sp.c0 * x0 + sp.cx * xo;
    if (nz) xn += sp.sigma * nz[e];
    return xn;
  };
  if (!echo_only) {
    for (int i = threadIdx.x; i < nP; i += 256) {
      const long e = (long)b * nP + i;
      x[e] = step(x[e], ratio * vhat[e] + (extra ? extra[e] : 0.f), noise, e);
    }
  }
  for (int i = threadIdx.x; i < eP; i += 256) {
    const long e = (long)b * eP + i;
    echo[e] = step(echo[e], ratio * echo[e], echo_noise, e);
  }
}

int renorm_step(float* x, const float* vhat, const float* cond, const float* extra, const float* noise, float* echo, const float* echo_noise,
                int B, int nP, int eP, const SamplerStep& sp, float renorm, int echo_only, hipStream_t st) {
  ProfScope prof(PROF_DECODER, 0.0, st);
  if (B <= 0 || (nP <= 0 && eP <= 0)) return 0;
  if (!echo || (!echo_only && (!x || !vhat || !cond))) return set_error(NOVA_ERR_ARG, "renorm_step: null pointer");
  hipLaunchKernelGGL(renorm_step_kernel, dim3(B), dim3(256), 0, st, x, vhat, cond, extra, noise, echo, echo_noise, nP, eP, sp, renorm, echo_only);
  return check_launch("renorm_step");
}

__global__ void scale_vector_kernel(float* v, int n, float f) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] *= f;
}

int scale_vector(float* v, int n, float f, hipStream_t st) {
  ProfScope prof(PROF_DECODER, 0.0, st);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(scale_vector_kernel, dim3((n + 255) / 256), dim3(256), 0, st, v, n, f);
  return check_launch("scale_vector");
}

int renorm_euler(float* x, const float* vhat, const float* cond, const float* extra, float* echo, int B, int n, int P, float dt,
                 float renorm, hipStream_t st) {
  ProfScope prof(PROF_DECODER, 0.0, st);
  if (B <= 0 || n <= 0) return 0;
  hipLaunchKernelGGL(renorm_euler_kernel, dim3(B), dim3(256), 0, st, x, vhat, cond, extra, echo, n * P, dt, renorm);
  return check_launch("renorm_euler");
}

// ------------------------------------------------------------------------------------------
// KV cache of the conditioning encoder for multi-frame generation (vision_transformer.py:55-60: k, v of the new
// rows are concatenated behind the cached ones). The fused QKV GEMM leaves [S*Lq, 3D] rows (q | k | v); the k | v
// part of row (s, l) is copied to cache[s][base + l][0:2D]   (cache: [S][cap][2D] per block).
template <typename T>
__global__ __launch_bounds__(256) void kv_append_kernel(const T* __restrict__ qkv, T* __restrict__ cache, long rows, int Lq,
                                                        int D, long cap, long base) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long s = row / Lq, l = row - s * Lq;
  const T* src = qkv + row * 3 * D + D;
  T* dst = cache + (s * cap + base + l) * 2 * D;
  constexpr int V = 16 / sizeof(T);
  for (int d = lane * V; d < 2 * D; d += 64 * V) *reinterpret_cast<u4v*>(dst + d) = *reinterpret_cast<const u4v*>(src + d);
}

int kv_append(const void* qkv, void* cache, int S, int Lq, int D, long cap, long base, int dtype, hipStream_t st) {
  ProfScope prof(PROF_TOKENS, 0.0, st);
  const long rows = (long)S * Lq;
  if (rows <= 0) return 0;
  const int V = dtype_is16(dtype) ? 8 : 4;
  if (D % V) return set_error(NOVA_ERR_SHAPE, "kv_append: D must be a multiple of %d", V);
  if (base < 0 || base + Lq > cap) return set_error(NOVA_ERR_SHAPE, "kv_append: rows [%ld, %ld) exceed the cache capacity %ld", base, base + Lq, cap);
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(kv_append_kernel<T>, grid, block, 0, st, (const T*)qkv, (T*)cache, rows, Lq, D, cap, base);
    return 0;
  });
  return check_launch("kv_append");
}

// out[r][:] = x[r][:] * (1 + mod[r][0:D]) + mod[r][D:2D]   (AdaLayerNorm with eps=None, i.e. no normalisation: the
// frame mixer of the conditioning encoder, normalization.py:41-46 + transformer_nova.py:87-89)
template <typename T>
__global__ __launch_bounds__(256) void modulate_rows_kernel(const T* __restrict__ x, const T* __restrict__ mod, T* __restrict__ out,
                                                            long total4, int D) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 4;
    const long r = e / D;
    const int d = (int)(e - r * D);
    const f4v xv = Vec4<T>::load(x + e);
    const f4v sc = Vec4<T>::load(mod + r * 2 * D + d), sh = Vec4<T>::load(mod + r * 2 * D + D + d);
    Vec4<T>::store(out + e, xv * (1.0f + sc) + sh);
  }
}

int modulate_rows(const void* x, const void* mod, void* out, long rows, int D, int dtype, hipStream_t st) {
  ProfScope prof(PROF_TOKENS, 0.0, st);
  if (rows <= 0) return 0;
  if (D % 4) return set_error(NOVA_ERR_SHAPE, "modulate_rows: D %% 4 != 0");
  const long total4 = rows * D / 4;
  const int blocks = (int)((total4 + 255) / 256 > 8192 ? 8192 : (total4 + 255) / 256);
  dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(modulate_rows_kernel<T>, dim3(blocks), dim3(256), 0, st, (const T*)x, (const T*)mod, (T*)out, total4, D);
    return 0;
  });
  return check_launch("modulate_rows");
}

}  // namespace nova
