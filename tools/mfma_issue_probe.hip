// Issue-port probe for the attention redesign (round 3): how many VALU fillers hide under one bf16 MFMA of either shape,
// at 1 / 2 / 4 waves per SIMD, on random operands (DVFS: MI355X_MICROARCH 'DVFS give-back' item 7).
//   hipcc -O3 --offload-arch=gfx950 mfma_issue_probe.hip -o mfma_issue_probe && ./mfma_issue_probe
// Per configuration prints shader cycles per MFMA-equivalent (32x32x16 = 1, 16x16x32 = 1/2), the in-kernel clock
// (s_memtime / s_memrealtime) and the TFLOP/s the chip delivered.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf8v;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) float f4v;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MFMAs are asm statements with dst == src C (the builtin form let the register allocator rotate the 16x16 accumulators
// through overlapping, misaligned ranges: 31 cycles per 16x16x32 instead of 16).
// SHAPE 0: 8 x v_mfma_f32_32x32x16_bf16 per iteration, SHAPE 1: 16 x v_mfma_f32_16x16x32_bf16 (same FLOPs, same 64x64..
// output footprint per wave: 32x32 tiles x 2 accumulators vs 16x16 x 8 accumulators). NADD plain VALU and NEXP
// transcendentals follow every 32x32x16-equivalent (i.e. are split over the two 16x16x32).
template <int SHAPE, int NADD, int NEXP>
__global__ __launch_bounds__(256) void probe(const bf8v* __restrict__ ab, float* __restrict__ out, long long* __restrict__ stamps, int iters) {
  const int lane = threadIdx.x & 63;
  bf8v a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = ab[(i * 64 + lane)]; b[i] = ab[((4 + i) * 64 + lane)]; }
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)lane * 0.001f + i;
  f16v acc32[2];
  f4v acc16[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc32[0][i] = 0.f; acc32[1][i] = 0.f; }
#pragma unroll
  for (int i = 0; i < 8; ++i) acc16[i] = f4v{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (SHAPE == 0) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc32[m & 1]) : "v"(a[m & 3]), "v"(b[(m >> 1) & 3]));
      } else {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc16[(2 * m) & 7]) : "v"(a[m & 3]), "v"(b[(m >> 1) & 3]));
      }
#pragma unroll
      for (int k = 0; k < (NADD + 1) / 2; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k & 3]) : "v"(f[4 + (k & 3)]));
#pragma unroll
      for (int k = 0; k < (NEXP + 1) / 2; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(f[4 + (k & 3)]));
      if (SHAPE == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc16[(2 * m + 1) & 7]) : "v"(a[(m + 1) & 3]), "v"(b[(m >> 1) & 3]));
#pragma unroll
      for (int k = 0; k < NADD / 2; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k & 3]) : "v"(f[4 + (k & 3)]));
#pragma unroll
      for (int k = 0; k < NEXP / 2; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(f[4 + (k & 3)]));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc32[0][i] + acc32[1][i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = r1 - r0;
  }
}

template <int SHAPE, int NADD, int NEXP>
static void run(const bf8v* ab, float* out, long long* stamps, int wps) {
  const int iters = 20000;
  const int blocks = 256 * wps;  // 256 CUs x wps blocks of 4 waves = wps waves per SIMD
  std::vector<long long> h(2 * blocks * 4);
  // warm (2 s of launches on random data would be the guide's protocol; a few 10-ms launches settle the clock enough to rank)
  for (int i = 0; i < 12; ++i) hipLaunchKernelGGL((probe<SHAPE, NADD, NEXP>), dim3(blocks), dim3(256), 0, 0, ab, out, stamps, iters);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  const int reps = 6;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<SHAPE, NADD, NEXP>), dim3(blocks), dim3(256), 0, 0, ab, out, stamps, iters);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(h.data(), stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  std::vector<double> cyc, clk;
  for (int w = 0; w < blocks * 4; ++w) { cyc.push_back((double)h[2 * w]); clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1); }
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  const double med_cyc = cyc[cyc.size() / 2], med_clk = clk[clk.size() / 2];
  const double mfma_eq = (double)iters * 8;  // 32x32x16 equivalents per wave
  const double flops = (double)blocks * 4 * mfma_eq * 2.0 * 32 * 32 * 16 * reps;
  printf("shape %s  waves/SIMD %d  fill/32x32-eq: %d add + %d exp   cycles per 32x32-eq per wave %.1f  (per SIMD %.1f)  clock %.2f GHz  %.0f TFLOP/s\n",
         SHAPE == 0 ? "32x32x16" : "16x16x32", wps, NADD, NEXP, med_cyc / mfma_eq, med_cyc / mfma_eq / wps, med_clk, flops / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

int main() {
  bf8v* ab; float* out; long long* stamps;
  const size_t nab = 8 * 64;
  std::vector<unsigned short> h(nab * 8);
  srand(1);
  for (auto& v : h) { float x = (float)rand() / RAND_MAX * 2.f - 1.f; unsigned u; memcpy(&u, &x, 4); v = (unsigned short)(u >> 16); }
  CK(hipMalloc(&ab, nab * 16)); CK(hipMemcpy(ab, h.data(), nab * 16, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
  CK(hipMalloc(&stamps, 256 * 8 * 4 * 2 * sizeof(long long)));
  for (int wps : {1, 2, 4}) {
    run<0, 0, 0>(ab, out, stamps, wps); run<1, 0, 0>(ab, out, stamps, wps);
    run<0, 4, 0>(ab, out, stamps, wps); run<1, 4, 0>(ab, out, stamps, wps);
    run<0, 6, 0>(ab, out, stamps, wps); run<1, 6, 0>(ab, out, stamps, wps);
    run<0, 4, 2>(ab, out, stamps, wps); run<1, 4, 2>(ab, out, stamps, wps);  // the attention loop's density at head_dim 64
    run<0, 8, 2>(ab, out, stamps, wps); run<1, 8, 2>(ab, out, stamps, wps);
    run<0, 0, 2>(ab, out, stamps, wps); run<1, 0, 2>(ab, out, stamps, wps);
  }
  return 0;
}
