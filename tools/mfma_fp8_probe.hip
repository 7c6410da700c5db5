// Issue probe for the fp8 GEMM mode (round 3): what a bare loop of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 operands, unit block
// scales - the instruction gemm256p_kernel<fp8, *> runs) holds on random operands, beside the bf16 16x16x32 loop, at 1 and 2
// waves per SIMD, with 0 / 2 / 4 plain VALU fillers per MFMA. Prints shader cycles per MFMA, the in-kernel clock and TFLOP/s.
//   hipcc -O3 --offload-arch=gfx950 mfma_fp8_probe.hip -o mfma_fp8_probe && ./mfma_fp8_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i8v;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8v;
typedef __attribute__((ext_vector_type(4))) float f4v;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// KIND 0: fp8 scaled 16x16x128 (8 operand registers each side), KIND 1: bf16 16x16x32 (4 each side). 8 accumulators, 8 MFMAs per
// iteration, NADD v_add_f32 after each.
template <int KIND, int NADD>
__global__ __launch_bounds__(256) void probe(const i8v* __restrict__ ab, float* __restrict__ out, long long* __restrict__ stamps, int iters) {
  const int lane = threadIdx.x & 63;
  i8v a[4], b[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = ab[i * 64 + lane];
#pragma unroll
  for (int i = 0; i < 2; ++i) b[i] = ab[(4 + i) * 64 + lane];
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)lane * 0.001f + i;
  f4v acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f4v{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (KIND == 0) {
        acc[m] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b[m >> 2], a[m & 3], acc[m], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      } else {
        const bf8v av = __builtin_bit_cast(bf8v, __builtin_shufflevector(a[m & 3], a[m & 3], 0, 1, 2, 3));
        const bf8v bv = __builtin_bit_cast(bf8v, __builtin_shufflevector(b[m >> 2], b[m >> 2], 0, 1, 2, 3));
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv, av, acc[m], 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < NADD; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k & 3]) : "v"(f[4 + (k & 3)]));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = r1 - r0;
  }
}

template <int KIND, int NADD>
static void run(const i8v* ab, float* out, long long* stamps, int wps) {
  const int iters = KIND == 0 ? 10000 : 20000;
  const int blocks = 256 * wps;
  std::vector<long long> h(2 * blocks * 4);
  for (int i = 0; i < 12; ++i) hipLaunchKernelGGL((probe<KIND, NADD>), dim3(blocks), dim3(256), 0, 0, ab, out, stamps, iters);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  const int reps = 6;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<KIND, NADD>), dim3(blocks), dim3(256), 0, 0, ab, out, stamps, iters);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(h.data(), stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  std::vector<double> cyc, clk;
  for (int w = 0; w < blocks * 4; ++w) { cyc.push_back((double)h[2 * w]); clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1); }
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  const double med_cyc = cyc[cyc.size() / 2], med_clk = clk[clk.size() / 2];
  const double n = (double)iters * 8;
  const double flops = (double)blocks * 4 * n * 2.0 * 16 * 16 * (KIND == 0 ? 128 : 32) * reps;
  printf("%s  waves/SIMD %d  %d v_add per MFMA   cycles per MFMA per wave %.1f  (per SIMD %.1f; the pipe needs %d)  clock %.2f GHz  %.0f TFLOP/s\n",
         KIND == 0 ? "fp8 scaled 16x16x128" : "bf16 16x16x32       ", wps, NADD, med_cyc / n, med_cyc / n / wps, KIND == 0 ? 32 : 16, med_clk,
         flops / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

int main() {
  i8v* ab; float* out; long long* stamps;
  const size_t nab = 6 * 64;
  std::vector<unsigned char> h(nab * 32);
  srand(1);
  // random e4m3 bytes without NaN encodings (0x7f / 0xff); read as bf16 pairs by the other kind they are random finite-ish values too
  for (auto& v : h) { unsigned char c = (unsigned char)(rand() & 0xff); if ((c & 0x7f) == 0x7f) c ^= 1; if ((c & 0x78) == 0x78) c ^= 0x40; v = c; }
  CK(hipMalloc(&ab, nab * 32)); CK(hipMemcpy(ab, h.data(), nab * 32, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, 256 * 4 * 256 * sizeof(float)));
  CK(hipMalloc(&stamps, 256 * 4 * 4 * 2 * sizeof(long long)));
  for (int wps : {1, 2}) {
    run<0, 0>(ab, out, stamps, wps); run<1, 0>(ab, out, stamps, wps);
    run<0, 2>(ab, out, stamps, wps); run<1, 1>(ab, out, stamps, wps);
    run<0, 4>(ab, out, stamps, wps); run<1, 2>(ab, out, stamps, wps);
  }
  return 0;
}
