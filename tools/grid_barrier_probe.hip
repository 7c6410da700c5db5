// What a device-wide barrier costs inside one persistent kernel on MI355X (round 3: pricing a fused denoising loop against its
// 20 launches per step). One 256-thread workgroup per CU; every barrier = all waves of the workgroup arrive (s_barrier), thread 0
// bumps an agent-scope counter and spins (bounded: a budget of polls, after which the kernel sets a flag and leaves - no wave
// can wait forever) until the counter reaches the round's target. Variants: the counter alone, and the counter plus the
// release / acquire fences a producer-consumer exchange through HBM-backed memory across XCDs needs.
//   hipcc -O3 --offload-arch=gfx950 grid_barrier_probe.hip -o grid_barrier_probe && ./grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool FENCES, bool PAYLOAD>
__global__ __launch_bounds__(256) void probe(unsigned* counter, unsigned* fail, float* buf, int rounds, int nwg_group, int groups) {
  // groups > 1: independent barriers among the workgroups that share blockIdx % groups (with groups = 8: one XCD each)
  const int g = blockIdx.x % groups;
  unsigned* ctr = counter + 64 * g;  // one counter per 256-byte line
  const int members = nwg_group;
  float acc = 0.f;
  for (int r = 1; r <= rounds; ++r) {
    if (PAYLOAD) {  // every workgroup publishes 1 KiB, then reads its neighbour's after the barrier
      buf[(size_t)blockIdx.x * 256 + threadIdx.x] = (float)r + threadIdx.x;
    }
    if (FENCES) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)members * (unsigned)r;
      int budget = 1 << 20;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && --budget > 0) __builtin_amdgcn_s_sleep(1);
      if (budget <= 0) atomicExch(fail, 1u);
    }
    __syncthreads();
    if (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (*reinterpret_cast<volatile unsigned*>(fail)) return;  // somebody gave up: everybody leaves
    if (PAYLOAD) {
      const int nb = (blockIdx.x + groups) % gridDim.x;
      acc += __builtin_nontemporal_load(buf + (size_t)nb * 256 + threadIdx.x);
    }
  }
  if (PAYLOAD) buf[(size_t)(gridDim.x + blockIdx.x) * 256 + threadIdx.x] = acc;
}

template <bool FENCES, bool PAYLOAD>
static void run(const char* what, unsigned* counter, unsigned* fail, float* buf, int nwg, int groups) {
  const int rounds = 2000;
  CK(hipMemset(counter, 0, 64 * 8 * sizeof(unsigned)));
  CK(hipMemset(fail, 0, sizeof(unsigned)));
  hipLaunchKernelGGL((probe<FENCES, PAYLOAD>), dim3(nwg), dim3(256), 0, 0, counter, fail, buf, 50, nwg / groups, groups);
  CK(hipDeviceSynchronize());
  CK(hipMemset(counter, 0, 64 * 8 * sizeof(unsigned)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((probe<FENCES, PAYLOAD>), dim3(nwg), dim3(256), 0, 0, counter, fail, buf, rounds, nwg / groups, groups);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned f = 0;
  CK(hipMemcpy(&f, fail, sizeof(unsigned), hipMemcpyDeviceToHost));
  printf("%-58s %3d workgroups in %d group(s): %.2f us per barrier%s\n", what, nwg, groups, ms * 1e3 / rounds, f ? "  (GAVE UP: poll budget exhausted)" : "");
  fflush(stdout);
}

int main() {
  unsigned *counter, *fail; float* buf;
  CK(hipMalloc(&counter, 64 * 8 * sizeof(unsigned)));
  CK(hipMalloc(&fail, sizeof(unsigned)));
  CK(hipMalloc(&buf, 2 * 256 * 256 * sizeof(float)));
  for (int nwg : {64, 128, 256}) {
    run<false, false>("counter only", counter, fail, buf, nwg, 1);
    run<true, false>("counter + agent-scope release/acquire fences", counter, fail, buf, nwg, 1);
    run<true, true>("fences + 1 KiB published and read by a neighbour", counter, fail, buf, nwg, 1);
  }
  run<false, false>("counter only, 8 independent groups (blockIdx % 8)", counter, fail, buf, 256, 8);
  run<true, true>("fences + payload, 8 independent groups (blockIdx % 8)", counter, fail, buf, 256, 8);
  return 0;
}
