// What the 256x256 GEMM epilogue's output stores cost a CU, by themselves (round 4: sizing the woven epilogue of gemm256s).
// One 512-thread workgroup per CU; per "tile" every wave issues 16 global_store_dwordx4 (1 KiB each, 128 KiB per workgroup) with the
// address pattern of the kernel's epilogue (out[M][N] bf16, a wave = 128 rows x 64 columns), then either waits for all of them
// (burst: the drain time of one tile, `idle` cycles of nothing between tiles as a main loop would leave) or issues them one every `gap` cycles (spread: the rate the CU sustains).
//   hipcc -O3 --offload-arch=gfx950 store_probe.hip -o store_probe && ./store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u4v;

// pattern 0: the shipped epilogue: one instruction = 16 rows x 64 B      (lane: row fr = lane & 15, 16-B chunk fg = lane >> 4)
// pattern 1: 8 rows x 128 B (full lines)                                  (row = lane >> 3, chunk = lane & 7)
// pattern 2: 2 rows x 512 B                                              (row = lane >> 5, chunk = lane & 31)  [wave = 32 rows x 256 cols]
// NT: nontemporal stores. WAVES: waves of the workgroup that store (the others idle at the barrier).
template <int PATTERN, bool NT>
__global__ __launch_bounds__(512) void probe(char* out, long long* stamps, int N, int tiles, int gap, int waves, int rows_total, int idle) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wr = wid >> 2, wc = wid & 3;
  const size_t rowbytes = (size_t)N * 2;
  const int ntn = N / 256;
  long long t_issue = 0, t_drain = 0;
  const u4v val = {(unsigned)threadIdx.x, 1u, 2u, 3u};
  for (int t = 0; t < tiles; ++t) {
    // tile walk: consecutive tiles of a workgroup go down the rows (as the persistent kernel's chunks do, roughly)
    const int tile = (blockIdx.x * tiles + t);
    const int tm = (tile / ntn) % (rows_total / 256), tn = tile % ntn;
    char* base = out + (size_t)tm * 256 * rowbytes + (size_t)tn * 512;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (wid < waves) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        char* p;
        if (PATTERN == 0) {  // i = mf * 2 + pr: row block mf (16 rows), 64-byte column group pr
          const int mf = i >> 1, pr = i & 1;
          p = base + (size_t)(wr * 128 + mf * 16 + (lane & 15)) * rowbytes + wc * 128 + pr * 64 + (lane >> 4) * 16;
        } else if (PATTERN == 1) {  // i: 8-row block of the wave's 128 rows, full 128-byte line of the wave's 64 columns
          p = base + (size_t)(wr * 128 + i * 8 + (lane >> 3)) * rowbytes + wc * 128 + (lane & 7) * 16;
        } else {  // the wave owns 32 rows x all 256 columns: i = 2-row block
          p = base + (size_t)(wid * 32 + i * 2 + (lane >> 5)) * rowbytes + (lane & 31) * 16;
        }
        if (NT) __builtin_nontemporal_store(val, reinterpret_cast<u4v*>(p));
        else *reinterpret_cast<u4v*>(p) = val;
        if (gap > 0) {
          const long long until = __builtin_amdgcn_s_memtime() + gap;
          while (__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t2 = __builtin_amdgcn_s_memtime();
    if (t > 0) { t_issue += t1 - t0; t_drain += t2 - t0; }
    if (idle > 0) {  // the main loop of the next tile: nothing is stored for `idle` cycles
      const long long until = __builtin_amdgcn_s_memtime() + idle;
      while (__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(8);
    }
  }
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = t_issue / (tiles - 1);
    stamps[blockIdx.x * 2 + 1] = t_drain / (tiles - 1);
  }
}

template <int PATTERN, bool NT>
static void run(const char* what, char* out, long long* stamps, int nwg, int N, int gap, int waves, int rows_total, int idle = 40000) {
  const int tiles = 24;
  hipLaunchKernelGGL((probe<PATTERN, NT>), dim3(nwg), dim3(512), 0, 0, out, stamps, N, 4, gap, waves, rows_total, idle);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((probe<PATTERN, NT>), dim3(nwg), dim3(512), 0, 0, out, stamps, N, tiles, gap, waves, rows_total, idle);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> h(nwg * 2);
  CK(hipMemcpy(h.data(), stamps, nwg * 2 * sizeof(long long), hipMemcpyDeviceToHost));
  std::vector<long long> iss, dr;
  for (int i = 0; i < nwg; ++i) { iss.push_back(h[2 * i]); dr.push_back(h[2 * i + 1]); }
  std::sort(iss.begin(), iss.end()); std::sort(dr.begin(), dr.end());
  const double bytes_tile = (double)waves * 16 * 1024;
  const double us_tile = ms * 1e3 / tiles;
  // s_memtime counts shader cycles (MI355X_MICROARCH.md, cycle constants)
  const double cyc = (double)dr[nwg / 2];
  printf("%-44s N=%5d wgs=%3d waves=%d gap=%5d idle=%5d | issue %6lld  drain %6lld cycles (median per tile) = %5.1f B/cycle/CU | %7.2f us/tile  %7.1f GB/s chip  clock %.2f GHz\n",
         what, N, nwg, waves, gap, idle, iss[nwg / 2], dr[nwg / 2], bytes_tile / cyc, us_tile, bytes_tile * nwg / (us_tile * 1e3), cyc / (us_tile * 1e3));
}

int main() {
  const int rows_total = 163840;
  const size_t bytes = (size_t)rows_total * 4096 * 2;  // room for N up to 4096
  char* out; long long* stamps;
  CK(hipMalloc(&out, bytes));
  CK(hipMalloc(&stamps, 4096 * sizeof(long long)));
  CK(hipMemset(out, 0, bytes));
  for (int N : {1024, 3072, 4096}) {
    run<0, false>("burst, shipped pattern (16 rows x 64 B)", out, stamps, 256, N, 0, 8, rows_total);
    run<1, false>("burst, full lines (8 rows x 128 B)", out, stamps, 256, N, 0, 8, rows_total);
    run<2, false>("burst, 2 rows x 512 B", out, stamps, 256, N, 0, 8, rows_total);
    run<0, true>("burst, shipped pattern, nontemporal", out, stamps, 256, N, 0, 8, rows_total);
  }
  // back to back (no idle time between tiles): the sustained store rate of the chip
  run<0, false>("back-to-back tiles, shipped pattern", out, stamps, 256, 3072, 0, 8, rows_total, 0);
  // chip-level or CU-level? fewer workgroups, same per-CU burst
  for (int nwg : {8, 32, 64, 128}) run<0, false>("burst, shipped pattern, fewer workgroups", out, stamps, nwg, 3072, 0, 8, rows_total);
  // half the waves store (same bytes per storing wave)
  run<0, false>("burst, 4 of 8 waves store", out, stamps, 256, 3072, 0, 4, rows_total);
  run<0, false>("burst, 2 of 8 waves store", out, stamps, 256, 3072, 0, 2, rows_total);
  // spread: one store per wave every `gap` shader cycles
  for (int gap : {50, 100, 200, 400, 800, 1600}) run<0, false>("spread, shipped pattern", out, stamps, 256, 3072, gap, 8, rows_total);
  for (int gap : {200, 800}) run<0, false>("spread, shipped pattern, N = 1024", out, stamps, 256, 1024, gap, 8, rows_total);
  return 0;
}
