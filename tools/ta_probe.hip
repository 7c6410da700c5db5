// What one vector-memory instruction costs the CU's address path (TA), by access shape (round 4: choosing the accumulator layout of
// the 256x256 GEMM's epilogue). One 512-thread workgroup per CU; every wave issues REPS instructions of one shape back to back and
// then waits (s_waitcnt vmcnt(0)); the median workgroup's cycles per wave-instruction are reported (8 waves issue concurrently, so
// "cycles per instruction per CU" = cycles / (8 x REPS)). All addresses stay inside a per-workgroup window of a few hundred KB that
// is touched once before the timed loop: the numbers are L2-hit request rates, not HBM bandwidth. `idle` cycles of nothing
// between repetitions keep the memory system from saturating.
//   hipcc -O3 --offload-arch=gfx950 ta_probe.hip -o ta_probe && ./ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u4v;
typedef __attribute__((ext_vector_type(2))) unsigned u2v;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

enum { ST_X4_SCATTER16 = 0,  // store dwordx4: lane (fr = l & 15 -> row, fg = l >> 4 -> 16-B chunk): 16 rows x 64 B  (shipped epilogue)
       ST_X4_ROWS4 = 1,      // store dwordx4: lane (j = l & 15 -> 16-B chunk, fg -> row): 4 rows x 256 B contiguous
       ST_X2_ROWS4 = 2,      // store dwordx2: lane (j -> 8-B chunk, fg -> row): 4 rows x 128 B contiguous
       ST_X4_LINES8 = 3,     // store dwordx4: lane (l >> 3 -> row, l & 7 -> chunk): 8 rows x 128 B
       LD_X4_SCATTER16 = 4,  // load dwordx4 of the shipped RoPE-table shape: lane (fr -> row of 256 B, fg -> 16-B chunk)
       LD_X4_ROWS4 = 5,      // load dwordx4: lane (j -> chunk, fg -> row): 4 rows x 256 B contiguous
       LD_X4_STRIDE32 = 6,   // load dwordx4: lane (j -> 16 B at byte 32 j, fg -> row): 4 rows, every other 16-B chunk of 512 B
       DMA_X4_LINES8 = 7 };  // global_load_lds_dwordx4: lane (l >> 3 -> row, l & 7 -> chunk): 8 rows x 128 B (the K loop's staging)

template <int SHAPE, int REPS>
__global__ __launch_bounds__(512) void probe(char* buf, long long* stamps, int rowbytes, int tiles, int idle) {
  __shared__ __attribute__((aligned(16))) char smem[8 * 1024 * 8];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  char* base = buf + (size_t)blockIdx.x * (256 * 1024);
  const int fr = lane & 15, fg = lane >> 4;
  long long total = 0;
  u4v keep = {0, 0, 0, 0};
  for (int t = 0; t < tiles; ++t) {
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < REPS; ++i) {
      // instruction i of a wave covers an R-row x W-byte block of the wave's own region; blocks of one burst are disjoint
      constexpr int R = (SHAPE == ST_X4_SCATTER16 || SHAPE == LD_X4_SCATTER16) ? 16 : (SHAPE == ST_X4_LINES8 || SHAPE == DMA_X4_LINES8) ? 8 : 4;
      constexpr int W = (SHAPE == ST_X4_SCATTER16 || SHAPE == LD_X4_SCATTER16) ? 64 : (SHAPE == ST_X4_LINES8 || SHAPE == DMA_X4_LINES8 || SHAPE == ST_X2_ROWS4) ? 128
                        : SHAPE == LD_X4_STRIDE32 ? 512 : 256;
      const int per_row = rowbytes / W;
      char* w = base + (size_t)wid * (32 * 1024) + (size_t)((i / per_row) * R) * rowbytes + (i % per_row) * W;
      const u4v val = {(unsigned)lane, (unsigned)i, 2u, 3u};
      if (SHAPE == ST_X4_SCATTER16) *reinterpret_cast<u4v*>(w + fr * rowbytes + fg * 16) = val;
      if (SHAPE == ST_X4_ROWS4) *reinterpret_cast<u4v*>(w + fg * rowbytes + fr * 16) = val;
      if (SHAPE == ST_X2_ROWS4) *reinterpret_cast<u2v*>(w + fg * rowbytes + fr * 8) = u2v{val[0], val[1]};
      if (SHAPE == ST_X4_LINES8) *reinterpret_cast<u4v*>(w + (lane >> 3) * rowbytes + (lane & 7) * 16) = val;
      if (SHAPE == LD_X4_SCATTER16) keep += *reinterpret_cast<volatile u4v*>(w + fr * rowbytes + fg * 16);
      if (SHAPE == LD_X4_ROWS4) keep += *reinterpret_cast<volatile u4v*>(w + fg * rowbytes + fr * 16);
      if (SHAPE == LD_X4_STRIDE32) keep += *reinterpret_cast<volatile u4v*>(w + fg * rowbytes + fr * 32);
      if (SHAPE == DMA_X4_LINES8)
        __builtin_amdgcn_global_load_lds(w + (lane >> 3) * rowbytes + (lane & 7) * 16, LDS_PTR(smem + wid * 8192 + (i & 7) * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (t > 1) total += t1 - t0;
    if (idle > 0) {
      const long long until = __builtin_amdgcn_s_memtime() + idle;
      while (__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(8);
    }
  }
  if (keep[0] == 0x12345678u) buf[0] = 1;  // keep the loads alive
  if (threadIdx.x == 0) stamps[blockIdx.x] = total / (tiles - 2);
}

template <int SHAPE, int REPS>
static void run(const char* what, char* buf, long long* stamps, int rowbytes, int idle) {
  const int nwg = 256, tiles = 20;
  hipLaunchKernelGGL((probe<SHAPE, REPS>), dim3(nwg), dim3(512), 0, 0, buf, stamps, rowbytes, tiles, idle);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((probe<SHAPE, REPS>), dim3(nwg), dim3(512), 0, 0, buf, stamps, rowbytes, tiles, idle);
  CK(hipDeviceSynchronize());
  std::vector<long long> h(nwg);
  CK(hipMemcpy(h.data(), stamps, nwg * sizeof(long long), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double cyc = (double)h[nwg / 2];
  const int bytes_instr = (SHAPE == ST_X2_ROWS4) ? 512 : 1024;
  printf("%-64s reps=%2d rowbytes=%5d idle=%5d | %7.0f cycles per burst = %6.1f cycles per wave-instruction per CU, %5.1f B/cycle/CU\n", what, REPS,
         rowbytes, idle, cyc, cyc / (8.0 * REPS), bytes_instr * 8.0 * REPS / cyc);
}

int main() {
  char* buf; long long* stamps;
  const size_t bytes = (size_t)256 * 256 * 1024 + (1 << 22);
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&stamps, 4096 * sizeof(long long)));
  CK(hipMemset(buf, 0, bytes));
  for (int idle : {20000}) {
    run<ST_X4_SCATTER16, 16>("store x4, 16 rows x 64 B, lane-scattered (shipped epilogue)", buf, stamps, 2048, idle);
    run<ST_X4_ROWS4, 16>("store x4, 4 rows x 256 B, 16 lanes contiguous", buf, stamps, 2048, idle);
    run<ST_X2_ROWS4, 32>("store x2, 4 rows x 128 B, 16 lanes contiguous (32 instr = same bytes)", buf, stamps, 2048, idle);
    run<ST_X4_LINES8, 16>("store x4, 8 rows x 128 B, 8 lanes contiguous", buf, stamps, 2048, idle);
    run<ST_X4_ROWS4, 16>("store x4, 4 rows x 256 B, 16 lanes contiguous, rows of 8 KB", buf, stamps, 8192, idle);
    run<ST_X2_ROWS4, 32>("store x2, 4 rows x 128 B, rows of 8 KB", buf, stamps, 8192, idle);
    run<LD_X4_SCATTER16, 32>("load x4, 16 rows x 64 B, lane-scattered (shipped RoPE rows)", buf, stamps, 256, idle);
    run<LD_X4_ROWS4, 32>("load x4, 4 rows x 256 B, 16 lanes contiguous", buf, stamps, 256, idle);
    run<LD_X4_STRIDE32, 32>("load x4, 4 rows, 16 B at lane stride 32 B", buf, stamps, 512, idle);
    run<LD_X4_ROWS4, 16>("load x4, 4 rows x 256 B, 16 lanes contiguous", buf, stamps, 256, idle);
    run<DMA_X4_LINES8, 8>("LDS-DMA x4, 8 rows x 128 B", buf, stamps, 2048, idle);
    run<DMA_X4_LINES8, 32>("LDS-DMA x4, 8 rows x 128 B", buf, stamps, 2048, idle);
  }
  return 0;
}
