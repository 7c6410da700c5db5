// Point-set metrics of the step after generation: the O(N M) pairwise-distance work behind the
// reference's Chamfer and EMD evaluations
//   compute_chamfer_distance / compute_emd_distance   test_optimize.py:354-415  (torch.cdist + min / linear assignment)
//   distChamfer / emd_approx                          train_newloss.py:316-372  (same structure on normalised points)
// Points are [B, n, 3] float32 (what NOVAPipeline's latent output flattens to). Both kernels are plain VALU work on
// 12-byte points: K = 3 is no MFMA shape (a 16-deep contraction would be 81 % padding), the arithmetic is in exact
// differences (x - y)^2 rather than cdist's |x|^2 + |y|^2 - 2 x.y expansion, which cancels for the near neighbours that
// Chamfer is about.
//   nn_dist        d[b, i] = min_j ||clamp(x[b, i]) - clamp(y[b, j])||     one thread per x point, y tiles through LDS
//   pairwise_dist  D[b, i, j] = ||clamp(x[b, i]) - clamp(y[b, j])||        the cost matrix of the assignment problem
#include "common.h"
#include "nova_internal.h"

namespace nova {

constexpr int PS_TILE = 1024;  // y points staged per LDS tile (12 KiB)

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// unit != 0: points are scaled to unit norm after clamping (distChamfer, train_newloss.py:325-337: x / max(|x|, 1e-8))
__device__ __forceinline__ void load_point(const float* p, float lo, float hi, int unit, float& a, float& b, float& c) {
  a = clampf(p[0], lo, hi);
  b = clampf(p[1], lo, hi);
  c = clampf(p[2], lo, hi);
  if (unit) {
    const float inv = 1.0f / fmaxf(sqrtf(a * a + b * b + c * c), 1e-8f);
    a *= inv;
    b *= inv;
    c *= inv;
  }
}

__global__ __launch_bounds__(256) void nn_dist_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                      float* __restrict__ d, int N, int M, float lo, float hi, int unit) {
  __shared__ float ys[PS_TILE * 3];
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float* xb = x + (size_t)b * N * 3;
  const float* yb = y + (size_t)b * M * 3;
  float x0 = 0.f, x1 = 0.f, x2 = 0.f;
  if (i < N) load_point(xb + (size_t)i * 3, lo, hi, unit, x0, x1, x2);
  float best = __builtin_huge_valf();
  for (int j0 = 0; j0 < M; j0 += PS_TILE) {
    const int cnt = min(PS_TILE, M - j0);
    __syncthreads();
    for (int j = threadIdx.x; j < cnt; j += 256) {
      float a, bb, c;
      load_point(yb + (size_t)(j0 + j) * 3, lo, hi, unit, a, bb, c);
      ys[3 * j] = a;
      ys[3 * j + 1] = bb;
      ys[3 * j + 2] = c;
    }
    __syncthreads();
    for (int j = 0; j < cnt; ++j) {  // every lane reads the same LDS address: broadcast, no conflicts
      const float e0 = x0 - ys[3 * j], e1 = x1 - ys[3 * j + 1], e2 = x2 - ys[3 * j + 2];
      best = fminf(best, fmaf(e2, e2, fmaf(e1, e1, e0 * e0)));
    }
  }
  if (i < N) d[(size_t)b * N + i] = sqrtf(best);
}

__global__ __launch_bounds__(256) void pairwise_dist_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            float* __restrict__ D, int N, int M, float lo, float hi) {
  // one workgroup = 16 x points x 256 y points... laid out so that consecutive lanes write consecutive j (coalesced rows)
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i0 = blockIdx.y * 16;
  const float* xb = x + (size_t)b * N * 3;
  const float* yb = y + (size_t)b * M * 3;
  float y0 = 0.f, y1 = 0.f, y2 = 0.f;
  if (j < M) load_point(yb + (size_t)j * 3, lo, hi, 0, y0, y1, y2);
  for (int ii = 0; ii < 16; ++ii) {
    const int i = i0 + ii;
    if (i >= N) break;
    float a, bb, c;
    load_point(xb + (size_t)i * 3, lo, hi, 0, a, bb, c);  // wave-uniform address: one scalar-like broadcast load
    const float e0 = a - y0, e1 = bb - y1, e2 = c - y2;
    if (j < M) D[((size_t)b * N + i) * M + j] = sqrtf(fmaf(e2, e2, fmaf(e1, e1, e0 * e0)));
  }
}

int pointset_nn_dist(const float* x, const float* y, float* d, int B, int N, int M, float lo, float hi, int unit, hipStream_t st) {
  if (B <= 0 || N <= 0) return 0;
  if (M <= 0) return set_error(NOVA_ERR_SHAPE, "pointset_nn_dist: empty target set");
  if (B > 65535) return set_error(NOVA_ERR_SHAPE, "pointset_nn_dist: batch %d too large", B);
  hipLaunchKernelGGL(nn_dist_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, x, y, d, N, M, lo, hi, unit);
  return check_launch("pointset_nn_dist");
}

int pointset_pairwise_dist(const float* x, const float* y, float* D, int B, int N, int M, float lo, float hi, hipStream_t st) {
  if (B <= 0 || N <= 0 || M <= 0) return 0;
  if (B > 65535 || (N + 15) / 16 > 65535) return set_error(NOVA_ERR_SHAPE, "pointset_pairwise_dist: grid too large");
  hipLaunchKernelGGL(pairwise_dist_kernel, dim3((M + 255) / 256, (N + 15) / 16, B), dim3(256), 0, st, x, y, D, N, M, lo, hi);
  return check_launch("pointset_pairwise_dist");
}

}  // namespace nova
