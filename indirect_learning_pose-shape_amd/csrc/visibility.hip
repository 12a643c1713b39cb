// K5 visibility mask: a z-buffer over a fixed grid of rounded pixel positions, one workgroup
// per mesh, the grid (grid_wh^2 x 8 B) and the per-vertex visible flags held in LDS.
//
// Reference: keras_smpl/compute_mask.py:12-108, stateless semantics:
// round (u,v) half-to-even (:22); winner of a grid cell = candidate of largest z, lowest vertex
// index on ties (tf.argmax, :98-103); an empty cell yields index 1 (:99); mask = 500 except
// winners = 1 (:65-70).  The reference's nested map_fn (4096 sequential cell scans of all
// vertices, :59-63) becomes one 64-bit LDS atomicMax per vertex on the key
// (orderable(z) << 32 | ~index) followed by one scan of the cells.
#include "common.h"

namespace smplr {

__global__ __launch_bounds__(1024) void visibility_kernel(const float *__restrict__ proj, int VP, int G,
                                                          int ref_compat, float *__restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long zbuf[];                 // G*G keys
  unsigned int *vis = reinterpret_cast<unsigned int *>(zbuf + (size_t)G * G);  // ceil(VP/32) words
  __shared__ int any_empty;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int cells = G * G, words = (VP + 31) / 32;
  for (int i = tid; i < cells; i += 1024) zbuf[i] = 0ull;
  for (int i = tid; i < words; i += 1024) vis[i] = 0u;
  if (tid == 0) any_empty = 0;
  __syncthreads();
  const float *p = proj + (size_t)n * VP * 3;
  const float fG = (float)G;
  for (int v = tid; v < VP; v += 1024) {
    const float pu = rintf(p[v * 3 + 0]);   // round half to even, like tf.round
    const float pv = rintf(p[v * 3 + 1]);
    if (pu >= 0.0f && pu < fG && pv >= 0.0f && pv < fG) {
      const int cell = (int)pv * G + (int)pu;
      const unsigned long long key =
          ((unsigned long long)orderable(p[v * 3 + 2]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)v);
      atomicMax(&zbuf[cell], key);
    }
  }
  __syncthreads();
  int empty = 0;
  for (int i = tid; i < cells; i += 1024) {
    const unsigned long long key = zbuf[i];
    if (key == 0ull) {
      empty = 1;
    } else {
      const unsigned int v = 0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull);
      atomicOr(&vis[v >> 5], 1u << (v & 31));
    }
  }
  if (empty) any_empty = 1;   // benign same-value race
  __syncthreads();
  if (tid == 0 && any_empty && ref_compat && VP > 1) atomicOr(&vis[0], 2u);   // vertex 1
  __syncthreads();
  float *m = mask + (size_t)n * VP;
  for (int v = tid; v < VP; v += 1024) m[v] = ((vis[v >> 5] >> (v & 31)) & 1u) ? 1.0f : 500.0f;
}

}  // namespace smplr

extern "C" int smplr_visibility(const float *proj, int B, int VP, int grid_wh, int ref_compat, float *mask,
                                void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && VP > 0 && grid_wh > 0 && grid_wh <= 128,
                "smplr_visibility: bad sizes B=%d VP=%d grid_wh=%d (max 128)", B, VP, grid_wh);
  if (B == 0) return 0;
  SMPLR_REQUIRE(proj && mask, "smplr_visibility: null pointer");
  const size_t lds = (size_t)grid_wh * grid_wh * 8 + (size_t)((VP + 31) / 32) * 4;
  SMPLR_REQUIRE(lds <= 150 * 1024, "smplr_visibility: grid + flags need %zu B of LDS (max 153600)", lds);
  if (lds > 48 * 1024) {
    static LdsAttrMemo memo = {};                     // once per (kernel, device): common.h
    int rc = ensure_lds_attr(reinterpret_cast<const void *>(visibility_kernel), lds, &memo, "visibility_kernel");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(visibility_kernel, dim3(B), dim3(1024), lds, as_stream(stream), proj, VP, grid_wh,
                     ref_compat, mask);
  SMPLR_LAUNCH_CHECK("smplr_visibility");
  return 0;
}
