// K2 blend-shape GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact f32 FMA chain).
//
//   forward : v_posed (B,N3) = coef (B,220) x blend (220,N3) + v_template      (N3 = 20670)
//   backward: dcoef  (B,220) = dv_posed (B,N3) x blend^T                         (split-K)
//
// Reference: keras_smpl/batch_smpl.py:106-108 (shape blend, K.dot) and :126-128 (pose blend,
// K.dot) fused into one contraction over [beta | pose_feature].
//
// Forward tiling: workgroup = 4 waves = 32 meshes x 384 columns, each wave a 32x96 strip
// (3 accumulator tiles).  coef^T sits in LDS ([k][mesh], row stride 33 -> conflict-free both
// ways); the B operand needs no transpose so each wave reads blend rows straight from
// global/L2 (two 128-B row segments per load).  The blockIdx -> (column block, mesh tile)
// map keeps the mesh tiles that share a column block on one XCD (blocks b and b+8 share an
// L2), so blend is fetched from HBM/MALL once and re-read from L2.
//
// Backward tiling: workgroup = 32 meshes x all 220 outputs (7 tiles) x one K-slice of
// columns; both operands are transposed through LDS in 64-column chunks; the 4 waves split
// each chunk's columns, are summed in LDS in a fixed order, and the per-slice partials are
// summed by a second kernel in slice order (deterministic, no atomics).
#include "common.h"

namespace smplr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KP = SMPLR_KPAD;  // 220
constexpr int FW_BM = 32, FW_WN = 96, FW_BN = 384;
constexpr int A_LD = 33;

__global__ __launch_bounds__(256) void blend_fwd_kernel(const float *__restrict__ coef,
                                                        const float *__restrict__ blend,
                                                        const float *__restrict__ vt, int B, int N3,
                                                        int ncb, int nmt, float *__restrict__ out) {
  __shared__ float sA[KP * A_LD];
  // XCD-aware map: within a group of 8*nmt consecutive blocks, the nmt blocks with equal
  // (bid % 8) take the same column block.
  const int bid = blockIdx.x;
  const int group = bid / (8 * nmt), within = bid % (8 * nmt);
  const int cb = group * 8 + (within & 7), mt = within >> 3;
  if (cb >= ncb) return;
  const int m0 = mt * FW_BM;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

  for (int e = tid; e < FW_BM * KP; e += 256) {
    const int i = e / KP, k = e % KP;
    const int m = m0 + i;
    sA[k * A_LD + i] = (m < B) ? coef[(size_t)m * KP + k] : 0.0f;
  }
  __syncthreads();

  const int i = lane & 31, h = lane >> 5;
  const int cw = cb * FW_BN + wave * FW_WN;
  int col[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int c = cw + t * 32 + i;
    col[t] = c < N3 ? c : N3 - 1;  // clamped; discarded in the epilogue
  }
  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

#pragma unroll 5
  for (int k0 = 0; k0 < KP; k0 += 2) {
    const float a = sA[(k0 + h) * A_LD + i];
    const float *brow = blend + (size_t)(k0 + h) * N3;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const float b = brow[col[t]];
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
  }

#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int c = cw + t * 32 + i;
    if (c < N3) {
      const float base = vt[c];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int m = m0 + row;
        if (m < B) out[(size_t)m * N3 + c] = acc[t][r] + base;
      }
    }
  }
}

// ---------------------------------------------------------------- backward (split-K)
constexpr int BW_KC = 64;            // columns per LDS chunk
constexpr int BW_LD = BW_KC + 1;     // 65: odd stride -> conflict-free transposed reads
constexpr int BW_NT = 7;             // 7 x 32 = 224 >= 220 outputs
constexpr int BW_NO = 224;

__global__ __launch_bounds__(256) void blend_bwd_kernel(const float *__restrict__ dvp,
                                                        const float *__restrict__ blend, int B, int N3,
                                                        int nslices, int nchunks,
                                                        float *__restrict__ part) {
  extern __shared__ float smem[];
  float *sD = smem;                    // [32][65]
  float *sB = smem + 32 * BW_LD;       // [220][65]   (reused as the [32][224] reduction buffer)
  const int slice = blockIdx.x, mt = blockIdx.y;
  const int m0 = mt * 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;
  // chunk range of this slice (balanced)
  const int c_beg = (int)((long long)nchunks * slice / nslices);
  const int c_end = (int)((long long)nchunks * (slice + 1) / nslices);

  f32x16 acc[BW_NT];
#pragma unroll
  for (int t = 0; t < BW_NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  for (int ch = c_beg; ch < c_end; ++ch) {
    const int col0 = ch * BW_KC;
    __syncthreads();
    for (int e = tid; e < 32 * BW_KC; e += 256) {
      const int r = e >> 6, c = e & 63;
      const int m = m0 + r, cc = col0 + c;
      sD[r * BW_LD + c] = (m < B && cc < N3) ? dvp[(size_t)m * N3 + cc] : 0.0f;
    }
    for (int e = tid; e < KP * BW_KC; e += 256) {
      const int r = e >> 6, c = e & 63;
      const int cc = col0 + c;
      sB[r * BW_LD + c] = (cc < N3) ? blend[(size_t)r * N3 + cc] : 0.0f;
    }
    __syncthreads();
    // wave w owns columns [16w, 16w+16) of the chunk: 8 k-steps of 2
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int kk = wave * 16 + s * 2 + h;
      const float a = sD[i * BW_LD + kk];
#pragma unroll
      for (int t = 0; t < BW_NT; ++t) {
        const int j = t * 32 + i;
        const float b = (j < KP) ? sB[j * BW_LD + kk] : 0.0f;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
      }
    }
  }
  // fixed-order reduction of the 4 waves through LDS, then one partial per (slice, mesh tile)
  float *sR = sB;  // [32][224]
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < BW_NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          float *p = &sR[row * BW_NO + t * 32 + i];
          *p = (w == 0) ? acc[t][r] : (*p + acc[t][r]);
        }
    }
  }
  __syncthreads();
  float *dst = part + ((size_t)slice * gridDim.y + mt) * (32 * BW_NO);
  for (int e = tid; e < 32 * BW_NO; e += 256) dst[e] = sR[e];
}

__global__ __launch_bounds__(256) void blend_bwd_reduce_kernel(const float *__restrict__ part, int B,
                                                               int nslices, int nmt,
                                                               float *__restrict__ dcoef) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * KP) return;
  const int m = e / KP, k = e % KP;
  const int mt = m >> 5, r = m & 31;
  float acc = 0.0f;
  for (int s = 0; s < nslices; ++s) acc += part[((size_t)s * nmt + mt) * (32 * BW_NO) + r * BW_NO + k];
  dcoef[e] = acc;
}

static int bwd_slices(int B, int N3) {
  const int nmt = (B + 31) / 32;
  const int nchunks = (N3 + BW_KC - 1) / BW_KC;
  int s = (256 + nmt - 1) / nmt;  // ~one block per CU
  if (s < 8) s = 8;
  if (s > nchunks) s = nchunks;
  return s;
}

}  // namespace smplr

extern "C" {

int smplr_blend_fwd(const float *coef, const float *blend, const float *v_template, int B, int N3,
                    float *v_posed, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && N3 > 0, "smplr_blend_fwd: bad sizes B=%d N3=%d", B, N3);
  if (B == 0) return 0;
  SMPLR_REQUIRE(coef && blend && v_template && v_posed, "smplr_blend_fwd: null pointer");
  const int ncb = (N3 + FW_BN - 1) / FW_BN, nmt = (B + FW_BM - 1) / FW_BM;
  const int grid = ((ncb + 7) / 8) * 8 * nmt;
  hipLaunchKernelGGL(blend_fwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream), coef, blend,
                     v_template, B, N3, ncb, nmt, v_posed);
  SMPLR_LAUNCH_CHECK("smplr_blend_fwd");
  return 0;
}

size_t smplr_blend_bwd_workspace(int B, int N3) {
  using namespace smplr;
  if (B <= 0 || N3 <= 0) return 0;
  const int nmt = (B + 31) / 32;
  return (size_t)bwd_slices(B, N3) * nmt * 32 * BW_NO * sizeof(float);
}

int smplr_blend_bwd(const float *dv_posed, const float *blend, int B, int N3, float *dcoef,
                    void *workspace, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && N3 > 0, "smplr_blend_bwd: bad sizes B=%d N3=%d", B, N3);
  if (B == 0) return 0;
  SMPLR_REQUIRE(dv_posed && blend && dcoef && workspace, "smplr_blend_bwd: null pointer");
  const int nmt = (B + 31) / 32, nchunks = (N3 + BW_KC - 1) / BW_KC, ns = bwd_slices(B, N3);
  const size_t lds = (size_t)(32 * BW_LD + KP * BW_LD) * sizeof(float);  // 65,520 B
  static bool attr_set = false;
  if (!attr_set) {
    SMPLR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(blend_bwd_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(blend_bwd_kernel, dim3(ns, nmt), dim3(256), lds, as_stream(stream), dv_posed, blend,
                     B, N3, ns, nchunks, reinterpret_cast<float *>(workspace));
  SMPLR_LAUNCH_CHECK("smplr_blend_bwd");
  hipLaunchKernelGGL(blend_bwd_reduce_kernel, dim3((B * KP + 255) / 256), dim3(256), 0,
                     as_stream(stream), reinterpret_cast<const float *>(workspace), B, ns, nmt, dcoef);
  SMPLR_LAUNCH_CHECK("smplr_blend_bwd(reduce)");
  return 0;
}

}  // extern "C"
