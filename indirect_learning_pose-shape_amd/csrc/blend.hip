// K2 blend-shape GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact f32 FMA chain).
//
//   forward : v_posed (B,N3) = coef (B,220) x blend (220,N3) + v_template      (N3 = 20670)
//   backward: dcoef  (B,220) = dv_posed (B,N3) x blend^T                         (split-K)
//
// Reference: keras_smpl/batch_smpl.py:106-108 (shape blend, K.dot) and :126-128 (pose blend,
// K.dot) fused into one contraction over [beta | pose_feature].
//
// Forward tiling: workgroup = 4 waves = 128 meshes x one 96-column strip, each wave a 32x96 tile
// (3 accumulator tiles) of a different mesh tile.  coef^T of all 128 meshes sits in LDS
// ([tile][k][mesh], row stride 33 -> conflict-free both ways); the B operand needs no transpose
// so each wave reads blend rows straight from global in MFMA layout (two 128-B row segments
// per load) and the four waves share those lines through L1.
//
// Backward tiling: workgroup = 32 meshes x all 220 outputs (7 tiles) x one slice of columns,
// against the TRANSPOSED constant blendT (N3 x 224) so that no operand of the big matrix needs a
// transpose; the 4 waves split the slice, are summed in LDS in a fixed order, and the per-slice
// partials are summed by a second kernel in slice order (deterministic, no atomics).
#include "common.h"

namespace smplr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KP = SMPLR_KPAD;  // 220
constexpr int FW_BM = 32, FW_WN = 96;
constexpr int A_LD = 33;

// grid (column strips of 96, groups of 128 meshes); the 4 waves of a block are the 4 mesh tiles of
// ONE strip, so they issue identical B-operand loads and the CU's L1 serves three of them: the
// big matrix crosses L2->L1 once per 128 meshes instead of once per 32.
__global__ __launch_bounds__(256) void blend_fwd_kernel(const float *__restrict__ coef,
                                                        const float *__restrict__ blend,
                                                        const float *__restrict__ vt, int B, int N3,
                                                        float *__restrict__ out) {
  extern __shared__ float sAall[];          // 4 x [220][33]: coef^T of the 4 mesh tiles
  const int strip = blockIdx.x, mg = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int mbase = mg * 4 * FW_BM;
  {
    const int rows = min(4 * FW_BM, B - mbase);          // meshes present in this group
    const float *src = coef + (size_t)mbase * KP;        // contiguous rows*220 floats
    // 28,160 contiguous floats, 110 per thread, in 2 batches of 55 loads in flight (a batch costs one
    // memory round trip however many loads it holds); (mesh, k) of element e = tid + 256 u advance
    // incrementally (256 = 220 + 36) instead of dividing per element
    const int nvalid = rows * KP;
    int i = tid / KP, k = tid - i * KP;
    for (int e0 = tid; e0 < 4 * FW_BM * KP; e0 += 256 * 55) {
      float v[55];
#pragma unroll
      for (int u = 0; u < 55; ++u) {
        const int e = e0 + u * 256;
        const float ld = src[min(e, nvalid - 1)];       // unconditional (clamped) load, select afterwards
        v[u] = (e < nvalid) ? ld : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 55; ++u) {
        sAall[(i >> 5) * (KP * A_LD) + k * A_LD + (i & 31)] = v[u];
        k += 36; i += 1;
        if (k >= KP) { k -= KP; i += 1; }
      }
    }
  }
  __syncthreads();
  const int m0 = mbase + wave * FW_BM;
  if (m0 >= B) return;                                   // this wave's mesh tile is empty
  const float *sA = sAall + wave * (KP * A_LD);

  const int i = lane & 31, h = lane >> 5;
  const int cw = strip * FW_WN;
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

  // B operand double-buffered in registers: batch q+1 (11 k-steps = 33 loads) is in flight while
  // the 33 MFMAs of batch q run, so one wave per SIMD is enough to cover the L2/HBM latency.
  constexpr int NB = 11;                       // k-steps per batch; 10 batches x 22 rows = 220
  float b0[NB][3], b1[NB][3], b2[NB][3];
  auto load_batch = [&](float (&b)[NB][3], int kb) {
#pragma unroll
    for (int s2 = 0; s2 < NB; ++s2) {
      const float *brow = blend + (size_t)(kb + 2 * s2 + h) * N3;
#pragma unroll
      for (int t = 0; t < 3; ++t) b[s2][t] = brow[col[t]];
    }
  };
  auto mma_batch = [&](const float (&b)[NB][3], int kb) {
    float a[NB];                                // the batch's A operands first: no LDS wait per k-step
#pragma unroll
    for (int s2 = 0; s2 < NB; ++s2) a[s2] = sA[(kb + 2 * s2 + h) * A_LD + i];
#pragma unroll
    for (int s2 = 0; s2 < NB; ++s2) {
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2], b[s2][t], acc[t], 0, 0, 0);
    }
  };
  // Three B-operand batches rotate: batches q+1 and q+2 (66 loads) are in flight under batch q's
  // 33 MFMAs (~1 us), which is what an L2-miss round trip takes.  sched_barrier pins the order:
  // left alone the scheduler sinks the loads next to their uses (8 in flight) and the loop stalls.
#define SMPLR_STEP(cur, nxt2, q)                                   \
  if ((q) + 2 < 10) load_batch(nxt2, ((q) + 2) * 2 * NB);         \
  __builtin_amdgcn_sched_barrier(0);                               \
  mma_batch(cur, (q) * 2 * NB);                                    \
  __builtin_amdgcn_sched_barrier(0);
  load_batch(b0, 0);
  load_batch(b1, 2 * NB);
  __builtin_amdgcn_sched_barrier(0);
  SMPLR_STEP(b0, b2, 0)
  SMPLR_STEP(b1, b0, 1)
  SMPLR_STEP(b2, b1, 2)
  SMPLR_STEP(b0, b2, 3)
  SMPLR_STEP(b1, b0, 4)
  SMPLR_STEP(b2, b1, 5)
  SMPLR_STEP(b0, b2, 6)
  SMPLR_STEP(b1, b0, 7)
  SMPLR_STEP(b2, b1, 8)
  SMPLR_STEP(b0, b2, 9)
#undef SMPLR_STEP

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
// dcoef[m][k] = sum_c dvp[m][c] * blendT[c][k].  blendT = blend transposed, rows padded to 224
// floats, so the B operand (lanes along k) is read straight from global in MFMA layout (two
// 128-B row segments per load) exactly like the forward; only the small A operand (32 meshes x
// 32 columns per stage) is transposed, through a wave-private LDS tile.  The 4 waves of a block
// split the block's column slice, so there is no block barrier in the main loop.
constexpr int BW_NT = 7;             // 7 x 32 = 224 >= 220 outputs
constexpr int BW_NO = 224;           // blendT row stride and partial row stride
constexpr int BW_ST = 32;            // columns per wave stage
constexpr int BW_LD = 33;

__global__ __launch_bounds__(256) void blend_bwd_kernel(const float *__restrict__ dvp,
                                                        const float *__restrict__ blendT, int B, int N3,
                                                        int cols_per_block, int nslices, int nmt,
                                                        float *__restrict__ part) {
  __shared__ float smem[32 * BW_NO];          // 28,672 B: 4 wave tiles (4 x 32 x 33) then the reduction buffer
  // XCD-aware map (as the forward): the nmt mesh tiles of one column slice share an XCD's L2.
  const int bid = blockIdx.x;
  const int group = bid / (8 * nmt), within = bid % (8 * nmt);
  const int slice = group * 8 + (within & 7), mt = within >> 3;
  if (slice >= nslices) return;
  const int m0 = mt * 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;
  float *sAw = smem + wave * (BW_ST * BW_LD);
  const int cw = cols_per_block / 4;
  const int c_beg = slice * cols_per_block + wave * cw, c_end = c_beg + cw;

  f32x16 acc[BW_NT];
#pragma unroll
  for (int t = 0; t < BW_NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  for (int c0 = c_beg; c0 < c_end; c0 += BW_ST) {
    if (c0 >= N3) break;                       // wave-uniform
    // stage A: 32 meshes x 32 columns, coalesced along c, stored [c][mesh]
    {
      float v[16];
      const int cc = c0 + i;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int m = m0 + it * 2 + h;
        v[it] = (m < B && cc < N3) ? dvp[(size_t)m * N3 + cc] : 0.0f;
      }
#pragma unroll
      for (int it = 0; it < 16; ++it) sAw[i * BW_LD + it * 2 + h] = v[it];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // 16 k-steps in 4 batches of 4; batch q+1's 28 B loads fly under batch q's 28 MFMAs
    constexpr int NBB = 4;
    float b0[NBB][BW_NT], b1[NBB][BW_NT];
    auto load_b = [&](float (&b)[NBB][BW_NT], int s0) {
#pragma unroll
      for (int s2 = 0; s2 < NBB; ++s2) {
        int cr = c0 + 2 * (s0 + s2) + h;
        cr = cr < N3 ? cr : N3 - 1;            // a == 0 there
        const float *brow = blendT + (size_t)cr * BW_NO + i;
#pragma unroll
        for (int t = 0; t < BW_NT; ++t) b[s2][t] = brow[t * 32];
      }
    };
    auto mma_b = [&](const float (&b)[NBB][BW_NT], int s0) {
      float a[NBB];
#pragma unroll
      for (int s2 = 0; s2 < NBB; ++s2) a[s2] = sAw[(2 * (s0 + s2) + h) * BW_LD + i];
#pragma unroll
      for (int s2 = 0; s2 < NBB; ++s2) {
#pragma unroll
        for (int t = 0; t < BW_NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2], b[s2][t], acc[t], 0, 0, 0);
      }
    };
    __builtin_amdgcn_sched_barrier(0);
    load_b(b0, 0);
    load_b(b1, 4);
    __builtin_amdgcn_sched_barrier(0);
    mma_b(b0, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_b(b0, 8);
    __builtin_amdgcn_sched_barrier(0);
    mma_b(b1, 4);
    __builtin_amdgcn_sched_barrier(0);
    load_b(b1, 12);
    __builtin_amdgcn_sched_barrier(0);
    mma_b(b0, 8);
    __builtin_amdgcn_sched_barrier(0);
    mma_b(b1, 12);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
  }
  // fixed-order reduction of the 4 waves through LDS, then one partial per (slice, mesh tile)
  float *sR = smem;  // [32][224]
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
  float *dst = part + ((size_t)slice * nmt + mt) * (32 * BW_NO);
  for (int e = tid; e < 32 * BW_NO; e += 256) dst[e] = sR[e];
}

__global__ __launch_bounds__(256) void blend_bwd_reduce_kernel(const float *__restrict__ part, int B,
                                                               int nslices, int nmt,
                                                               float *__restrict__ dcoef) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * KP) return;
  const int m = e / KP, k = e % KP;
  const int mt = m >> 5, r = m & 31;
  const float *p = part + (size_t)mt * (32 * BW_NO) + r * BW_NO + k;
  const size_t stride = (size_t)nmt * (32 * BW_NO);
  float acc = 0.0f;
#pragma unroll 8
  for (int s = 0; s < nslices; ++s) acc += p[s * stride];
  dcoef[e] = acc;
}

static void bwd_geometry(int B, int N3, int *nslices, int *cols_per_block) {
  const int nmt = (B + 31) / 32;
  int target = (256 + nmt - 1) / nmt;          // ~one block per CU
  if (target < 8) target = 8;
  int cpb = (N3 + target - 1) / target;
  cpb = (cpb + 127) / 128 * 128;               // 4 waves x stages of 32 columns
  *cols_per_block = cpb;
  *nslices = (N3 + cpb - 1) / cpb;
}

BlendBwdGeom blend_bwd_geom(int B, int N3) {
  BlendBwdGeom g;
  g.nmt = (B + 31) / 32;
  bwd_geometry(B, N3, &g.nslices, &g.cols_per_block);
  g.part_floats = (size_t)g.nslices * g.nmt * 32 * BW_NO;
  return g;
}

int launch_blend_bwd_partials(const float *dv_posed, const float *blend_t, int B, int N3, float *part,
                              hipStream_t st) {
  const BlendBwdGeom g = blend_bwd_geom(B, N3);
  const int grid = ((g.nslices + 7) / 8) * 8 * g.nmt;
  hipLaunchKernelGGL(blend_bwd_kernel, dim3(grid), dim3(256), 0, st, dv_posed, blend_t, B, N3, g.cols_per_block,
                     g.nslices, g.nmt, part);
  SMPLR_LAUNCH_CHECK("blend_bwd_kernel");
  return 0;
}

}  // namespace smplr

extern "C" {

int smplr_blend_fwd(const float *coef, const float *blend, const float *v_template, int B, int N3,
                    float *v_posed, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && N3 > 0, "smplr_blend_fwd: bad sizes B=%d N3=%d", B, N3);
  if (B == 0) return 0;
  SMPLR_REQUIRE(coef && blend && v_template && v_posed, "smplr_blend_fwd: null pointer");
  const int nstrips = (N3 + FW_WN - 1) / FW_WN, ngroups = (B + 4 * FW_BM - 1) / (4 * FW_BM);
  const size_t lds = (size_t)4 * KP * A_LD * sizeof(float);     // 116,160 B
  static bool attr_set = false;
  if (!attr_set) {
    SMPLR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(blend_fwd_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(blend_fwd_kernel, dim3(nstrips, ngroups), dim3(256), lds, as_stream(stream), coef, blend,
                     v_template, B, N3, v_posed);
  SMPLR_LAUNCH_CHECK("smplr_blend_fwd");
  return 0;
}

size_t smplr_blend_bwd_workspace(int B, int N3) {
  using namespace smplr;
  if (B <= 0 || N3 <= 0) return 0;
  int ns, cpb;
  bwd_geometry(B, N3, &ns, &cpb);
  return (size_t)ns * ((B + 31) / 32) * 32 * BW_NO * sizeof(float);
}

int smplr_blend_bwd(const float *dv_posed, const float *blend_t, int B, int N3, float *dcoef,
                    void *workspace, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && N3 > 0, "smplr_blend_bwd: bad sizes B=%d N3=%d", B, N3);
  if (B == 0) return 0;
  SMPLR_REQUIRE(dv_posed && blend_t && dcoef && workspace, "smplr_blend_bwd: null pointer");
  const BlendBwdGeom g = blend_bwd_geom(B, N3);
  const int ns = g.nslices, nmt = g.nmt;
  int rc = launch_blend_bwd_partials(dv_posed, blend_t, B, N3, reinterpret_cast<float *>(workspace),
                                     as_stream(stream));
  if (rc) return rc;
  hipLaunchKernelGGL(blend_bwd_reduce_kernel, dim3((B * KP + 255) / 256), dim3(256), 0,
                     as_stream(stream), reinterpret_cast<const float *>(workspace), B, ns, nmt, dcoef);
  SMPLR_LAUNCH_CHECK("smplr_blend_bwd(reduce)");
  return 0;
}

}  // extern "C"
