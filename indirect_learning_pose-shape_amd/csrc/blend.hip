// K2 blend-shape GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact f32 FMA chain).
//
//   forward : v_posed (B,N3) = coef (B,220) x blend (220,N3) + v_template      (N3 = 20670)
//   backward: dcoef  (B,220) = dv_posed (B,N3) x blend^T                         (split-K)
//
// Reference: keras_smpl/batch_smpl.py:106-108 (shape blend, K.dot) and :126-128 (pose blend,
// K.dot) fused into one contraction over [beta | pose_feature].
//
// Forward tiling: workgroup = 4 waves = 128 meshes x one 96-column tile, each wave a 32x96 tile
// (3 accumulators); both operands are read from global memory directly in MFMA layout (no LDS).
//
// Backward tiling: workgroup = 32 meshes x all 220 outputs (7 tiles) x one slice of columns,
// against the TRANSPOSED constant blendT (N3 x 224) so that no operand of the big matrix needs a
// transpose; the 4 waves split the slice, are summed in LDS in a fixed order, and the per-slice
// partials are summed by a second kernel in slice order (deterministic, no atomics).
#include "common.h"

namespace smplr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KP = SMPLR_KPAD;  // 220
constexpr int FW_BM = 32;
constexpr int FW_NT = 3;      // column tiles (accumulators) per wave: B is loaded NT columns per lane
constexpr int FW_BN = 32 * FW_NT;
constexpr int FW_NB = 5;      // k-steps (of 2 rows) per register batch: 22 batches x 5 x 2 = 220
constexpr int FW_DEPTH = 6;   // batches in flight ahead of the one being multiplied (60 loads <= 63)

template <int N> struct VecN;
template <> struct VecN<1> { typedef float type; };
template <> struct VecN<2> { typedef float type __attribute__((ext_vector_type(2), aligned(4))); };
template <> struct VecN<3> { typedef float type __attribute__((ext_vector_type(3), aligned(4))); };
template <> struct VecN<4> { typedef float type __attribute__((ext_vector_type(4), aligned(4))); };

// Forward: grid (column tiles of 96, groups of 128 meshes); workgroup = 4 waves = the 4 mesh tiles
// of ONE column tile, each wave three 32x32 accumulators.  Both operands are read straight from
// global memory in MFMA layout, nothing is staged or transposed:
//   A (coef, k-major (220, ldc)): lane (i, h) reads coef[2s+h][m0+i]
//   B (blend, row-major (220, N3)): lane (i, h) reads blend[2s+h][c0 + 3i .. 3i+2] with ONE dwordx3
//     load; accumulator t owns the columns c0 + 3i + t (any column permutation is a valid tile)
// so every wave-load covers whole contiguous row segments and a k-step costs 2 loads for 3 MFMAs.
// A wave may have 63 loads outstanding: 6 batches of 5 k-steps (60 loads = 90 MFMAs ~ 2.4 us of
// matrix work) run ahead of the batch being multiplied, which covers an HBM round trip without
// help from other waves (there is about one wave per SIMD).  The four waves issue identical B
// loads (L1 serves three of them), coef (112 KB) stays in L2.  Measured alone: 16 us, against
// 29 us for the previous tiling that staged coef^T through 116 KB of LDS per workgroup.
__global__ __launch_bounds__(256) void blend_fwd_kernel(const float *__restrict__ coef,
                                                        const float *__restrict__ blend,
                                                        const float *__restrict__ vt, int B, int N3,
                                                        int ldc, float *__restrict__ out) {
  typedef typename VecN<FW_NT>::type bvec;
  constexpr int NBATCH = KP / (2 * FW_NB);
  static_assert(NBATCH * FW_NB * 2 == KP, "batches cover K exactly");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = (blockIdx.y * 4 + wave) * FW_BM;
  if (m0 >= B) return;                                   // this wave's mesh tile is empty
  const int i = lane & 31, h = lane >> 5;
  const int c = blockIdx.x * FW_BN + FW_NT * i;
  const int cc = c + FW_NT <= N3 ? c : N3 - FW_NT;       // clamped; discarded in the epilogue
  const float *ap = coef + (size_t)h * ldc + m0 + i;     // m0 + i < ldc (padded columns are never stored)
  const float *bp = blend + (size_t)h * N3 + cc;
  const size_t arow = (size_t)ldc, brow = (size_t)N3;

  f32x16 acc[FW_NT];
#pragma unroll
  for (int t = 0; t < FW_NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  // ring of DEPTH+1 register batches, fully unrolled (static indices); sched_barrier pins each
  // batch of loads above the MFMAs it overlaps (left alone the scheduler sinks loads to their uses)
  float a[FW_DEPTH + 1][FW_NB];
  bvec b[FW_DEPTH + 1][FW_NB];
#define SMPLR_LOAD_BATCH(slot, q)                                                        \
  _Pragma("unroll") for (int s2 = 0; s2 < FW_NB; ++s2) {                                 \
    const int kk = 2 * ((q) * FW_NB + s2);                                               \
    a[slot][s2] = ap[(size_t)kk * arow];                                                 \
    b[slot][s2] = *reinterpret_cast<const bvec *>(bp + (size_t)kk * brow);               \
  }
#pragma unroll
  for (int q = 0; q < FW_DEPTH; ++q) { SMPLR_LOAD_BATCH(q % (FW_DEPTH + 1), q) }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < NBATCH; ++q) {
    if (q + FW_DEPTH < NBATCH) { SMPLR_LOAD_BATCH((q + FW_DEPTH) % (FW_DEPTH + 1), q + FW_DEPTH) }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < FW_NB; ++s2)
#pragma unroll
      for (int t = 0; t < FW_NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q % (FW_DEPTH + 1)][s2], b[q % (FW_DEPTH + 1)][s2][t],
                                                      acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef SMPLR_LOAD_BATCH

  float base[FW_NT];
#pragma unroll
  for (int t = 0; t < FW_NT; ++t) base[t] = vt[min(c + t, N3 - 1)];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    const int m = m0 + row;
    if (m < B) {
      float *o = out + (size_t)m * N3 + c;
      if (c + FW_NT <= N3) {
        bvec v;
#pragma unroll
        for (int t = 0; t < FW_NT; ++t) v[t] = acc[t][r] + base[t];
        *reinterpret_cast<bvec *>(o) = v;
      } else {
#pragma unroll
        for (int t = 0; t < FW_NT; ++t)
          if (c + t < N3) o[t] = acc[t][r] + base[t];
      }
    }
  }
}

// ---------------------------------------------------------------- backward (split-K)
// dcoef[m][o] = sum_c dvp[m][c] * blendT[c][o].  blendT = blend transposed, rows padded to 224
// floats.  Same recipe as the forward, no LDS in the main loop:
//   A (dvp, row-major (B, N3)): lane (i, h) reads the float4 dvp[m0+i][8g+4h .. 8g+4h+3]; MFMA t of
//     group g contracts c = 8g + 4h + t over h (a permutation of the c order inside the sum)
//   B (blendT (N3, 224)): lane (i, h) reads blendT[8g+4h+t][7i .. 7i+6] with a dwordx4 + a dwordx3
//     load; accumulator u owns the outputs 7i + u
// so a group of 8 columns costs 9 loads for 28 MFMAs, and 5 groups (45 loads, 140 MFMAs ~ 3.7 us
// of matrix work) run ahead of the one being multiplied.  The 4 waves of a block split the block's
// column slice and are summed in LDS in a fixed order; per-slice partials are summed in slice
// order by pose_bwd (fused path) or blend_bwd_reduce_kernel: deterministic, no atomics.
constexpr int BW_NT = 7;             // 7 x 32 = 224 >= 220 outputs
constexpr int BW_NO = 224;           // blendT row stride and partial row stride
constexpr int BW_G = 8;              // columns per group
constexpr int BW_DEPTH = 5;          // groups in flight ahead of the one being multiplied

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));

__global__ __launch_bounds__(256) void blend_bwd_kernel(const float *__restrict__ dvp,
                                                        const float *__restrict__ blendT, int B, int N3,
                                                        int cols_per_block, int nslices, int nmt,
                                                        float *__restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float sR[];   // 4 x [32][224] floats = 114,688 B: one tile per wave
  // XCD-aware map: the nmt mesh tiles of one column slice (same blendT rows) share an XCD's L2.
  const int bid = blockIdx.x;
  const int group = bid / (8 * nmt), within = bid % (8 * nmt);
  const int slice = group * 8 + (within & 7), mt = within >> 3;
  if (slice >= nslices) return;
  const int m0 = mt * 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;
  const int cw = cols_per_block / 4;          // multiple of BW_G
  const int c_beg = slice * cols_per_block + wave * cw;
  const int c_end = min(c_beg + cw, N3);
  const int ncols = c_beg < c_end ? c_end - c_beg : 0;
  const int ngroups = ncols / BW_G;           // full groups (wave-uniform); the matrix' ragged end is handled below

  f32x16 acc[BW_NT];
#pragma unroll
  for (int t = 0; t < BW_NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  const int mr = min(m0 + i, B - 1);          // rows beyond B: clamped, their results are never stored
  const float *arow = dvp + (size_t)mr * N3;
  const float *bcol = blendT + 7 * i;

  // group g of this wave: lane (i, h) covers columns c_beg + 8g + 4h .. +3
  f32x4u a[BW_DEPTH + 1];
  f32x4u bx[BW_DEPTH + 1][4];
  f32x3u by[BW_DEPTH + 1][4];
#define SMPLR_LOAD_GROUP(slot, g)                                                          \
  {                                                                                        \
    const int c0_ = c_beg + (g) * BW_G + 4 * h;                                            \
    a[slot] = *reinterpret_cast<const f32x4u *>(arow + c0_);                               \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) {                                     \
      const float *br_ = bcol + (size_t)(c0_ + t_) * BW_NO;                                \
      bx[slot][t_] = *reinterpret_cast<const f32x4u *>(br_);                               \
      by[slot][t_] = *reinterpret_cast<const f32x3u *>(br_ + 4);                           \
    }                                                                                      \
  }
#define SMPLR_MMA_GROUP(slot)                                                              \
  _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) {                                       \
    _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_)                                       \
      acc[u_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot][t_], bx[slot][t_][u_], acc[u_], 0, 0, 0);         \
    _Pragma("unroll") for (int u_ = 0; u_ < 3; ++u_)                                       \
      acc[4 + u_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot][t_], by[slot][t_][u_], acc[4 + u_], 0, 0, 0); \
  }
  // ring of DEPTH+1 register groups; the loop is unrolled by the ring length so slots are static
#pragma unroll
  for (int g = 0; g < BW_DEPTH; ++g)
    if (g < ngroups) SMPLR_LOAD_GROUP(g, g)
  __builtin_amdgcn_sched_barrier(0);
  for (int g0 = 0; g0 < ngroups; g0 += BW_DEPTH + 1) {
#pragma unroll
    for (int u = 0; u < BW_DEPTH + 1; ++u) {
      const int g = g0 + u;
      if (g < ngroups) {                      // wave-uniform
        if (g + BW_DEPTH < ngroups) SMPLR_LOAD_GROUP((u + BW_DEPTH) % (BW_DEPTH + 1), g + BW_DEPTH)
        __builtin_amdgcn_sched_barrier(0);
        SMPLR_MMA_GROUP(u)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#undef SMPLR_LOAD_GROUP
#undef SMPLR_MMA_GROUP
  if (ncols - ngroups * BW_G > 0) {           // wave-uniform: only the wave that owns the end of the matrix
    // (N3 = 8q + 6): clamped scalar loads, surplus A elements zeroed
    const int c0 = c_beg + ngroups * BW_G + 4 * h;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = c0 + t, cl = min(c, N3 - 1);
      const float av = (c < c_end) ? arow[cl] : 0.0f;
      const float *br = bcol + (size_t)cl * BW_NO;
      const f32x4u x = *reinterpret_cast<const f32x4u *>(br);
      const f32x3u y = *reinterpret_cast<const f32x3u *>(br + 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, x[u], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 3; ++u) acc[4 + u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, y[u], acc[4 + u], 0, 0, 0);
    }
  }
  // the 4 waves park their tiles side by side in LDS (all at once: 4 x 28 KB), then every thread
  // sums its elements over the waves in a fixed order and stores the block's partial coalesced;
  // accumulator u of lane (i, h) holds output column 7i + u (stride 7: conflict-free)
  float *mine = sR + wave * (32 * BW_NO);
#pragma unroll
  for (int t = 0; t < BW_NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      mine[row * BW_NO + 7 * i + t] = acc[t][r];
    }
  __syncthreads();
  float *dst = part + ((size_t)slice * nmt + mt) * (32 * BW_NO);
  for (int e = tid * 4; e < 32 * BW_NO; e += 256 * 4) {
    const float4 s0 = *reinterpret_cast<const float4 *>(sR + e);
    const float4 s1 = *reinterpret_cast<const float4 *>(sR + 32 * BW_NO + e);
    const float4 s2 = *reinterpret_cast<const float4 *>(sR + 2 * 32 * BW_NO + e);
    const float4 s3 = *reinterpret_cast<const float4 *>(sR + 3 * 32 * BW_NO + e);
    float4 o;
    o.x = ((s0.x + s1.x) + s2.x) + s3.x;
    o.y = ((s0.y + s1.y) + s2.y) + s3.y;
    o.z = ((s0.z + s1.z) + s2.z) + s3.z;
    o.w = ((s0.w + s1.w) + s2.w) + s3.w;
    *reinterpret_cast<float4 *>(dst + e) = o;
  }
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
  cpb = (cpb + 4 * BW_G - 1) / (4 * BW_G) * (4 * BW_G);   // 4 waves x groups of 8 columns
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
  const size_t lds = (size_t)4 * 32 * BW_NO * sizeof(float);
  static LdsAttrMemo memo = {};                       // once per (kernel, device): common.h
  {
    int rc = ensure_lds_attr(reinterpret_cast<const void *>(blend_bwd_kernel), lds, &memo, "blend_bwd_kernel");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(blend_bwd_kernel, dim3(grid), dim3(256), lds, st, dv_posed, blend_t, B, N3, g.cols_per_block,
                     g.nslices, g.nmt, part);
  SMPLR_LAUNCH_CHECK("blend_bwd_kernel");
  return 0;
}

int launch_blend_bwd_reduce(const float *part, int B, int nslices, int nmt, float *dcoef, hipStream_t st) {
  hipLaunchKernelGGL(blend_bwd_reduce_kernel, dim3((B * KP + 255) / 256), dim3(256), 0, st, part, B, nslices, nmt,
                     dcoef);
  SMPLR_LAUNCH_CHECK("blend_bwd_reduce_kernel");
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
  const int ntiles = (N3 + FW_BN - 1) / FW_BN, ngroups = (B + 4 * FW_BM - 1) / (4 * FW_BM);
  hipLaunchKernelGGL(blend_fwd_kernel, dim3(ntiles, ngroups), dim3(256), 0, as_stream(stream), coef, blend,
                     v_template, B, N3, smplr_coef_ld(B), v_posed);
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
  return launch_blend_bwd_reduce(reinterpret_cast<const float *>(workspace), B, ns, nmt, dcoef, as_stream(stream));
}

}  // extern "C"
