// K2' blend-shape GEMMs on the bf16 matrix cores with fp32-grade operands ("bf16x3").
//
//   forward : v_posed (B,N3) = coef (B,220) x blend (220,N3) + v_template      (N3 = 20670)
//   backward: dcoef  (B,220) = dv_posed (B,N3) x blend^T                         (split-K)
//
// Reference: keras_smpl/batch_smpl.py:106-108 and :126-128 (the two K.dot), as in blend.hip.
//
// gfx950 has no reduced-precision fast path for fp32 operands: v_mfma_f32_32x32x2_f32 runs at the
// vector rate, 1/16 of the bf16 matrix rate, and the two GEMMs above are a fifth of the decoder
// step at B = 128.  Here every fp32 operand is written as the exact sum of three bf16 numbers,
//      a = a_h + a_m + a_l,   a_h = bf16(a), a_m = bf16(a - a_h), a_l = bf16(a - a_h - a_m)
// (round to nearest; 3 x 8 significant bits cover the 24 of an fp32, the residuals are exact in
// fp32), and a product a*b as the six partial products of relative size >= 2^-16,
//      a_h b_h  +  (a_h b_m + a_m b_h)  +  (a_h b_l + a_m b_m + a_l b_h),
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The three
// dropped products are below 2^-24 of |a b|, i.e. below the rounding of an fp32 product itself.
// The leading product and the five corrections go to separate accumulators that are added once
// at the end, so small terms are not absorbed one by one into a large sum.  Six bf16 MFMAs of
// K = 16 take 192 cycles against 512 for the eight fp32 MFMAs of K = 2 they replace, which leaves
// each GEMM with its launch, one round trip, 3.4 us of matrix work and the drain of its output.
// Measured error against float64: tests/test_gpu_parity.py::test_blend3_matches_fp32_path and
// ::test_blend3_wide_dynamic_range (the decoder's vertex error is unchanged at 3.3e-7).
//
// The constant operand is split and laid out in MFMA fragment order ONCE (smplr_blend3_pack), so
// a wave reads each 64-lane fragment with one coalesced 1-KB global_load_dwordx4.  The per-step
// operands: the forward's coef arrives already split, as A-fragments written by pose_fwd (coef3:
// once per mesh instead of once per column tile); the backward's dv_posed is fp32 in memory and
// split in registers (v_cvt_pk_bf16_f32 + exact fp32 residuals).
//   forward operand  [96-col tile][k-tile 14][t 3][split 3][lane 64][8 bf16]:
//       element j of lane (i, h) = split_s(blend[16 kt + 8h + j][96 ct + 3i + t])
//   backward operand [k-tile ceil(N3/16)][out tile 7][split 3][lane 64][8 bf16]:
//       element j of lane (i, h) = split_s(blend[32u + i][16 kt + 8h + j])
// (zeros beyond row 219 / column N3 - 1).
#include "common.h"
#include "pose_device.h"

namespace smplr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));

constexpr int KP = SMPLR_KPAD;       // 220
constexpr int NKT = 14;              // forward k-tiles of 16 (224 >= 220)
constexpr int F3_BN = 96;            // forward column tile (3 MFMA tiles; lane i owns columns 3i..3i+2)
constexpr int B3_NO = 224;           // backward outputs (7 tiles) = partial row stride

#define SMPLR_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
// the six partial products of one (A fragment, B fragment triple); smallest first into `lo`
#define SMPLR_MFMA_X3(A, bh, bm, bl, hi, lo) \
  lo = SMPLR_MFMA16(A.l, bh, lo);            \
  lo = SMPLR_MFMA16(A.h, bl, lo);            \
  lo = SMPLR_MFMA16(A.m, bm, lo);            \
  lo = SMPLR_MFMA16(A.m, bh, lo);            \
  lo = SMPLR_MFMA16(A.h, bm, lo);            \
  hi = SMPLR_MFMA16(A.h, bh, hi);

// ------------------------------------------------------------------------------------------------
// one-off packing of the constant (not on the hot path)
__global__ __launch_bounds__(256) void blend3_pack_fwd_kernel(const float *__restrict__ blend, int N3, int ntile,
                                                              uint4 *__restrict__ pk) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;           // (ct, kt, t, lane)
  if (e >= (size_t)ntile * NKT * 3 * 64) return;
  const int lane = e & 63, t = (e >> 6) % 3, kt = (e / 192) % NKT, ct = e / (192 * NKT);
  const int i = lane & 31, h = lane >> 5;
  const int col = ct * F3_BN + 3 * i + t;
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kt * 16 + 8 * h + j;
    x[j] = (k < KP && col < N3) ? blend[(size_t)k * N3 + col] : 0.0f;
  }
  const Frag3 f = split8(x);
  uint4 *o = pk + ((size_t)(ct * NKT + kt) * 9 + t * 3) * 64 + lane;
  o[0] = __builtin_bit_cast(uint4, f.h);
  o[64] = __builtin_bit_cast(uint4, f.m);
  o[128] = __builtin_bit_cast(uint4, f.l);
}

__global__ __launch_bounds__(256) void blend3_pack_bwd_kernel(const float *__restrict__ blend, int N3, int nkt,
                                                              uint4 *__restrict__ pk) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;           // (kt, u, lane)
  if (e >= (size_t)nkt * 7 * 64) return;
  const int lane = e & 63, u = (e >> 6) % 7, kt = e / (64 * 7);
  const int i = lane & 31, h = lane >> 5;
  const int o_ = 32 * u + i;
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = kt * 16 + 8 * h + j;
    x[j] = (o_ < KP && c < N3) ? blend[(size_t)o_ * N3 + c] : 0.0f;
  }
  const Frag3 f = split8(x);
  uint4 *o = pk + ((size_t)(kt * 7 + u) * 3) * 64 + lane;
  o[0] = __builtin_bit_cast(uint4, f.h);
  o[64] = __builtin_bit_cast(uint4, f.m);
  o[128] = __builtin_bit_cast(uint4, f.l);
}

// ------------------------------------------------------------------------------------------------
// coef (220, ld) k-major fp32 -> coef3, the A-fragment layout smplr_pose_fwd writes directly
// ([mesh tile][k-tile 14][split 3][lane 64] x 16 B; lane (i, h) = mesh 32 mt + i, k = 16 kt + 8h + j).
__global__ __launch_bounds__(256) void coef3_pack_kernel(const float *__restrict__ coef, int B, int ldc,
                                                         u32x4 *__restrict__ coef3) {
  const int e = blockIdx.x * 256 + threadIdx.x;                      // (mt, kt, lane)
  const int nmt = (B + 31) / 32;
  if (e >= nmt * NKT * 64) return;
  const int lane = e & 63, kt = (e >> 6) % NKT, mt = e / (64 * NKT);
  const int i = lane & 31, h = lane >> 5, m = mt * 32 + i;
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kt * 16 + 8 * h + j;
    x[j] = (k < KP && m < B) ? coef[(size_t)k * ldc + m] : 0.0f;
  }
  const Frag3 f = split8(x);
  u32x4 *o = coef3 + ((size_t)(mt * NKT + kt) * 3) * 64 + lane;
  o[0] = __builtin_bit_cast(u32x4, f.h);
  o[64] = __builtin_bit_cast(u32x4, f.m);
  o[128] = __builtin_bit_cast(u32x4, f.l);
}

// ------------------------------------------------------------------------------------------------
// Forward.  grid (column tiles of 96, groups of 128 meshes); workgroup = 4 waves = the 4 mesh
// tiles of one column tile (identical requests for the constant: L1 serves three of the four), each
// wave a 32 x 96 tile = 3 (leading, correction) accumulator pairs.  Per k-tile a lane requests
// 3 x 16 B of coef3 and 9 x 16 B of the packed constant, every request a contiguous 1 KB per wave;
// F3_DEPTH k-tiles are in flight ahead of the one being multiplied.  Both operands arrive as
// fragments: no conversion and no address arithmetic in the loop.
// Measured (B = 128, rocprofv3): 14 us against 18.7 us for the fp32 matrix-core kernel; with the
// constant served from L2 it takes 12 us and without its stores 10 us, so what is left is launch,
// the first round trip, 252 MFMAs and the drain of 10.6 MB of output.  Rejected on the way (same
// numerics): the constant shared through LDS with each wave fetching a quarter (16.4 us), eight
// waves splitting K in pairs with an LDS hand-over (15.4 us).
// (two k-tiles ahead and at most 256 registers: TWO workgroups per CU, so that one's first round trip and output drain
// hide under the other's loop - what counts once the grid is several rounds deep: B = 2 048 143.6 -> 136.4 us, and
// 13.0 -> 12.4 us at B = 128; three ahead needed 308 registers, one wave per SIMD)
constexpr int F3_DEPTH = 2;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void blend3_fwd_kernel(const u32x4 *__restrict__ coef3,
                                                         const u32x4 *__restrict__ pk,
                                                         const float *__restrict__ vt, int B, int N3,
                                                         float *__restrict__ out) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = (blockIdx.y * 4 + wave) * 32;
  if (m0 >= B) return;                                   // this wave's mesh tile is empty
  const int i = lane & 31, h = lane >> 5;
  const int c = blockIdx.x * F3_BN + 3 * i;
  const u32x4 *ap = coef3 + (size_t)(m0 >> 5) * (NKT * 3 * 64) + lane;
  const u32x4 *bp = pk + (size_t)blockIdx.x * (NKT * 9 * 64) + lane;

  f32x16 hi[3], lo[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { hi[t][r] = 0.0f; lo[t][r] = 0.0f; }

  u32x4 a[F3_DEPTH + 1][3], b[F3_DEPTH + 1][9];
#define SMPLR_LOAD_KT(slot, kt)                                                         \
  {                                                                                     \
    _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_) a[slot][s_] = ap[((kt) * 3 + s_) * 64]; \
    _Pragma("unroll") for (int j_ = 0; j_ < 9; ++j_) b[slot][j_] = bp[((kt) * 9 + j_) * 64]; \
  }
#pragma unroll
  for (int kt = 0; kt < F3_DEPTH; ++kt) { SMPLR_LOAD_KT(kt % (F3_DEPTH + 1), kt) }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    if (kt + F3_DEPTH < NKT) { SMPLR_LOAD_KT((kt + F3_DEPTH) % (F3_DEPTH + 1), kt + F3_DEPTH) }
    __builtin_amdgcn_sched_barrier(0);
    const int s = kt % (F3_DEPTH + 1);
    Frag3 A;
    A.h = __builtin_bit_cast(bf16x8, a[s][0]);
    A.m = __builtin_bit_cast(bf16x8, a[s][1]);
    A.l = __builtin_bit_cast(bf16x8, a[s][2]);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const bf16x8 bh = __builtin_bit_cast(bf16x8, b[s][t * 3 + 0]);
      const bf16x8 bm = __builtin_bit_cast(bf16x8, b[s][t * 3 + 1]);
      const bf16x8 bl = __builtin_bit_cast(bf16x8, b[s][t * 3 + 2]);
      SMPLR_MFMA_X3(A, bh, bm, bl, hi[t], lo[t])
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef SMPLR_LOAD_KT

  float base[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) base[t] = vt[min(c + t, N3 - 1)];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    const int m = m0 + row;
    if (m < B) {
      float *o = out + (size_t)m * N3 + c;
      if (c + 3 <= N3) {
        f32x3u v;
#pragma unroll
        for (int t = 0; t < 3; ++t) v[t] = (hi[t][r] + lo[t][r]) + base[t];
        *reinterpret_cast<f32x3u *>(o) = v;
      } else {
#pragma unroll
        for (int t = 0; t < 3; ++t)
          if (c + t < N3) o[t] = (hi[t][r] + lo[t][r]) + base[t];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Pose + blend forward in ONE launch (what the decoder runs; pose_fwd_kernel + blend3_fwd_kernel remain as the
// stand-alone entry points and give the same bits).  pose_fwd is a chain of dependent steps on one wave per mesh -
// 10 us at B = 128 for 5 kFLOP per mesh - and the GEMM waited for it only because it read the coefficient
// fragments that kernel wrote.  Here
//   * the first ceil(B / 8) workgroups ARE pose_fwd (one wave per mesh: Rs, J, A, J_transformed), and
//   * a GEMM workgroup is 8 waves, two per SIMD.  Phase 1, all eight: the coefficient rows [beta | Rs[1:] - I | 0] of
//     the workgroup's 4 x 32 meshes into 29 KB of LDS per mesh tile - Rodrigues of two joints per mesh and step with
//     the same `rodrigues()` as pose_fwd, six steps per wave (wave w and its partner w + 4 share tile w), wave w having
//     requested its first fragments of the constant beforehand.  ONE workgroup barrier.  Phase 2, waves 0..3 (the
//     others leave): the GEMM tile as in blend3_fwd_kernel, each k-tile's eight values per lane read from LDS and
//     split into the three bf16 terms with the same `split8()` - the operand is the one pose_fwd would have
//     written, bit for bit.  (A wave alone on a SIMD issues one vector instruction per 5 cycles whatever the matrix
//     pipe does: the ~2 000 vector instructions of a tile's rows cost 6 us on the multiplying wave alone, in front
//     of the loop or interleaved with it alike; on two waves per SIMD in front of the barrier about 2.)
// Nothing is handed from workgroup to workgroup: the two roles write different outputs, and the chain's latency is
// hidden under the GEMM on compute units the GEMM leaves idle (216 + 16 workgroups at B = 128 on 256 CUs).
constexpr int FC_LD = 228;           // floats per mesh row of the staged coefficients (16-B aligned rows, bank-skewed)
constexpr int F3F_DEPTH = 2;         // k-tiles of the constant in flight (two waves per SIMD: 256 registers each)
constexpr int PB_MPB = 8;            // meshes per workgroup in the pose role

__global__ __launch_bounds__(512) void pose_blend3_fwd_kernel(
    const float *__restrict__ x, int x_stride, int num_cam, int B, const float *__restrict__ J_template,
    const float *__restrict__ J_dirs, const int *__restrict__ parents, const u32x4 *__restrict__ pk,
    const float *__restrict__ vt, int N3, int ntiles, int npose, float *__restrict__ Rs_out,
    float *__restrict__ J_out, float *__restrict__ A_out, float *__restrict__ newJ_out, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((int)blockIdx.x < npose) {                           // ---- role 1: pose_fwd for 8 meshes
    PoseLds *lds = reinterpret_cast<PoseLds *>(smem);
    float *sCoef = smem + (PB_MPB * sizeof(PoseLds) + 15) / 16 * 4;
    pose_fwd_wave(x, x_stride, num_cam, B, blockIdx.x * PB_MPB + wave, lane, J_template, J_dirs, parents, nullptr, 0,
                  nullptr, Rs_out, J_out, A_out, newJ_out, lds[wave], sCoef + wave * 224);
    return;
  }
  // ---- role 2: a 96-column tile of the GEMM for 4 x 32 meshes
  const int gb = blockIdx.x - npose;
  const int ct = gb % ntiles, grp = gb / ntiles;
  const int mw = wave & 3;                                  // mesh tile of this wave (multiplier w, producer w + 4)
  const int m0 = (grp * 4 + mw) * 32;
  const bool empty = m0 >= B;                               // this pair's mesh tile is empty: only the barrier to keep
  const int i = lane & 31, h = lane >> 5;
  float *sc = smem + mw * (32 * FC_LD);
  if (empty) {
    __syncthreads();
    return;
  }
  // ---- phase 1, all eight waves: the coefficient rows.  Lane (i, h) takes joint 2 s + 1 + h of mesh i in step s; the
  // multiplier wave w does steps 0..5 of its tile (after requesting its first fragments of the constant), its
  // partner w + 4 steps 6..11 and the betas / padding (rows beyond B repeat mesh B - 1)
  const int c = ct * F3_BN + 3 * i;
  const u32x4 *bp = pk + (size_t)ct * (NKT * 9 * 64) + lane;
  u32x4 b[F3F_DEPTH + 1][9];
#define SMPLR_LOAD_B(slot, kt) \
  { _Pragma("unroll") for (int j_ = 0; j_ < 9; ++j_) b[slot][j_] = bp[((kt) * 9 + j_) * 64]; }
  if (wave < 4) {
#pragma unroll
    for (int kt = 0; kt < F3F_DEPTH; ++kt) { SMPLR_LOAD_B(kt % (F3F_DEPTH + 1), kt) }
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    const int s0 = wave < 4 ? 0 : 6;
    float th[6][3];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int j = min(2 * (s0 + q) + 1 + h, 23);
      const float *xr = x + (size_t)min(m0 + i, B - 1) * x_stride + num_cam + 3 * j;
      th[q][0] = xr[0]; th[q][1] = xr[1]; th[q][2] = xr[2];
    }
    if (wave >= 4) {
      float be[5];
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        const int e = lane + 64 * t, mi = e / 10, k = e - 10 * mi;
        be[t] = x[(size_t)min(m0 + mi, B - 1) * x_stride + num_cam + 72 + k];
      }
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        const int e = lane + 64 * t, mi = e / 10, k = e - 10 * mi;
        sc[mi * FC_LD + k] = be[t];
      }
      for (int e = lane; e < 32 * 7; e += 64) sc[(e / 7) * FC_LD + 217 + e % 7] = 0.0f;       // 217..223
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int j = 2 * (s0 + q) + 1 + h;
      float R[9];
      rodrigues(th[q], R);
      if (j <= 23) {
#pragma unroll
        for (int e = 0; e < 9; ++e)
          sc[i * FC_LD + 10 + 9 * (j - 1) + e] = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
      }
    }
  }
  // All rows are in LDS before any multiplier reads one: the workgroup barrier is the only hand-over.  A per-step
  // counter in LDS with the multiplier starting on k-tile kt as soon as steps 0..kt were published was WRONG about
  // once in 30 launches - a tile of v_posed off by 1e-4, run-to-run (tools/probes/det_step.py) - with the stores
  // drained (`s_waitcnt lgkmcnt(0)`) and even read back before the counter's store; a delay of 1 280 cycles before
  // publishing made it disappear, waiting for two or three steps more made it rarer.  Not understood, so not
  // used: nothing but `s_barrier` orders one wave's LDS stores before another wave's loads here.
  __syncthreads();
  if (wave >= 4) return;
  // ---- phase 2, waves 0..3: multiply
  f32x16 hi[3], lo[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { hi[t][r] = 0.0f; lo[t][r] = 0.0f; }
  const float *arow = sc + i * FC_LD + 8 * h;
  // the eight coefficients of k-tile kt for this lane
#define SMPLR_WAIT_READ(kt, x0_, x1_)                                                                       \
  {                                                                                                         \
    x0_ = *reinterpret_cast<const float4 *>(arow + 16 * (kt));                                              \
    x1_ = *reinterpret_cast<const float4 *>(arow + 16 * (kt) + 4);                                          \
  }
  float4 x0, x1;
  SMPLR_WAIT_READ(0, x0, x1)
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    if (kt + F3F_DEPTH < NKT) { SMPLR_LOAD_B((kt + F3F_DEPTH) % (F3F_DEPTH + 1), kt + F3F_DEPTH) }
    __builtin_amdgcn_sched_barrier(0);
    const int s = kt % (F3F_DEPTH + 1);
    const float xk[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    const Frag3 A = split8(xk);
    if (kt + 1 < NKT) SMPLR_WAIT_READ(kt + 1, x0, x1)        // the next tile's values arrive under this tile's MFMAs
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const bf16x8 bh = __builtin_bit_cast(bf16x8, b[s][t * 3 + 0]);
      const bf16x8 bm = __builtin_bit_cast(bf16x8, b[s][t * 3 + 1]);
      const bf16x8 bl = __builtin_bit_cast(bf16x8, b[s][t * 3 + 2]);
      SMPLR_MFMA_X3(A, bh, bm, bl, hi[t], lo[t])
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef SMPLR_WAIT_READ
#undef SMPLR_LOAD_B

  float base[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) base[t] = vt[min(c + t, N3 - 1)];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    const int m = m0 + row;
    if (m < B) {
      float *o = out + (size_t)m * N3 + c;
      if (c + 3 <= N3) {
        f32x3u v;
#pragma unroll
        for (int t = 0; t < 3; ++t) v[t] = (hi[t][r] + lo[t][r]) + base[t];
        *reinterpret_cast<f32x3u *>(o) = v;
      } else {
#pragma unroll
        for (int t = 0; t < 3; ++t)
          if (c + t < N3) o[t] = (hi[t][r] + lo[t][r]) + base[t];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Backward (split-K over column slices).  Workgroup = (slice, mesh tile of 32); its 4 waves own
// DIFFERENT output tiles (wave w: tiles w and w + 4; 7 tiles, so wave 3 has one), hence no
// reduction inside the block and no LDS: the dv_posed fragment (8 consecutive floats of row
// m0 + i, two dwordx4, split in registers) is requested by all four waves (L1), the packed constant
// only by its owner.  Per-slice partials (slice, mesh tile, 32, 224) are summed in slice order by
// pose_bwd (fused path) or blend_bwd_reduce_kernel: deterministic, no atomics.
// Measured (B = 128): 14 us against 18.3 us for the fp32 matrix-core kernel; eight waves splitting
// each slice in halves with an LDS hand-over: 15.7 us.
// Two builds of the loop: B3_DEPTH k-tiles in flight ahead of the one being multiplied.  Five (288 registers, one
// workgroup per CU) when the grid is several rounds deep and the constant streams from L2 at its limit; two (190
// registers, amdgpu_waves_per_eu 2: TWO workgroups per CU, one's first round trip and partial stores under the other's
// loop) below 512 meshes - B = 128: 12.1 us against 13.1, B = 512: 47.7 / 47.7, B = 2 048: 200.8 against 182.5.
// From 512 meshes on a workgroup takes MT = 2 mesh tiles (three ahead, 324 + 128 registers): B = 512: 38.0 against 47.3,
// 1 024: 67.6 against 91.7, 2 048: 132 against 181 us (four tiles: 216 - half the workgroups and a ring that spills).
constexpr int B3_BIG_DEPTH = 3;   // k-tiles ahead in the two-tile form (ring of 4 x 40 registers + 128 accumulators)

template <int B3_DEPTH, int WPE, int MT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void blend3_bwd_kernel(
    const float *__restrict__ dvp, const u32x4 *__restrict__ pk, int B, int N3, int ktps, int nslices, int nmt,
    float *__restrict__ part) {
  // MT mesh tiles of 32 per workgroup (nmt counts the workgroups' tile GROUPS): every fragment of the constant a wave
  // requests meets MT fragments of dv_posed - from 512 meshes on the kernel is bound by what L2 delivers of the
  // constant (each XCD streams its slices' 3.4 MB once per tile group), and two tiles per group halve that.
  // XCD-aware map: the nmt tile groups of one column slice (same constant rows) share an XCD's L2.
  const int bid = blockIdx.x;
  const int group = bid / (8 * nmt), within = bid % (8 * nmt);
  const int slice = group * 8 + (within & 7), mt = within >> 3;
  if (slice >= nslices) return;
  const int m0 = mt * 32 * MT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int u0 = wave, u1 = wave + 4 < 7 ? wave + 4 : 6;      // wave 3: second tile is a discarded copy of tile 6
  const int nkt = (N3 + 15) / 16, nfull = N3 / 16;
  const int kt_beg = slice * ktps;
  const int kt_end = min(kt_beg + ktps, nkt);
  const int kt_main = min(kt_end, nfull);                     // whole k-tiles; the matrix' ragged last tile comes after
  const int n = kt_main > kt_beg ? kt_main - kt_beg : 0;

  f32x16 hi[MT][2], lo[MT][2];
#pragma unroll
  for (int j = 0; j < MT; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { hi[j][t][r] = 0.0f; lo[j][t][r] = 0.0f; }

  const float *arow[MT];
  int mr[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    mr[j] = min(m0 + 32 * j + i, B - 1);                      // rows beyond B: clamped, never stored
    arow[j] = dvp + (size_t)mr[j] * N3 + 8 * h;
  }
  const u32x4 *b0 = pk + (size_t)u0 * 192 + lane, *b1 = pk + (size_t)u1 * 192 + lane;

  f32x4u ra0[B3_DEPTH + 1][MT], ra1[B3_DEPTH + 1][MT];
  u32x4 rb[B3_DEPTH + 1][6];
  // (requests are unconditional with clamped tile indices: a branch around them would park the ring in scratch)
#define SMPLR_LOAD_KT(slot, g)                                                    \
  {                                                                               \
    const int kt_ = min(kt_beg + (g), nfull - 1);                                 \
    _Pragma("unroll") for (int j_ = 0; j_ < MT; ++j_) {                           \
      ra0[slot][j_] = *reinterpret_cast<const f32x4u *>(arow[j_] + (size_t)kt_ * 16);     \
      ra1[slot][j_] = *reinterpret_cast<const f32x4u *>(arow[j_] + (size_t)kt_ * 16 + 4); \
    }                                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_) {                            \
      rb[slot][s_] = b0[((size_t)kt_ * 21 + s_) * 64];                            \
      rb[slot][3 + s_] = b1[((size_t)kt_ * 21 + s_) * 64];                        \
    }                                                                             \
  }
#define SMPLR_MMA_KT(slot)                                                         \
  {                                                                               \
    _Pragma("unroll") for (int j_ = 0; j_ < MT; ++j_) {                           \
      const float x_[8] = {ra0[slot][j_][0], ra0[slot][j_][1], ra0[slot][j_][2], ra0[slot][j_][3],  \
                           ra1[slot][j_][0], ra1[slot][j_][1], ra1[slot][j_][2], ra1[slot][j_][3]}; \
      const Frag3 A = split8(x_);                                                 \
      _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) {                          \
        const bf16x8 bh = __builtin_bit_cast(bf16x8, rb[slot][3 * t_ + 0]);       \
        const bf16x8 bm = __builtin_bit_cast(bf16x8, rb[slot][3 * t_ + 1]);       \
        const bf16x8 bl = __builtin_bit_cast(bf16x8, rb[slot][3 * t_ + 2]);       \
        SMPLR_MFMA_X3(A, bh, bm, bl, hi[j_][t_], lo[j_][t_])                      \
      }                                                                           \
    }                                                                             \
  }
#pragma unroll
  for (int g = 0; g < B3_DEPTH; ++g) { SMPLR_LOAD_KT(g, min(g, max(n - 1, 0))) }
  __builtin_amdgcn_sched_barrier(0);
  for (int g0 = 0; g0 < n; g0 += B3_DEPTH + 1) {
#pragma unroll
    for (int u = 0; u < B3_DEPTH + 1; ++u) {
      const int g = g0 + u;
      if (g < n) {                             // block-uniform
        SMPLR_LOAD_KT((u + B3_DEPTH) % (B3_DEPTH + 1), min(g + B3_DEPTH, n - 1))
        __builtin_amdgcn_sched_barrier(0);
        SMPLR_MMA_KT(u)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#undef SMPLR_LOAD_KT
#undef SMPLR_MMA_KT
  if (kt_end > kt_main) {                      // block-uniform: the slice that owns the end of the matrix (N3 % 16 != 0)
    const int kt = nfull;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      float x[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int cc = kt * 16 + 8 * h + q;    // columns >= N3 meet zeros in the packed constant
        x[q] = dvp[(size_t)mr[j] * N3 + min(cc, N3 - 1)];
      }
      const Frag3 A = split8(x);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const u32x4 *bq = pk + ((size_t)kt * 21 + (t == 0 ? u0 : u1) * 3) * 64 + lane;
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bq[0]);
        const bf16x8 bm = __builtin_bit_cast(bf16x8, bq[64]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, bq[128]);
        SMPLR_MFMA_X3(A, bh, bm, bl, hi[j][t], lo[j][t])
      }
    }
  }
  // accumulator register r of lane (i, h) = mesh row (r&3) + 8(r>>2) + 4h, output 32u + i; the partials keep their
  // layout (slice, mesh tile of 32, 32, B3_NO) whatever MT
  const int ntile = (B + 31) / 32;
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int tile = mt * MT + j;
    if (tile >= ntile) break;                  // block-uniform: the last group of an odd tile count
    float *dst = part + ((size_t)slice * ntile + tile) * (32 * B3_NO);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t == 1 && wave + 4 >= 7) break;      // wave-uniform
      const int u = t == 0 ? u0 : u1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        dst[row * B3_NO + 32 * u + i] = hi[j][t][r] + lo[j][t][r];
      }
    }
  }
}

// k-tiles per slice: about one workgroup per CU, and at most 60 slices so that pose_bwd sums a
// mesh's partials with one batch of loads.
Blend3BwdGeom blend3_bwd_geom(int B, int N3) {
  Blend3BwdGeom g;
  g.nmt = (B + 31) / 32;
  const int nkt = (N3 + 15) / 16;
  const int ngrp = B >= 512 ? (g.nmt + 1) / 2 : g.nmt;     // workgroups per slice (two mesh tiles each from 512 meshes on)
  int target = (256 + ngrp - 1) / ngrp;
  if (target < 8) target = 8;
  if (target > 60) target = 60;
  g.ktps = (nkt + target - 1) / target;
  g.nslices = (nkt + g.ktps - 1) / g.ktps;
  g.part_floats = (size_t)g.nslices * g.nmt * 32 * B3_NO;
  return g;
}

int launch_blend3_bwd_partials(const float *dv_posed, const void *pk_bwd, int B, int N3, float *part,
                               hipStream_t st) {
  const Blend3BwdGeom g = blend3_bwd_geom(B, N3);
  if (B < 512) {
    const int grid = ((g.nslices + 7) / 8) * 8 * g.nmt;
    hipLaunchKernelGGL((blend3_bwd_kernel<2, 2, 1>), dim3(grid), dim3(256), 0, st, dv_posed,
                       reinterpret_cast<const u32x4 *>(pk_bwd), B, N3, g.ktps, g.nslices, g.nmt, part);
  } else {
    const int ngrp = (g.nmt + 1) / 2;          // two mesh tiles per workgroup
    const int grid = ((g.nslices + 7) / 8) * 8 * ngrp;
    hipLaunchKernelGGL((blend3_bwd_kernel<B3_BIG_DEPTH, 1, 2>), dim3(grid), dim3(256), 0, st, dv_posed,
                       reinterpret_cast<const u32x4 *>(pk_bwd), B, N3, g.ktps, g.nslices, ngrp, part);
  }
  SMPLR_LAUNCH_CHECK("blend3_bwd_kernel");
  return 0;
}

}  // namespace smplr

extern "C" {

size_t smplr_blend3_fwd_bytes(int N3) {
  using namespace smplr;
  if (N3 <= 0) return 0;
  return (size_t)((N3 + F3_BN - 1) / F3_BN) * NKT * 9 * 64 * sizeof(uint4);
}

size_t smplr_blend3_bwd_bytes(int N3) {
  if (N3 <= 0) return 0;
  return (size_t)((N3 + 15) / 16) * 7 * 3 * 64 * sizeof(uint4);
}

int smplr_blend3_pack(const float *blend, int N3, void *pk_fwd, void *pk_bwd, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(N3 > 0, "smplr_blend3_pack: bad size N3=%d", N3);
  SMPLR_REQUIRE(blend && (pk_fwd || pk_bwd), "smplr_blend3_pack: null pointer");
  if (pk_fwd) {
    const int ntile = (N3 + F3_BN - 1) / F3_BN;
    const size_t n = (size_t)ntile * NKT * 3 * 64;
    hipLaunchKernelGGL(blend3_pack_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                       blend, N3, ntile, reinterpret_cast<uint4 *>(pk_fwd));
    SMPLR_LAUNCH_CHECK("smplr_blend3_pack(fwd)");
  }
  if (pk_bwd) {
    const int nkt = (N3 + 15) / 16;
    const size_t n = (size_t)nkt * 7 * 64;
    hipLaunchKernelGGL(blend3_pack_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                       blend, N3, nkt, reinterpret_cast<uint4 *>(pk_bwd));
    SMPLR_LAUNCH_CHECK("smplr_blend3_pack(bwd)");
  }
  return 0;
}

int smplr_coef3_pack(const float *coef, int B, void *coef3, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0, "smplr_coef3_pack: bad size B=%d", B);
  if (B == 0) return 0;
  SMPLR_REQUIRE(coef && coef3, "smplr_coef3_pack: null pointer");
  const int n = (B + 31) / 32 * NKT * 64;
  hipLaunchKernelGGL(coef3_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), coef, B,
                     smplr_coef_ld(B), reinterpret_cast<u32x4 *>(coef3));
  SMPLR_LAUNCH_CHECK("smplr_coef3_pack");
  return 0;
}

int smplr_blend3_fwd(const void *coef3, const void *pk_fwd, const float *v_template, int B, int N3,
                     float *v_posed, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && N3 > 0, "smplr_blend3_fwd: bad sizes B=%d N3=%d", B, N3);
  if (B == 0) return 0;
  SMPLR_REQUIRE(coef3 && pk_fwd && v_template && v_posed, "smplr_blend3_fwd: null pointer");
  const int ntiles = (N3 + F3_BN - 1) / F3_BN, ngroups = (B + 127) / 128;
  hipLaunchKernelGGL(blend3_fwd_kernel, dim3(ntiles, ngroups), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const u32x4 *>(coef3), reinterpret_cast<const u32x4 *>(pk_fwd), v_template, B,
                     N3, v_posed);
  SMPLR_LAUNCH_CHECK("smplr_blend3_fwd");
  return 0;
}

int smplr_pose_blend3_fwd(const float *x, int x_stride, int num_cam, int B, const float *J_template,
                          const float *J_dirs, const int32_t *parents, const void *pk_fwd, const float *v_template,
                          int N3, float *Rs, float *J, float *A, float *J_transformed, float *v_posed, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && N3 > 0 && num_cam >= 0 && num_cam <= 16 && x_stride >= num_cam + 82,
                "smplr_pose_blend3_fwd: bad sizes B=%d N3=%d num_cam=%d x_stride=%d", B, N3, num_cam, x_stride);
  if (B == 0) return 0;
  SMPLR_REQUIRE(x && J_template && J_dirs && parents && pk_fwd && v_template && Rs && J && A && J_transformed && v_posed,
                "smplr_pose_blend3_fwd: null pointer");
  const int ntiles = (N3 + F3_BN - 1) / F3_BN, ngroups = (B + 127) / 128, npose = (B + PB_MPB - 1) / PB_MPB;
  const size_t lds_gemm = (size_t)4 * 32 * FC_LD * sizeof(float);
  const size_t lds_pose = (PB_MPB * sizeof(PoseLds) + 15) / 16 * 16 + (size_t)PB_MPB * 224 * sizeof(float);
  const size_t lds = lds_gemm > lds_pose ? lds_gemm : lds_pose;
  static LdsAttrMemo memo = {};                       // once per (kernel, device): common.h
  {
    int rc = ensure_lds_attr(reinterpret_cast<const void *>(pose_blend3_fwd_kernel), lds, &memo, "pose_blend3_fwd_kernel");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(pose_blend3_fwd_kernel, dim3(npose + ntiles * ngroups), dim3(512), lds, as_stream(stream), x, x_stride,
                     num_cam, B, J_template, J_dirs, parents, reinterpret_cast<const u32x4 *>(pk_fwd), v_template, N3,
                     ntiles, npose, Rs, J, A, J_transformed, v_posed);
  SMPLR_LAUNCH_CHECK("smplr_pose_blend3_fwd");
  return 0;
}

size_t smplr_blend3_bwd_workspace(int B, int N3) {
  if (B <= 0 || N3 <= 0) return 0;
  return smplr::blend3_bwd_geom(B, N3).part_floats * sizeof(float);
}

int smplr_blend3_bwd(const float *dv_posed, const void *pk_bwd, int B, int N3, float *dcoef,
                     void *workspace, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && N3 >= 16, "smplr_blend3_bwd: bad sizes B=%d N3=%d (N3 >= 16)", B, N3);
  if (B == 0) return 0;
  SMPLR_REQUIRE(dv_posed && pk_bwd && dcoef && workspace, "smplr_blend3_bwd: null pointer");
  int rc = launch_blend3_bwd_partials(dv_posed, pk_bwd, B, N3, reinterpret_cast<float *>(workspace),
                                      as_stream(stream));
  if (rc) return rc;
  const Blend3BwdGeom g = blend3_bwd_geom(B, N3);
  return launch_blend_bwd_reduce(reinterpret_cast<const float *>(workspace), B, g.nslices, g.nmt, dcoef,
                                 as_stream(stream));
}

}  // extern "C"
