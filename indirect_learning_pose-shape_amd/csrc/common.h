// Shared helpers for libsmplraster_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/smplraster.h"

namespace smplr {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Argument check -> negative code + message; launch check -> positive hipError_t.
#define SMPLR_REQUIRE(cond, ...)                       \
  do {                                                 \
    if (!(cond)) {                                     \
      ::smplr::set_error(__VA_ARGS__);                 \
      return SMPLR_EINVAL;                             \
    }                                                  \
  } while (0)

#define SMPLR_LAUNCH_CHECK(name)                                              \
  do {                                                                        \
    hipError_t e__ = hipGetLastError();                                       \
    if (e__ != hipSuccess) {                                                  \
      ::smplr::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return (int)e__;                                                        \
    }                                                                         \
  } while (0)

#define SMPLR_HIP(call)                                                       \
  do {                                                                        \
    hipError_t e__ = (call);                                                  \
    if (e__ != hipSuccess) {                                                  \
      ::smplr::set_error("%s failed: %s", #call, hipGetErrorString(e__));     \
      return (int)e__;                                                        \
    }                                                                         \
  } while (0)

constexpr int WAVE = 64;

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) ONCE per (kernel, device): the attribute belongs to the device's copy of
// a kernel, so a process-wide flag leaves a second device's launch without its LDS (found in round 3).  A call site keeps
// one LdsAttrMemo per kernel; the ordinal is hipGetDevice()'s - or, for the one-GPU test box, the fake one set through
// smplr_debug_device_ordinal() (tests/test_gpu_baseline_sizes.py walks the table with it).  Returns 0 or a hipError_t.
struct LdsAttrMemo { size_t set[64]; };
int ensure_lds_attr(const void *fn, size_t lds, LdsAttrMemo *memo, const char *what);

// Stores of tensors that nothing in the step reads again (verts, proj, mask, the scores): non-temporal, so that they
// do not push the 54 MB of packed blend constants and the step's own intermediates out of the Infinity Cache
// (the step moves 317 MB against its 256 MB): A/B on one box, default step 0.1298 / 0.1289 against 0.1307 / 0.1301 ms,
// both heads 0.167 against 0.172.  -DSMPLR_PLAIN_OUT_STORES restores plain stores (tools/build_variant.sh).
#ifdef SMPLR_PLAIN_OUT_STORES
#define SMPLR_OUT_STORE(ptr, val) (*(ptr) = (val))
#else
#define SMPLR_OUT_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#endif
// (non-temporal LOADS of dseg, the step's one large read-once input, measured no gain: 0.1303 against 0.1298 ms)

// In-kernel timelines (tools/probes/*_timeline.py).  Built with -DSMPLR_TL (make CXXFLAGS+=-DSMPLR_TL) a kernel that
// declares SMPLR_TL_WAVE(buffer, waves-per-workgroup, workgroup) has one lane per wave stamp the shader clock into a
// __device__ buffer (32 words per wave: 0 = entry, 28 = the 100 MHz wall clock, 29 = HW_ID, 30 = XCC_ID) at every
// SMPLR_TL_STAMP(i), and the library exports smplr_tl_read_<kernel>().  Without the flag all of it compiles to nothing.
#ifdef SMPLR_TL
#define SMPLR_TL_WAVE(buf, nwaves, wg, nwg)                                                                       \
  unsigned *tl__ = ((threadIdx.x & 63) == 1 && (wg) < (nwg)) ? (buf) + ((size_t)(wg) * (nwaves) + (threadIdx.x >> 6)) * 32 \
                                                             : nullptr;                                           \
  if (tl__) {                                                                                                     \
    tl__[0] = (unsigned)clock64();                                                                                \
    tl__[28] = (unsigned)wall_clock64();                                                                          \
    tl__[29] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));                                              \
    tl__[30] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));                                             \
  }
#define SMPLR_TL_PTR(buf, nwaves, wg, nwg)                                                                        \
  unsigned *tl__ = ((threadIdx.x & 63) == 1 && (wg) < (nwg)) ? (buf) + ((size_t)(wg) * (nwaves) + (threadIdx.x >> 6)) * 32 \
                                                             : nullptr;
#define SMPLR_TL_STAMP(i)                        \
  do {                                           \
    if (tl__) tl__[i] = (unsigned)clock64();     \
  } while (0)
#define SMPLR_TL_EXPORT(name, buf, words)                                                                \
  extern "C" int smplr_tl_read_##name(unsigned *host, int n) {                                           \
    if (n > (words)) n = (words);                                                                        \
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(buf), (size_t)n * 4, 0, hipMemcpyDeviceToHost);     \
  }
#else
#define SMPLR_TL_WAVE(buf, nwaves, wg, nwg)
#define SMPLR_TL_PTR(buf, nwaves, wg, nwg)
#define SMPLR_TL_STAMP(i) \
  do {                    \
  } while (0)
#define SMPLR_TL_EXPORT(name, buf, words)
#endif

// Internal pieces of the fused SMPL backward (smplr_smpl_bwd, pose.hip): the partial-sum producers
// without their stand-alone reduction kernels.
struct BlendBwdGeom { int nslices, cols_per_block, nmt; size_t part_floats; };
BlendBwdGeom blend_bwd_geom(int B, int N3);
int launch_blend_bwd_partials(const float *dv_posed, const float *blend_t, int B, int N3, float *part,
                              hipStream_t st);
int launch_blend_bwd_reduce(const float *part, int B, int nslices, int nmt, float *dcoef, hipStream_t st);
// the same GEMM on the bf16 matrix cores with 3-way split operands (blend3.hip)
struct Blend3BwdGeom { int nslices, ktps, nmt; size_t part_floats; };
Blend3BwdGeom blend3_bwd_geom(int B, int N3);
int launch_blend3_bwd_partials(const float *dv_posed, const void *pk_bwd, int B, int N3, float *part,
                               hipStream_t st);
int skin_bwd_nblk(int V);
// The segmentation backward's per-row-block slot sums (raster.hip), which the skinning backward can gather
// by vertex instead of reading a merged dproj: part (B, nsplit, SB_NWIN, SB_SLOTS, 2), vslot (B, VP) = the
// record slot of each vertex (-1: none), nsplit = ceil(W / seg_bwd_rows(B, W)).
constexpr int SB_SLOTS = 4096;   // 32 KB of LDS accumulators per window
constexpr int SB_NWIN = 5;       // slot windows the partial buffer holds: S <= 20480
constexpr int SB_ROWS = 8;       // rows (strips) per block at small batch ...
constexpr int SB_ROWS_BIG = 24;  // ... and while that gives every CU one to four blocks: fewer zero / write-out passes
                                 // over the accumulators and fewer partials per vertex for the gather
                                 // (B = 128: seg_bwd -1.0 us, skin_bwd -0.4; at B = 32 it would cost 4 us)
inline int seg_bwd_rows(int B, int W) {
  // seg_bwd + skin_bwd at W = 48, 8 against 24 rows: B = 128 42.4 / 41.0 us, 256: 81.9 / 78.3, 512: 153.3 / 156.9,
  // 2048: 578 / 601 (round 3's kernels) - the tall blocks pay off from one block per CU on
  // (round 4, with the pipelined row walk: tall blocks at every batch from there on - B = 2 048: seg_bwd 260 -> 289 us,
  // skin_bwd 261 -> 233 (two partials to gather per vertex instead of six), step 1.442 -> 1.432 ms; B = 512: 0.3897 -> 0.3871;
  // SMPLR_SEGBWD_TALL_TO=1024 restores round 3's upper end)
  static const long long tall_to = getenv("SMPLR_SEGBWD_TALL_TO") ? atoll(getenv("SMPLR_SEGBWD_TALL_TO")) : (1ll << 40);
  static const int tall_env = getenv("SMPLR_SEGBWD_TALL_ROWS") ? atoi(getenv("SMPLR_SEGBWD_TALL_ROWS")) : 0;   // (A/B runs: 2..24)
  // the tall height divides the image where it can (W = 64: 24 + 24 + 16 rows left the launch waiting for its two
  // full-height blocks per mesh - whole step at B = 128: 0.1698 ms with 24, 0.1691 with 8, 0.1605 with 16 rows)
  int tall = SB_ROWS_BIG;
  for (int r = SB_ROWS_BIG; r >= 12; r -= 2)
    if (W % r == 0) { tall = r; break; }
  if (tall_env >= 2 && tall_env <= SB_ROWS_BIG) tall = tall_env;
  const long long n = (long long)B * ((W + tall - 1) / tall);          // tall blocks of the launch
  return (n >= 256 && n < tall_to) ? tall : SB_ROWS;
}
struct SegGrad { const float *part; const int16_t *vslot; int nsplit; };
int launch_skin_bwd_partials(const float *dverts, const float *dproj, SegGrad sg, const float *v_posed,
                             const float *lbs_weights, const float *lbs_top4, const float *A, const float *cam,
                             int x_stride, int B, int V, int vs, float *dv_posed, float *part, hipStream_t st,
                             int *nblk_out /* partial blocks per mesh it wrote: pose_bwd / the reduce sum that many */);

// Linear-blend skinning of one vertex from its <= 4 (weight, joint) pairs against the mesh's 24 x 12 joint matrix in
// LDS (sAj: 72 float4), then the orthographic projection.  One definition for the two kernels that must
// agree bit for bit: skin_fwd_kernel (skin.hip) and the binning kernel that skins its own vertices (raster.hip).
__device__ __forceinline__ void skin_T_sparse(const float4 *sAj, const float4 ww, const float4 jj, float T[12]) {
  const float w[4] = {ww.x, ww.y, ww.z, ww.w};
  const int jx[4] = {(int)jj.x, (int)jj.y, (int)jj.z, (int)jj.w};
#pragma unroll
  for (int e = 0; e < 12; ++e) T[e] = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#if defined(SMPLR_KO_SKIN_NOLDS)      // knock-outs (tools/build_variant.sh; wrong results on purpose): no LDS read at all ...
    const float4 r0 = make_float4(jj.x, jj.y, jj.z, jj.w), r1 = make_float4(jj.y, jj.x, jj.w, jj.z), r2 = make_float4(jj.w, jj.z, jj.y, jj.x);
#elif defined(SMPLR_KO_SKIN_JOINT0)   // ... / every lane reads joint k's rows (pure broadcast: no bank conflict)
    const float4 r0 = sAj[k * 3], r1 = sAj[k * 3 + 1], r2 = sAj[k * 3 + 2];
#else
    const float4 r0 = sAj[jx[k] * 3], r1 = sAj[jx[k] * 3 + 1], r2 = sAj[jx[k] * 3 + 2];
#endif
    const float wj = w[k];
    T[0] = fmaf(wj, r0.x, T[0]); T[1] = fmaf(wj, r0.y, T[1]); T[2] = fmaf(wj, r0.z, T[2]); T[3] = fmaf(wj, r0.w, T[3]);
    T[4] = fmaf(wj, r1.x, T[4]); T[5] = fmaf(wj, r1.y, T[5]); T[6] = fmaf(wj, r1.z, T[6]); T[7] = fmaf(wj, r1.w, T[7]);
    T[8] = fmaf(wj, r2.x, T[8]); T[9] = fmaf(wj, r2.y, T[9]); T[10] = fmaf(wj, r2.z, T[10]); T[11] = fmaf(wj, r2.w, T[11]);
  }
}
// (Plain expressions on purpose: hipcc contracts them the same way in both kernels - checked in the ISA, and
// test_skinning_inside_the_binning_kernel_equals_separate_calls compares the outputs bit for bit at five sizes - and
// the way it did before the function was shared, so the projected positions, and with them every near-tie of the
// rasterisers against the float64 oracle, are the ones the parity tests were tuned on.)
__device__ __forceinline__ void skin_apply(const float T[12], float p0, float p1, float p2, float &X, float &Y, float &Z) {
  X = T[0] * p0 + T[1] * p1 + T[2] * p2 + T[3];
  Y = T[4] * p0 + T[5] * p1 + T[6] * p2 + T[7];
  Z = T[8] * p0 + T[9] * p1 + T[10] * p2 + T[11];
}
// (u, v) = cam[2:4] + (X, Y) * cam[0:2]  (projection.py:62-79)
__device__ __forceinline__ float project_u(float X, float c0, float c2) { return c2 + X * c0; }

// What the binning kernel needs to skin its own vertices (smplr_skin_vis_seg_fwd); all NULL otherwise.
struct SkinIn { const float *v_posed, *top4, *A, *cam; int x_stride; float *verts, *proj; };

// Depth as an unsigned key whose order is the float order (visibility z-buffer, compute_mask.py:98-103).
__device__ __forceinline__ unsigned int orderable(float z) {
  z += 0.0f;  // -0 -> +0 so that equal depths compare equal (tf.argmax treats them as ties)
  const unsigned int b = __float_as_uint(z);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// ---- bf16x3: an fp32 number as the exact sum of three bf16 numbers (blend3.hip, pose.hip) ----
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Two fp32 values -> three dwords of packed bf16 pairs (value 0 in the low half).
struct Split2 { unsigned h, m, l; };
__device__ __forceinline__ Split2 split2(float x0, float x1) {
  const f32x2v v = {x0, x1};
  const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2v r = {x0 - __uint_as_float(hu << 16), x1 - __uint_as_float(hu & 0xffff0000u)};
  const unsigned mu = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  const f32x2v r2 = {r.x - __uint_as_float(mu << 16), r.y - __uint_as_float(mu & 0xffff0000u)};
  const unsigned lu = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
  return {hu, mu, lu};
}

struct Frag3 { bf16x8 h, m, l; };
__device__ __forceinline__ Frag3 split8(const float x[8]) {
  u32x4 h, m, l;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const Split2 s = split2(x[2 * d], x[2 * d + 1]);
    h[d] = s.h;
    m[d] = s.m;
    l[d] = s.l;
  }
  Frag3 f;
  f.h = __builtin_bit_cast(bf16x8, h);
  f.m = __builtin_bit_cast(bf16x8, m);
  f.l = __builtin_bit_cast(bf16x8, l);
  return f;
}

// ---- focal loss pieces shared by the loss head (loss.hip) and the rasteriser's loss epilogue (raster.hip) ----
constexpr float K_EPS = 1e-7f;   // keras.backend.epsilon()  (focal_loss.py:17)
__device__ __forceinline__ float pow_gamma(float x, float gamma) {
  // (1-p)^gamma; the reference's only values are 2 (focal) and, for cross-entropy, 0
  if (gamma == 2.0f) return x * x;
  if (gamma == 0.0f) return 1.0f;
  if (gamma == 1.0f) return x;
  return powf(x, gamma);
}
__device__ __forceinline__ float dpow_gamma(float x, float gamma) {   // d/dx x^gamma
  if (gamma == 2.0f) return 2.0f * x;
  if (gamma == 0.0f) return 0.0f;
  if (gamma == 1.0f) return 1.0f;
  return gamma * powf(x, gamma - 1.0f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace smplr
