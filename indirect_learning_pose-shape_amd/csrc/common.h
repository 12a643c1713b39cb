// Shared helpers for libsmplraster_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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

// Internal pieces of the fused SMPL backward (smplr_smpl_bwd, pose.hip): the partial-sum producers
// without their stand-alone reduction kernels.
struct BlendBwdGeom { int nslices, cols_per_block, nmt; size_t part_floats; };
BlendBwdGeom blend_bwd_geom(int B, int N3);
int launch_blend_bwd_partials(const float *dv_posed, const float *blend_t, int B, int N3, float *part,
                              hipStream_t st);
int skin_bwd_nblk(int V);
int launch_skin_bwd_partials(const float *dverts, const float *dproj, const float *v_posed,
                             const float *lbs_weights, const float *lbs_top4, const float *A, const float *cam,
                             int x_stride, int B, int V, int vs, float *dv_posed, float *part, hipStream_t st);

// Depth as an unsigned key whose order is the float order (visibility z-buffer, compute_mask.py:98-103).
__device__ __forceinline__ unsigned int orderable(float z) {
  z += 0.0f;  // -0 -> +0 so that equal depths compare equal (tf.argmax treats them as ties)
  const unsigned int b = __float_as_uint(z);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace smplr
