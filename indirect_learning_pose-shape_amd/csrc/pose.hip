// K1 pose kernels: Rodrigues + pose feature + joints-from-betas + 24-joint kinematic chain,
// forward and backward.  Forward: one 64-lane wavefront per mesh, 4 meshes per workgroup; the
// chain's 23 dependent 3x4 products run out of LDS with lanes = matrix elements, parent(i) by
// v_readlane.  Backward: a 512-thread workgroup per mesh sums the producers' partials (one sum
// per thread, all its loads in flight), then one wavefront walks the chain backwards.
//
// Reference: keras_smpl/batch_smpl.py:255-276 (batch_rodrigues), :230-253 (batch_skew),
// :122 (pose_feature), :106-115 (J from v_shaped; here J = J_template + J_dirs*beta, which is
// the same linear map evaluated in the other association order), :168-228 (global rigid).
#include <stdarg.h>
#include "common.h"
#include "pose_device.h"

namespace smplr {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

__global__ __launch_bounds__(MPB * 64) void pose_fwd_kernel(
    const float *__restrict__ x, int x_stride, int num_cam, int B,
    const float *__restrict__ J_template, const float *__restrict__ J_dirs,
    const int *__restrict__ parents, float *__restrict__ coef, int ldc, u32x4 *__restrict__ coef3,
    float *__restrict__ Rs_out, float *__restrict__ J_out, float *__restrict__ A_out,
    float *__restrict__ newJ_out) {
  __shared__ PoseLds lds[MPB];
  __shared__ float sCoef[MPB][224];             // this mesh's coefficient column (for the bf16x3 fragments)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  pose_fwd_wave(x, x_stride, num_cam, B, blockIdx.x * MPB + wave, lane, J_template, J_dirs, parents, coef, ldc, coef3,
                Rs_out, J_out, A_out, newJ_out, lds[wave], sCoef[wave]);
}

constexpr int PBW = 512;   // pose backward: threads per block (one mesh): one partial sum per thread

// The whole backward of one mesh (pose_device.h: pose_bwd_body) - what smplr_pose_bwd and the fp32-GEMM path run.
__global__ __launch_bounds__(PBW) void pose_bwd_kernel(PoseBwdArgs a) {
  __shared__ PoseLds lds1;
  __shared__ float sJd[720];           // J_dirs (the d beta loop walks all of it)
  pose_bwd_body<PBW, POSE_BWD_FULL>(a, blockIdx.x, lds1, sJd);
}

// The TAIL of the backward when its chain ran as a role of the blend GEMM's launch (blend3.hip:
// blend3_bwd_chain_kernel): sums the GEMM's split-K partials, adds them to the chain's dR / d beta terms and does
// Rodrigues' backward.  Bit for bit what pose_bwd_kernel computes.
constexpr int PBT = 256;
__global__ __launch_bounds__(PBT) void pose_bwd_tail_kernel(PoseBwdArgs a) {
  __shared__ PoseLds lds1;
  __shared__ float sJd[1];
  pose_bwd_body<PBT, POSE_BWD_TAIL>(a, blockIdx.x, lds1, sJd);
}

}  // namespace smplr

extern "C" {

int smplr_abi_version(void) { return SMPLR_ABI_VERSION; }
const char *smplr_last_error(void) { return smplr::g_err; }

int smplr_coef_ld(int B) { return B > 0 ? (B + 31) / 32 * 32 : 0; }
size_t smplr_coef3_bytes(int B) { return B > 0 ? (size_t)((B + 31) / 32) * 14 * 3 * 64 * 16 : 0; }

int smplr_pose_fwd(const float *x, int x_stride, int num_cam, int B, const float *J_template,
                   const float *J_dirs, const int32_t *parents, float *coef, void *coef3, float *Rs, float *J,
                   float *A, float *J_transformed, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && num_cam >= 0 && num_cam <= 16 && x_stride >= num_cam + 82,
                "smplr_pose_fwd: bad sizes B=%d num_cam=%d x_stride=%d", B, num_cam, x_stride);
  if (B == 0) return 0;
  SMPLR_REQUIRE(x && J_template && J_dirs && parents && (coef || coef3) && Rs && J && A && J_transformed,
                "smplr_pose_fwd: null pointer");
  hipLaunchKernelGGL(pose_fwd_kernel, dim3((B + MPB - 1) / MPB), dim3(MPB * 64), 0, as_stream(stream),
                     x, x_stride, num_cam, B, J_template, J_dirs, parents, coef, smplr_coef_ld(B),
                     reinterpret_cast<u32x4 *>(coef3), Rs, J, A, J_transformed);
  SMPLR_LAUNCH_CHECK("smplr_pose_fwd");
  return 0;
}

int smplr_pose_bwd(const float *x, int x_stride, int num_cam, int B, const float *J_dirs,
                   const int32_t *parents, const float *Rs, const float *J, const float *A,
                   const float *dcoef, const float *dA, const float *dJ_transformed, const float *dcam,
                   float *dx, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && num_cam >= 0 && num_cam <= 16 && x_stride >= num_cam + 82,
                "smplr_pose_bwd: bad sizes B=%d num_cam=%d x_stride=%d", B, num_cam, x_stride);
  if (B == 0) return 0;
  SMPLR_REQUIRE(x && J_dirs && parents && Rs && J && A && dcoef && dA && dx, "smplr_pose_bwd: null pointer");
  const PoseBwdArgs a{x, x_stride, num_cam, B, J_dirs, parents, Rs, J, A, dcoef, dA, dJ_transformed, dcam, dx,
                      nullptr, 0, nullptr, 0, 0, 0, nullptr};
  hipLaunchKernelGGL(pose_bwd_kernel, dim3(B), dim3(PBW), 0, as_stream(stream), a);
  SMPLR_LAUNCH_CHECK("smplr_pose_bwd");
  return 0;
}

static size_t align256(size_t b) { return (b + 255) / 256 * 256; }

size_t smplr_smpl_bwd_workspace(int B, int V) {
  using namespace smplr;
  if (B <= 0 || V <= 0) return 0;
  const size_t pf = blend_bwd_geom(B, 3 * V).part_floats, pf3 = blend3_bwd_geom(B, 3 * V).part_floats;
  return align256((size_t)B * V * 3 * sizeof(float)) + align256((size_t)B * skin_bwd_nblk(V) * 292 * sizeof(float)) +
         align256((pf > pf3 ? pf : pf3) * sizeof(float)) + align256((size_t)B * POSE_MID * sizeof(float));
}

int smplr_smpl_bwd(const float *dverts, const float *dproj, const float *seg_part, const int16_t *seg_vslot,
                   int seg_nsplit, const float *dJ_transformed, const float *x,
                   int x_stride, int num_cam, int B, int V, int vertex_sampling, const float *blend_t,
                   const void *blend3_bwd, const float *lbs_weights, const float *lbs_top4, const float *J_dirs, const int32_t *parents, const float *Rs,
                   const float *J, const float *A, const float *v_posed, float *dx, void *workspace,
                   void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && V > 0 && num_cam >= 0 && num_cam <= 16 && x_stride >= num_cam + 82 &&
                    vertex_sampling >= 1 && vertex_sampling <= V,
                "smplr_smpl_bwd: bad sizes B=%d V=%d num_cam=%d x_stride=%d vs=%d", B, V, num_cam, x_stride,
                vertex_sampling);
  if (B == 0) return 0;
  SMPLR_REQUIRE(!blend3_bwd || V >= 6, "smplr_smpl_bwd: the bf16x3 GEMM needs 3V >= 16");
  SMPLR_REQUIRE(x && (blend_t || blend3_bwd) && lbs_weights && J_dirs && parents && Rs && J && A && v_posed && dx && workspace,
                "smplr_smpl_bwd: null pointer");
  SMPLR_REQUIRE((seg_part != nullptr) == (seg_vslot != nullptr) && (!seg_part || seg_nsplit > 0),
                "smplr_smpl_bwd: seg_part, seg_vslot and seg_nsplit go together");
  const bool has_proj = dproj || seg_part;
  SMPLR_REQUIRE(dverts || has_proj, "smplr_smpl_bwd: need dverts and/or dproj / the segmentation slot sums");
  SMPLR_REQUIRE(!has_proj || (num_cam >= 4), "smplr_smpl_bwd: dproj needs the 4 camera columns");
  hipStream_t st = as_stream(stream);
  char *base = reinterpret_cast<char *>(workspace);
  float *dv_posed = reinterpret_cast<float *>(base);
  float *skin_part = reinterpret_cast<float *>(base + align256((size_t)B * V * 3 * sizeof(float)));
  float *blend_part = reinterpret_cast<float *>(reinterpret_cast<char *>(skin_part) +
                                                align256((size_t)B * skin_bwd_nblk(V) * 292 * sizeof(float)));
  int rc = launch_skin_bwd_partials(dverts, dproj, SegGrad{seg_part, seg_vslot, seg_nsplit}, v_posed, lbs_weights,
                                    lbs_top4, A, has_proj ? x : nullptr, x_stride, B, V, vertex_sampling, dv_posed,
                                    skin_part, st);
  if (rc) return rc;
  int nslices, nmt;
  if (blend3_bwd) {                              // bf16x3 operands (blend3.hip); else the fp32 matrix-core GEMM
    // The GEMM's launch also runs the part of the pose backward that needs only the skinning partials (sums, the
    // chain in closed form) as a second workgroup role; what is left for the last launch is the tail
    const Blend3BwdGeom g3 = blend3_bwd_geom(B, 3 * V);
    const size_t pf = blend_bwd_geom(B, 3 * V).part_floats;
    float *mid = reinterpret_cast<float *>(reinterpret_cast<char *>(blend_part) +
                                           align256((pf > g3.part_floats ? pf : g3.part_floats) * sizeof(float)));
    const PoseBwdArgs a{x, x_stride, num_cam, B, J_dirs, parents, Rs, J, A, nullptr, nullptr, dJ_transformed, nullptr,
                        dx, skin_part, skin_bwd_nblk(V), blend_part, g3.nslices, g3.nmt, has_proj ? 1 : 0, mid};
    rc = launch_blend3_bwd_chain(dv_posed, blend3_bwd, B, 3 * V, blend_part, a, st);
    if (rc) return rc;
    hipLaunchKernelGGL(pose_bwd_tail_kernel, dim3(B), dim3(PBT), 0, st, a);
    SMPLR_LAUNCH_CHECK("smplr_smpl_bwd(pose tail)");
    return 0;
  } else {
    rc = launch_blend_bwd_partials(dv_posed, blend_t, B, 3 * V, blend_part, st);
    const BlendBwdGeom g = blend_bwd_geom(B, 3 * V);
    nslices = g.nslices;
    nmt = g.nmt;
  }
  if (rc) return rc;
  const PoseBwdArgs a{x, x_stride, num_cam, B, J_dirs, parents, Rs, J, A, nullptr, nullptr, dJ_transformed, nullptr,
                      dx, skin_part, skin_bwd_nblk(V), blend_part, nslices, nmt, has_proj ? 1 : 0, nullptr};
  hipLaunchKernelGGL(pose_bwd_kernel, dim3(B), dim3(PBW), 0, st, a);
  SMPLR_LAUNCH_CHECK("smplr_smpl_bwd(pose)");
  return 0;
}

}  // extern "C"
