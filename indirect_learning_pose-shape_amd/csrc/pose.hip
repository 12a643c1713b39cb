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

static int g_fake_ordinal = -1;      // >= 0: what ensure_lds_attr takes for the current device (tests)
static int g_lds_attr_sets = 0;      // hipFuncSetAttribute calls made through ensure_lds_attr

int ensure_lds_attr(const void *fn, size_t lds, LdsAttrMemo *memo, const char *what) {
  int dev = 0;
  if (g_fake_ordinal >= 0) {
    dev = g_fake_ordinal;
  } else {
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
      set_error("%s: hipGetDevice failed: %s", what, hipGetErrorString(e));
      return (int)e;
    }
  }
  const bool known = dev >= 0 && dev < 64;
  if (known && memo->set[dev] >= lds) return 0;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    set_error("%s: hipFuncSetAttribute(%zu B of LDS) failed: %s", what, lds, hipGetErrorString(e));
    return (int)e;
  }
  ++g_lds_attr_sets;
  if (known) memo->set[dev] = lds;
  return 0;
}

// Given dR (gradient wrt the 9 entries of R) return dtheta.
__device__ __forceinline__ void rodrigues_bwd(const float t[3], const float dR[9], float dt[3]) {
  const float e[3] = {t[0] + 1e-8f, t[1] + 1e-8f, t[2] + 1e-8f};
  const float angle = sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
  const float inv = 1.0f / angle;
  const float r[3] = {t[0] * inv, t[1] * inv, t[2] * inv};
  float s, c;
  sincosf(angle, &s, &c);
  const float oc = 1.0f - c;
  // d/d angle: -s*I + s*r r^T + c*K
  const float tr = dR[0] + dR[4] + dR[8];
  float rDr = 0.f;  // sum_ij dR_ij r_i r_j
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) rDr += dR[i * 3 + j] * r[i] * r[j];
  // sum_ij dR_ij K_ij with K = skew(r)
  const float w0 = dR[7] - dR[5], w1 = dR[2] - dR[6], w2 = dR[3] - dR[1];
  const float dK = w0 * r[0] + w1 * r[1] + w2 * r[2];
  float da = -s * tr + s * rDr + c * dK;
  // d/d r_k: (1-c) * ((dR r)_k + (dR^T r)_k) + s * w_k
  float dr[3];
  const float w[3] = {w0, w1, w2};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) a += dR[k * 3 + j] * r[j] + dR[j * 3 + k] * r[j];
    dr[k] = oc * a + s * w[k];
  }
  // r = t/angle  ->  dt += dr/angle ; dangle -= dr.t / angle^2
  da -= (dr[0] * t[0] + dr[1] * t[1] + dr[2] * t[2]) * inv * inv;
#pragma unroll
  for (int k = 0; k < 3; ++k) dt[k] = dr[k] * inv + da * e[k] * inv;
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

__global__ __launch_bounds__(PBW) void pose_bwd_kernel(
    const float *__restrict__ x, int x_stride, int num_cam, int B,
    const float *__restrict__ J_dirs, const int *__restrict__ parents,
    const float *__restrict__ Rs_in, const float *__restrict__ J_in, const float *__restrict__ A_in,
    const float *__restrict__ dcoef, const float *__restrict__ dA, const float *__restrict__ dnewJ,
    const float *__restrict__ dcam, float *__restrict__ dx,
    // fused mode (dA == nullptr): sum the skinning partials (B,nblk,292) and the split-K partials
    // (ns,nmt,32,224) of the blend GEMM here, in fixed order, instead of in two more launches
    const float *__restrict__ skin_part, int nblk, const float *__restrict__ blend_part, int ns, int nmt,
    int want_dcam) {
  // One mesh per 256-thread block: all four waves stage the inputs and sum the producers' partials
  // (512 entries, two per thread, their 27 / 54 loads all in flight together: ~3 memory round trips);
  // then wave 0 alone walks the chain.
  __shared__ PoseLds lds1;
  __shared__ float sJd[720];           // J_dirs (the d beta loop walks all of it)
  PoseLds &L = lds1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int n = blockIdx.x;
  const bool live = true;
  const size_t nn = n;
  // this lane's joint angles, needed only by the Rodrigues backward at the very end: requested with everything
  // else (a load there sat alone on the critical path)
  float th[3] = {0.f, 0.f, 0.f};
  int tdesc = 0, tchild = 0;                       // the SMPL tree's masks for joint `tid`, requested with the rest
  if (tid < 24) {
    const float *xr0 = x + nn * x_stride + num_cam + 3 * tid;
    th[0] = xr0[0]; th[1] = xr0[1]; th[2] = xr0[2];
    tdesc = SMPL_TREE_DESC[tid];
    tchild = SMPL_TREE_CHILD[tid];
  }
  for (int e = tid; e < 720; e += PBW) sJd[e] = J_dirs[e];
  for (int e = tid; e < 216; e += PBW) L.Rs[e / 9][e % 9] = Rs_in[nn * 216 + e];
  for (int e = tid; e < 72; e += PBW) L.J[e / 3][e % 3] = J_in[nn * 72 + e];
  for (int e = tid; e < 288; e += PBW) L.G[e / 12][e % 12] = A_in[nn * 288 + e];  // G.R = A.R
  if (dA) {
    for (int e = tid; e < 288; e += PBW) L.dA[e / 12][e % 12] = dA[nn * 288 + e];
    for (int e = tid; e < 220; e += PBW) L.dcoef[e] = dcoef[nn * SMPLR_KPAD + e];
    if (tid < 4) L.dcam[tid] = dcam ? dcam[nn * 4 + tid] : 0.0f;
  } else {
    // 512 sums, one per thread, each with up to PB_INFLIGHT of its partials requested at once
    // (clamped addresses, surplus terms replaced by 0): one round trip for the whole reduction.
    // The order of the additions is the producers' slice / block order, as in the stand-alone
    // reduce kernels.
    constexpr int PB_INFLIGHT = 60;
    for (int e = tid; e < 512; e += PBW) {
      const bool skin = e < 292;
      const int cnt = skin ? nblk : ns;
      const size_t mt = nn >> 5, r = nn & 31;
      const float *p = skin ? skin_part + (nn * nblk) * 292 + e
                            : blend_part + (mt * 32 + r) * 224 + min(e - 292, 219);
      const size_t stride = skin ? (size_t)292 : (size_t)nmt * 32 * 224;
      float acc = 0.0f;
      for (int s0 = 0; s0 < cnt; s0 += PB_INFLIGHT) {
        float v[PB_INFLIGHT];
#pragma unroll
        for (int u = 0; u < PB_INFLIGHT; ++u) v[u] = p[(size_t)min(s0 + u, cnt - 1) * stride];
#pragma unroll
        for (int u = 0; u < PB_INFLIGHT; ++u) acc += (s0 + u < cnt) ? v[u] : 0.0f;
      }
      if (e < 288) L.dA[e / 12][e % 12] = acc;
      else if (e < 292) L.dcam[e - 288] = want_dcam ? acc : 0.0f;
      else if (e - 292 < 220) L.dcoef[e - 292] = acc;
    }
  }
  __syncthreads();
  const int par = parents[lane < 24 ? lane : 0];
  if (tid < 24) {
    const int i = tid;
    const float *dAi = &L.dA[i][0];
    float dAt[3] = {dAi[3], dAi[7], dAi[11]};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) L.dGR[i][r * 3 + c] = dAi[r * 4 + c] - dAt[r] * L.J[i][c];
      L.dGt[i][r] = dAt[r] + (dnewJ ? dnewJ[nn * 72 + i * 3 + r] : 0.0f);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
      L.dJ[i][c] = -(L.G[i][0 * 4 + c] * dAt[0] + L.G[i][1 * 4 + c] * dAt[1] + L.G[i][2 * 4 + c] * dAt[2]);
    // t_i = A_i.t + G_i.R J_i, the world position of joint i (dA_i has just been consumed, by this thread)
#pragma unroll
    for (int r = 0; r < 3; ++r)
      L.dA[i][9 + r] = L.G[i][r * 4 + 3] + (L.G[i][r * 4 + 0] * L.J[i][0] + L.G[i][r * 4 + 1] * L.J[i][1] +
                                            L.G[i][r * 4 + 2] * L.J[i][2]);
  }
  if (tid < 64) {
    // masks of the tree: the standard SMPL tree's from the tables, any other tree's by walking up (lanes = joints)
    const bool smpl_tree = is_smpl_tree(par, lane);                     // wave-uniform
    int desc = tdesc, child = tchild;
    if (!smpl_tree) {
      int anc = 1 << (lane & 31), p = lane < 24 ? par : -1;
      for (int it = 0; it < 23; ++it) {                               // uniform trip count: shuffles need every lane
        anc |= (p >= 0) ? 1 << p : 0;
        const int pp = __shfl(par, p >= 0 ? p : 0, 64);
        p = (p >= 0) ? pp : -1;
      }
      desc = 0;
      child = 0;
      for (int d = 0; d < 24; ++d) {
        desc |= ((__shfl(anc, d, 64) >> (lane & 31)) & 1) << d;
        child |= (__shfl(par, d, 64) == lane ? 1 : 0) << d;
      }
    }
    if (lane < 24) {
      L.desc[lane] = desc;
      L.child[lane] = child;
      L.par[lane] = par;
    }
  }
  __syncthreads();
  // The chain, in closed form.  The recursion (children before parents)
  //     dGR_p += dGR_i R_i^T + dGt_i (x) (J_i - J_p),   dGt_p += dGt_i,   dR_i = G_p.R^T dGR_i
  // is 23 dependent LDS round trips on ONE wave (6.8 us of this kernel's 13.4).  Written in the world frame -
  // H_i = dGR_i G_i.R^T, with G_c.R R_c^T = G_p.R and J_c - J_p = G_p.R^T (t_c - t_p), t = world joint positions - it
  // telescopes into sums over subtrees:
  //     dGt_i = sum_{d in sub(i)} dGt_d^0,        H_i = sum_{d in sub(i)} Z_d - Y_i,
  //     Z_d = dGR_d^0 G_d.R^T + Y_d,   Y_d = dGt_d (x) (t_d - t_parent(d))  (Y_root = 0),
  //     dR_i = G_p.R^T H_i G_i.R   (root: H_0 G_0.R),
  //     dJ_i = dJ_i^0 + G_p.R^T dGt_i - sum_{c child of i} G_i.R^T dGt_c   (root: + dGt_0),
  // i.e. five phases whose items (joint x matrix element, one per thread of the workgroup) are independent, the
  // subtree sums taken in index order from each joint's descendant mask: the same gradient, a fixed summation
  // order, all eight waves instead of one.
  // scratch: L.dA[i][0..8] = X_i, then T_i = H_i G_i.R;  L.dA[i][9..11] = t_i;  L.tmpv = bone vectors, then G_p.R^T dGt;  L.dGR = H
  if (tid < 72) {                                          // P1: dGt over subtrees
    const int i = tid / 3, r = tid - 3 * i, m = L.desc[i];
    // (every joint is read, the mask picks the terms: unconditional reads are all in flight together, a test around
    // each read serialises 24 LDS round trips)
    float v[24];
#pragma unroll
    for (int d = 0; d < 24; ++d) v[d] = L.dGt[d][r];
    float acc = 0.0f;
#pragma unroll
    for (int d = 0; d < 24; ++d) acc += ((m >> d) & 1) ? v[d] : 0.0f;
    L.dGtF[i][r] = acc;
  }
  __syncthreads();
  const int ci = tid / 9, ce = tid - 9 * ci, cr = ce / 3, cc = ce - 3 * cr;   // this thread's (joint, row, column) for tid < 216
  if (tid < 216) {                                         // P2: X_d = dGR_d^0 G_d.R^T
    L.dA[ci][ce] = L.dGR[ci][cr * 3 + 0] * L.G[ci][cc * 4 + 0] + L.dGR[ci][cr * 3 + 1] * L.G[ci][cc * 4 + 1] +
                   L.dGR[ci][cr * 3 + 2] * L.G[ci][cc * 4 + 2];
  } else if (tid >= 256 && tid < 328) {                    //     and the bone vectors t_d - t_parent(d) (root: 0)
    const int it = tid - 256, d = it / 3, c = it - 3 * d, p = L.par[d];
    L.tmpv[d][c] = (p >= 0) ? L.dA[d][9 + c] - L.dA[p][9 + c] : 0.0f;
  }
  __syncthreads();
  if (tid < 216) {                                         // P3: H_i = sum_{sub(i)} X_d + sum_{sub(i), d != i} Y_d
    // (Y_i itself is left out of the sum rather than added and subtracted again: it is of the size of the result)
    const int m = L.desc[ci], my = m & ~(1 << ci);
    float vx[24], vg[24], vt[24];
#pragma unroll
    for (int d = 0; d < 24; ++d) { vx[d] = L.dA[d][ce]; vg[d] = L.dGtF[d][cr]; vt[d] = L.tmpv[d][cc]; }
    float acc = 0.0f;
#pragma unroll
    for (int d = 0; d < 24; ++d) {
      acc += ((m >> d) & 1) ? vx[d] : 0.0f;
      acc += ((my >> d) & 1) ? vg[d] * vt[d] : 0.0f;
    }
    L.dGR[ci][ce] = acc;
  }
  __syncthreads();
  if (tid < 216) {                                         // P4: T_i = H_i G_i.R
    L.dA[ci][ce] = L.dGR[ci][cr * 3 + 0] * L.G[ci][0 * 4 + cc] + L.dGR[ci][cr * 3 + 1] * L.G[ci][1 * 4 + cc] +
                   L.dGR[ci][cr * 3 + 2] * L.G[ci][2 * 4 + cc];
  } else if (tid >= 256 && tid < 328) {                    //     and G_p.R^T dGt_i
    const int it = tid - 256, i = it / 3, c = it - 3 * i, p = L.par[i];
    L.tmpv[i][c] = (p >= 0) ? L.G[p][0 * 4 + c] * L.dGtF[i][0] + L.G[p][1 * 4 + c] * L.dGtF[i][1] +
                                  L.G[p][2 * 4 + c] * L.dGtF[i][2]
                            : L.dGtF[i][c];
  }
  __syncthreads();
  if (tid < 216) {                                         // P5: dR_i = G_p.R^T T_i
    const int p = L.par[ci];
    L.dR[ci][ce] = (p >= 0) ? L.G[p][0 * 4 + cr] * L.dA[ci][0 * 3 + cc] + L.G[p][1 * 4 + cr] * L.dA[ci][1 * 3 + cc] +
                                  L.G[p][2 * 4 + cr] * L.dA[ci][2 * 3 + cc]
                            : L.dA[ci][ce];
  } else if (tid >= 256 && tid < 328) {                    //     and dJ_i
    const int it = tid - 256, i = it / 3, c = it - 3 * i, m = L.child[i];
    float v[24];
#pragma unroll
    for (int d = 0; d < 24; ++d) v[d] = L.tmpv[d][c];
    float acc = L.dJ[i][c] + L.tmpv[i][c];
#pragma unroll
    for (int d = 0; d < 24; ++d) acc -= ((m >> d) & 1) ? v[d] : 0.0f;
    L.dJ[i][c] = acc;
  }
  __syncthreads();
  if (tid >= 64) return;                                 // the rest is one wavefront's work
  // d beta = dcoef[0..9] + J_dirs^T dJ: 10 x 72 products, over 60 lanes (6 chunks of 12 per beta; the 10 serial
  // 72-term sums took 0.6 us between two divergent branches), chunk sums parked in L.dA (free by now)
  float *scratch = &L.dA[0][0];
  if (live && lane < 60) {
    const int k = lane / 6, part = lane - 6 * k;
    float acc = 0.0f;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      const int e = part * 12 + q;
      acc += L.dJ[e / 3][e % 3] * sJd[e * 10 + k];
    }
    scratch[lane] = acc;
  }
  wave_sync();
  if (live) {
    float *dxr = dx + nn * x_stride;
    const float *dc = L.dcoef;
    if (lane < 24) {
      float g[9];
#pragma unroll
      for (int e = 0; e < 9; ++e) g[e] = L.dR[lane][e] + (lane >= 1 ? dc[10 + 9 * (lane - 1) + e] : 0.0f);
      float dt[3];
      rodrigues_bwd(th, g, dt);
#pragma unroll
      for (int k = 0; k < 3; ++k) dxr[num_cam + 3 * lane + k] = dt[k];
    } else if (lane >= 32 && lane < 42) {
      const int k = lane - 32;
      float acc = dc[k];
#pragma unroll
      for (int part = 0; part < 6; ++part) acc += scratch[k * 6 + part];
      dxr[num_cam + 72 + k] = acc;
    } else if (lane >= 48 && lane < 48 + num_cam) {
      const int cidx = lane - 48;
      dxr[cidx] = (cidx < 4) ? L.dcam[cidx] : 0.0f;
    }
  }
}

}  // namespace smplr

extern "C" {

int smplr_abi_version(void) { return SMPLR_ABI_VERSION; }

int smplr_debug_device_ordinal(int fake) {
  const int prev = smplr::g_fake_ordinal;
  smplr::g_fake_ordinal = fake < 0 ? -1 : (fake < 64 ? fake : 63);
  return prev;
}
int smplr_debug_lds_attr_sets(void) { return smplr::g_lds_attr_sets; }
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
  hipLaunchKernelGGL(pose_bwd_kernel, dim3(B), dim3(PBW), 0, as_stream(stream),
                     x, x_stride, num_cam, B, J_dirs, parents, Rs, J, A, dcoef, dA, dJ_transformed, dcam, dx,
                     (const float *)nullptr, 0, (const float *)nullptr, 0, 0, 0);
  SMPLR_LAUNCH_CHECK("smplr_pose_bwd");
  return 0;
}

static size_t align256(size_t b) { return (b + 255) / 256 * 256; }

size_t smplr_smpl_bwd_workspace(int B, int V) {
  using namespace smplr;
  if (B <= 0 || V <= 0) return 0;
  const size_t pf = blend_bwd_geom(B, 3 * V).part_floats, pf3 = blend3_bwd_geom(B, 3 * V).part_floats;
  return align256((size_t)B * V * 3 * sizeof(float)) + align256((size_t)B * skin_bwd_nblk(V) * 292 * sizeof(float)) +
         align256((pf > pf3 ? pf : pf3) * sizeof(float));
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
  int skin_nblk = 0;
  int rc = launch_skin_bwd_partials(dverts, dproj, SegGrad{seg_part, seg_vslot, seg_nsplit}, v_posed, lbs_weights,
                                    lbs_top4, A, has_proj ? x : nullptr, x_stride, B, V, vertex_sampling, dv_posed,
                                    skin_part, st, &skin_nblk);
  if (rc) return rc;
  int nslices, nmt;
  if (blend3_bwd) {                              // bf16x3 operands (blend3.hip); else the fp32 matrix-core GEMM
    rc = launch_blend3_bwd_partials(dv_posed, blend3_bwd, B, 3 * V, blend_part, st);
    const Blend3BwdGeom g3 = blend3_bwd_geom(B, 3 * V);
    nslices = g3.nslices;
    nmt = g3.nmt;
  } else {
    rc = launch_blend_bwd_partials(dv_posed, blend_t, B, 3 * V, blend_part, st);
    const BlendBwdGeom g = blend_bwd_geom(B, 3 * V);
    nslices = g.nslices;
    nmt = g.nmt;
  }
  if (rc) return rc;
  hipLaunchKernelGGL(pose_bwd_kernel, dim3(B), dim3(PBW), 0, st, x, x_stride, num_cam, B,
                     J_dirs, parents, Rs, J, A, (const float *)nullptr, (const float *)nullptr, dJ_transformed,
                     (const float *)nullptr, dx, skin_part, skin_nblk, blend_part, nslices, nmt,
                     has_proj ? 1 : 0);
  SMPLR_LAUNCH_CHECK("smplr_smpl_bwd(pose)");
  return 0;
}

}  // extern "C"
