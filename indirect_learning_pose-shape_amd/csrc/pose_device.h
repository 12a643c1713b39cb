// Device pieces of the pose forward shared by pose.hip (pose_fwd_kernel) and blend3.hip (the fused pose + blend
// kernel): LDS layout, wave-level sync, the SMPL tree by levels, Rodrigues, and the forward of one mesh on one wave.
#pragma once
#include "common.h"

namespace smplr {

constexpr int MPB = 4;  // meshes (waves) per block

// Each wave owns its mesh's LDS region, so ordering LDS traffic inside the wave is enough.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct PoseLds {
  float Rs[24][9];
  float J[24][3];
  float G[24][12];   // rows 0..2 of the 4x4 world transform: [R | t]
  float dGR[24][9];
  float dGt[24][3];
  float dJ[24][3];
  float dR[24][9];
  float dA[24][12];     // backward inputs, summed from the producers' partials when fused
  float dcoef[220];
  float dcam[4];
  float dGtF[24][3];    // backward: subtree sums of dGt
  float tmpv[24][3];    // backward: G_parent.R^T dGtF_i
  int desc[24], child[24], par[24];   // backward: descendants (incl. self) / children of joint i as bit masks, parent
};

// ---- the SMPL kinematic tree by LEVELS ------------------------------------------------------------------
// The 23 joint updates of the chain are serial only along a path of the tree: joints of equal depth are
// independent.  For the standard SMPL tree (8 levels of at most 5 joints) the schedule is a compile-time
// table: in pose_fwd 12 lanes per joint (one per element of its 3 x 4 transform) do a whole level at once, i.e. 8
// dependent LDS round trips instead of 23 (the serial chain measured 4 us of the kernel's 7.5; by levels the
// kernel went from 11.4 to 10.2 us).  A wave checks its `parents` against the table (one compare + ballot) and
// any other tree takes the serial loop, which computes the same values.  The backward chain stays serial: by
// levels it needs a second phase per level in which parents collect their children's contributions in index
// order, 20 phases of ~0.3 us instead of 23 steps - measured 15.3 against 14.7 us for the whole kernel.
__constant__ const signed char SMPL_TREE_PARENT[24] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14,
                                                       16, 17, 18, 19, 20, 21};
// joint i's descendants (itself included) and children as bit masks (the backward's closed-form chain)
__constant__ const int SMPL_TREE_DESC[24] = {0xffffff, 0x492, 0x924, 0xfff248, 0x490, 0x920, 0xfff240, 0x480, 0x900, 0xfff200,
                                             0x400, 0x800, 0x9000, 0x552000, 0xaa4000, 0x8000, 0x550000, 0xaa0000,
                                             0x540000, 0xa80000, 0x500000, 0xa00000, 0x400000, 0x800000};
__constant__ const int SMPL_TREE_CHILD[24] = {0xe, 0x10, 0x20, 0x40, 0x80, 0x100, 0x200, 0x400, 0x800, 0x7000, 0x0, 0x0,
                                              0x8000, 0x10000, 0x20000, 0x0, 0x40000, 0x80000, 0x100000, 0x200000,
                                              0x400000, 0x800000, 0x0, 0x0};
constexpr int TREE_LEVELS = 8;
// level (1-based - 1) -> its joints (-1 = none), at most 5
constexpr int TREE_LVL[TREE_LEVELS][5] = {{1, 2, 3, -1, -1},      {4, 5, 6, -1, -1},   {7, 8, 9, -1, -1},
                                          {10, 11, 12, 13, 14},   {15, 16, 17, -1, -1}, {18, 19, -1, -1, -1},
                                          {20, 21, -1, -1, -1},   {22, 23, -1, -1, -1}};
constexpr int TREE_PAR[24] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21};
// the joint of this lane's slot (lane / 12) in a 5-entry list, as selects on compile-time constants
__device__ __forceinline__ int slot_joint(const int (&js)[5], int slot) {
  int i = js[0];
#pragma unroll
  for (int t = 1; t < 5; ++t) i = (slot == t) ? js[t] : i;
  return slot < 5 ? i : -1;
}
__device__ __forceinline__ int slot_parent(const int (&js)[5], int slot) {
  int p = js[0] >= 0 ? TREE_PAR[js[0]] : 0;
#pragma unroll
  for (int t = 1; t < 5; ++t) p = (slot == t && js[t] >= 0) ? TREE_PAR[js[t]] : p;
  return p;
}
// wave-uniform: do the 24 lanes' parents equal the SMPL tree?
__device__ __forceinline__ bool is_smpl_tree(int par, int lane) {
  const int want = SMPL_TREE_PARENT[lane < 24 ? lane : 0];
  return __ballot(lane < 24 && par != want) == 0ull;
}

// R = cos*I + (1-cos)*r r^T + sin*skew(r),  angle = |theta + 1e-8|, r = theta/angle.
// Two kernels evaluate this for the same joint angles (the pose kernel for Rs, the blend GEMM's waves for their
// coefficient rows) and must get the same bits, so nothing here is left to the compiler's choice of what to
// contract into an fma in which context: contraction is off and every product and sum is written out.
__device__ __forceinline__ void rodrigues(const float t[3], float R[9]) {
#pragma clang fp contract(off)
  const float e0 = t[0] + 1e-8f, e1 = t[1] + 1e-8f, e2 = t[2] + 1e-8f;
  const float angle = sqrtf((e0 * e0 + e1 * e1) + e2 * e2);
  const float inv = 1.0f / angle;                       // one IEEE division; r = theta / angle to 1 ulp
  const float rx = t[0] * inv, ry = t[1] * inv, rz = t[2] * inv;
  float s, c;
  sincosf(angle, &s, &c);
  const float oc = 1.0f - c;
  const float ox = oc * rx, oy = oc * ry, oz = oc * rz;
  const float sx = s * rx, sy = s * ry, sz = s * rz;
  R[0] = c + ox * rx;
  R[1] = ox * ry - sz;
  R[2] = ox * rz + sy;
  R[3] = oy * rx + sz;
  R[4] = c + oy * ry;
  R[5] = oy * rz - sx;
  R[6] = oz * rx - sy;
  R[7] = oz * ry + sx;
  R[8] = c + oz * rz;
}

// The forward of one mesh on one 64-lane wavefront (pose_fwd_kernel's body; also a role of the fused
// pose + blend kernel in blend3.hip): Rodrigues, pose feature, joints from betas, the chain, A and J_transformed.
// L / sc: this wave's LDS (transforms; the mesh's coefficient column).  coef / coef3 may be NULL.
__device__ __forceinline__ void pose_fwd_wave(
    const float *__restrict__ x, int x_stride, int num_cam, int B, int n, int lane,
    const float *__restrict__ J_template, const float *__restrict__ J_dirs,
    const int *__restrict__ parents, float *__restrict__ coef, int ldc, u32x4 *__restrict__ coef3,
    float *__restrict__ Rs_out, float *__restrict__ J_out, float *__restrict__ A_out,
    float *__restrict__ newJ_out, PoseLds &L, float *sc) {
  const bool live = n < B;
  // the kinematic tree, one entry per lane, fetched once: the chain loop below takes parent(i) with
  // v_readlane instead of paying a scalar-load round trip per joint
  const int par = parents[lane < 24 ? lane : 0];
  const float *xr = x + (size_t)(live ? n : 0) * x_stride;
  const float *beta = xr + num_cam + 72;

  if (live) {
    float *cf = coef ? coef + n : nullptr;   // k-major: coef[k][n], row stride ldc (optional)
    if (lane < 24) {
      float t[3] = {xr[num_cam + 3 * lane], xr[num_cam + 3 * lane + 1], xr[num_cam + 3 * lane + 2]};
      float R[9];
      rodrigues(t, R);
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        L.Rs[lane][e] = R[e];
        Rs_out[((size_t)n * 24 + lane) * 9 + e] = R[e];
      }
      if (lane >= 1) {
#pragma unroll
        for (int e = 0; e < 9; ++e) {
          const float pf = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
          sc[10 + 9 * (lane - 1) + e] = pf;
          if (cf) cf[(size_t)(10 + 9 * (lane - 1) + e) * ldc] = pf;
        }
      }
    }
    if (lane < 10) {
      sc[lane] = beta[lane];
      if (cf) cf[(size_t)lane * ldc] = beta[lane];
    }
    if (lane >= 10 && lane < 17) sc[207 + lane] = 0.0f;                        // 217..223
    if (cf && lane >= 10 && lane < 13) cf[(size_t)(207 + lane) * ldc] = 0.0f;  // 217..219
    for (int e = lane; e < 72; e += 64) {
      float acc = J_template[e];
#pragma unroll
      for (int k = 0; k < 10; ++k) acc += J_dirs[e * 10 + k] * beta[k];
      L.J[e / 3][e % 3] = acc;
      J_out[(size_t)n * 72 + e] = acc;
    }
  }
  wave_sync();
  if (live && coef3 && lane < 28) {
    // the same column as bf16x3 MFMA A-fragments (blend3.hip): lane = (k-tile, half) of this mesh's row
    // of its 32-mesh tile; [mesh tile][k-tile 14][split 3][lane 64] x 16 B
    const int kt = lane >> 1, hh = lane & 1;
    float xk[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xk[j] = sc[kt * 16 + 8 * hh + j];
    const Frag3 f = split8(xk);
    u32x4 *o = coef3 + ((size_t)((n >> 5) * 14 + kt) * 3) * 64 + hh * 32 + (n & 31);
    o[0] = __builtin_bit_cast(u32x4, f.h);
    o[64] = __builtin_bit_cast(u32x4, f.m);
    o[128] = __builtin_bit_cast(u32x4, f.l);
  }
  // root
  if (live && lane < 12) {
    const int r = lane >> 2, c = lane & 3;
    L.G[0][lane] = (c < 3) ? L.Rs[0][r * 3 + c] : L.J[0][r];
  }
  wave_sync();
  if (is_smpl_tree(par, lane)) {                              // wave-uniform
    const int slot = lane / 12, el = lane - 12 * slot;        // lanes 0..59: (joint slot, element of its 3 x 4)
    const int r = el >> 2, c = el & 3;
#pragma unroll
    for (int lv = 0; lv < TREE_LEVELS; ++lv) {
      const int i = slot_joint(TREE_LVL[lv], slot), p = slot_parent(TREE_LVL[lv], slot);
      if (live && i > 0) {
        float acc;
        if (c < 3) {
          acc = L.G[p][r * 4 + 0] * L.Rs[i][0 * 3 + c] + L.G[p][r * 4 + 1] * L.Rs[i][1 * 3 + c] +
                L.G[p][r * 4 + 2] * L.Rs[i][2 * 3 + c];
        } else {
          acc = L.G[p][r * 4 + 0] * (L.J[i][0] - L.J[p][0]) + L.G[p][r * 4 + 1] * (L.J[i][1] - L.J[p][1]) +
                L.G[p][r * 4 + 2] * (L.J[i][2] - L.J[p][2]) + L.G[p][r * 4 + 3];
        }
        L.G[i][el] = acc;
      }
      wave_sync();
    }
  } else {
#pragma unroll
    for (int i = 1; i < 24; ++i) {
      const int p = __builtin_amdgcn_readlane(par, i);
      if (live && lane < 12) {
        const int r = lane >> 2, c = lane & 3;
        float acc;
        if (c < 3) {
          acc = L.G[p][r * 4 + 0] * L.Rs[i][0 * 3 + c] + L.G[p][r * 4 + 1] * L.Rs[i][1 * 3 + c] +
                L.G[p][r * 4 + 2] * L.Rs[i][2 * 3 + c];
        } else {
          acc = L.G[p][r * 4 + 0] * (L.J[i][0] - L.J[p][0]) + L.G[p][r * 4 + 1] * (L.J[i][1] - L.J[p][1]) +
                L.G[p][r * 4 + 2] * (L.J[i][2] - L.J[p][2]) + L.G[p][r * 4 + 3];
        }
        L.G[i][lane] = acc;
      }
      wave_sync();
    }
  }
  if (live) {
    for (int e = lane; e < 288; e += 64) {
      const int j = e / 12, rc = e % 12, r = rc >> 2, c = rc & 3;
      float v;
      if (c < 3) {
        v = L.G[j][rc];
      } else {
        v = L.G[j][r * 4 + 3] - (L.G[j][r * 4 + 0] * L.J[j][0] + L.G[j][r * 4 + 1] * L.J[j][1] +
                                 L.G[j][r * 4 + 2] * L.J[j][2]);
      }
      A_out[(size_t)n * 288 + e] = v;
    }
    for (int e = lane; e < 72; e += 64) newJ_out[(size_t)n * 72 + e] = L.G[e / 3][(e % 3) * 4 + 3];
  }
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


// ---- the backward of one mesh (pose_bwd_kernel's body), in three forms -------------------------------------------
//   FULL  (NT = 512): everything - what smplr_pose_bwd and the fp32-GEMM path of smplr_smpl_bwd run.
//   CHAIN (NT = 256): the part that needs only the SKINNING backward's partials - their sums (dA, dcam), the
//          kinematic chain's backward in closed form (dR, dJ) and the J_dirs^T dJ chunk sums of d beta - written
//          to `mid` (B, 280): [dR 216 | chunk sums 60 | dcam 4].  Runs as a role of the blend GEMM's backward
//          launch (blend3.hip), on the SIMD slots its one-wave-per-SIMD workgroups leave free: 2.7 us of chain and
//          the skinning partials' round trip leave the critical path.
//   TAIL  (NT = 256): the rest - the blend GEMM's split-K partial sums (dcoef), dR + d(pose feature), Rodrigues'
//          backward, d beta, dcam -> dx.
// Every item is computed by the same expression in each form (items are looped over NT threads), so CHAIN followed
// by TAIL gives the bits of FULL.
enum { POSE_BWD_FULL = 0, POSE_BWD_CHAIN = 1, POSE_BWD_TAIL = 2 };
constexpr int POSE_MID = 280;

struct PoseBwdArgs {
  const float *x; int x_stride, num_cam, B;
  const float *J_dirs; const int *parents;
  const float *Rs_in, *J_in, *A_in;
  const float *dcoef, *dA, *dnewJ, *dcam;      // granular mode (dA != NULL): already reduced inputs
  float *dx;
  // fused mode (dA == NULL): sum the skinning partials (B,nblk,292) and the split-K partials (ns,nmt,32,224) of the
  // blend GEMM here, in fixed order, instead of in two more launches
  const float *skin_part; int nblk; const float *blend_part; int ns, nmt;
  int want_dcam;
  float *mid;                                  // CHAIN writes, TAIL reads (B, POSE_MID)
};

template <int NT, int MODE>
__device__ __forceinline__ void pose_bwd_body(const PoseBwdArgs &a, int n, PoseLds &L, float *sJd) {
  const int tid = threadIdx.x, lane = tid & 63;
  const bool live = true;
  const size_t nn = n;
  const int num_cam = a.num_cam;
  // this lane's joint angles, needed only by the Rodrigues backward at the very end: requested with everything
  // else (a load there sat alone on the critical path)
  float th[3] = {0.f, 0.f, 0.f};
  int tdesc = 0, tchild = 0;                       // the SMPL tree's masks for joint `tid`, requested with the rest
  if (tid < 24) {
    if (MODE != POSE_BWD_CHAIN) {
      const float *xr0 = a.x + nn * a.x_stride + num_cam + 3 * tid;
      th[0] = xr0[0]; th[1] = xr0[1]; th[2] = xr0[2];
    }
    tdesc = SMPL_TREE_DESC[tid];
    tchild = SMPL_TREE_CHILD[tid];
  }
  if (MODE != POSE_BWD_TAIL) {
    for (int e = tid; e < 720; e += NT) sJd[e] = a.J_dirs[e];
    for (int e = tid; e < 216; e += NT) L.Rs[e / 9][e % 9] = a.Rs_in[nn * 216 + e];
    for (int e = tid; e < 72; e += NT) L.J[e / 3][e % 3] = a.J_in[nn * 72 + e];
    for (int e = tid; e < 288; e += NT) L.G[e / 12][e % 12] = a.A_in[nn * 288 + e];  // G.R = A.R
  }
  float *scratch = &L.dA[0][0];                    // (the d beta chunk sums live here at the end: dA is free by then)
  if (MODE == POSE_BWD_TAIL) {
    // what the chain role left: dR, the chunk sums, dcam
    for (int e = tid; e < POSE_MID; e += NT) {
      const float v = a.mid[nn * POSE_MID + e];
      if (e < 216) L.dR[e / 9][e % 9] = v;
      else if (e < 276) scratch[e - 216] = v;
      else L.dcam[e - 276] = v;
    }
  }
  if (a.dA) {
    for (int e = tid; e < 288; e += NT) L.dA[e / 12][e % 12] = a.dA[nn * 288 + e];
    for (int e = tid; e < 220; e += NT) L.dcoef[e] = a.dcoef[nn * SMPLR_KPAD + e];
    if (tid < 4) L.dcam[tid] = a.dcam ? a.dcam[nn * 4 + tid] : 0.0f;
  } else {
    // 512 sums, each with up to PB_INFLIGHT of its partials requested at once (clamped addresses, surplus terms
    // replaced by 0): one round trip for the whole reduction.  The order of the additions is the producers' slice /
    // block order, as in the stand-alone reduce kernels.  CHAIN takes the skinning's 292, TAIL the GEMM's 220.
    constexpr int PB_INFLIGHT = 60;
    const int e_beg = MODE == POSE_BWD_TAIL ? 292 : 0, e_end = MODE == POSE_BWD_CHAIN ? 292 : 512;
    for (int e = e_beg + tid; e < e_end; e += NT) {
      const bool skin = e < 292;
      const int cnt = skin ? a.nblk : a.ns;
      const size_t mt = nn >> 5, r = nn & 31;
      const float *p = skin ? a.skin_part + (nn * a.nblk) * 292 + e
                            : a.blend_part + (mt * 32 + r) * 224 + min(e - 292, 219);
      const size_t stride = skin ? (size_t)292 : (size_t)a.nmt * 32 * 224;
      float acc = 0.0f;
      for (int s0 = 0; s0 < cnt; s0 += PB_INFLIGHT) {
        float v[PB_INFLIGHT];
#pragma unroll
        for (int u = 0; u < PB_INFLIGHT; ++u) v[u] = p[(size_t)min(s0 + u, cnt - 1) * stride];
#pragma unroll
        for (int u = 0; u < PB_INFLIGHT; ++u) acc += (s0 + u < cnt) ? v[u] : 0.0f;
      }
      if (e < 288) L.dA[e / 12][e % 12] = acc;
      else if (e < 292) L.dcam[e - 288] = a.want_dcam ? acc : 0.0f;
      else if (e - 292 < 220) L.dcoef[e - 292] = acc;
    }
  }
  __syncthreads();
  if (MODE != POSE_BWD_TAIL) {
    const int par = a.parents[lane < 24 ? lane : 0];
    if (tid < 24) {
      const int i = tid;
      const float *dAi = &L.dA[i][0];
      float dAt[3] = {dAi[3], dAi[7], dAi[11]};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) L.dGR[i][r * 3 + c] = dAi[r * 4 + c] - dAt[r] * L.J[i][c];
        L.dGt[i][r] = dAt[r] + (a.dnewJ ? a.dnewJ[nn * 72 + i * 3 + r] : 0.0f);
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
    // is 23 dependent LDS round trips on ONE wave.  Written in the world frame -
    // H_i = dGR_i G_i.R^T, with G_c.R R_c^T = G_p.R and J_c - J_p = G_p.R^T (t_c - t_p), t = world joint positions - it
    // telescopes into sums over subtrees:
    //     dGt_i = sum_{d in sub(i)} dGt_d^0,        H_i = sum_{d in sub(i)} Z_d - Y_i,
    //     Z_d = dGR_d^0 G_d.R^T + Y_d,   Y_d = dGt_d (x) (t_d - t_parent(d))  (Y_root = 0),
    //     dR_i = G_p.R^T H_i G_i.R   (root: H_0 G_0.R),
    //     dJ_i = dJ_i^0 + G_p.R^T dGt_i - sum_{c child of i} G_i.R^T dGt_c   (root: + dGt_0),
    // i.e. five phases whose items (joint x matrix element) are independent, the subtree sums taken in index order from
    // each joint's descendant mask: the same gradient, a fixed summation order, the whole workgroup instead of one wave.
    // Items: group A = 216 (joint, row, column) items, group B = 72 (joint, component) items; with 512 threads
    // threads 256.. take group B beside group A, with 256 the first 72 threads take it after their group-A item.
    // scratch: L.dA[i][0..8] = X_i, then T_i = H_i G_i.R;  L.dA[i][9..11] = t_i;  L.tmpv = bone vectors, then G_p.R^T dGt;  L.dGR = H
    const int bi = NT >= 512 ? tid - 256 : tid;          // this thread's group-B item (valid: 0 <= bi < 72)
    const bool hasB = bi >= 0 && bi < 72;
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
    }
    float bone = 0.0f;
    if (hasB) {                                              //     and the bone vectors t_d - t_parent(d) (root: 0)
      const int d = bi / 3, c = bi - 3 * d, p = L.par[d];
      bone = (p >= 0) ? L.dA[d][9 + c] - L.dA[p][9 + c] : 0.0f;   // (columns 9..11: not written by P2)
      L.tmpv[d][c] = bone;
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
    }
    if (hasB) {                                              //     and G_p.R^T dGt_i
      const int i = bi / 3, c = bi - 3 * i, p = L.par[i];
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
    }
    if (hasB) {                                              //     and dJ_i
      const int i = bi / 3, c = bi - 3 * i, m = L.child[i];
      float v[24];
#pragma unroll
      for (int d = 0; d < 24; ++d) v[d] = L.tmpv[d][c];
      float acc = L.dJ[i][c] + L.tmpv[i][c];
#pragma unroll
      for (int d = 0; d < 24; ++d) acc -= ((m >> d) & 1) ? v[d] : 0.0f;
      L.dJ[i][c] = acc;
    }
    __syncthreads();
    if (tid >= 64) {
      if (MODE != POSE_BWD_CHAIN) return;                  // the rest is one wavefront's work
    } else if (live && lane < 60) {
      // d beta = dcoef[0..9] + J_dirs^T dJ: 10 x 72 products, over 60 lanes (6 chunks of 12 per beta), chunk sums
      // parked in L.dA (free by now)
      const int k = lane / 6, part = lane - 6 * k;
      float acc = 0.0f;
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const int e = part * 12 + q;
        acc += L.dJ[e / 3][e % 3] * sJd[e * 10 + k];
      }
      scratch[lane] = acc;
    }
    if (MODE == POSE_BWD_CHAIN) {
      __syncthreads();
      for (int e = tid; e < POSE_MID; e += NT)
        a.mid[nn * POSE_MID + e] = e < 216 ? L.dR[e / 9][e % 9] : (e < 276 ? scratch[e - 216] : L.dcam[e - 276]);
      return;
    }
    wave_sync();
  } else {
    if (tid >= 64) return;                                 // TAIL: the rest is one wavefront's work
  }
  if (live) {
    float *dxr = a.dx + nn * a.x_stride;
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
