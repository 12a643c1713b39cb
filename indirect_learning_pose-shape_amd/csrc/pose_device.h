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

}  // namespace smplr
