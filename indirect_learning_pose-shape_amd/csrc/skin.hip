// K3 linear-blend skinning (+ orthographic projection epilogue) and its backward, plus the
// stand-alone projection op.
//
// Reference: keras_smpl/batch_smpl.py:135-145 (W tiled to (N,6890,24), T = W x A, v_homo =
// T x [v_posed;1]) and keras_smpl/projection.py:54-81.  The reference materialises W (0.66 MB
// per mesh) and T (0.44 MB per mesh); here a lane owns one vertex, holds its (up to 4 non-zero,
// else all 24) skinning weights in registers and blends the mesh's 24x12 joint matrix out of
// LDS; every global operand of a block is requested before its first barrier.
//
// Backward: dv_posed = T^T g per vertex on the VALU;  dA[j] = sum_v w[v][j] * (g (x) [v_posed;1])
// is a (24 x Vchunk) x (Vchunk x 12) product per mesh and runs on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32, two 16-joint tiles, K = 64 vertices per wave; the weights operand is
// read from the weight table directly in MFMA layout), reduced across the block's waves in LDS
// and across blocks in a fixed order by pose_bwd / skin_bwd_reduce_kernel (no atomics).
#include "common.h"

namespace smplr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SK_MB = 1;       // meshes per thread in the forward (weights stay in registers)
constexpr int SKB_T = 256;     // backward block: 256 vertices (4 waves)

// top4 (V,8): per vertex its (up to) 4 non-zero skinning weights and their joint indices (as floats),
// or NULL.  Real SMPL rows have <= 4 non-zeros, so T = sum_j w_j A_j needs 4 x 12 FMAs, not 24 x 12;
// dropping exact-zero terms in index order leaves every fp32 result bit-identical.  Rows with more
// non-zeros make the host pass NULL and the dense path runs.
template <bool SPARSE>
__global__ __launch_bounds__(256) void skin_fwd_kernel(const float *__restrict__ v_posed,
                                                       const float *__restrict__ lbs,
                                                       const float *__restrict__ top4,
                                                       const float *__restrict__ A,
                                                       const float *__restrict__ cam, int x_stride,
                                                       int B, int V, int vs, int VP,
                                                       float *__restrict__ verts,
                                                       float *__restrict__ proj) {
  __shared__ float4 sAj[SPARSE ? 72 : 1];
  const int v = blockIdx.x * 256 + threadIdx.x;
  const bool live = v < V;
  const int vc = live ? v : V - 1;
  float w[SPARSE ? 1 : 24];
  float4 ww = {0.f, 0.f, 0.f, 0.f}, jj = {0.f, 0.f, 0.f, 0.f};
  if (SPARSE) {
    const float4 *tp = reinterpret_cast<const float4 *>(top4 + (size_t)vc * 8);
    ww = tp[0];
    jj = tp[1];
  } else {
    const float4 *wp = reinterpret_cast<const float4 *>(lbs + (size_t)vc * 24);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const float4 t = wp[q];
      w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
    }
  }
  const bool sampled = (vs <= 1) || (v % vs == 0);
  const int vp_idx = (vs <= 1) ? v : v / vs;
  static_assert(SK_MB == 1, "one mesh per block");
  {
    const int n = blockIdx.y;                // block-uniform
    const float *An = A + (size_t)n * 288;
    // every global operand is requested before the barrier below: one round trip per block
    const float *vp = v_posed + ((size_t)n * V + vc) * 3;
    const float p0 = vp[0], p1 = vp[1], p2 = vp[2];
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    if (proj) {
      const float *c = cam + (size_t)n * x_stride;
      c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3];
    }
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.0f;
    if (SPARSE) {
      if (threadIdx.x < 72) sAj[threadIdx.x] = reinterpret_cast<const float4 *>(An)[threadIdx.x];
      __syncthreads();
      skin_T_sparse(sAj, ww, jj, T);
    } else {
#pragma unroll
      for (int j = 0; j < 24; ++j)
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = fmaf(w[SPARSE ? 0 : j], An[j * 12 + e], T[e]);
    }
    float X, Y, Z;
    skin_apply(T, p0, p1, p2, X, Y, Z);
    if (live) {
      if (verts) {
        float *o = verts + ((size_t)n * V + v) * 3;
        SMPLR_OUT_STORE(&o[0], X); SMPLR_OUT_STORE(&o[1], Y); SMPLR_OUT_STORE(&o[2], Z);
      }
      if (proj && sampled) {                       // (default policy: the binning / silhouette kernels read it next)
        float *o = proj + ((size_t)n * VP + vp_idx) * 3;
        o[0] = project_u(X, c0, c2);
        o[1] = project_u(Y, c1, c3);
        o[2] = Z;
      }
    }
  }
}

// (7 waves per SIMD: 3456 workgroups of 4 waves at B = 128 are then 1.93 rounds of the 1792 that fit, not 2.25 of 1536;
// needs <= 72 registers: 65 with T built row by row)
// One block = 256 vertices x SKB_MB meshes (the 24 skinning weights of a vertex are loaded once and
// kept in registers / LDS for all of them).  part layout per (mesh, block): 288 dA + 4 dcam floats.
#ifdef SMPLR_TL
constexpr int TL_SKIN_WG = 3456;
__device__ unsigned g_tl_skin[TL_SKIN_WG * (SKB_T / 64) * 32];
#endif
constexpr int SKB_PART = 292;
constexpr int SKB_MB = 1;

template <bool SPARSE>
__global__ __launch_bounds__(SKB_T) __attribute__((amdgpu_waves_per_eu(SPARSE ? 7 : 4, 8))) void skin_bwd_kernel(
    const float *__restrict__ dverts, const float *__restrict__ dproj, const float *__restrict__ v_posed,
    const float *__restrict__ lbs, const float *__restrict__ top4, const float *__restrict__ A,
    const float *__restrict__ cam,
    int x_stride, int B, int V, int vs, int VP, float *__restrict__ dv_posed, float *__restrict__ part,
    const float *__restrict__ seg_part, const short *__restrict__ seg_vslot, int seg_nsplit) {
  __shared__ float sG[SKB_T][4];    // g (3) per vertex
  __shared__ float sP[SKB_T][4];    // [v_posed;1]
  __shared__ float sRed[SKB_T / 64][SKB_PART];
  __shared__ float4 sAj[72];        // this mesh's 24 x 12 joint matrix
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int v = blockIdx.x * SKB_T + tid;
  const bool live = v < V;
  const int vc = live ? v : V - 1;
  const bool sampled = (vs <= 1) || (v % vs == 0);
  const int vpi = (vs <= 1) ? v : v / vs;

  static_assert(SKB_MB == 1, "one mesh per block");
  const int n = blockIdx.y;                   // block-uniform
  SMPLR_TL_WAVE(g_tl_skin, SKB_T / 64, blockIdx.y * gridDim.x + blockIdx.x, TL_SKIN_WG)
  // Every global operand of the block is requested here, before the first barrier: the weights,
  // the mesh's joint matrix, the vertex and its incoming gradients (clamped, unconditional loads:
  // a per-lane condition around a load costs a branch and a drained vmcnt each), so the block
  // pays one round trip to memory instead of three.
  float w[SPARSE ? 1 : 24];
  if (!SPARSE) {
    const float4 *wp = reinterpret_cast<const float4 *>(lbs + (size_t)vc * 24);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const float4 t = wp[q];
      w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
    }
  }
  const int li = lane & 15, lk = lane >> 4;
  float w4[4] = {0.f, 0.f, 0.f, 0.f};
  int jx[4] = {0, 0, 0, 0};
  if (SPARSE) {
    const float4 *tp = reinterpret_cast<const float4 *>(top4 + (size_t)vc * 8);
    const float4 ww = tp[0], jj = tp[1];
    w4[0] = ww.x; w4[1] = ww.y; w4[2] = ww.z; w4[3] = ww.w;
    jx[0] = (int)jj.x; jx[1] = (int)jj.y; jx[2] = (int)jj.z; jx[3] = (int)jj.w;
  }
  const float4 aj = reinterpret_cast<const float4 *>(A + (size_t)n * 288)[tid < 72 ? tid : 71];
  const float *vp = v_posed + ((size_t)n * V + vc) * 3;
  const float p0 = vp[0], p1 = vp[1], p2 = vp[2];
  float gv0 = 0.f, gv1 = 0.f, gv2 = 0.f, gp0 = 0.f, gp1 = 0.f, gp2 = 0.f, ck0 = 0.f, ck1 = 0.f;
  if (dverts) {                               // block-uniform
    const float *d = dverts + ((size_t)n * V + vc) * 3;
    gv0 = d[0]; gv1 = d[1]; gv2 = d[2];
  }
  const bool has_proj = dproj || seg_vslot;   // block-uniform
  if (dproj) {                                // block-uniform
    const float *d = dproj + ((size_t)n * VP + min(vpi, VP - 1)) * 3;
    gp0 = d[0]; gp1 = d[1]; gp2 = d[2];
  }
  if (has_proj) {
    const float *c = cam + (size_t)n * x_stride;
    ck0 = c[0]; ck1 = c[1];
  }
  if (seg_vslot) {                            // block-uniform
    // d(seg)/d(proj) of this vertex = its record slot's sums over the segmentation backward's row blocks,
    // added in block order (what seg_bwd_merge_kernel would have stored in dproj); one extra hop: slot -> sums
    const int slot = seg_vslot[(size_t)n * VP + min(vpi, VP - 1)];
    const int sl = max(slot, 0), win = sl / SB_SLOTS;
    const float *sp = seg_part + ((size_t)n * seg_nsplit * SB_NWIN + win) * (SB_SLOTS * 2) + (sl - win * SB_SLOTS) * 2;
    float sx = 0.0f, sy = 0.0f;
    if (seg_nsplit <= 2) {                    // block-uniform: large batches (24 rows per row block, W <= 48)
      const float2 t0 = *reinterpret_cast<const float2 *>(sp);
      const float2 t1 = *reinterpret_cast<const float2 *>(sp + (size_t)(seg_nsplit - 1) * (SB_NWIN * SB_SLOTS * 2));
      sx = t0.x + (seg_nsplit > 1 ? t1.x : 0.0f);
      sy = t0.y + (seg_nsplit > 1 ? t1.y : 0.0f);
    } else {
      constexpr int GC = 6;                   // row blocks requested together (W = 48 at 8 rows: all six)
      for (int s0 = 0; s0 < seg_nsplit; s0 += GC) {
        float2 t[GC];
#pragma unroll
        for (int u = 0; u < GC; ++u)
          t[u] = *reinterpret_cast<const float2 *>(sp + (size_t)min(s0 + u, seg_nsplit - 1) * (SB_NWIN * SB_SLOTS * 2));
#pragma unroll
        for (int u = 0; u < GC; ++u) {
          sx += (s0 + u < seg_nsplit) ? t[u].x : 0.0f;
          sy += (s0 + u < seg_nsplit) ? t[u].y : 0.0f;
        }
      }
    }
    if (slot >= 0) { gp0 += sx; gp1 += sy; }
  }
  if (tid < 72) sAj[tid] = aj;
  SMPLR_TL_STAMP(1);
  __syncthreads();
  SMPLR_TL_STAMP(2);

  const int cr = li >> 2, cc = li & 3;   // dT component j = li = r*4+c  (valid for li < 12)

  {
    // T = sum_j w_j A_j with the mesh's joint matrix staged in LDS (broadcast ds_read_b128) and
    // w_j taken from LDS as well: as scalar operands the 288 matrix entries need more SGPRs than
    // exist (the compiler then spills through v_readlane or falls back to 288 vector loads).
    // ... one ROW of T at a time (4 live values instead of 12: the kernel's register peak was here, and 72 registers
    // are what a seventh wave per SIMD needs): row r gives dv_posed its g_r terms, rows 0 and 1 the projected X and Y
    const bool proj_on = live && has_proj && sampled;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    float dku = 0.f, dkv = 0.f, du0 = 0.f, dv0 = 0.f;
    if (live) { g0 = gv0; g1 = gv1; g2 = gv2; }
    // (explicit fmas: the sparse and the dense instantiation must round alike, whatever the compiler would contract)
    if (proj_on) { g0 = fmaf(ck0, gp0, g0); g1 = fmaf(ck1, gp1, g1); g2 += gp2; du0 = gp0; dv0 = gp1; }
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
      for (int jq = 0; jq < (SPARSE ? 4 : 24); ++jq) {
        const int j = SPARSE ? jx[jq & 3] : jq;
        const float wj = SPARSE ? w4[jq & 3] : w[SPARSE ? 0 : jq];
        const float4 rr = sAj[j * 3 + r];
        t0 = fmaf(wj, rr.x, t0); t1 = fmaf(wj, rr.y, t1); t2 = fmaf(wj, rr.z, t2); t3 = fmaf(wj, rr.w, t3);
      }
      const float gr = r == 0 ? g0 : (r == 1 ? g1 : g2);
      dp0 = fmaf(t0, gr, dp0); dp1 = fmaf(t1, gr, dp1); dp2 = fmaf(t2, gr, dp2);
      if (r < 2 && proj_on) {
        const float XY = fmaf(t2, p2, fmaf(t1, p1, fmaf(t0, p0, t3)));
        if (r == 0) dku = XY * gp0; else dkv = XY * gp1;
      }
    }
    if (live) {
      float *o = dv_posed + ((size_t)n * V + v) * 3;
      o[0] = dp0; o[1] = dp1; o[2] = dp2;
    }
    // A operand of the dA product, straight from the (L2-resident) weight table in MFMA layout:
    // lane (li, lk) of step s holds w[vertex 64 wave + 4 s + lk][joint li] and [joint 16 + li]
    // (4 rows x 64-B segments per load); 32 loads per wave, requested once T is done (their latency
    // overlaps the barrier; asked for at the top they cost 32 live registers = one wave per SIMD).
    SMPLR_TL_STAMP(3);
    float wa0[8], wa1[8];        // a ring of 8 steps: steps 8..15 are requested as the first eight are consumed
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(lbs), 0, V * 24 * 4, 0x00020000);
    const int vb = blockIdx.x * SKB_T + wave * 64 + lk;
    const int o0 = (vb * 24 + li) * 4, o1 = (vb * 24 + 16 + (li & 7)) * 4;
    {
      // buffer loads: a 128-bit descriptor of the weight table (wave-uniform) + ONE 32-bit byte offset per lane
      // + an immediate per step, rows past the end of the table read as 0 (the hardware's range check; their
      // vertices carry g = 0 anyway) - instead of a clamp and a 64-bit multiply-add per request (a fifth of
      // the kernel's vector instructions)
  #pragma unroll
      for (int sI = 0; sI < 8; ++sI) {
        wa0[sI] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o0 + sI * 384, 0, 0));
        wa1[sI] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o1 + sI * 384, 0, 0));
      }
    }
    sG[tid][0] = g0; sG[tid][1] = g1; sG[tid][2] = g2; sG[tid][3] = 0.f;
    sP[tid][0] = p0; sP[tid][1] = p1; sP[tid][2] = p2; sP[tid][3] = 1.0f;
    __syncthreads();
    SMPLR_TL_STAMP(4);

    // dA tile on the matrix cores: D[joint][comp] += sum_k w[vk][joint] * g[vk][comp>>2]*ph[vk][comp&3]
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int t = wave * 64 + s * 4 + lk;     // tail vertices carry g = 0 and finite (clamped) weights
      const float a0 = wa0[s & 7];
      const float a1 = (li < 8) ? wa1[s & 7] : 0.0f;
      const float b = (li < 12) ? sG[t][cr] * sP[t][cc] : 0.0f;
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc1, 0, 0, 0);
      if (s < 8) {
        wa0[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o0 + (s + 8) * 384, 0, 0));
        wa1[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o1 + (s + 8) * 384, 0, 0));
      }
    }
    SMPLR_TL_STAMP(5);
    // C/D layout 16x16: col = lane&15 (component), row = (lane>>4)*4 + reg (joint in tile)
    if (li < 12) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j0 = lk * 4 + r;
        sRed[wave][j0 * 12 + li] = acc0[r];
        if (j0 < 8) sRed[wave][(16 + j0) * 12 + li] = acc1[r];
      }
    }
    const float r0 = wave_sum(dku), r1 = wave_sum(dkv), r2 = wave_sum(du0), r3 = wave_sum(dv0);
    if (lane == 0) {
      sRed[wave][288] = r0; sRed[wave][289] = r1; sRed[wave][290] = r2; sRed[wave][291] = r3;
    }
    __syncthreads();
    SMPLR_TL_STAMP(6);
    for (int e = tid; e < SKB_PART; e += SKB_T) {
      float acc = 0.f;
#pragma unroll
      for (int wv = 0; wv < SKB_T / 64; ++wv) acc += sRed[wv][e];
      part[((size_t)n * gridDim.x + blockIdx.x) * SKB_PART + e] = acc;
    }
    SMPLR_TL_STAMP(7);
  }
}

__global__ __launch_bounds__(320) void skin_bwd_reduce_kernel(const float *__restrict__ part, int nblk,
                                                              float *__restrict__ dA,
                                                              float *__restrict__ dcam) {
  const int n = blockIdx.x, e = threadIdx.x;
  if (e >= SKB_PART) return;
  float acc = 0.f;
  for (int b = 0; b < nblk; ++b) acc += part[((size_t)n * nblk + b) * SKB_PART + e];
  if (e < 288) dA[(size_t)n * 288 + e] = acc;
  else if (dcam) dcam[(size_t)n * 4 + (e - 288)] = acc;
}

// ---------------------------------------------------------------- stand-alone projection
__global__ __launch_bounds__(256) void project_fwd_kernel(const float *__restrict__ verts,
                                                          const float *__restrict__ cam, int x_stride,
                                                          int V, int vs, int VP, float *__restrict__ proj) {
  const int n = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= VP) return;
  const float *c = cam + (size_t)n * x_stride;
  const float *p = verts + ((size_t)n * V + (size_t)i * vs) * 3;
  float *o = proj + ((size_t)n * VP + i) * 3;
  o[0] = c[2] + p[0] * c[0];
  o[1] = c[3] + p[1] * c[1];
  o[2] = p[2];
}

__global__ __launch_bounds__(256) void project_bwd_kernel(const float *__restrict__ dproj,
                                                          const float *__restrict__ verts,
                                                          const float *__restrict__ cam, int x_stride,
                                                          int V, int vs, int VP, float *__restrict__ dverts,
                                                          float *__restrict__ dcam) {
  __shared__ float red[4][4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float *c = cam + (size_t)n * x_stride;
  const float ku = c[0], kv = c[1];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int v = tid; v < V; v += 256) {
    float *o = dverts + ((size_t)n * V + v) * 3;
    if (vs <= 1 || v % vs == 0) {
      const float *d = dproj + ((size_t)n * VP + (vs <= 1 ? v : v / vs)) * 3;
      const float *p = verts + ((size_t)n * V + v) * 3;
      const float du = d[0], dv = d[1];
      o[0] = ku * du; o[1] = kv * dv; o[2] = d[2];
      s0 += p[0] * du; s1 += p[1] * dv; s2 += du; s3 += dv;
    } else {
      o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
    }
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
  if ((tid & 63) == 0) {
    red[tid >> 6][0] = s0; red[tid >> 6][1] = s1; red[tid >> 6][2] = s2; red[tid >> 6][3] = s3;
  }
  __syncthreads();
  if (tid < 4) dcam[(size_t)n * 4 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// ------------------------------------------------------------------------------------------------
// The skinning backward over RECORDS instead of vertices (round 5).  When the only gradient that reaches the vertices is
// the part rasteriser's (seg_bwd's slot sums; no d verts, no merged d proj: the decoder's training step), a vertex without
// a record - hidden and further than 0.208 px from every pixel centre: 78 % of them - has g = 0: its dv_posed row is
// zero and it adds nothing to dA or the camera sums, yet skin_bwd_kernel loads its weights and v_posed row, blends its T
// and feeds it to the dA product like any other.  Here a workgroup takes 1 024 consecutive vertices, compacts those with
// a record ((vertex, slot) in vertex order: ballots + prefix sums, no atomics, so the order - and with it every sum - is
// the same on every launch) and runs skin_bwd_kernel's arithmetic on the compact list alone: the slot sums straight by
// slot (one hop instead of vertex -> slot -> sums), T rows 0 and 1 (row 2 multiplies g_z = 0), the dA tile on the matrix
// cores with the weights operand rebuilt from the records' (weight, joint) quadruples in LDS.  dv_posed is bit for bit
// skin_bwd_kernel's (rows without a record are stored as zeros); dA and the camera sums add the same terms in another
// order.  Measured (rocprofv3, same box): B = 128 19.3-20.1 -> 16.1 us, B = 2 048 233-236 -> 138 us; the step 0.1253 ->
// 0.1229 ms and 1.404 -> 1.297 ms (1.58 M meshes/s).  What the in-kernel stamps taught on the way
// (tools/probes/skinrec_timeline.py): gfx950 counts loads and stores in ONE in-order counter, so the zero rows stored
// during the compaction and the records' rows stored before the dA product made the next loads' waits 11.5 k and 4.7 k
// clocks long - every store now comes behind the kernel's last load; gathering the dense weight rows per record (as
// skin_bwd_kernel reads them) was a third dependent round trip of 5 k clocks.  A wave-private variant (no workgroup
// barriers before the final sum) and 512- / 2 048-vertex workgroups were slower (tools: SMPLR_SKIN_BWD_REC=0 runs the
// per-vertex kernel for A/B).
#ifdef SMPLR_TL
constexpr int TL_SKINREC_WG = 896;
__device__ unsigned g_tl_skinrec[TL_SKINREC_WG * 4 * 32];
#endif
constexpr int SKR_T = 256;         // threads
constexpr int SKR_VPT = 4;         // vertices per thread in the compaction pass
constexpr int SKR_CHUNK = SKR_T * SKR_VPT;

__global__ __launch_bounds__(SKR_T) __attribute__((amdgpu_waves_per_eu(4, 8))) void skin_bwd_rec_kernel(
    const float *__restrict__ v_posed, const float *__restrict__ top4,
    const float *__restrict__ A, const float *__restrict__ cam, int x_stride, int B, int V, int vs, int VP,
    float *__restrict__ dv_posed, float *__restrict__ part, const float *__restrict__ seg_part,
    const short *__restrict__ seg_vslot, int seg_nsplit) {
  __shared__ float sG[SKR_T][4];    // (g_x, g_y) per record of the round
  __shared__ float sP[SKR_T][4];    // [v_posed; 1]
  __shared__ float4 sW[SKR_T];      // its (up to) four skinning weights (zeros for a lane past the list) ...
  __shared__ float4 sJ[SKR_T];      // ... and their joints
  __shared__ float sRed[SKR_T / 64][SKB_PART];
  __shared__ float4 sAj[72];
  __shared__ int sVid[SKR_CHUNK];
  __shared__ short sSlot[SKR_CHUNK];
  __shared__ int sCnt[SKR_VPT * (SKR_T / 64)];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = blockIdx.y, v0 = blockIdx.x * SKR_CHUNK;
  SMPLR_TL_WAVE(g_tl_skinrec, SKR_T / 64, blockIdx.y * gridDim.x + blockIdx.x, TL_SKINREC_WG)
  const float4 aj = reinterpret_cast<const float4 *>(A + (size_t)n * 288)[tid < 72 ? tid : 71];
  const float *cm = cam + (size_t)n * x_stride;
  const float ck0 = cm[0], ck1 = cm[1];
  int slot[SKR_VPT];
#pragma unroll
  for (int i = 0; i < SKR_VPT; ++i) {
    const int v = v0 + i * SKR_T + tid;
    const bool sampled = v < V && ((vs <= 1) || (v % vs == 0));
    const int vpi = (vs <= 1) ? v : v / vs;
    slot[i] = seg_vslot[(size_t)n * VP + min(vpi, VP - 1)];
    if (!sampled) slot[i] = -1;
  }
  if (tid < 72) sAj[tid] = aj;
  SMPLR_TL_STAMP(1);
  int pre[SKR_VPT];
#pragma unroll
  for (int i = 0; i < SKR_VPT; ++i) {
    const unsigned long long m = __ballot(slot[i] >= 0);
    pre[i] = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) sCnt[i * (SKR_T / 64) + wave] = __popcll(m);
  }
  SMPLR_TL_STAMP(2);
  __syncthreads();
  SMPLR_TL_STAMP(3);
  int R = 0;
  {
    int base[SKR_VPT] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < SKR_VPT * (SKR_T / 64); ++k) {
      const int c = sCnt[k];
#pragma unroll
      for (int i = 0; i < SKR_VPT; ++i)
        if (k < i * (SKR_T / 64) + wave) base[i] += c;
      R += c;
    }
#pragma unroll
    for (int i = 0; i < SKR_VPT; ++i) {
      const int v = v0 + i * SKR_T + tid;
      if (slot[i] >= 0) {
        sVid[base[i] + pre[i]] = v;
        sSlot[base[i] + pre[i]] = (short)slot[i];
      }
    }
  }
  SMPLR_TL_STAMP(4);
  __syncthreads();
  SMPLR_TL_STAMP(5);
  const int li = lane & 15, lk = lane >> 4;
  const int cr = li >> 2, cc = li & 3;
  const float fl0 = (float)li, fl1 = (float)(16 + li);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float dku = 0.f, dkv = 0.f, du0 = 0.f, dv0 = 0.f;
  for (int r0 = 0; r0 < R; r0 += SKR_T) {                  // (block-uniform; one round for up to 256 records)
    const int idx = r0 + tid;
    const bool valid = idx < R;
    const int vid = sVid[min(idx, R - 1)], sl = sSlot[min(idx, R - 1)];
    const float4 *tp = reinterpret_cast<const float4 *>(top4 + (size_t)vid * 8);
    const float4 ww = tp[0], jj = tp[1];
    const float *vp = v_posed + ((size_t)n * V + vid) * 3;
    const float p0 = vp[0], p1 = vp[1], p2 = vp[2];
    // the slot's sums over seg_bwd's row blocks, in block order (skin_bwd_kernel's gather, without the vertex -> slot hop)
    const int win = sl / SB_SLOTS;
    const float *sp = seg_part + ((size_t)n * seg_nsplit * SB_NWIN + win) * (SB_SLOTS * 2) + (sl - win * SB_SLOTS) * 2;
    float sx = 0.0f, sy = 0.0f;
    if (seg_nsplit <= 2) {
      const float2 t0 = *reinterpret_cast<const float2 *>(sp);
      const float2 t1 = *reinterpret_cast<const float2 *>(sp + (size_t)(seg_nsplit - 1) * (SB_NWIN * SB_SLOTS * 2));
      sx = t0.x + (seg_nsplit > 1 ? t1.x : 0.0f);
      sy = t0.y + (seg_nsplit > 1 ? t1.y : 0.0f);
    } else {
      constexpr int GC = 6;
      for (int s0 = 0; s0 < seg_nsplit; s0 += GC) {
        float2 t[GC];
#pragma unroll
        for (int u = 0; u < GC; ++u)
          t[u] = *reinterpret_cast<const float2 *>(sp + (size_t)min(s0 + u, seg_nsplit - 1) * (SB_NWIN * SB_SLOTS * 2));
#pragma unroll
        for (int u = 0; u < GC; ++u) {
          sx += (s0 + u < seg_nsplit) ? t[u].x : 0.0f;
          sy += (s0 + u < seg_nsplit) ? t[u].y : 0.0f;
        }
      }
    }
    const float gp0 = valid ? sx : 0.0f, gp1 = valid ? sy : 0.0f;
    const float g0 = fmaf(ck0, gp0, 0.0f), g1 = fmaf(ck1, gp1, 0.0f);
    const float w4[4] = {ww.x, ww.y, ww.z, ww.w};
    const int jx[4] = {(int)jj.x, (int)jj.y, (int)jj.z, (int)jj.w};
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {                          // (row 2 of T only meets g_z = 0)
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) {
        const float4 rr = sAj[jx[jq] * 3 + r];
        t0 = fmaf(w4[jq], rr.x, t0); t1 = fmaf(w4[jq], rr.y, t1); t2 = fmaf(w4[jq], rr.z, t2); t3 = fmaf(w4[jq], rr.w, t3);
      }
      const float gr = r == 0 ? g0 : g1;
      dp0 = fmaf(t0, gr, dp0); dp1 = fmaf(t1, gr, dp1); dp2 = fmaf(t2, gr, dp2);
      const float XY = fmaf(t2, p2, fmaf(t1, p1, fmaf(t0, p0, t3)));
      if (r == 0) dku += XY * gp0; else dkv += XY * gp1;
    }
    du0 += gp0;
    dv0 += gp1;
    sG[tid][0] = g0; sG[tid][1] = g1; sG[tid][2] = 0.f; sG[tid][3] = 0.f;
    sP[tid][0] = p0; sP[tid][1] = p1; sP[tid][2] = p2; sP[tid][3] = 1.0f;
    sW[tid] = valid ? ww : make_float4(0.f, 0.f, 0.f, 0.f);
    sJ[tid] = jj;
    SMPLR_TL_STAMP(6);
    __syncthreads();
    SMPLR_TL_STAMP(7);
    if (r0 + wave * 64 < R) {                              // (wave-uniform: this wave's 64 records hold at least one)
      // dA tile: D[joint][comp] += sum_k w[record k][joint] * g[k][comp >> 2] * [v_posed; 1][k][comp & 3].  Lane (li, lk)
      // of step s holds record 64 wave + 4 s + lk: its weight for joint li (and 16 + li) - rebuilt from the record's
      // (weight, joint) quadruple in LDS: the quadruple's weight for that joint, else 0, the very numbers
      // skin_bwd_kernel gathers from the dense (V, 24) table (a third dependent round trip here: 5 k clocks)
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int t = wave * 64 + s * 4 + lk;
        const float4 wq = sW[t], jq = sJ[t];
        const float a0 = ((jq.x == fl0 ? wq.x : 0.0f) + (jq.y == fl0 ? wq.y : 0.0f)) +
                         ((jq.z == fl0 ? wq.z : 0.0f) + (jq.w == fl0 ? wq.w : 0.0f));
        const float a1 = ((jq.x == fl1 ? wq.x : 0.0f) + (jq.y == fl1 ? wq.y : 0.0f)) +
                         ((jq.z == fl1 ? wq.z : 0.0f) + (jq.w == fl1 ? wq.w : 0.0f));
        const float b = (li < 12) ? sG[t][cr] * sP[t][cc] : 0.0f;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(li < 8 ? a1 : 0.0f, b, acc1, 0, 0, 0);
      }
    }
    // (every store of the kernel comes behind its loads: gfx950 counts loads and stores in ONE in-order counter, so a
    // wait for a load is a wait for every store the wave issued before it - with the zero rows stored during the
    // compaction and the records' rows before the dA product, the waves sat 11.5 k and 4.7 k clocks in those two waits)
    if (valid) {
      float *o = dv_posed + ((size_t)n * V + vid) * 3;
      o[0] = dp0; o[1] = dp1; o[2] = dp2;
    }
    SMPLR_TL_STAMP(8);
    __syncthreads();                                       // (sG / sP / sRv are rewritten by the next round)
  }
  // the rows of the vertices without a record
#pragma unroll
  for (int i = 0; i < SKR_VPT; ++i) {
    const int v = v0 + i * SKR_T + tid;
    if (slot[i] < 0 && v < V) {
      float *o = dv_posed + ((size_t)n * V + v) * 3;
      o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f;
    }
  }
  SMPLR_TL_STAMP(9);
  if (li < 12) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j0 = lk * 4 + r;
      sRed[wave][j0 * 12 + li] = acc0[r];
      if (j0 < 8) sRed[wave][(16 + j0) * 12 + li] = acc1[r];
    }
  }
  const float q0 = wave_sum(dku), q1 = wave_sum(dkv), q2 = wave_sum(du0), q3 = wave_sum(dv0);
  if (lane == 0) {
    sRed[wave][288] = q0; sRed[wave][289] = q1; sRed[wave][290] = q2; sRed[wave][291] = q3;
  }
  __syncthreads();
  SMPLR_TL_STAMP(10);
  for (int e = tid; e < SKB_PART; e += SKR_T) {
    float acc = 0.f;
#pragma unroll
    for (int wv = 0; wv < SKR_T / 64; ++wv) acc += sRed[wv][e];
    part[((size_t)n * gridDim.x + blockIdx.x) * SKB_PART + e] = acc;
  }
  SMPLR_TL_STAMP(11);
#ifdef SMPLR_TL
  if (tl__) tl__[12] = (unsigned)R;
#endif
}

int skin_bwd_nblk(int V) { return (V + SKB_T - 1) / SKB_T; }

int launch_skin_bwd_partials(const float *dverts, const float *dproj, SegGrad sg, const float *v_posed,
                             const float *lbs_weights, const float *lbs_top4, const float *A, const float *cam,
                             int x_stride, int B, int V, int vs, float *dv_posed, float *part, hipStream_t st,
                             int *nblk_out) {
  const int VP = (V + vs - 1) / vs;
  // SMPLR_SKIN_BWD_REC=0: the per-vertex kernel for every case (A/B runs)
  static const bool rec_on = !(getenv("SMPLR_SKIN_BWD_REC") && atoi(getenv("SMPLR_SKIN_BWD_REC")) == 0);
  if (rec_on && !dverts && !dproj && sg.part && lbs_top4) {
    const dim3 grid((V + SKR_CHUNK - 1) / SKR_CHUNK, B);
    hipLaunchKernelGGL(skin_bwd_rec_kernel, grid, dim3(SKR_T), 0, st, v_posed, lbs_top4, A, cam, x_stride, B, V, vs, VP,
                       dv_posed, part, sg.part, reinterpret_cast<const short *>(sg.vslot), sg.nsplit);
    SMPLR_LAUNCH_CHECK("skin_bwd_rec_kernel");
    *nblk_out = (int)grid.x;
    return 0;
  }
  *nblk_out = skin_bwd_nblk(V);
  const dim3 grid(skin_bwd_nblk(V), (B + SKB_MB - 1) / SKB_MB);
  if (lbs_top4)
    hipLaunchKernelGGL(skin_bwd_kernel<true>, grid, dim3(SKB_T), 0, st, dverts, dproj, v_posed, lbs_weights, lbs_top4,
                       A, cam, x_stride, B, V, vs, VP, dv_posed, part, sg.part, reinterpret_cast<const short *>(sg.vslot),
                       sg.nsplit);
  else
    hipLaunchKernelGGL(skin_bwd_kernel<false>, grid, dim3(SKB_T), 0, st, dverts, dproj, v_posed, lbs_weights, lbs_top4,
                       A, cam, x_stride, B, V, vs, VP, dv_posed, part, sg.part, reinterpret_cast<const short *>(sg.vslot),
                       sg.nsplit);
  SMPLR_LAUNCH_CHECK("skin_bwd_kernel");
  return 0;
}

}  // namespace smplr

extern "C" {

static int check_vs(int V, int vs) { return vs >= 1 && vs <= V; }

int smplr_skin_fwd(const float *v_posed, const float *lbs_weights, const float *lbs_top4, const float *A,
                   const float *cam, int x_stride, int B, int V, int vertex_sampling, float *verts,
                   float *proj, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && V > 0 && check_vs(V, vertex_sampling), "smplr_skin_fwd: bad sizes B=%d V=%d vs=%d",
                B, V, vertex_sampling);
  if (B == 0) return 0;
  SMPLR_REQUIRE(v_posed && lbs_weights && A && (verts || proj), "smplr_skin_fwd: null pointer");
  SMPLR_REQUIRE(!proj || (cam && x_stride >= 4), "smplr_skin_fwd: proj requested without camera rows");
  const int VP = (V + vertex_sampling - 1) / vertex_sampling;
  dim3 grid((V + 255) / 256, (B + SK_MB - 1) / SK_MB);
  if (lbs_top4)
    hipLaunchKernelGGL(skin_fwd_kernel<true>, grid, dim3(256), 0, as_stream(stream), v_posed, lbs_weights, lbs_top4, A,
                       cam, x_stride, B, V, vertex_sampling, VP, verts, proj);
  else
    hipLaunchKernelGGL(skin_fwd_kernel<false>, grid, dim3(256), 0, as_stream(stream), v_posed, lbs_weights, lbs_top4, A,
                       cam, x_stride, B, V, vertex_sampling, VP, verts, proj);
  SMPLR_LAUNCH_CHECK("smplr_skin_fwd");
  return 0;
}

size_t smplr_skin_bwd_workspace(int B, int V) {
  using namespace smplr;
  if (B <= 0 || V <= 0) return 0;
  return (size_t)B * ((V + SKB_T - 1) / SKB_T) * SKB_PART * sizeof(float);
}

int smplr_skin_bwd(const float *dverts, const float *dproj, const float *v_posed,
                   const float *lbs_weights, const float *lbs_top4, const float *A, const float *cam,
                   int x_stride, int B, int V,
                   int vertex_sampling, float *dv_posed, float *dA, float *dcam, void *workspace,
                   void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && V > 0 && check_vs(V, vertex_sampling), "smplr_skin_bwd: bad sizes B=%d V=%d vs=%d",
                B, V, vertex_sampling);
  if (B == 0) return 0;
  SMPLR_REQUIRE(v_posed && lbs_weights && A && dv_posed && dA && workspace, "smplr_skin_bwd: null pointer");
  SMPLR_REQUIRE(dverts || dproj, "smplr_skin_bwd: need dverts and/or dproj");
  SMPLR_REQUIRE(!dproj || (cam && x_stride >= 4), "smplr_skin_bwd: dproj given without camera rows");
  int nblk = 0;
  int rc = launch_skin_bwd_partials(dverts, dproj, SegGrad{nullptr, nullptr, 0}, v_posed, lbs_weights, lbs_top4, A, cam,
                                    x_stride, B, V, vertex_sampling, dv_posed, reinterpret_cast<float *>(workspace),
                                    as_stream(stream), &nblk);
  if (rc) return rc;
  hipLaunchKernelGGL(skin_bwd_reduce_kernel, dim3(B), dim3(320), 0, as_stream(stream),
                     reinterpret_cast<const float *>(workspace), nblk, dA, dcam);
  SMPLR_LAUNCH_CHECK("smplr_skin_bwd(reduce)");
  return 0;
}

int smplr_project_fwd(const float *verts, const float *cam, int x_stride, int B, int V,
                      int vertex_sampling, float *proj, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && V > 0 && check_vs(V, vertex_sampling) && x_stride >= 4,
                "smplr_project_fwd: bad sizes B=%d V=%d vs=%d x_stride=%d", B, V, vertex_sampling, x_stride);
  if (B == 0) return 0;
  SMPLR_REQUIRE(verts && cam && proj, "smplr_project_fwd: null pointer");
  const int VP = (V + vertex_sampling - 1) / vertex_sampling;
  hipLaunchKernelGGL(project_fwd_kernel, dim3((VP + 255) / 256, B), dim3(256), 0, as_stream(stream), verts,
                     cam, x_stride, V, vertex_sampling, VP, proj);
  SMPLR_LAUNCH_CHECK("smplr_project_fwd");
  return 0;
}

int smplr_project_bwd(const float *dproj, const float *verts, const float *cam, int x_stride, int B, int V,
                      int vertex_sampling, float *dverts, float *dcam, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && V > 0 && check_vs(V, vertex_sampling) && x_stride >= 4,
                "smplr_project_bwd: bad sizes B=%d V=%d vs=%d x_stride=%d", B, V, vertex_sampling, x_stride);
  if (B == 0) return 0;
  SMPLR_REQUIRE(dproj && verts && cam && dverts && dcam, "smplr_project_bwd: null pointer");
  const int VP = (V + vertex_sampling - 1) / vertex_sampling;
  hipLaunchKernelGGL(project_bwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), dproj, verts, cam,
                     x_stride, V, vertex_sampling, VP, dverts, dcam);
  SMPLR_LAUNCH_CHECK("smplr_project_bwd");
  return 0;
}

}  // extern "C"

#ifdef SMPLR_TL
SMPLR_TL_EXPORT(skin, smplr::g_tl_skin, smplr::TL_SKIN_WG * (smplr::SKB_T / 64) * 32)
SMPLR_TL_EXPORT(skinrec, smplr::g_tl_skinrec, smplr::TL_SKINREC_WG * 4 * 32)
#endif
