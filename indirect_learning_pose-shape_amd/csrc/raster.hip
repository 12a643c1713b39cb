// K6/K7 soft rasterisers: 31-part segmentation and silhouette, forward and backward.
//
// Reference: keras_smpl/projects_to_seg.py:34-69 and keras_smpl/projects_to_silhouette.py:20-42.
// The reference materialises (N, W^2, n_p, 2) tiles per part and takes max_v exp(-m_v d_v);
// exp is monotone, so that is exp(-min_v m_v d_v): a masked nearest-vertex search.
//
// Segmentation forward = two kernels.
//  (1) seg_bin_kernel, one workgroup per mesh (optionally with compute_mask's z-buffer fused in
//      front, and the mesh's vertices staged in LDS): splits the part-major vertex list by reach.
//      In fp32 exp(-x) == 0 for x >= 104, so a vertex with mask m only matters within
//      104/m pixels.  m > 208 ("local": the invisible vertices, m = 500) reaches at most its
//      nearest pixel centre -> one (pixel, part, x, vertex) record, counting-sorted by pixel.
//      m <= 208 ("global": the visible vertices, m = 1) are compacted part-major, in table
//      order, each part padded to a multiple of 4 with +inf sentinels, as (u, v, m^2, vertex).
//      This drops the pair count from 2304 x 6879 to 2304 x (#visible ~ 570) per mesh without
//      changing a single fp32 result.
//  (2) raster_fwd_kernel: a lane owns one pixel; a workgroup = 256 consecutive pixels of one mesh
//      x 4 contiguous ranges of parts (16 waves), the ranges cut per mesh so that each holds a
//      quarter of its global records.  The records are copied to LDS once per block (field-major)
//      together with (v - row)^2 for the image rows the block touches, so a pair costs a subtract
//      and an fma on the packed fp32 pipe; eight records per step, the running minimum carried
//      through v_min3, the winning group re-evaluated once per (pixel, part) for the first arg-min;
//      score = exp(-sqrt(key)) into a pixel-major LDS tile.  After one barrier all waves merge the
//      pixels' local records (8 lanes per pixel, LDS atomic max on the score bits) and write the
//      NHWC outputs as whole 128-B / 64-B pixel rows.  Record lists too long for the tables use
//      the plain LDS copy, those too long for LDS scalar loads.
//
// Backward (seg_bwd_kernel + seg_bwd_merge_kernel): lanes = channels of a pixel exactly as the
// NHWC tensors lie in memory (coalesced; the 32 lanes of a pixel hit 31 different parts, hence
// different vertices); the score is recomputed from the arg-min record (no re-read of seg);
// runs of pixels that share an arg-min are summed in registers and reach the block's LDS slot
// accumulators (ds_add_f32) only at run boundaries; per-block slot sums are merged in a fixed
// order and scattered to the vertices with plain stores (no global atomics, no memset).
#include <hip/hip_ext.h>
#include <algorithm>
#include "common.h"

namespace smplr {
__host__ __device__ constexpr int goff_stride(int P) { return P + 2 + 32; }   // ints per mesh in `goff`
#ifndef SMPLR_BIN_Q0
#define SMPLR_BIN_Q0 3
#endif
constexpr int BIN_Q0 = SMPLR_BIN_Q0;   // seg_bin_kernel<.., SKIN>: vertices per thread whose operands are requested before the first barrier

constexpr int CH = SMPLR_CHUNK;      // 8: silhouette list padding
template <auto Kernel>
static int lds_attr(size_t lds);
constexpr int RT = 256;              // pixels (threads) per silhouette raster block
constexpr float X_ZERO = 104.0f;     // expf(-x) rounds to 0 in fp32 for x >= 104
constexpr float M_LOCAL = 208.0f;    // m > 208 => 104/m < 0.5 px: only the nearest pixel centre
constexpr int GP = 4;                // global-list group size (padding granule)
constexpr int BIN_T = 1024;
#ifdef SMPLR_TL
constexpr int TL_BIN_WG = 128, TL_SEGBWD_WG = 256;
__device__ unsigned g_tl_bin[TL_BIN_WG * (BIN_T / 64) * 32];
__device__ unsigned g_tl_segbwd[TL_SEGBWD_WG * 12 * 32];
constexpr int TL_RASTER_WG = 1152;
__device__ unsigned g_tl_raster[TL_RASTER_WG * 16 * 32];
constexpr int TL_SILHPX_WG = 256;
__device__ unsigned g_tl_silhpx[TL_SILHPX_WG * 16 * 32];
#endif
constexpr int IPT_MAX = 8;           // part-table slots per bin thread: K <= 8192

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_excl_scan(int val, int *s_wave /*[BIN_T/64]*/, int *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = val;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < BIN_T / 64; ++w) {
    const int t = s_wave[w];
    if (w < wave) base += t;
    tot += t;
  }
  *total = tot;
  return base + inc - val;
}

struct Slot {
  int cls;      // 0 skip, 1 global, 2 local
  int pos, pix;
  float u, v, m, x;
};

__device__ __forceinline__ Slot classify(float u, float v, float m, int pos, int W) {
  Slot s;
  s.pos = pos;
  s.u = u;
  s.v = v;
  s.m = m;
  s.cls = 1;
  s.pix = 0;
  s.x = 0.f;
  if (s.m > M_LOCAL) {
    s.cls = 0;
    const float c = rintf(s.u), r = rintf(s.v);
    if (c >= 0.0f && c <= (float)(W - 1) && r >= 0.0f && r <= (float)(W - 1)) {
      const float du = s.u - c, dv = s.v - r;
      // v_sqrt_f32 (1 ulp), as the pair loop and the backward compute it; the IEEE sequence was a fifth of this
      // kernel's per-slot instructions
      s.x = __builtin_amdgcn_sqrtf(fmaf(du, du, dv * dv) * (s.m * s.m));
      if (s.x < X_ZERO) {
        s.cls = 2;
        s.pix = (int)r * W + (int)c;
      }
    }
  }
  return s;
}

// rec[n] (S = Kpad + K slots of (u, v, m^2, vertex)): [0, goff[P]) the global list, part-major,
// padded per part; [goff[P], goff[P] + L) the local records in pixel order.  Saved for backward.
// scratch per mesh: goff[P+1] | lstart[npix+1] | lrec[K] uint2 (x bits, part)
// VIS = true fuses compute_mask (visibility.hip's kernel, same arithmetic) in front: the z-buffer
// over the vgrid x vgrid grid and the per-vertex flags live in LDS after the pixel counters, the
// mask is written out (it is an output of the decoder) and classification reads the flags.
// STAGE = true keeps every vertex' (u, v) in LDS as well (2 VP floats): the workgroup then makes
// ONE round trip to global memory - its vertices (coalesced) and its part-table slots, requested
// together at the top - and the per-slot gathers of classification become LDS reads.
// SKIN = true (with VIS and STAGE): the workgroup skins and projects its mesh's vertices itself (skin_fwd_kernel's
// arithmetic, common.h) from v_posed, the sparse weights and the joint matrices, writes verts and proj out and goes on
// with the values in registers: the skinning launch, its ramp and the re-read of proj go away.
template <bool VIS, bool STAGE, bool SKIN>
__global__ __launch_bounds__(BIN_T) void seg_bin_kernel(const float *__restrict__ proj,
                                                        float *__restrict__ mask,
                                                        const int *__restrict__ part_pos,
                                                        const int *__restrict__ part_off, int P, int K,
                                                        int VP, int W, int S, float4 *__restrict__ G,
                                                        int *__restrict__ goff, int *__restrict__ lstart,
                                                        uint2 *__restrict__ lrec, int vgrid, int ref_compat,
                                                        short *__restrict__ vslot, SkinIn sk) {
  // (16-B aligned: the 64-bit z-buffer keys behind the counters need 8, whatever the static LDS in front)
  extern __shared__ __attribute__((aligned(16))) int s_cnt[];   // npix | VIS: z-buffer keys, visible flags | STAGE: u[VP], v[VP]
  __shared__ int s_poff[33], s_gstart[33], s_gpad[33], s_wave[BIN_T / 64], s_gb[BIN_T];
  __shared__ int s_any_empty, s_nonunit;
  __shared__ float4 sAj[SKIN ? 72 : 1];
  static_assert(!SKIN || (VIS && STAGE), "the skinning form is built for the decoder's path only");
  const int n = blockIdx.x, tid = threadIdx.x;
  SMPLR_TL_WAVE(g_tl_bin, BIN_T / 64, n, TL_BIN_WG)
  const int npix = W * W;
  const float *pj = proj + (size_t)n * VP * 3;
  float *mk = mask + (size_t)n * VP;
  const int cells = VIS ? vgrid * vgrid : 0, words = VIS ? (VP + 31) / 32 : 0;
  unsigned long long *zbuf = reinterpret_cast<unsigned long long *>(s_cnt + ((npix + 1) & ~1));
  unsigned int *vis = reinterpret_cast<unsigned int *>(zbuf + cells);
  float *sU = reinterpret_cast<float *>(vis + words), *sV = sU + VP;
  // vertex -> record slot (for the backward's gather by vertex).  Round 4: written straight to global memory - -1
  // everywhere up front (coalesced, under the first requests), a record's slot by the placement (a 2-B store per
  // record, nothing waits for it; the barriers in between order the two stores to one address) - instead of a map
  // staged in 13.8 KB of LDS and copied out behind one more barrier: 2.8 k of the kernel's 48 k clocks.
  short *vsl = vslot ? vslot + (size_t)n * VP : nullptr;

  // ---- every global operand of the block, requested up front
  const int ipt = (K + BIN_T - 1) / BIN_T;      // <= IPT_MAX (checked by the launcher)
  const int k0 = tid * ipt, k1 = min(K, k0 + ipt);
  int pos[IPT_MAX];
  if (!SKIN) {                                  // (the skinning form asks after its vertices are done: registers)
#pragma unroll
    for (int j = 0; j < IPT_MAX; ++j) pos[j] = (j < ipt) ? part_pos[min(k0 + j, K - 1)] : 0;
  }
  constexpr int VPT = SKIN ? 7 : 8;             // vertices per thread and trip: 8192 per trip (skinning: 7168, one trip)
  float vu[VPT], vv[VPT], vz[VPT];
  float4 tw[SKIN ? VPT : 1], tj[SKIN ? VPT : 1], aj = {0.f, 0.f, 0.f, 0.f};
  float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
  if (SKIN) {                                   // (VP <= 7 BIN_T: one trip, checked by the launcher)
    aj = reinterpret_cast<const float4 *>(sk.A + (size_t)n * 288)[tid < 72 ? tid : 71];
    const float *c = sk.cam + (size_t)n * sk.x_stride;
    c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3];
    const float *vp = sk.v_posed + (size_t)n * VP * 3;
    // (the first BIN_Q0 vertices' operands here, the rest behind the barrier: the CU's address unit takes 7.5 k clocks
    // for all 35 requests of every thread, and the skinning that follows is bound by LDS reads - the later vertices'
    // requests are worked off under the first ones' skinning instead of in front of the barrier)
#pragma unroll
    for (int q = 0; q < BIN_Q0; ++q) {
      const int v = min(tid + q * BIN_T, VP - 1);
      const float4 *tp = reinterpret_cast<const float4 *>(sk.top4 + (size_t)v * 8);
      tw[q] = tp[0];
      tj[q] = tp[1];
      vu[q] = vp[v * 3 + 0];                    // the posed vertex for now
      vv[q] = vp[v * 3 + 1];
      vz[q] = vp[v * 3 + 2];
    }
  } else if (VIS || STAGE) {
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
      const int v = min(tid + q * BIN_T, VP - 1);
      vu[q] = pj[v * 3 + 0];
      vv[q] = pj[v * 3 + 1];
      vz[q] = VIS ? pj[v * 3 + 2] : 0.0f;
    }
  }
  if (tid <= P) s_poff[tid] = part_off[tid];
  if (tid == 0) { s_nonunit = 0; s_any_empty = 0; }
  for (int i = tid; i < npix; i += BIN_T) s_cnt[i] = 0;
  if (vslot && !SKIN)                                         // block-uniform
    for (int i = tid; i < VP; i += BIN_T) vsl[i] = -1;
  if (VIS) {
    for (int i = tid; i < cells; i += BIN_T) zbuf[i] = 0ull;
    for (int i = tid; i < words; i += BIN_T) vis[i] = 0u;
  }
  if (SKIN && tid < 72) sAj[tid] = aj;
  SMPLR_TL_STAMP(1);
  __syncthreads();
  SMPLR_TL_STAMP(2);
  if (SKIN) {
    const float *vp = sk.v_posed + (size_t)n * VP * 3;
#pragma unroll
    for (int q = BIN_Q0; q < VPT; ++q) {
      const int v = min(tid + q * BIN_T, VP - 1);
      const float4 *tp = reinterpret_cast<const float4 *>(sk.top4 + (size_t)v * 8);
      tw[q] = tp[0];
      tj[q] = tp[1];
      vu[q] = vp[v * 3 + 0];
      vv[q] = vp[v * 3 + 1];
      vz[q] = vp[v * 3 + 2];
    }
    if (vslot)                                                // block-uniform
      for (int i = tid; i < VP; i += BIN_T) vsl[i] = -1;
    __builtin_amdgcn_sched_barrier(0);
    float *vo = sk.verts + (size_t)n * VP * 3, *po = sk.proj + (size_t)n * VP * 3;
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
      const int v = tid + q * BIN_T;
      float T[12], X, Y, Z;
      skin_T_sparse(sAj, tw[q], tj[q], T);
      skin_apply(T, vu[q], vv[q], vz[q], X, Y, Z);
      vu[q] = project_u(X, c0, c2);
      vv[q] = project_u(Y, c1, c3);
      vz[q] = Z;
      if (v < VP) {                                // (either output may be NULL: block-uniform)
        if (sk.verts) { SMPLR_OUT_STORE(&vo[v * 3 + 0], X); SMPLR_OUT_STORE(&vo[v * 3 + 1], Y); SMPLR_OUT_STORE(&vo[v * 3 + 2], Z); }
        if (sk.proj) { SMPLR_OUT_STORE(&po[v * 3 + 0], vu[q]); SMPLR_OUT_STORE(&po[v * 3 + 1], vv[q]); SMPLR_OUT_STORE(&po[v * 3 + 2], Z); }
      }
      __builtin_amdgcn_sched_barrier(0);          // one vertex at a time: seven T matrices at once do not fit the registers
    }
#pragma unroll
    for (int j = 0; j < IPT_MAX; ++j) pos[j] = (j < ipt) ? part_pos[min(k0 + j, K - 1)] : 0;
  }
  if (VIS || STAGE) {
    const float fG = (float)vgrid;
    for (int base = 0; base < VP; base += VPT * BIN_T) {
      if (base > 0) {                           // VP > 8192: further trips (block-uniform)
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
          const int v = min(base + tid + q * BIN_T, VP - 1);
          vu[q] = pj[v * 3 + 0];
          vv[q] = pj[v * 3 + 1];
          vz[q] = VIS ? pj[v * 3 + 2] : 0.0f;
        }
      }
#pragma unroll
      for (int q = 0; q < VPT; ++q) {
        const int v = base + tid + q * BIN_T;
        if (v < VP) {
          if (STAGE) { sU[v] = vu[q]; sV[v] = vv[q]; }
          if (VIS) {
            const float pu = rintf(vu[q]);      // round half to even, like tf.round (compute_mask.py:22)
            const float pv = rintf(vv[q]);
            if (pu >= 0.0f && pu < fG && pv >= 0.0f && pv < fG) {
              const int cell = (int)pv * vgrid + (int)pu;
              const unsigned long long key = ((unsigned long long)orderable(vz[q]) << 32) |
                                             (unsigned long long)(0xFFFFFFFFu - (unsigned)v);
              atomicMax(&zbuf[cell], key);
            }
          }
        }
      }
    }
    SMPLR_TL_STAMP(3);
    __syncthreads();
    SMPLR_TL_STAMP(4);
  }
  bool vertex1 = false;                          // an empty cell makes vertex 1 visible (compute_mask.py:99)
  if (VIS) {
    int empty = 0;
    for (int i = tid; i < cells; i += BIN_T) {
      const unsigned long long key = zbuf[i];
      if (key == 0ull) {
        empty = 1;
      } else {
        const unsigned int v = 0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull);
        atomicOr(&vis[v >> 5], 1u << (v & 31));
      }
    }
    if (empty) s_any_empty = 1;   // benign same-value race
    SMPLR_TL_STAMP(5);
    __syncthreads();
    SMPLR_TL_STAMP(6);
    vertex1 = s_any_empty && ref_compat && VP > 1;
    if (mask)                                      // (NULL with SKIN when the caller does not want it: block-uniform)
      for (int v = tid; v < VP; v += BIN_T)
        SMPLR_OUT_STORE(&mk[v], (((vis[v >> 5] >> (v & 31)) & 1u) || (vertex1 && v == 1)) ? 1.0f : 500.0f);
  }
  SMPLR_TL_STAMP(7);
  float4 *Gn = G + (size_t)n * S;
  int *goffn = goff + (size_t)n * goff_stride(P);
  int *lstartn = lstart + (size_t)n * (npix + 1);
  uint2 *lrecn = lrec + (size_t)n * K;

  // pass 1: classify each of this thread's slots ONCE (results stay in registers for the later
  // passes), count
  Slot sl[IPT_MAX];
  int gcnt = 0;
  unsigned gbits = 0;                            // bit j: this thread's slot j is a global record
#pragma unroll
  for (int j = 0; j < IPT_MAX; ++j) {
    const int k = k0 + j;
    sl[j].cls = 0;
    if (j < ipt && k < k1) {
      const int ps = pos[j];
      const float m = VIS ? ((((vis[ps >> 5] >> (ps & 31)) & 1u) || (vertex1 && ps == 1)) ? 1.0f : 500.0f) : mk[ps];
      const float u = STAGE ? sU[ps] : pj[ps * 3], v = STAGE ? sV[ps] : pj[ps * 3 + 1];
      sl[j] = classify(u, v, m, ps, W);
    }
    if (sl[j].cls == 1) {
      ++gcnt;
      gbits |= 1u << j;
      if (sl[j].m != 1.0f) s_nonunit = 1;     // benign same-value race; read after the scans' barriers
    } else if (sl[j].cls == 2) {
      atomicAdd(&s_cnt[sl[j].pix], 1);
    }
  }
  // the part of this thread's first slot (largest p with poff[p] <= k0), for the placement
  int p0 = 0;
  if (k0 < k1) {
    int lo = 0, hi = P;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_poff[mid] <= k0) lo = mid; else hi = mid;
    }
    p0 = lo;
  }
  SMPLR_TL_STAMP(8);
  __syncthreads();                               // pixel counters complete
  SMPLR_TL_STAMP(9);
  // ONE block scan for both prefixes: global records per thread (low half) and local records per thread's
  // pixel range (high half); K <= 8192 keeps either total below 2^16
  const int ept = (npix + BIN_T - 1) / BIN_T;
  const int e0 = tid * ept, e1 = min(npix, e0 + ept);
  int loc = 0;
  for (int e = e0; e < e1; ++e) loc += s_cnt[e];
  int tot2;
  const int base2 = block_excl_scan(gcnt | (loc << 16), s_wave, &tot2);
  const int gbase = base2 & 0xffff, gtotal = tot2 & 0xffff, ltotal = tot2 >> 16;
  SMPLR_TL_STAMP(10);
  s_gb[tid] = gbase | (int)(gbits << 16);
  {
    int run = base2 >> 16;                       // counting sort offsets over pixels
    for (int e = e0; e < e1; ++e) {
      const int c = s_cnt[e];
      s_cnt[e] = run;            // becomes the placement cursor
      lstartn[e] = run;
      run += c;
    }
    if (tid == 0) lstartn[npix] = ltotal;
  }
  SMPLR_TL_STAMP(11);
  __syncthreads();
  SMPLR_TL_STAMP(12);
  if (tid < 128) {
    // global prefix at each part's first slot: the owning thread's base + its global flags below that slot
    // (empty parts share a slot; parts that start at K take the total); then the padded part offsets (P <= 31)
    const int l = tid & 63;
    int gs = gtotal;
    const int kk = s_poff[l <= P ? l : P];
    if (l < P && kk < K) {
      const int t = kk / ipt, j = kk - t * ipt;
      const int w = s_gb[t];
      gs = (w & 0xffff) + __popc(((unsigned)w >> 16) & ((1u << j) - 1u));
    }
    const int gnext = __shfl_down(gs, 1, 64);
    const int cnt = (l < P) ? (gnext - gs + GP - 1) / GP * GP : 0;
    if (tid < 64) {
      int inc = cnt;
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (tid >= o) inc += t;
      }
      if (tid <= P) {
        s_gstart[tid] = gs;
        s_gpad[tid] = inc - cnt;               // tid == P: cnt = 0, inc = total
        goffn[tid] = inc - cnt;
      }
    } else {
      // the block's second wave, beside the prefix: the parts in order of record count (largest first, ties by part
      // number) for the rasteriser, whose waves take them from this list as they become free (raster2_fwd_kernel)
      const int key = l < P ? ((cnt << 5) | (31 - l)) : -1;                    // distinct keys; cnt < 2^16
      int rank = 0;
#pragma unroll
      for (int q = 0; q < 31; ++q) rank += (__builtin_amdgcn_readlane(key, q) > key) ? 1 : 0;
      if (l < P) goffn[P + 2 + rank] = l;
    }
  }
  SMPLR_TL_STAMP(13);
  __syncthreads();
  SMPLR_TL_STAMP(14);
  // pass 3: placement
  {
    int p = p0;
    int run = gbase;
#pragma unroll
    for (int j = 0; j < IPT_MAX; ++j) {
      const int k = k0 + j;
      if (!(j < ipt && k < k1)) continue;
      while (k >= s_poff[p + 1]) ++p;
      const Slot s = sl[j];
      if (s.cls == 1) {
        const int slot = s_gpad[p] + (run - s_gstart[p]);
        Gn[slot] = make_float4(s.u, s.v, s.m * s.m, __int_as_float(s.pos));
        if (vslot) vsl[s.pos] = (short)slot;
        ++run;
      } else if (s.cls == 2) {
        const int dst = atomicAdd(&s_cnt[s.pix], 1);
        lrecn[dst] = make_uint2(__float_as_uint(s.x), (unsigned)p);
        Gn[s_gpad[P] + dst] = make_float4(s.u, s.v, s.m * s.m, __int_as_float(s.pos));
        if (vslot) vsl[s.pos] = (short)(s_gpad[P] + dst);
      }
    }
  }
  SMPLR_TL_STAMP(15);
  // header: used slots | 1 if some far-reaching record has a weight other than 1 (else the pair loop skips m^2) | length
  // of the far-reaching list, padded per part (what the rasteriser's table has to hold: smplr_seg_raster_plan)
  if (tid == 0) goffn[P + 1] = s_nonunit;
  if (tid == 0)
    Gn[S - 1] = make_float4(__int_as_float(s_gpad[P] + lstartn[npix]), __int_as_float(s_nonunit), __int_as_float(s_gpad[P]),
                            __int_as_float(-1));
  // sentinels in the padding
  if (tid < P) {
    const int cnt = s_gstart[tid + 1] - s_gstart[tid];
    for (int i = s_gpad[tid] + cnt; i < s_gpad[tid + 1]; ++i)
      Gn[i] = make_float4(INFINITY, INFINITY, 1.0f, __int_as_float(-1));
  }
  SMPLR_TL_STAMP(16);
  SMPLR_TL_STAMP(17);
  SMPLR_TL_STAMP(18);
}

__device__ __forceinline__ float pair_key(const float4 a, float fc, float fr) {
  const float du = a.x - fc, dv = a.y - fr;
  return fmaf(du, du, dv * dv) * a.z;
}

// exp(-x) for x >= 0 on the transcendental unit (v_exp_f32; rel. error ~ 1e-7 * (1 + x))
__device__ __forceinline__ float fast_exp_neg(float x) { return __expf(-x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

constexpr int NG = 4;            // waves per 64-pixel group, each walking a contiguous range of parts (8: 55.7 us, 4: 48 us)
constexpr int SLD = 33;          // score tile row stride (floats per pixel: 32 channels + 1, conflict-free by lane)
constexpr int ALD = 34;          // arg tile row stride (shorts per pixel: 17 dwords)
#ifndef SMPLR_RASTER_BT
#define SMPLR_RASTER_BT 1024
#endif
constexpr int RTS = SMPLR_RASTER_BT / NG;   // pixels per segmentation raster block
constexpr int WPT = RTS / 64;    // 64-pixel sub-tiles per block
constexpr int PART_COST = 16;    // fixed cost of a part in the balance, in records (exp, sqrt, winner re-scan; 4: +0.4 us)
constexpr int NREC = 1024;      // records of a mesh's global list that fit the block's LDS copy (per field)
// LDS arena of a block, in floats: u[NREC] | v[NREC] | m^2[NREC] | tables of (v - row)^2, one row of the table per
// image row the block touches; with unit weights the tables start over m^2 (never read then)
#ifdef SMPLR_NO_TBL
constexpr int ARENA = 3 * NREC;
#else
constexpr int ARENA = (SMPLR_RASTER_BT >= 1024) ? 7232 : 6912;
#endif

// One vertex against this lane's pixel: strict '<' keeps the first arg-min in list order.
#define SMPLR_PAIR(rec, slot)                                   \
  {                                                             \
    const float key_ = pair_key(rec, fc, fr);                   \
    const bool lt_ = key_ < best;                               \
    best = lt_ ? key_ : best;                                   \
    bslot = lt_ ? (slot) : bslot;                               \
  }

// Launder a wave-uniform index so the optimiser cannot fold a prefetch back into its use.
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+s"(v));
  return v;
}

// Scalar (wave-uniform) record loads issued from inline asm so that they can be double-buffered:
// SMEM returns out of order, so hipcc makes every use of a scalar load wait lgkmcnt(0), which
// also drains a prefetch issued in between.  Here a group of 4 records (64 B) is fetched with one
// s_load_dwordx16 while the previous group is evaluated, and the wait is placed by hand right
// before the new group's first use.  The compiler never touches a group between its load and
// its wait (the "+s" on the wait statement is the group's only way to its uses).
typedef float f32x16s __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void sload_group(f32x16s &dst, const float4 *p) {
  asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(dst) : "s"(p) : "memory");
  __builtin_amdgcn_sched_barrier(0);   // keep the other group's VALU work BELOW the prefetch
}
__device__ __forceinline__ void swait_group(f32x16s &v) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v) : : "memory");
}
#define SMPLR_GROUP(grp, k)                                                        \
  SMPLR_PAIR(make_float4(grp[0], grp[1], grp[2], grp[3]), (k))                     \
  SMPLR_PAIR(make_float4(grp[4], grp[5], grp[6], grp[7]), (k) + 1)                 \
  SMPLR_PAIR(make_float4(grp[8], grp[9], grp[10], grp[11]), (k) + 2)               \
  SMPLR_PAIR(make_float4(grp[12], grp[13], grp[14], grp[15]), (k) + 3)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Keys of two records against this lane's pixel (v_pk_add/mul/fma_f32).  Same roundings as
// pair_key: sub, sub, mul, fma, mul; UNIT drops the final multiply when every weight is 1 (x*1 = x).
template <bool UNIT>
__device__ __forceinline__ f32x2 pair_key2(f32x2 u, f32x2 v, f32x2 m2, f32x2 fc2, f32x2 fr2) {
  const f32x2 du = u - fc2, dv = v - fr2;
  const f32x2 t = dv * dv;
  const f32x2 d2 = __builtin_elementwise_fma(du, du, t);
  return UNIT ? d2 : d2 * m2;
}

// Records [beg, end) of the block's LDS copy (field-major: u | v | m^2, NREC floats each) against
// this lane's pixel, 4 records per step: only the group minimum is tracked (strict '<': the first
// minimal group wins); bav = byte offset of the winning group (the LDS address operand is a VGPR
// anyway, so it doubles as the tracked id).
#define SMPLR_LDS_GROUP(off, fld) \
  (*reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(base + (off) + (fld) * NREC * 4, 16)))
template <bool UNIT>
__device__ __forceinline__ float group_min(const char *base, unsigned av, f32x2 fc2, f32x2 fr2) {
  const f32x4 u = SMPLR_LDS_GROUP(av, 0), v = SMPLR_LDS_GROUP(av, 1);
  f32x4 m = {1.f, 1.f, 1.f, 1.f};
  if (!UNIT) m = SMPLR_LDS_GROUP(av, 2);
  const f32x2 k01 = pair_key2<UNIT>(u.xy, v.xy, m.xy, fc2, fr2);
  const f32x2 k23 = pair_key2<UNIT>(u.zw, v.zw, m.zw, fc2, fr2);
  return fminf(fminf(k01.x, k01.y), fminf(k23.x, k23.y));
}
template <bool UNIT>
__device__ __forceinline__ void lds_scan(const char *base, int beg, int end, f32x2 fc2, f32x2 fr2, float &best,
                                         unsigned &bav) {
  unsigned av = (unsigned)beg * 4u;
  asm volatile("" : "+v"(av));
  for (int k = beg; k < end; k += GP) {
    const float ma = group_min<UNIT>(base, av, fc2, fr2);
    const bool la = ma < best;
    best = la ? ma : best;
    bav = la ? av : bav;
    av += GP * 4;
  }
}

// The same scan with (v - row)^2 read from the block's row table instead of being recomputed per pixel:
// rowoff = byte offset of this lane's image row in the table.  Keys are bit-identical to pair_key2's
// (the table entry IS its t = dv * dv).
template <bool UNIT>
__device__ __forceinline__ void tbl_keys(const char *base, unsigned av, unsigned tv, f32x2 fc2, f32x2 &k01, f32x2 &k23) {
  const f32x4 u = SMPLR_LDS_GROUP(av, 0);
  const f32x4 t = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(base + tv, 16));
  const f32x2 du01 = u.xy - fc2, du23 = u.zw - fc2;
  k01 = __builtin_elementwise_fma(du01, du01, t.xy);
  k23 = __builtin_elementwise_fma(du23, du23, t.zw);
  if (!UNIT) {
    const f32x4 m = SMPLR_LDS_GROUP(av, 2);
    k01 = k01 * m.xy;
    k23 = k23 * m.zw;
  }
}
// Two groups per step (one address update each for the record and the table pointer); the running minimum is
// carried through v_min3 and a group is the new winner iff it lowered it (strict, so the first minimal group in
// list order wins, as in lds_scan).  The id kept for the first group of a step is its record offset av, for the
// second the step's table pointer tv (>= TBL_ID, no extra register or instruction): tbl_group() decodes both.
template <bool UNIT>
__device__ __forceinline__ void lds_scan_tbl(const char *base, int beg, int end, unsigned rowoff, f32x2 fc2,
                                             float &best, unsigned &bav) {
  unsigned av = (unsigned)beg * 4u, tv = rowoff + (unsigned)beg * 4u;
  asm volatile("" : "+v"(av));
  asm volatile("" : "+v"(tv));
  int k = beg;
  for (; k + 2 * GP <= end; k += 2 * GP) {
    f32x2 a01, a23, b01, b23;
    tbl_keys<UNIT>(base, av, tv, fc2, a01, a23);
    tbl_keys<UNIT>(base, av + GP * 4, tv + GP * 4, fc2, b01, b23);
    const float na = fminf(fminf(a01.x, a01.y), fminf(fminf(a23.x, a23.y), best));
#ifndef SMPLR_KO_TRACK
    bav = na < best ? av : bav;
#endif
    const float nb = fminf(fminf(b01.x, b01.y), fminf(fminf(b23.x, b23.y), na));
#ifndef SMPLR_KO_TRACK
    bav = nb < na ? tv : bav;
#endif
    best = nb;
    av += 2 * GP * 4;
    tv += 2 * GP * 4;
  }
  if (k < end) {
    f32x2 a01, a23;
    tbl_keys<UNIT>(base, av, tv, fc2, a01, a23);
    const float na = fminf(fminf(a01.x, a01.y), fminf(fminf(a23.x, a23.y), best));
    bav = na < best ? av : bav;
    best = na;
  }
}
// record offset (bytes) of the winning group from the id lds_scan_tbl kept
__device__ __forceinline__ unsigned tbl_group(unsigned id, unsigned rowoff) {
  return id >= NREC * 4u ? id - rowoff + GP * 4u : id;
}

// A wave's parts [ps, pe) in table mode, specialised by the weights so that no mode is tested per part; offv =
// the part offsets, one per lane.  A non-empty part always has a finite key (its pads come after real records),
// so the winner re-scan is unconditional and only the final selects look at best < inf.
template <bool UNIT>
__device__ __forceinline__ void scan_parts_tbl(const char *base, int offv, int ps, int pe, unsigned rowoff,
                                               f32x2 fc2, float *myS, short *myA) {
  int beg = __builtin_amdgcn_readlane(offv, ps);
  for (int p = ps; p < pe; ++p) {
    const int end = __builtin_amdgcn_readlane(offv, p + 1);
    float best = INFINITY;
    int bslot = -1;
    if (beg < end) {
      unsigned bav = (unsigned)beg * 4u;           // (no group lowers an infinite best: the first one is looked at)
#ifdef SMPLR_KO_PAIRS
      lds_scan_tbl<UNIT>(base, beg, beg + GP, rowoff, fc2, best, bav);
#else
      lds_scan_tbl<UNIT>(base, beg, end, rowoff, fc2, best, bav);
#endif
#if defined(SMPLR_KO_RESCAN) || defined(SMPLR_KO_TRACK)
      bslot = (int)(bav >> 2);
    }
    if (false) {
      unsigned bav = 0;
#endif
      // the winning group is looked at once more for the first record that attains the minimum
      const unsigned wav = tbl_group(bav, rowoff);
      f32x2 k01, k23;
      tbl_keys<UNIT>(base, wav, rowoff + wav, fc2, k01, k23);
      const int w23 = (k23.x == best) ? 2 : 3, w13 = (k01.y == best) ? 1 : w23;
      const int w = (int)(wav >> 2) + ((k01.x == best) ? 0 : w13);
      bslot = (best < INFINITY) ? w : -1;
    }
    // (no test for an empty part: sqrt(inf) = inf and v_exp_f32(-inf) = +0 exactly)
#ifdef SMPLR_KO_FINAL
    myS[p] = best;
#else
    myS[p] = fast_exp_neg(fast_sqrt(best));
#endif
    myA[p] = (short)bslot;
    beg = end;
  }
}

// Sum over each aligned group of 8 lanes, the same bits in all 8 (fixed tree: lane^1, lane^2, other quad).
__device__ __forceinline__ float sum8_dpp(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));    // quad_perm 1,0,3,2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));    // quad_perm 2,3,0,1
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));   // row_half_mirror
  return v;
}

__device__ __forceinline__ float max8_dpp(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false)));
  return v;
}

// Block = 256 pixels of one mesh x 4 part ranges = 16 waves: wave (g, w) evaluates the parts of range g
// for pixels [64w, 64w+64) of the tile.  What bounds this kernel at small batch is the per-wave
// dependent chain (LDS reads -> VALU -> exp -> LDS, part after part), so the parts are spread over 4
// waves instead of walked by one.  The 4 ranges are contiguous and cut so that each holds about a
// quarter of the mesh's visible records (the block waits for its slowest wave, and the records per
// fixed group of 8 channels differ several-fold: a torso facing the camera against a hidden arm).
// Scores meet in a pixel-major LDS tile; after one barrier all 16 waves write it out, and the
// background channel comes from the sum over the tile row (fixed tree, independent of the cuts).
// LOSS: the loss head's forward as the epilogue (model.py:119-120 Reshape + softmax, focal_loss.py:10-46 at an integer
// class map): a pixel's 32 raw scores sit in 8 adjacent lanes at write-out time, so its softmax denominator, the
// labelled class' probability and the per-pixel loss cost two 8-lane tree sums - and the (B, W, W, 32) score tensor
// need not be written at all (seg = NULL): the backward (seg_bwd_kernel<.., LOSS>) rebuilds d loss / d score of every
// channel from 16 bytes per pixel left here (`stats`, k = q_t softmax_t: k / sum exp(score) | k x the background's share
// (delta_0t - softmax_0 where the clip's gate is open, else 0) | k | label) instead of reading a 128-B row of dseg.
// Scores lie in [0, 1]: the softmax needs no max shift.
// vmax (optional, with or without the loss): per pixel the largest of its 31 part scores, as the output lies - for the
// silhouette rasteriser an upper bound of the distance to the nearest vertex (-log of it: a score is exp(-m d), m >= 1),
// which spares it its own search for one (smplr_silh_fwd_hint).
struct LossOut { const int *labels; const float *class_w; float gamma; float *loss; float4 *stats; float *vmax; };

// (amdgpu_num_sgpr: two blocks of 16 waves share a CU, 8 waves per SIMD, and that holds up to 80 scalar registers per
// wave only - 800 per SIMD, allotted in 16s, 16 more per wave for the trap handler the runtime installs - although the
// compiler's own table reports "Occupancy: 8" up to 102: a build of the LOSS variant with 83 ran ONE block per CU and
// took 46.6 us instead of 37.3 with fewer instructions (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES halved).)
#ifdef SMPLR_RASTER_NO_SGPR_CAP
#define SMPLR_RASTER_SGPRS
#else
#define SMPLR_RASTER_SGPRS __attribute__((amdgpu_num_sgpr(80)))
#endif
template <bool LOSS>
__global__ __launch_bounds__(RTS * NG) SMPLR_RASTER_SGPRS void raster_fwd_kernel(const float4 *__restrict__ G,
                                                             const int *__restrict__ goff,
                                                             const int *__restrict__ lstart,
                                                             const uint2 *__restrict__ lrec, int P, int K,
                                                             int S, int W, int B, int ntiles,
                                                             float *__restrict__ seg, short *__restrict__ arg,
                                                             unsigned wmagic, LossOut lo) {
  __shared__ float sS[RTS * SLD];
  __shared__ short sA[RTS * ALD];
  __shared__ f32x4 sRec[ARENA / 4];      // records, field-major: u[NREC] | v[NREC] | m^2[NREC]; row tables
  // XCD-aware map: mesh m lives on XCD m % 8 (blocks b and b+8 share an L2), its tiles are
  // consecutive there, so a mesh's record list is fetched into one L2 and re-read from it.
  const int bid = blockIdx.x;
  const int xcd = bid & 7, idx = bid >> 3;
  const int n = (idx / ntiles) * 8 + xcd, tile = idx % ntiles;
  if (n >= B) return;                                    // block-uniform
  const int tid = threadIdx.x, lane = tid & 63;
  SMPLR_TL_WAVE(g_tl_raster, 16, n * ntiles + tile, TL_RASTER_WG)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keep it scalar
  const int g = wave / WPT, pw = wave % WPT;             // part range, pixel sub-tile
  const int pt = pw * 64 + lane;                         // pixel within the tile
  const int npix = W * W;
  const int q = tile * RTS + pt;
  const int qc = q < npix ? q : npix - 1;
  // q / W for q < W^2 <= 25600 as a multiply and a shift (wmagic = ceil(2^24 / W), exact there): the
  // compiler's sequence for a division by a run-time W is ~20 instructions, three times per lane
  const int r = (int)(((unsigned)qc * wmagic) >> 24), c = qc - r * W;
  const float fc = (float)c, fr = (float)r;
  const float4 *Gn = G + (size_t)n * S;
  const int *goffn = goff + (size_t)n * goff_stride(P);
  const int C = P + 1;
  // everything the block needs from global memory is requested up front (one round trip): the
  // pixel's local-record range, the list length, the unit-weight flag, the part offsets (-> LDS)
  // (for the merge and write-out phase a pixel belongs to 8 adjacent lanes: item e = it * threads + tid is pixel
  // e / 8 of the tile, channels 4 (e % 8) ...)
  constexpr int NIT = 8 / NG;
  const int sub = tid & 7;
  int l0a[NIT], l1a[NIT];
  int lab[NIT];                                          // LOSS: the label of each of this lane's merge pixels
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int qq = tile * RTS + ((it * (RTS * NG) + tid) >> 3);
    const int *lp = lstart + (size_t)n * (npix + 1) + (qq < npix ? qq : npix - 1);
    l0a[it] = lp[0];
    l1a[it] = qq < npix ? lp[1] : 0;                     // pixels past the image merge nothing
    lab[it] = 0;
    if (LOSS) {                                          // labels lie as the output does: rows flipped
      const int qs = qq < npix ? qq : npix - 1;
      const int rr = (int)(((unsigned)qs * wmagic) >> 24), cc = qs - rr * W;
      lab[it] = lo.labels[(size_t)n * npix + (unsigned)((W - 1 - rr) * W + cc)];
    }
  }
  const uint2 *lrecn = lrec + (size_t)n * K;
  const int lbase = goffn[P];
  const bool unit_m = goffn[P + 1] == 0;                 // every far-reaching weight is 1 (block-uniform)
  __shared__ int sOff[40];
  if (tid <= P) sOff[tid] = goffn[tid];
  // the mesh's global record list (typically ~600 records) is copied to LDS once per block, one
  // array per field, and read by its 16 waves four records at a time with broadcast ds_read_b128
  // (in-order, counted waits); longer lists use the scalar-load path below.  Both evaluate the
  // same fp32 expressions.
  float *const frec = reinterpret_cast<float *>(sRec);
#pragma unroll
  for (int i = tid; i < NREC; i += RTS * NG) {
    // thread i copies record i before the list length is even known (slots beyond it hold stale
    // bytes nobody reads), so the copy shares the first round trip to memory
    const float4 t = Gn[min(i, S - 1)];
    frec[i] = t.x;
    frec[NREC + i] = t.y;
    frec[2 * NREC + i] = t.z;
  }
  const bool in_lds = lbase <= NREC;                     // block-uniform
  // (v - row)^2 of every record for the image rows this block touches (6 at W = 48), so that a pair costs
  // a subtract and an fma instead of two subtracts, a multiply and an fma; used when the tables fit
  const int row0 = (int)(((unsigned)min(tile * RTS, npix - 1) * wmagic) >> 24);
  const int nrows = (int)(((unsigned)min(tile * RTS + RTS - 1, npix - 1) * wmagic) >> 24) - row0 + 1;
  // table row stride: consecutive rows (the most a 16-lane read group spans) must not share banks
  const int lb4 = (lbase + 3) & ~3;
  const int RS = ((lb4 & 63) >= 4 && (lb4 & 63) <= 60) ? lb4 : lb4 + 4;
  const int toff = unit_m ? 2 * NREC : 3 * NREC;
#ifdef SMPLR_NO_TBL
  const bool tbl = false;
#else
  const bool tbl = in_lds && nrows * RS <= ARENA - toff;  // block-uniform
#endif
  SMPLR_TL_STAMP(1);
  __syncthreads();
  SMPLR_TL_STAMP(2);
  if (tbl) {
    // thread -> (group of 4 records k4 = tid % 256, rows tid / 256, + 4, ...): lbase <= NREC = 1 024 records
    const int n4 = (lbase + 3) >> 2, k4 = tid & 255;
    if (k4 < n4) {
      const f32x4 v = sRec[NREC / 4 + k4];
      for (int j = tid >> 8; j < nrows; j += (RTS * NG) >> 8) {
        const float frj = (float)(row0 + j);
        const f32x4 dv = v - frj;
        *reinterpret_cast<f32x4 *>(frec + toff + j * RS + 4 * k4) = dv * dv;
      }
    }
    __syncthreads();
  }
  SMPLR_TL_STAMP(3);
  const unsigned rowoff = (unsigned)(toff + (r - row0) * RS) * 4u;
  // the first 8 local records of each of this lane's merge pixels are fetched now and used after the pair loop
  uint2 lr0[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) lr0[it] = lrecn[min(l0a[it] + sub, K - 1)];
  float wlab[NIT];                                       // LOSS: the labelled class' weight (focal_loss.py:20-41)
#pragma unroll
  for (int it = 0; it < NIT; ++it)
    wlab[it] = (LOSS && lo.class_w) ? lo.class_w[min(max(lab[it], 0), 31)] : 1.0f;
  const f32x2 fc2 = {fc, fc}, fr2 = {fr, fr};
  float *myS = &sS[pt * SLD + 1];                        // indexed by part (channel = part + 1)
  short *myA = &sA[pt * ALD + 1];

  // this wave's parts [ps, pe): part p belongs to range g when the midpoint of its span in the cost
  // prefix c[p] = offset[p] + PART_COST p falls into the g-th quarter (the cuts are monotone and cover [0, P))
  int ps, pe;
  {
    const int lp = lane < P ? lane : 0;                  // P <= 31 parts
    const int mid2 = sOff[lp] + sOff[lp + 1] + PART_COST * (2 * lp + 1);     // 2 x midpoint
    const int total = __builtin_amdgcn_readfirstlane(sOff[P]) + PART_COST * P;
    const unsigned long long b0 = __ballot(lane < P && 2 * mid2 < total * g);
    const unsigned long long b1 = __ballot(lane < P && 2 * mid2 < total * (g + 1));
    ps = (g == 0) ? 0 : __popcll(b0);
    pe = (g == NG - 1) ? P : __popcll(b1);
  }
#ifdef SMPLR_TL
  const int ps0 = ps, pe0 = pe;
#endif

  {
    // the part offsets sit in a VGPR, one per lane (P + 1 <= 32), and a part's range is a v_readlane away
    // instead of an LDS round trip per part
    const int offv = sOff[lane <= P ? lane : P];
    const char *base = reinterpret_cast<const char *>(sRec);
    if (tbl) {                                             // block-uniform: the standard case
      if (unit_m) scan_parts_tbl<true>(base, offv, ps, pe, rowoff, fc2, myS, myA);
      else scan_parts_tbl<false>(base, offv, ps, pe, rowoff, fc2, myS, myA);
      pe = ps;                                             // nothing left for the generic loop
    }
    int beg = __builtin_amdgcn_readlane(offv, ps);
    for (int p = ps; p < pe; ++p) {
      const int end = __builtin_amdgcn_readlane(offv, p + 1);
      float best = INFINITY;
      int bslot = -1;
      if (in_lds) {
        if (beg < end) {
          unsigned bav = 0xffffffffu;
          if (unit_m) lds_scan<true>(base, beg, end, fc2, fr2, best, bav);
          else lds_scan<false>(base, beg, end, fc2, fr2, best, bav);
          // the winning group is looked at once more for the first record that attains the minimum
          if (bav != 0xffffffffu) {
            const f32x4 u = SMPLR_LDS_GROUP(bav, 0), v = SMPLR_LDS_GROUP(bav, 1);
            f32x4 m = {1.f, 1.f, 1.f, 1.f};                // x * 1 = x: the unit-weight scan's keys exactly
            if (!unit_m) m = SMPLR_LDS_GROUP(bav, 2);
            const f32x2 k01 = pair_key2<false>(u.xy, v.xy, m.xy, fc2, fr2);
            const f32x2 k23 = pair_key2<false>(u.zw, v.zw, m.zw, fc2, fr2);
            const int w23 = (k23.x == best) ? 2 : 3, w13 = (k01.y == best) ? 1 : w23;
            bslot = (int)(bav >> 2) + ((k01.x == best) ? 0 : w13);
          }
        }
      } else if (beg < end) {
        // two record groups in flight: group k+4 is being fetched while group k is evaluated
        f32x16s ga, gb;
        sload_group(ga, Gn + beg);
        swait_group(ga);
        int k = beg;
        while (true) {
          sload_group(gb, Gn + ((k + GP < end) ? k + GP : k));
          SMPLR_GROUP(ga, k)
          swait_group(gb);
          k += GP;
          if (k >= end) break;
          sload_group(ga, Gn + ((k + GP < end) ? k + GP : k));
          SMPLR_GROUP(gb, k)
          swait_group(ga);
          k += GP;
          if (k >= end) break;
        }
      }
      myS[p] = (best < INFINITY) ? fast_exp_neg(fast_sqrt(best)) : 0.0f;
      myA[p] = (short)bslot;
      beg = end;
    }
  }
  SMPLR_TL_STAMP(4);
  __syncthreads();
  SMPLR_TL_STAMP(5);
  // The tile now holds every part's best visible vertex.  All 16 waves merge the local records (invisible
  // vertices that round to the pixel) and write the tile out: 8 lanes per pixel, each taking every 8th record
  // of the pixel's list, then 4 channels of its row (coalesced 128-B / 64-B pixel rows).  A record replaces the
  // tile's score only if strictly larger (ties keep the earlier winner, global before local): an LDS atomic max
  // on the score bits (scores are >= 0, so the integer order is the float order) whose return value tells the
  // lane whether it raised the slot; the slot read back tells it whether a later lane of the same step raised
  // it further.  LDS operations of one wave execute in order, so no barrier separates merge and write-out.
  // LOSS: what the pixel of merge step `it` needs for its loss, in all 8 of its lanes; finished after the loop
  float den_[NIT], st_[NIT], eg_[NIT];
  unsigned po_[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = it * (RTS * NG) + tid;
    const int pl = e >> 3, c4 = (e & 7) * 4;
    {
      int *rowS = reinterpret_cast<int *>(&sS[pl * SLD + 1]);
      short *rowA = &sA[pl * ALD + 1];
      const int l1 = l1a[it];
      int i = l0a[it] + sub;
      uint2 rec = lr0[it];
#ifdef SMPLR_KO_MERGE
      while (false) {
#else
      while (__any(i < l1)) {
#endif
        const uint2 nxt = lrecn[min(i + 8, K - 1)];        // next step's record, in flight during this one
        if (i < l1) {
          const int sc = __float_as_int(fast_exp_neg(__uint_as_float(rec.x)));
          const int p = (int)rec.y;
          const int old = atomicMax(&rowS[p], sc);
          const int fin = __hip_atomic_load(&rowS[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // (a ds_read, not a flat load)
          if (old < sc && fin == sc) rowA[p] = (short)(lbase + i);
        }
        rec = nxt;
        i += 8;
      }
    }
    const float *ts = &sS[pl * SLD + c4];
    const short *ta = &sA[pl * ALD + c4];
    float v[4];
    short a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      v[t] = ts[t];
      a[t] = ta[t];
    }
    if (c4 == 0) v[0] = 0.0f;                              // the tile holds nothing for channel 0 ...
    if (C != 32) {                                         // ... nor for slots >= C (block-uniform: not the reference's 31 parts)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (c4 + t >= C) {
          v[t] = 0.0f;
          a[t] = (short)-1;
        }
      }
    }
    const float sum = sum8_dpp((v[0] + v[1]) + (v[2] + v[3]));   // over the pixel's parts (all lanes take part)
    float vmx = 0.0f;
    if (lo.vmax) vmx = max8_dpp(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));   // (block-uniform; channel 0 holds 0 here)
    if (c4 == 0) {
      v[0] = 1.0f - fminf(fmaxf(sum, 0.0f), 1.0f);         // background (:61-64)
      a[0] = (sum >= 0.0f && sum <= 1.0f) ? 1 : 0;         // clip pass-through gate
    }
    const int qq = tile * RTS + pl;
    if (LOSS) {                                            // (C == 32: checked by the launcher; all lanes take part)
      den_[it] = sum8_dpp((__expf(v[0]) + __expf(v[1])) + (__expf(v[2]) + __expf(v[3])));
      const int t = lab[it];
      const float vt = (t & 2) ? ((t & 1) ? v[3] : v[2]) : ((t & 1) ? v[1] : v[0]);
      st_[it] = sum8_dpp(c4 == (t & ~3) ? vt : 0.0f);      // the labelled class' score in all 8 lanes (+ exact zeros)
      // the background's exp (what it contributes to every channel's gradient) where the clip's gate is open, else a
      // negative number, from the pixel's lane 0 to its lanes 0 .. 3 (quad_perm 0,0,0,0)
      const float eg = a[0] ? __expf(v[0]) : -1.0f;
      eg_[it] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(eg), 0x00, 0xF, 0xF, false));
    }
    unsigned po = ~0u;
    if (qq < npix) {
      const int rr = (int)(((unsigned)qq * wmagic) >> 24), cc = qq - rr * W;
      po = (unsigned)((W - 1 - rr) * W + cc);              // rows flipped (:68); mesh base + 32-bit offset
    }
    if (LOSS) po_[it] = po;
    if (qq < npix && c4 < C) {
      if (lo.vmax && c4 == 0) lo.vmax[(size_t)n * npix + po] = vmx;
      float *so = seg + (size_t)n * npix * C + (po * (unsigned)C + (unsigned)c4);
      if (LOSS && !seg) {                                  // (block-uniform) the scores stay on the chip
      } else if (c4 + 3 < C && (C & 3) == 0) {
        SMPLR_OUT_STORE(reinterpret_cast<f32x4 *>(so), (f32x4{v[0], v[1], v[2], v[3]}));
      } else {
        for (int t = 0; t < 4; ++t)
          if (c4 + t < C) so[t] = v[t];
      }
      short4 o4;
      o4.x = a[0]; o4.y = a[1]; o4.z = a[2]; o4.w = a[3];
      *reinterpret_cast<short4 *>(arg + (size_t)n * npix * 32 + (po * 32u + (unsigned)c4)) = o4;
    }
  }
  if (LOSS) {
    // The per-pixel end of the loss (a dozen transcendental and clip steps) once for all merge steps of the lane: lane
    // `it` of a pixel's 8 finishes the pixel of step `it`, so the NIT pixels share one pass of the instructions
    // instead of running them NIT times in 8 lanes each.
    static_assert(NIT <= 4, "the background's lane reaches its quad only");
    float den = den_[0], st = st_[0], eg = eg_[0], w = wlab[0];
    int t = lab[0];
    unsigned po = po_[0];
#pragma unroll
    for (int it = 1; it < NIT; ++it) {
      if (sub == it) {
        den = den_[it]; st = st_[it]; eg = eg_[it]; w = wlab[it];
        t = lab[it];
        po = po_[it];
      }
    }
    // (v_rcp_f32 / v_log_f32: 1 ulp and ~1e-7 absolute in log2 on p in [1e-7, 1) - far inside the loss head's 1e-4
    // bar - where the IEEE division and logf() were a fifth of this phase's instructions; the raw instruction, not
    // __logf(): p >= 1e-7 is never denormal, and the library form spends 12 instructions on that case and on a
    // two-term product with ln 2)
    const float inv = __builtin_amdgcn_rcpf(den);
    const float sm = __expf(st) * inv;
    const float p = fminf(fmaxf(sm, K_EPS), 1.0f - K_EPS);                     // focal_loss.py:17
    const bool inside = sm >= K_EPS && sm <= 1.0f - K_EPS && (unsigned)t < 32u;  // (a label outside the classes: no loss)
    const float om = 1.0f - p, lg = __builtin_amdgcn_logf(p) * 0.6931471806f;
    const float pg = pow_gamma(om, lo.gamma);
    const float ls = (unsigned)t < 32u ? pg * ((-lg) * w) : 0.0f;              // :18, :41, :43-44
    // d loss / d softmax_t (the clip passes gradient on [eps, 1 - eps] only) x softmax_t: with it
    // d loss / d score_c = (q_t softmax_t) (delta_ct - softmax_c)
    const float k1 = inside ? (w * (dpow_gamma(om, lo.gamma) * lg - pg * __builtin_amdgcn_rcpf(p))) * sm : 0.0f;
    // what the background contributes to every channel's gradient where the clip's gate is open, per unit of k1
    const float gbu = eg >= 0.0f ? ((t == 0 ? 1.0f : 0.0f) - eg * inv) : 0.0f;
    if (sub < NIT && po != ~0u) {
      lo.loss[(size_t)n * npix + po] = ls;
      lo.stats[(size_t)n * npix + po] = make_float4(k1 * inv, k1 * gbu, k1, __int_as_float(t));
    }
  }
  SMPLR_TL_STAMP(6);
#ifdef SMPLR_TL
  if (tl__) { tl__[7] = (unsigned)(pe0 - ps0); tl__[8] = (unsigned)g; }
#endif
}

// ------------------------------------------------------------------------------------------------
// raster2_fwd_kernel: the same rasteriser with TWO pixels per lane (round 4).
//
// What the counters said about raster_fwd_kernel (profiles/raster_sq.json, tools/ab_kernel_b.sh knock-outs): of 99.6
// vector instructions per (64-pixel wave, part) the pair arithmetic was under a third; with all but one record group
// per part knocked out the kernel still took 23.6 of its 33.2 us.  The fixed cost per (wave, part) - part bookkeeping,
// two dependent LDS round trips, the winner re-scan, address arithmetic - is what two pixels per lane attack:
//  * a lane owns the pixels (2Q, c) and (2Q + 1, c): they share du = u - c, so a group of 4 records costs
//    2 v_pk_add + 4 v_pk_fma + 4 v_min3 + 2 (cmp, cndmask) = 14 vector instructions for 8 pairs (was 9 for 4), three
//    ds_read_b128 instead of four, and every per-part scalar / branch / setup instruction serves 128 pixels;
//  * the block's table is laid out [group of 4 records][row][4]: row 0 = u, row 1 + j = (v - (row0 + j))^2, so a
//    lane's two table rows are ADJACENT (one address register, immediate offsets 0 / 16) and the next group lies a
//    compile-time R * 16 bytes on: four groups per trip through the loop with immediate offsets only, two address
//    adds per 16 records instead of two per 8; the group id that is tracked is a scalar counter;
//  * the table is built straight from the global records (thread = record: u and R - 1 squares) behind ONE barrier
//    (was: copy, barrier, build, barrier);
//  * the winner re-scan decodes nothing (the tracked id IS the group index);
//  * the waves of a pixel slice share a list of the parts, largest first (seg_bin_kernel ranks them), and draw from
//    it as they finish (scan2_parts): the block's longest wave runs 1.11 x its mean instead of 1.32 x;
//  * two block shapes, PL pair-lanes x NG2 part ranges: 128 x 8 (two blocks per CU) and 64 x 10 (three) for batches
//    that would not fill two rounds of the large one (raster2_shape).
// Keys, tie rules, merge and write-out are raster_fwd_kernel's, expression for expression: outputs are bit-identical
// (tools/probes/seg_hash.py).  A list longer than the table (TREC(R) records) goes through it in chunks (round 5:
// scan2_parts_chunk below - rounds 1-4 walked such lists with scalar loads, 1.7 x slower at 3 000 records); blocks with a
// weight other than 1 in the list, or more than 10 image rows under them, still walk the global record list with scalar
// loads - exact, slow, and not what the reference's masks ({1, 500}) or sizes (48, 64) produce.
constexpr int PLN = 128;                 // pair-lanes per block: 2 x 64 (the large-batch shape)
// The block shape by batch: 1 = 128 pair-lanes x 8 part ranges (two blocks per CU), 2 = 64 x 10 (three per CU).  The
// small blocks cost a second table build per 256 pixels and pay while the large ones would leave CUs idle or
// half-filled: up to about two rounds of the large shape (512 blocks on 256 CUs at a time).
__host__ inline int raster2_shape(int B, int W, int K) {
  const long long nl = (long long)((W + 1) / 2) * W, blocks1 = (long long)B * ((nl + PLN - 1) / PLN);
  // (W = 48: up to 100 meshes; step A/B at B = 96: -1.2 %, at 112: +1.5 %.  W = 64, where a mesh has 60 % more records in
  // reach and 32 small blocks to build a table for: the small shape loses at every batch - B = 16: +6 %, 48: +12 %.
  // With every fifth vertex (K = 1 376, 400 records in reach) it loses as well - B = 96: +1.8 %, 32: +0.4 %: the full
  // part table only)
  return (W <= 48 && K >= 6000 && blocks1 <= 900) ? 2 : 1;
}
#ifndef SMPLR_R2_STATIC
#define SMPLR_R2_STATIC 1
#endif
constexpr int R2_STATIC = SMPLR_R2_STATIC;   // parts a wave takes by the static deal before it draws from the shared list (1..4)
constexpr int R2_MAX = 11;               // table rows + 1 of the largest instantiation
__host__ __device__ constexpr int trec_of(int AR, int R) { return (AR / (4 * R)) * 4; }   // records an arena of AR floats holds at R rows per group

template <int R>
__device__ __forceinline__ void scan2_body(const char *tb, unsigned vu, unsigned vt, int off, int gid, f32x2 fc2,
                                           float &bestA, float &bestB, int &gA, int &gB) {
  const f32x4 u = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(tb + vu + off, 16));
  const f32x4 t0 = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(tb + vt + off, 16));
  const f32x4 t1 = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(tb + vt + off + 16, 16));
  const f32x2 du01 = u.xy - fc2, du23 = u.zw - fc2;
  const f32x2 a01 = __builtin_elementwise_fma(du01, du01, t0.xy), a23 = __builtin_elementwise_fma(du23, du23, t0.zw);
  const f32x2 b01 = __builtin_elementwise_fma(du01, du01, t1.xy), b23 = __builtin_elementwise_fma(du23, du23, t1.zw);
  const float na = fminf(fminf(a01.x, a01.y), fminf(fminf(a23.x, a23.y), bestA));
  const float nb = fminf(fminf(b01.x, b01.y), fminf(fminf(b23.x, b23.y), bestB));
  // strict: the first group that attains the minimum keeps it.  (Written as "not less ? old : new" so that the
  // scalar group counter can be the select's SGPR operand - v_cndmask takes one only as its "false" value.)
  gA = !(na < bestA) ? gA : gid;
  gB = !(nb < bestB) ? gB : gid;
  bestA = na;
  bestB = nb;
}

// first record of group g (table row ra) whose key equals best -> its slot
template <int R>
__device__ __forceinline__ int rescan2(const char *tb, int g, unsigned va, f32x2 fc2, float best) {
  const unsigned wa = __umul24((unsigned)g, (unsigned)(R * 16));   // (v_mul_u32_u24: full rate; g < 2^24)
  const f32x4 u = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(tb + wa, 16));
  const f32x4 t = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(tb + wa + va, 16));
  const f32x2 du01 = u.xy - fc2, du23 = u.zw - fc2;
  const f32x2 k01 = __builtin_elementwise_fma(du01, du01, t.xy), k23 = __builtin_elementwise_fma(du23, du23, t.zw);
  const int w23 = (k23.x == best) ? 2 : 3, w13 = (k01.y == best) ? 1 : w23;
  const int w = g * 4 + ((k01.x == best) ? 0 : w13);
  return (best < INFINITY) ? w : -1;
}

// A wave's parts for its 64 pair-lanes, unit weights, table mode.  offv: the part offsets, one per lane; ordv: the parts
// by size, one per lane.  Wave g of NG takes the list entries g, 2 NG - 1 - g, ... (boustrophedon) for its first
// R2_STATIC parts and the rest from the counter the block's waves of this pixel half share (asked for before the part
// in hand is scanned: the answer is there when it is needed).
template <int R, int NG>
__device__ __forceinline__ int scan2_parts(const char *tb, int offv, int ordv, int *ctr, int g, int P, bool lane0,
                                           unsigned va, f32x2 fc2, float *myS, short *myA) {
  constexpr int GB = R * 16;             // bytes per group
  int done = 1, r = g;
#ifdef SMPLR_KO2_SCAN
  if (r < P) {
#else
  while (r < P) {
#endif
    // (the draw as a bare ds_add_rtn_u32 from lane 0: through __hip_atomic_fetch_add the compiler's wave-aggregation of
    // atomics - mbcnt, a second exec detour, a count, a broadcast - wrapped it in a dozen instructions and, worse, waited
    // for the answer on the spot; here nothing waits before the part in hand has been scanned.  LDS operations return in
    // order, so every wait the compiler places for its own reads covers this older one too.)
    int rn = 0;
    if (done >= R2_STATIC) {
      if (lane0) {
        const unsigned ca = (unsigned)(size_t)(__attribute__((address_space(3))) int *)ctr;
        asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(rn) : "v"(ca), "v"(1) : "memory");
      }
    }
    const int p = __builtin_amdgcn_readlane(ordv, r);
    const int beg = __builtin_amdgcn_readlane(offv, p), end = __builtin_amdgcn_readlane(offv, p + 1);
    float bestA = INFINITY, bestB = INFINITY;
    int sA_ = -1, sB_ = -1;
    if (beg < end) {
      int gid = beg >> 2;
      const int ngrp = (end - beg) >> 2;
      int gA = gid, gB = gid;
      unsigned vu = (unsigned)gid * GB, vt = vu + va;
      asm volatile("" : "+v"(vu));
      asm volatile("" : "+v"(vt));
      int g = 0;
      for (; g + 4 <= ngrp; g += 4) {
        scan2_body<R>(tb, vu, vt, 0, gid, fc2, bestA, bestB, gA, gB);
        scan2_body<R>(tb, vu, vt, GB, gid + 1, fc2, bestA, bestB, gA, gB);
        scan2_body<R>(tb, vu, vt, 2 * GB, gid + 2, fc2, bestA, bestB, gA, gB);
        scan2_body<R>(tb, vu, vt, 3 * GB, gid + 3, fc2, bestA, bestB, gA, gB);
        vu += 4 * GB;
        vt += 4 * GB;
        gid += 4;
      }
      if ((ngrp - g) & 2) {
        scan2_body<R>(tb, vu, vt, 0, gid, fc2, bestA, bestB, gA, gB);
        scan2_body<R>(tb, vu, vt, GB, gid + 1, fc2, bestA, bestB, gA, gB);
        vu += 2 * GB;
        vt += 2 * GB;
        gid += 2;
      }
      if ((ngrp - g) & 1) scan2_body<R>(tb, vu, vt, 0, gid, fc2, bestA, bestB, gA, gB);
      sA_ = rescan2<R>(tb, gA, va, fc2, bestA);
      sB_ = rescan2<R>(tb, gB, va + 16, fc2, bestB);
    }
    // (no test for an empty part: sqrt(inf) = inf and v_exp_f32(-inf) = +0 exactly)
    myS[p] = fast_exp_neg(fast_sqrt(bestA));
    myS[SLD + p] = fast_exp_neg(fast_sqrt(bestB));
    myA[p] = (short)sA_;
    myA[ALD + p] = (short)sB_;
    const int sn = (done & 1) ? (done + 1) * NG - 1 - g : done * NG + g;       // the wave's done-th entry of the static deal
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rn) : : "memory");               // (the draw has long returned)
    r = (done < R2_STATIC) ? sn : __builtin_amdgcn_readfirstlane(rn);
    ++done;
  }
  return done - 1;
}

// The same for a record list LONGER than the table (round 5): the list goes through the table in chunks of `trec`
// records, [c0, c1) this time, and a (pixel, part)'s running (smallest key, its record slot) waits in the block's score /
// arg tiles between chunks.  A part is STARTED by the chunk its first record lies in (the last chunk also starts the parts
// that begin at the very end of the list: the empty ones), CONTINUED by every later chunk it reaches into, and FINISHED -
// its key turned into the score - by the chunk its last record lies in.  Keys, the strict '<' across groups and the
// first-equal rule inside the winning group are scan2_parts': the winner over the chunks is the first record in list order
// that attains the minimum, bit for bit what one pass over a table of the whole list (or raster_fwd_kernel) gives.  The
// waves draw the parts as above in every chunk; parts the chunk does not touch cost a draw and two compares.
template <int R, int NG>
__device__ __forceinline__ int scan2_parts_chunk(const char *tb, int offv, int ordv, int *ctr, int g, int P, bool lane0,
                                                 unsigned va, f32x2 fc2, float *myS, short *myA, int c0, int c1,
                                                 bool last) {
  constexpr int GB = R * 16;
  int done = 1, r = g;
  while (r < P) {
    int rn = 0;
    if (done >= R2_STATIC) {
      if (lane0) {
        const unsigned ca = (unsigned)(size_t)(__attribute__((address_space(3))) int *)ctr;
        asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(rn) : "v"(ca), "v"(1) : "memory");
      }
    }
    const int p = __builtin_amdgcn_readlane(ordv, r);
    const int beg = __builtin_amdgcn_readlane(offv, p), end = __builtin_amdgcn_readlane(offv, p + 1);
    const bool start = beg >= c0 && (beg < c1 || last), cont = beg < c0 && end > c0;       // wave-uniform
    if (start || cont) {
      float bestA = INFINITY, bestB = INFINITY;
      int sA_ = -1, sB_ = -1;
      if (cont) {
        bestA = myS[p];
        bestB = myS[SLD + p];
        sA_ = myA[p];
        sB_ = myA[ALD + p];
      }
      const int b = max(beg, c0), e = min(end, c1);
      if (b < e) {
        int gid = (b - c0) >> 2;                             // group of the TABLE (chunk-relative)
        const int ngrp = (e - b) >> 2;
        int gA = -1, gB = -1;                                // no group of this chunk has lowered the minimum yet
        unsigned vu = (unsigned)gid * GB, vt = vu + va;
        asm volatile("" : "+v"(vu));
        asm volatile("" : "+v"(vt));
        int k = 0;
        for (; k + 4 <= ngrp; k += 4) {
          scan2_body<R>(tb, vu, vt, 0, gid, fc2, bestA, bestB, gA, gB);
          scan2_body<R>(tb, vu, vt, GB, gid + 1, fc2, bestA, bestB, gA, gB);
          scan2_body<R>(tb, vu, vt, 2 * GB, gid + 2, fc2, bestA, bestB, gA, gB);
          scan2_body<R>(tb, vu, vt, 3 * GB, gid + 3, fc2, bestA, bestB, gA, gB);
          vu += 4 * GB;
          vt += 4 * GB;
          gid += 4;
        }
        if ((ngrp - k) & 2) {
          scan2_body<R>(tb, vu, vt, 0, gid, fc2, bestA, bestB, gA, gB);
          scan2_body<R>(tb, vu, vt, GB, gid + 1, fc2, bestA, bestB, gA, gB);
          vu += 2 * GB;
          vt += 2 * GB;
          gid += 2;
        }
        if ((ngrp - k) & 1) scan2_body<R>(tb, vu, vt, 0, gid, fc2, bestA, bestB, gA, gB);
        // (a lane whose minimum this chunk did not lower keeps the slot it came with: the earlier record wins a tie)
        const int nA = rescan2<R>(tb, max(gA, 0), va, fc2, bestA), nB = rescan2<R>(tb, max(gB, 0), va + 16, fc2, bestB);
        sA_ = gA >= 0 ? nA + c0 : sA_;
        sB_ = gB >= 0 ? nB + c0 : sB_;
      }
      const bool fin = end <= c1;                            // (the last chunk ends at the list's end: always)
      myS[p] = fin ? fast_exp_neg(fast_sqrt(bestA)) : bestA;
      myS[SLD + p] = fin ? fast_exp_neg(fast_sqrt(bestB)) : bestB;
      myA[p] = (short)sA_;
      myA[ALD + p] = (short)sB_;
    }
    const int sn = (done & 1) ? (done + 1) * NG - 1 - g : done * NG + g;
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rn) : : "memory");
    r = (done < R2_STATIC) ? sn : __builtin_amdgcn_readfirstlane(rn);
    ++done;
  }
  return done - 1;
}

template <bool LOSS, int NG2, int PL>
__global__ __launch_bounds__(PL * NG2, (PL * NG2 > 512 ? 8 : 4)) SMPLR_RASTER_SGPRS void raster2_fwd_kernel(
    const float4 *__restrict__ G, const int *__restrict__ goff, const int *__restrict__ lstart,
    const uint2 *__restrict__ lrec, int P, int K, int S, int W, int B, int ntiles, float *__restrict__ seg,
    short *__restrict__ arg, unsigned wmagic, LossOut lo) {
  constexpr int NT = PL * NG2;           // threads
  constexpr int TS = 2 * PL;             // pixels of the block's tile
  constexpr int PW = PL / 64;            // waves per part range (64 pair-lanes each)
  __shared__ float sS[TS * SLD];
  __shared__ short sA[TS * ALD];
  constexpr int AR = PL == 64 ? 6144 : ARENA;            // floats of the table's arena (three 64-lane blocks per CU: 51 KB each)
  __shared__ f32x4 sTab[AR / 4];         // [group][row 0 = u | rows 1.. = (v - row)^2][4 records]
  __shared__ int sCtr[PW];               // next free entry of the part list, per 64 pair-lanes
  const int bid = blockIdx.x;
  const int xcd = bid & 7, idx = bid >> 3;
  const int n = (idx / ntiles) * 8 + xcd, tile = idx % ntiles;
  if (n >= B) return;                                    // block-uniform
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < PW) sCtr[tid] = R2_STATIC * NG2;
  SMPLR_TL_WAVE(g_tl_raster, 16, n * ntiles + tile, TL_RASTER_WG)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave / PW, pw = wave % PW;               // part range, 64-lane slice of the block
  const int npix = W * W;
  const int nq = (W + 1) >> 1, nl = nq * W;              // row pairs, pair-lanes of an image
  // this lane's pixel pair: pair-lane L = Q W + c -> pixels (2Q, c), (2Q + 1, c)
  const int Lb = pw * 64 + lane;
  const int L = min(tile * PL + Lb, nl - 1);
  const int Q = (int)(((unsigned)L * wmagic) >> 24), c = L - Q * W;     // L < W^2 <= 25600: exact (see raster_fwd_kernel)
  // rows under the block: pairs Qf .. Ql
  const int Qf = (int)(((unsigned)min(tile * PL, nl - 1) * wmagic) >> 24);
  const int Ql = (int)(((unsigned)min(tile * PL + PL - 1, nl - 1) * wmagic) >> 24);
  const int row0 = 2 * Qf, nrows = 2 * (Ql - Qf + 1);    // (an odd W's last pair has a phantom row W: built, never written out)
  const float4 *Gn = G + (size_t)n * S;
  const int *goffn = goff + (size_t)n * goff_stride(P);
  const int C = P + 1;
  // (the records and offsets first: the table build and the block's first barrier wait for them, the items' list
  // bounds are not needed before the write-out)
  const uint2 *lrecn = lrec + (size_t)n * K;
  const int lbase = goffn[P];
  const bool unit_m = goffn[P + 1] == 0;                 // every far-reaching weight is 1 (block-uniform)
  const int goffv = goffn[min(lane, P)];                 // the part offsets, one per lane, in every wave
  const int ordv = goffn[P + 2 + (lane & 31)];           // ... and the parts by size (seg_bin_kernel)
  // table rows per group at this block: the smallest instantiation that holds its image rows
  const int Rb = nrows <= 4 ? 5 : nrows <= 6 ? 7 : nrows <= 8 ? 9 : 11;
  const int trec = trec_of(AR, Rb);
  // thread i asks for record i (and i + NT ... while the arena could hold it) before the list length is known: the
  // records share the block's first round trip to memory; slots beyond the list hold stale bytes nobody reads
  constexpr int NH = (trec_of(AR, 5) + NT - 1) / NT;
  float4 rcs[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#ifdef SMPLR_KO2_RECORDS                                 // (knock-out builds, tools/ab_kernel_b.sh: wrong results on purpose)
    rcs[h] = make_float4((float)(tid & 31), (float)(tid >> 5), 1.f, 0.f);
#else
    rcs[h] = (h == 0 || tid + h * NT < trec) ? Gn[min(tid + h * NT, S - 1)] : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
  // block-uniform: the table form (unit weights, at most 10 image rows under the block); a list longer than the table
  // goes through it in chunks of trec records (scan2_parts_chunk)
  const bool tblm = unit_m && nrows <= R2_MAX - 1;
  const bool tbl = tblm && lbase <= trec, chunked = tblm && !tbl;
  // merge / write-out items: item e = it * NT + tid is tile pixel e / 4 (= 2 x pair-lane + row of the pair); its lane
  // sub4 = e % 4 takes the channel chunks sub4 and sub4 + 4 (channels 4 sub4 .. and 16 + 4 sub4 ..): four lanes per
  // pixel, ONE item per thread at 1 024 threads - the per-item fixed cost (pixel decode, list bounds, addresses) of
  // raster_fwd_kernel's 8-lanes-per-pixel form once per 8 channels instead of once per 4
  constexpr int NIT = (TS * 4 + NT - 1) / NT;
  const int sub = tid & 3;
  int l0a[NIT], l1a[NIT], qqa[NIT];
  int lab[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int pl = (it * NT + tid) >> 2;
    const int Li = tile * PL + (pl >> 1);
    const int Lc = min(Li, nl - 1);
    const int Qi = (int)(((unsigned)Lc * wmagic) >> 24), ci = Lc - Qi * W;
    const int ri = 2 * Qi + (pl & 1);
    const bool ok = Li < nl && ri < W && (TS * 4 % NT == 0 || pl < TS);       // (threads past the tile's items: none)
    const int qq = ok ? ri * W + ci : -1;                // the item's pixel (row-major, unflipped), -1: none
    qqa[it] = qq;
    const int *lp = lstart + (size_t)n * (npix + 1) + (ok ? qq : 0);
    l0a[it] = lp[0];
    l1a[it] = ok ? lp[1] : 0;                            // pixels past the image merge nothing
    lab[it] = 0;
    if (LOSS) {                                          // labels lie as the output does: rows flipped
      const int rr = ok ? ri : 0;
      lab[it] = lo.labels[(size_t)n * npix + (unsigned)((W - 1 - rr) * W + ci)];
    }
  }
  SMPLR_TL_STAMP(1);
  // the table of the `cnt` records in rcs (thread i: records i, i + NT, ...)
  auto build_table = [&](int cnt) {
    float *tab = reinterpret_cast<float *>(sTab);
    const float fr0 = (float)row0;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int i = tid + h * NT;
      const float4 rcd = rcs[h];
      if (i < cnt) {
        float *dst = tab + ((i >> 2) * Rb) * 4 + (i & 3);
        dst[0] = rcd.x;
#pragma unroll
        for (int j = 0; j < R2_MAX - 1; ++j) {
          if (j < Rb - 1) {                              // (Rb is block-uniform)
            const float dv = rcd.y - (fr0 + (float)j);
            dst[(1 + j) * 4] = dv * dv;
          }
        }
      }
    }
  };
  if (tblm) build_table(min(lbase, trec));
  SMPLR_TL_STAMP(2);
  __syncthreads();
  SMPLR_TL_STAMP(3);
  // the first 4 local records of each of this lane's merge pixels are fetched now and used after the pair loop
  uint2 lr0[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) lr0[it] = lrecn[min(l0a[it] + sub, K - 1)];
  float wlab[NIT];                                       // LOSS: the labelled class' weight (focal_loss.py:20-41)
#pragma unroll
  for (int it = 0; it < NIT; ++it)
    wlab[it] = (LOSS && lo.class_w) ? lo.class_w[min(max(lab[it], 0), 31)] : 1.0f;
  const float fc = (float)c;
  const f32x2 fc2 = {fc, fc};
  float *myS = &sS[(2 * Lb) * SLD + 1];                  // indexed by part (channel = part + 1); second pixel at + SLD
  short *myA = &sA[(2 * Lb) * ALD + 1];

  // Which parts this wave scans: the block's NG2 waves of a pixel half share a list of the parts, largest first
  // (seg_bin_kernel sorts them); wave g starts on entry g and takes the next free entry whenever it finishes one - list
  // scheduling, longest first.  (Round 3 cut the part list into NG2 contiguous runs of equal estimated cost: with 31
  // parts on 8 ranges the longest range ran 1.32 x the mean and the block waited for it at the barrier.)  Which wave
  // evaluates a (pixel, part) does not enter the result.
  int ndone = 0;
  (void)ndone;                                           // (timeline builds record it)
  {
    const int offv = goffv;
    const char *tb = reinterpret_cast<const char *>(sTab);
    const unsigned va = (unsigned)(1 + 2 * (Q - Qf)) * 16u;      // byte offset of the upper pixel's table row in a group
    if (tbl) {
      if (Rb == 5) ndone = scan2_parts<5, NG2>(tb, offv, ordv, &sCtr[pw], g, P, lane == 0, va, fc2, myS, myA);
      else if (Rb == 7) ndone = scan2_parts<7, NG2>(tb, offv, ordv, &sCtr[pw], g, P, lane == 0, va, fc2, myS, myA);
      else if (Rb == 9) ndone = scan2_parts<9, NG2>(tb, offv, ordv, &sCtr[pw], g, P, lane == 0, va, fc2, myS, myA);
      else ndone = scan2_parts<11, NG2>(tb, offv, ordv, &sCtr[pw], g, P, lane == 0, va, fc2, myS, myA);
    } else if (chunked) {
      for (int c0 = 0;;) {
        const int c1 = min(c0 + trec, lbase);
        const bool last = c1 == lbase;
        int nd;
        if (Rb == 5) nd = scan2_parts_chunk<5, NG2>(tb, offv, ordv, &sCtr[pw], g, P, lane == 0, va, fc2, myS, myA, c0, c1, last);
        else if (Rb == 7) nd = scan2_parts_chunk<7, NG2>(tb, offv, ordv, &sCtr[pw], g, P, lane == 0, va, fc2, myS, myA, c0, c1, last);
        else if (Rb == 9) nd = scan2_parts_chunk<9, NG2>(tb, offv, ordv, &sCtr[pw], g, P, lane == 0, va, fc2, myS, myA, c0, c1, last);
        else nd = scan2_parts_chunk<11, NG2>(tb, offv, ordv, &sCtr[pw], g, P, lane == 0, va, fc2, myS, myA, c0, c1, last);
        ndone += nd;
        if (last) break;
        c0 = c1;
        // the next chunk's records (asked for here, not before the scan: eight registers the scan would have to hold
        // took the kernel past its 64 and into scratch), under the wait for the block's slowest wave
#pragma unroll
        for (int h = 0; h < NH; ++h)
          rcs[h] = (h == 0 || tid + h * NT < trec) ? Gn[min(c0 + tid + h * NT, S - 1)] : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();                                     // every wave is done with the table and with the draw counter
        if (tid < PW) sCtr[tid] = R2_STATIC * NG2;
        build_table(min(lbase - c0, trec));
        __syncthreads();
      }
    } else {
      // (weights other than 1, or more than 10 image rows under the block)
      // the global record list by scalar loads, one record at a time (strict '<': the first arg-min in list order)
      const float fr0 = (float)(2 * Q), fr1 = fr0 + 1.0f;
      for (int r = g; r < P; r += NG2) {                   // (no balancing here)
        const int p = __builtin_amdgcn_readlane(ordv, r);
        const int beg = __builtin_amdgcn_readlane(offv, p), end = __builtin_amdgcn_readlane(offv, p + 1);
        float bestA = INFINITY, bestB = INFINITY;
        int sA_ = -1, sB_ = -1;
        for (int k = beg; k < end; ++k) {
          const float4 rcd = Gn[k];
          const float ka = pair_key(rcd, fc, fr0), kb = pair_key(rcd, fc, fr1);
          const bool la = ka < bestA, lb = kb < bestB;
          bestA = la ? ka : bestA;
          sA_ = la ? k : sA_;
          bestB = lb ? kb : bestB;
          sB_ = lb ? k : sB_;
        }
        myS[p] = (bestA < INFINITY) ? fast_exp_neg(fast_sqrt(bestA)) : 0.0f;
        myS[SLD + p] = (bestB < INFINITY) ? fast_exp_neg(fast_sqrt(bestB)) : 0.0f;
        myA[p] = (short)sA_;
        myA[ALD + p] = (short)sB_;
      }
    }
  }
  SMPLR_TL_STAMP(4);
  __syncthreads();
  SMPLR_TL_STAMP(5);
  // Merge of the local records and write-out.  raster_fwd_kernel's scheme (LDS atomic max on the score bits, ties keep
  // the earlier winner, global before local) with FOUR lanes per pixel: a lane takes every 4th record of the pixel's list,
  // then the channel chunks sub and sub + 4.  The background's sum keeps raster_fwd_kernel's tree bit for bit: chunk sums
  // s_j = (v0 + v1) + (v2 + v3); Qlo = (s0 + s1) + (s2 + s3), Qhi = (s4 + s5) + (s6 + s7) by two quad exchanges each
  // (there: the 8-lane tree's first two steps); sum = Qlo + Qhi (there: lane 0 + lane 7 of the half-mirror step, and
  // fp32 addition commutes).
  auto quad_sum = [](float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));    // quad_perm 1,0,3,2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));    // quad_perm 2,3,0,1
    return v;
  };
  auto quad_max = [](float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)));
    return v;
  };
  float den_[NIT], st_[NIT], eg_[NIT];
  unsigned po_[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = it * NT + tid;
    const int pl = (TS * 4 % NT == 0) ? (e >> 2) : min(e >> 2, TS - 1), cA = sub * 4, cB = 16 + sub * 4;
    {
      int *rowS = reinterpret_cast<int *>(&sS[pl * SLD + 1]);
      short *rowA = &sA[pl * ALD + 1];
      const int l1 = l1a[it];
      int i = l0a[it] + sub;
      uint2 rec = lr0[it];
#ifdef SMPLR_KO2_MERGE
      while (false) {
#else
      while (__any(i < l1)) {
#endif
        const uint2 nxt = lrecn[min(i + 4, K - 1)];        // next step's record, in flight during this one
        if (i < l1) {
          const int sc = __float_as_int(fast_exp_neg(__uint_as_float(rec.x)));
          const int p = (int)rec.y;
          const int old = atomicMax(&rowS[p], sc);
          const int fin = __hip_atomic_load(&rowS[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // (a ds_read, not a flat load)
          if (old < sc && fin == sc) rowA[p] = (short)(lbase + i);
        }
        rec = nxt;
        i += 4;
      }
    }
    const float *ts = &sS[pl * SLD];
    const short *ta = &sA[pl * ALD];
    float va[4], vb[4];
    short aa[4], ab[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      va[t] = ts[cA + t];
      vb[t] = ts[cB + t];
      aa[t] = ta[cA + t];
      ab[t] = ta[cB + t];
    }
    if (sub == 0) va[0] = 0.0f;                            // the tile holds nothing for channel 0 ...
    if (C != 32) {                                         // ... nor for slots >= C (block-uniform: not the reference's 31 parts)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (cA + t >= C) { va[t] = 0.0f; aa[t] = (short)-1; }
        if (cB + t >= C) { vb[t] = 0.0f; ab[t] = (short)-1; }
      }
    }
    const float sum = quad_sum((va[0] + va[1]) + (va[2] + va[3])) + quad_sum((vb[0] + vb[1]) + (vb[2] + vb[3]));
    float vmx = 0.0f;
    if (lo.vmax)                                           // (block-uniform; channel 0 holds 0 here)
      vmx = quad_max(fmaxf(fmaxf(fmaxf(va[0], va[1]), fmaxf(va[2], va[3])), fmaxf(fmaxf(vb[0], vb[1]), fmaxf(vb[2], vb[3]))));
    if (sub == 0) {
      va[0] = 1.0f - fminf(fmaxf(sum, 0.0f), 1.0f);        // background (:61-64)
      aa[0] = (sum >= 0.0f && sum <= 1.0f) ? 1 : 0;        // clip pass-through gate
    }
    const int qq = qqa[it];
    if (LOSS) {                                            // (C == 32: checked by the launcher; all lanes take part)
      den_[it] = quad_sum((__expf(va[0]) + __expf(va[1])) + (__expf(va[2]) + __expf(va[3]))) +
                 quad_sum((__expf(vb[0]) + __expf(vb[1])) + (__expf(vb[2]) + __expf(vb[3])));
      const int t = lab[it];
      const float vta = (t & 2) ? ((t & 1) ? va[3] : va[2]) : ((t & 1) ? va[1] : va[0]);
      const float vtb = (t & 2) ? ((t & 1) ? vb[3] : vb[2]) : ((t & 1) ? vb[1] : vb[0]);
      // the labelled class' score in all 4 lanes (+ exact zeros)
      st_[it] = quad_sum(cA == (t & ~3) ? vta : 0.0f) + quad_sum(cB == (t & ~3) ? vtb : 0.0f);
      // the background's exp where the clip's gate is open, else a negative number, from the pixel's lane 0 to all 4
      const float eg = aa[0] ? __expf(va[0]) : -1.0f;
      eg_[it] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(eg), 0x00, 0xF, 0xF, false));
    }
    unsigned po = ~0u;
    if (qq >= 0) {
      const int rr = (int)(((unsigned)qq * wmagic) >> 24), cc = qq - rr * W;
      po = (unsigned)((W - 1 - rr) * W + cc);              // rows flipped (:68); mesh base + 32-bit offset
    }
    if (LOSS) po_[it] = po;
#ifdef SMPLR_KO2_WRITE
    if (qq >= 0 && sum == 12345.678f) {
#else
    if (qq >= 0) {
#endif
      if (lo.vmax && sub == 0) lo.vmax[(size_t)n * npix + po] = vmx;
      float *so = seg + (size_t)n * npix * C + po * (unsigned)C;
      if (LOSS && !seg) {                                  // (block-uniform) the scores stay on the chip
      } else if (C == 32) {
        SMPLR_OUT_STORE(reinterpret_cast<f32x4 *>(so + cA), (f32x4{va[0], va[1], va[2], va[3]}));
        SMPLR_OUT_STORE(reinterpret_cast<f32x4 *>(so + cB), (f32x4{vb[0], vb[1], vb[2], vb[3]}));
      } else {
        for (int t = 0; t < 4; ++t) {
          if (cA + t < C) so[cA + t] = va[t];
          if (cB + t < C) so[cB + t] = vb[t];
        }
      }
      short *ao = arg + (size_t)n * npix * 32 + po * 32u;
      short4 o4;
      o4.x = aa[0]; o4.y = aa[1]; o4.z = aa[2]; o4.w = aa[3];
      *reinterpret_cast<short4 *>(ao + cA) = o4;
      o4.x = ab[0]; o4.y = ab[1]; o4.z = ab[2]; o4.w = ab[3];
      *reinterpret_cast<short4 *>(ao + cB) = o4;
    }
  }
  if (LOSS) {
    // the per-pixel end of the loss once for all merge steps of the lane: lane `it` of a pixel's 4 finishes step `it`
    static_assert(!LOSS || NIT <= 4, "a pixel has four lanes");
    float den = den_[0], st = st_[0], eg = eg_[0], w = wlab[0];
    int t = lab[0];
    unsigned po = po_[0];
#pragma unroll
    for (int it = 1; it < NIT; ++it) {
      if (sub == it) {
        den = den_[it]; st = st_[it]; eg = eg_[it]; w = wlab[it];
        t = lab[it];
        po = po_[it];
      }
    }
    const float inv = __builtin_amdgcn_rcpf(den);
    const float sm = __expf(st) * inv;
    const float p = fminf(fmaxf(sm, K_EPS), 1.0f - K_EPS);                     // focal_loss.py:17
    const bool inside = sm >= K_EPS && sm <= 1.0f - K_EPS && (unsigned)t < 32u;  // (a label outside the classes: no loss)
    const float om = 1.0f - p, lg = __builtin_amdgcn_logf(p) * 0.6931471806f;
    const float pg = pow_gamma(om, lo.gamma);
    const float ls = (unsigned)t < 32u ? pg * ((-lg) * w) : 0.0f;              // :18, :41, :43-44
    const float k1 = inside ? (w * (dpow_gamma(om, lo.gamma) * lg - pg * __builtin_amdgcn_rcpf(p))) * sm : 0.0f;
    const float gbu = eg >= 0.0f ? ((t == 0 ? 1.0f : 0.0f) - eg * inv) : 0.0f;
    if (sub < NIT && po != ~0u) {
      lo.loss[(size_t)n * npix + po] = ls;
      lo.stats[(size_t)n * npix + po] = make_float4(k1 * inv, k1 * gbu, k1, __int_as_float(t));
    }
  }
  SMPLR_TL_STAMP(6);
#ifdef SMPLR_TL
  if (tl__) { tl__[7] = (unsigned)ndone; tl__[8] = (unsigned)g; }
#endif
}

// ------------------------------------------------------------------------------------------------
// Segmentation backward.  grid (ceil(W/rows), B), rows = 8 or 24 strips x 32 channels (block b of a mesh takes rows
// b, b + nblocks, ...): a 32-lane
// group walks one output row at a time, lane = channel, so dseg/arg are read as whole 128-B /
// 64-B pixel rows (coalesced) and neighbouring lanes hit different parts.  Along a row the
// arg-min of a part changes rarely, so each lane sums the run of pixels that share a slot in
// registers and only touches the LDS accumulator (ds_add_f32 runs at ~1 lane/clk) at run
// boundaries.  The arg-min record comes from the mesh's compact list (a few KB: L1-resident) as
// one 16-B gather; the score is recomputed from it (seg is not re-read).
// Per-block slot sums go to a partial buffer with plain coalesced stores and are summed in fixed
// order by seg_bwd_merge_kernel, which scatters each slot to its vertex (one slot per vertex, so
// plain stores; the blocks of a mesh zero its dproj rows first, so there is no memset and no
// global atomic).  A mesh with more than SB_SLOTS records (only when most vertices are marked
// visible) is walked once per window of SB_SLOTS slots.  Run-to-run differences are confined to
// the order in which a block's strips reach a slot's LDS accumulator (last-ulp rounding).
// (SB_SLOTS = 4096 accumulators per window, SB_NWIN = 5 windows, 8 or 24 rows per block by batch size: common.h)
constexpr int SB_U = 8;          // pixels in flight per lane
constexpr int SB_PF = 12;        // pixels of a row requested at kernel entry (>= SB_U)

// (unconditional: a run that ends has a non-zero sum except by cancellation, and the walk starts on slot 0 with a sum
// of zero, so the tests that used to guard this - slot valid, sum non-zero - only cost their instructions, in a kernel
// whose SIMDs are 88 % busy issuing)
// (cur >= 0 by construction for finite cotangents; a NaN / inf in dseg makes kk of a masked pixel (slot -1) a NaN,
// which passes `kk != 0`: the max keeps that garbage sum inside the accumulators instead of in front of them)
__device__ __forceinline__ void seg_flush(float *acc, int cur, float sx, float sy) {
  cur = max(cur, 0);
  atomicAdd(&acc[cur * 2], sx);
  atomicAdd(&acc[cur * 2 + 1], sy);
}

// Deterministic form: the run sums are added as 64-bit fixed-point integers (ds_add_u64).  Integer addition is
// associative, so the slot sums do not depend on the order in which the block's strips reach an accumulator - bit
// for bit the same result on every launch - and `scale` (a power of two chosen per block from max|dseg| and the
// largest record weight, see seg_bwd_kernel) keeps 2^-41 of the largest possible term as the resolution, far
// below an fp32 sum's own rounding.
__device__ __forceinline__ void seg_flush_det(unsigned long long *acc, int cur, float sx, float sy, float scale) {
  cur = max(cur, 0);
  atomicAdd(&acc[cur * 2], (unsigned long long)__float2ll_rn(sx * scale));
  atomicAdd(&acc[cur * 2 + 1], (unsigned long long)__float2ll_rn(sy * scale));
}

// One row strip (a 32-lane group, lane = channel) over its W pixels for one slot window.  MW =
// false is the standard single-window case (no window test per pixel).  The arithmetic is
// branch-free (a masked pixel contributes kk = 0); the only divergent step is the run boundary.
// FAST: C == 32 and W a multiple of SB_U (the reference's sizes): no clamping, and a pixel's dseg / arg elements
// sit at compile-time byte offsets (128 B / 64 B per pixel) from ONE address per lane and batch - the general
// form spent 29 % of the kernel's vector instructions on 64-bit address arithmetic, in a kernel that is
// vector-issue bound.
#ifdef SMPLR_TL
#define SMPLR_TL_ROW SMPLR_TL_PTR(g_tl_segbwd, 12, blockIdx.y * gridDim.x + blockIdx.x, (W <= 80 ? TL_SEGBWD_WG : 0))
#else
#define SMPLR_TL_ROW
#endif
template <bool MW, bool FAST, bool DET>
__device__ __forceinline__ void seg_bwd_row(const float *__restrict__ dseg, const short *__restrict__ arg,
                                            const float4 *__restrict__ R, int rbytes, float *acc, size_t row0,
                                            int W, int C, int ch, float fr, int base, float scale,
                                            const int *pa, const float *pg) {
  unsigned long long *acc64 = reinterpret_cast<unsigned long long *>(acc);
  const int chc = FAST ? ch : min(ch, C - 1);
  const bool chok = ch >= 1 && ch < C;
  const __amdgpu_buffer_rsrc_t rrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(R), 0, rbytes, 0x00020000);
  const short *arow = arg + row0 * 32 + ch;                    // FAST: + 32 (c0 + u) shorts
  const float *grow = dseg + row0 * 32 + ch;                   // FAST (C == 32): + 32 (c0 + u) floats
  int cur = 0;                                   // (slot 0 with a sum of zero: the first flush adds nothing)
  float sx = 0.0f, sy = 0.0f;
  SMPLR_TL_ROW
  for (int c0 = 0; c0 < W; c0 += SB_U) {
    int a[SB_U];
    float g[SB_U];
    if (FAST) {
      const short *ab = arow + c0 * 32;
      const float *gb = grow + c0 * 32;
#pragma unroll
      for (int u = 0; u < SB_U; ++u) {
        if (c0 == 0) {                            // (uniform) the first batch was requested at kernel entry
          a[u] = pa[u];
          g[u] = pg[u];
        } else {
          a[u] = ab[u * 32];
          g[u] = gb[u * 32];
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < SB_U; ++u) {
        const int cc = (c0 + u < W) ? c0 + u : W - 1;
        const size_t po = row0 + cc;
        a[u] = arg[po * 32 + ch];
        g[u] = dseg[po * C + chc];                // unconditional load (slots >= C are masked below)
      }
    }
    float4 rv[SB_U];
#pragma unroll
    for (int u = 0; u < SB_U; ++u) {
      // the channel-0 lane of this pixel holds the clip's gate (1 = the background's gradient passes) and that
      // gradient: what every channel subtracts is selected there and broadcast once
      g[u] = g[u] - __shfl((a[u] == 1) ? g[u] : 0.0f, 0, 32);
      if (!FAST && !(chok && c0 + u < W)) a[u] = -1;
      if (MW) {
        a[u] -= base;                             // another window's slot -> masked
        if (a[u] >= SB_SLOTS) a[u] = -1;
      }
      // the 16-B record of the arg-min slot as a buffer load: 32-bit offset, and slot -1 (masked) falls outside
      // the descriptor's range and reads as zeros (kk = 0 below) - no clamp, no 64-bit address per gather
      rv[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrs, (base + a[u]) * 16, 0, 0));
    }
    SMPLR_TL_STAMP(3 + c0 / SB_U * 2);           // (batches 0..9: words 3..22)
    if (FAST && !chok) continue;                  // channel 0 has no part (its lanes have served the broadcast above)
#pragma unroll
    for (int u = 0; u < SB_U; ++u) {
      const float fc = (float)(c0 + u);
      const float du = rv[u].x - fc, dv = rv[u].y - fr;
      const float d2 = fmaf(du, du, dv * dv);
      // d score / d(u,v) = -score m (p - q) / d, score = exp(-m d).  With t = (m d)^2 and r = 1 / sqrt(t): m d = t r
      // and m / d = m^2 r - two transcendental instructions per pixel (v_rsq, v_exp) instead of three (v_sqrt, v_exp,
      // v_rcp): they issue at a quarter of the rate and were a third of the vector time of this vector-bound loop.
      // d = 0: t is lifted to 1e-37, kk is large but finite and multiplies du = dv = 0: the gradient is 0, not NaN.
      const float t = d2 * rv[u].z;
      const float r = __builtin_amdgcn_rsqf(fmaxf(t, 1e-37f));
      float kk = (-g[u] * fast_exp_neg(t * r)) * (rv[u].z * r);
      // kk != 0 says it all: a masked slot (-1) read a record of zeros (m^2 = 0); exp underflows beyond 104
      // (with slot windows a slot below the window is a valid record of another window: tested there)
      if (MW && a[u] < 0) kk = 0.0f;
      const bool on = kk != 0.0f;
      if (on && a[u] != cur) {
        if (DET) seg_flush_det(acc64, cur, sx, sy, scale);
        else seg_flush(acc, cur, sx, sy);
        cur = a[u];
        sx = 0.0f;
        sy = 0.0f;
      }
      sx = fmaf(kk, du, sx);
      sy = fmaf(kk, dv, sy);
    }
    SMPLR_TL_STAMP(4 + c0 / SB_U * 2);
  }
  if (DET) seg_flush_det(acc64, cur, sx, sy, scale);
  else seg_flush(acc, cur, sx, sy);
}

// The FAST row walk (C == 32, W = 8 NB) as a software pipeline over its NB batches of SB_U pixels (round 4).  The
// in-kernel stamps of round 2 (profiles/r02_timelines.txt) show a batch as two halves of equal length: ~2.0 k clocks
// of memory latency (its arg / dseg rows, then the DEPENDENT gather of the arg-min records) with the vector unit idle,
// and ~2.2 k clocks of vector work (score, run sums, flushes) with nothing in flight - at three waves per SIMD neither
// half hides the other.  Here batch b + 2's rows are requested and batch b + 1's records gathered BEFORE batch b is
// summed: a wave's walk is as long as its vector work alone.  Fully unrolled (W = 48: 8 batches of 6 pixels, W = 64: 16 of 4 - the registers of three batches in flight
// beside the walk's own 59 must stay within the 168 a 768-thread block allows), so the three
// batches in flight are register names, not copies.  Same arithmetic, same order of flushes as seg_bwd_row.
typedef float f32x3g __attribute__((ext_vector_type(3)));
template <int U, int NB, bool DET>
__device__ __forceinline__ void seg_bwd_row_pipe(const float *__restrict__ dseg, const short *__restrict__ arg,
                                                 const float4 *__restrict__ R, int rbytes, float *acc, size_t row0,
                                                 int ch, float fr, float scale, const int *pa, const float *pg) {
  unsigned long long *acc64 = reinterpret_cast<unsigned long long *>(acc);
  const bool chok = ch >= 1;
  const __amdgpu_buffer_rsrc_t rrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(R), 0, rbytes, 0x00020000);
  const short *arow = arg + row0 * 32 + ch;
  const float *grow = dseg + row0 * 32 + ch;
  int a[NB][U];
  float g[NB][U];
  // (u, v, m^2): 12 of a record's 16 bytes - the vertex id is not used here (round 5: B = 2 048 285 -> 282 us)
  f32x3g rv[NB][U];
  int cur = 0;                                   // (slot 0 with a sum of zero: the first flush adds nothing)
  float sx = 0.0f, sy = 0.0f;
#ifdef SMPLR_TL
  constexpr int W = U * NB;                      // (the stamp macro's window test)
#endif
  SMPLR_TL_ROW
  // rows of batch b_ (arg-min slots + cotangents)
  // (the batch offset passes through an empty asm with a memory clobber: the loads are speculatable, and unrolled the
  // compiler otherwise hoists EVERY batch's loads to the top of the function - 185 spilled registers)
#define SMPLR_SB_LOAD(b_)                                                   \
  {                                                                         \
    int o_ = (b_) * U * 32;                                              \
    asm volatile("" : "+v"(o_) : : "memory");                               \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                         \
      a[b_][u] = arow[o_ + u * 32];                                         \
      g[b_][u] = grow[o_ + u * 32];                                         \
    }                                                                       \
  }
  // the channel-0 lane's gate and gradient broadcast, then the dependent gather of batch b_'s arg-min records
#define SMPLR_SB_GATHER(b_)                                                                                     \
  asm volatile("" : : : "memory");                                                                              \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                               \
    g[b_][u] = g[b_][u] - __shfl((a[b_][u] == 1) ? g[b_][u] : 0.0f, 0, 32);                                     \
    rv[b_][u] = __builtin_bit_cast(f32x3g, __builtin_amdgcn_raw_buffer_load_b96(rrs, a[b_][u] * 16, 0, 0));     \
  }
  static_assert(NB > 1 && 2 * U <= SB_PF, "the first TWO batches come from the SB_PF pixels requested at kernel entry");
#pragma unroll
  for (int u = 0; u < U; ++u) {                  // batches 0 and 1 were requested at kernel entry: the walk starts with
    a[0][u] = pa[u];                             // both gathers instead of a round trip for batch 1's rows
    g[0][u] = pg[u];
    a[1][u] = pa[U + u];
    g[1][u] = pg[U + u];
  }
  SMPLR_SB_GATHER(0)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    // (scheduling fences: unrolled, the compiler would otherwise hoist EVERY batch's loads to the top - 185 spills)
    __builtin_amdgcn_sched_barrier(0);
    if (b + 2 < NB) { SMPLR_SB_LOAD(b + 2) }
    if (b + 1 < NB) { SMPLR_SB_GATHER(b + 1) }
    __builtin_amdgcn_sched_barrier(0);
    SMPLR_TL_STAMP(3 + b * 2);
    if (chok) {                                  // channel 0 has no part (its lanes have served the broadcasts above)
      // (the batch's first column through an empty asm: as compile-time constants the 48 columns became 48 packed
      // {-column, -row} operands hoisted out of the window loop - and spilled)
      float fb = (float)(b * U);
      asm volatile("" : "+v"(fb));
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float fc = fb + (float)u;
        const float du = rv[b][u].x - fc, dv = rv[b][u].y - fr;
        const float d2 = fmaf(du, du, dv * dv);
        const float t = d2 * rv[b][u].z;
        const float r = __builtin_amdgcn_rsqf(fmaxf(t, 1e-37f));      // (see seg_bwd_row: m d = t r, m / d = m^2 r)
        const float kk = (-g[b][u] * fast_exp_neg(t * r)) * (rv[b][u].z * r);
        const bool on = kk != 0.0f;
        if (on && a[b][u] != cur) {
          if (DET) seg_flush_det(acc64, cur, sx, sy, scale);
          else seg_flush(acc, cur, sx, sy);
          cur = a[b][u];
          sx = 0.0f;
          sy = 0.0f;
        }
        sx = fmaf(kk, du, sx);
        sy = fmaf(kk, dv, sy);
      }
    }
    SMPLR_TL_STAMP(4 + b * 2);
  }
#undef SMPLR_SB_LOAD
#undef SMPLR_SB_GATHER
  if (DET) seg_flush_det(acc64, cur, sx, sy, scale);
  else seg_flush(acc, cur, sx, sy);
}

// The row walk when the loss head's backward is fused in (seg_bwd_kernel<.., LOSS = true>): d loss / d score of a
// channel is rebuilt per pixel from what the forward's loss epilogue left (raster_fwd_kernel<true>),
//   g_c = A (delta_ct - softmax_c) - g_background,   A = dloss q_t softmax_t,   softmax_c = exp(score_c) / sum exp,
// with score_c the lane's own recomputed score: every lane of a pixel's 32-lane group reads the same 16 B of `stats`
// (k / sum exp | k x the background's share | k = q_t softmax_t | label) and 4 B of dloss - one request per group -
// instead of its own 4 B of a 128-B row of dseg, and folds them at once into the two numbers it needs per pixel,
// c1 = A delta_ct - g_background and c2 = A / sum exp (g_c = c1 - c2 exp(score_c)).
struct LossIn { const float *dloss; const float4 *stats; };

// one batch of SB_U pixels of a row: a = arg-min slots, dl = dloss, st = stats of the pixels c0 .. c0 + SB_U - 1
template <bool MW, bool DET>
__device__ __forceinline__ void seg_bwd_batch_loss(int c0, int (&a)[SB_U], const float (&dl)[SB_U],
                                                   const float4 (&st)[SB_U], __amdgpu_buffer_rsrc_t rrs, float *acc,
                                                   int W, bool chok, int ch, float fr, int base, float scale, int &cur,
                                                   float &sx, float &sy) {
  unsigned long long *acc64 = reinterpret_cast<unsigned long long *>(acc);
  float c1[SB_U], c2[SB_U];
  float4 rv[SB_U];
#pragma unroll
  for (int u = 0; u < SB_U; ++u) {
    // stats = (k / sum exp, k x the background's share, k, label), k = q_t softmax_t: with dl = dloss
    //   g_c = dl (k delta_ct - k share_0) - dl (k / sum exp) exp(score_c) = c1 - c2 exp(score_c)
    c1[u] = dl[u] * ((__float_as_int(st[u].w) == ch ? st[u].z : 0.0f) - st[u].y);
    c2[u] = dl[u] * st[u].x;
    if (!(chok && c0 + u < W)) a[u] = -1;
    if (MW) {
      a[u] -= base;
      if (a[u] >= SB_SLOTS) a[u] = -1;
    }
    rv[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrs, (base + a[u]) * 16, 0, 0));
  }
#pragma unroll
  for (int u = 0; u < SB_U; ++u) {
    const float fc = (float)(c0 + u);
    const float du = rv[u].x - fc, dv = rv[u].y - fr;
    const float d2 = fmaf(du, du, dv * dv);
    const float t = d2 * rv[u].z;
    const float r = __builtin_amdgcn_rsqf(fmaxf(t, 1e-37f));      // (see seg_bwd_row: m d = t r, m / d = m^2 r)
    const float sc = fast_exp_neg(t * r);
    const float g = c1[u] - c2[u] * __expf(sc);
    float kk = (-g * sc) * (rv[u].z * r);                          // (a masked slot read zeros: m^2 = 0, kk = 0)
    if (MW && a[u] < 0) kk = 0.0f;
    const bool on = kk != 0.0f;
    if (on && a[u] != cur) {
      if (DET) seg_flush_det(acc64, cur, sx, sy, scale);
      else seg_flush(acc, cur, sx, sy);
      cur = a[u];
      sx = 0.0f;
      sy = 0.0f;
    }
    sx = fmaf(kk, du, sx);
    sy = fmaf(kk, dv, sy);
  }
}

// pa / pg: the first batch's arg-min slots and dloss as requested at kernel entry (FAST); its stats were only touched
// there (one dword per lane = the batch's 128 B: an L1 hit now) - held in registers through the barrier the 32 dwords
// per lane push the kernel past its register budget and the compiler parks them in scratch, behind a wait for the
// very requests they were to overlap.
template <bool MW, bool FAST, bool DET>
__device__ __forceinline__ void seg_bwd_row_loss(LossIn li, const short *__restrict__ arg, const float4 *__restrict__ R,
                                                 int rbytes, float *acc, size_t row0, int W, int C, int ch, float fr,
                                                 int base, float scale, const int *pa, const float *pg) {
  const bool chok = ch >= 1 && ch < C;
  const __amdgpu_buffer_rsrc_t rrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(R), 0, rbytes, 0x00020000);
  const short *arow = arg + row0 * 32 + ch;
  const float *drow = li.dloss + row0;
  const float4 *srow = li.stats + row0;
  int cur = 0;                                   // (slot 0 with a sum of zero: the first flush adds nothing)
  float sx = 0.0f, sy = 0.0f;
  for (int c0 = 0; c0 < W; c0 += SB_U) {
    int a[SB_U];
    float dl[SB_U];
    float4 st[SB_U];
    if (FAST) {                                    // W a multiple of SB_U: one address per stream, immediate offsets
      const short *ab = arow + c0 * 32;
      const float *db = drow + c0;
      const float4 *sb = srow + c0;
#pragma unroll
      for (int u = 0; u < SB_U; ++u) {
        if (c0 == 0) {                             // (uniform)
          a[u] = pa[u];
          dl[u] = pg[u];
        } else {
          a[u] = ab[u * 32];
          dl[u] = db[u];
        }
        st[u] = sb[u];
      }
    } else {                                       // (pixels past the row's end repeat its last one and are masked)
#pragma unroll
      for (int u = 0; u < SB_U; ++u) {
        const int cc = min(c0 + u, W - 1);
        a[u] = arow[cc * 32];
        dl[u] = drow[cc];
        st[u] = srow[cc];
      }
    }
    seg_bwd_batch_loss<MW, DET>(c0, a, dl, st, rrs, acc, W, chok, ch, fr, base, scale, cur, sx, sy);
  }
  unsigned long long *acc64 = reinterpret_cast<unsigned long long *>(acc);
  if (DET) seg_flush_det(acc64, cur, sx, sy, scale);
  else seg_flush(acc, cur, sx, sy);
}

// seg_bwd_row_pipe for the fused loss head: per pixel the lane needs its arg-min slot (2 B), dloss (4 B) and stats
// (16 B, the same for all 32 lanes of the pixel's group) and then its record.  Three stages in flight: rows of batch
// b + 2 requested; batch b + 1's rows folded into (c1, c2) - g_c = c1 - c2 exp(score_c), see seg_bwd_batch_loss - and
// its records gathered; batch b summed.  Four pixels per batch keep that within the block's 168 registers.
template <int U, int NB, bool DET>
__device__ __forceinline__ void seg_bwd_row_loss_pipe(LossIn li, const short *__restrict__ arg, const float4 *__restrict__ R,
                                                      int rbytes, float *acc, size_t row0, int ch, float fr, float scale,
                                                      const int *pa, const float *pg) {
  unsigned long long *acc64 = reinterpret_cast<unsigned long long *>(acc);
  const bool chok = ch >= 1;
  const __amdgpu_buffer_rsrc_t rrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(R), 0, rbytes, 0x00020000);
  const short *arow = arg + row0 * 32 + ch;
  const float *drow = li.dloss + row0;
  const float4 *srow = li.stats + row0;
  static_assert(U <= SB_U, "the first batch comes from the SB_U pixels requested at kernel entry");
  int a[NB][U];
  float dl[NB][U], c1[NB][U], c2[NB][U];
  float4 st[NB][U], rv[NB][U];
  int cur = 0;                                   // (slot 0 with a sum of zero: the first flush adds nothing)
  float sx = 0.0f, sy = 0.0f;
#define SMPLR_SBL_LOAD(b_, first_)                                          \
  {                                                                         \
    int o_ = (b_) * U;                                                      \
    asm volatile("" : "+v"(o_) : : "memory");                               \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                         \
      if (first_) {                                                         \
        a[b_][u] = pa[u];                                                   \
        dl[b_][u] = pg[u];                                                  \
      } else {                                                              \
        a[b_][u] = arow[(o_ + u) * 32];                                     \
        dl[b_][u] = drow[o_ + u];                                           \
      }                                                                     \
      st[b_][u] = srow[o_ + u];                                             \
    }                                                                       \
  }
#define SMPLR_SBL_GATHER(b_)                                                                                    \
  asm volatile("" : : : "memory");                                                                              \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                               \
    c1[b_][u] = dl[b_][u] * ((__float_as_int(st[b_][u].w) == ch ? st[b_][u].z : 0.0f) - st[b_][u].y);           \
    c2[b_][u] = dl[b_][u] * st[b_][u].x;                                                                        \
    if (!chok) a[b_][u] = -1;                                                                                   \
    rv[b_][u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrs, a[b_][u] * 16, 0, 0));    \
  }
  SMPLR_SBL_LOAD(0, true)
  if (NB > 1) { SMPLR_SBL_LOAD(1, false) }
  SMPLR_SBL_GATHER(0)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    __builtin_amdgcn_sched_barrier(0);
    if (b + 2 < NB) { SMPLR_SBL_LOAD(b + 2, false) }
    if (b + 1 < NB) { SMPLR_SBL_GATHER(b + 1) }
    __builtin_amdgcn_sched_barrier(0);
    float fb = (float)(b * U);
    asm volatile("" : "+v"(fb));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float fc = fb + (float)u;
      const float du = rv[b][u].x - fc, dv = rv[b][u].y - fr;
      const float d2 = fmaf(du, du, dv * dv);
      const float t = d2 * rv[b][u].z;
      const float r = __builtin_amdgcn_rsqf(fmaxf(t, 1e-37f));      // (see seg_bwd_row: m d = t r, m / d = m^2 r)
      const float sc = fast_exp_neg(t * r);
      const float g = c1[b][u] - c2[b][u] * __expf(sc);
      const float kk = (-g * sc) * (rv[b][u].z * r);                 // (a masked slot read zeros: m^2 = 0, kk = 0)
      const bool on = kk != 0.0f;
      if (on && a[b][u] != cur) {
        if (DET) seg_flush_det(acc64, cur, sx, sy, scale);
        else seg_flush(acc, cur, sx, sy);
        cur = a[b][u];
        sx = 0.0f;
        sy = 0.0f;
      }
      sx = fmaf(kk, du, sx);
      sy = fmaf(kk, dv, sy);
    }
  }
#undef SMPLR_SBL_LOAD
#undef SMPLR_SBL_GATHER
  if (DET) seg_flush_det(acc64, cur, sx, sy, scale);
  else seg_flush(acc, cur, sx, sy);
}

template <bool DET, bool LOSS>
__global__ __launch_bounds__(32 * SB_ROWS_BIG) void seg_bwd_kernel(const float *__restrict__ dseg,
                                                      const short *__restrict__ arg,
                                                      const float4 *__restrict__ rec, int S, int VP, int W,
                                                      int P, float *__restrict__ dproj,
                                                      float *__restrict__ part, int rows, LossIn li, int pipe) {
  // SB_SLOTS x 2 accumulators: fp32 (32 KB), or 64-bit fixed point in the deterministic form (64 KB)
  extern __shared__ __attribute__((aligned(16))) float acc[];
  unsigned long long *acc64 = reinterpret_cast<unsigned long long *>(acc);
  __shared__ unsigned s_gmax, s_m2max;
  const int n = blockIdx.y, tid = threadIdx.x, nthr = 32 * rows;   // a 32-lane group per row of the block
  SMPLR_TL_WAVE(g_tl_segbwd, 12, blockIdx.y * gridDim.x + blockIdx.x, (W <= 80 ? TL_SEGBWD_WG : 0))
  const float4 *R = rec + (size_t)n * S;
  const int nslots = __float_as_int(R[S - 1].x);
  const int C = P + 1, npix = W * W;
  const int ch = tid & 31, strip = tid >> 5;
  // Output (flipped) row of this strip.  Rows across the body end more runs (flushes) than rows of background, and a
  // workgroup waits for its slowest wave, the launch for its slowest workgroup: the row blocks of a mesh take
  // interleaved rows, and the two strips of a wave an early and a late one of the block's.
  const int kidx = (strip & 1) ? rows - 1 - (strip >> 1) : (strip >> 1);
  const int ro = blockIdx.x + gridDim.x * kidx;
  const float fr = (float)(W - 1 - ro);
  const size_t row0 = (size_t)n * npix + (size_t)ro * W;
  const bool fast = C == 32 && W % SB_U == 0;               // block-uniform
  // The first batch of the row walk is requested here, behind the header: its trip to HBM (2 us at the head of a
  // 20 us kernel that otherwise streams at 4.8 TB/s) then runs under the zeroing of the accumulators and its barrier.
  // (round 4: the first SB_PF = 12 pixels - two batches of the pipelined walk, which then starts with two gathers)
  int pa[SB_PF];
  float pg[SB_PF];
  float warm = 0.0f;                                          // LOSS: touches the first batch's stats (see seg_bwd_row_loss)
#pragma unroll
  for (int u = 0; u < SB_PF; ++u) { pa[u] = 0; pg[u] = 0.0f; }
  if (fast && ro < W) {
#pragma unroll
    for (int u = 0; u < SB_PF; ++u) {
      if (LOSS && u >= SB_U) continue;                        // (the loss walk takes its first batch only)
      const int uu = min(u, W - 1);
      pa[u] = arg[(row0 + uu) * 32 + ch];
      pg[u] = LOSS ? li.dloss[row0 + uu] : dseg[(row0 + uu) * 32 + ch];
    }
    if (LOSS) warm = reinterpret_cast<const float *>(li.stats + row0)[ch];
  }
  if (dproj) {                                                // (NULL: the consumer gathers the slot sums itself)
    // this block's share of the mesh's dproj rows := 0 (the merge kernel then stores the sums)
    float *dp = dproj + (size_t)n * VP * 3;
    const int tot = VP * 3, per = (tot + gridDim.x - 1) / gridDim.x;
    const int z0 = blockIdx.x * per, z1 = min(tot, z0 + per);
    for (int i = z0 + tid; i < z1; i += nthr) dp[i] = 0.0f;
  }
  float scale = 1.0f, inv_scale = 1.0f;
  if (DET) {
    // Bound of one term: |g - g0| m |du| / d <= 2 max|dseg| max(m); a slot collects at most rows x W of them.
    // max is order-independent, so the scale itself is reproducible.  (bit patterns of non-negative floats order
    // like the floats; a NaN / inf cotangent gives a NaN / inf bound and garbage either way)
    if (tid == 0) { s_gmax = 0u; s_m2max = 0u; }
    __syncthreads();
    unsigned gm = 0u, mm = 0u;
    if (ro < W) {
      if (LOSS)        // |g_c| = |A (delta_ct - softmax_c) - g_background| <= 2 |A|, A = dloss q_t softmax_t
        for (int i = ch; i < W; i += 32) gm = max(gm, __float_as_uint(fabsf(2.0f * li.dloss[row0 + i] * li.stats[row0 + i].z)));
      else
        for (int i = ch; i < W * C; i += 32) gm = max(gm, __float_as_uint(fabsf(dseg[row0 * C + i])));
    }
    for (int i = tid; i < nslots; i += nthr) mm = max(mm, __float_as_uint(fabsf(R[i].z)));
    atomicMax(&s_gmax, gm);
    atomicMax(&s_m2max, mm);
    __syncthreads();
    int eg, em;
    frexpf(__uint_as_float(s_gmax), &eg);                      // max|g| < 2^eg
    frexpf(fmaxf(__uint_as_float(s_m2max), 1.0f), &em);        // max m^2 < 2^em -> max m < 2^((em + 1) / 2)
    int terms = 1;
    while ((1 << terms) < rows * W) ++terms;                   // rows x W <= 2^terms
    // |sum| < 2^(1 + eg + (em + 1) / 2 + terms) must stay below 2^62
    const int e = min(max(60 - eg - (em + 1) / 2 - terms, -100), 100);
    scale = ldexpf(1.0f, e);
    inv_scale = ldexpf(1.0f, -e);
  }
  SMPLR_TL_STAMP(1);
  const int nwin = (nslots + SB_SLOTS - 1) / SB_SLOTS;    // 1 in the standard pipeline
  for (int win = 0; win < nwin; ++win) {
    const int base = win * SB_SLOTS;
    const int nsl = min(nslots - base, SB_SLOTS);
    if (win > 0) __syncthreads();
    if (DET) for (int i = tid; i < nsl * 2; i += nthr) acc64[i] = 0ull;
    else for (int i = tid; i < nsl * 2; i += nthr) acc[i] = 0.0f;
    __syncthreads();
    SMPLR_TL_STAMP(2);
    if (ro < W) {
      // (the compiler would otherwise start on the first batch - and wait for it - in front of the barrier)
#pragma unroll
      for (int u = 0; u < SB_PF; ++u) asm volatile("" : "+v"(pa[u]), "+v"(pg[u]));
      if (LOSS) {
        asm volatile("" : "+v"(warm));
        if (nwin == 1 && fast && W == 48 && pipe)
          seg_bwd_row_loss_pipe<4, 12, DET>(li, arg, R, S * 16, acc, row0, ch, fr, scale, pa, pg);
        else if (nwin == 1 && fast && W == 64 && pipe)
          seg_bwd_row_loss_pipe<4, 16, DET>(li, arg, R, S * 16, acc, row0, ch, fr, scale, pa, pg);
        else if (nwin == 1 && fast)
          seg_bwd_row_loss<false, true, DET>(li, arg, R, S * 16, acc, row0, W, C, ch, fr, 0, scale, pa, pg);
        else if (nwin == 1)
          seg_bwd_row_loss<false, false, DET>(li, arg, R, S * 16, acc, row0, W, C, ch, fr, 0, scale, pa, pg);
        else
          seg_bwd_row_loss<true, false, DET>(li, arg, R, S * 16, acc, row0, W, C, ch, fr, base, scale, pa, pg);
      } else if (nwin == 1 && fast && W == 48 && pipe)
        seg_bwd_row_pipe<6, 8, DET>(dseg, arg, R, S * 16, acc, row0, ch, fr, scale, pa, pg);
      else if (nwin == 1 && fast && W == 64 && pipe)
        seg_bwd_row_pipe<4, 16, DET>(dseg, arg, R, S * 16, acc, row0, ch, fr, scale, pa, pg);
      else if (nwin == 1 && fast)
        seg_bwd_row<false, true, DET>(dseg, arg, R, S * 16, acc, row0, W, C, ch, fr, 0, scale, pa, pg);
      else if (nwin == 1)
        seg_bwd_row<false, false, DET>(dseg, arg, R, S * 16, acc, row0, W, C, ch, fr, 0, scale, pa, pg);
      else seg_bwd_row<true, false, DET>(dseg, arg, R, S * 16, acc, row0, W, C, ch, fr, base, scale, pa, pg);
    }
    SMPLR_TL_STAMP(24);
    __syncthreads();
    SMPLR_TL_STAMP(25);
    float *dst = part + (((size_t)n * gridDim.x + blockIdx.x) * SB_NWIN + win) * (SB_SLOTS * 2);
    if (DET) for (int i = tid; i < nsl * 2; i += nthr) dst[i] = (float)(long long)acc64[i] * inv_scale;
    else for (int i = tid; i < nsl * 2; i += nthr) dst[i] = acc[i];
  }
  SMPLR_TL_STAMP(26);
}

__global__ __launch_bounds__(256) void seg_bwd_merge_kernel(const float *__restrict__ part,
                                                            const float4 *__restrict__ rec, int S, int VP,
                                                            int nsplit, float *__restrict__ dproj) {
  const int n = blockIdx.y;
  const float4 *R = rec + (size_t)n * S;
  const int nslots = __float_as_int(R[S - 1].x);
  for (int slot = blockIdx.x * 256 + threadIdx.x; slot < nslots; slot += SB_SLOTS) {
    // the record and the partials are requested together (one round trip); the sentinel test comes after
    const int v = __float_as_int(R[slot].w);
    const int win = slot / SB_SLOTS;
    const float *p = part + ((size_t)n * nsplit * SB_NWIN + win) * (SB_SLOTS * 2) + (slot - win * SB_SLOTS) * 2;
    float sx = 0.0f, sy = 0.0f;
    for (int s = 0; s < nsplit; ++s) {
      const float2 t = *reinterpret_cast<const float2 *>(p + (size_t)s * SB_NWIN * (SB_SLOTS * 2));
      sx += t.x;
      sy += t.y;
    }
    if (v >= 0) {                                            // v < 0: padding sentinel
      float *o = dproj + ((size_t)n * VP + v) * 3;
      o[0] = sx;
      o[1] = sy;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Silhouette: every vertex is "global" with weight 1/1.2 (no mask): brute force over all of them.
__global__ __launch_bounds__(256) void silh_prep_kernel(const float *__restrict__ proj, int VP, int KP,
                                                        float4 *__restrict__ sorted) {
  const int n = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  if (k >= KP) return;
  float4 o = make_float4(INFINITY, INFINITY, 1.0f, __int_as_float(-1));
  if (k < VP) {
    const float *p = proj + ((size_t)n * VP + k) * 3;
    o = make_float4(p[0], p[1], 1.0f, __int_as_float(k));
  }
  sorted[(size_t)n * KP + k] = o;
}

__global__ __launch_bounds__(RT) void silh_fwd_kernel(const float4 *__restrict__ sorted, int KP, int W,
                                                      float *__restrict__ out, int *__restrict__ arg_out) {
  const int n = blockIdx.y;
  const int q = blockIdx.x * RT + threadIdx.x;
  const int npix = W * W;
  const bool live = q < npix;
  const int qc = live ? q : npix - 1;
  const int r = qc / W, c = qc - r * W;
  const float fc = (float)c, fr = (float)r;
  const float4 *S = sorted + (size_t)n * KP;
  float best = INFINITY;
  int bestk = 0;
  for (int k = 0; k < KP; k += CH) {
    float cm = pair_key(S[k], fc, fr);
#pragma unroll
    for (int j = 1; j < CH; ++j) cm = fminf(cm, pair_key(S[k + j], fc, fr));
    if (cm < best) { best = cm; bestk = k; }
  }
  int pos = -1;
  float score = 0.0f;
  if (best < INFINITY) {
#pragma unroll
    for (int j = CH - 1; j >= 0; --j) {
      const float4 a = S[bestk + j];
      if (pair_key(a, fc, fr) == best) pos = __float_as_int(a.w);
    }
    score = expf(-sqrtf(best) / 1.2f);
  }
  if (live) {
    const size_t o = ((size_t)n * W + (W - 1 - r)) * W + c;   // rows flipped (:42)
    out[o * 2 + 0] = 1.0f - score;
    out[o * 2 + 1] = score;
    arg_out[o] = pos;
  }
}

// ------------------------------------------------------------------------------------------------
// Pruned silhouette forward (exact): one workgroup per mesh, everything in LDS, four lanes per pixel.
// Vertices are binned into 1-px cells (cell = rounded position, on a window of the image plus SM px
// of margin; the rest are "outliers", always evaluated) and kept in LDS sorted by cell.  For a pixel:
//  (1) the nearest OCCUPIED cell centre, by an exact distance transform of the occupancy grid: per
//      cell row the nearest occupied column comes from the row's occupancy bits (clz / ctz), then
//      the minimum over the rows;
//  (2) the vertices of that cell give a real distance d1 (<= Dmin + 0.7072: a vertex lies within
//      0.7072 px of its cell centre);
//  (3) any closer vertex lives in a cell whose centre is within R = d1 + 0.7072 of the pixel, so
//      only the occupied cells inside that disc are evaluated: the set bits of each row's mask
//      within the disc's chord (9 cells of ~10 vertices inside the body, a thin arc outside it).
// The four lanes of a pixel take every fourth row in (1) and (3) and every fourth vertex in (2),
// then reduce with two xor-shuffles; 16 neighbouring pixels (a 4 x 4 tile) share a wave, so its
// lanes walk similar rows.  The earlier version scanned ALL occupied cells per wave with scalar
// loads from global memory (any lane's candidate was everybody's work, and every record group cost
// an L2 round trip): 57 + 212 us at B = 128.  Ties go to the lowest vertex index, as the dense
// formulation's arg-max does (keys are packed (d^2 bits, index) and compared as 64-bit integers).
constexpr int SM = 8;            // margin of the cell window around the image
constexpr int SILH_WMAX = 96;    // (W + 16)^2 cell offsets + the vertex records must fit LDS
constexpr int SF_T = 1024;

static size_t silh_fused_lds(int VP, int W) {
  const int GW = W + 2 * SM;
  return (size_t)((GW * GW + 2) & ~1) * 4 + (size_t)2 * GW * 8 + (size_t)VP * 12;
}

// distance (in columns) from cx to the nearest set bit of a 128-bit row mask; 1 << 20 if the row is empty
__device__ __forceinline__ int nearest_bit(unsigned long long m0, unsigned long long m1, int cx) {
  int dl = 1 << 20, dr = 1 << 20;
  {
    unsigned long long lo = m0, hi = m1;                 // bits <= cx
    if (cx < 63) { lo &= (2ull << cx) - 1ull; hi = 0ull; }
    else if (cx == 63) hi = 0ull;
    else if (cx < 127) hi &= (2ull << (cx - 64)) - 1ull;
    if (hi) dl = cx - (127 - __clzll((long long)hi));
    else if (lo) dl = cx - (63 - __clzll((long long)lo));
  }
  {
    unsigned long long lo = m0, hi = m1;                 // bits >= cx
    if (cx < 64) lo &= ~((1ull << cx) - 1ull);
    else { lo = 0ull; hi &= ~((1ull << (cx - 64)) - 1ull); }
    if (lo) dr = (__ffsll((long long)lo) - 1) - cx;
    else if (hi) dr = 64 + (__ffsll((long long)hi) - 1) - cx;
  }
  return dl <= dr ? -dl : dr;                            // signed offset to the nearest occupied column
}

// the same for a row of at most 64 cells
__device__ __forceinline__ int nearest_bit1(unsigned long long m, int cx) {
  const unsigned long long le = m & ((cx < 63) ? ((2ull << cx) - 1ull) : ~0ull);   // bits <= cx
  const unsigned long long ge = m & ~((1ull << cx) - 1ull);                          // bits >= cx
  const int dl = le ? cx - (63 - __clzll((long long)le)) : (1 << 20);
  const int dr = ge ? (__ffsll((long long)ge) - 1) - cx : (1 << 20);
  return dl <= dr ? -dl : dr;
}

constexpr int LPP = 4;     // lanes per pixel (B = 128, W = 48: 2 lanes 49 us, 4: 48.5, 8: 60, 16: 87)
__device__ __forceinline__ unsigned long long quad_min(unsigned long long k) {
#pragma unroll
  for (int o = 1; o < LPP; o <<= 1) {
    const unsigned int lo = __shfl_xor((unsigned int)k, o, 64), hi = __shfl_xor((unsigned int)(k >> 32), o, 64);
    const unsigned long long other = ((unsigned long long)hi << 32) | lo;
    k = other < k ? other : k;
  }
  return k;
}

template <bool ONEWORD>   // ONEWORD: the cell window is at most 64 wide (W <= 48): one mask word per row
__global__ __launch_bounds__(SF_T) void silh_fused_kernel(const float *__restrict__ proj, int VP, int W,
                                                          float *__restrict__ out, int *__restrict__ arg_out) {
  // 16-B aligned: the 64-bit row masks behind the counters need 8, whatever the static LDS in front
  extern __shared__ __attribute__((aligned(16))) int s_cnt[];   // cells (+2, even) | row masks | u[VP] | v[VP] | index[VP]
  __shared__ int s_next_tile;
  if (threadIdx.x == 0) s_next_tile = 0;             // (ordered by the binning's barriers)
  __shared__ int s_wave[SF_T / 64];
  __shared__ int s_nout;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int GW = W + 2 * SM, cells = GW * GW;
  // gridDim.y workgroups share a mesh (each bins it for itself and takes every gridDim.y-th tile):
  // with fewer meshes than CUs the pixel phase, not the binning, is what there is to spread
  unsigned long long *rowmask = reinterpret_cast<unsigned long long *>(s_cnt + ((cells + 2) & ~1));
  float *sU = reinterpret_cast<float *>(rowmask + 2 * GW), *sV = sU + VP;
  int *sI = reinterpret_cast<int *>(sV + VP);
  const float *pj = proj + (size_t)n * VP * 3;
  // ---- binning: every vertex requested up front
  float pu[IPT_MAX], pv[IPT_MAX];
#pragma unroll
  for (int j = 0; j < IPT_MAX; ++j) {
    const int v = min(tid + j * SF_T, VP - 1);
    pu[j] = pj[v * 3];
    pv[j] = pj[v * 3 + 1];
  }
  for (int i = tid; i < cells; i += SF_T) s_cnt[i] = 0;
  for (int i = tid; i < 2 * GW; i += SF_T) rowmask[i] = 0ull;
  if (tid == 0) s_nout = 0;
  __syncthreads();
  int pc[IPT_MAX];
#pragma unroll
  for (int j = 0; j < IPT_MAX; ++j) {
    const int v = tid + j * SF_T;
    pc[j] = -2;                                  // no vertex
    if (v < VP) {
      const float cx = rintf(pu[j]) + (float)SM, cy = rintf(pv[j]) + (float)SM;
      if (cx >= 0.0f && cx < (float)GW && cy >= 0.0f && cy < (float)GW) {
        pc[j] = (int)cy * GW + (int)cx;
        atomicAdd(&s_cnt[pc[j]], 1);
        atomicOr(&rowmask[2 * (int)cy + ((int)cx >> 6)], 1ull << ((int)cx & 63));
      } else {
        pc[j] = -1;                              // outlier (also NaN positions)
        atomicAdd(&s_nout, 1);
      }
    }
  }
  __syncthreads();
  // exclusive scan of the counts -> placement cursors; after placement s_cnt[e] = end of cell e
  // (= start of cell e + 1), so one array serves as both
  const int ept = (cells + SF_T - 1) / SF_T;
  const int e0 = tid * ept, e1 = min(cells, e0 + ept);
  int lc = 0;
  for (int e = e0; e < e1; ++e) lc += s_cnt[e];
  int tot_v;
  int run_v = block_excl_scan(lc, s_wave, &tot_v);
  for (int e = e0; e < e1; ++e) {
    const int c = s_cnt[e];
    s_cnt[e] = run_v;
    run_v += c;
  }
  const int nout = s_nout;
  __syncthreads();
  if (tid == 0) s_nout = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < IPT_MAX; ++j) {
    const int v = tid + j * SF_T;
    int dst = -1;
    if (pc[j] >= 0) dst = atomicAdd(&s_cnt[pc[j]], 1);
    else if (pc[j] == -1) dst = tot_v + atomicAdd(&s_nout, 1);
    if (dst >= 0) { sU[dst] = pu[j]; sV[dst] = pv[j]; sI[dst] = v; }
  }
  __syncthreads();
  // ---- pixels: a wave takes 4 x 4 tiles, 4 lanes per pixel
  const int lane = tid & 63;
  const int sub = lane & (LPP - 1), pq = lane / LPP;       // lane of the pixel's group, pixel of the tile
  constexpr int TH = 64 / LPP / 4;                         // tile: 4 pixels wide, TH high
  const int tpr = (W + 3) / 4, ntile = tpr * ((W + TH - 1) / TH);
#define SMPLR_SILH_VERTEX(i_)                                                                   \
  {                                                                                             \
    const float du_ = sU[i_] - fc, dv_ = sV[i_] - fr;                                           \
    const unsigned long long k_ =                                                               \
        ((unsigned long long)__float_as_uint(fmaf(du_, du_, dv_ * dv_)) << 32) | (unsigned int)sI[i_]; \
    best = k_ < best ? k_ : best;                                                               \
  }
  // Tiles are handed out through a counter in LDS, not round-robin: tiles over the body cost several times a
  // background tile, and the workgroup waits for its slowest wave (W = 48: 51.5 -> 48.5 us, W = 64: 107 -> 76 us at
  // B = 128; in image order - starting at the middle rows measured the same, from both ends inwards 4 us worse).
  const int nloc = (ntile - (int)blockIdx.y + (int)gridDim.y - 1) / (int)gridDim.y;
  for (;;) {
    int t = 0;
    if (lane == 0) t = atomicAdd(&s_next_tile, 1);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= nloc) break;
    const int tile = t * (int)gridDim.y + (int)blockIdx.y;
    const int ty = tile / tpr, tx = tile - ty * tpr;
    const int r_ = ty * TH + (pq >> 2), c_ = tx * 4 + (pq & 3);
    const bool live = r_ < W && c_ < W;
    const int r = min(r_, W - 1), c = min(c_, W - 1);      // clamped lanes repeat a border pixel
    const float fc = (float)c, fr = (float)r;
    const int cx = c + SM, cy = r + SM;
    unsigned long long best = ~0ull;                       // (d^2 bits << 32) | vertex index; d^2 >= 0: bit order = value order
    // (1) nearest occupied cell centre: rows sub, sub + 4, ...
    unsigned long long near = ~0ull;                       // (d^2 bits << 32) | cell
    // rows cy, cy +- 1, cy +- 2, ... (this lane: offsets sub, sub + 4, ...); a lane stops once the
    // row offset alone exceeds its own best (such rows cannot beat it, hence not the quad's minimum)
    for (int k = sub; k < GW; k += LPP) {
      const float fk = (float)k;
      if ((unsigned long long)__float_as_uint(fk * fk) << 32 > near) break;
#pragma unroll
      for (int sgn = 0; sgn < 2; ++sgn) {
        const int y = sgn ? cy - k : cy + k;
        if (y < 0 || y >= GW || (sgn && k == 0)) continue;
        const unsigned long long m0 = rowmask[2 * y], m1 = ONEWORD ? 0ull : rowmask[2 * y + 1];
        if ((m0 | m1) == 0ull) continue;
        const int off = ONEWORD ? nearest_bit1(m0, cx) : nearest_bit(m0, m1, cx);
        const float dx = (float)off;
        const unsigned long long kk =
            ((unsigned long long)__float_as_uint(fmaf(dx, dx, fk * fk)) << 32) | (unsigned int)(y * GW + cx + off);
        near = kk < near ? kk : near;
      }
    }
    near = quad_min(near);
    if (near != ~0ull) {
      // (2) the nearest cell's vertices, every fourth one per lane
      {
        const int e = (int)(near & 0xffffffffull);
        const int i0 = e ? s_cnt[e - 1] : 0, i1 = s_cnt[e];
        for (int i = i0 + sub; i < i1; i += LPP) SMPLR_SILH_VERTEX(i)
        best = quad_min(best);
      }
      // (3) every occupied cell whose centre is within R = d1 + 0.7072 (+ rounding slack)
      const float d1 = sqrtf(__uint_as_float((unsigned int)(best >> 32)));
      const float R = d1 + 0.7072f;
      const float R2 = R * R * 1.0001f;
      const int rad = (int)R + 1;
      const int ylo = max(0, cy - rad), yhi = min(GW - 1, cy + rad);
      for (int y = ylo + sub; y <= yhi; y += LPP) {
        const float dy = (float)(y - cy);
        const float rem = R2 - dy * dy;
        if (rem < 0.0f) continue;
        const int w = (int)sqrtf(rem) + 1;                 // generous: every cell is tested exactly below
        const int xlo = max(0, cx - w), xhi = min(GW - 1, cx + w);
        unsigned long long m0 = rowmask[2 * y], m1 = ONEWORD ? 0ull : rowmask[2 * y + 1];
        if (xlo < 64) m0 &= ~((1ull << xlo) - 1ull); else { m0 = 0ull; m1 &= ~((1ull << (xlo - 64)) - 1ull); }
        if (xhi < 63) { m0 &= (2ull << xhi) - 1ull; m1 = 0ull; }
        else if (xhi == 63) m1 = 0ull;
        else if (xhi < 127) m1 &= (2ull << (xhi - 64)) - 1ull;
        for (int half = 0; half < (ONEWORD ? 1 : 2); ++half) {
          unsigned long long m = half ? m1 : m0;
          while (m) {
            const int x = (__ffsll((long long)m) - 1) + 64 * half;
            m &= m - 1ull;
            const float dx = (float)(x - cx);
            if (fmaf(dx, dx, dy * dy) <= R2) {
              const int e = y * GW + x;
              const int i0 = e ? s_cnt[e - 1] : 0, i1 = s_cnt[e];
              for (int i = i0; i < i1; ++i) SMPLR_SILH_VERTEX(i)
            }
          }
        }
      }
    }
    for (int i = tot_v + sub; i < tot_v + nout; i += LPP) SMPLR_SILH_VERTEX(i)     // outliers: always
    best = quad_min(best);
    if (live && sub == 0) {
      float score = 0.0f;
      int pos = -1;
      if (best != ~0ull) {
        score = expf(-sqrtf(__uint_as_float((unsigned int)(best >> 32))) / 1.2f);
        pos = (int)(best & 0xffffffffull);
      }
      const size_t o = ((size_t)n * W + (W - 1 - r)) * W + c;   // rows flipped (:42)
      out[o * 2 + 0] = 1.0f - score;
      out[o * 2 + 1] = score;
      arg_out[o] = pos;
    }
  }
#undef SMPLR_SILH_VERTEX
}

// ------------------------------------------------------------------------------------------------
// Pruned silhouette forward, one LANE per pixel, candidates shared by the 64 pixels of a wave's 8 x 8 tile (cell
// windows of at most 64 columns: W <= 48, the reference's silhouette size, train_stage2_silhouette.py:349-354).
// Same binning and the same exact pruning idea as silh_fused_kernel, re-cut around what the counters showed: that
// kernel issued 18.6 M vector wave-instructions at B = 128 - more than the whole 31-part rasteriser - in nested
// per-lane loops over rows, cells and vertices (profiles/r02_silh_*).  Here:
//  (0) binning also leaves, per pixel of the image, the vertex of the pixel's OWN cell nearest to it (a pixel centre
//      is its cell's centre: one 64-bit LDS atomic min per vertex, order-independent), and per row of the cell
//      grid the signed offset from every column to the row's nearest occupied cell (one byte per cell);
//  (1) a pixel whose own cell is occupied starts from that vertex at distance d <= 0.7072 - and is done unless
//      d > 0.5, since every other cell's square lies at least half a cell away;
//  (2) any other pixel walks the rows outwards from its own: nearest occupied cell q0 at squared centre distance
//      D2 = min(off^2 + k^2), one byte read per row, until k^2 > (sqrt(D2) + 1.4143)^2, keeping as bits of one word
//      the rows that hold a cell within that bound; the first vertex of q0 gives a real distance d <= sqrt(D2) + 0.7072;
//  (3) a vertex of cell (x, y) lies within half a cell of its centre, so it is at least hypot(max(|x - cx| - 0.5, 0),
//      max(|y - cy| - 0.5, 0)) from the pixel: only cells whose square comes within d can hold the nearest vertex.
//      Those are few (about three sparse cells of the outline per exterior pixel) and nearly the same for
//      neighbouring pixels, so the lanes OR their candidate cells into a 64 x 64-bit map in LDS (one word per cell
//      row, owned by the wave), lane y then takes row y's word and looks up the record range of its first run of
//      cells, and EVERY lane evaluates every record of every run - wave-uniform loops over ranges handed round by
//      v_readlane, broadcast LDS reads, no per-lane walks; an extra candidate can only lower a lane's minimum
//      towards the truth.
// Keys are (d^2 bits, vertex index) compared as 64-bit integers, exactly as in silh_fused_kernel: the same d^2
// expression, ties to the lowest vertex index - the two kernels give bit-identical outputs.
constexpr int SPX_TILE = 8;      // 8 x 8 pixels per wave
constexpr int SPX_EMPTY = 127;   // offset-table entry of an empty row

// LDS (bytes): cell starts | offset table | row words | own-cell keys | per-wave candidate maps | records.
// Cell rows have a stride of GW + 1 and the own-cell rows of W + 1: vertices that follow each other in the mesh sit
// above each other as often as side by side, and a stride of 64 words put all of those on one bank.
struct SpxLds { size_t gtab, rowmask, own, ubm, rec, total; int GWP, WP; };
static SpxLds silh_px_layout(int VP, int W) {
  SpxLds L;
  const int GW = W + 2 * SM;
  L.GWP = GW + 1;
  L.WP = W + 1;
  size_t off = (size_t)((GW * L.GWP + 4) & ~3) * 4;
  L.gtab = off;     off += (size_t)((GW * L.GWP + 15) & ~15);
  L.rowmask = off;  off += (size_t)((GW + 1) & ~1) * 8;
  L.own = off;      off += (size_t)W * L.WP * 8;
  L.ubm = off;      off += (size_t)(SF_T / 64) * 64 * 8;
  off = (off + 15) & ~(size_t)15;
  L.rec = off;      off += (size_t)VP * 16;
  L.total = off;
  return L;
}

// hint (optional, (B, W, W) as the output lies): per pixel a score exp(-x) with x >= the distance to SOME vertex - the
// 31-part rasteriser's largest part score of the pixel (raster_fwd_kernel's vmax: exp(-m d) of a real vertex, m >= 1).
// A pixel whose own cell is empty then takes -log(hint) as its search radius instead of walking the rows for the
// nearest occupied cell (step (2): a third of this kernel's time); the candidates of step (3) are a superset of
// those the nearest vertex' cell belongs to either way, so the result is the same bit for bit.
__global__ __launch_bounds__(SF_T) void silh_px_kernel(const float *__restrict__ proj, int VP, int W, SpxLds L,
                                                       float *__restrict__ out, int *__restrict__ arg_out,
                                                       const float *__restrict__ hint) {
  extern __shared__ __attribute__((aligned(16))) int s_cnt[];
  __shared__ int s_next_tile;
  if (threadIdx.x == 0) s_next_tile = 0;             // (ordered by the binning's barriers)
  __shared__ int s_wave[SF_T / 64];
  __shared__ int s_nout;
  const int n = blockIdx.x, tid = threadIdx.x;
  SMPLR_TL_WAVE(g_tl_silhpx, 16, blockIdx.x * gridDim.y + blockIdx.y, TL_SILHPX_WG)
  const int GW = W + 2 * SM, GWP = L.GWP, WP = L.WP, cells = GW * GWP;        // GW <= 64
  char *lds = reinterpret_cast<char *>(s_cnt);
  signed char *gtab = reinterpret_cast<signed char *>(lds + L.gtab);
  unsigned long long *rowmask = reinterpret_cast<unsigned long long *>(lds + L.rowmask);
  unsigned long long *own = reinterpret_cast<unsigned long long *>(lds + L.own);
  unsigned long long *ubm = reinterpret_cast<unsigned long long *>(lds + L.ubm) + (tid >> 6) * 64;
  float4 *sRec = reinterpret_cast<float4 *>(lds + L.rec);
  const float *pj = proj + (size_t)n * VP * 3;
  // ---- binning: every vertex requested up front
  float pu[IPT_MAX], pv[IPT_MAX];
#pragma unroll
  for (int j = 0; j < IPT_MAX; ++j) {
    const int v = min(tid + j * SF_T, VP - 1);
    pu[j] = pj[v * 3];
    pv[j] = pj[v * 3 + 1];
  }
  for (int i = tid; i <= cells; i += SF_T) s_cnt[i] = 0;
  for (int i = tid; i < W * WP; i += SF_T) own[i] = ~0ull;
  if (tid < GW) rowmask[tid] = 0ull;
  if (tid == 0) s_nout = 0;
  __syncthreads();
  SMPLR_TL_STAMP(1);
  int pc[IPT_MAX], rank[IPT_MAX];
#pragma unroll
  for (int j = 0; j < IPT_MAX; ++j) {
    const int v = tid + j * SF_T;
    pc[j] = -2;                                  // no vertex
    rank[j] = 0;
    if (v < VP) {
      const float ru = rintf(pu[j]), rv = rintf(pv[j]);
      const float cx = ru + (float)SM, cy = rv + (float)SM;
      if (cx >= 0.0f && cx < (float)GW && cy >= 0.0f && cy < (float)GW) {
        pc[j] = (int)cy * GWP + (int)cx;
        rank[j] = atomicAdd(&s_cnt[pc[j]], 1);              // arrival order within the cell
        if (ru >= 0.0f && ru < (float)W && rv >= 0.0f && rv < (float)W) {
          // the pixel at this cell's centre: its key for this vertex, as the pixel itself would compute it
          const float du = pu[j] - ru, dv = pv[j] - rv;
          atomicMin(&own[(int)rv * WP + (int)ru],
                    ((unsigned long long)__float_as_uint(fmaf(du, du, dv * dv)) << 32) | (unsigned int)v);
        }
      } else {
        pc[j] = -1;                              // outlier (also NaN positions)
        rank[j] = atomicAdd(&s_nout, 1);
      }
    }
  }
  __syncthreads();
  SMPLR_TL_STAMP(2);
  // exclusive scan of the counts: s_cnt[e] = start of cell e, s_cnt[cells] = vertices inside the window; the
  // occupied cells set their bit of the row words.  Thread (row tid / 16, segment tid % 16) takes the segment's cells
  // of its row - row-major order is thread order, and no cell index is ever divided by the row length (GW <= 64 rows)
  const int by = tid >> 4, bs = tid & 15;
  const int cpt = (GWP + 15) >> 4;
  const int bx0 = bs * cpt, bx1 = by < GW ? min(GWP, bx0 + cpt) : bx0;
  int lc = 0;
  {
    unsigned long long bits = 0ull;
    for (int x = bx0; x < bx1; ++x) {
      const int c = s_cnt[by * GWP + x];
      lc += c;
      if (c) bits |= 1ull << x;
    }
    if (bits) atomicOr(&rowmask[by], bits);
  }
  int tot_v;
  int run_v = block_excl_scan(lc, s_wave, &tot_v);          // (its barriers also publish the row words)
  for (int x = bx0; x < bx1; ++x) {
    const int c = s_cnt[by * GWP + x];
    s_cnt[by * GWP + x] = run_v;
    run_v += c;
  }
  if (tid == 0) s_cnt[cells] = tot_v;
  const int nout = s_nout;
  // offsets to the nearest occupied cell of each row (the pad column is never read)
  if (by < GW) {
    const unsigned long long m = rowmask[by];
    const int gpt = (GW + 15) >> 4;
    for (int x = bs * gpt; x < min(GW, bs * gpt + gpt); ++x)
      gtab[by * GWP + x] = (signed char)(m ? nearest_bit1(m, x) : SPX_EMPTY);
  }
  __syncthreads();
  SMPLR_TL_STAMP(3);
  // placement by rank
#pragma unroll
  for (int j = 0; j < IPT_MAX; ++j) {
    const int v = tid + j * SF_T;
    int dst = -1;
    if (pc[j] >= 0) dst = s_cnt[pc[j]] + rank[j];
    else if (pc[j] == -1) dst = tot_v + rank[j];
    if (dst >= 0) sRec[dst] = make_float4(pu[j], pv[j], __int_as_float(v), 0.0f);
  }
  __syncthreads();
  SMPLR_TL_STAMP(4);
#ifdef SMPLR_TL
  int tl_k = 0;
  unsigned tl_a = 0, tl_b = 0, tl_c = 0, tl_t0 = 0, tl_t1 = 0, tl_t2 = 0;     // clocks in steps (1)-(2), (3), the ranges
#define SMPLR_TL_CLK(x) x = (unsigned)clock64()
#else
#define SMPLR_TL_CLK(x)
#endif
  // ---- pixels: a wave takes 8 x 8 tiles, handed out through a counter (tiles on the outline cost more)
  const int lane = tid & 63;
  const int tpr = (W + SPX_TILE - 1) / SPX_TILE, ntile = tpr * tpr;
  const int nloc = (ntile - (int)blockIdx.y + (int)gridDim.y - 1) / (int)gridDim.y;
#define SMPLR_SPX_KEY(rec_)                                                                               \
  (((unsigned long long)__float_as_uint(fmaf((rec_).x - fc, (rec_).x - fc, ((rec_).y - fr) * ((rec_).y - fr))) << 32) | \
   (unsigned int)__float_as_int((rec_).z))
  // records [i0_, i1_) (wave-uniform, i0_ < i1_) against every lane's pixel: four broadcast reads in flight per
  // step; a step's surplus slots repeat the range's last record (the same key again: harmless)
#define SMPLR_SPX_RANGE(i0_, i1_)                                                                         \
  for (int i_ = (i0_); i_ < (i1_); i_ += 4) {                                                             \
    const int l_ = (i1_) - 1;                                                                             \
    const float4 ra_ = sRec[i_], rb_ = sRec[min(i_ + 1, l_)], rc_ = sRec[min(i_ + 2, l_)], rd_ = sRec[min(i_ + 3, l_)]; \
    const unsigned long long ka_ = SMPLR_SPX_KEY(ra_), kb_ = SMPLR_SPX_KEY(rb_), kc_ = SMPLR_SPX_KEY(rc_), kd_ = SMPLR_SPX_KEY(rd_); \
    const unsigned long long kab_ = ka_ < kb_ ? ka_ : kb_, kcd_ = kc_ < kd_ ? kc_ : kd_;                  \
    const unsigned long long k4_ = kab_ < kcd_ ? kab_ : kcd_;                                             \
    best = k4_ < best ? k4_ : best;                                                                       \
  }
  for (;;) {
    int t = 0;
    if (lane == 0) t = atomicAdd(&s_next_tile, 1);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= nloc) break;
    const int tile = t * (int)gridDim.y + (int)blockIdx.y;
    const int ty = tile / tpr, tx = tile - ty * tpr;
    const int r_ = ty * SPX_TILE + (lane >> 3), c_ = tx * SPX_TILE + (lane & 7);
    const bool live = r_ < W && c_ < W;
    const int r = min(r_, W - 1), c = min(c_, W - 1);      // clamped lanes repeat a border pixel
    const float fc = (float)c, fr = (float)r;
    const int cx = c + SM, cy = r + SM;
    // this wave's candidate map, one word per cell row.  The lanes talk to each other through it, so every access
    // is an atomic operation to the compiler (with plain accesses it forwards a lane's own zero to its read)
    SMPLR_TL_CLK(tl_t0);
    __hip_atomic_store(&ubm[lane], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    unsigned long long best = own[r * WP + c];             // (d^2 bits << 32) | vertex index; ~0: own cell empty
    unsigned long long rows = 0ull;
    float lim = -1.0f;                                      // squared search radius (< 0: nothing to search)
    float hs = 0.0f;                                        // the hint's score for this pixel (0: none)
    if (hint) hs = hint[((size_t)n * W + (W - 1 - r)) * W + c];
    if (best != ~0ull) {
      // (1) own cell occupied: other cells matter only if the nearest own vertex is farther than half a cell
      const float d2 = __uint_as_float((unsigned int)(best >> 32));
      if (d2 > 0.25f) {
        lim = d2 * 1.000001f;
        rows = 7ull << (cy - 1);                            // rows cy - 1 .. cy + 1 (cy >= SM)
      }
    } else if (hs > 1e-30f && hs <= 1.0f) {
      // (a hint outside (1e-30, 1] - a denormal, a NaN, garbage handed to smplr_silh_fwd_hint - is no hint: -log of it
      // would make the row window overflow and the pixel would come back empty instead of exact; step (2) searches)
      // (2') some vertex lies within -log(hs) of the pixel (+ 1e-3 for the approximate exp / log): every row in reach
      const float b = 1e-3f - __logf(hs);
      lim = b * b * 1.0001f;
      rows = ~0ull;
    } else {
      // (2) nearest occupied cell + the rows that can hold a candidate
      int best2 = 1 << 30, q0 = -1;
      float bound = INFINITY;
      const signed char *gcol = gtab + cx;
      for (int k = 0; k < GW; ++k) {
        const int kk = k * k;
        if ((float)kk > bound) break;
#pragma unroll
        for (int sgn = 0; sgn < 2; ++sgn) {
          const int y = sgn ? cy - k : cy + k;
          if (y < 0 || y >= GW || (sgn && k == 0)) continue;
          const int off = gcol[y * GWP];
          if (off == SPX_EMPTY) continue;
          const int d2 = off * off + kk;
          if (d2 < best2) {
            best2 = d2;
            q0 = y * GWP + cx + off;
            const float rr = __builtin_amdgcn_sqrtf((float)d2) + 1.4143f;
            bound = rr * rr * 1.0001f;
          }
          if ((float)d2 <= bound) rows |= 1ull << y;
        }
      }
      if (q0 >= 0) {
        const float4 r0 = sRec[s_cnt[q0]];                  // any vertex of q0: an upper bound of the answer
        best = SMPLR_SPX_KEY(r0);
        lim = __uint_as_float((unsigned int)(best >> 32)) * 1.000001f;
      } else {
        rows = 0ull;
      }
    }
    SMPLR_TL_CLK(tl_t1);
    // (3) A cell (x, y) is a candidate iff max(|x - cx| - 0.5, 0)^2 + max(|y - cy| - 0.5, 0)^2 <= lim, i.e. row by
    // row |y - cy| <= 0.5 + sqrt(lim) and |x - cx| <= 0.5 + sqrt(lim - dym^2) (1e-4 covers the approximate roots).
    // The lanes OR their intervals into the wave's map; the occupied cells are selected when the map is read back.
    if (lim >= 0.0f) {
      const int yr = (int)(__builtin_amdgcn_sqrtf(lim) + 0.5001f);
      const int ylo = max(0, cy - yr), yhi = min(GW - 1, cy + yr);
      if (rows == ~0ull) {
        // (2'): every row in reach, one after the other: a row whose NEAREST occupied cell (one byte of the offset
        // table) lies outside the row's interval holds no candidate and costs a dozen instructions
        const signed char *g = gtab + ylo * GWP + cx;
        for (int y = ylo; y <= yhi; ++y, g += GWP) {
          const int off = *g;
          const float dym = fmaxf((float)abs(y - cy) - 0.5f, 0.0f);
          const float rem = lim - dym * dym;
          if (rem < 0.0f) continue;
          const int w = (int)(__builtin_amdgcn_sqrtf(rem) + 0.5001f);
          if (abs(off) > w) continue;                       // (an empty row's entry is 127: beyond any radius)
          const int xlo = max(0, cx - w), xhi = min(GW - 1, cx + w);
          unsigned long long m = ~((1ull << xlo) - 1ull);
          if (xhi < 63) m &= (2ull << xhi) - 1ull;
          atomicOr(&ubm[y], m);
        }
      } else {
        unsigned long long keep = ~((1ull << ylo) - 1ull);
        if (yhi < 63) keep &= (2ull << yhi) - 1ull;
        rows &= keep;
        while (rows) {                                      // this lane's rows: its candidate cells into the wave's map
          const int y = __ffsll((long long)rows) - 1;
          rows &= rows - 1ull;
          const float dym = fmaxf((float)abs(y - cy) - 0.5f, 0.0f);
          const float rem = lim - dym * dym;
          if (rem < 0.0f) continue;
          const int w = (int)(__builtin_amdgcn_sqrtf(rem) + 0.5001f);
          const int xlo = max(0, cx - w), xhi = min(GW - 1, cx + w);
          unsigned long long m = ~((1ull << xlo) - 1ull);
          if (xhi < 63) m &= (2ull << xhi) - 1ull;
          atomicOr(&ubm[y], m);
        }
      }
    }
    SMPLR_TL_CLK(tl_t2);
    // LDS operations of one wave execute in order: the map is complete when lane y reads row y's word
    unsigned long long um = __hip_atomic_load(&ubm[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    um &= lane < GW ? rowmask[lane] : 0ull;
    while (__ballot(um != 0ull)) {
      // lane y: the first run of consecutive candidate cells of row y = one contiguous range of records
      int i0 = 0, i1 = 0;
      if (um) {
        const int x0 = __ffsll((long long)um) - 1;
        const unsigned long long inv = ~(um >> x0);
        const int len = inv ? __ffsll((long long)inv) - 1 : 64 - x0;
        um = (x0 + len >= 64) ? 0ull : (um >> (x0 + len)) << (x0 + len);
        const int e = lane * GWP + x0;
        i0 = s_cnt[e];
        i1 = s_cnt[e + len];
      }
      unsigned long long todo = __ballot(i1 > i0);
      while (todo) {                                        // wave-uniform: every lane evaluates every range
        const int l = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const int a0 = __builtin_amdgcn_readlane(i0, l), a1 = __builtin_amdgcn_readlane(i1, l);
        SMPLR_SPX_RANGE(a0, a1)
      }
    }
    if (nout > 0) SMPLR_SPX_RANGE(tot_v, tot_v + nout)       // outliers: always
    if (live) {
      float score = 0.0f;
      int pos = -1;
      if (best != ~0ull) {
        score = expf(-sqrtf(__uint_as_float((unsigned int)(best >> 32))) / 1.2f);
        pos = (int)(best & 0xffffffffull);
      }
      const size_t o = ((size_t)n * W + (W - 1 - r)) * W + c;   // rows flipped (:42)
      *reinterpret_cast<float2 *>(out + o * 2) = make_float2(1.0f - score, score);
      arg_out[o] = pos;
    }
#ifdef SMPLR_TL
    tl_a += tl_t1 - tl_t0; tl_b += tl_t2 - tl_t1; tl_c += (unsigned)clock64() - tl_t2;
    if (tl__ && tl_k < 6) {                                 // per tile: its index and the clock at its end
      tl__[8 + 2 * tl_k] = (unsigned)tile;
      tl__[9 + 2 * tl_k] = (unsigned)clock64();
    }
    ++tl_k;
#endif
  }
  SMPLR_TL_STAMP(5);
#ifdef SMPLR_TL
  if (tl__) { tl__[6] = (unsigned)tl_k; tl__[20] = tl_a; tl__[21] = tl_b; tl__[22] = tl_c; }
#endif
#undef SMPLR_SPX_RANGE
#undef SMPLR_SPX_KEY
#undef SMPLR_TL_CLK
}

// DET: the per-vertex sums as 64-bit fixed point (see seg_flush_det): bit-reproducible whatever the order in which
// the 1 024 threads' pixels reach a vertex' accumulator.
template <bool DET>
__global__ __launch_bounds__(1024) void silh_bwd_kernel(const float *__restrict__ dsilh,
                                                        const float *__restrict__ silh,
                                                        const int *__restrict__ arg,
                                                        const float *__restrict__ proj, int VP, int W,
                                                        float *__restrict__ dproj) {
  // gridDim.y workgroups share a mesh: each owns a contiguous range of VERTICES (its accumulators, its rows of
  // dproj) and walks all the pixels, taking those whose arg-max vertex is its own - with fewer meshes than compute
  // units what there is to spread is the zeroing and the 82 KB of dproj per mesh, the pixel walk is short
  extern __shared__ __attribute__((aligned(16))) float acc[];
  unsigned long long *acc64 = reinterpret_cast<unsigned long long *>(acc);
  __shared__ unsigned s_gmax;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int per = (VP + (int)gridDim.y - 1) / (int)gridDim.y;
  const int v0 = (int)blockIdx.y * per, v1 = min(VP, v0 + per), nv = max(v1 - v0, 0);
  if (DET) {
    for (int i = tid; i < nv * 2; i += 1024) acc64[i] = 0ull;
    if (tid == 0) s_gmax = 0u;
  } else {
    for (int i = tid; i < nv * 2; i += 1024) acc[i] = 0.0f;
  }
  __syncthreads();
  const int npix = W * W;
  float scale = 1.0f, inv_scale = 1.0f;
  if (DET) {
    unsigned gm = 0u;
    for (int i = tid; i < npix * 2; i += 1024) gm = max(gm, __float_as_uint(fabsf(dsilh[(size_t)n * npix * 2 + i])));
    atomicMax(&s_gmax, gm);
    __syncthreads();
    int eg, terms = 1;
    frexpf(__uint_as_float(s_gmax), &eg);
    while ((1 << terms) < npix) ++terms;
    // a term is |g1 - g0| s / 1.2 |du| / d < 2^(1 + eg); a vertex collects at most W^2 <= 2^terms of them
    const int e = min(max(60 - eg - terms, -100), 100);
    scale = ldexpf(1.0f, e);
    inv_scale = ldexpf(1.0f, -e);
  }
  const float *pj = proj + (size_t)n * VP * 3;
  for (int o = tid; o < npix; o += 1024) {
    const size_t po = (size_t)n * npix + o;
    const int v = arg[po];
    if (v < v0 || v >= v1) continue;                       // (-1: no vertex) another workgroup's vertex
    const float g = dsilh[po * 2 + 1] - dsilh[po * 2];
    const float sc = silh[po * 2 + 1];
    const int ro = o / W, cc = o - ro * W;
    const float fr = (float)(W - 1 - ro), fc = (float)cc;
    const float du = pj[v * 3] - fc, dv = pj[v * 3 + 1] - fr;
    const float d = sqrtf(fmaf(du, du, dv * dv));
    const float k = -g * sc / 1.2f;
    if (d > 0.0f && k != 0.0f) {
      const float kk = k / d;
      const int a = (v - v0) * 2;
      if (DET) {
        atomicAdd(&acc64[a], (unsigned long long)__float2ll_rn(kk * du * scale));
        atomicAdd(&acc64[a + 1], (unsigned long long)__float2ll_rn(kk * dv * scale));
      } else {
        atomicAdd(&acc[a], kk * du);
        atomicAdd(&acc[a + 1], kk * dv);
      }
    }
  }
  __syncthreads();
  float *o = dproj + ((size_t)n * VP + v0) * 3;
  for (int i = tid; i < nv * 3; i += 1024) {
    const int v = i / 3, c = i - v * 3;
    if (DET) o[i] = (c < 2) ? (float)(long long)acc64[v * 2 + c] * inv_scale : 0.0f;
    else o[i] = (c < 2) ? acc[v * 2 + c] : 0.0f;
  }
}

// More than 48 KB of dynamic LDS needs the kernel's attribute raised - once per (kernel, device), common.h's memo (one per
// instantiation: the kernel is the template argument)
template <auto Kernel>
static int lds_attr(size_t lds) {
  static LdsAttrMemo memo = {};
  if (lds <= 48 * 1024) return 0;
  return ensure_lds_attr(reinterpret_cast<const void *>(Kernel), lds, &memo, "raster.hip");
}

struct SegWs {
  size_t goff_off, lstart_off, lrec_off, total;
};

// LDS of the binning workgroup: pixel counters [+ z-buffer keys and visible flags with the fused mask] = base, the
// vertex -> slot map = slot, and - when it still fits - every vertex' (u, v) (stage).
struct BinLds { size_t base, slot, total; bool stage; };
static BinLds bin_lds(int VP, int W, int grid_wh /* 0: mask not fused */, bool with_vslot) {
  BinLds b;
  b.base = (size_t)((W * W + 1) & ~1) * sizeof(int);
  if (grid_wh > 0) b.base += (size_t)grid_wh * grid_wh * 8 + (size_t)((VP + 31) / 32) * 4;
  (void)with_vslot;
  b.slot = 0;                                  // (rounds 1-3 staged the vertex -> slot map here; it goes straight to memory now)
  b.stage = b.base + b.slot + (size_t)VP * 8 <= 150 * 1024;
  b.total = b.base + b.slot + (b.stage ? (size_t)VP * 8 : 0);
  return b;
}

// global list (padded per part) + local records + one spare group whose last slot is the header
static int seg_slots(int P, int K) { return ((K + (GP - 1) * P + 3) / 4 * 4) + (K + 3) / 4 * 4 + GP; }

static SegWs seg_ws_layout(int B, int W, int P, int K) {
  SegWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  w.goff_off = take((size_t)B * goff_stride(P) * sizeof(int));   // part offsets [P+1] | unit-weight flag | parts by size [32]
  w.lstart_off = take((size_t)B * ((size_t)W * W + 1) * sizeof(int));
  w.lrec_off = take((size_t)B * K * sizeof(uint2));
  w.total = off;
  return w;
}

}  // namespace smplr

extern "C" {

int smplr_seg_slots(int P, int K) { return (P > 0 && K > 0) ? smplr::seg_slots(P, K) : 0; }

size_t smplr_seg_workspace(int B, int VP, int W, int P, int K) {
  if (B <= 0 || VP <= 0 || W <= 0 || P <= 0 || K <= 0) return 0;
  return smplr::seg_ws_layout(B, W, P, K).total;
}

namespace smplr {
// stage 1: binning (optionally with compute_mask fused in front) -> rec, workspace (part offsets, pixel lists), vslot
static int seg_bin_impl(const char *fn, const float *proj, float *mask, bool fuse_vis, int grid_wh, int ref_compat,
                        int B, int VP, int W, const int32_t *part_pos, const int32_t *part_off, int P, int K,
                        void *workspace, float *rec, int16_t *vslot, void *stream, SkinIn sk = SkinIn{}) {
  SMPLR_REQUIRE(B >= 0 && VP > 0 && VP <= 32767 && W > 0 && W <= 160 && P >= 1 && P <= 31 && K > 0 && K <= BIN_T * IPT_MAX,
                "%s: bad sizes B=%d VP=%d W=%d (max 160) P=%d (max 31) K=%d", fn, B, VP, W, P, K);
  SMPLR_REQUIRE(!fuse_vis || (grid_wh > 0 && grid_wh <= 128), "%s: bad grid_wh=%d (max 128)", fn, grid_wh);
  if (B == 0) return 0;
  const bool skin = sk.v_posed != nullptr;
  // (the skinning form reads v_posed, not proj, and keeps the mask in LDS: there proj, verts and mask are optional outputs)
  SMPLR_REQUIRE((skin || (proj && mask)) && part_pos && part_off && workspace && rec, "%s: null pointer", fn);
  SMPLR_REQUIRE(!skin || (sk.top4 && sk.A && sk.cam && sk.x_stride >= 4 && sk.proj == proj && fuse_vis &&
                          VP <= 7 * BIN_T),
                "%s: the skinning form needs the sparse weights, A, camera rows, the fused mask and V <= %d",
                fn, 7 * BIN_T);
  hipStream_t st = as_stream(stream);
  const SegWs ws = seg_ws_layout(B, W, P, K);
  const int S = seg_slots(P, K);
  char *base = reinterpret_cast<char *>(workspace);
  float4 *G = reinterpret_cast<float4 *>(rec);
  int *goff = reinterpret_cast<int *>(base + ws.goff_off);
  int *lstart = reinterpret_cast<int *>(base + ws.lstart_off);
  uint2 *lrec = reinterpret_cast<uint2 *>(base + ws.lrec_off);
  // LDS: pixel counters | fused mask: z-buffer keys + visible flags | staged (u, v) of every vertex
  const BinLds bl = bin_lds(VP, W, fuse_vis ? grid_wh : 0, vslot != nullptr);
  SMPLR_REQUIRE(bl.base <= 150 * 1024, "%s: pixel counters + grid + flags need %zu B of LDS (max 153600)", fn, bl.base);
  SMPLR_REQUIRE(bl.base + bl.slot <= 150 * 1024, "%s: LDS budget exceeded (%zu B)", fn, bl.base + bl.slot);
  const bool stage = bl.stage;
  const size_t lds = bl.total;
#define SMPLR_BIN_LAUNCH(VIS_, STAGE_, SKIN_)                                                                 \
  {                                                                                                           \
    int rc = lds_attr<&seg_bin_kernel<VIS_, STAGE_, SKIN_>>(lds);          \
    if (rc) return rc;                                                                                        \
    hipLaunchKernelGGL((seg_bin_kernel<VIS_, STAGE_, SKIN_>), dim3(B), dim3(BIN_T), lds, st, proj, mask,      \
                       part_pos, part_off, P, K, VP, W, S, G, goff, lstart, lrec, fuse_vis ? grid_wh : 1,     \
                       ref_compat, reinterpret_cast<short *>(vslot), sk);                                     \
  }
  SMPLR_REQUIRE(!skin || stage, "%s: the skinning form needs the staged (u, v) to fit LDS", fn);
  if (skin) SMPLR_BIN_LAUNCH(true, true, true)
  else if (fuse_vis && stage) SMPLR_BIN_LAUNCH(true, true, false)
  else if (fuse_vis) SMPLR_BIN_LAUNCH(true, false, false)
  else if (stage) SMPLR_BIN_LAUNCH(false, true, false)
  else SMPLR_BIN_LAUNCH(false, false, false)
#undef SMPLR_BIN_LAUNCH
  SMPLR_LAUNCH_CHECK(fn);
  return 0;
}

// stage 2: the pair loop + merge + write-out over a binned workspace
// kernel_ms != NULL: the launch carries start / stop events (hipExtLaunchKernel) and the call WAITS for the kernel and
// returns its own duration - begin to end on the device, what rocprofv3's kernel trace reports, without the dispatch
// gap an event pair around a launch includes.  A measurement aid for bench.py's roofline only.
static int seg_raster_impl(const char *fn, int B, int W, int P, int K, const void *workspace, const float *rec,
                           float *seg, int16_t *arg, void *stream, LossOut lo = LossOut{}, float *kernel_ms = nullptr) {
  SMPLR_REQUIRE(B >= 0 && W > 0 && W <= 160 && P >= 1 && P <= 31 && K > 0 && K <= BIN_T * IPT_MAX,
                "%s: bad sizes B=%d W=%d (max 160) P=%d (max 31) K=%d", fn, B, W, P, K);
  const bool with_loss = lo.loss != nullptr;
  SMPLR_REQUIRE(!with_loss || (P == 31 && lo.gamma >= 0.0f),
                "%s: the loss epilogue is the 32-class head's (P = 31, gamma >= 0): P=%d gamma=%g", fn, P, (double)lo.gamma);
  if (B == 0) return 0;
  SMPLR_REQUIRE(workspace && rec && (seg || with_loss) && arg, "%s: null pointer (seg may be NULL only with a loss)", fn);
  SMPLR_REQUIRE(!with_loss || (lo.labels && lo.stats), "%s: the loss epilogue needs labels and stats", fn);
  const SegWs ws = seg_ws_layout(B, W, P, K);
  const int S = seg_slots(P, K);
  const char *base = reinterpret_cast<const char *>(workspace);
  // SMPLR_RASTER=1: the one-pixel-per-lane kernel of rounds 1-3 (kept for A/B runs); SMPLR_RASTER_SHAPE: the two-pixel
  // kernel's block, 0 = by batch (below), 1 = 128 pair-lanes x 8 part ranges, 2 = 64 x 10, 3 = 128 x 4
  static const int version = getenv("SMPLR_RASTER") ? atoi(getenv("SMPLR_RASTER")) : 2;
  static const int shape_env = getenv("SMPLR_RASTER_SHAPE") ? atoi(getenv("SMPLR_RASTER_SHAPE")) : 0;
  if (version != 1) {
    const int shape = shape_env ? shape_env : raster2_shape(B, W, K);
    const int pl = shape == 2 ? 64 : PLN;
    const int nl = ((W + 1) / 2) * W, nt2 = (nl + pl - 1) / pl;
    const int grid2 = 8 * ((B + 7) / 8) * nt2;
    const unsigned wm = (unsigned)(((1u << 24) + W - 1) / W);
    const float4 *Gp = reinterpret_cast<const float4 *>(rec);
    const int *goffp = reinterpret_cast<const int *>(base + ws.goff_off);
    const int *lsp = reinterpret_cast<const int *>(base + ws.lstart_off);
    const uint2 *lrp = reinterpret_cast<const uint2 *>(base + ws.lrec_off);
    short *argp = reinterpret_cast<short *>(arg);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (kernel_ms) {
      SMPLR_HIP(hipEventCreate(&e0));
      SMPLR_HIP(hipEventCreate(&e1));
    }
#define SMPLR_RASTER2_LAUNCH(LOSS_, NG_, PL_)                                                                       \
  {                                                                                                                 \
    if (kernel_ms)                                                                                                  \
      hipExtLaunchKernelGGL((raster2_fwd_kernel<LOSS_, NG_, PL_>), dim3(grid2), dim3(PL_ * NG_), 0, as_stream(stream), \
                            e0, e1, 0, Gp, goffp, lsp, lrp, P, K, S, W, B, nt2, seg, argp, wm, lo);                 \
    else                                                                                                            \
      hipLaunchKernelGGL((raster2_fwd_kernel<LOSS_, NG_, PL_>), dim3(grid2), dim3(PL_ * NG_), 0, as_stream(stream), \
                         Gp, goffp, lsp, lrp, P, K, S, W, B, nt2, seg, argp, wm, lo);                               \
  }
#define SMPLR_RASTER2_SHAPES(LOSS_)                                                                                 \
  {                                                                                                                 \
    if (shape == 2) SMPLR_RASTER2_LAUNCH(LOSS_, 10, 64)                                                             \
    else if (shape == 3) SMPLR_RASTER2_LAUNCH(LOSS_, 4, 128)                                                        \
    else SMPLR_RASTER2_LAUNCH(LOSS_, 8, 128)                                                                        \
  }
    if (with_loss) SMPLR_RASTER2_SHAPES(true) else SMPLR_RASTER2_SHAPES(false)
#undef SMPLR_RASTER2_SHAPES
#undef SMPLR_RASTER2_LAUNCH
    SMPLR_LAUNCH_CHECK(fn);
    if (kernel_ms) {
      SMPLR_HIP(hipEventSynchronize(e1));
      SMPLR_HIP(hipEventElapsedTime(kernel_ms, e0, e1));
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
    }
    return 0;
  }
  const int ntiles = (W * W + RTS - 1) / RTS;
  const int grid = 8 * ((B + 7) / 8) * ntiles;
#define SMPLR_RASTER_LAUNCH(LOSS_)                                                                             \
  hipLaunchKernelGGL(raster_fwd_kernel<LOSS_>, dim3(grid), dim3(RTS * NG), 0, as_stream(stream),               \
                     reinterpret_cast<const float4 *>(rec), reinterpret_cast<const int *>(base + ws.goff_off), \
                     reinterpret_cast<const int *>(base + ws.lstart_off),                                      \
                     reinterpret_cast<const uint2 *>(base + ws.lrec_off), P, K, S, W, B, ntiles, seg,          \
                     reinterpret_cast<short *>(arg), (unsigned)(((1u << 24) + W - 1) / W), lo)
  if (kernel_ms) {
    hipEvent_t e0, e1;
    SMPLR_HIP(hipEventCreate(&e0));
    SMPLR_HIP(hipEventCreate(&e1));
    hipExtLaunchKernelGGL(raster_fwd_kernel<false>, dim3(grid), dim3(RTS * NG), 0, as_stream(stream), e0, e1, 0,
                          reinterpret_cast<const float4 *>(rec), reinterpret_cast<const int *>(base + ws.goff_off),
                          reinterpret_cast<const int *>(base + ws.lstart_off),
                          reinterpret_cast<const uint2 *>(base + ws.lrec_off), P, K, S, W, B, ntiles, seg,
                          reinterpret_cast<short *>(arg), (unsigned)(((1u << 24) + W - 1) / W), lo);
    SMPLR_LAUNCH_CHECK(fn);
    SMPLR_HIP(hipEventSynchronize(e1));
    SMPLR_HIP(hipEventElapsedTime(kernel_ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
  }
  if (with_loss) SMPLR_RASTER_LAUNCH(true);
  else SMPLR_RASTER_LAUNCH(false);
#undef SMPLR_RASTER_LAUNCH
  SMPLR_LAUNCH_CHECK(fn);
  return 0;
}

static int seg_fwd_impl(const char *fn, const float *proj, float *mask, bool fuse_vis, int grid_wh, int ref_compat,
                        int B, int VP, int W, const int32_t *part_pos, const int32_t *part_off, int P, int K,
                        void *workspace, float *seg, int16_t *arg, float *rec, int16_t *vslot, void *stream) {
  SMPLR_REQUIRE(B == 0 || (seg && arg), "%s: null pointer", fn);
  int rc = seg_bin_impl(fn, proj, mask, fuse_vis, grid_wh, ref_compat, B, VP, W, part_pos, part_off, P, K, workspace,
                        rec, vslot, stream);
  if (rc) return rc;
  return seg_raster_impl(fn, B, W, P, K, workspace, rec, seg, arg, stream);
}
}  // namespace smplr

int smplr_seg_bin(const float *proj, float *mask, int B, int VP, int W, int grid_wh, int ref_compat,
                  const int32_t *part_pos, const int32_t *part_off, int P, int K, void *workspace, float *rec,
                  int16_t *vslot, void *stream) {
  return smplr::seg_bin_impl("smplr_seg_bin", proj, mask, grid_wh > 0, grid_wh, ref_compat, B, VP, W, part_pos,
                             part_off, P, K, workspace, rec, vslot, stream);
}

int smplr_seg_raster(int B, int W, int P, int K, const void *workspace, const float *rec, float *seg, int16_t *arg,
                     void *stream) {
  return smplr::seg_raster_impl("smplr_seg_raster", B, W, P, K, workspace, rec, seg, arg, stream);
}

int smplr_seg_fwd(const float *proj, const float *mask, int B, int VP, int W, const int32_t *part_pos,
                  const int32_t *part_off, int P, int K, void *workspace, float *seg, int16_t *arg,
                  float *rec, int16_t *vslot, void *stream) {
  return smplr::seg_fwd_impl("smplr_seg_fwd", proj, const_cast<float *>(mask), false, 0, 0, B, VP, W, part_pos,
                             part_off, P, K, workspace, seg, arg, rec, vslot, stream);
}

int smplr_vis_seg_fwd(const float *proj, int B, int VP, int W, int grid_wh, int ref_compat,
                      const int32_t *part_pos, const int32_t *part_off, int P, int K, void *workspace,
                      float *mask, float *seg, int16_t *arg, float *rec, int16_t *vslot, void *stream) {
  return smplr::seg_fwd_impl("smplr_vis_seg_fwd", proj, mask, true, grid_wh, ref_compat, B, VP, W, part_pos,
                             part_off, P, K, workspace, seg, arg, rec, vslot, stream);
}

int smplr_skin_vis_seg_fits(int V, int W, int grid_wh) {
  if (V <= 0 || V > 7 * smplr::BIN_T || W <= 0 || W > 160 || grid_wh <= 0 || grid_wh > 128) return 0;
  const smplr::BinLds b = smplr::bin_lds(V, W, grid_wh, true);
  return (b.base + b.slot <= 150 * 1024 && b.stage) ? 1 : 0;
}

int smplr_skin_vis_seg_fwd(const float *v_posed, const float *lbs_top4, const float *A, const float *cam, int x_stride,
                           int B, int V, int W, int grid_wh, int ref_compat, const int32_t *part_pos,
                           const int32_t *part_off, int P, int K, void *workspace, float *verts, float *proj,
                           float *mask, float *seg, int16_t *arg, float *rec, int16_t *vslot, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B <= 0 || (v_posed && lbs_top4 && A && cam), "smplr_skin_vis_seg_fwd: null pointer");
  const SkinIn sk{v_posed, lbs_top4, A, cam, x_stride, verts, proj};
  int rc = seg_bin_impl("smplr_skin_vis_seg_fwd", proj, mask, true, grid_wh, ref_compat, B, V, W, part_pos, part_off, P,
                        K, workspace, rec, vslot, stream, sk);
  if (rc) return rc;
  return seg_raster_impl("smplr_skin_vis_seg_fwd", B, W, P, K, workspace, rec, seg, arg, stream);
}

int smplr_seg_raster_timed(int B, int W, int P, int K, const void *workspace, const float *rec, float *seg, int16_t *arg,
                           float *kernel_ms, void *stream) {
  SMPLR_REQUIRE(kernel_ms != nullptr, "smplr_seg_raster_timed: null kernel_ms");
  return smplr::seg_raster_impl("smplr_seg_raster_timed", B, W, P, K, workspace, rec, seg, arg, stream, smplr::LossOut{},
                                kernel_ms);
}

int smplr_seg_raster_plan(int B, int W, int P, int K, int32_t *info, int32_t *tile_records) {
  using namespace smplr;
  if (B <= 0 || W <= 0 || W > 160 || P < 1 || P > 31 || K <= 0 || K > BIN_T * IPT_MAX) return 0;
  static const int shape_env = getenv("SMPLR_RASTER_SHAPE") ? atoi(getenv("SMPLR_RASTER_SHAPE")) : 0;
  const int shape = shape_env ? shape_env : raster2_shape(B, W, K);
  const int pl = shape == 2 ? 64 : PLN, ng = shape == 2 ? 10 : shape == 3 ? 4 : 8;
  const int ar = pl == 64 ? 6144 : ARENA;                  // raster2_fwd_kernel's AR
  const int nq = (W + 1) / 2, nl = nq * W, nt = (nl + pl - 1) / pl;
  int tmax = 0, tmin = 1 << 30, tall = 0;
  for (int t = 0; t < nt; ++t) {                           // the kernel's own row count per tile
    const int Qf = std::min(t * pl, nl - 1) / W, Ql = std::min(t * pl + pl - 1, nl - 1) / W;
    const int nrows = 2 * (Ql - Qf + 1);
    const int Rb = nrows <= 4 ? 5 : nrows <= 6 ? 7 : nrows <= 8 ? 9 : 11;
    const int tr = nrows <= R2_MAX - 1 ? trec_of(ar, Rb) : 0;
    if (tile_records) tile_records[t] = tr;
    if (tr) { tmax = std::max(tmax, tr); tmin = std::min(tmin, tr); } else tall = 1;
  }
  if (info) {
    info[0] = pl; info[1] = ng; info[2] = nt; info[3] = goff_stride(P);
    info[4] = tmax; info[5] = tmin == (1 << 30) ? 0 : tmin; info[6] = tall; info[7] = 0;
  }
  return nt;
}

int smplr_seg_raster_ex(int B, int W, int P, int K, const void *workspace, const float *rec, const int32_t *labels,
                        const float *class_w, float gamma, float *seg, int16_t *arg, float *loss, float *stats,
                        float *vmax, void *stream) {
  return smplr::seg_raster_impl("smplr_seg_raster_ex", B, W, P, K, workspace, rec, seg, arg, stream,
                                smplr::LossOut{labels, class_w, gamma, loss, reinterpret_cast<float4 *>(stats), vmax});
}

int smplr_skin_vis_seg_fwd_ex(const float *v_posed, const float *lbs_top4, const float *A, const float *cam,
                              int x_stride, int B, int V, int W, int grid_wh, int ref_compat, const int32_t *part_pos,
                              const int32_t *part_off, int P, int K, void *workspace, const int32_t *labels,
                              const float *class_w, float gamma, float *verts, float *proj, float *mask, float *seg,
                              int16_t *arg, float *rec, int16_t *vslot, float *loss, float *stats, float *vmax,
                              void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B <= 0 || (v_posed && lbs_top4 && A && cam), "smplr_skin_vis_seg_fwd_ex: null pointer");
  const SkinIn sk{v_posed, lbs_top4, A, cam, x_stride, verts, proj};
  int rc = seg_bin_impl("smplr_skin_vis_seg_fwd_ex", proj, mask, true, grid_wh, ref_compat, B, V, W, part_pos, part_off,
                        P, K, workspace, rec, vslot, stream, sk);
  if (rc) return rc;
  return seg_raster_impl("smplr_skin_vis_seg_fwd_ex", B, W, P, K, workspace, rec, seg, arg, stream,
                         LossOut{labels, class_w, gamma, loss, reinterpret_cast<float4 *>(stats), vmax});
}

int smplr_seg_bwd_nsplit(int B, int W) {
  if (B <= 0 || W <= 0) return 0;
  const int rows = smplr::seg_bwd_rows(B, W);
  return (W + rows - 1) / rows;
}

size_t smplr_seg_bwd_workspace(int B, int W) {
  if (B <= 0 || W <= 0) return 0;
  const int nsplit = smplr_seg_bwd_nsplit(B, W);
  return (size_t)B * nsplit * smplr::SB_NWIN * smplr::SB_SLOTS * 2 * sizeof(float);
}

namespace smplr {
static int seg_bwd_impl(const char *fn, const float *dseg, LossIn li, const int16_t *arg, const float *rec, int B, int VP,
                        int W, int P, int K, float *dproj, void *workspace, int deterministic, void *stream) {
  SMPLR_REQUIRE(B >= 0 && VP > 0 && VP <= 32767 && W > 0 && W <= 160 && P >= 1 && P <= 31 && K > 0 && K <= 16000,
                "%s: bad sizes B=%d VP=%d W=%d P=%d K=%d", fn, B, VP, W, P, K);
  const bool with_loss = li.dloss != nullptr;
  SMPLR_REQUIRE(!with_loss || P == 31, "%s: the loss head has 32 classes (P = 31), not P=%d", fn, P);
  if (B == 0) return 0;
  SMPLR_REQUIRE((with_loss ? li.stats != nullptr : dseg != nullptr) && arg && rec && workspace, "%s: null pointer", fn);
  hipStream_t st = as_stream(stream);
  const int rows = seg_bwd_rows(B, W), nsplit = (W + rows - 1) / rows;
  const int S = seg_slots(P, K);
  SMPLR_REQUIRE(S <= SB_NWIN * SB_SLOTS, "%s: %d record slots exceed %d", fn, S, SB_NWIN * SB_SLOTS);
#define SMPLR_SEGBWD_LAUNCH(DET_, LOSS_, lds_)                                                                       \
  hipLaunchKernelGGL((seg_bwd_kernel<DET_, LOSS_>), dim3(nsplit, B), dim3(32 * rows), lds_, st, dseg,                 \
                     reinterpret_cast<const short *>(arg), reinterpret_cast<const float4 *>(rec), S, VP, W, P, dproj, \
                     reinterpret_cast<float *>(workspace), rows, li, pipe)
  static const int pipe = getenv("SMPLR_SEGBWD_PIPE") ? atoi(getenv("SMPLR_SEGBWD_PIPE")) : 1;   // 0: the unpipelined row walk (A/B runs)
  if (deterministic) {
    const size_t lds = (size_t)SB_SLOTS * 2 * sizeof(unsigned long long);
    int rc = with_loss ? lds_attr<&seg_bwd_kernel<true, true>>(lds)
                       : lds_attr<&seg_bwd_kernel<true, false>>(lds);
    if (rc) return rc;
    if (with_loss) SMPLR_SEGBWD_LAUNCH(true, true, lds);
    else SMPLR_SEGBWD_LAUNCH(true, false, lds);
  } else {
    const size_t lds = (size_t)SB_SLOTS * 2 * sizeof(float);
    if (with_loss) SMPLR_SEGBWD_LAUNCH(false, true, lds);
    else SMPLR_SEGBWD_LAUNCH(false, false, lds);
  }
#undef SMPLR_SEGBWD_LAUNCH
  SMPLR_LAUNCH_CHECK(fn);
  if (!dproj) return 0;                        // slot sums only: smplr_smpl_bwd gathers them by vertex
  hipLaunchKernelGGL(seg_bwd_merge_kernel, dim3(SB_SLOTS / 256, B), dim3(256), 0, st,
                     reinterpret_cast<const float *>(workspace), reinterpret_cast<const float4 *>(rec), S, VP, nsplit,
                     dproj);
  SMPLR_LAUNCH_CHECK(fn);
  return 0;
}
}  // namespace smplr

int smplr_seg_bwd(const float *dseg, const int16_t *arg, const float *rec, int B, int VP, int W, int P, int K,
                  float *dproj, void *workspace, int deterministic, void *stream) {
  return smplr::seg_bwd_impl("smplr_seg_bwd", dseg, smplr::LossIn{nullptr, nullptr}, arg, rec, B, VP, W, P, K, dproj,
                             workspace, deterministic, stream);
}

int smplr_seg_loss_bwd(const float *dloss, const float *stats, const int16_t *arg, const float *rec, int B, int VP, int W,
                       int P, int K, float *dproj, void *workspace, int deterministic, void *stream) {
  SMPLR_REQUIRE(B <= 0 || dloss, "smplr_seg_loss_bwd: null dloss");
  return smplr::seg_bwd_impl("smplr_seg_loss_bwd", nullptr, smplr::LossIn{dloss, reinterpret_cast<const float4 *>(stats)},
                             arg, rec, B, VP, W, P, K, dproj, workspace, deterministic, stream);
}

size_t smplr_silh_workspace(int B, int VP, int W) {
  if (B <= 0 || VP <= 0 || W <= 0) return 0;
  const int KP = (VP + smplr::CH - 1) / smplr::CH * smplr::CH;
  return (size_t)B * KP * 4 * sizeof(float);     // only the brute-force fallback uses it
}

int smplr_silh_fwd(const float *proj, int B, int VP, int W, float *silh, int32_t *arg, void *workspace,
                   void *stream) {
  return smplr_silh_fwd_hint(proj, nullptr, B, VP, W, silh, arg, workspace, stream);
}

int smplr_silh_fwd_hint(const float *proj, const float *hint, int B, int VP, int W, float *silh, int32_t *arg,
                        void *workspace, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && VP > 0 && W > 0 && W <= 1024, "smplr_silh_fwd: bad sizes B=%d VP=%d W=%d", B, VP, W);
  if (B == 0) return 0;
  SMPLR_REQUIRE(proj && silh && arg && workspace, "smplr_silh_fwd: null pointer");
  hipStream_t st = as_stream(stream);
  if (W + 2 * SM <= 64 && VP <= SF_T * IPT_MAX && silh_px_layout(VP, W).total <= 159 * 1024) {
    const SpxLds L = silh_px_layout(VP, W);
    const int nsplit = B >= 256 ? 1 : (B >= 128 ? 2 : 4);      // one workgroup per CU (256 CUs)
    int rc = lds_attr<&silh_px_kernel>(L.total);
    if (rc) return rc;
    hipLaunchKernelGGL(silh_px_kernel, dim3(B, nsplit), dim3(SF_T), L.total, st, proj, VP, W, L, silh, arg, hint);
    SMPLR_LAUNCH_CHECK("smplr_silh_fwd");
    return 0;
  }
  if (W <= SILH_WMAX && VP <= SF_T * IPT_MAX && silh_fused_lds(VP, W) <= 150 * 1024) {
    const size_t lds = silh_fused_lds(VP, W);
    const int nsplit = B >= 256 ? 1 : (B >= 128 ? 2 : 4);      // one workgroup per CU (256 CUs)
    if (W + 2 * SM <= 64) {
      int rc = lds_attr<&silh_fused_kernel<true>>(lds);
      if (rc) return rc;
      hipLaunchKernelGGL(silh_fused_kernel<true>, dim3(B, nsplit), dim3(SF_T), lds, st, proj, VP, W, silh, arg);
    } else {
      int rc = lds_attr<&silh_fused_kernel<false>>(lds);
      if (rc) return rc;
      hipLaunchKernelGGL(silh_fused_kernel<false>, dim3(B, nsplit), dim3(SF_T), lds, st, proj, VP, W, silh, arg);
    }
    SMPLR_LAUNCH_CHECK("smplr_silh_fwd");
    return 0;
  }
  const int KP = (VP + CH - 1) / CH * CH;
  hipLaunchKernelGGL(silh_prep_kernel, dim3((KP + 255) / 256, B), dim3(256), 0, st, proj, VP, KP,
                     reinterpret_cast<float4 *>(workspace));
  SMPLR_LAUNCH_CHECK("smplr_silh_fwd(prep)");
  hipLaunchKernelGGL(silh_fwd_kernel, dim3((W * W + RT - 1) / RT, B), dim3(RT), 0, st,
                     reinterpret_cast<const float4 *>(workspace), KP, W, silh, arg);
  SMPLR_LAUNCH_CHECK("smplr_silh_fwd");
  return 0;
}

int smplr_silh_bwd(const float *dsilh, const float *silh, const int32_t *arg, const float *proj, int B,
                   int VP, int W, float *dproj, int deterministic, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && VP > 0 && W > 0 && W <= 1024, "smplr_silh_bwd: bad sizes B=%d VP=%d W=%d", B, VP, W);
  if (B == 0) return 0;
  SMPLR_REQUIRE(dsilh && silh && arg && proj && dproj, "smplr_silh_bwd: null pointer");
  const int nsplit = B >= 512 ? 1 : (B >= 128 ? 2 : 4);      // workgroups per mesh (vertex ranges)
  const int per = (VP + nsplit - 1) / nsplit;
  const size_t lds = (size_t)per * 2 * (deterministic ? sizeof(unsigned long long) : sizeof(float));
  SMPLR_REQUIRE(lds <= 150 * 1024, "smplr_silh_bwd: VP=%d needs %zu B of LDS", VP, lds);
  if (deterministic) {
    int rc = lds_attr<&silh_bwd_kernel<true>>(lds);
    if (rc) return rc;
    hipLaunchKernelGGL(silh_bwd_kernel<true>, dim3(B, nsplit), dim3(1024), lds, as_stream(stream), dsilh, silh, arg, proj,
                       VP, W, dproj);
  } else {
    int rc = lds_attr<&silh_bwd_kernel<false>>(lds);
    if (rc) return rc;
    hipLaunchKernelGGL(silh_bwd_kernel<false>, dim3(B, nsplit), dim3(1024), lds, as_stream(stream), dsilh, silh, arg, proj,
                       VP, W, dproj);
  }
  SMPLR_LAUNCH_CHECK("smplr_silh_bwd");
  return 0;
}

}  // extern "C"

#ifdef SMPLR_TL
SMPLR_TL_EXPORT(bin, smplr::g_tl_bin, smplr::TL_BIN_WG * (smplr::BIN_T / 64) * 32)
SMPLR_TL_EXPORT(segbwd, smplr::g_tl_segbwd, smplr::TL_SEGBWD_WG * 12 * 32)
SMPLR_TL_EXPORT(raster, smplr::g_tl_raster, smplr::TL_RASTER_WG * 16 * 32)
SMPLR_TL_EXPORT(silhpx, smplr::g_tl_silhpx, smplr::TL_SILHPX_WG * 16 * 32)
#endif
