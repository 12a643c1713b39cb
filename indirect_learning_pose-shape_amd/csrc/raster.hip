// K6/K7 soft rasterisers: 31-part segmentation and silhouette, forward and backward.
//
// Reference: keras_smpl/projects_to_seg.py:34-69 and keras_smpl/projects_to_silhouette.py:20-42.
// The reference materialises (N, W^2, n_p, 2) tiles per part and takes max_v exp(-m_v d_v);
// since exp is monotone that is exp(-min_v m_v d_v): a masked nearest-vertex search.
//
// Forward: a lane owns one pixel, a workgroup 256 consecutive pixels of one mesh.  The mesh's
// vertices are first gathered part-major into (u, v, m^2, m) records (`sorted`, each part
// padded to a multiple of 8 with +inf sentinels), so the pair loop reads wave-uniform records
// (scalar loads, no LDS, no vector memory) and spends its time on the fp32 VALU:
// key = m^2 * ((u-c)^2 + (v-r)^2), running minimum per 8-vertex chunk; the winning chunk is
// re-evaluated once per (pixel, part) to recover the first arg-min, and the score is
// exp(-m * sqrt(d2)) of that vertex.  Scores and arg-mins are transposed through LDS so the
// NHWC outputs are written as whole 128-B pixel rows.
//
// Backward: lanes = (pixel, channel) exactly as the NHWC tensors lie in memory (coalesced,
// and neighbouring lanes hit different parts, hence different vertices); per-mesh gradient
// accumulation in LDS (V' x 2 floats) with ds_add_f32, one plain store pass at the end.
#include "common.h"

namespace smplr {

constexpr int CH = SMPLR_CHUNK;   // 8
constexpr int RT = 256;           // pixels (threads) per block
constexpr int SLD = 33;           // LDS transpose stride

// sorted[n][k] = (u, v, m*m, m) for slot k; pos < 0 -> sentinel (+inf, +inf, 1, 1).
__global__ __launch_bounds__(256) void seg_prep_kernel(const float *__restrict__ proj,
                                                       const float *__restrict__ mask,
                                                       const int *__restrict__ part_pos, int VP, int KP,
                                                       float4 *__restrict__ sorted) {
  const int n = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  if (k >= KP) return;
  const int pos = part_pos ? part_pos[k] : (k < VP ? k : -1);
  float4 o;
  if (pos < 0) {
    o = make_float4(INFINITY, INFINITY, 1.0f, 1.0f);
  } else {
    const float *p = proj + ((size_t)n * VP + pos) * 3;
    const float m = mask ? mask[(size_t)n * VP + pos] : 1.0f;
    o = make_float4(p[0], p[1], m * m, m);
  }
  sorted[(size_t)n * KP + k] = o;
}

__device__ __forceinline__ float pair_key(const float4 a, float fc, float fr) {
  const float du = a.x - fc, dv = a.y - fr;
  return fmaf(du, du, dv * dv) * a.z;
}

// SILH = false: P parts, outputs seg (B,W,W,P+1) + arg (B,W,W,32) int16.
// SILH = true : one part = all vertices, outputs silh (B,W,W,2) + arg (B,W,W) int32.
template <bool SILH>
__global__ __launch_bounds__(RT) void raster_fwd_kernel(const float4 *__restrict__ sorted,
                                                        const int *__restrict__ part_pos,
                                                        const int *__restrict__ part_off, int P, int KP,
                                                        int W, float *__restrict__ out,
                                                        void *__restrict__ arg_out) {
  __shared__ float sS[SILH ? 1 : (RT / 64) * 64 * SLD];
  __shared__ short sA[SILH ? 1 : (RT / 64) * 64 * SLD];
  const int n = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q = blockIdx.x * RT + tid;
  const int npix = W * W;
  const bool live = q < npix;
  const int r = q / W, c = q - r * W;
  const float fc = (float)c, fr = (float)r;
  const float4 *S = sorted + (size_t)n * KP;
  const int C = P + 1;

  float sum = 0.0f;
  for (int p = 0; p < P; ++p) {
    const int beg = SILH ? 0 : part_off[p], end = SILH ? KP : part_off[p + 1];
    float best = INFINITY;
    int bestk = beg;
    for (int k = beg; k < end; k += CH) {
      float cm = pair_key(S[k], fc, fr);
#pragma unroll
      for (int j = 1; j < CH; ++j) cm = fminf(cm, pair_key(S[k + j], fc, fr));
      if (cm < best) { best = cm; bestk = k; }
    }
    // resolve the first arg-min inside the winning chunk (per-lane addresses: vector loads)
    int sel = 0;
    float4 win = make_float4(INFINITY, INFINITY, 1.0f, 1.0f);
    if (beg < end) {
#pragma unroll
      for (int j = CH - 1; j >= 0; --j) {
        const float4 a = S[bestk + j];
        if (pair_key(a, fc, fr) == best) { sel = j; win = a; }
      }
    }
    const float du = win.x - fc, dv = win.y - fr;
    const float d = sqrtf(fmaf(du, du, dv * dv));
    int pos = SILH ? (bestk + sel) : part_pos[bestk + sel];
    float score;
    if (SILH) {
      score = expf(-d / 1.2f);
    } else {
      score = expf(-(d * win.w));
    }
    if (!(best < INFINITY)) { score = 0.0f; pos = 0; }   // empty part / all sentinels / NaN
    if (SILH) {
      if (live) {
        const size_t o = ((size_t)n * W + (W - 1 - r)) * W + c;   // rows flipped (:42)
        out[o * 2 + 0] = 1.0f - score;
        out[o * 2 + 1] = score;
        reinterpret_cast<int *>(arg_out)[o] = pos;
      }
    } else {
      sum += score;
      sS[(wave * 64 + lane) * SLD + 1 + p] = score;
      sA[(wave * 64 + lane) * SLD + p] = (short)pos;
    }
  }
  if (SILH) return;
  // background = 1 - clip(sum, 0, 1) (:61-64); gate = clip passes gradient (0 <= sum <= 1)
  sS[(wave * 64 + lane) * SLD + 0] = 1.0f - fminf(fmaxf(sum, 0.0f), 1.0f);
  for (int p = P; p < 31; ++p) sA[(wave * 64 + lane) * SLD + p] = 0;
  sA[(wave * 64 + lane) * SLD + 31] = (sum >= 0.0f && sum <= 1.0f) ? 1 : 0;
  // wave-private tiles: no block barrier needed, but order LDS writes before the reads
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  const int q0 = blockIdx.x * RT + wave * 64;
  short *arg = reinterpret_cast<short *>(arg_out);
  for (int e = lane; e < 64 * 32; e += 64) {
    const int pl = e >> 5, ch = e & 31;
    const int qq = q0 + pl;
    if (qq < npix) {
      const int rr = qq / W, cc = qq - rr * W;
      const size_t o = ((size_t)n * W + (W - 1 - rr)) * W + cc;   // rows flipped (:68)
      if (ch < C) out[o * C + ch] = sS[(wave * 64 + pl) * SLD + ch];
      arg[o * 32 + ch] = sA[(wave * 64 + pl) * SLD + ch];
    }
  }
}

// dproj (B,VP,3); one block per mesh, LDS accumulators acc[VP*2].
__global__ __launch_bounds__(1024) void seg_bwd_kernel(const float *__restrict__ dseg,
                                                       const float *__restrict__ seg,
                                                       const short *__restrict__ arg,
                                                       const float *__restrict__ proj,
                                                       const float *__restrict__ mask, int VP, int W, int P,
                                                       float *__restrict__ dproj) {
  extern __shared__ float acc[];   // VP*2
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < VP * 2; i += 1024) acc[i] = 0.0f;
  __syncthreads();
  const int C = P + 1, npix = W * W;
  const float *pj = proj + (size_t)n * VP * 3;
  const float *mk = mask + (size_t)n * VP;
  // element e = (pixel, slot) over 32 slots per pixel; slot s<P is part s (channel s+1)
  for (int e = tid; e < npix * 32; e += 1024) {
    const int o = e >> 5, s = e & 31;        // o = output pixel index (flipped row-major)
    const size_t po = (size_t)n * npix + o;
    const short a = arg[po * 32 + s];
    const int gate = __shfl((int)a, 31, 32);     // slot 31 of this pixel (32-lane groups)
    if (s < P) {
      const float g0 = dseg[po * C];
      const float g = dseg[po * C + 1 + s] - (gate ? g0 : 0.0f);
      const float sc = seg[po * C + 1 + s];
      const int ro = o / W, cc = o - ro * W;
      const float fr = (float)(W - 1 - ro), fc = (float)cc;
      const int v = (unsigned short)a;
      const float du = pj[v * 3] - fc, dv = pj[v * 3 + 1] - fr;
      const float d = sqrtf(fmaf(du, du, dv * dv));
      const float k = -g * sc * mk[v];
      if (d > 0.0f && k != 0.0f) {
        const float kk = k / d;
        atomicAdd(&acc[v * 2], kk * du);
        atomicAdd(&acc[v * 2 + 1], kk * dv);
      }
    }
  }
  __syncthreads();
  float *o = dproj + (size_t)n * VP * 3;
  for (int i = tid; i < VP * 3; i += 1024) {
    const int v = i / 3, c = i - v * 3;
    o[i] = (c < 2) ? acc[v * 2 + c] : 0.0f;
  }
}

__global__ __launch_bounds__(1024) void silh_bwd_kernel(const float *__restrict__ dsilh,
                                                        const float *__restrict__ silh,
                                                        const int *__restrict__ arg,
                                                        const float *__restrict__ proj, int VP, int W,
                                                        float *__restrict__ dproj) {
  extern __shared__ float acc[];
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < VP * 2; i += 1024) acc[i] = 0.0f;
  __syncthreads();
  const int npix = W * W;
  const float *pj = proj + (size_t)n * VP * 3;
  for (int o = tid; o < npix; o += 1024) {
    const size_t po = (size_t)n * npix + o;
    const float g = dsilh[po * 2 + 1] - dsilh[po * 2];
    const float sc = silh[po * 2 + 1];
    const int v = arg[po];
    const int ro = o / W, cc = o - ro * W;
    const float fr = (float)(W - 1 - ro), fc = (float)cc;
    const float du = pj[v * 3] - fc, dv = pj[v * 3 + 1] - fr;
    const float d = sqrtf(fmaf(du, du, dv * dv));
    const float k = -g * sc / 1.2f;
    if (d > 0.0f && k != 0.0f) {
      const float kk = k / d;
      atomicAdd(&acc[v * 2], kk * du);
      atomicAdd(&acc[v * 2 + 1], kk * dv);
    }
  }
  __syncthreads();
  float *o = dproj + (size_t)n * VP * 3;
  for (int i = tid; i < VP * 3; i += 1024) {
    const int v = i / 3, c = i - v * 3;
    o[i] = (c < 2) ? acc[v * 2 + c] : 0.0f;
  }
}

static int set_lds_attr(const void *fn, size_t lds) {
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return (int)e;
    }
  }
  return 0;
}

}  // namespace smplr

extern "C" {

int smplr_seg_fwd(const float *proj, const float *mask, int B, int VP, int W, const int32_t *part_pos,
                  const int32_t *part_off, int P, int KP, float *sorted, float *seg, int16_t *arg,
                  void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && VP > 0 && VP <= 32767 && W > 0 && W <= 1024 && P >= 1 && P <= 31 && KP > 0 &&
                    KP % CH == 0,
                "smplr_seg_fwd: bad sizes B=%d VP=%d W=%d P=%d KP=%d", B, VP, W, P, KP);
  if (B == 0) return 0;
  SMPLR_REQUIRE(proj && mask && part_pos && part_off && sorted && seg && arg, "smplr_seg_fwd: null pointer");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(seg_prep_kernel, dim3((KP + 255) / 256, B), dim3(256), 0, st, proj, mask, part_pos, VP,
                     KP, reinterpret_cast<float4 *>(sorted));
  SMPLR_LAUNCH_CHECK("smplr_seg_fwd(prep)");
  hipLaunchKernelGGL(raster_fwd_kernel<false>, dim3((W * W + RT - 1) / RT, B), dim3(RT), 0, st,
                     reinterpret_cast<const float4 *>(sorted), part_pos, part_off, P, KP, W, seg,
                     reinterpret_cast<void *>(arg));
  SMPLR_LAUNCH_CHECK("smplr_seg_fwd");
  return 0;
}

int smplr_seg_bwd(const float *dseg, const float *seg, const int16_t *arg, const float *proj,
                  const float *mask, int B, int VP, int W, int P, float *dproj, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && VP > 0 && VP <= 32767 && W > 0 && W <= 1024 && P >= 1 && P <= 31,
                "smplr_seg_bwd: bad sizes B=%d VP=%d W=%d P=%d", B, VP, W, P);
  if (B == 0) return 0;
  SMPLR_REQUIRE(dseg && seg && arg && proj && mask && dproj, "smplr_seg_bwd: null pointer");
  const size_t lds = (size_t)VP * 2 * sizeof(float);
  SMPLR_REQUIRE(lds <= 150 * 1024, "smplr_seg_bwd: VP=%d needs %zu B of LDS", VP, lds);
  int rc = set_lds_attr(reinterpret_cast<const void *>(seg_bwd_kernel), lds);
  if (rc) return rc;
  hipLaunchKernelGGL(seg_bwd_kernel, dim3(B), dim3(1024), lds, as_stream(stream), dseg, seg,
                     reinterpret_cast<const short *>(arg), proj, mask, VP, W, P, dproj);
  SMPLR_LAUNCH_CHECK("smplr_seg_bwd");
  return 0;
}

size_t smplr_silh_workspace(int B, int VP) {
  if (B <= 0 || VP <= 0) return 0;
  const int KP = (VP + smplr::CH - 1) / smplr::CH * smplr::CH;
  return (size_t)B * KP * 4 * sizeof(float);
}

int smplr_silh_fwd(const float *proj, int B, int VP, int W, float *silh, int32_t *arg, void *workspace,
                   void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && VP > 0 && W > 0 && W <= 1024, "smplr_silh_fwd: bad sizes B=%d VP=%d W=%d", B, VP, W);
  if (B == 0) return 0;
  SMPLR_REQUIRE(proj && silh && arg && workspace, "smplr_silh_fwd: null pointer");
  const int KP = (VP + CH - 1) / CH * CH;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(seg_prep_kernel, dim3((KP + 255) / 256, B), dim3(256), 0, st, proj,
                     (const float *)nullptr, (const int *)nullptr, VP, KP, reinterpret_cast<float4 *>(workspace));
  SMPLR_LAUNCH_CHECK("smplr_silh_fwd(prep)");
  hipLaunchKernelGGL(raster_fwd_kernel<true>, dim3((W * W + RT - 1) / RT, B), dim3(RT), 0, st,
                     reinterpret_cast<const float4 *>(workspace), (const int *)nullptr, (const int *)nullptr, 1,
                     KP, W, silh, reinterpret_cast<void *>(arg));
  SMPLR_LAUNCH_CHECK("smplr_silh_fwd");
  return 0;
}

int smplr_silh_bwd(const float *dsilh, const float *silh, const int32_t *arg, const float *proj, int B,
                   int VP, int W, float *dproj, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(B >= 0 && VP > 0 && W > 0 && W <= 1024, "smplr_silh_bwd: bad sizes B=%d VP=%d W=%d", B, VP, W);
  if (B == 0) return 0;
  SMPLR_REQUIRE(dsilh && silh && arg && proj && dproj, "smplr_silh_bwd: null pointer");
  const size_t lds = (size_t)VP * 2 * sizeof(float);
  SMPLR_REQUIRE(lds <= 150 * 1024, "smplr_silh_bwd: VP=%d needs %zu B of LDS", VP, lds);
  int rc = set_lds_attr(reinterpret_cast<const void *>(silh_bwd_kernel), lds);
  if (rc) return rc;
  hipLaunchKernelGGL(silh_bwd_kernel, dim3(B), dim3(1024), lds, as_stream(stream), dsilh, silh, arg, proj, VP,
                     W, dproj);
  SMPLR_LAUNCH_CHECK("smplr_silh_bwd");
  return 0;
}

}  // extern "C"
