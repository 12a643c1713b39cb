// PReLU (per-channel slope) forward and backward for the ENet encoder that feeds the path
// (reference: encoders/encoder_enet_simple.py:21,37,50,58,79 - `PReLU(shared_axes=[1, 2])`; SURVEY.md
// 8(f) next-1).  The encoder's convolutions and batch norms stay on stock MIOpen / rocBLAS; this
// one activation is here because its stock backward is half of the reference train step on this
// GPU (torch materialises a full-size per-element weight gradient and reduces it afterwards:
// 1.16 ms per call, 67 calls per step).  Here one pass reads x and dy, writes dx and keeps the
// slope gradient in registers: a workgroup owns one (image, channel) plane chunk, reduces its
// sum and stores ONE partial; a second kernel adds the partials of a channel in a fixed order.
// NCHW, fp32.  HBM-bound: forward 8 B/element, backward 12 B/element.
#include "common.h"

namespace smplr {

constexpr int PR_T = 256;
constexpr int PR_CHUNK = 4096;       // elements of a plane per workgroup (16 per thread)

__global__ __launch_bounds__(PR_T) void prelu_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                         int C, int HW, int chunks, float *__restrict__ y) {
  const long long plane = blockIdx.x / chunks;
  const int chunk = blockIdx.x - (int)(plane * chunks);
  const float a = w[plane % C];
  const size_t base = (size_t)plane * HW;
  const int e0 = chunk * PR_CHUNK, e1 = min(HW, e0 + PR_CHUNK);
  if (((HW | e0) & 3) == 0) {                         // plane rows are 16-B aligned: float4 path
    const float4 *xv = reinterpret_cast<const float4 *>(x + base);
    float4 *yv = reinterpret_cast<float4 *>(y + base);
    for (int i = e0 / 4 + threadIdx.x; i < e1 / 4; i += PR_T) {
      float4 v = xv[i];
      v.x = v.x > 0.f ? v.x : a * v.x; v.y = v.y > 0.f ? v.y : a * v.y;
      v.z = v.z > 0.f ? v.z : a * v.z; v.w = v.w > 0.f ? v.w : a * v.w;
      yv[i] = v;
    }
  } else {
    for (int i = e0 + threadIdx.x; i < e1; i += PR_T) {
      const float v = x[base + i];
      y[base + i] = v > 0.f ? v : a * v;
    }
  }
}

__global__ __launch_bounds__(PR_T) void prelu_bwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ gy, int C, int HW, int chunks,
                                                         float *__restrict__ gx, float *__restrict__ part) {
  __shared__ float red[PR_T / 64];
  const long long plane = blockIdx.x / chunks;
  const int chunk = blockIdx.x - (int)(plane * chunks);
  const float a = w[plane % C];
  const size_t base = (size_t)plane * HW;
  const int e0 = chunk * PR_CHUNK, e1 = min(HW, e0 + PR_CHUNK);
  float s = 0.f;
  if (((HW | e0) & 3) == 0) {
    const float4 *xv = reinterpret_cast<const float4 *>(x + base), *gv = reinterpret_cast<const float4 *>(gy + base);
    float4 *ov = reinterpret_cast<float4 *>(gx + base);
    for (int i = e0 / 4 + threadIdx.x; i < e1 / 4; i += PR_T) {
      const float4 v = xv[i], g = gv[i];
      float4 o;
      o.x = v.x > 0.f ? g.x : a * g.x; s += v.x > 0.f ? 0.f : g.x * v.x;
      o.y = v.y > 0.f ? g.y : a * g.y; s += v.y > 0.f ? 0.f : g.y * v.y;
      o.z = v.z > 0.f ? g.z : a * g.z; s += v.z > 0.f ? 0.f : g.z * v.z;
      o.w = v.w > 0.f ? g.w : a * g.w; s += v.w > 0.f ? 0.f : g.w * v.w;
      ov[i] = o;
    }
  } else {
    for (int i = e0 + threadIdx.x; i < e1; i += PR_T) {
      const float v = x[base + i], g = gy[base + i];
      gx[base + i] = v > 0.f ? g : a * g;
      s += v > 0.f ? 0.f : g * v;
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// gw[c] = sum over images n and chunks of part[(n*C + c)*chunks + chunk], in index order
__global__ __launch_bounds__(64) void prelu_bwd_reduce_kernel(const float *__restrict__ part, long long N, int C,
                                                              int chunks, float *__restrict__ gw) {
  const int c = blockIdx.x, lane = threadIdx.x;
  const long long per = N * chunks;                   // partials of this channel
  float s = 0.f;
  for (long long i = lane; i < per; i += 64) {
    const long long n = i / chunks, ch = i - n * chunks;
    s += part[(n * C + c) * chunks + ch];
  }
  s = wave_sum(s);                                     // xor butterfly: fixed order
  if (lane == 0) gw[c] = s;
}

static int prelu_chunks(int HW) { return (HW + PR_CHUNK - 1) / PR_CHUNK; }

}  // namespace smplr

extern "C" int smplr_prelu_fwd(const float *x, const float *w, long long N, int C, int HW, float *y, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(N >= 0 && C > 0 && HW > 0 && N * C * (long long)prelu_chunks(HW) < (1ll << 31),
                "smplr_prelu_fwd: bad sizes N=%lld C=%d HW=%d", N, C, HW);
  if (N == 0) return 0;
  SMPLR_REQUIRE(x && w && y, "smplr_prelu_fwd: null pointer");
  const int chunks = prelu_chunks(HW);
  hipLaunchKernelGGL(prelu_fwd_kernel, dim3((unsigned)(N * C * chunks)), dim3(PR_T), 0, as_stream(stream), x, w, C, HW,
                     chunks, y);
  SMPLR_LAUNCH_CHECK("smplr_prelu_fwd");
  return 0;
}

extern "C" size_t smplr_prelu_bwd_workspace(long long N, int C, int HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  return (size_t)N * C * smplr::prelu_chunks(HW) * sizeof(float);
}

extern "C" int smplr_prelu_bwd(const float *x, const float *w, const float *gy, long long N, int C, int HW, float *gx,
                               float *gw, void *workspace, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(N >= 0 && C > 0 && HW > 0 && N * C * (long long)prelu_chunks(HW) < (1ll << 31),
                "smplr_prelu_bwd: bad sizes N=%lld C=%d HW=%d", N, C, HW);
  SMPLR_REQUIRE(gw != nullptr, "smplr_prelu_bwd: null gw");
  if (N == 0) {
    SMPLR_HIP(hipMemsetAsync(gw, 0, (size_t)C * sizeof(float), as_stream(stream)));
    return 0;
  }
  SMPLR_REQUIRE(x && w && gy && gx && workspace, "smplr_prelu_bwd: null pointer");
  const int chunks = prelu_chunks(HW);
  float *part = reinterpret_cast<float *>(workspace);
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3((unsigned)(N * C * chunks)), dim3(PR_T), 0, as_stream(stream), x, w, gy, C,
                     HW, chunks, gx, part);
  SMPLR_LAUNCH_CHECK("smplr_prelu_bwd");
  hipLaunchKernelGGL(prelu_bwd_reduce_kernel, dim3(C), dim3(64), 0, as_stream(stream), part, N, C, chunks, gw);
  SMPLR_LAUNCH_CHECK("smplr_prelu_bwd(reduce)");
  return 0;
}
