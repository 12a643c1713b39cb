// Training-mode batch normalisation (+ optional per-channel PReLU) for the ENet encoder that feeds the
// path, forward and backward (reference: encoders/encoder_enet_simple.py:19-21,35-37,48-50,56-58 -
// `BatchNormalization(momentum=0.1)` followed by `PReLU(shared_axes=[1, 2])`).
// The encoder's convolutions stay on stock MIOpen / rocBLAS.  The stock batch norm parallelises over
// channels only: on ENet's 16-channel 128x128 and 64x64 maps (B = 256: 268 / 67 MB per tensor) it moves
// 0.5-0.75 TB/s and is a third of the reference train step on this GPU (27.5 of 75 ms, plus 4.8 ms of
// PReLU).  Here every pass is cut into (image, channel, 4096-element chunk) workgroups like act.hip:
//   forward : stats (sum x, sum x^2 per chunk) -> finalize (per channel, fixed order, in double; running
//             statistics updated as torch.nn.BatchNorm2d does) -> apply y = (x - mean) rstd gamma + beta,
//             z = y > 0 ? y : a y;
//   backward: with x_hat and y recomputed from x (nothing but mean / rstd is saved),
//             dy = dz (y > 0 ? 1 : a);  partial sums of dy, dy x_hat, dz y [y <= 0] -> finalize ->
//             dx = gamma rstd (dy - mean(dy) - x_hat mean(dy x_hat)).
// NCHW, fp32, HBM-bound: forward 3 passes over the tensor, backward 5 (the unfused pair: 5 and 8).
// Every reduction has a fixed order (no atomics): results are run-to-run identical.
#include "common.h"

namespace smplr {

constexpr int BN_T = 256;
constexpr int BN_CHUNK = 4096;       // elements of a plane per workgroup (16 per thread)

static int bn_chunks(int HW) { return (HW + BN_CHUNK - 1) / BN_CHUNK; }

__device__ __forceinline__ void block_store3(float s0, float s1, float s2, float *red, float *dst, int n) {
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w * 3] = s0; red[w * 3 + 1] = s1; red[w * 3 + 2] = s2; }
  __syncthreads();
  if (threadIdx.x < n) {
    const int k = threadIdx.x;
    dst[k] = ((red[k] + red[3 + k]) + red[6 + k]) + red[9 + k];
  }
}

// part[(plane * chunks + chunk) * 2 + {0, 1}] = sum x, sum x^2 of the chunk
__global__ __launch_bounds__(BN_T) void bn_stats_kernel(const float *__restrict__ x, int HW, int chunks,
                                                        float *__restrict__ part) {
  __shared__ float red[12];
  const long long plane = blockIdx.x / chunks;
  const int chunk = blockIdx.x - (int)(plane * chunks);
  const size_t base = (size_t)plane * HW;
  const int e0 = chunk * BN_CHUNK, e1 = min(HW, e0 + BN_CHUNK);
  float s = 0.f, q = 0.f;
  if (((HW | e0) & 3) == 0) {
    const float4 *xv = reinterpret_cast<const float4 *>(x + base);
    for (int i = e0 / 4 + threadIdx.x; i < e1 / 4; i += BN_T) {
      const float4 v = xv[i];
      s += (v.x + v.y) + (v.z + v.w);
      q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  } else {
    for (int i = e0 + threadIdx.x; i < e1; i += BN_T) {
      const float v = x[base + i];
      s += v;
      q += v * v;
    }
  }
  block_store3(s, q, 0.f, red, part + (size_t)blockIdx.x * 2, 2);
}

// per channel: the chunk sums of all images in index order (thread-strided, then a fixed tree), in double
__global__ __launch_bounds__(BN_T) void bn_finalize_kernel(const float *__restrict__ part, long long N, int C,
                                                           int chunks, long long M, float eps, float momentum,
                                                           float *__restrict__ mean, float *__restrict__ rstd,
                                                           float *__restrict__ run_mean,
                                                           float *__restrict__ run_var) {
  __shared__ double rs[BN_T], rq[BN_T];
  const int c = blockIdx.x;
  const long long per = N * chunks;
  double s = 0.0, q = 0.0;
  for (long long i = threadIdx.x; i < per; i += BN_T) {
    const long long n = i / chunks, ch = i - n * chunks;
    const float *p = part + ((n * C + c) * chunks + ch) * 2;
    s += (double)p[0];
    q += (double)p[1];
  }
  rs[threadIdx.x] = s;
  rq[threadIdx.x] = q;
  __syncthreads();
  for (int o = BN_T / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      rs[threadIdx.x] += rs[threadIdx.x + o];
      rq[threadIdx.x] += rq[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double m = rs[0] / (double)M;
    double var = rq[0] / (double)M - m * m;             // biased (population) variance normalises
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * (float)m;
    if (run_var) {                                       // torch keeps the unbiased estimate
      const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
      run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (float)unb;
    }
  }
}

template <bool PRELU>
__global__ __launch_bounds__(BN_T) void bn_apply_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta,
                                                        const float *__restrict__ slope,
                                                        const float *__restrict__ mean,
                                                        const float *__restrict__ rstd, int C, int HW, int chunks,
                                                        float *__restrict__ z) {
  const long long plane = blockIdx.x / chunks;
  const int chunk = blockIdx.x - (int)(plane * chunks);
  const int c = (int)(plane % C);
  const float sc = rstd[c] * gamma[c], sh = beta[c] - mean[c] * sc;   // y = x sc + sh
  const float a = PRELU ? slope[c] : 1.0f;
  const size_t base = (size_t)plane * HW;
  const int e0 = chunk * BN_CHUNK, e1 = min(HW, e0 + BN_CHUNK);
  if (((HW | e0) & 3) == 0) {
    const float4 *xv = reinterpret_cast<const float4 *>(x + base);
    float4 *zv = reinterpret_cast<float4 *>(z + base);
    for (int i = e0 / 4 + threadIdx.x; i < e1 / 4; i += BN_T) {
      const float4 v = xv[i];
      float4 y;
      y.x = fmaf(v.x, sc, sh); y.y = fmaf(v.y, sc, sh); y.z = fmaf(v.z, sc, sh); y.w = fmaf(v.w, sc, sh);
      if (PRELU) {
        y.x = y.x > 0.f ? y.x : a * y.x; y.y = y.y > 0.f ? y.y : a * y.y;
        y.z = y.z > 0.f ? y.z : a * y.z; y.w = y.w > 0.f ? y.w : a * y.w;
      }
      zv[i] = y;
    }
  } else {
    for (int i = e0 + threadIdx.x; i < e1; i += BN_T) {
      float y = fmaf(x[base + i], sc, sh);
      if (PRELU) y = y > 0.f ? y : a * y;
      z[base + i] = y;
    }
  }
}

// dy and x_hat of one element, and its contribution to the slope gradient
template <bool PRELU>
__device__ __forceinline__ void bn_elem(float xv, float dz, float mu, float rs, float g, float b, float a, float &xh,
                                        float &dy, float &da) {
  xh = (xv - mu) * rs;
  if (PRELU) {
    const float sc = rs * g;
    const float y = fmaf(xv, sc, b - mu * sc);       // exactly the forward's y (same sign test)
    dy = y > 0.f ? dz : a * dz;
    da = y > 0.f ? 0.f : dz * y;
  } else {
    dy = dz;
    da = 0.f;
  }
}

// part[(plane * chunks + chunk) * 3 + {0, 1, 2}] = sum dy, sum dy x_hat, sum dz y [y <= 0]
template <bool PRELU>
__global__ __launch_bounds__(BN_T) void bn_bwd_stats_kernel(const float *__restrict__ x, const float *__restrict__ dz,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta,
                                                            const float *__restrict__ slope,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ rstd, int C, int HW, int chunks,
                                                            float *__restrict__ part) {
  __shared__ float red[12];
  const long long plane = blockIdx.x / chunks;
  const int chunk = blockIdx.x - (int)(plane * chunks);
  const int c = (int)(plane % C);
  const float mu = mean[c], rs = rstd[c], g = gamma[c], b = beta[c], a = PRELU ? slope[c] : 1.0f;
  const size_t base = (size_t)plane * HW;
  const int e0 = chunk * BN_CHUNK, e1 = min(HW, e0 + BN_CHUNK);
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#define SMPLR_BN_ACC(XV, DZ)                          \
  {                                                   \
    float xh_, dy_, da_;                              \
    bn_elem<PRELU>(XV, DZ, mu, rs, g, b, a, xh_, dy_, da_); \
    s1 += dy_;                                        \
    s2 = fmaf(dy_, xh_, s2);                          \
    s3 += da_;                                        \
  }
  if (((HW | e0) & 3) == 0) {
    const float4 *xv = reinterpret_cast<const float4 *>(x + base), *gv = reinterpret_cast<const float4 *>(dz + base);
    for (int i = e0 / 4 + threadIdx.x; i < e1 / 4; i += BN_T) {
      const float4 v = xv[i], d = gv[i];
      SMPLR_BN_ACC(v.x, d.x) SMPLR_BN_ACC(v.y, d.y) SMPLR_BN_ACC(v.z, d.z) SMPLR_BN_ACC(v.w, d.w)
    }
  } else {
    for (int i = e0 + threadIdx.x; i < e1; i += BN_T) SMPLR_BN_ACC(x[base + i], dz[base + i])
  }
#undef SMPLR_BN_ACC
  block_store3(s1, s2, s3, red, part + (size_t)blockIdx.x * 3, 3);
}

__global__ __launch_bounds__(BN_T) void bn_bwd_finalize_kernel(const float *__restrict__ part, long long N, int C,
                                                               int chunks, long long M,
                                                               float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                               float *__restrict__ dslope,
                                                               float *__restrict__ k12) {
  __shared__ double r1[BN_T], r2[BN_T], r3[BN_T];
  const int c = blockIdx.x;
  const long long per = N * chunks;
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (long long i = threadIdx.x; i < per; i += BN_T) {
    const long long n = i / chunks, ch = i - n * chunks;
    const float *p = part + ((n * C + c) * chunks + ch) * 3;
    s1 += (double)p[0];
    s2 += (double)p[1];
    s3 += (double)p[2];
  }
  r1[threadIdx.x] = s1; r2[threadIdx.x] = s2; r3[threadIdx.x] = s3;
  __syncthreads();
  for (int o = BN_T / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      r1[threadIdx.x] += r1[threadIdx.x + o];
      r2[threadIdx.x] += r2[threadIdx.x + o];
      r3[threadIdx.x] += r3[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    dbeta[c] = (float)r1[0];
    dgamma[c] = (float)r2[0];
    if (dslope) dslope[c] = (float)r3[0];
    k12[2 * c] = (float)(r1[0] / (double)M);
    k12[2 * c + 1] = (float)(r2[0] / (double)M);
  }
}

template <bool PRELU>
__global__ __launch_bounds__(BN_T) void bn_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ dz,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta,
                                                            const float *__restrict__ slope,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ rstd,
                                                            const float *__restrict__ k12, int C, int HW, int chunks,
                                                            float *__restrict__ dx) {
  const long long plane = blockIdx.x / chunks;
  const int chunk = blockIdx.x - (int)(plane * chunks);
  const int c = (int)(plane % C);
  const float mu = mean[c], rs = rstd[c], g = gamma[c], b = beta[c], a = PRELU ? slope[c] : 1.0f;
  const float k1 = k12[2 * c], k2 = k12[2 * c + 1], gr = g * rs;
  const size_t base = (size_t)plane * HW;
  const int e0 = chunk * BN_CHUNK, e1 = min(HW, e0 + BN_CHUNK);
#define SMPLR_BN_DX(XV, DZ, OUT)                      \
  {                                                   \
    float xh_, dy_, da_;                              \
    bn_elem<PRELU>(XV, DZ, mu, rs, g, b, a, xh_, dy_, da_); \
    OUT = gr * ((dy_ - k1) - xh_ * k2);               \
  }
  if (((HW | e0) & 3) == 0) {
    const float4 *xv = reinterpret_cast<const float4 *>(x + base), *gv = reinterpret_cast<const float4 *>(dz + base);
    float4 *ov = reinterpret_cast<float4 *>(dx + base);
    for (int i = e0 / 4 + threadIdx.x; i < e1 / 4; i += BN_T) {
      const float4 v = xv[i], d = gv[i];
      float4 o;
      SMPLR_BN_DX(v.x, d.x, o.x) SMPLR_BN_DX(v.y, d.y, o.y) SMPLR_BN_DX(v.z, d.z, o.z) SMPLR_BN_DX(v.w, d.w, o.w)
      ov[i] = o;
    }
  } else {
    for (int i = e0 + threadIdx.x; i < e1; i += BN_T) SMPLR_BN_DX(x[base + i], dz[base + i], dx[base + i])
  }
#undef SMPLR_BN_DX
}

// ---- residual form: out = prelu(scale[plane] * (gamma x_hat + beta) + other, slope) --------------------
// The tail of an ENet bottleneck (encoder_enet_simple.py:56-79): BatchNormalization -> SpatialDropout2D
// (scale[plane] = 0 or 1/(1-p) per (image, channel)) -> add the other branch -> PReLU, as ONE pass over
// the tensor instead of four (3 tensor-passes forward instead of 9, 8 backward instead of 10).
__global__ __launch_bounds__(BN_T) void bn_res_apply_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta,
                                                            const float *__restrict__ scale,
                                                            const float *__restrict__ other,
                                                            const float *__restrict__ slope,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ rstd, int C, int HW, int chunks,
                                                            float *__restrict__ out) {
  const long long plane = blockIdx.x / chunks;
  const int chunk = blockIdx.x - (int)(plane * chunks);
  const int c = (int)(plane % C);
  const float ps = scale ? scale[plane] : 1.0f;
  const float sc = rstd[c] * gamma[c], sh = beta[c] - mean[c] * sc;   // y = x sc + sh
  const float a = slope[c];
  const size_t base = (size_t)plane * HW;
  const int e0 = chunk * BN_CHUNK, e1 = min(HW, e0 + BN_CHUNK);
#define SMPLR_RES(XV, OV, OUT)                                \
  {                                                           \
    const float pre_ = fmaf(ps, fmaf(XV, sc, sh), OV);        \
    OUT = pre_ > 0.f ? pre_ : a * pre_;                       \
  }
  if (((HW | e0) & 3) == 0) {
    const float4 *xv = reinterpret_cast<const float4 *>(x + base), *ov = reinterpret_cast<const float4 *>(other + base);
    float4 *zv = reinterpret_cast<float4 *>(out + base);
    for (int i = e0 / 4 + threadIdx.x; i < e1 / 4; i += BN_T) {
      const float4 v = xv[i], o = ov[i];
      float4 z;
      SMPLR_RES(v.x, o.x, z.x) SMPLR_RES(v.y, o.y, z.y) SMPLR_RES(v.z, o.z, z.z) SMPLR_RES(v.w, o.w, z.w)
      zv[i] = z;
    }
  } else {
    for (int i = e0 + threadIdx.x; i < e1; i += BN_T) SMPLR_RES(x[base + i], other[base + i], out[base + i])
  }
#undef SMPLR_RES
}

// d_pre, x_hat and the slope-gradient term of one element of the residual form
__device__ __forceinline__ void bn_res_elem(float xv, float ov, float dout, float mu, float rs, float g, float b, float ps,
                                            float a, float &xh, float &dpre, float &da) {
  xh = (xv - mu) * rs;
  const float sc = rs * g;
  const float pre = fmaf(ps, fmaf(xv, sc, b - mu * sc), ov);   // exactly the forward's value (same sign test)
  dpre = pre > 0.f ? dout : a * dout;
  da = pre > 0.f ? 0.f : dout * pre;
}

__global__ __launch_bounds__(BN_T) void bn_res_bwd_stats_kernel(const float *__restrict__ x, const float *__restrict__ dout,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ beta,
                                                                const float *__restrict__ scale,
                                                                const float *__restrict__ other,
                                                                const float *__restrict__ slope,
                                                                const float *__restrict__ mean,
                                                                const float *__restrict__ rstd, int C, int HW,
                                                                int chunks, float *__restrict__ part) {
  __shared__ float red[12];
  const long long plane = blockIdx.x / chunks;
  const int chunk = blockIdx.x - (int)(plane * chunks);
  const int c = (int)(plane % C);
  const float ps = scale ? scale[plane] : 1.0f;
  const float mu = mean[c], rs = rstd[c], g = gamma[c], b = beta[c], a = slope[c];
  const size_t base = (size_t)plane * HW;
  const int e0 = chunk * BN_CHUNK, e1 = min(HW, e0 + BN_CHUNK);
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#define SMPLR_RES_ACC(XV, OV, DZ)                                     \
  {                                                                   \
    float xh_, dp_, da_;                                              \
    bn_res_elem(XV, OV, DZ, mu, rs, g, b, ps, a, xh_, dp_, da_);      \
    const float dy_ = ps * dp_;                                       \
    s1 += dy_;                                                        \
    s2 = fmaf(dy_, xh_, s2);                                          \
    s3 += da_;                                                        \
  }
  if (((HW | e0) & 3) == 0) {
    const float4 *xv = reinterpret_cast<const float4 *>(x + base), *ov = reinterpret_cast<const float4 *>(other + base),
                 *gv = reinterpret_cast<const float4 *>(dout + base);
    for (int i = e0 / 4 + threadIdx.x; i < e1 / 4; i += BN_T) {
      const float4 v = xv[i], o = ov[i], d = gv[i];
      SMPLR_RES_ACC(v.x, o.x, d.x) SMPLR_RES_ACC(v.y, o.y, d.y) SMPLR_RES_ACC(v.z, o.z, d.z) SMPLR_RES_ACC(v.w, o.w, d.w)
    }
  } else {
    for (int i = e0 + threadIdx.x; i < e1; i += BN_T) SMPLR_RES_ACC(x[base + i], other[base + i], dout[base + i])
  }
#undef SMPLR_RES_ACC
  block_store3(s1, s2, s3, red, part + (size_t)blockIdx.x * 3, 3);
}

__global__ __launch_bounds__(BN_T) void bn_res_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ dout,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ beta,
                                                                const float *__restrict__ scale,
                                                                const float *__restrict__ other,
                                                                const float *__restrict__ slope,
                                                                const float *__restrict__ mean,
                                                                const float *__restrict__ rstd,
                                                                const float *__restrict__ k12, int C, int HW, int chunks,
                                                                float *__restrict__ dx, float *__restrict__ dother) {
  const long long plane = blockIdx.x / chunks;
  const int chunk = blockIdx.x - (int)(plane * chunks);
  const int c = (int)(plane % C);
  const float ps = scale ? scale[plane] : 1.0f;
  const float mu = mean[c], rs = rstd[c], g = gamma[c], b = beta[c], a = slope[c];
  const float k1 = k12[2 * c], k2 = k12[2 * c + 1], gr = g * rs;
  const size_t base = (size_t)plane * HW;
  const int e0 = chunk * BN_CHUNK, e1 = min(HW, e0 + BN_CHUNK);
#define SMPLR_RES_DX(XV, OV, DZ, DX, DO)                              \
  {                                                                   \
    float xh_, dp_, da_;                                              \
    bn_res_elem(XV, OV, DZ, mu, rs, g, b, ps, a, xh_, dp_, da_);      \
    DO = dp_;                                                         \
    DX = gr * ((ps * dp_ - k1) - xh_ * k2);                           \
  }
  if (((HW | e0) & 3) == 0) {
    const float4 *xv = reinterpret_cast<const float4 *>(x + base), *ov = reinterpret_cast<const float4 *>(other + base),
                 *gv = reinterpret_cast<const float4 *>(dout + base);
    float4 *dxv = reinterpret_cast<float4 *>(dx + base), *dov = reinterpret_cast<float4 *>(dother + base);
    for (int i = e0 / 4 + threadIdx.x; i < e1 / 4; i += BN_T) {
      const float4 v = xv[i], o = ov[i], d = gv[i];
      float4 r, q;
      SMPLR_RES_DX(v.x, o.x, d.x, r.x, q.x) SMPLR_RES_DX(v.y, o.y, d.y, r.y, q.y)
      SMPLR_RES_DX(v.z, o.z, d.z, r.z, q.z) SMPLR_RES_DX(v.w, o.w, d.w, r.w, q.w)
      dxv[i] = r;
      dov[i] = q;
    }
  } else {
    for (int i = e0 + threadIdx.x; i < e1; i += BN_T)
      SMPLR_RES_DX(x[base + i], other[base + i], dout[base + i], dx[base + i], dother[base + i])
  }
#undef SMPLR_RES_DX
}

static size_t bn_ws_floats(long long N, int C, int HW) { return (size_t)N * C * bn_chunks(HW) * 3 + (size_t)C * 2; }

}  // namespace smplr

extern "C" {

size_t smplr_bn_workspace(long long N, int C, int HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  return smplr::bn_ws_floats(N, C, HW) * sizeof(float);
}

int smplr_bn_fwd(const float *x, const float *gamma, const float *beta, const float *slope, long long N, int C,
                 int HW, float eps, float momentum, float *running_mean, float *running_var, float *z,
                 float *save_mean, float *save_rstd, void *workspace, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(N >= 0 && C > 0 && HW > 0 && N * C * (long long)bn_chunks(HW) < (1ll << 31) && eps > 0.0f,
                "smplr_bn_fwd: bad sizes N=%lld C=%d HW=%d eps=%g", N, C, HW, (double)eps);
  if (N == 0) return 0;
  SMPLR_REQUIRE(x && gamma && beta && z && save_mean && save_rstd && workspace, "smplr_bn_fwd: null pointer");
  const int chunks = bn_chunks(HW);
  const unsigned grid = (unsigned)(N * C * chunks);
  float *part = reinterpret_cast<float *>(workspace);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(grid), dim3(BN_T), 0, st, x, HW, chunks, part);
  SMPLR_LAUNCH_CHECK("smplr_bn_fwd(stats)");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(BN_T), 0, st, part, N, C, chunks, N * (long long)HW, eps, momentum,
                     save_mean, save_rstd, running_mean, running_var);
  SMPLR_LAUNCH_CHECK("smplr_bn_fwd(finalize)");
  if (slope)
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(grid), dim3(BN_T), 0, st, x, gamma, beta, slope, save_mean, save_rstd,
                       C, HW, chunks, z);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(grid), dim3(BN_T), 0, st, x, gamma, beta, slope, save_mean,
                       save_rstd, C, HW, chunks, z);
  SMPLR_LAUNCH_CHECK("smplr_bn_fwd(apply)");
  return 0;
}

int smplr_bn_bwd(const float *x, const float *gamma, const float *beta, const float *slope, const float *save_mean,
                 const float *save_rstd, const float *dz, long long N, int C, int HW, float *dx, float *dgamma,
                 float *dbeta, float *dslope, void *workspace, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(N >= 0 && C > 0 && HW > 0 && N * C * (long long)bn_chunks(HW) < (1ll << 31),
                "smplr_bn_bwd: bad sizes N=%lld C=%d HW=%d", N, C, HW);
  SMPLR_REQUIRE(dgamma && dbeta && (!slope || dslope), "smplr_bn_bwd: null gradient output");
  hipStream_t st = as_stream(stream);
  if (N == 0) {
    SMPLR_HIP(hipMemsetAsync(dgamma, 0, (size_t)C * sizeof(float), st));
    SMPLR_HIP(hipMemsetAsync(dbeta, 0, (size_t)C * sizeof(float), st));
    if (dslope) SMPLR_HIP(hipMemsetAsync(dslope, 0, (size_t)C * sizeof(float), st));
    return 0;
  }
  SMPLR_REQUIRE(x && gamma && beta && save_mean && save_rstd && dz && dx && workspace, "smplr_bn_bwd: null pointer");
  const int chunks = bn_chunks(HW);
  const unsigned grid = (unsigned)(N * C * chunks);
  float *part = reinterpret_cast<float *>(workspace);
  float *k12 = part + (size_t)N * C * chunks * 3;
  if (slope)
    hipLaunchKernelGGL(bn_bwd_stats_kernel<true>, dim3(grid), dim3(BN_T), 0, st, x, dz, gamma, beta, slope, save_mean,
                       save_rstd, C, HW, chunks, part);
  else
    hipLaunchKernelGGL(bn_bwd_stats_kernel<false>, dim3(grid), dim3(BN_T), 0, st, x, dz, gamma, beta, slope, save_mean,
                       save_rstd, C, HW, chunks, part);
  SMPLR_LAUNCH_CHECK("smplr_bn_bwd(stats)");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(BN_T), 0, st, part, N, C, chunks, N * (long long)HW, dgamma,
                     dbeta, slope ? dslope : nullptr, k12);
  SMPLR_LAUNCH_CHECK("smplr_bn_bwd(finalize)");
  if (slope)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(grid), dim3(BN_T), 0, st, x, dz, gamma, beta, slope, save_mean,
                       save_rstd, k12, C, HW, chunks, dx);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(grid), dim3(BN_T), 0, st, x, dz, gamma, beta, slope, save_mean,
                       save_rstd, k12, C, HW, chunks, dx);
  SMPLR_LAUNCH_CHECK("smplr_bn_bwd(apply)");
  return 0;
}

int smplr_bn_res_fwd(const float *x, const float *gamma, const float *beta, const float *plane_scale,
                     const float *other, const float *slope, long long N, int C, int HW, float eps, float momentum,
                     float *running_mean, float *running_var, float *out, float *save_mean, float *save_rstd,
                     void *workspace, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(N >= 0 && C > 0 && HW > 0 && N * C * (long long)bn_chunks(HW) < (1ll << 31) && eps > 0.0f,
                "smplr_bn_res_fwd: bad sizes N=%lld C=%d HW=%d eps=%g", N, C, HW, (double)eps);
  if (N == 0) return 0;
  SMPLR_REQUIRE(x && gamma && beta && other && slope && out && save_mean && save_rstd && workspace,
                "smplr_bn_res_fwd: null pointer");
  const int chunks = bn_chunks(HW);
  const unsigned grid = (unsigned)(N * C * chunks);
  float *part = reinterpret_cast<float *>(workspace);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(grid), dim3(BN_T), 0, st, x, HW, chunks, part);
  SMPLR_LAUNCH_CHECK("smplr_bn_res_fwd(stats)");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(BN_T), 0, st, part, N, C, chunks, N * (long long)HW, eps, momentum,
                     save_mean, save_rstd, running_mean, running_var);
  SMPLR_LAUNCH_CHECK("smplr_bn_res_fwd(finalize)");
  hipLaunchKernelGGL(bn_res_apply_kernel, dim3(grid), dim3(BN_T), 0, st, x, gamma, beta, plane_scale, other, slope,
                     save_mean, save_rstd, C, HW, chunks, out);
  SMPLR_LAUNCH_CHECK("smplr_bn_res_fwd(apply)");
  return 0;
}

int smplr_bn_res_bwd(const float *x, const float *gamma, const float *beta, const float *plane_scale,
                     const float *other, const float *slope, const float *save_mean, const float *save_rstd,
                     const float *dout, long long N, int C, int HW, float *dx, float *dother, float *dgamma,
                     float *dbeta, float *dslope, void *workspace, void *stream) {
  using namespace smplr;
  SMPLR_REQUIRE(N >= 0 && C > 0 && HW > 0 && N * C * (long long)bn_chunks(HW) < (1ll << 31),
                "smplr_bn_res_bwd: bad sizes N=%lld C=%d HW=%d", N, C, HW);
  SMPLR_REQUIRE(dgamma && dbeta && dslope, "smplr_bn_res_bwd: null gradient output");
  hipStream_t st = as_stream(stream);
  if (N == 0) {
    SMPLR_HIP(hipMemsetAsync(dgamma, 0, (size_t)C * sizeof(float), st));
    SMPLR_HIP(hipMemsetAsync(dbeta, 0, (size_t)C * sizeof(float), st));
    SMPLR_HIP(hipMemsetAsync(dslope, 0, (size_t)C * sizeof(float), st));
    return 0;
  }
  SMPLR_REQUIRE(x && gamma && beta && other && slope && save_mean && save_rstd && dout && dx && dother && workspace,
                "smplr_bn_res_bwd: null pointer");
  const int chunks = bn_chunks(HW);
  const unsigned grid = (unsigned)(N * C * chunks);
  float *part = reinterpret_cast<float *>(workspace);
  float *k12 = part + (size_t)N * C * chunks * 3;
  hipLaunchKernelGGL(bn_res_bwd_stats_kernel, dim3(grid), dim3(BN_T), 0, st, x, dout, gamma, beta, plane_scale, other,
                     slope, save_mean, save_rstd, C, HW, chunks, part);
  SMPLR_LAUNCH_CHECK("smplr_bn_res_bwd(stats)");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(BN_T), 0, st, part, N, C, chunks, N * (long long)HW, dgamma,
                     dbeta, dslope, k12);
  SMPLR_LAUNCH_CHECK("smplr_bn_res_bwd(finalize)");
  hipLaunchKernelGGL(bn_res_bwd_apply_kernel, dim3(grid), dim3(BN_T), 0, st, x, dout, gamma, beta, plane_scale, other,
                     slope, save_mean, save_rstd, k12, C, HW, chunks, dx, dother);
  SMPLR_LAUNCH_CHECK("smplr_bn_res_bwd(apply)");
  return 0;
}

}  // extern "C"
