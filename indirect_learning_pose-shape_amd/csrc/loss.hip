// K8 loss head: Reshape(W^2, C) + softmax + per-pixel categorical focal loss, forward and backward.
//
// Reference: model.py:119-120 (Reshape + Activation('softmax')), focal_loss.py:10-46
// (clip to [eps, 1-eps], -y log p, optional class weights, (1-p)^gamma, sum over classes), and the
// silhouette head's Keras 'categorical_crossentropy' (train_stage2_silhouette.py:85-86,226-229),
// which is the same expression with gamma = 0 and no weights (its rescale of y_pred by its row sum
// is the identity on a softmax output).
//
// The reference materialises softmax, clip, log, pow and their products as (N, W^2, C) tensors
// (>= 6 HBM round trips of the 295 KB/mesh score tensor forward, as many backward).  Here the
// score tensor is read once forward (loss out: 4 B/pixel) and once backward (gradient out), the
// softmax being recomputed rather than stored.  Both kernels are HBM-bound: lanes map onto the
// NHWC tensor exactly as it lies in memory, VEC consecutive channels per lane (16-B accesses),
// a pixel's C channels on C/VEC adjacent lanes, reductions by xor-butterfly (fixed order).
#include "common.h"

namespace smplr {

template <int GL>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = 1; o < GL; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
template <int GL>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = 1; o < GL; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int VEC>
struct VecT;
template <>
struct VecT<4> {
  using type = float4;
};
template <>
struct VecT<2> {
  using type = float2;
};

template <int VEC>
__device__ __forceinline__ void load_vec(const float *p, float (&o)[VEC]) {
  const typename VecT<VEC>::type t = *reinterpret_cast<const typename VecT<VEC>::type *>(p);
  const float *f = reinterpret_cast<const float *>(&t);
#pragma unroll
  for (int i = 0; i < VEC; ++i) o[i] = f[i];
}
template <int VEC>
__device__ __forceinline__ void store_vec(float *p, const float (&o)[VEC]) {
  typename VecT<VEC>::type t;
  float *f = reinterpret_cast<float *>(&t);
#pragma unroll
  for (int i = 0; i < VEC; ++i) f[i] = o[i];
  *reinterpret_cast<typename VecT<VEC>::type *>(p) = t;
}

// One lane = VEC channels of one pixel; GL = C / VEC lanes per pixel.
// MODE 0: integer class map (labels), MODE 1: dense y_true (N, C) (one-hot or soft).
template <int C, int VEC, int MODE, bool BWD>
__global__ __launch_bounds__(256) void focal_kernel(const float *__restrict__ logits,
                                                    const int *__restrict__ labels,
                                                    const float *__restrict__ y_true,
                                                    const float *__restrict__ class_w, float gamma,
                                                    const float *__restrict__ dloss, long long npix,
                                                    float *__restrict__ loss, float *__restrict__ probs,
                                                    float *__restrict__ dlogits) {
  constexpr int GL = C / VEC;
  const long long nlane = npix * GL;
  const long long stride = (long long)gridDim.x * 256;
  float w[VEC];
  {
    const int c0 = (threadIdx.x % GL) * VEC;     // 256 % GL == 0 and stride % GL == 0: fixed per lane
#pragma unroll
    for (int i = 0; i < VEC; ++i) w[i] = class_w ? class_w[c0 + i] : 1.0f;
  }
  // every lane of a pixel group takes the same trip count (nlane and stride are multiples of GL)
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nlane; e += stride) {
    const long long pix = e / GL;
    const int c0 = (int)(e - pix * GL) * VEC;
    float s[VEC], y[VEC];
    load_vec<VEC>(logits + e * VEC, s);
    if (MODE == 1) {
      load_vec<VEC>(y_true + e * VEC, y);
    } else {
      const int t = labels[pix];
#pragma unroll
      for (int i = 0; i < VEC; ++i) y[i] = (c0 + i == t) ? 1.0f : 0.0f;
    }
    float mx = s[0];
#pragma unroll
    for (int i = 1; i < VEC; ++i) mx = fmaxf(mx, s[i]);
    mx = group_max<GL>(mx);
    float ex[VEC], den = 0.0f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      ex[i] = expf(s[i] - mx);
      den += ex[i];
    }
    den = group_sum<GL>(den);
    const float inv = 1.0f / den;
    float sm[VEC], p[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      sm[i] = ex[i] * inv;
      p[i] = fminf(fmaxf(sm[i], K_EPS), 1.0f - K_EPS);       // focal_loss.py:17
    }
    if (!BWD) {
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < VEC; ++i)                           // :18, :41, :43
        acc += pow_gamma(1.0f - p[i], gamma) * ((-y[i] * logf(p[i])) * w[i]);
      acc = group_sum<GL>(acc);                               // :44
      if (c0 == 0) loss[pix] = acc;
      if (probs) store_vec<VEC>(probs + e * VEC, sm);
    } else {
      // q_c = dL/d(softmax_c); the clip passes gradient on [eps, 1-eps] only
      float q[VEC], dot = 0.0f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const bool inside = sm[i] >= K_EPS && sm[i] <= 1.0f - K_EPS;
        const float om = 1.0f - p[i];
        const float d = (y[i] * w[i]) * (dpow_gamma(om, gamma) * logf(p[i]) - pow_gamma(om, gamma) / p[i]);
        q[i] = inside ? d : 0.0f;
        dot += q[i] * sm[i];
      }
      dot = group_sum<GL>(dot);
      const float g = dloss[pix];
      float o[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = g * (sm[i] * (q[i] - dot));
      store_vec<VEC>(dlogits + e * VEC, o);
    }
  }
}

template <int C, int VEC, bool BWD>
static int launch_focal(const float *logits, const int *labels, const float *y_true, const float *class_w,
                        float gamma, const float *dloss, long long npix, float *loss, float *probs,
                        float *dlogits, hipStream_t st) {
  constexpr int GL = C / VEC;
  const long long nlane = npix * GL;
  long long blocks = (nlane + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;     // 16 blocks per CU, grid-stride beyond
  if (labels) {
    hipLaunchKernelGGL((focal_kernel<C, VEC, 0, BWD>), dim3((unsigned)blocks), dim3(256), 0, st, logits, labels,
                       y_true, class_w, gamma, dloss, npix, loss, probs, dlogits);
  } else {
    hipLaunchKernelGGL((focal_kernel<C, VEC, 1, BWD>), dim3((unsigned)blocks), dim3(256), 0, st, logits, labels,
                       y_true, class_w, gamma, dloss, npix, loss, probs, dlogits);
  }
  return 0;
}

static int check_focal(const char *fn, const float *logits, const int *labels, const float *y_true, long long npix,
                       int C, float gamma) {
  SMPLR_REQUIRE(npix >= 0 && npix < (1ll << 40), "%s: bad pixel count %lld", fn, npix);
  SMPLR_REQUIRE(C == 32 || C == 2, "%s: C=%d; the reference's heads have 32 (parts) or 2 (silhouette) classes",
                fn, C);
  SMPLR_REQUIRE(gamma >= 0.0f, "%s: gamma=%g must be >= 0", fn, (double)gamma);
  if (npix == 0) return 0;
  SMPLR_REQUIRE(logits != nullptr, "%s: null logits", fn);
  SMPLR_REQUIRE((labels != nullptr) != (y_true != nullptr), "%s: pass exactly one of labels / y_true", fn);
  return 0;
}

}  // namespace smplr

extern "C" int smplr_focal_fwd(const float *logits, const int *labels, const float *y_true, const float *class_w,
                               float gamma, long long npix, int C, float *loss, float *probs, void *stream) {
  using namespace smplr;
  if (int rc = check_focal("smplr_focal_fwd", logits, labels, y_true, npix, C, gamma)) return rc;
  if (npix == 0) return 0;
  SMPLR_REQUIRE(loss != nullptr, "smplr_focal_fwd: null loss");
  if (C == 32)
    launch_focal<32, 4, false>(logits, labels, y_true, class_w, gamma, nullptr, npix, loss, probs, nullptr,
                               as_stream(stream));
  else
    launch_focal<2, 2, false>(logits, labels, y_true, class_w, gamma, nullptr, npix, loss, probs, nullptr,
                              as_stream(stream));
  SMPLR_LAUNCH_CHECK("smplr_focal_fwd");
  return 0;
}

extern "C" int smplr_focal_bwd(const float *logits, const int *labels, const float *y_true, const float *class_w,
                               float gamma, const float *dloss, long long npix, int C, float *dlogits,
                               void *stream) {
  using namespace smplr;
  if (int rc = check_focal("smplr_focal_bwd", logits, labels, y_true, npix, C, gamma)) return rc;
  if (npix == 0) return 0;
  SMPLR_REQUIRE(dloss != nullptr && dlogits != nullptr, "smplr_focal_bwd: null pointer");
  if (C == 32)
    launch_focal<32, 4, true>(logits, labels, y_true, class_w, gamma, dloss, npix, nullptr, nullptr, dlogits,
                              as_stream(stream));
  else
    launch_focal<2, 2, true>(logits, labels, y_true, class_w, gamma, dloss, npix, nullptr, nullptr, dlogits,
                             as_stream(stream));
  SMPLR_LAUNCH_CHECK("smplr_focal_bwd");
  return 0;
}
