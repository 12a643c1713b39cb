// Probe: phases of blend_bwd timed alone.  MODE 0 = full kernel, 1 = no LDS reduction / partial store (acc summed to one
// value per lane), 2 = MFMA only (no operand loads), 3 = loads only.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
constexpr int BW_NT = 7, BW_NO = 224, BW_G = 8;
template <int MODE, int BW_DEPTH>
__global__ __launch_bounds__(256) void k(const float *__restrict__ dvp, const float *__restrict__ blendT, int B, int N3,
                                         int cols_per_block, int nslices, int nmt, float *__restrict__ part) {
  __shared__ float sR[32 * BW_NO];
  const int bid = blockIdx.x;
  const int group = bid / (8 * nmt), within = bid % (8 * nmt);
  const int slice = group * 8 + (within & 7), mt = within >> 3;
  if (slice >= nslices) return;
  const int m0 = mt * 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;
  const int cw = cols_per_block / 4;
  const int c_beg = slice * cols_per_block + wave * cw;
  const int c_end = min(c_beg + cw, N3);
  const int ncols = c_beg < c_end ? c_end - c_beg : 0;
  const int ngroups = ncols / BW_G;
  f32x16 acc[BW_NT];
#pragma unroll
  for (int t = 0; t < BW_NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  const int mr = min(m0 + i, B - 1);
  const float *arow = dvp + (size_t)mr * N3;
  const float *bcol = blendT + 7 * i;
  f32x4u a[BW_DEPTH + 1];
  f32x4u bx[BW_DEPTH + 1][4];
  f32x3u by[BW_DEPTH + 1][4];
#define LOAD_GROUP(slot, g)                                                              \
  {                                                                                      \
    const int c0_ = c_beg + (g) * BW_G + 4 * h;                                          \
    if (MODE == 2) {                                                                     \
      a[slot] = (f32x4u){1.f, 2.f, 3.f, (float)c0_};                                     \
      _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) { bx[slot][t_] = a[slot]; by[slot][t_] = (f32x3u){1.f, 2.f, 3.f}; } \
    } else {                                                                             \
      a[slot] = *reinterpret_cast<const f32x4u *>(arow + c0_);                           \
      _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) {                                 \
        const float *br_ = bcol + (size_t)(c0_ + t_) * BW_NO;                            \
        bx[slot][t_] = *reinterpret_cast<const f32x4u *>(br_);                           \
        by[slot][t_] = *reinterpret_cast<const f32x3u *>(br_ + 4);                       \
      }                                                                                  \
    }                                                                                    \
  }
#define MMA_GROUP(slot)                                                                  \
  _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) {                                     \
    if (MODE == 3) {                                                                     \
      _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) acc[u_][t_] += a[slot][t_] * bx[slot][t_][u_]; \
      _Pragma("unroll") for (int u_ = 0; u_ < 3; ++u_) acc[4 + u_][t_] += a[slot][t_] * by[slot][t_][u_]; \
    } else {                                                                             \
      _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_)                                   \
        acc[u_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot][t_], bx[slot][t_][u_], acc[u_], 0, 0, 0); \
      _Pragma("unroll") for (int u_ = 0; u_ < 3; ++u_)                                   \
        acc[4 + u_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot][t_], by[slot][t_][u_], acc[4 + u_], 0, 0, 0); \
    }                                                                                    \
  }
#pragma unroll
  for (int g = 0; g < BW_DEPTH; ++g)
    if (g < ngroups) LOAD_GROUP(g, g)
  __builtin_amdgcn_sched_barrier(0);
  for (int g0 = 0; g0 < ngroups; g0 += BW_DEPTH + 1) {
#pragma unroll
    for (int u = 0; u < BW_DEPTH + 1; ++u) {
      const int g = g0 + u;
      if (g < ngroups) {
        if (g + BW_DEPTH < ngroups) LOAD_GROUP((u + BW_DEPTH) % (BW_DEPTH + 1), g + BW_DEPTH)
        __builtin_amdgcn_sched_barrier(0);
        MMA_GROUP(u)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (MODE == 1 || MODE == 3) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < BW_NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[t][r];
    part[(size_t)blockIdx.x * 256 + tid] = s;
    return;
  }
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < BW_NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          float *p = &sR[row * BW_NO + 7 * i + t];
          *p = (w == 0) ? acc[t][r] : (*p + acc[t][r]);
        }
    }
  }
  __syncthreads();
  float *dst = part + ((size_t)slice * nmt + mt) * (32 * BW_NO);
  for (int e = tid; e < 32 * BW_NO; e += 256) dst[e] = sR[e];
}
template <typename F>
static float timeit(F f, float *flush, size_t flush_n) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  (void)hipDeviceSynchronize();
  float tot = 0.f;
  for (int i = 0; i < 10; ++i) {
    if (flush) (void)hipMemsetAsync(flush, i, flush_n, 0);
    (void)hipEventRecord(e0);
    f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    tot += ms;
  }
  return tot / 10 * 1e3f;
}
int main() {
  const int B = 128, N3 = 20670, nmt = 4;
  float *dvp, *blendT, *part, *flush;
  const size_t flush_n = 512u << 20;
  (void)hipMalloc(&dvp, (size_t)B * N3 * 4); (void)hipMalloc(&blendT, (size_t)N3 * BW_NO * 4);
  (void)hipMalloc(&part, (size_t)64 << 20); (void)hipMalloc(&flush, flush_n);
  (void)hipMemset(dvp, 0, (size_t)B * N3 * 4); (void)hipMemset(blendT, 0, (size_t)N3 * BW_NO * 4);
#define RUN(MODE, DEPTH, CPB)                                                                                   \
  {                                                                                                             \
    const int cpb = CPB, ns = (N3 + cpb - 1) / cpb, grid = ((ns + 7) / 8) * 8 * nmt;                            \
    auto f = [&] { hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(grid), dim3(256), 0, 0, dvp, blendT, B, N3, cpb, ns, nmt, part); }; \
    printf("MODE=%d DEPTH=%d cpb=%d (blocks %d): hot %.1f us  cold %.1f us\n", MODE, DEPTH, cpb, ns * nmt,      \
           timeit(f, nullptr, 0), timeit(f, flush, flush_n));                                                   \
  }
  RUN(0, 5, 352) RUN(1, 5, 352) RUN(2, 5, 352) RUN(3, 5, 352)
  RUN(0, 3, 352) RUN(0, 5, 320) RUN(0, 5, 384) RUN(0, 5, 704) RUN(0, 5, 192)
  return 0;
}
