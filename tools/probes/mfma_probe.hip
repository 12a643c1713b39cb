// Probe: issue rate of v_mfma_f32_32x32x2_f32 / 16x16x4_f32 per SIMD, 1..4 waves per SIMD, no memory traffic.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_probe.hip -o gpurun_out/mfma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(256) void k32(float *out, int iters, float a, float b) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k16(float *out, int iters, float a, float b) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <typename F>
static float timeit(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5 * 1e3f;
}
int main() {
  float *out; hipMalloc(&out, 256 * 256 * 16 * sizeof(float));
  const int iters = 64;   // 512 MFMAs per chain
  for (int bpc = 1; bpc <= 4; ++bpc) {
    const int blocks = 256 * bpc;   // bpc blocks of 4 waves per CU -> bpc waves per SIMD
    float t1 = timeit([&] { hipLaunchKernelGGL(k32<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    float t2 = timeit([&] { hipLaunchKernelGGL(k32<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    float t3 = timeit([&] { hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    const double n1 = 512.0 * bpc, n2 = 1024.0 * bpc;
    printf("waves/SIMD=%d  32x32x2 1 chain: %.1f us (%.0f cyc/MFMA/SIMD @2.4GHz)  2 chains: %.1f us (%.0f)  16x16x4: %.1f us (%.0f)\n",
           bpc, t1, t1 * 2400 / n1, t2, t2 * 2400 / n2, t3, t3 * 2400 / n1);
  }
  return 0;
}
