// Probe: cycles one fp32 vector instruction occupies a SIMD's issue port with 1, 2, 4 waves per SIMD (4, 8, 16 waves
// per CU) - plain v_fma_f32 and the packed forms the rasteriser's pair loop uses (v_pk_add_f32 / v_pk_fma_f32) and
// v_min3_f32.  No memory traffic; 8 independent chains per lane so that dependency latency never limits issue.
// The in-kernel clock comes from s_memtime / s_memrealtime (100 MHz), so cycles are shader cycles, not wall guesses.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/valu_issue_probe.hip -o gpurun_out/valu_issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int UNROLL = 64;    // instructions per loop trip and kind

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *stamps, int iters, float a, float b) {
  float x[8];
  f32x2 y[8];
  for (int i = 0; i < 8; ++i) { x[i] = a + i; y[i] = f32x2{a + i, b - i}; }
  const f32x2 a2 = {a, a}, b2 = {b, b};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int c = u & 7;
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
      if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[c]) : "v"(a2), "v"(b2));
      if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[c]) : "v"(a2));
      if (KIND == 3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
      if (KIND == 4) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
      if (KIND == 5) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
      if (KIND == 6) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[c]) : "v"(a), "v"(b) : "vcc");   // 2 instr
      if (KIND == 7) asm volatile("v_exp_f32 %0, %0" : "+v"(x[c]));
      if (KIND == 8) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[c]));
      if (KIND == 9) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
      if (KIND == 10) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[c]) : "v"(a2));
      if (KIND == 11) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[c]) : "v"(a));
      if (KIND == 12) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
      if (KIND == 13) asm volatile("v_mov_b32 %0, %1" : "+v"(x[c]) : "v"(a));
      if (KIND == 14) asm volatile("v_fma_f32 %0, %0, %0, %1" : "+v"(x[c]) : "v"(b));     // (du, du, t): two distinct sources
      if (KIND == 15) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(y[c]) : "v"(a2)); // the loop's shape: fma(du, du, t)
      if (KIND == 16) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
      if (KIND == 17) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
      if (KIND == 18) asm volatile("v_min_f32 %0, %0, %1\n\tv_min_f32 %0, %0, %2" : "+v"(x[c]) : "v"(a), "v"(b));   // 2 instr
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0;
    stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
  }
}

static int cmp(const void *a, const void *b) {
  const double x = *(const double *)a, y = *(const double *)b;
  return x < y ? -1 : x > y;
}

template <int KIND>
static void run(const char *name, float *out, unsigned long long *stamps, unsigned long long *h) {
  const int iters = 256;
  for (int wps = 1; wps <= 8; wps <<= 1) {
    const int blocks = 256 * wps;                       // wps blocks of 4 waves per CU -> wps waves per SIMD
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, stamps, iters, 1.0001f, 0.5f);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, stamps, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipMemcpy(h, stamps, sizeof(unsigned long long) * blocks * 8, hipMemcpyDeviceToHost);
    const int nw = blocks * 4;
    double *cyc = (double *)malloc(sizeof(double) * nw), *ghz = (double *)malloc(sizeof(double) * nw);
    for (int w = 0; w < nw; ++w) { cyc[w] = (double)h[2 * w]; ghz[w] = (double)h[2 * w] / ((double)h[2 * w + 1] * 10.0); }
    qsort(cyc, nw, sizeof(double), cmp);
    qsort(ghz, nw, sizeof(double), cmp);
    const double n = (double)iters * UNROLL;
    // a wave's lifetime holds its own n instructions and those of the (wps - 1) waves sharing its SIMD
    printf("%-14s waves/SIMD=%d  median wave: %.0f cycles for %d instr -> %.2f cycles per instruction PER SIMD (%.2f per wave), "
           "in-kernel clock %.2f GHz\n", name, wps, cyc[nw / 2], (int)n, cyc[nw / 2] / (n * wps), cyc[nw / 2] / n, ghz[nw / 2]);
    free(cyc); free(ghz);
  }
}

int main() {
  float *out; unsigned long long *stamps;
  hipMalloc(&out, 2048 * 256 * sizeof(float));
  hipMalloc(&stamps, 2048 * 8 * sizeof(unsigned long long));
  unsigned long long *h = (unsigned long long *)malloc(2048 * 8 * sizeof(unsigned long long));
  run<0>("v_fma_f32", out, stamps, h);
  run<1>("v_pk_fma_f32", out, stamps, h);
  run<2>("v_pk_add_f32", out, stamps, h);
  run<3>("v_min3_f32", out, stamps, h);
  run<4>("v_min_f32", out, stamps, h);
  run<5>("v_sub_f32", out, stamps, h);
  run<6>("cmp+cndmask(x2)", out, stamps, h);
  run<7>("v_exp_f32", out, stamps, h);
  run<8>("v_sqrt_f32", out, stamps, h);
  run<9>("v_min3_u32", out, stamps, h);
  run<10>("v_pk_mul_f32", out, stamps, h);
  run<11>("v_mul_u32_u24", out, stamps, h);
  run<12>("v_min_u32", out, stamps, h);
  run<13>("v_mov_b32", out, stamps, h);
  run<14>("v_fma(x,x,b)", out, stamps, h);
  run<15>("pk_fma(a,a,y)", out, stamps, h);
  run<16>("v_max3_f32", out, stamps, h);
  run<17>("v_med3_f32", out, stamps, h);
  run<18>("v_min_f32(x2)", out, stamps, h);
  return 0;
}
