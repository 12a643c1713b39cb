// Probe: what a dependent chain of N tiny kernels costs per kernel when replayed from a HIP graph (the floor
// under every kernel of the step), and for a kernel with one global round trip.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_empty(float *p) { if (p == nullptr) p[0] = 1.f; }
__global__ void k_touch(float *p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main() {
  float *buf; (void)hipMalloc(&buf, 64 << 20);
  (void)hipMemset(buf, 0, 64 << 20);
  hipStream_t st; (void)hipStreamCreate(&st);
  for (int variant = 0; variant < 3; ++variant) {
    const int N = 10;
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < N; ++i) {
      if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, buf);
      else if (variant == 1) hipLaunchKernelGGL(k_touch, dim3(128), dim3(256), 0, st, buf, 128 * 256);
      else hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, st, buf, 4096 * 256);   // 4 MB r/w
    }
    (void)hipStreamEndCapture(st, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 5; ++i) (void)hipGraphLaunch(ge, st);
    (void)hipStreamSynchronize(st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < 50; ++i) (void)hipGraphLaunch(ge, st);
    (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("variant %d (%s): %.2f us per kernel in a %d-kernel graph\n", variant,
           variant == 0 ? "empty, 1 wave" : variant == 1 ? "128 blocks, 1 round trip" : "4096 blocks, 4 MB r/w", ms * 1e3 / 50 / N, N);
  }
  return 0;
}
