// Probe: producer -> consumer hand-overs, every word checked.
//
// Part 1 (LDS, between waves of ONE workgroup): round 2 dropped a step counter in LDS as the hand-over between
// the coefficient producers and the multiplying waves of pose_blend3_fwd_kernel because it was wrong about once in
// 30 launches, cause unknown ("A counter in LDS is NOT a safe hand-over").  This pins the cause by
// running the candidate protocols 10^5 times each, under LDS and VALU noise from the other waves of the CU:
//   P1  ONE producer wave: ds_write payload -> s_waitcnt lgkmcnt(0) -> volatile store of the step; consumer:
//       volatile poll -> plain loads.  (the "obvious" form)
//   P2  the same with release / acquire atomics at workgroup scope
//   P3  TWO producer waves each write half of the payload (wave 1 late by a varying amount, as under uneven load);
//       wave 0 alone publishes after ITS OWN wait - the flaw a step counter written by one of several producers has
//   P4  two producer waves, each adds 1 to the counter (release) after its own wait; the consumer waits for 2 per step
// A protocol passes with 0 mismatching words.  The payload is 4 KB (16 B per lane per wave-store, 4 stores), new
// values every step; the consumer acknowledges a step through a second word so that the producers may overwrite.
//
// Part 2 (global memory, between TWO workgroups on different CUs): the exchange a two-workgroups-per-mesh binning
// kernel would need (each workgroup hands the other 32 KB - its half's z-buffer - and waits for the other's), in
// the form MI355X_MICROARCH.md prescribes (plain stores -> every wave's vmcnt(0) -> barrier -> lane-0 agent release
// -> vmcnt(0) -> relaxed agent flag; consumer: relaxed poll -> agent acquire -> vmcnt(0) -> barrier -> plain loads),
// 256 workgroups of 1 024 threads, pairs (2k, 2k + 1), every spin bounded.  Reports mismatching words and the
// shader-clock cost of one exchange (median / p90 over pairs and steps).
//
// Build + run: hipcc -O3 --offload-arch=gfx950 tools/probes/handover_probe.hip -o gpurun_out/handover_probe && gpurun_out/handover_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
constexpr int PAY = 1024;          // payload dwords (4 KB)
constexpr unsigned SPIN_MAX = 1u << 22;

__device__ __forceinline__ unsigned word(unsigned step, unsigned i, unsigned wg) { return step * 2654435761u + i * 40503u + wg; }

// waves: 0, 1 = producers, 2 = consumer, 3.. = noise (LDS reads/writes + VALU on a scratch region)
template <int PROTO>
__global__ __launch_bounds__(512) void lds_handover(int steps, unsigned *bad_out, unsigned *timeout_out) {
  __shared__ unsigned pay[PAY];
  __shared__ unsigned flag, ack;
  __shared__ unsigned noise[2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { flag = 0; ack = 0; }
  for (int i = tid; i < 2048; i += 512) noise[i] = i;
  __syncthreads();
  unsigned bad = 0, timed_out = 0;
  const unsigned wg = blockIdx.x;
  if (wave <= 1) {
    const bool two = PROTO >= 3;
    if (wave == 1 && !two) return;
    for (int s = 1; s <= steps; ++s) {
      // wait until the consumer has read step s - 1
      unsigned spin = 0;
      while (__hip_atomic_load(&ack, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)(s - 1) && ++spin < SPIN_MAX) {}
      if (spin >= SPIN_MAX) { timed_out = 1; break; }
      if (two && wave == 1) {                                        // uneven load: producer 1 is late by a varying amount
        unsigned z = (unsigned)s * 747796405u + wg;
        for (unsigned k = 0; k < (z >> 26); ++k) __builtin_amdgcn_s_sleep(4);
      }
      const int half = two ? PAY / 2 : PAY, base = two ? wave * (PAY / 2) : 0;
      for (int i = lane * 4; i < half; i += 256) {
        uint4 v;
        v.x = word(s, base + i, wg); v.y = word(s, base + i + 1, wg); v.z = word(s, base + i + 2, wg); v.w = word(s, base + i + 3, wg);
        *reinterpret_cast<uint4 *>(&pay[base + i]) = v;
      }
      if (PROTO == 1 || PROTO == 3) {
        __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): this wave's LDS stores are done
        if (wave == 0 && lane == 0) *reinterpret_cast<volatile unsigned *>(&flag) = (unsigned)s;
      } else if (PROTO == 2) {
        if (lane == 0) __hip_atomic_store(&flag, (unsigned)s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {                                                     // PROTO 4: every producer wave counts itself in
        if (lane == 0) __hip_atomic_fetch_add(&flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  } else if (wave == 2) {
    for (int s = 1; s <= steps; ++s) {
      const unsigned want = PROTO == 4 ? 2u * s : (unsigned)s;
      unsigned spin = 0;
      if (PROTO == 1 || PROTO == 3) {
        while (*reinterpret_cast<volatile unsigned *>(&flag) < want && ++spin < SPIN_MAX) {}
      } else {
        while (__hip_atomic_load(&flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want && ++spin < SPIN_MAX) {}
      }
      if (spin >= SPIN_MAX) { timed_out = 1; break; }
      for (int i = lane * 4; i < PAY; i += 256) {
        const uint4 v = *reinterpret_cast<const uint4 *>(&pay[i]);
        bad += (v.x != word(s, i, wg)) + (v.y != word(s, i + 1, wg)) + (v.z != word(s, i + 2, wg)) + (v.w != word(s, i + 3, wg));
      }
      // all lanes' reads are done before the producers may overwrite
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) __hip_atomic_store(&ack, (unsigned)s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  } else {
    // noise: keep the LDS pipe and the SIMDs busy while the others talk (bounded by the consumer's progress)
    unsigned x = tid, spin = 0;
    while (__hip_atomic_load(&ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)steps && ++spin < SPIN_MAX) {
      x = x * 1664525u + noise[(x >> 7) & 2047];
      noise[(tid * 4 + (x & 3)) & 2047] = x;
    }
    if (x == 0xdeadbeefu) bad_out[0] = 1;                            // keep x alive
  }
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if (lane == 0 && bad) atomicAdd(&bad_out[blockIdx.x], bad);
  if (lane == 0 && timed_out) atomicAdd(timeout_out, 1u);
}

// ---- part 2: two workgroups exchange 32 KB through global memory, both directions, `steps` times
constexpr int XW = 8192;           // dwords per direction (32 KB)
template <bool SC1>   // SC1: write-through stores + L1-bypassing loads (the guide's R1 form) instead of release / acquire fences
__global__ __launch_bounds__(1024) void wg_exchange(unsigned *buf /*[nwg][2][XW]*/, unsigned *flags /*[nwg]*/, int steps, int stride,
                                                   unsigned *bad_out, unsigned *timeout_out,
                                                   unsigned long long *cost /*[nwg][steps]*/) {
  const int tid = threadIdx.x, wg = blockIdx.x, peer = wg ^ stride;   // 1: neighbours (different XCDs), 8: same XCD
  __shared__ unsigned s_to;
  if (tid == 0) s_to = 0;
  __syncthreads();
  unsigned bad = 0;
  for (int s = 1; s <= steps; ++s) {
    unsigned *mine = buf + ((size_t)wg * 2 + (s & 1)) * XW;            // double-buffered by step parity: no ack needed
    const unsigned *theirs = buf + ((size_t)peer * 2 + (s & 1)) * XW;
    const unsigned long long t0 = clock64();
    for (int i = tid * 4; i < XW; i += 4096) {
      uint4 v;
      v.x = word(s, i, wg); v.y = word(s, i + 1, wg); v.z = word(s, i + 2, wg); v.w = word(s, i + 3, wg);
      if (SC1) {
        const u32x4v vv = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(mine + i), "v"(vv) : "memory");
      }
      else *reinterpret_cast<uint4 *>(mine + i) = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (!SC1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __hip_atomic_store(&flags[wg], (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spin = 0;
      while (__hip_atomic_load(&flags[peer], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)s && ++spin < SPIN_MAX)
        __builtin_amdgcn_s_sleep(1);
      if (spin >= SPIN_MAX) s_to = 1;
      if (!SC1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    if (s_to) break;                                                // (block-uniform: set before the barrier)
    for (int i = tid * 4; i < XW; i += 4096) {
      uint4 v;
      if (SC1) {
        u32x4v vv;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(vv) : "v"(theirs + i) : "memory");
        v.x = vv.x; v.y = vv.y; v.z = vv.z; v.w = vv.w;
      } else {
        v = *reinterpret_cast<const uint4 *>(theirs + i);
      }
      bad += (v.x != word(s, i, peer)) + (v.y != word(s, i + 1, peer)) + (v.z != word(s, i + 2, peer)) + (v.w != word(s, i + 3, peer));
    }
    if (tid == 0) cost[(size_t)wg * steps + (s - 1)] = clock64() - t0;
  }
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if ((tid & 63) == 0 && bad) atomicAdd(&bad_out[wg], bad);
  if (tid == 0 && s_to) atomicAdd(timeout_out, 1u);
}

template <int PROTO>
static void run_lds(const char *what, int steps) {
  const int nwg = 512;
  unsigned *bad, *to;
  CHECK(hipMalloc(&bad, nwg * 4)); CHECK(hipMalloc(&to, 4));
  CHECK(hipMemset(bad, 0, nwg * 4)); CHECK(hipMemset(to, 0, 4));
  hipLaunchKernelGGL(lds_handover<PROTO>, dim3(nwg), dim3(512), 0, 0, steps, bad, to);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned> h(nwg);
  unsigned hto = 0;
  CHECK(hipMemcpy(h.data(), bad, nwg * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&hto, to, 4, hipMemcpyDeviceToHost));
  unsigned long long words = 0; int wgs = 0;
  for (unsigned v : h) { words += v; wgs += v != 0; }
  printf("LDS  P%d %-78s %d workgroups x %d hand-overs x %d words: %llu mismatching words in %d workgroups, %u timeouts\n",
         PROTO, what, nwg, steps, PAY, words, wgs, hto);
  CHECK(hipFree(bad)); CHECK(hipFree(to));
}

int main() {
  const int steps = 100000 / 512 + 200;          // x 512 workgroups >= 10^5 hand-overs per protocol, each under noise
  run_lds<1>("one producer wave: waitcnt lgkmcnt(0) + volatile step word, plain loads", steps);
  run_lds<2>("one producer wave: release store / acquire load (workgroup scope)", steps);
  run_lds<3>("TWO producer waves, wave 0 alone publishes after its own wait (round 2's form)", steps);
  run_lds<4>("two producer waves, each counts itself in (release add); consumer waits for 2", steps);

  const int nwg = 256, xsteps = 400;
  unsigned long long total_bad = 0;
  for (int variant = 0; variant < 4; ++variant) {
    const int stride = (variant & 1) ? 8 : 1;
    const bool sc1 = variant >= 2;
    unsigned *buf, *flags, *bad, *to; unsigned long long *cost;
    CHECK(hipMalloc(&buf, (size_t)nwg * 2 * XW * 4)); CHECK(hipMalloc(&flags, nwg * 4)); CHECK(hipMalloc(&bad, nwg * 4));
    CHECK(hipMalloc(&to, 4)); CHECK(hipMalloc(&cost, (size_t)nwg * xsteps * 8));
    CHECK(hipMemset(buf, 0, (size_t)nwg * 2 * XW * 4)); CHECK(hipMemset(flags, 0, nwg * 4)); CHECK(hipMemset(bad, 0, nwg * 4));
    CHECK(hipMemset(to, 0, 4));
    if (sc1) hipLaunchKernelGGL(wg_exchange<true>, dim3(nwg), dim3(1024), 0, 0, buf, flags, xsteps, stride, bad, to, cost);
    else hipLaunchKernelGGL(wg_exchange<false>, dim3(nwg), dim3(1024), 0, 0, buf, flags, xsteps, stride, bad, to, cost);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> hb(nwg); unsigned hto = 0; std::vector<unsigned long long> hc((size_t)nwg * xsteps);
    CHECK(hipMemcpy(hb.data(), bad, nwg * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&hto, to, 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hc.data(), cost, hc.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long words = 0; for (unsigned v : hb) words += v;
    std::sort(hc.begin(), hc.end());
    printf("GLOBAL two workgroups (block b and b ^ %d: %s) exchange 32 KB each way (%s, "
           "one flag per workgroup): %d pairs x %d steps: %llu mismatching words, %u timeouts; cost per exchange "
           "(write + flag + wait + read) median %llu, p90 %llu, max %llu shader clocks\n",
           stride, stride == 1 ? "different XCDs" : "same XCD",
           sc1 ? "sc1 write-through stores, drained; sc1 loads" : "plain stores, agent release / acquire fences", nwg / 2, xsteps, words, hto, hc[hc.size() / 2],
           hc[hc.size() * 9 / 10], hc.back());
    total_bad += words + hto;
    CHECK(hipFree(buf)); CHECK(hipFree(flags)); CHECK(hipFree(bad)); CHECK(hipFree(to)); CHECK(hipFree(cost));
  }
  return total_bad ? 1 : 0;
}
