// Experiment: cost of a cluster barrier (NU workgroups per cluster) with agent-scope release/acquire,
// all workgroups co-resident (grid <= 256, 1 per CU).  Bounded spins: never hangs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__global__ __launch_bounds__(256) void kbar(unsigned* counters, float* payload, float* sink, int nu, int steps, int mode,
                                           unsigned* tmo) {
  extern __shared__ char lds[];  // force 1 WG/CU with big LDS
  const int cluster = blockIdx.x / nu, me = blockIdx.x % nu;
  unsigned* cnt = counters + cluster * 32;  // 128 B apart
  float* pay = payload + (size_t)cluster * nu * 256;
  float acc = 0.f;
  for (int s = 0; s < steps; ++s) {
    // produce 1 KB per WG
    pay[me * 256 + threadIdx.x] = (float)(s + me);
    if (mode == 0) {  // plain stores + release fence + counter; consumer acquire fence
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(cnt, 1u, RLX_AGENT);
        const unsigned target = (unsigned)nu * (s + 1);
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, RLX_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 20)) { atomicExch(tmo, 1u); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
    }
    // consume: read every peer's 1 KB (plain loads)
    for (int p = 0; p < nu; ++p) acc += pay[p * 256 + threadIdx.x];
    __syncthreads();  // WAR on payload next step is protected by the next barrier round? add a 2nd barrier below
    if (mode == 0) {  // second barrier (consumption done) so producers may overwrite: use second counter
      unsigned* cnt2 = cnt + 16;
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt2, 1u, RLX_AGENT);
        const unsigned target = (unsigned)nu * (s + 1);
        unsigned spins = 0;
        while (__hip_atomic_load(cnt2, RLX_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 20)) { atomicExch(tmo, 2u); break; }
        }
      }
      __syncthreads();
    }
  }
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  unsigned *cnt, *tmo; float *pay, *sink;
  hipMalloc(&cnt, 4096 * 4); hipMalloc(&tmo, 4); hipMalloc(&pay, 256 * 256 * 4); hipMalloc(&sink, 256 * 256 * 4);
  for (int nu : {16, 32, 256}) {
    int grid = 256, steps = 200;
    hipMemset(cnt, 0, 4096 * 4); hipMemset(tmo, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)kbar, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL(kbar, dim3(grid), dim3(256), 100 * 1024, 0, cnt, pay, sink, nu, 5, 0, tmo);
    hipDeviceSynchronize();
    hipMemset(cnt, 0, 4096 * 4);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kbar, dim3(grid), dim3(256), 100 * 1024, 0, cnt, pay, sink, nu, steps, 0, tmo);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned t; hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost);
    std::vector<float> h(256); hipMemcpy(h.data(), sink, 1024, hipMemcpyDeviceToHost);
    // expected acc for block 0, thread 0: sum_s sum_p (s+p)
    double exp = 0; for (int s = 0; s < steps; ++s) for (int p = 0; p < nu; ++p) exp += s + p;
    printf("cluster of %3d WGs: %.2f us per step (2 barriers + 1KB/WG exchange), timeout=%u, check %s (%.0f vs %.0f)\n", nu,
           ms * 1e3 / steps, t, h[0] == (float)exp ? "OK" : "MISMATCH", h[0], exp);
  }
  return 0;
}
