// Experiment: cost of a one-barrier-per-step exchange among NU co-resident workgroups when the cluster lives on ONE XCD
// (blocks with the same blockIdx % 8: workgroups are dealt round-robin over the XCDs), flag polling instead of an
// atomic counter, fresh addresses every step (no WAR barrier), L1-bypassing (sc1) consumer loads.
//   mode 0: agent-scope release/acquire fences (portable), consecutive-block clusters (spread over all XCDs)
//   mode 1: agent-scope fences, XCD-local clusters
//   mode 2: no fences: vmcnt(0) + flag store, sc1 loads, XCD-local clusters (valid only if the cluster shares an L2)
// Bounded spins: never hangs.  Every consumer checks every peer's payload each step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__device__ __forceinline__ float load_sc1(const float* p) {
  float v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

constexpr int PAY = 512;  // floats per workgroup per step (2 KB)

__global__ __launch_bounds__(256) void kbar(unsigned* flags, float* payload, unsigned* bad, unsigned* xcc, int nu, int steps,
                                           int mode, unsigned* tmo, int nread) {
  extern __shared__ char lds[];  // force 1 WG/CU with big LDS
  int cluster, me;
  if (mode == 0) {
    cluster = blockIdx.x / nu, me = blockIdx.x % nu;
  } else {
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    cluster = xcd * (32 / nu) + slot / nu, me = slot % nu;
  }
  if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id();
  unsigned* fl = flags + cluster * 64;  // one 256-B block of flags per cluster
  const int tid = threadIdx.x;
  unsigned nbad = 0;
  for (int s = 0; s < steps; ++s) {
    float* pay = payload + ((size_t)s * gridDim.x + cluster * nu) * PAY;  // fresh region every step
    pay[me * PAY + tid] = (float)(s * 1000 + me);
    pay[me * PAY + 256 + tid] = (float)(s * 1000 + me) + 0.5f;
    if (mode <= 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < 64) {
      if (tid == 0) __hip_atomic_store(fl + me, (unsigned)(s + 1), RLX_AGENT);
      unsigned spins = 0;
      while (true) {
        unsigned v = tid < nu ? __hip_atomic_load(fl + tid, RLX_AGENT) : 0xffffffffu;
        if (__all(v >= (unsigned)(s + 1))) break;
        if (++spins > (1u << 18)) {
          atomicExch(tmo, 1u);
          break;
        }
      }
    }
    __syncthreads();
    if (mode <= 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (int p = 0; p < nread; ++p) {
      float a, b;
      if (mode == 2) {
        a = __hip_atomic_load(pay + p * PAY + tid, RLX_AGENT);
        b = __hip_atomic_load(pay + p * PAY + 256 + tid, RLX_AGENT);
      } else {
        a = pay[p * PAY + tid];
        b = pay[p * PAY + 256 + tid];
      }
      nbad += (a != (float)(s * 1000 + p)) + (b != (float)(s * 1000 + p) + 0.5f);
    }
  }
  if (nbad) atomicAdd(bad, nbad);
}

int main() {
  unsigned *flags, *tmo, *bad, *xcc;
  float* pay;
  const int grid = 256, steps = 200;
  hipMalloc(&flags, 64 * 64 * 4);
  hipMalloc(&tmo, 4);
  hipMalloc(&bad, 4);
  hipMalloc(&xcc, grid * 4);
  hipMalloc(&pay, (size_t)steps * grid * PAY * 4);
  hipFuncSetAttribute((const void*)kbar, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode)
    for (int nu : {8, 16, 32}) for (int nr = 0; nr < 2; ++nr) {
      float best = 1e9f;
      unsigned t = 0, b = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(flags, 0, 64 * 64 * 4);
        hipMemset(tmo, 0, 4);
        hipMemset(bad, 0, 4);
        hipMemset(pay, 0xff, (size_t)steps * grid * PAY * 4);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kbar, dim3(grid), dim3(256), 100 * 1024, 0, flags, pay, bad, xcc, nu, steps, mode, tmo, nr ? nu : 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        unsigned tt, bb;
        hipMemcpy(&tt, tmo, 4, hipMemcpyDeviceToHost);
        hipMemcpy(&bb, bad, 4, hipMemcpyDeviceToHost);
        t |= tt;
        b += bb;
      }
      printf("mode %d  cluster of %2d WGs: %.2f us per step (1 barrier + %d KB read per WG), timeout=%u, bad=%u\n", mode, nu,
             best * 1e3 / steps, (nr ? nu : 1) * PAY * 4 / 1024, t, b);
    }
  std::vector<unsigned> x(grid);
  hipMemcpy(x.data(), xcc, grid * 4, hipMemcpyDeviceToHost);
  int mism = 0;
  for (int i = 0; i < grid; ++i) mism += (x[i] != (unsigned)(i % 8));
  printf("XCC_ID == blockIdx %% 8 for %d of %d blocks; first 16:", grid - mism, grid);
  for (int i = 0; i < 16; ++i) printf(" %u", x[i]);
  printf("\n");
  return 0;
}
