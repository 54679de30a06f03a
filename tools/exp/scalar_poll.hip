// Experiment: does polling the cluster flags with SCALAR loads (s_load_dwordx16 glc: through the scalar path to L2, its
// own lgkmcnt) take the flag poll out from behind in-flight VECTOR loads (loads return in order per wave)?  Each step a
// workgroup first issues cold HBM loads (like the epilogue operands of the LSTM kernels), then polls.
//   mode 0: vector poll (global_load sc1), mode 1: scalar poll (s_load_dwordx16 glc)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(unsigned* flags, float* payload, const uint4* cold, unsigned* bad, int steps, int mode,
                                         int ncold, unsigned* tmo, unsigned* sink) {
  extern __shared__ char lds[];
  const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
  const int cluster = xcd * 2 + slot / 16, me = slot % 16;
  unsigned* fl = flags + cluster * 64;
  const int tid = threadIdx.x;
  unsigned nbad = 0, acc = 0;
  for (int s = 0; s < steps; ++s) {
    float* pay = payload + ((size_t)s * gridDim.x + cluster * 16) * 256;
    pay[me * 256 + tid] = (float)(s * 1000 + me);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(fl + me, (unsigned)(s + 1), RLX_AGENT);
    // cold loads issued BEFORE the poll (never touched before: HBM latency)
    uint4 c[4];
    for (int i = 0; i < ncold; ++i) c[i] = cold[((size_t)(s * gridDim.x + blockIdx.x) * 4 + i) * 256 + tid];
    unsigned spins = 0;
    if (mode == 0) {
      const int lane = tid & 63;
      while (true) {
        unsigned v = lane < 16 ? __hip_atomic_load(fl + lane, RLX_AGENT) : 0xffffffffu;
        if (__all(v >= (unsigned)(s + 1))) break;
        if (++spins > (1u << 18)) { atomicExch(tmo, 1u); break; }
      }
    } else {
      while (true) {
        u32x16 f;
        asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(f) : "s"(fl) : "memory");
        unsigned m = f[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) m = m < f[i] ? m : f[i];
        if (m >= (unsigned)(s + 1)) break;
        if (++spins > (1u << 18)) { atomicExch(tmo, 1u); break; }
      }
    }
    asm volatile("" ::: "memory");
    for (int p = 0; p < 16; ++p) {
      float a = __hip_atomic_load(pay + p * 256 + tid, RLX_AGENT);
      nbad += (a != (float)(s * 1000 + p));
    }
    for (int i = 0; i < ncold; ++i) acc += c[i].x;
  }
  if (nbad) atomicAdd(bad, nbad);
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const int grid = 256, steps = 100;
  unsigned *flags, *tmo, *bad, *sink;
  float* pay;
  uint4* cold;
  hipMalloc(&flags, 64 * 64 * 4); hipMalloc(&tmo, 4); hipMalloc(&bad, 4); hipMalloc(&sink, 4);
  hipMalloc(&pay, (size_t)steps * grid * 256 * 4);
  const size_t coldb = (size_t)steps * grid * 4 * 256 * 16;
  hipMalloc(&cold, coldb);
  hipMemset(cold, 1, coldb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int ncold : {0, 4})
    for (int mode = 0; mode < 2; ++mode) {
      float best = 1e9f; unsigned t = 0, b = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(flags, 0, 64 * 64 * 4); hipMemset(tmo, 0, 4); hipMemset(bad, 0, 4);
        hipMemset(pay, 0xff, (size_t)steps * grid * 256 * 4);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 100 * 1024, 0, flags, pay, cold, bad, steps, mode, ncold, tmo, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        unsigned tt, bb; hipMemcpy(&tt, tmo, 4, hipMemcpyDeviceToHost); hipMemcpy(&bb, bad, 4, hipMemcpyDeviceToHost);
        t |= tt; b += bb;
      }
      printf("cold loads %d  %s poll: %.2f us per step, timeout=%u bad=%u\n", ncold, mode ? "scalar" : "vector", best * 1e3 / steps, t, b);
    }
  return 0;
}
