// Experiment: how fast can every CU read a 64-KB block that workgroups of its own XCD have just written (the cluster
// exchange of lstm_cluster.hip), as a function of the load flavour and of how many workgroups share the block.
//   kind 0: global_load_dwordx4 (plain)     kind 1: buffer_load_dwordx4 sc1     kind 2: global_load_lds_dwordx4 sc1
//   kind 3: buffer_load sc1, but each wave instruction covers 16 rows x 64 B of a row-major [32 rows][2 KB] block (the MFMA
//           fragment pattern of a row-major operand) instead of 1 KB contiguous
//   share 1: every workgroup reads its own block; share 16: the 16 members of a cluster read the same block
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int KB = 64;               // bytes read per workgroup per round = KB * 1024
constexpr int ROUNDS = 20;

template <int KIND>
__global__ __launch_bounds__(256) void kread(uint4* data, unsigned* flags, unsigned long long* clk, unsigned* sink, int share) {
  extern __shared__ char lds[];
  const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;          // 32 slots per XCD
  const int cluster = xcd * 2 + slot / 16, me = slot % 16;
  const int tid = threadIdx.x;
  unsigned* fl = flags + cluster * 64;
  unsigned acc = 0;
  unsigned long long t_read = 0;
  for (int rd = 0; rd < ROUNDS; ++rd) {
    // region of this round: cluster region = 16 members x 4 KB (share 16) or this workgroup's own 64 KB (share 1)
    uint4* reg = data + ((size_t)rd * 256 + (share == 16 ? cluster * 16 : blockIdx.x)) * (KB * 1024 / 16 / (share == 16 ? 16 : 1));
    // write my part: share 16 -> my 4 KB of the cluster's 64 KB; share 1 -> all 64 KB of my own block
    const int nw = (share == 16 ? 4 : 64) * 1024 / 16;
    uint4* mine = share == 16 ? reg + me * nw : reg;
    for (int i = tid; i < nw; i += 256) mine[i] = uint4{(unsigned)rd, (unsigned)i, 0u, 1u};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(fl + me, (unsigned)(rd + 1), RLX_AGENT);
    if (tid < 64) {
      unsigned spins = 0;
      while (true) {
        unsigned v = tid < 16 ? __hip_atomic_load(fl + tid, RLX_AGENT) : 0xffffffffu;
        if (__all(v >= (unsigned)(rd + 1))) break;
        if (++spins > (1u << 20)) break;
      }
    }
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    const uint4* src = share == 16 ? data + ((size_t)rd * 256 + cluster * 16) * (KB * 1024 / 16 / 16) : reg;
    // 64 KB = 4096 x 16 B: 16 loads per thread, all in flight
    uint4 v[16];
    if constexpr (KIND == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = src[k * 256 + tid];
    } else if constexpr (KIND == 1) {
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, (k * 256 + tid) * 16, 0, 16);
        v[k] = uint4{w.x, w.y, w.z, w.w};
      }
    } else if constexpr (KIND == 3) {
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
      // the 64 KB as 32 rows x 2 KB; wave w takes k-steps [8w, 8w+8) (64 B each) of ... rows: lane (r = lane&15, q = lane>>4)
      const int lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
#pragma unroll
      for (int k = 0; k < 16; ++k) {  // k < 8: rows 0..15, k >= 8: rows 16..31; k-step (w*8 + k%8)
        const int row = (k >> 3) * 16 + r, ks = w * 8 + (k & 7);
        u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, row * 2048 + ks * 64 + q * 16, 0, 16);
        v[k] = uint4{x.x, x.y, x.z, x.w};
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + k * 256 + tid),
                                         (void __attribute__((address_space(3)))*)(lds + k * 4096 + (tid >> 6) * 1024), 16, 0, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = *(const uint4*)(lds + k * 4096 + tid * 16);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += v[k].x + v[k].w;
    asm volatile("" : "+v"(acc));
    t_read += wall_clock64() - t0;
    __syncthreads();
  }
  if (tid == 0) clk[blockIdx.x] = t_read;
  if (acc == 0xdeadbeef) sink[0] = acc;
}

int main() {
  uint4* data;
  unsigned *flags, *sink;
  unsigned long long* clk;
  const size_t bytes = (size_t)ROUNDS * 256 * KB * 1024;
  hipMalloc(&data, bytes);
  hipMalloc(&flags, 64 * 64 * 4);
  hipMalloc(&sink, 4);
  hipMalloc(&clk, 256 * 8);
  auto run = [&](int kind, int share) {
    hipMemset(flags, 0, 64 * 64 * 4);
    hipMemset(data, 0xff, bytes);
    hipDeviceSynchronize();
    if (kind == 0) hipLaunchKernelGGL(kread<0>, dim3(256), dim3(256), 65536, 0, data, flags, clk, sink, share);
    if (kind == 1) hipLaunchKernelGGL(kread<1>, dim3(256), dim3(256), 65536, 0, data, flags, clk, sink, share);
    if (kind == 3) hipLaunchKernelGGL(kread<3>, dim3(256), dim3(256), 65536, 0, data, flags, clk, sink, share);
    if (kind == 2) hipLaunchKernelGGL(kread<2>, dim3(256), dim3(256), 65536, 0, data, flags, clk, sink, share);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), clk, 256 * 8, hipMemcpyDeviceToHost);
    double tot = 0, mx = 0;
    for (auto c : h) { tot += c; if (c > mx) mx = c; }
    const double us = tot / 256 / ROUNDS * 0.01, usmax = mx / ROUNDS * 0.01;
    printf("kind %d share %2d: %.2f us per 64 KB read (slowest workgroup %.2f) = %.1f GB/s per CU\n", kind, share, us, usmax, 65536 / us / 1e3);
  };
  hipFuncSetAttribute((const void*)kread<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int share : {1, 16})
    for (int kind = 0; kind < 4; ++kind) run(kind, share);
  return 0;
}
