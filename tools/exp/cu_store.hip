// Experiment: what does a burst of 16-byte-per-lane global stores cost the waves that issue it, as a function of how many CUs burst
// at once and of the burst's size?  (The forward recurrence's saved-for-backward stores: 28 KB per CU and burst, two bursts per
// step, every CU at about the same time.)  Each workgroup (256 threads = 4 waves, one per SIMD) writes KB kilobytes of its own
// region per round, rounds separated by ~2 us of s_sleep; the time from the first store's issue to (a) the last store's ISSUE and
// (b) s_waitcnt vmcnt(0) is clocked by wave 0.  grid = number of workgroups (one per CU for <= 256).
// hipcc --offload-arch=gfx950 -O3 tools/exp/cu_store.hip -o tools/exp/cu_store && tools/exp/cu_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ROUNDS = 24;

template <int KB>
__global__ __launch_bounds__(256) void kstore(uint4* data, unsigned long long* clk) {
  const int tid = threadIdx.x;
  constexpr int PER = KB * 1024 / 16 / 256;  // 16-byte stores per thread and round
  unsigned long long t_issue = 0, t_done = 0;
  for (int rd = 0; rd < ROUNDS; ++rd) {
    uint4* reg = data + ((size_t)rd * gridDim.x + blockIdx.x) * (KB * 1024 / 16);
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
#pragma unroll
    for (int k = 0; k < PER; ++k) reg[k * 256 + tid] = uint4{(unsigned)rd, (unsigned)k, (unsigned)tid, 1u};
    const unsigned long long t1 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = wall_clock64();
    if (rd >= 4) t_issue += t1 - t0, t_done += t2 - t0;
    for (int i = 0; i < 24; ++i) __builtin_amdgcn_s_sleep(16);  // ~2 us between bursts
  }
  if (tid == 0) clk[blockIdx.x * 2] = t_issue, clk[blockIdx.x * 2 + 1] = t_done;
}

template <int KB>
void run(int grid) {
  uint4* data;
  unsigned long long* clk;
  const size_t bytes = (size_t)ROUNDS * grid * KB * 1024;
  hipMalloc(&data, bytes);
  hipMalloc(&clk, grid * 16);
  hipLaunchKernelGGL(kstore<KB>, dim3(grid), dim3(256), 0, 0, data, clk);
  hipLaunchKernelGGL(kstore<KB>, dim3(grid), dim3(256), 0, 0, data, clk);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 2);
  hipMemcpy(h.data(), clk, grid * 16, hipMemcpyDeviceToHost);
  double ti = 0, td = 0;
  for (int i = 0; i < grid; ++i) ti += h[2 * i], td += h[2 * i + 1];
  ti = ti / grid / (ROUNDS - 4) * 0.01, td = td / grid / (ROUNDS - 4) * 0.01;  // us (100 MHz clock)
  printf("%3d workgroups x %2d KB per burst: issue %.2f us, until acknowledged %.2f us  (%.1f GB/s per CU, %.2f TB/s over the chip)\n", grid, KB,
         ti, td, KB * 1024 / td * 1e-3, KB * 1024.0 * grid / td * 1e-6);
  hipFree(data);
  hipFree(clk);
}

int main() {
  for (int grid : {1, 8, 32, 64, 128, 256}) run<28>(grid);
  for (int grid : {1, 256}) run<14>(grid);
  for (int grid : {1, 256}) run<56>(grid);
  return 0;
}
