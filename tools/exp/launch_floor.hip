// Experiment: per-kernel cost of a dependent chain of launches replayed from a hipGraph, vs kernel "weight"
// (workgroups, static LDS, kernarg bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { float* p; long pad[100]; };
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
template <int LDS> __global__ __launch_bounds__(256) void heavy(Big b) {
  __shared__ char s[LDS];
  if (threadIdx.x == 0) s[blockIdx.x % LDS] = 1;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) b.p[0] += (float)s[0];
}
template <class F> float run(F launch, int n) {
  hipStream_t st; hipStreamCreate(&st);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < n; ++i) launch(st);
  hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st); for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / (5 * n);
}
int main() {
  float* p; hipMalloc(&p, 1024); hipMemset(p, 0, 1024);
  Big b = {}; b.p = p;
  printf("tiny  <<<1,64>>>            : %.2f us per dependent launch\n", run([&](hipStream_t s) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, p); }, 400));
  printf("tiny  <<<128,256>>>         : %.2f us\n", run([&](hipStream_t s) { hipLaunchKernelGGL(tiny, dim3(128), dim3(256), 0, s, p); }, 400));
  printf("heavy <<<128,256>>> 1KB LDS, 808B kernarg : %.2f us\n", run([&](hipStream_t s) { hipLaunchKernelGGL(heavy<1024>, dim3(128), dim3(256), 0, s, b); }, 400));
  printf("heavy <<<128,256>>> 64KB LDS, 808B kernarg: %.2f us\n", run([&](hipStream_t s) { hipLaunchKernelGGL(heavy<65536>, dim3(128), dim3(256), 0, s, b); }, 400));
  printf("heavy <<<1024,256>>> 40KB LDS             : %.2f us\n", run([&](hipStream_t s) { hipLaunchKernelGGL(heavy<40960>, dim3(1024), dim3(256), 0, s, b); }, 400));
  return 0;
}
