// trace.h -- optional per-launch HIP-event timing of the step-cell kernels (bench.py's roofline leg).
// Off by default; when on, every traced launch is bracketed by two events recorded on the launch stream and
// tagged with its kind and its ALGORITHMIC FLOPs.  Not for use under graph capture.
#pragma once
#include <hip/hip_runtime.h>

namespace fh {
enum TraceKind { kTraceFwdCell = 0, kTraceBwdCell = 1, kTraceGemm = 2 };
bool trace_on();
int trace_begin(hipStream_t st, int kind, double flops);  // returns a slot (or -1)
void trace_end(hipStream_t st, int slot);
}  // namespace fh
