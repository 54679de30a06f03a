// trace.hip -- see trace.h.  The only process-global state of the library, and only while enabled.
#include "trace.h"

#include <mutex>
#include <vector>

#include "../../include/fhvae_hip.h"

namespace fh {
struct Rec {
  hipEvent_t e0, e1;
  int kind;
  double flops;
  bool closed;
};
static std::mutex g_mu;
static bool g_on = false;
static std::vector<Rec> g_recs;        // events are created once and reused
static size_t g_used = 0;

bool trace_on() { return g_on; }

int trace_begin(hipStream_t st, int kind, double flops) {
  if (!g_on) return -1;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_used == g_recs.size()) {
    if (g_recs.size() >= 65536) return -1;
    Rec r = {};
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return -1;
    g_recs.push_back(r);
  }
  Rec& r = g_recs[g_used];
  r.kind = kind;
  r.flops = flops;
  r.closed = false;
  hipEventRecord(r.e0, st);
  return (int)g_used++;
}

void trace_end(hipStream_t st, int slot) {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  Rec& r = g_recs[slot];
  hipEventRecord(r.e1, st);
  r.closed = true;
}
}  // namespace fh

using namespace fh;

extern "C" int fhvae_trace_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_on = on != 0;
  g_used = 0;
  return FHVAE_OK;
}

extern "C" int64_t fhvae_trace_collect(float* ms, int32_t* kind, double* flops, int64_t cap) {
  std::lock_guard<std::mutex> lk(g_mu);
  int64_t n = 0;
  for (size_t i = 0; i < g_used && n < cap; ++i) {
    Rec& r = g_recs[i];
    if (!r.closed) continue;
    if (hipEventSynchronize(r.e1) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) continue;
    if (ms) ms[n] = t;
    if (kind) kind[n] = r.kind;
    if (flops) flops[n] = r.flops;
    ++n;
  }
  g_used = 0;
  return n;
}
