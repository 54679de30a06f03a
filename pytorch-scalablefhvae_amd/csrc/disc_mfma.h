// disc_mfma.h -- argument block and host entry points of the matrix-core K5 kernels (disc_mfma.hip: exact-f32 MFMA;
// disc_lp.hip: bf16 MFMA with split operands, the bf16 compute mode).
#pragma once
#include "common.h"

namespace fh {

struct DiscMfmaArgs {
  const float* X;  // stationary [NX, D]
  const float* Y;  // streamed   [NY, D]
  int NX, NY;
  float c;
  int x_is_query;
  const int64_t* idx;  // per QUERY target row (global); query b hits local row idx[b] - row0
  int64_t row0;
  const float* rmax;   // per query (MODE 1)
  const float* rsum;
  const float* gsc;    // device scalar
  float gmul;
  float2* part;        // MODE 0: [nchunks][NX]
  float* G;            // MODE 1: [NX, D] accumulated with atomics
  // MODE 2 (one pass, no atomics): G = partials [chunks][NX, D] of the stationary side's gradient; the streamed side's
  // G2 = [x-tiles][NY, D] partial sums of w[y,x] X[x] and WY = [x-tiles][NY] of w[y,x] (reduced by disc_mfma.hip's kernels)
  float* G2;
  float* WY;
  int chunk;           // streamed vectors per workgroup (multiple of 64)
};

// streamed vectors per workgroup for about `target` workgroups (see disc_mfma.hip)
int mfma_chunk(int64_t nx, int64_t ny, int target = 1024);

bool disc_mfma_supported(int64_t B, int64_t S, int64_t D);
int64_t disc_mfma_ws_bytes(int64_t B, int64_t S);
// lp != 0: the bf16 split-operand kernels (D == 32 only; otherwise the f32 ones run)
int disc_mfma_fwd(const float* q, const float* table, const int64_t* idx, int64_t row0, float c, float2* part, int* nchunks,
                  int64_t B, int64_t S, int64_t D, int lp, hipStream_t st);
// ws / ws_bytes: workspace (or NULL / 0): with it, dq AND dtable wanted and a kernel that has the one-pass form, both gradients
// come from one recomputation of the logits, the queries in groups of as many 256-query tiles as the workspace holds the
// partial sums of; otherwise (or with less than one tile's worth) one pass per gradient.  disc_onepass_ws_bytes = the
// recommended size: the whole problem in one group up to kOnePassWsCap.
constexpr int64_t kOnePassWsCap = 3LL << 29;  // 1.5 GiB
int64_t disc_onepass_ws_bytes(int64_t B, int64_t S, int64_t D);
int disc_mfma_bwd(const float* q, const float* table, const int64_t* idx, int64_t row0, float c, const float* rmax,
                  const float* rsum, const float* gsc, float gmul, float* dq, float* dtable, float* ws, int64_t ws_bytes,
                  int64_t B, int64_t S, int64_t D, int lp, hipStream_t st);
// disc_lp.hip
void disc_lp_launch(const DiscMfmaArgs& a, int mode, dim3 grid, hipStream_t st);

}  // namespace fh
