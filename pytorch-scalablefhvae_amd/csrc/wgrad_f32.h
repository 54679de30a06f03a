// wgrad_f32.h -- long-contraction f32 weight-gradient GEMM (wgrad_f32.hip): C[M,N] += A[K,M]^T . B[K,N], K = T*B rows, exact-f32 MFMA.
#pragma once
#include "common.h"

namespace fh {

struct WgProblem32 {
  const float* A;  // [K, lda]: the contraction index is the ROW (dgates: k = t*B + b, m = gate column)
  const float* B;  // [K, ldb] (hidden states / inputs: n = feature column)
  float* C;        // [M, ldc], accumulated
  int64_t lda, ldb, ldc;
  int M, N, K;
  // filled by launch_wgrad32
  int m_tiles, n_tiles, splitk, ksteps_per;
  int shared_c;
};

constexpr int kMaxWg32Problems = 16;

bool wgrad32_eligible(const WgProblem32& p);
// any number of eligible problems: grouped by tile class, split over K so that one launch fills the chip once
int launch_wgrad32(const WgProblem32* ps, int n, hipStream_t st);

}  // namespace fh
