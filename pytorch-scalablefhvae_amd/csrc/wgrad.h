// wgrad.h -- long-contraction bf16 weight-gradient GEMM (wgrad.hip): C[M,N] += A[K,M]^T . B[K,N], K = T*B rows.
#pragma once
#include "common.h"

namespace fh {

struct WgProblem {
  const u16* A;  // [K, lda] bf16, contraction index is the ROW (dgates: k = t*B + b, m = gate column)
  const u16* B;  // [K, ldb] bf16 (hidden states / inputs: n = feature column)
  float* C;      // [M, ldc] f32, accumulated with atomics
  int64_t lda, ldb, ldc;
  int M, N, K;
  // filled by launch_wgrad
  int m_tiles, n_tiles, splitk, ksteps_per;
  int shared_c;  // another problem of the same launch accumulates into the same C: atomics even without a K split
  int a_col0;    // A points a_col0 columns INTO the rows of its buffer (a column slice): the buffer ends that much earlier
};

constexpr int kMaxWgProblems = 16;
struct WgGroup {
  int n;
  int base[kMaxWgProblems + 1];  // problem i owns the logical workgroups [base[i], base[i+1])
  WgProblem p[kMaxWgProblems];
};

// alignment / range preconditions of the kernel (16-byte LDS-DMA pieces, 32-bit buffer offsets)
bool wgrad_eligible(const WgProblem& p);
// any number of eligible problems: grouped by tile class, split over K so that one launch fills the chip once
int launch_wgrad(const WgProblem* ps, int n, hipStream_t st);

}  // namespace fh
