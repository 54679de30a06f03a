// proj.h -- bf16 projection GEMM C[M,N] = A[M,K] . B[N,K]^T (+ bias), f32 out (proj.hip).
#pragma once
#include "common.h"

namespace fh {

// preconditions: 16-byte aligned bases, K % 64 == 0, N % 4 == 0, lda / ldb multiples of 8, operands below 2^31 / 2^30 bytes
bool proj_eligible(const void* a, int64_t lda, const void* b, int64_t ldb, const float* c, int64_t ldc, int64_t M, int64_t N, int64_t K);
int launch_proj(const void* a, int64_t lda, const void* b, int64_t ldb, float* c, int64_t ldc, const float* bias, int64_t M, int64_t N,
                int64_t K, hipStream_t st, const float* bias2 = nullptr, int bias_split = 0);

}  // namespace fh
