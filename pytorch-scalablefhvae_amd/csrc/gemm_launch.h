// gemm_launch.h -- host-side launcher of the generic GEMM kernel (gemm.hip), shared by lstm.hip.
#pragma once
#include "gemm_core.h"

namespace fh {

struct GemmParams {
  Seg seg[2];
  int M, N;
  float* C;        // f32 output [M,N] (may be NULL if Clp given)
  float* C2;       // optional: output rows >= c_split go to C2[(row - c_split), :] (two matrices stacked along M)
  int c_split;
  int64_t ldc;
  u16* Clp;        // optional bf16 copy of the output
  int64_t ldclp;
  const float* bias;   // [N] added once (may be NULL)
  const float* bias2;  // second [N] vector (b_ih + b_hh)
  int relu;
  int mode;    // 0: C = v   1: C += v   2: atomicAdd(C, v)  (required when splitk > 1)
  int splitk;  // number of K slices (gridDim.z)
};

int launch_gemm(const GemmParams& p, int dtype, hipStream_t st);
constexpr int kMaxGroup = 4;
struct GemmGroup {
  int n;
  int zbase[kMaxGroup + 1];  // problem i owns blockIdx.z in [zbase[i], zbase[i+1]) = its K slices
  GemmParams p[kMaxGroup];
};
// several independent problems in one launch where they share a shape class (else one launch each)
int launch_gemm_group(const GemmParams* ps, int n, int dtype, hipStream_t st);
int pick_splitk(int64_t M, int64_t N, int64_t K);
// split == 0: db and db2 both receive all N column sums; split > 0: columns < split -> db, the others -> db2
int launch_colsum(const void* g, int dtype, int64_t ldg, float* db, float* db2, int64_t M, int64_t N, hipStream_t st,
                  int64_t split = 0);

}  // namespace fh
