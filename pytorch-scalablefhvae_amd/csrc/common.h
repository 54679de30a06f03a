// common.h -- shared host/device helpers for libfhvae_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fhvae_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#define FH_CHECK_PTR(p) \
  do {                  \
    if ((p) == nullptr) return FHVAE_ERR_NULL; \
  } while (0)
#define FH_CHECK_POS(v) \
  do {                  \
    if ((v) <= 0) return FHVAE_ERR_SHAPE; \
  } while (0)
#define FH_CHECK_I32(v) \
  do {                  \
    if ((v) > 0x7fffffffLL) return FHVAE_ERR_LIMIT; \
  } while (0)

// after a kernel launch: surface launch-configuration errors as positive hipError_t codes
static inline int fh_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FHVAE_OK : (int)e;
}

static inline int64_t fh_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// f32 -> bf16 (round to nearest even; NaN stays NaN via the compiler's cvt)
__device__ __forceinline__ u16 f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(u16, b);
}
__device__ __forceinline__ float bf2f(u16 h) {
  return __builtin_bit_cast(float, ((uint32_t)h) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Gate nonlinearities on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1 ulp each): the libm
// expf/tanhf expand to ~30-40 instructions with branches and made the cell epilogue cost more than its GEMM.
// Absolute error ~1e-7 on values in (-1,1): inside the 1e-4 parity budget (checked by the LSTM parity tests).
// (__builtin_amdgcn_rcpf = one v_rcp_f32, 1 ulp; __frcp_rn expands to the ten-instruction correctly-rounded division)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
  // tanh(x) = 1 - 2/(1+exp(2x)); exp overflow -> rcp(inf) = 0 -> 1, underflow -> 1-2 = -1
  // (an explicit fma: left to -ffp-contract the compiler fused this differently in different unrolled copies of an
  //  epilogue, and a row's result then depended on which tile slot of a wave it occupied)
  return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)), 1.0f);
}
