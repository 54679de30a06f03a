// data.hip -- the two HBM-streaming kernels of the "next" rows of SURVEY section 8f:
//   fhvae_segment_gather : cut (B,T,F) training segments out of an utterance pool resident in HBM, with mean/variance
//                          normalisation fused (replaces NumpyDataset.__getitem__ + apply_mvn + the DataLoader collate,
//                          datasets.py:100-105, :214-223, train_model.py:379-395);
//   fhvae_mu2_estimate   : closed-form per-sequence mu2 posterior mean, a segmented reduction over the sequence index
//                          (utils.estimate_mu2_dict, utils.py:45-60).
#include "common.h"

namespace fh {

// one thread = V consecutive features of one output frame (V = 4: 16-byte accesses when F % 4 == 0); output
// batch-major (B,T,F) and/or time-major (T,B,F)
template <int V>
__global__ void segment_gather_kernel(const float* __restrict__ pool, const int64_t* __restrict__ start,
                                      const float* __restrict__ mean, const float* __restrict__ inv_std,
                                      float* __restrict__ out_btf, float* __restrict__ out_tbf, int64_t B, int64_t T_,
                                      int64_t F, int64_t pool_frames, int32_t* oob) {
  const int64_t FV = F / V;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T_ * FV) return;
  const int64_t f = (i % FV) * V, t = (i / FV) % T_, b = i / (FV * T_);
  const int64_t fr = start[b] + t;
  float v[V];
#pragma unroll
  for (int e = 0; e < V; ++e) v[e] = 0.f;
  if (fr >= 0 && fr < pool_frames) {
    if constexpr (V == 4) {
      const float4 u = *(const float4*)(pool + fr * F + f);
      v[0] = u.x, v[1] = u.y, v[2] = u.z, v[3] = u.w;
    } else {
      v[0] = pool[fr * F + f];
    }
    if (mean) {
#pragma unroll
      for (int e = 0; e < V; ++e) v[e] = (v[e] - mean[f + e]) * inv_std[f + e];
    }
  } else if (oob && f == 0) {
    atomicOr(oob, 1);
  }
  const int64_t o1 = (b * T_ + t) * F + f, o2 = (t * B + b) * F + f;
  if constexpr (V == 4) {
    if (out_btf) *(float4*)(out_btf + o1) = make_float4(v[0], v[1], v[2], v[3]);
    if (out_tbf) *(float4*)(out_tbf + o2) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    if (out_btf) out_btf[o1] = v[0];
    if (out_tbf) out_tbf[o2] = v[0];
  }
}

__global__ void mu2_accum_kernel(const float* __restrict__ z, const int64_t* __restrict__ idx, float* __restrict__ zsum,
                                 float* __restrict__ cnt, int64_t N, int64_t S, int64_t D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * D) return;
  const int64_t n = i / D, d = i % D;
  const int64_t s = idx[n];
  if (s < 0 || s >= S) return;
  atomicAdd(zsum + s * D + d, z[i]);
  if (d == 0) atomicAdd(cnt + s, 1.f);
}

__global__ void mu2_finalize_kernel(const float* __restrict__ zsum, const float* __restrict__ cnt, float* __restrict__ mu2,
                                    int64_t S, int64_t D, float ratio) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S * D) return;
  const float n = cnt[i / D];
  mu2[i] = n > 0.f ? zsum[i] / (n + ratio) : 0.f;  // utils.py:57-59
}

}  // namespace fh

using namespace fh;

extern "C" int fhvae_segment_gather(const float* pool, int64_t pool_frames, const int64_t* start, const float* mean,
                                    const float* inv_std, float* out_btf, float* out_tbf, int64_t B, int64_t T, int64_t F,
                                    int32_t* oob_flag, void* stream) {
  FH_CHECK_PTR(pool);
  FH_CHECK_PTR(start);
  if (!out_btf && !out_tbf) return FHVAE_ERR_NULL;
  if ((mean == nullptr) != (inv_std == nullptr)) return FHVAE_ERR_NULL;
  FH_CHECK_POS(pool_frames);
  FH_CHECK_POS(B);
  FH_CHECK_POS(T);
  FH_CHECK_POS(F);
  const bool vec = F % 4 == 0 && ((((uintptr_t)pool | (uintptr_t)out_btf | (uintptr_t)out_tbf) & 15) == 0);
  if (vec) {
    const int64_t n = B * T * (F / 4);
    hipLaunchKernelGGL(segment_gather_kernel<4>, dim3((unsigned)fh_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pool, start,
                       mean, inv_std, out_btf, out_tbf, B, T, F, pool_frames, oob_flag);
  } else {
    const int64_t n = B * T * F;
    hipLaunchKernelGGL(segment_gather_kernel<1>, dim3((unsigned)fh_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pool, start,
                       mean, inv_std, out_btf, out_tbf, B, T, F, pool_frames, oob_flag);
  }
  return fh_launch_status();
}

extern "C" int fhvae_mu2_accumulate(const float* z2_mu, const int64_t* idx, float* zsum, float* count, int64_t N, int64_t S,
                                    int64_t D, void* stream) {
  FH_CHECK_PTR(z2_mu);
  FH_CHECK_PTR(idx);
  FH_CHECK_PTR(zsum);
  FH_CHECK_PTR(count);
  FH_CHECK_POS(N);
  FH_CHECK_POS(S);
  FH_CHECK_POS(D);
  hipLaunchKernelGGL(mu2_accum_kernel, dim3((unsigned)fh_cdiv(N * D, 256)), dim3(256), 0, (hipStream_t)stream, z2_mu, idx, zsum,
                     count, N, S, D);
  return fh_launch_status();
}

extern "C" int fhvae_mu2_finalize(const float* zsum, const float* count, float* mu2, int64_t S, int64_t D, float ratio,
                                  void* stream) {
  FH_CHECK_PTR(zsum);
  FH_CHECK_PTR(count);
  FH_CHECK_PTR(mu2);
  FH_CHECK_POS(S);
  FH_CHECK_POS(D);
  hipLaunchKernelGGL(mu2_finalize_kernel, dim3((unsigned)fh_cdiv(S * D, 256)), dim3(256), 0, (hipStream_t)stream, zsum, count,
                     mu2, S, D, ratio);
  return fh_launch_status();
}
