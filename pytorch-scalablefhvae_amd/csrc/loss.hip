// loss.hip -- the HBM-streaming half of the hot path:
//   K3 fused lower bound (simple_fhvae.py:105-116), K4 mu2 gather (:53), K5 discriminative
//   log-sum-exp cross-entropy over the mu2 table (:119-122), fused Adam (train_model.py:409-411),
//   layout utilities.
#include "common.h"
#include "disc_mfma.h"
#include <cstdlib>

namespace fh {

// prior constants of the reference (simple_fhvae.py:22-23, :88): float32(log 1) and float32(log 0.25)
__device__ constexpr float kPz2Logvar = -1.3862943649291992f;  // np.log(0.5**2).astype(np.float32)
__device__ constexpr float kLog2Pi = 1.8378770664093453f;

// ---------------------------------------------------------------------------------------------
// layout utilities
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void to_time_major_kernel(const float* __restrict__ x, T* __restrict__ o, float* __restrict__ of, int64_t B,
                                     int64_t T_, int64_t F) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // index into (T,B,F)
  if (i >= B * T_ * F) return;
  int64_t f = i % F, b = (i / F) % B, t = i / (F * B);
  float v = x[(b * T_ + t) * F + f];
  if constexpr (sizeof(T) == 4)
    o[i] = v;
  else
    o[i] = f2bf(v);
  if (of) of[i] = v;
}

// the same with four features per thread (F % 4 == 0, 16-byte aligned buffers): 16-byte loads and f32 stores, 8-byte bf16 stores
template <typename T>
__global__ void to_time_major4_kernel(const float* __restrict__ x, T* __restrict__ o, float* __restrict__ of, int B, int T_, int F4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // index into (T,B,F/4)
  if (i >= (int64_t)B * T_ * F4) return;
  const int f = (int)(i % F4);
  const int64_t tb = i / F4;
  const int b = (int)(tb % B), t = (int)(tb / B);
  const float4 v = ((const float4*)x)[((int64_t)b * T_ + t) * F4 + f];
  if constexpr (sizeof(T) == 4) {
    ((float4*)o)[i] = v;
  } else {
    ((uint2*)o)[i] = uint2{(uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16), (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16)};
  }
  if (of) ((float4*)of)[i] = v;
}

__global__ void cast_bf16_kernel(const float* __restrict__ s, u16* __restrict__ d, u16* __restrict__ dt, int64_t R,
                                 int64_t C) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * C) return;
  u16 v = f2bf(s[i]);
  if (d) d[i] = v;
  if (dt) dt[(i % C) * R + i / C] = v;
}

// ---------------------------------------------------------------------------------------------
// K4 gather / scatter-add
// ---------------------------------------------------------------------------------------------
__global__ void gather_fwd_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx, int64_t off,
                                  float* __restrict__ out, int64_t B, int64_t S, int64_t D, int32_t* oob) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  int64_t b = i / D, d = i % D;
  int64_t s = idx[b] - off;
  if (s < 0 || s >= S) {
    out[i] = 0.f;
    if (oob && d == 0) atomicOr(oob, 1);
    return;
  }
  out[i] = table[s * D + d];
}

__global__ void gather_bwd_kernel(const float* __restrict__ dmu2, const int64_t* __restrict__ idx, int64_t off,
                                  float* __restrict__ dtable, int64_t B, int64_t S, int64_t D, float scale) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  int64_t b = i / D, d = i % D;
  int64_t s = idx[b] - off;
  if (s < 0 || s >= S) return;
  atomicAdd(dtable + s * D + d, scale * dmu2[i]);
}

// ---------------------------------------------------------------------------------------------
// K3 fused lower bound: one wave per segment
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ float nll_term(float x, float mu, float lv) {
  const float df = x - mu;
  return lv + df * df / expf(lv);  // log_gauss without the constant and the -0.5 (simple_fhvae.py:58-60)
}

__global__ __launch_bounds__(256) void elbo_fwd_kernel(fhvae_elbo_desc d) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= d.B) return;
  const float* x = d.x + b * d.x_sb;
  const float* xm = d.x_mu + b * d.xo_sb;
  const float* xl = d.x_lv + b * d.xo_sb;
  float s = 0.f;
  const bool vec = (d.F % 4 == 0) && (d.x_sb % 4 == 0) && (d.x_st % 4 == 0) && (d.xo_sb % 4 == 0) &&
                   (d.xo_st % 4 == 0) && ((((uintptr_t)d.x | (uintptr_t)d.x_mu | (uintptr_t)d.x_lv) & 15) == 0);
  if (vec) {
    const int F4 = (int)(d.F / 4);
    const int n4 = (int)d.T * F4;
    for (int i = lane; i < n4; i += 64) {
      const int t = i / F4, f = (i % F4) * 4;
      const float4 a = *(const float4*)(x + t * d.x_st + f);
      const float4 m = *(const float4*)(xm + t * d.xo_st + f);
      const float4 l = *(const float4*)(xl + t * d.xo_st + f);
      s += nll_term(a.x, m.x, l.x) + nll_term(a.y, m.y, l.y) + nll_term(a.z, m.z, l.z) + nll_term(a.w, m.w, l.w);
    }
  } else {
    const int n = (int)(d.T * d.F);
    for (int i = lane; i < n; i += 64) {
      const int t = i / (int)d.F, f = i % (int)d.F;
      s += nll_term(x[t * d.x_st + f], xm[t * d.xo_st + f], xl[t * d.xo_st + f]);
    }
  }
  s = wave_sum(s);
  const float log_px_z = -0.5f * ((float)(d.T * d.F) * kLog2Pi + s);

  // KL(q(z1|x) || N(0,1)) and KL(q(z2|x) || N(mu2, 0.25)), log N(mu2; 0, 1)   (simple_fhvae.py:62-69,106-112)
  const float v2 = expf(kPz2Logvar);
  float k1 = 0.f, k2 = 0.f, pm = 0.f;
  for (int j = lane; j < d.D1; j += 64) {
    const float mu = d.z1_mu[b * d.D1 + j], lv = d.z1_lv[b * d.D1 + j];
    k1 += 1.f + lv - 0.f - (mu * mu + expf(lv)) / 1.f;
  }
  for (int j = lane; j < d.D2; j += 64) {
    const float mu = d.z2_mu[b * d.D2 + j], lv = d.z2_lv[b * d.D2 + j], m2 = d.mu2[b * d.D2 + j];
    const float df = mu - m2;
    k2 += 1.f + lv - kPz2Logvar - (df * df + expf(lv)) / v2;
    pm += kLog2Pi + m2 * m2;
  }
  k1 = 0.5f * wave_sum(k1);  // neg_kld = -sum(kld) = 0.5 * sum(...)
  k2 = 0.5f * wave_sum(k2);
  pm = -0.5f * wave_sum(pm);
  if (lane == 0) {
    const float ns = d.num_segs ? (float)d.num_segs[b] : (float)d.nsegs_scalar;
    d.log_px_z[b] = log_px_z;
    d.neg_kld_z1[b] = k1;
    d.neg_kld_z2[b] = k2;
    d.log_pmu2[b] = pm;
    d.lower_bound[b] = log_px_z + k1 + k2 + pm / ns;
  }
}

__device__ __forceinline__ void elbo_dx4(const float4& a, const float4& mu, const float4& lv, float gpx, float4& gm, float4& gl) {
  const float av[4] = {a.x, a.y, a.z, a.w}, mv[4] = {mu.x, mu.y, mu.z, mu.w}, lvv[4] = {lv.x, lv.y, lv.z, lv.w};
  float m[4], l[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float df = av[k] - mv[k], iv = 1.f / expf(lvv[k]);
    m[k] = gpx * df * iv;
    l[k] = gpx * -0.5f * (1.f - df * df * iv);
  }
  gm = make_float4(m[0], m[1], m[2], m[3]);
  gl = make_float4(l[0], l[1], l[2], l[3]);
}

struct ElboG {
  float gpx, gk1, gk2, gpm;
};
__device__ __forceinline__ ElboG elbo_upstream(const fhvae_elbo_bwd_desc& bd, int64_t b) {
  const fhvae_elbo_desc& d = bd.f;
  const float glb = bd.g_lower_bound ? bd.g_lower_bound[b] : 0.f;
  ElboG g;
  g.gpx = glb + (bd.g_log_px_z ? bd.g_log_px_z[b] : 0.f);
  g.gk1 = glb + (bd.g_neg_kld_z1 ? bd.g_neg_kld_z1[b] : 0.f);
  g.gk2 = glb + (bd.g_neg_kld_z2 ? bd.g_neg_kld_z2[b] : 0.f);
  const float ns = d.num_segs ? (float)d.num_segs[b] : (float)d.nsegs_scalar;
  g.gpm = glb / ns + (bd.g_log_pmu2 ? bd.g_log_pmu2[b] : 0.f);
  return g;
}
// the latent terms' gradients of segment b (one wave)
__device__ __forceinline__ void elbo_dz(const fhvae_elbo_bwd_desc& bd, int64_t b, int lane, const ElboG& g) {
  const fhvae_elbo_desc& d = bd.f;
  const float v2 = expf(kPz2Logvar);
  for (int j = lane; j < d.D1; j += 64) {
    const int64_t o = b * d.D1 + j;
    const float mu = d.z1_mu[o], lv = d.z1_lv[o];
    bd.d_z1_mu[o] = -mu * g.gk1;
    bd.d_z1_lv[o] = 0.5f * (1.f - expf(lv)) * g.gk1;
  }
  for (int j = lane; j < d.D2; j += 64) {
    const int64_t o = b * d.D2 + j;
    const float mu = d.z2_mu[o], lv = d.z2_lv[o], m2 = d.mu2[o];
    const float df = (mu - m2) / v2;
    bd.d_z2_mu[o] = -df * g.gk2;
    bd.d_z2_lv[o] = 0.5f * (1.f - expf(lv) / v2) * g.gk2;
    bd.d_mu2[o] = df * g.gk2 + (bd.reference_detach ? 0.f : -m2 * g.gpm);
  }
}

__global__ __launch_bounds__(256) void elbo_bwd_kernel(fhvae_elbo_bwd_desc bd) {
  const fhvae_elbo_desc& d = bd.f;
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= d.B) return;
  const ElboG g = elbo_upstream(bd, b);
  if (bd.d_x_mu && bd.d_x_lv && !bd.reference_detach) {
    const float* x = d.x + b * d.x_sb;
    const float* xm = d.x_mu + b * d.xo_sb;
    const float* xl = d.x_lv + b * d.xo_sb;
    float* dm = bd.d_x_mu + b * d.xo_sb;
    float* dl = bd.d_x_lv + b * d.xo_sb;
    const bool vec = (d.F % 4 == 0) && (d.x_sb % 4 == 0) && (d.x_st % 4 == 0) && (d.xo_sb % 4 == 0) && (d.xo_st % 4 == 0) &&
                     ((((uintptr_t)d.x | (uintptr_t)d.x_mu | (uintptr_t)d.x_lv | (uintptr_t)bd.d_x_mu | (uintptr_t)bd.d_x_lv) & 15) == 0);
    if (vec) {
      const int F4 = (int)(d.F / 4), n4 = (int)d.T * F4;
      for (int i = lane; i < n4; i += 64) {
        const int t = i / F4, f = (i % F4) * 4;
        const int64_t o = t * d.xo_st + f;
        float4 gm, gl;
        elbo_dx4(*(const float4*)(x + t * d.x_st + f), *(const float4*)(xm + o), *(const float4*)(xl + o), g.gpx, gm, gl);
        *(float4*)(dm + o) = gm;
        *(float4*)(dl + o) = gl;
      }
    } else {
      const int n = (int)(d.T * d.F);
      for (int i = lane; i < n; i += 64) {
        const int t = i / (int)d.F, f = i % (int)d.F;
        const int64_t o = t * d.xo_st + f;
        const float df = x[t * d.x_st + f] - xm[o];
        const float iv = 1.f / expf(xl[o]);
        dm[o] = g.gpx * df * iv;
        dl[o] = g.gpx * -0.5f * (1.f - df * df * iv);
      }
    }
  }
  elbo_dz(bd, b, lane, g);
}

// The same with the bf16 pair copy of [d_x_mu | d_x_lv] and its column sums (fhvae_elbo_bwd_desc::d_x_pair_lp).  Two waves per
// segment, two segments per workgroup (twice the waves of the kernel above: the passes of a wave are latency-bound); lanes ->
// (row of the pass, 4 columns), so every lane keeps ONE column group over all its rows and the column sums stay in registers
// (F = 80: 3 rows x 20 groups per pass, 4 lanes idle); a wave's passes are loaded four at a time before any is used.
// Column sums: one partial row per workgroup (no atomics; fhvae_gauss_head_bwd_pair reduces the rows).
__global__ __launch_bounds__(256) void elbo_bwd_pair_kernel(fhvae_elbo_bwd_desc bd) {
  const fhvae_elbo_desc& d = bd.f;
  __shared__ float cs[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 2 + (wave >> 1);
  const int half = wave & 1;
  const int F = (int)d.F, F4 = F / 4, RP = 64 / F4, T = (int)d.T;
  const int rsub = lane / F4, f = (lane % F4) * 4;
  const bool act = lane < RP * F4 && b < d.B;
  float4 sm = make_float4(0.f, 0.f, 0.f, 0.f), sl = sm;
  if (b < d.B) {
    const ElboG g = elbo_upstream(bd, b);
    const float* x = d.x + b * d.x_sb;
    const float* xm = d.x_mu + b * d.xo_sb;
    const float* xl = d.x_lv + b * d.xo_sb;
    float* dm = bd.d_x_mu + b * d.xo_sb;
    float* dl = bd.d_x_lv + b * d.xo_sb;
    u16* gp = (u16*)bd.d_x_pair_lp;
    const int npass = (T + RP - 1) / RP;
    for (int p0 = half; p0 < npass; p0 += 8) {  // this wave's passes p0, p0 + 2, p0 + 4, p0 + 6
      float4 a[4], mu[4], lv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = (p0 + 2 * k) * RP + rsub;
        if (act && t < T) {
          const int64_t o = t * d.xo_st + f;
          a[k] = *(const float4*)(x + t * d.x_st + f);
          mu[k] = *(const float4*)(xm + o);
          lv[k] = *(const float4*)(xl + o);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = (p0 + 2 * k) * RP + rsub;
        if (act && t < T) {
          const int64_t o = t * d.xo_st + f;
          float4 gm, gl;
          elbo_dx4(a[k], mu[k], lv[k], g.gpx, gm, gl);
          *(float4*)(dm + o) = gm;
          *(float4*)(dl + o) = gl;
          u16* row = gp + ((int64_t)t * d.B + b) * bd.ld_pair;
          *(uint2*)(row + f) = uint2{(uint32_t)f2bf(gm.x) | ((uint32_t)f2bf(gm.y) << 16), (uint32_t)f2bf(gm.z) | ((uint32_t)f2bf(gm.w) << 16)};
          *(uint2*)(row + F + f) = uint2{(uint32_t)f2bf(gl.x) | ((uint32_t)f2bf(gl.y) << 16), (uint32_t)f2bf(gl.z) | ((uint32_t)f2bf(gl.w) << 16)};
          sm.x += gm.x, sm.y += gm.y, sm.z += gm.z, sm.w += gm.w;
          sl.x += gl.x, sl.y += gl.y, sl.z += gl.z, sl.w += gl.w;
        }
      }
    }
    const int padc = (int)(bd.ld_pair - 2 * F) / 8;  // the zero padding of each row, in 16-byte pieces
    for (int i = lane + 64 * half; i < T * padc; i += 128) {
      const int t = i / padc, c = i % padc;
      *(uint4*)(gp + ((int64_t)t * d.B + b) * bd.ld_pair + 2 * F + c * 8) = make_uint4(0u, 0u, 0u, 0u);
    }
    if (half == 0) elbo_dz(bd, b, lane, g);
  }
  // column sums: lanes of one column group across the RP row slots (shuffles stay inside the wave), then the 4 waves through LDS
  for (int j = lane; j < 2 * F; j += 64) cs[wave][j] = 0.f;
  __syncthreads();
  if (lane < RP * F4) {
    atomicAdd(&cs[wave][f + 0], sm.x), atomicAdd(&cs[wave][f + 1], sm.y), atomicAdd(&cs[wave][f + 2], sm.z), atomicAdd(&cs[wave][f + 3], sm.w);
    atomicAdd(&cs[wave][F + f + 0], sl.x), atomicAdd(&cs[wave][F + f + 1], sl.y), atomicAdd(&cs[wave][F + f + 2], sl.z),
        atomicAdd(&cs[wave][F + f + 3], sl.w);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * F; j += 256)
    bd.d_x_colsum[(int64_t)blockIdx.x * 2 * F + j] = (cs[0][j] + cs[1][j]) + (cs[2][j] + cs[3][j]);
}

// ---------------------------------------------------------------------------------------------
// K5 discriminative log-sum-exp cross-entropy.
// Forward: thread = query b (q row in registers), table rows are wave-uniform -> scalar loads
// (s_load_dwordx*), so per (b,s) pair the VALU does only the 2*D sub/fma and the online-LSE
// update; nothing of size B*S is written.  grid = (query tiles of 256) x (row chunks).
// ---------------------------------------------------------------------------------------------
struct DiscPlan {
  int chunk;    // table rows per workgroup
  int nchunks;
  int btiles;
};
static inline DiscPlan disc_plan(int64_t B, int64_t S) {
  DiscPlan p;
  p.btiles = (int)fh_cdiv(B, 256);
  int64_t want = fh_cdiv(1024, p.btiles);  // aim at ~1024 workgroups
  int64_t chunk = fh_cdiv(S, want);
  chunk = fh_cdiv(chunk, 8) * 8;
  if (chunk < 8) chunk = 8;
  p.chunk = (int)chunk;
  p.nchunks = (int)fh_cdiv(S, chunk);
  return p;
}

template <int D>
__device__ __forceinline__ float sqdist(const float (&q)[D], const float* __restrict__ trow) {
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int d = 0; d < D; d += 2) {
    const float d0 = q[d] - trow[d], d1 = q[d + 1] - trow[d + 1];
    a0 = fmaf(d0, d0, a0);
    a1 = fmaf(d1, d1, a1);
  }
  return a0 + a1;
}

template <int D>
__global__ __launch_bounds__(256) void disc_fwd_kernel(const float* __restrict__ q, const float* __restrict__ table,
                                                       float c, float2* __restrict__ part, int B, int S, int chunk) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  const int bb = b < B ? b : B - 1;
  float qr[D];
#pragma unroll
  for (int d = 0; d < D; ++d) qr[d] = q[(int64_t)bb * D + d];
  const int s0 = blockIdx.y * chunk;
  const int s1 = min(S, s0 + chunk);
  float m = -INFINITY, sum = 0.f;
  for (int s = s0; s < s1; s += 8) {
    float l[8];
    float gm = -INFINITY;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int su = s + u;
      // rows past the chunk end are clamped (uniform scalar address) and masked to -inf
      const float* trow = table + (int64_t)(su < s1 ? su : s1 - 1) * D;
      l[u] = su < s1 ? -c * sqdist<D>(qr, trow) : -INFINITY;
      gm = fmaxf(gm, l[u]);
    }
    if (gm > m) {
      sum *= __expf(m - gm);  // m = -inf on the first group: exp(-inf) = 0
      m = gm;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += __expf(l[u] - m);
  }
  if (b < B) part[(int64_t)blockIdx.y * B + b] = make_float2(m, sum);
}

// one WAVE per query: lanes stride over the chunk partials (a thread-per-query loop was a chain of nchunks
// dependent L2 loads: 165 us for 144 chunks), then a wave-level (max, sum) merge; lane 0 also evaluates the
// target logit.
template <int D>
__global__ __launch_bounds__(256) void disc_combine_kernel(const float* __restrict__ q, const float* __restrict__ table,
                                                           const int64_t* __restrict__ idx, int64_t row0, float c,
                                                           const float2* __restrict__ part, int nchunks,
                                                           float* __restrict__ row_max, float* __restrict__ row_sum,
                                                           float* __restrict__ tgt, int B, int S, int own_excluded) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float m = -INFINITY, sum = 0.f;
  for (int k = lane; k < nchunks; k += 64) {
    const float2 p = part[(int64_t)k * B + b];
    if (p.x > m) {
      sum = sum * __expf(m - p.x) + p.y;
      m = p.x;
    } else {
      sum += p.y * __expf(p.x - m);
    }
  }
  const float gm = wave_max(m);
  sum = m == -INFINITY ? 0.f : sum * __expf(m - gm);
  sum = wave_sum(sum);
  // target logit with EXACTLY the arithmetic of disc_fwd_kernel (same sqdist order), so that a target that is the
  // row maximum gives (max - target) == 0 bit for bit
  const int64_t s = idx[b] - row0;
  if (lane == 0) {
    float t = 0.f;
    if (s >= 0 && s < S) {
      float qr[D];
#pragma unroll
      for (int d = 0; d < D; ++d) qr[d] = q[(int64_t)b * D + d];
      t = -c * sqdist<D>(qr, table + s * D);
      if (own_excluded) {
        // the MFMA kernels left the query's own row out of the partials (disc_mfma.hip): its exact logit joins here
        const float nm = fmaxf(gm, t);
        sum = (gm == -INFINITY ? 0.f : sum * __expf(gm - nm)) + __expf(t - nm);
        row_max[b] = nm;
        row_sum[b] = sum;
        tgt[b] = t;
        return;
      }
    }
    row_max[b] = gm;
    row_sum[b] = sum;
    tgt[b] = t;
  }
}

// Backward of the (query, own row) pairs the MFMA kernels leave out: w = g (p_own - 1), p_own = exp(target - max) / sum with the
// DIRECT-form target logit; dq[b] += -2c w (q_b - t_y), dtable[y] += +2c w (q_b - t_y).  One thread per (query, 4 dims).
template <int D>
__global__ __launch_bounds__(256) void disc_own_bwd_kernel(const float* __restrict__ q, const float* __restrict__ table,
                                                           const int64_t* __restrict__ idx, int64_t row0, float c,
                                                           const float* __restrict__ rmax, const float* __restrict__ rsum,
                                                           const float* __restrict__ gsc, float gmul, float* __restrict__ dq,
                                                           float* __restrict__ dtable, int B, int S) {
  constexpr int PER = D / 4;  // threads per query
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int b = tid / PER, part = tid % PER;
  if (b >= B) return;
  const int64_t s = idx[b] - row0;
  if (s < 0 || s >= S) return;
  float qr[D];
#pragma unroll
  for (int d = 0; d < D; ++d) qr[d] = q[(int64_t)b * D + d];
  const float t = -c * sqdist<D>(qr, table + s * D);
  const float w = (*gsc) * gmul * (__expf(t - rmax[b]) / rsum[b] - 1.f);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = part * 4 + k;
    const float diff = qr[d] - table[s * D + d];
    if (dq) atomicAdd(dq + (int64_t)b * D + d, -2.f * c * w * diff);
    if (dtable) atomicAdd(dtable + s * D + d, 2.f * c * w * diff);
  }
}

// single-workgroup deterministic mean of (max + log(sumexp) - target)
__global__ __launch_bounds__(256) void ce_mean_kernel(const float* __restrict__ row_max, const float* __restrict__ row_sum,
                                                      const float* __restrict__ tgt, float* __restrict__ out, int B, float scale) {
  __shared__ float red[4];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) s += (row_max[b] - tgt[b]) + logf(row_sum[b]);  // exact 0 + log s when the target row is the max
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = scale * ((red[0] + red[1] + red[2] + red[3]) / (float)B);
}

// loss = -(mean(lower_bound) + alpha * log_qy)  (train_model.py:243-251) in one launch, and its backward in one
__global__ __launch_bounds__(256) void loss_fwd_kernel(const float* __restrict__ lb, const float* __restrict__ log_qy, float alpha,
                                                       float* __restrict__ out, int B, int* __restrict__ nan_flag) {
  __shared__ float red[4];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) s += lb[b];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float mean_lb = (red[0] + red[1] + red[2] + red[3]) / (float)B;
    *out = -(mean_lb + alpha * log_qy[0]);
    // sticky divergence word (train_model.py:464-466 tests isnan(lower_bound).any() on the host every batch: a NaN in any
    // element makes the mean NaN, so the same condition is recorded here without a host sync per step)
    if (nan_flag && mean_lb != mean_lb) atomicOr(nan_flag, 1);
  }
}
__global__ void loss_bwd_kernel(const float* __restrict__ g, float alpha, float* __restrict__ d_lb, float* __restrict__ d_qy,
                                int64_t B) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float gv = g ? g[0] : 1.f;
  if (i < B) d_lb[i] = -gv / (float)B;
  if (i == 0 && d_qy) d_qy[0] = -alpha * gv;
}

// ---- helpers of the row-sharded table's exchange (dist_shard.py): one launch each instead of 4-6 elementwise ATen launches
// pack: out[b] = [q[b, 0..D) | bits of int32(idx[b])] -- the queries and their row indices travel in ONE all-gather
__global__ void shard_pack_kernel(const float* __restrict__ q, const int64_t* __restrict__ idx, float* __restrict__ out, int64_t B, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * (D + 1)) return;
  const int64_t b = i / (D + 1);
  const int d = (int)(i - b * (D + 1));
  out[i] = d < D ? q[b * D + d] : __int_as_float((int)idx[b]);
}
__global__ void shard_unpack_kernel(const float* __restrict__ pk, float* __restrict__ q, int64_t* __restrict__ idx, int64_t N, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * (D + 1)) return;
  const int64_t b = i / (D + 1);
  const int d = (int)(i - b * (D + 1));
  if (d < D)
    q[b * D + d] = pk[i];
  else
    idx[b] = (int64_t)__float_as_int(pk[i]);
}
// merge of the W ranks' K5 partials (parts[w] = [max | sumexp | target], N each): m = max_w, s = sum_w sumexp_w exp(max_w - m),
// t = sum_w target_w.  An empty shard's (-inf, 0, 0) contributes exp(-inf) * 0 = 0.
__global__ void disc_merge_kernel(const float* __restrict__ parts, float* __restrict__ m, float* __restrict__ s, float* __restrict__ t,
                                  int W, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float mx = -INFINITY;
  for (int w = 0; w < W; ++w) mx = fmaxf(mx, parts[((int64_t)w * 3 + 0) * N + i]);
  float ss = 0.f, tt = 0.f;
  for (int w = 0; w < W; ++w) {
    const float rm = parts[((int64_t)w * 3 + 0) * N + i], rs = parts[((int64_t)w * 3 + 1) * N + i];
    ss += rs > 0.f ? rs * __expf(rm - mx) : 0.f;
    tt += parts[((int64_t)w * 3 + 2) * N + i];
  }
  m[i] = mx, s[i] = ss, t[i] = tt;
}
// backward buffer [dq * scale | dmu2 rows of the local queries (zeros elsewhere)], N x 2D
__global__ void shard_bwd_pack_kernel(const float* __restrict__ dq, float scale, const float* __restrict__ dmu2, int64_t own0, int64_t nown,
                                      float* __restrict__ out, int64_t N, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * 2 * D) return;
  const int64_t b = i / (2 * D);
  const int d = (int)(i - b * 2 * D);
  float v = 0.f;
  if (d < D) {
    if (dq) v = dq[b * D + d] * scale;
  } else if (dmu2 && b >= own0 && b < own0 + nown) {
    v = dmu2[(b - own0) * D + (d - D)];
  }
  out[i] = v;
}
// ... and back: dq of the local queries, dmu2 of all queries (contiguous for the scatter)
__global__ void shard_bwd_unpack_kernel(const float* __restrict__ buf, int64_t own0, int64_t nown, float* __restrict__ dq_local,
                                        float* __restrict__ dmu2_all, int64_t N, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * 2 * D) return;
  const int64_t b = i / (2 * D);
  const int d = (int)(i - b * 2 * D);
  if (d < D) {
    if (dq_local && b >= own0 && b < own0 + nown) dq_local[(b - own0) * D + d] = buf[i];
  } else if (dmu2_all) {
    dmu2_all[b * D + (d - D)] = buf[i];
  }
}

// Backward, query side: dq[b,:] = -2c * sum_s w_bs (q_b - t_s), w = g (p - onehot)
template <int D>
__global__ __launch_bounds__(256) void disc_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ table,
                                                          const int64_t* __restrict__ idx, int64_t row0, float c,
                                                          const float* __restrict__ rmax, const float* __restrict__ rsum,
                                                          const float* __restrict__ gsc, float gmul,
                                                          float* __restrict__ dq, int B, int S, int chunk) {
  __shared__ float tr[256][D + 1];
  const int b = blockIdx.x * 256 + threadIdx.x;
  const int bb = b < B ? b : B - 1;
  float qr[D], V[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    qr[d] = q[(int64_t)bb * D + d];
    V[d] = 0.f;
  }
  const float g = *gsc * gmul;
  // p = exp(logit - max) / sumexp: the max is one of the logits exactly, so the subtraction is exact for
  // the rows that matter (an lse = max + log(sum) would carry the ulp of |max| ~ 1e3 into every p)
  const float mb = rmax[bb], inv_s = 1.f / rsum[bb];
  const int64_t tg = idx[bb] - row0;
  const int s0 = blockIdx.y * chunk, s1 = min(S, s0 + chunk);
  for (int s = s0; s < s1; ++s) {
    const float* trow = table + (int64_t)s * D;
    float df[D];
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int d = 0; d < D; d += 2) {
      df[d] = qr[d] - trow[d];
      df[d + 1] = qr[d + 1] - trow[d + 1];
      a0 = fmaf(df[d], df[d], a0);
      a1 = fmaf(df[d + 1], df[d + 1], a1);
    }
    const float lg = -c * (a0 + a1);
    const float w = g * (__expf(lg - mb) * inv_s - (s == tg ? 1.f : 0.f));
#pragma unroll
    for (int d = 0; d < D; ++d) V[d] = fmaf(w, df[d], V[d]);
  }
  // transpose through LDS so the atomics go out as contiguous rows (MI355X_MICROARCH.md, float atomics)
#pragma unroll
  for (int d = 0; d < D; ++d) tr[threadIdx.x][d] = -2.f * c * V[d];
  __syncthreads();
  for (int i = threadIdx.x; i < 256 * D; i += 256) {
    const int r = i / D, d = i % D;
    const int br = blockIdx.x * 256 + r;
    if (br < B) atomicAdd(dq + (int64_t)br * D + d, tr[r][d]);
  }
}

// Backward, table side: thread = table row s (row in registers), queries are wave-uniform.
// dtable[s,:] += 2c * sum_b w_bs (q_b - t_s)
template <int D>
__global__ __launch_bounds__(256) void disc_bwd_dt_kernel(const float* __restrict__ q, const float* __restrict__ table,
                                                          const int64_t* __restrict__ idx, int64_t row0, float c,
                                                          const float* __restrict__ rmax, const float* __restrict__ rsum,
                                                          const float* __restrict__ gsc, float gmul,
                                                          float* __restrict__ dtable, int B, int S, int bchunk) {
  __shared__ float tr[256][D + 1];
  const int s = blockIdx.x * 256 + threadIdx.x;
  const int ss = s < S ? s : S - 1;
  float t[D], U[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    t[d] = table[(int64_t)ss * D + d];
    U[d] = 0.f;
  }
  const float g = *gsc * gmul;
  const int b0 = blockIdx.y * bchunk, b1 = min(B, b0 + bchunk);
  for (int b = b0; b < b1; ++b) {
    const float* qrow = q + (int64_t)b * D;
    float df[D];
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int d = 0; d < D; d += 2) {
      df[d] = qrow[d] - t[d];
      df[d + 1] = qrow[d + 1] - t[d + 1];
      a0 = fmaf(df[d], df[d], a0);
      a1 = fmaf(df[d + 1], df[d + 1], a1);
    }
    const float lg = -c * (a0 + a1);
    const float w = g * (__expf(lg - rmax[b]) / rsum[b] - ((int64_t)s == idx[b] - row0 ? 1.f : 0.f));
#pragma unroll
    for (int d = 0; d < D; ++d) U[d] = fmaf(w, df[d], U[d]);
  }
#pragma unroll
  for (int d = 0; d < D; ++d) tr[threadIdx.x][d] = 2.f * c * U[d];
  __syncthreads();
  for (int i = threadIdx.x; i < 256 * D; i += 256) {
    const int r = i / D, d = i % D;
    const int sr = blockIdx.x * 256 + r;
    if (sr < S) atomicAdd(dtable + (int64_t)sr * D + d, tr[r][d]);
  }
}

// ---------------------------------------------------------------------------------------------
// Adam
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float adam_one(float& p, float g, float& m, float& v, float lr_bc1, float rs_bc2, float b1,
                                          float b2, float eps, float gscale) {
  const float gi = g * gscale;
  m = b1 * m + (1.f - b1) * gi;
  v = b2 * v + (1.f - b2) * gi * gi;
  const float denom = sqrtf(v) * rs_bc2 + eps;
  p = p - lr_bc1 * (m / denom);
  return p;
}

// 16 bytes per lane per stream (the arena is 64-element aligned); scalar tail
// flags: FHVAE_ADAM_ZERO_GRAD -- g is cleared behind its use (the next backward accumulates into zeros: no memset launch);
// FHVAE_ADAM_ADVANCE -- the step is step[0] + 1 and the last workgroup to finish stores it (every workgroup has read step[0]
// before it counts itself in, so nobody reads the new value), instead of an increment launch in front of this one.  Counting in
// two levels -- 64 group words, one per 128-byte line: step[32 (1 + (block & 63))], whose last arrivals count in step[1] --:
// 3320 device-scope atomics on ONE word took 140 us, on 64 words of two adjacent lines still 85 (~30 ns each: atomics of one
// LINE are serialised at the memory side); 52 per line on 64 lines + 64 on the top word are a chain of ~3 us
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, u16* __restrict__ plp, int64_t n, float lr, float b1, float b2,
                            float eps, float gscale, int flags, int32_t* __restrict__ step) {
  const int t_i = *(volatile int32_t*)step + ((flags & FHVAE_ADAM_ADVANCE) ? 1 : 0);
  const float t = (float)t_i;
  const bool zg = (flags & FHVAE_ADAM_ZERO_GRAD) != 0;
  const float lr_bc1 = lr / (1.f - powf(b1, t)), rs_bc2 = 1.f / sqrtf(1.f - powf(b2, t));
  const int64_t n4 = n / 4;
  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    for (int64_t k = i; k < n4; k += stride) {
      float4 pp = ((float4*)p)[k], mm = ((float4*)m)[k], vv = ((float4*)v)[k];
      const float4 gg = ((const float4*)g)[k];
      adam_one(pp.x, gg.x, mm.x, vv.x, lr_bc1, rs_bc2, b1, b2, eps, gscale);
      adam_one(pp.y, gg.y, mm.y, vv.y, lr_bc1, rs_bc2, b1, b2, eps, gscale);
      adam_one(pp.z, gg.z, mm.z, vv.z, lr_bc1, rs_bc2, b1, b2, eps, gscale);
      adam_one(pp.w, gg.w, mm.w, vv.w, lr_bc1, rs_bc2, b1, b2, eps, gscale);
      ((float4*)p)[k] = pp;
      ((float4*)m)[k] = mm;
      ((float4*)v)[k] = vv;
      if (zg) ((float4*)g)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (plp) {
        plp[4 * k] = f2bf(pp.x);
        plp[4 * k + 1] = f2bf(pp.y);
        plp[4 * k + 2] = f2bf(pp.z);
        plp[4 * k + 3] = f2bf(pp.w);
      }
    }
    i += n4 * 4;  // tail elements
    for (int64_t k = i; k < n; k += stride) {
      const float r = adam_one(p[k], g[k], m[k], v[k], lr_bc1, rs_bc2, b1, b2, eps, gscale);
      if (zg) g[k] = 0.f;
      if (plp) plp[k] = f2bf(r);
    }
  } else {
    for (int64_t k = i; k < n; k += stride) {
      const float r = adam_one(p[k], g[k], m[k], v[k], lr_bc1, rs_bc2, b1, b2, eps, gscale);
      if (zg) g[k] = 0.f;
      if (plp) plp[k] = f2bf(r);
    }
  }
  if (flags & FHVAE_ADAM_ADVANCE) {
    __syncthreads();  // (every thread of this workgroup has its t)
    if (threadIdx.x == 0) {
      const unsigned grp = blockIdx.x & 63u, ngrp = gridDim.x < 64u ? gridDim.x : 64u;
      const unsigned members = (gridDim.x - grp + 63u) / 64u;  // workgroups with this group's low bits
      if (atomicAdd((unsigned*)(step + 32 * (1 + grp)), 1u) == members - 1) {
        step[32 * (1 + grp)] = 0;
        if (atomicAdd((unsigned*)(step + 1), 1u) == ngrp - 1) {
          step[1] = 0;
          step[0] = t_i;
        }
      }
    }
  }
}

}  // namespace fh

using namespace fh;

extern "C" int fhvae_abi_version(void) { return FHVAE_ABI_VERSION; }

extern "C" const char* fhvae_strerror(int code) {
  switch (code) {
    case FHVAE_OK: return "ok";
    case FHVAE_ERR_NULL: return "required pointer is NULL";
    case FHVAE_ERR_SHAPE: return "bad or inconsistent dimension";
    case FHVAE_ERR_DTYPE: return "unsupported dtype for this entry point";
    case FHVAE_ERR_ALIGN: return "pointer or leading dimension not aligned";
    case FHVAE_ERR_LIMIT: return "dimension exceeds kernel index range";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

extern "C" int fhvae_to_time_major(const float* x_btf, void* x_tbf, float* x_tbf_f32, int64_t B, int64_t T, int64_t F,
                                   int dtype, void* stream) {
  FH_CHECK_PTR(x_btf);
  FH_CHECK_PTR(x_tbf);
  FH_CHECK_POS(B);
  FH_CHECK_POS(T);
  FH_CHECK_POS(F);
  const int64_t n = B * T * F;
  if (dtype != FHVAE_F32 && dtype != FHVAE_BF16) return FHVAE_ERR_DTYPE;
  if (F % 4 == 0 && B <= INT32_MAX && T <= INT32_MAX && ((((uintptr_t)x_btf) | ((uintptr_t)x_tbf) | ((uintptr_t)x_tbf_f32)) & 15) == 0) {
    dim3 grid4((unsigned)fh_cdiv(n / 4, 256));
    if (dtype == FHVAE_F32)
      hipLaunchKernelGGL((to_time_major4_kernel<float>), grid4, dim3(256), 0, (hipStream_t)stream, x_btf, (float*)x_tbf, x_tbf_f32, (int)B,
                         (int)T, (int)(F / 4));
    else
      hipLaunchKernelGGL((to_time_major4_kernel<u16>), grid4, dim3(256), 0, (hipStream_t)stream, x_btf, (u16*)x_tbf, x_tbf_f32, (int)B, (int)T,
                         (int)(F / 4));
    return fh_launch_status();
  }
  dim3 grid((unsigned)fh_cdiv(n, 256));
  if (dtype == FHVAE_F32)
    hipLaunchKernelGGL((to_time_major_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, x_btf, (float*)x_tbf,
                       x_tbf_f32, B, T, F);
  else if (dtype == FHVAE_BF16)
    hipLaunchKernelGGL((to_time_major_kernel<u16>), grid, dim3(256), 0, (hipStream_t)stream, x_btf, (u16*)x_tbf, x_tbf_f32,
                       B, T, F);
  else
    return FHVAE_ERR_DTYPE;
  return fh_launch_status();
}

extern "C" int fhvae_cast_bf16(const float* src, void* dst, void* dst_t, int64_t R, int64_t C, void* stream) {
  FH_CHECK_PTR(src);
  if (!dst && !dst_t) return FHVAE_ERR_NULL;
  FH_CHECK_POS(R);
  FH_CHECK_POS(C);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)fh_cdiv(R * C, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (u16*)dst, (u16*)dst_t, R, C);
  return fh_launch_status();
}

extern "C" int fhvae_mu2_gather_fwd(const float* table, const int64_t* idx, int64_t idx_offset, float* mu2, int64_t B,
                                    int64_t S, int64_t D, int32_t* oob_flag, void* stream) {
  FH_CHECK_PTR(table);
  FH_CHECK_PTR(idx);
  FH_CHECK_PTR(mu2);
  FH_CHECK_POS(B);
  FH_CHECK_POS(S);
  FH_CHECK_POS(D);
  hipLaunchKernelGGL(gather_fwd_kernel, dim3((unsigned)fh_cdiv(B * D, 256)), dim3(256), 0, (hipStream_t)stream, table, idx,
                     idx_offset, mu2, B, S, D, oob_flag);
  return fh_launch_status();
}

extern "C" int fhvae_mu2_gather_bwd(const float* dmu2, const int64_t* idx, int64_t idx_offset, float* dtable, int64_t B,
                                    int64_t S, int64_t D, float scale, void* stream) {
  FH_CHECK_PTR(dmu2);
  FH_CHECK_PTR(idx);
  FH_CHECK_PTR(dtable);
  FH_CHECK_POS(B);
  FH_CHECK_POS(S);
  FH_CHECK_POS(D);
  hipLaunchKernelGGL(gather_bwd_kernel, dim3((unsigned)fh_cdiv(B * D, 256)), dim3(256), 0, (hipStream_t)stream, dmu2, idx,
                     idx_offset, dtable, B, S, D, scale);
  return fh_launch_status();
}

static int check_elbo(const fhvae_elbo_desc* d) {
  FH_CHECK_PTR(d);
  FH_CHECK_POS(d->B);
  FH_CHECK_POS(d->T);
  FH_CHECK_POS(d->F);
  FH_CHECK_POS(d->D1);
  FH_CHECK_POS(d->D2);
  FH_CHECK_I32(d->T * d->F);
  FH_CHECK_PTR(d->x);
  FH_CHECK_PTR(d->x_mu);
  FH_CHECK_PTR(d->x_lv);
  FH_CHECK_PTR(d->z1_mu);
  FH_CHECK_PTR(d->z1_lv);
  FH_CHECK_PTR(d->z2_mu);
  FH_CHECK_PTR(d->z2_lv);
  FH_CHECK_PTR(d->mu2);
  if (!d->num_segs && !(d->nsegs_scalar > 0)) return FHVAE_ERR_SHAPE;
  return FHVAE_OK;
}

extern "C" int fhvae_elbo_fwd(const fhvae_elbo_desc* d, void* stream) {
  int e = check_elbo(d);
  if (e) return e;
  FH_CHECK_PTR(d->lower_bound);
  FH_CHECK_PTR(d->log_px_z);
  FH_CHECK_PTR(d->neg_kld_z1);
  FH_CHECK_PTR(d->neg_kld_z2);
  FH_CHECK_PTR(d->log_pmu2);
  hipLaunchKernelGGL(elbo_fwd_kernel, dim3((unsigned)fh_cdiv(d->B, 4)), dim3(256), 0, (hipStream_t)stream, *d);
  return fh_launch_status();
}

extern "C" int fhvae_elbo_bwd(const fhvae_elbo_bwd_desc* d, void* stream) {
  FH_CHECK_PTR(d);
  int e = check_elbo(&d->f);
  if (e) return e;
  FH_CHECK_PTR(d->d_z1_mu);
  FH_CHECK_PTR(d->d_z1_lv);
  FH_CHECK_PTR(d->d_z2_mu);
  FH_CHECK_PTR(d->d_z2_lv);
  FH_CHECK_PTR(d->d_mu2);
  if (!d->reference_detach && (!d->d_x_mu || !d->d_x_lv)) return FHVAE_ERR_NULL;
  if (d->d_x_pair_lp) {  // the bf16 pair copy + column sums: vector layout with time-major rows only
    const fhvae_elbo_desc& f = d->f;
    FH_CHECK_PTR(d->d_x_colsum);
    if (d->reference_detach || !d->d_x_mu) return FHVAE_ERR_NULL;
    if (f.F % 4 || f.F > 256 || d->ld_pair < 2 * f.F || (d->ld_pair - 2 * f.F) % 8 || d->ld_pair % 8) return FHVAE_ERR_SHAPE;
    if ((f.x_sb % 4) || (f.x_st % 4) || (f.xo_sb % 4) || (f.xo_st % 4)) return FHVAE_ERR_ALIGN;
    if ((((uintptr_t)f.x | (uintptr_t)f.x_mu | (uintptr_t)f.x_lv | (uintptr_t)d->d_x_mu | (uintptr_t)d->d_x_lv | (uintptr_t)d->d_x_pair_lp) & 15))
      return FHVAE_ERR_ALIGN;
  }
  if (d->d_x_pair_lp)
    hipLaunchKernelGGL(elbo_bwd_pair_kernel, dim3((unsigned)fh_cdiv(d->f.B, 2)), dim3(256), 0, (hipStream_t)stream, *d);
  else
    hipLaunchKernelGGL(elbo_bwd_kernel, dim3((unsigned)fh_cdiv(d->f.B, 4)), dim3(256), 0, (hipStream_t)stream, *d);
  return fh_launch_status();
}

// disc_mfma.hip / disc_lp.hip: the matrix-core forms for large (B x S), D == 32 (declared in disc_mfma.h)
  // namespace fh

extern "C" int64_t fhvae_disc_lse_bwd_ws_bytes(int64_t B, int64_t S, int64_t D) {
  return (B > 0 && S > 0 && D > 0 && disc_mfma_supported(B, S, D)) ? disc_onepass_ws_bytes(B, S, D) : 0;
}

extern "C" int64_t fhvae_elbo_colsum_rows(int64_t B) { return B > 0 ? fh_cdiv(B, 2) : 0; }

extern "C" int64_t fhvae_disc_lse_ws_bytes(int64_t B, int64_t S) {
  if (B <= 0 || S <= 0) return 0;
  DiscPlan p = disc_plan(B, S);
  const int64_t a = (int64_t)p.nchunks * B * (int64_t)sizeof(float2), b = disc_mfma_ws_bytes(B, S);
  return a > b ? a : b;
}

#define DISC_DISPATCH(D_, CALL) \
  switch (D_) {                 \
    case 4: { constexpr int DD = 4; CALL; } break;   \
    case 8: { constexpr int DD = 8; CALL; } break;   \
    case 16: { constexpr int DD = 16; CALL; } break; \
    case 32: { constexpr int DD = 32; CALL; } break; \
    case 64: { constexpr int DD = 64; CALL; } break; \
    default: return FHVAE_ERR_SHAPE;                 \
  }

extern "C" int fhvae_disc_lse_fwd(const float* q, const float* table, const int64_t* idx, int64_t row0, float inv_two_var,
                                  float* row_max, float* row_sumexp, float* tgt_logit, float* ce_mean, float ce_scale, void* ws,
                                  int64_t B, int64_t S, int64_t D, int dtype, void* stream) {
  if (dtype != FHVAE_F32 && dtype != FHVAE_BF16) return FHVAE_ERR_DTYPE;
  FH_CHECK_PTR(q);
  FH_CHECK_PTR(table);
  FH_CHECK_PTR(idx);
  FH_CHECK_PTR(row_max);
  FH_CHECK_PTR(row_sumexp);
  FH_CHECK_PTR(tgt_logit);
  FH_CHECK_PTR(ws);
  FH_CHECK_POS(B);
  FH_CHECK_POS(S);
  FH_CHECK_I32(B);
  FH_CHECK_I32(S);
  hipStream_t st = (hipStream_t)stream;
  DiscPlan p = disc_plan(B, S);
  float2* part = (float2*)ws;
  int e, own_excluded = 0;
  if (disc_mfma_supported(B, S, D) && !getenv("FHVAE_DISC_VALU")) {
    e = disc_mfma_fwd(q, table, idx, row0, inv_two_var, part, &p.nchunks, B, S, D, dtype == FHVAE_BF16, st);
    own_excluded = 1;
  } else {
    dim3 grid((unsigned)p.btiles, (unsigned)p.nchunks);
    DISC_DISPATCH(D, hipLaunchKernelGGL((disc_fwd_kernel<DD>), grid, dim3(256), 0, st, q, table, inv_two_var, part, (int)B,
                                        (int)S, p.chunk));
    e = fh_launch_status();
  }
  if (e) return e;
  DISC_DISPATCH(D, hipLaunchKernelGGL((disc_combine_kernel<DD>), dim3((unsigned)fh_cdiv(B, 4)), dim3(256), 0, st, q, table, idx,
                                      row0, inv_two_var, part, p.nchunks, row_max, row_sumexp, tgt_logit, (int)B, (int)S,
                                      own_excluded));
  e = fh_launch_status();
  if (e) return e;
  if (ce_mean) {
    hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, st, row_max, row_sumexp, tgt_logit, ce_mean, (int)B, ce_scale);
    e = fh_launch_status();
  }
  return e;
}

extern "C" int fhvae_shard_pack(const float* q, const int64_t* idx, float* out, int64_t B, int64_t D, void* stream) {
  FH_CHECK_PTR(q);
  FH_CHECK_PTR(idx);
  FH_CHECK_PTR(out);
  FH_CHECK_POS(B);
  FH_CHECK_POS(D);
  hipLaunchKernelGGL(shard_pack_kernel, dim3((unsigned)fh_cdiv(B * (D + 1), 256)), dim3(256), 0, (hipStream_t)stream, q, idx, out, B, (int)D);
  return fh_launch_status();
}
extern "C" int fhvae_shard_unpack(const float* packed, float* q, int64_t* idx, int64_t N, int64_t D, void* stream) {
  FH_CHECK_PTR(packed);
  FH_CHECK_PTR(q);
  FH_CHECK_PTR(idx);
  FH_CHECK_POS(N);
  FH_CHECK_POS(D);
  hipLaunchKernelGGL(shard_unpack_kernel, dim3((unsigned)fh_cdiv(N * (D + 1), 256)), dim3(256), 0, (hipStream_t)stream, packed, q, idx, N, (int)D);
  return fh_launch_status();
}
extern "C" int fhvae_disc_merge_partials(const float* parts, float* row_max, float* row_sumexp, float* tgt_logit, int64_t W, int64_t N,
                                         void* stream) {
  FH_CHECK_PTR(parts);
  FH_CHECK_PTR(row_max);
  FH_CHECK_PTR(row_sumexp);
  FH_CHECK_PTR(tgt_logit);
  FH_CHECK_POS(W);
  FH_CHECK_POS(N);
  FH_CHECK_I32(W);
  hipLaunchKernelGGL(disc_merge_kernel, dim3((unsigned)fh_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, parts, row_max, row_sumexp,
                     tgt_logit, (int)W, N);
  return fh_launch_status();
}
extern "C" int fhvae_shard_bwd_pack(const float* dq_all, float dq_scale, const float* dmu2_local, int64_t own0, int64_t n_own, float* out,
                                    int64_t N, int64_t D, void* stream) {
  FH_CHECK_PTR(out);
  FH_CHECK_POS(N);
  FH_CHECK_POS(D);
  hipLaunchKernelGGL(shard_bwd_pack_kernel, dim3((unsigned)fh_cdiv(N * 2 * D, 256)), dim3(256), 0, (hipStream_t)stream, dq_all, dq_scale,
                     dmu2_local, own0, n_own, out, N, (int)D);
  return fh_launch_status();
}
extern "C" int fhvae_shard_bwd_unpack(const float* buf, int64_t own0, int64_t n_own, float* dq_local, float* dmu2_all, int64_t N, int64_t D,
                                      void* stream) {
  FH_CHECK_PTR(buf);
  FH_CHECK_POS(N);
  FH_CHECK_POS(D);
  hipLaunchKernelGGL(shard_bwd_unpack_kernel, dim3((unsigned)fh_cdiv(N * 2 * D, 256)), dim3(256), 0, (hipStream_t)stream, buf, own0, n_own,
                     dq_local, dmu2_all, N, (int)D);
  return fh_launch_status();
}

extern "C" int fhvae_disc_ce_mean(const float* row_max, const float* row_sumexp, const float* tgt_logit, float* ce_mean,
                                  float ce_scale, int64_t B, void* stream) {
  FH_CHECK_PTR(row_max);
  FH_CHECK_PTR(row_sumexp);
  FH_CHECK_PTR(tgt_logit);
  FH_CHECK_PTR(ce_mean);
  FH_CHECK_POS(B);
  FH_CHECK_I32(B);
  hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, row_max, row_sumexp, tgt_logit, ce_mean,
                     (int)B, ce_scale);
  return fh_launch_status();
}

extern "C" int fhvae_disc_lse_bwd(const float* q, const float* table, const int64_t* idx, int64_t row0, float inv_two_var,
                                  const float* row_max, const float* row_sumexp, const float* g_scale, float g_mul,
                                  float* dq, float* dtable, void* ws, int64_t ws_bytes, int64_t B, int64_t S, int64_t D, int dtype,
                                  void* stream) {
  if (dtype != FHVAE_F32 && dtype != FHVAE_BF16) return FHVAE_ERR_DTYPE;
  FH_CHECK_PTR(q);
  FH_CHECK_PTR(table);
  FH_CHECK_PTR(idx);
  FH_CHECK_PTR(row_max);
  FH_CHECK_PTR(row_sumexp);
  FH_CHECK_PTR(g_scale);
  FH_CHECK_POS(B);
  FH_CHECK_POS(S);
  FH_CHECK_I32(B);
  FH_CHECK_I32(S);
  hipStream_t st = (hipStream_t)stream;
  if (disc_mfma_supported(B, S, D) && !getenv("FHVAE_DISC_VALU")) {
    if (ws && (((uintptr_t)ws) & 15)) return FHVAE_ERR_ALIGN;
    int e = disc_mfma_bwd(q, table, idx, row0, inv_two_var, row_max, row_sumexp, g_scale, g_mul, dq, dtable, (float*)ws,
                          ws ? ws_bytes : 0, B, S, D, dtype == FHVAE_BF16, st);
    if (e) return e;
    if (dq || dtable) {
      DISC_DISPATCH(D, hipLaunchKernelGGL((disc_own_bwd_kernel<DD>), dim3((unsigned)fh_cdiv(B * (DD / 4), 256)), dim3(256), 0, st, q,
                                          table, idx, row0, inv_two_var, row_max, row_sumexp, g_scale, g_mul, dq, dtable, (int)B,
                                          (int)S));
      e = fh_launch_status();
    }
    return e;
  }
  if (dq) {
    hipError_t he = hipMemsetAsync(dq, 0, (size_t)(B * D) * sizeof(float), st);
    if (he != hipSuccess) return (int)he;
    DiscPlan p = disc_plan(B, S);
    dim3 grid((unsigned)p.btiles, (unsigned)p.nchunks);
    DISC_DISPATCH(D, hipLaunchKernelGGL((disc_bwd_dq_kernel<DD>), grid, dim3(256), 0, st, q, table, idx, row0, inv_two_var,
                                        row_max, row_sumexp, g_scale, g_mul, dq, (int)B, (int)S, p.chunk));
    int e = fh_launch_status();
    if (e) return e;
  }
  if (dtable) {
    DiscPlan p = disc_plan(S, B);  // roles swapped: threads = rows, chunks over queries
    dim3 grid((unsigned)p.btiles, (unsigned)p.nchunks);
    DISC_DISPATCH(D, hipLaunchKernelGGL((disc_bwd_dt_kernel<DD>), grid, dim3(256), 0, st, q, table, idx, row0, inv_two_var,
                                        row_max, row_sumexp, g_scale, g_mul, dtable, (int)B, (int)S, p.chunk));
    int e = fh_launch_status();
    if (e) return e;
  }
  return FHVAE_OK;
}

extern "C" int fhvae_adam_step(float* p, float* g, float* m, float* v, void* p_lp, int64_t n, float lr, float beta1,
                               float beta2, float eps, float grad_scale, int flags, int32_t* step_count, void* stream) {
  if (flags & ~(FHVAE_ADAM_ZERO_GRAD | FHVAE_ADAM_ADVANCE)) return FHVAE_ERR_SHAPE;
  FH_CHECK_PTR(p);
  FH_CHECK_PTR(g);
  FH_CHECK_PTR(m);
  FH_CHECK_PTR(v);
  FH_CHECK_PTR(step_count);
  FH_CHECK_POS(n);
  int64_t blocks = fh_cdiv(fh_cdiv(n, 4), 256);
  if (blocks > 8192) blocks = 8192;  // grid-stride beyond that
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (u16*)p_lp, n, lr,
                     beta1, beta2, eps, grad_scale, flags, step_count);
  return fh_launch_status();
}

extern "C" int fhvae_loss_fwd(const float* lower_bound, const float* log_qy, float alpha, float* loss, int64_t B, int32_t* nan_flag,
                              void* stream) {
  FH_CHECK_PTR(lower_bound);
  FH_CHECK_PTR(log_qy);
  FH_CHECK_PTR(loss);
  FH_CHECK_POS(B);
  FH_CHECK_I32(B);
  hipLaunchKernelGGL(loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lower_bound, log_qy, alpha, loss, (int)B, nan_flag);
  return fh_launch_status();
}

extern "C" int fhvae_loss_bwd(const float* g_loss, float alpha, float* d_lower_bound, float* d_log_qy, int64_t B, void* stream) {
  FH_CHECK_PTR(d_lower_bound);
  FH_CHECK_POS(B);
  hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)fh_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, g_loss, alpha, d_lower_bound,
                     d_log_qy, B);
  return fh_launch_status();
}
