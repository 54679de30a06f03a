// gemm.hip -- generic linear-layer kernels on the shared MFMA engine (gemm_core.h):
//   fhvae_linear_fwd / fhvae_linear_bwd / fhvae_gauss_head_reparam_fwd / fhvae_gauss_reparam_bwd.
// Reference ops replaced: nn.Linear (+ReLU) at simple_fhvae.py:127-134 and the Gaussian layer at
// simple_fhvae.py:193-216.
#include "gemm_launch.h"
#include "proj.h"
#include "wgrad.h"

#include <algorithm>
#include <cstdlib>

namespace fh {

// ---------------------------------------------------------------------------------------------
// generic GEMM kernel: C = epi(sum_seg A.B^T)
// ---------------------------------------------------------------------------------------------
// four consecutive columns of one row (col % 4 == 0): 16-byte stores when the output allows it
__device__ __forceinline__ void gemm_store4(const GemmParams& p, bool vec, int row, int col, f32x4 v) {
  if (p.relu) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
  }
  if (vec && col + 3 < p.N) {
    if (p.C) {
      float* c = (p.C2 && row >= p.c_split) ? p.C2 + (int64_t)(row - p.c_split) * p.ldc + col : p.C + (int64_t)row * p.ldc + col;
      if (p.mode == 0) {
        *(f32x4*)c = v;
      } else if (p.mode == 1) {
        *(f32x4*)c = *(const f32x4*)c + v;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(c + i, v[i]);
      }
    }
    if (p.Clp)
      *(uint2*)(p.Clp + (int64_t)row * p.ldclp + col) =
          uint2{(uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16), (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16)};
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (col + i >= p.N) break;
    float* c = p.C ? ((p.C2 && row >= p.c_split) ? p.C2 + (int64_t)(row - p.c_split) * p.ldc + col + i : p.C + (int64_t)row * p.ldc + col + i) : nullptr;
    if (c) {
      if (p.mode == 0)
        *c = v[i];
      else if (p.mode == 1)
        *c += v[i];
      else
        atomicAdd(c, v[i]);
    }
    if (p.Clp) p.Clp[(int64_t)row * p.ldclp + col + i] = f2bf(v[i]);
  }
}

__device__ __forceinline__ void gemm_store(const GemmParams& p, int row, int col, float v) {
  if (p.relu) v = fmaxf(v, 0.f);
  if (p.C) {
    float* c = (p.C2 && row >= p.c_split) ? p.C2 + (int64_t)(row - p.c_split) * p.ldc + col : p.C + (int64_t)row * p.ldc + col;
    if (p.mode == 0)
      *c = v;
    else if (p.mode == 1)
      *c += v;
    else
      atomicAdd(c, v);
  }
  if (p.Clp) p.Clp[(int64_t)row * p.ldclp + col] = f2bf(v);
}

// one output tile (bx, by) of problem p, K slice bz of p.splitk
// SWAP (outputs written once, no split-K atomics): MFMA roles swapped so that a lane holds 4 consecutive columns of a row and
// row-major f32 / bf16 outputs leave as 16- / 8-byte stores (the unswapped layout gives 4-byte stores; its split-K atomics,
// 64 contiguous bytes per 16 lanes, stay as they are: swapped they scatter and ran 25 % slower)
template <typename T, int BM, int BN, int WM, int WN, int CH, bool AKC, bool BKC, bool DMA, bool SWAP>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, int bx, int by, int bz, char* smem) {
  using TL = Tile<T, BM, BN, WM, WN, CH>;
  constexpr int TM = TL::TM, TN = TL::TN;
  constexpr bool kDma = DMA && AKC && BKC;
  constexpr int NBUF = CH >= 32 ? 2 : 1;
  const int m0 = by * BM, n0 = bx * BN;
  const int nkb = num_kblocks<T, CH>(p.seg);
  // split-K: contiguous ranges of panels per z-slice
  const int per = (nkb + p.splitk - 1) / p.splitk;
  const int it0 = bz * per;
  const int it1 = min(nkb, it0 + per);
  if (it0 >= it1 && bz > 0) return;

  f32x4 acc[TM][TN];
  zero_acc(acc);
  RowIdent arm{p.M}, brm{p.N};
  bool dma = false;
  if constexpr (kDma)
    dma = p.splitk == 1 && m0 + BM <= p.M && n0 + BN <= p.N && seg_glds_ok<T>(p.seg[0], TL::BK) && seg_glds_ok<T>(p.seg[1], TL::BK);
  if (dma) {
    if constexpr (kDma) mainloop_glds<T, BM, BN, WM, WN, CH, NBUF, SWAP>(acc, p.seg, m0, n0, arm, brm, smem);
  } else {
    mainloop<T, BM, BN, WM, WN, CH, AKC, BKC, SWAP>(acc, p.seg, m0, p.M, n0, p.N, arm, brm, it0, it1, smem);
  }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const bool first = bz == 0;
  if constexpr (!SWAP) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int col = n0 + wn * (TN * 16) + tn * 16 + (lane & 15);
        if (col >= p.N) continue;
        float add = 0.f;
        if (first) {
          if (p.bias) add += p.bias[col];
          if (p.bias2) add += p.bias2[col];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm * (TM * 16) + tm * 16 + (lane >> 4) * 4 + r;
          if (row < p.M) gemm_store(p, row, col, acc[tm][tn][r] + add);
        }
      }
    return;
  }
  // swapped MFMA roles: acc[tm][tn][r] = C[m = tm*16 + (lane & 15)][n = tn*16 + (lane >> 4)*4 + r]
  const bool vec = (p.ldc & 3) == 0 && (p.N & 3) == 0 && (!p.C || ((uintptr_t)p.C & 15) == 0) && (!p.C2 || ((uintptr_t)p.C2 & 15) == 0) &&
                   (!p.Clp || ((p.ldclp & 3) == 0 && ((uintptr_t)p.Clp & 7) == 0));
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int row = m0 + wm * (TM * 16) + tm * 16 + (lane & 15);
    if (row >= p.M) continue;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = n0 + wn * (TN * 16) + tn * 16 + (lane >> 4) * 4;
      if (col >= p.N) continue;
      f32x4 v = acc[tm][tn];
      if (first) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (col + r >= p.N) break;
          float add = 0.f;
          if (p.bias) add += p.bias[col + r];
          if (p.bias2) add += p.bias2[col + r];
          v[r] += add;
        }
      }
      gemm_store4(p, vec, row, col, v);
    }
  }
}

template <typename T, int BM, int BN, int WM, int WN, int CH, bool AKC, bool BKC, bool DMA = false, bool SWAP = false>
__global__ __launch_bounds__(kThreads) void gemm_kernel(GemmParams p) {
  using TL = Tile<T, BM, BN, WM, WN, CH>;
  // LDS-DMA main loop for interior KC/KC tiles without split-K.  A separate instantiation (it needs 128 KB of LDS for
  // its two buffers): launched only for small grids, where one workgroup per CU is all there is and the latency of a
  // panel is the whole cost (256x1024x512 bf16: 8.1 -> 7.0 us); larger grids prefer 2+ workgroups per CU.
  constexpr bool kDma = DMA && AKC && BKC;
  constexpr int NBUF = CH >= 32 ? 2 : 1;
  using GT = GldsTile<T, BM, BN, WM, WN, CH, NBUF>;
  __shared__ __attribute__((aligned(16))) char smem[(kDma && GT::SMEM > TL::SMEM) ? GT::SMEM : TL::SMEM];
  // XCD-aware tile order (guide T1, bijective form): workgroups are dealt round-robin over the 8 XCDs by linear id;
  // remap so that each XCD owns a CONTIGUOUS range of logical tiles (x fastest, then y, then the K slice): tiles that
  // share an operand panel then share one private L2 (a split-K weight gradient: one K slice per XCD instead of
  // every slice fetched by 4-8 XCDs; performance only, any placement is correct).
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const int gx = gridDim.x, gy = gridDim.y, nb = gx * gy * (int)gridDim.z;
    if (nb >= 16) {
      const int lin = bx + gx * (by + gy * bz);
      const int q = nb >> 3, r = nb & 7, xcd = lin & 7;
      const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
      bx = logical % gx;
      by = (logical / gx) % gy;
      bz = logical / (gx * gy);
    }
  }
  gemm_tile<T, BM, BN, WM, WN, CH, AKC, BKC, DMA, SWAP>(p, bx, by, bz, smem);
}

// Several independent problems of one shape class in ONE launch (blockIdx.z selects problem and K slice): the weight
// gradients of a net, the two linear layers of a Gaussian head.  At small batches each of them is a latency-bound launch
// of a few workgroups; together they fill the chip once.
template <typename T, int BM, int BN, int WM, int WN, int CH, bool AKC, bool BKC, bool SWAP = false>
__global__ __launch_bounds__(kThreads) void gemm_group_kernel(GemmGroup g) {
  using TL = Tile<T, BM, BN, WM, WN, CH>;
  __shared__ __attribute__((aligned(16))) char smem[TL::SMEM];
  int i = 0, bz = blockIdx.z;
  while (i + 1 < g.n && bz >= g.zbase[i + 1]) ++i;
  bz -= g.zbase[i];
  const GemmParams& p = g.p[i];
  if ((int)blockIdx.y * BM >= p.M || (int)blockIdx.x * BN >= p.N) return;
  gemm_tile<T, BM, BN, WM, WN, CH, AKC, BKC, false, SWAP>(p, blockIdx.x, blockIdx.y, bz, smem);
}

// Scalar fallback for shapes that break the 16-byte staging preconditions (odd K / leading dimension /
// unaligned views: tiny test shapes and the F=6 toy models).  One thread per output element.
template <typename T>
__global__ void gemm_slow_kernel(GemmParams p) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)p.M * p.N) return;
  const int row = (int)(i / p.N), col = (int)(i % p.N);
  float acc = 0.f;
  for (int si = 0; si < 2; ++si) {
    const Seg& s = p.seg[si];
    const T* A = (const T*)s.A;
    const T* B = (const T*)s.B;
    const int64_t ar = s.a_rmod > 0 ? row % s.a_rmod : row;
    for (int k = 0; k < s.K; ++k) {
      const T a = s.a_kc ? A[ar * s.lda + k] : A[(int64_t)k * s.lda + ar];
      const T b = s.b_kc ? B[(int64_t)col * s.ldb + k] : B[(int64_t)k * s.ldb + col];
      if constexpr (sizeof(T) == 4)
        acc = fmaf(a, b, acc);
      else
        acc = fmaf(bf2f(a), bf2f(b), acc);
    }
  }
  if (p.bias) acc += p.bias[col];
  if (p.bias2) acc += p.bias2[col];
  gemm_store(p, row, col, acc);
}

template <typename T, int BM, int BN, int CH>
static void launch_fast(const GemmParams& p, dim3 grid, hipStream_t st) {
  const int akc = p.seg[0].K > 0 ? p.seg[0].a_kc : p.seg[1].a_kc;
  const int bkc = p.seg[0].K > 0 ? p.seg[0].b_kc : p.seg[1].b_kc;
  if (akc && bkc) {
    if (CH == 32 && BM == 64 && p.splitk == 1 && (int64_t)grid.x * grid.y <= 256)
      hipLaunchKernelGGL((gemm_kernel<T, BM, BN, 2, 2, CH, true, true, CH == 32 && BM == 64, true>), grid, dim3(kThreads), 0, st, p);
    else if (p.splitk == 1 && p.mode != 2)
      hipLaunchKernelGGL((gemm_kernel<T, BM, BN, 2, 2, CH, true, true, false, true>), grid, dim3(kThreads), 0, st, p);
    else
      hipLaunchKernelGGL((gemm_kernel<T, BM, BN, 2, 2, CH, true, true>), grid, dim3(kThreads), 0, st, p);
  }
  else if (!akc && !bkc)
    hipLaunchKernelGGL((gemm_kernel<T, BM, BN, 2, 2, CH, false, false>), grid, dim3(kThreads), 0, st, p);
  else if constexpr (sizeof(T) == 4) {
    if (akc)
      hipLaunchKernelGGL((gemm_kernel<T, BM, BN, 2, 2, CH, true, false>), grid, dim3(kThreads), 0, st, p);
    else
      hipLaunchKernelGGL((gemm_kernel<T, BM, BN, 2, 2, CH, false, true>), grid, dim3(kThreads), 0, st, p);
  }
}

// splitk == 0 on entry means "choose": only legal with mode 1 (accumulate), switched to atomics when split
static int auto_splitk(int64_t tiles, int64_t panels) {
  if (tiles >= 192 || panels < 4) return 1;
  int64_t s = fh_cdiv(512, tiles);
  if (s > panels / 2) s = panels / 2;
  if (s < 1) s = 1;
  if (s > 128) s = 128;
  return (int)s;
}

int launch_gemm(const GemmParams& p_in, int dtype, hipStream_t st) {
  GemmParams p = p_in;
  if (p.M <= 0 || p.N <= 0) return FHVAE_ERR_SHAPE;
  if (dtype != FHVAE_F32 && dtype != FHVAE_BF16) return FHVAE_ERR_DTYPE;
  const bool want_auto = p.splitk == 0;
  if (want_auto && p.mode != 1) return FHVAE_ERR_SHAPE;
  if (p.splitk < 1) p.splitk = 1;
  // both segments must share the operand orientation (true for every caller)
  if (p.seg[0].K > 0 && p.seg[1].K > 0 && (p.seg[0].a_kc != p.seg[1].a_kc || p.seg[0].b_kc != p.seg[1].b_kc))
    return FHVAE_ERR_SHAPE;
  const bool bf = dtype == FHVAE_BF16;
  bool fast = bf ? (seg_fast_ok<u16>(p.seg[0], p.M, p.N) && seg_fast_ok<u16>(p.seg[1], p.M, p.N))
                 : (seg_fast_ok<float>(p.seg[0], p.M, p.N) && seg_fast_ok<float>(p.seg[1], p.M, p.N));
  if (bf)
    for (int s = 0; s < 2; ++s)
      if (p.seg[s].K > 0 && p.seg[s].a_kc != p.seg[s].b_kc) fast = false;  // bf16 engine: KC/KC or KM/KM only
  if (!fast) {
    p.splitk = 1;  // the fallback does not split: same semantics with one slice
    const int64_t n = (int64_t)p.M * p.N;
    if (bf)
      hipLaunchKernelGGL(gemm_slow_kernel<u16>, dim3((unsigned)fh_cdiv(n, 256)), dim3(256), 0, st, p);
    else
      hipLaunchKernelGGL(gemm_slow_kernel<float>, dim3((unsigned)fh_cdiv(n, 256)), dim3(256), 0, st, p);
    return fh_launch_status();
  }
  const int ktot = p.seg[0].K + p.seg[1].K;
  const int kmax = p.seg[0].K > p.seg[1].K ? p.seg[0].K : p.seg[1].K;
  const int epc = bf ? 8 : 4;
  // 128x128 tiles (half the L2->LDS traffic per FLOP) once the problem offers enough of them
  // (measured: with fewer than ~2 workgroups per CU the per-CU load bandwidth, not the aggregate L2 traffic,
  // is the limit, and 64x64 tiles on more CUs win: 2048x1024x512 bf16 takes 11 us with 64^2, 21 us with 128^2)
  // long-K bf16 weight gradients (KM/KM, K = T*B >= 16384): 128x64 tiles with 128-k panels (a third less operand traffic per
  // FLOP than 64x64; 56 KB of LDS = two workgroups per CU) and exactly as many K slices as fill the chip twice over
  // (512 workgroups).  1024x256 over K = 40960: 52 -> 43 us; other slice counts were slower (256: -6 %, 768: -4 % per step).
  if (bf && want_auto && ktot >= 16384 && p.M >= 128 && p.seg[1].K == 0 && !p.seg[0].a_kc && !p.seg[0].b_kc) {
    const int64_t tiles = fh_cdiv(p.M, 128) * fh_cdiv(p.N, 64);
    int64_t sk = fh_cdiv(512, tiles);
    const int64_t panels = fh_cdiv(ktot, 128);
    if (sk > panels / 2) sk = panels / 2;
    if (sk < 1) sk = 1;
    p.splitk = (int)sk;
    if (p.splitk > 1) p.mode = 2;
    dim3 grid((unsigned)fh_cdiv(p.N, 64), (unsigned)fh_cdiv(p.M, 128), (unsigned)p.splitk);
    hipLaunchKernelGGL((gemm_kernel<u16, 128, 64, 4, 1, 16, false, false>), grid, dim3(kThreads), 0, st, p);
    return fh_launch_status();
  }
  const bool big = kmax > 16 * epc && fh_cdiv(p.M, 128) * fh_cdiv(p.N, 128) >= 512;
  const int tb = big ? 128 : 64;
  const int ch = big ? 16 : (kmax <= 16 * epc ? 8 : 32);
  if (want_auto) {
    p.splitk = auto_splitk(fh_cdiv(p.M, tb) * fh_cdiv(p.N, tb), fh_cdiv(ktot, ch * epc));
    if (p.splitk > 1) p.mode = 2;
  }
  dim3 grid((unsigned)fh_cdiv(p.N, tb), (unsigned)fh_cdiv(p.M, tb), (unsigned)p.splitk);
  if (!bf) {
    if (big)
      launch_fast<float, 128, 128, 16>(p, grid, st);
    else if (ch == 8)
      launch_fast<float, 64, 64, 8>(p, grid, st);
    else
      launch_fast<float, 64, 64, 32>(p, grid, st);
  } else {
    if (big)
      launch_fast<u16, 128, 128, 16>(p, grid, st);
    else if (ch == 8)
      launch_fast<u16, 64, 64, 8>(p, grid, st);
    else
      launch_fast<u16, 64, 64, 32>(p, grid, st);
  }
  return fh_launch_status();
}

// n <= kMaxGroup problems in one launch when they share dtype, orientation and the 64x64 / 512-byte-panel configuration
// and take the branch-free staging path; otherwise (and for n == 1) one launch each.  splitk == 0 (auto) is resolved per
// problem against the group's total tile count.
int launch_gemm_group(const GemmParams* ps, int n, int dtype, hipStream_t st) {
  if (n <= 0) return FHVAE_OK;
  const bool bf = dtype == FHVAE_BF16;
  const int epc = bf ? 8 : 4;
  bool ok = n > 1 && n <= kMaxGroup;
  int akc = -1, bkc = -1;
  int64_t tiles = 0;
  for (int i = 0; i < n && ok; ++i) {
    const GemmParams& p = ps[i];
    if (p.M <= 0 || p.N <= 0) return FHVAE_ERR_SHAPE;
    const bool fast = bf ? (seg_fast_ok<u16>(p.seg[0], p.M, p.N) && seg_fast_ok<u16>(p.seg[1], p.M, p.N))
                         : (seg_fast_ok<float>(p.seg[0], p.M, p.N) && seg_fast_ok<float>(p.seg[1], p.M, p.N));
    const int kmax = p.seg[0].K > p.seg[1].K ? p.seg[0].K : p.seg[1].K;
    const int a = p.seg[0].K > 0 ? p.seg[0].a_kc : p.seg[1].a_kc, b = p.seg[0].K > 0 ? p.seg[0].b_kc : p.seg[1].b_kc;
    if (p.seg[0].K > 0 && p.seg[1].K > 0 && (p.seg[0].a_kc != p.seg[1].a_kc || p.seg[0].b_kc != p.seg[1].b_kc)) ok = false;
    if (!fast || kmax <= 16 * epc || (p.splitk == 0 && p.mode != 1)) ok = false;
    if (p.seg[0].K + p.seg[1].K > 16384) ok = false;  // long contractions are bandwidth-bound: separate launches with the
                                                       // XCD-aware tile order measured faster (B = 2048: 616k vs 604k segments/s)
    if (bf && a != b) ok = false;
    if (akc < 0) akc = a, bkc = b;
    if (a != akc || b != bkc) ok = false;
    tiles += fh_cdiv(p.M, 64) * fh_cdiv(p.N, 64);
  }
  if (!ok) {
    for (int i = 0; i < n; ++i) {
      int e = launch_gemm(ps[i], dtype, st);
      if (e) return e;
    }
    return FHVAE_OK;
  }
  GemmGroup g = {};
  g.n = n;
  unsigned gx = 1, gy = 1;
  for (int i = 0; i < n; ++i) {
    GemmParams p = ps[i];
    if (p.splitk == 0) {
      // enough K slices that the group as a whole offers ~3 workgroups per CU (long contractions are bandwidth-bound:
      // they want the occupancy; short ones are latency-bound: they want the parallelism)
      const int64_t panels = fh_cdiv(p.seg[0].K + p.seg[1].K, 32 * epc);
      int64_t sk = panels < 4 ? 1 : fh_cdiv(768, tiles);
      if (sk > panels / 2) sk = panels / 2;
      if (sk < 1) sk = 1;
      if (sk > 128) sk = 128;
      p.splitk = (int)sk;
      if (p.splitk > 1) p.mode = 2;
    }
    if (p.splitk < 1) p.splitk = 1;
    g.p[i] = p;
    g.zbase[i + 1] = g.zbase[i] + p.splitk;
    gx = gx > (unsigned)fh_cdiv(p.N, 64) ? gx : (unsigned)fh_cdiv(p.N, 64);
    gy = gy > (unsigned)fh_cdiv(p.M, 64) ? gy : (unsigned)fh_cdiv(p.M, 64);
  }
  dim3 grid(gx, gy, (unsigned)g.zbase[n]);
  bool once = true;  // every output written once (no split-K atomics): the 16-byte-store epilogue
  for (int i = 0; i < n; ++i) once = once && g.p[i].splitk == 1 && g.p[i].mode != 2;
  if (bf) {
    if (akc && once)
      hipLaunchKernelGGL((gemm_group_kernel<u16, 64, 64, 2, 2, 32, true, true, true>), grid, dim3(kThreads), 0, st, g);
    else if (akc)
      hipLaunchKernelGGL((gemm_group_kernel<u16, 64, 64, 2, 2, 32, true, true>), grid, dim3(kThreads), 0, st, g);
    else
      hipLaunchKernelGGL((gemm_group_kernel<u16, 64, 64, 2, 2, 32, false, false>), grid, dim3(kThreads), 0, st, g);
  } else if (akc && bkc && once)
    hipLaunchKernelGGL((gemm_group_kernel<float, 64, 64, 2, 2, 32, true, true, true>), grid, dim3(kThreads), 0, st, g);
  else if (akc && bkc)
    hipLaunchKernelGGL((gemm_group_kernel<float, 64, 64, 2, 2, 32, true, true>), grid, dim3(kThreads), 0, st, g);
  else if (!akc && !bkc)
    hipLaunchKernelGGL((gemm_group_kernel<float, 64, 64, 2, 2, 32, false, false>), grid, dim3(kThreads), 0, st, g);
  else if (akc)
    hipLaunchKernelGGL((gemm_group_kernel<float, 64, 64, 2, 2, 32, true, false>), grid, dim3(kThreads), 0, st, g);
  else
    hipLaunchKernelGGL((gemm_group_kernel<float, 64, 64, 2, 2, 32, false, true>), grid, dim3(kThreads), 0, st, g);
  return fh_launch_status();
}

// kept for callers that want an explicit split (none at present)
int pick_splitk(int64_t M, int64_t N, int64_t K) {
  return auto_splitk(fh_cdiv(M, 64) * fh_cdiv(N, 64), fh_cdiv(K, 128));
}

// ---------------------------------------------------------------------------------------------
// small elementwise / reduction helpers
// ---------------------------------------------------------------------------------------------
__global__ void relu_mask_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ y, int64_t ldy,
                                 float* __restrict__ out, int64_t M, int64_t N) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  int64_t m = i / N, n = i % N;
  out[i] = y[m * ldy + n] > 0.f ? dy[m * lddy + n] : 0.f;
}

// db[n] += sum_m g[m][n].  Thread = one 16-byte column group (4 f32 / 8 bf16) x one of 8 row lanes; a block
// covers 32 column groups; rows are strided over gridDim.y; 8 row lanes reduce through LDS, then atomics.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ g, int64_t ldg, float* __restrict__ db,
                                                     float* __restrict__ db2, int64_t M, int64_t N, int64_t split) {
  constexpr int E = 16 / (int)sizeof(T);
  __shared__ float red[8][32 * E + 1];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int64_t n0 = ((int64_t)blockIdx.x * 32 + cg) * E;
  float s[E];
#pragma unroll
  for (int e = 0; e < E; ++e) s[e] = 0.f;
  const bool vec = (ldg % E == 0) && ((((uintptr_t)g) & 15) == 0) && n0 + E <= N;
  // (unrolled: the row loads are independent; one at a time they were a chain of HBM round trips -- 13 MB took 20.8 us)
#pragma unroll 4
  for (int64_t m = (int64_t)blockIdx.y * 8 + rl; m < M; m += (int64_t)gridDim.y * 8) {
    if (vec) {
      const uint4 u = *(const uint4*)(g + m * ldg + n0);
      if constexpr (sizeof(T) == 4) {
        s[0] += __uint_as_float(u.x);
        s[1] += __uint_as_float(u.y);
        s[2] += __uint_as_float(u.z);
        s[3] += __uint_as_float(u.w);
      } else {
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[2 * e] += __uint_as_float(w[e] << 16);
          s[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (n0 + e < N) {
          if constexpr (sizeof(T) == 4)
            s[e] += g[m * ldg + n0 + e];
          else
            s[e] += bf2f(g[m * ldg + n0 + e]);
        }
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) red[rl][cg * E + e] = s[e];
  __syncthreads();
  for (int c = threadIdx.x; c < 32 * E; c += 256) {
    const int64_t n = (int64_t)blockIdx.x * 32 * E + c;
    if (n >= N) continue;
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += red[r][c];
    if (split > 0) {  // two vectors side by side: columns < split -> db, the others -> db2
      float* d = n < split ? (db ? db + n : nullptr) : (db2 ? db2 + (n - split) : nullptr);
      if (d) atomicAdd(d, t);
    } else {
      if (db) atomicAdd(db + n, t);
      if (db2) atomicAdd(db2 + n, t);
    }
  }
}

int launch_colsum(const void* g, int dtype, int64_t ldg, float* db, float* db2, int64_t M, int64_t N, hipStream_t st, int64_t split) {
  if (!db && !db2) return FHVAE_OK;
  const int E = dtype == FHVAE_F32 ? 4 : 8;
  int64_t gx = fh_cdiv(N, 32 * E);
  int64_t gy = fh_cdiv(M, 8 * 8);
  const int64_t want = fh_cdiv(1024, gx);
  if (gy > want) gy = want;
  if (gy < 1) gy = 1;
  dim3 grid((unsigned)gx, (unsigned)gy);
  if (dtype == FHVAE_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)g, ldg, db, db2, M, N, split);
  else
    hipLaunchKernelGGL(colsum_kernel<u16>, grid, dim3(256), 0, st, (const u16*)g, ldg, db, db2, M, N, split);
  return fh_launch_status();
}

__global__ void reparam_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                   const float* __restrict__ eps, float* __restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = mu[i] + eps[i] * expf(0.5f * lv[i]);  // simple_fhvae.py:214-216
}

__global__ void reparam_bwd_kernel(const float* __restrict__ d_mu, const float* __restrict__ d_lv,
                                   const float* __restrict__ d_s, const float* __restrict__ eps,
                                   const float* __restrict__ lv, float* __restrict__ g_mu, float* __restrict__ g_lv,
                                   int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float ds = d_s ? d_s[i] : 0.f;
  g_mu[i] = (d_mu ? d_mu[i] : 0.f) + ds;
  float e = (d_s && eps) ? ds * eps[i] * 0.5f * expf(0.5f * lv[i]) : 0.f;
  g_lv[i] = (d_lv ? d_lv[i] : 0.f) + e;
}

// g[M, 2D] = [g_mu | g_lv]: the two linear layers' upstream gradients side by side, so that one contraction / one
// column sum serves both
template <typename TO>
__global__ void reparam_bwd_cat_kernel(const float* __restrict__ d_mu, const float* __restrict__ d_lv,
                                       const float* __restrict__ d_s, const float* __restrict__ eps,
                                       const float* __restrict__ lv, TO* __restrict__ g, int64_t n, int D) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t m = i / D, c = i - m * D;
  float ds = d_s ? d_s[i] : 0.f;
  float e = (d_s && eps) ? ds * eps[i] * 0.5f * expf(0.5f * lv[i]) : 0.f;
  const float gm = (d_mu ? d_mu[i] : 0.f) + ds, gl = (d_lv ? d_lv[i] : 0.f) + e;
  if constexpr (sizeof(TO) == 4) {
    g[m * 2 * D + c] = gm;
    g[m * 2 * D + D + c] = gl;
  } else {
    g[m * 2 * D + c] = f2bf(gm);
    g[m * 2 * D + D + c] = f2bf(gl);
  }
}

// the stacked bf16 operands of a Gaussian head: wl [2D,K] and wt [K,ldt] = [w_mu^T | w_lv^T | 0]
__global__ void head_pair_weights_kernel(const float* __restrict__ w_mu, const float* __restrict__ w_lv, u16* __restrict__ wl,
                                         u16* __restrict__ wt, int64_t ldt, int D, int K) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ldt * K) return;
  const int r = (int)(i / K), k = (int)(i - (int64_t)r * K);  // r: stacked output row (consecutive threads walk k: coalesced reads)
  u16 v = 0;
  if (r < 2 * D) {
    v = f2bf(r < D ? w_mu[(int64_t)r * K + k] : w_lv[(int64_t)(r - D) * K + k]);
    if (wl) wl[(int64_t)r * K + k] = v;
  }
  if (wt) wt[(int64_t)k * ldt + r] = v;
}

__global__ void reparam_pair_fwd_kernel(const float* __restrict__ out, int64_t ldo, const float* __restrict__ eps,
                                        float* __restrict__ smp, float* __restrict__ mu_c, float* __restrict__ lv_c, int64_t n, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t m = i / D, c = i - m * D;
  const float mu = out[m * ldo + c], lv = out[m * ldo + D + c];
  smp[i] = mu + eps[i] * expf(0.5f * lv);  // simple_fhvae.py:214-216
  if (mu_c) mu_c[i] = mu;
  if (lv_c) lv_c[i] = lv;
}

// thread = 8 consecutive columns of a row of g (one 16-byte store); columns [2D, ldg) are zero.  DB: the workgroup's rows
// (256 / (ldg / 8) whole rows) are also summed per column through LDS and added to the two bias gradients, one atomic per column
// and workgroup -- the column-sum launch that used to follow (colsum_kernel over g) is gone
template <bool DB>
__global__ __launch_bounds__(256) void reparam_bwd_pair_kernel(const float* __restrict__ d_mu, const float* __restrict__ d_lv,
                                                               const float* __restrict__ d_s, int64_t ld_s, const float* __restrict__ eps,
                                                               const float* __restrict__ lv, int64_t ld_lv, u16* __restrict__ g,
                                                               int64_t ldg, int64_t M, int D, float* __restrict__ db_mu,
                                                               float* __restrict__ db_lv) {
  __shared__ float tile[DB ? 2048 : 1];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per = (int)(ldg / 8);
  const bool live = i < M * per;
  if (!DB && !live) return;
  const int64_t m = i / per;
  const int c0 = (int)(i - m * per) * 8;
  union {
    u16 h[8];
    uint4 v;
  } o;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c0 + k;
    float v = 0.f;
    if (live && c < D) {
      v = (d_mu ? d_mu[m * D + c] : 0.f) + (d_s ? d_s[m * ld_s + c] : 0.f);
    } else if (live && c < 2 * D) {
      const int64_t j = m * D + (c - D);
      v = (d_lv ? d_lv[j] : 0.f) + ((d_s && eps) ? d_s[m * ld_s + (c - D)] * eps[j] * 0.5f * expf(0.5f * lv[m * ld_lv + (c - D)]) : 0.f);
    }
    o.h[k] = f2bf(v);
    if (DB) tile[threadIdx.x * 8 + k] = bf2f(o.h[k]);  // (the rounded value: what the weight-gradient contraction multiplies too)
  }
  if (live) *(uint4*)(g + m * ldg + c0) = o.v;
  if (DB) {
    __syncthreads();
    const int rows = 256 / per;  // whole rows of this workgroup: flat index row * ldg + column
    for (int c = threadIdx.x; c < 2 * D; c += 256) {
      float sum = 0.f;
      for (int r = 0; r < rows; ++r) sum += tile[r * (int)ldg + c];
      float* dst = c < D ? (db_mu ? db_mu + c : nullptr) : (db_lv ? db_lv + (c - D) : nullptr);
      if (dst) atomicAdd(dst, sum);
    }
  }
}

// a[0..D) += column sums of src[rows][2D] (columns [0, D)), b likewise (columns [D, 2D)).  grid = (64-column groups, row
// chunks): wave w of chunk y takes rows (4y + w) + 4 gridDim.y k, eight independent loads in flight; one atomic per column and
// workgroup.
__global__ __launch_bounds__(256) void add_split_kernel(const float* __restrict__ src, int rows, float* __restrict__ a,
                                                        float* __restrict__ b, int D) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const int step = 4 * gridDim.y;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < 2 * D) {
    for (int r = blockIdx.y * 4 + wave; r < rows; r += 8 * step) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (r + k * step < rows) acc[k] += src[(int64_t)(r + k * step) * 2 * D + i];
    }
  }
  part[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (wave == 0 && i < 2 * D) {
    const float v = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    if (i < D) {
      if (a) atomicAdd(a + i, v);
    } else if (b) {
      atomicAdd(b + (i - D), v);
    }
  }
}

}  // namespace fh

using namespace fh;

extern "C" int fhvae_linear_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* b, float* y,
                                int64_t ldy, void* y_lp, int64_t M, int64_t K, int64_t N, int relu, int dtype,
                                void* stream) {
  FH_CHECK_PTR(x);
  FH_CHECK_PTR(w);
  if (!y && !y_lp) return FHVAE_ERR_NULL;
  FH_CHECK_POS(M);
  FH_CHECK_POS(K);
  FH_CHECK_POS(N);
  FH_CHECK_I32(M);
  FH_CHECK_I32(K);
  FH_CHECK_I32(N);
  if (ldx < K || ldw < K || (y && ldy < N)) return FHVAE_ERR_SHAPE;
  GemmParams p = {};
  p.seg[0] = Seg{x, ldx, 1, w, ldw, 1, (int)K};
  p.M = (int)M;
  p.N = (int)N;
  p.C = y;
  p.ldc = ldy;
  p.Clp = (u16*)y_lp;
  p.ldclp = N;
  p.bias = b;
  p.relu = relu;
  p.splitk = 1;
  return launch_gemm(p, dtype, (hipStream_t)stream);
}

extern "C" int fhvae_linear_bwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* y, int64_t ldy,
                                const float* dy, int64_t lddy, float* dy_masked, float* dx, int64_t lddx, float* dw,
                                int64_t lddw, float* db, int64_t M, int64_t K, int64_t N, int relu, int dx_accumulate,
                                void* stream) {
  FH_CHECK_PTR(dy);
  FH_CHECK_POS(M);
  FH_CHECK_POS(K);
  FH_CHECK_POS(N);
  FH_CHECK_I32(M);
  FH_CHECK_I32(K);
  FH_CHECK_I32(N);
  hipStream_t st = (hipStream_t)stream;
  const float* g = dy;
  int64_t ldg = lddy;
  if (relu) {
    FH_CHECK_PTR(y);
    FH_CHECK_PTR(dy_masked);
    int64_t n = M * N;
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)fh_cdiv(n, 256)), dim3(256), 0, st, dy, lddy, y, ldy, dy_masked, M, N);
    int e = fh_launch_status();
    if (e) return e;
    g = dy_masked;
    ldg = N;
  }
  if (dx) {
    FH_CHECK_PTR(w);
    // dx[M,K] = g[M,N] . w[N,K]: contraction over N; w is the KM operand B(k_out, n) = w[n*ldw + k_out]
    GemmParams p = {};
    p.seg[0] = Seg{g, ldg, 1, w, ldw, 0, (int)N};
    p.M = (int)M;
    p.N = (int)K;
    p.C = dx;
    p.ldc = lddx;
    p.mode = dx_accumulate ? 1 : 0;
    p.splitk = 1;
    int e = launch_gemm(p, FHVAE_F32, st);
    if (e) return e;
  }
  if (dw) {
    FH_CHECK_PTR(x);
    // dw[N,K] += g^T . x : contraction over M, both operands KM
    GemmParams p = {};
    p.seg[0] = Seg{g, ldg, 0, x, ldx, 0, (int)M};
    p.M = (int)N;
    p.N = (int)K;
    p.C = dw;
    p.ldc = lddw;
    p.splitk = 0;  // auto
    p.mode = 1;
    int e = launch_gemm(p, FHVAE_F32, st);
    if (e) return e;
  }
  if (db) {
    int e = launch_colsum(g, FHVAE_F32, ldg, db, nullptr, M, N, st);
    if (e) return e;
  }
  return FHVAE_OK;
}

extern "C" int fhvae_gauss_head_reparam_fwd(const void* h, int64_t ldh, const void* w_mu, const void* w_lv,
                                            const float* b_mu, const float* b_lv, const float* eps, float* mu,
                                            float* logvar, float* sample, int64_t M, int64_t K, int64_t D, int dtype,
                                            void* stream) {
  FH_CHECK_PTR(h);
  FH_CHECK_PTR(w_mu);
  FH_CHECK_PTR(w_lv);
  FH_CHECK_PTR(mu);
  FH_CHECK_PTR(logvar);
  if (eps && !sample) return FHVAE_ERR_NULL;
  FH_CHECK_POS(M);
  FH_CHECK_POS(K);
  FH_CHECK_POS(D);
  FH_CHECK_I32(M);
  FH_CHECK_I32(K);
  FH_CHECK_I32(D);
  if (dtype != FHVAE_F32 && dtype != FHVAE_BF16) return FHVAE_ERR_DTYPE;
  GemmParams ps[2] = {};
  for (int i = 0; i < 2; ++i) {  // mu and logvar share the operand h: one grouped launch
    ps[i].seg[0] = Seg{h, ldh, 1, i == 0 ? w_mu : w_lv, K, 1, (int)K, 0};
    ps[i].M = (int)M;
    ps[i].N = (int)D;
    ps[i].C = i == 0 ? mu : logvar;
    ps[i].ldc = D;
    ps[i].bias = i == 0 ? b_mu : b_lv;
    ps[i].splitk = 1;
  }
  int e = launch_gemm_group(ps, 2, dtype, (hipStream_t)stream);
  if (e) return e;
  if (eps) {
    int64_t n = M * D;
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3((unsigned)fh_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, mu, logvar,
                       eps, sample, n);
    return fh_launch_status();
  }
  return FHVAE_OK;
}

// mu | logvar of a Gaussian head WITHOUT sampling (the per-frame decoder head, simple_fhvae.py:98-103) as ONE projection: the two
// bf16 weight matrices stacked [2D, K], the outputs side by side in out[M, 2D] (mu = columns [0, D), logvar = [D, 2D))
extern "C" int fhvae_gauss_head_pair_fwd(const void* h_lp, int64_t ldh, const void* w_pair_lp, const float* b_mu, const float* b_lv,
                                         float* out, int64_t ldo, int64_t M, int64_t K, int64_t D, void* stream) {
  FH_CHECK_PTR(h_lp);
  FH_CHECK_PTR(w_pair_lp);
  FH_CHECK_PTR(out);
  FH_CHECK_POS(M);
  FH_CHECK_POS(K);
  FH_CHECK_POS(D);
  FH_CHECK_I32(M);
  hipStream_t st = (hipStream_t)stream;
  if (proj_eligible(h_lp, ldh, w_pair_lp, K, out, ldo, M, 2 * D, K) && D % 4 == 0)
    return launch_proj(h_lp, ldh, w_pair_lp, K, out, ldo, b_mu, M, 2 * D, K, st, b_lv, (int)D);
  GemmParams ps[2] = {};  // shapes the projection kernel does not take: the generic engine, one grouped launch
  for (int i = 0; i < 2; ++i) {
    ps[i].seg[0] = Seg{h_lp, ldh, 1, (const u16*)w_pair_lp + (int64_t)i * D * K, K, 1, (int)K, 0};
    ps[i].M = (int)M;
    ps[i].N = (int)D;
    ps[i].C = out + i * D;
    ps[i].ldc = ldo;
    ps[i].bias = i == 0 ? b_mu : b_lv;
    ps[i].splitk = 1;
  }
  return launch_gemm_group(ps, 2, FHVAE_BF16, st);
}

extern "C" int fhvae_head_pair_weights(const float* w_mu, const float* w_lv, void* wl_pair, void* wt_pair, int64_t ldt, int64_t D,
                                       int64_t K, void* stream) {
  FH_CHECK_PTR(w_mu);
  FH_CHECK_PTR(w_lv);
  if (!wl_pair && !wt_pair) return FHVAE_ERR_NULL;
  FH_CHECK_POS(D);
  FH_CHECK_POS(K);
  if (ldt < 2 * D) return FHVAE_ERR_SHAPE;
  hipLaunchKernelGGL(head_pair_weights_kernel, dim3((unsigned)fh_cdiv(ldt * K, 256)), dim3(256), 0, (hipStream_t)stream, w_mu, w_lv,
                     (u16*)wl_pair, (u16*)wt_pair, ldt, (int)D, (int)K);
  return fh_launch_status();
}

extern "C" int fhvae_gauss_reparam_pair_fwd(const float* out, int64_t ldo, const float* eps, float* sample, float* mu, float* logvar,
                                            int64_t M, int64_t D, void* stream) {
  FH_CHECK_PTR(out);
  FH_CHECK_PTR(eps);
  FH_CHECK_PTR(sample);
  FH_CHECK_POS(M);
  FH_CHECK_POS(D);
  if (ldo < 2 * D) return FHVAE_ERR_SHAPE;
  hipLaunchKernelGGL(reparam_pair_fwd_kernel, dim3((unsigned)fh_cdiv(M * D, 256)), dim3(256), 0, (hipStream_t)stream, out, ldo, eps,
                     sample, mu, logvar, M * D, (int)D);
  return fh_launch_status();
}

extern "C" int fhvae_gauss_reparam_bwd_pair(const float* d_mu, const float* d_logvar, const float* d_sample, int64_t ld_s, const float* eps,
                                            const float* logvar, int64_t ld_lv, void* g_lp, int64_t ldg, float* db_mu, float* db_lv,
                                            int64_t M, int64_t D, void* stream) {
  FH_CHECK_PTR(g_lp);
  FH_CHECK_POS(M);
  FH_CHECK_POS(D);
  if (d_sample && (!eps || !logvar)) return FHVAE_ERR_NULL;
  if (d_sample && ld_s < D) return FHVAE_ERR_SHAPE;
  if (ldg < 2 * D || ldg % 8) return FHVAE_ERR_SHAPE;
  if (((uintptr_t)g_lp) & 15) return FHVAE_ERR_ALIGN;
  const dim3 grid((unsigned)fh_cdiv(M * (ldg / 8), 256));
  hipStream_t st = (hipStream_t)stream;
  if ((db_mu || db_lv) && 256 % (ldg / 8) == 0) {  // whole rows per workgroup: the bias gradients' column sums ride along
    hipLaunchKernelGGL(reparam_bwd_pair_kernel<true>, grid, dim3(256), 0, st, d_mu, d_logvar, d_sample, ld_s, eps, logvar, ld_lv, (u16*)g_lp,
                       ldg, M, (int)D, db_mu, db_lv);
    return fh_launch_status();
  }
  hipLaunchKernelGGL(reparam_bwd_pair_kernel<false>, grid, dim3(256), 0, st, d_mu, d_logvar, d_sample, ld_s, eps, logvar, ld_lv, (u16*)g_lp, ldg,
                     M, (int)D, nullptr, nullptr);
  int e = fh_launch_status();
  if (e || !(db_mu || db_lv)) return e;
  return launch_colsum(g_lp, FHVAE_BF16, ldg, db_mu, db_lv, M, 2 * D, st, D);
}

extern "C" int fhvae_gauss_head_bwd_pair(const void* h_lp, int64_t ldh, const void* wt_pair, int64_t ldt, const void* g_lp, int64_t ldg,
                                         const float* col_sum, int64_t col_sum_rows, float* dh, int64_t lddh, float* dw_mu, float* dw_lv, float* db_mu,
                                         float* db_lv, int64_t M, int64_t K, int64_t D, void* stream) {
  FH_CHECK_PTR(g_lp);
  FH_CHECK_POS(M);
  FH_CHECK_POS(K);
  FH_CHECK_POS(D);
  FH_CHECK_I32(M);
  FH_CHECK_I32(K);
  FH_CHECK_I32(2 * D);
  if (ldg < 2 * D || ldt < 2 * D) return FHVAE_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const u16* g = (const u16*)g_lp;
  int e;
  if (dh) {  // dh[M,K] = g[M, 0..ldg) . wt_pair[K, 0..ldg)^T (the zero padding of both contributes nothing)
    FH_CHECK_PTR(wt_pair);
    if (ldt == ldg && proj_eligible(g, ldg, wt_pair, ldt, dh, lddh, M, K, ldg)) {
      e = launch_proj(g, ldg, wt_pair, ldt, dh, lddh, nullptr, M, K, ldg, st);
    } else {
      GemmParams p = {};
      p.seg[0] = Seg{g, ldg, 1, wt_pair, ldt, 1, (int)(2 * D)};
      p.M = (int)M;
      p.N = (int)K;
      p.C = dh;
      p.ldc = lddh;
      p.splitk = 1;
      e = launch_gemm(p, FHVAE_BF16, st);
    }
    if (e) return e;
  }
  if (dw_mu || dw_lv) {
    FH_CHECK_PTR(h_lp);
    FH_CHECK_PTR(dw_mu);
    FH_CHECK_PTR(dw_lv);
    WgProblem wp[2] = {};
    for (int i = 0; i < 2; ++i) {
      wp[i].A = g + i * D, wp[i].B = (const u16*)h_lp, wp[i].C = i == 0 ? dw_mu : dw_lv;
      wp[i].lda = ldg, wp[i].ldb = ldh, wp[i].ldc = K;
      wp[i].M = (int)D, wp[i].N = (int)K, wp[i].K = (int)M;
      wp[i].a_col0 = (int)(i * D);
    }
    if (wgrad_eligible(wp[0]) && wgrad_eligible(wp[1])) {
      e = launch_wgrad(wp, 2, st);
    } else {
      GemmParams p = {};
      p.seg[0] = Seg{g, ldg, 0, h_lp, ldh, 0, (int)M};
      p.M = (int)(2 * D);
      p.N = (int)K;
      p.C = dw_mu;
      p.C2 = dw_lv;
      p.c_split = (int)D;
      p.ldc = K;
      p.mode = 1;
      p.splitk = 0;
      e = launch_gemm(p, FHVAE_BF16, st);
    }
    if (e) return e;
  }
  if (db_mu || db_lv) {
    if (col_sum) {
      if (col_sum_rows <= 0 || col_sum_rows > INT32_MAX) return FHVAE_ERR_SHAPE;
      const unsigned gy = (unsigned)std::min<int64_t>(32, fh_cdiv(col_sum_rows, 32));
      hipLaunchKernelGGL(add_split_kernel, dim3((unsigned)fh_cdiv(2 * D, 64), gy), dim3(256), 0, st, col_sum, (int)col_sum_rows, db_mu,
                         db_lv, (int)D);
      return fh_launch_status();
    }
    return launch_colsum(g, FHVAE_BF16, ldg, db_mu, db_lv, M, 2 * D, st, D);
  }
  return FHVAE_OK;
}

extern "C" int fhvae_gauss_reparam_bwd(const float* d_mu, const float* d_logvar, const float* d_sample, const float* eps,
                                       const float* logvar, float* g_mu, float* g_lv, int64_t n, void* stream) {
  FH_CHECK_PTR(g_mu);
  FH_CHECK_PTR(g_lv);
  FH_CHECK_POS(n);
  if (d_sample && (!eps || !logvar)) return FHVAE_ERR_NULL;
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3((unsigned)fh_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, d_mu, d_logvar,
                     d_sample, eps, logvar, g_mu, g_lv, n);
  return fh_launch_status();
}

extern "C" int fhvae_gauss_head_bwd(const float* h, int64_t ldh, const float* w_mu, const float* w_lv, const float* d_mu,
                                    const float* d_logvar, const float* d_sample, const float* eps, const float* logvar,
                                    float* g_ws, float* dh, int64_t lddh, float* dw_mu, float* dw_lv, float* db_mu,
                                    float* db_lv, int64_t M, int64_t K, int64_t D, void* stream) {
  FH_CHECK_PTR(g_ws);
  FH_CHECK_POS(M);
  FH_CHECK_POS(K);
  FH_CHECK_POS(D);
  FH_CHECK_I32(M);
  FH_CHECK_I32(K);
  FH_CHECK_I32(2 * D);
  if (d_sample && (!eps || !logvar)) return FHVAE_ERR_NULL;
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = M * D;
  hipLaunchKernelGGL(reparam_bwd_cat_kernel<float>, dim3((unsigned)fh_cdiv(n, 256)), dim3(256), 0, st, d_mu, d_logvar, d_sample, eps,
                     logvar, g_ws, n, (int)D);
  int e = fh_launch_status();
  if (e) return e;
  if (dh) {  // dh[M,K] = g_mu . W_mu + g_lv . W_lv: two K-segments of one contraction, the weights as KM operands
    FH_CHECK_PTR(w_mu);
    FH_CHECK_PTR(w_lv);
    GemmParams p = {};
    p.seg[0] = Seg{g_ws, 2 * D, 1, w_mu, K, 0, (int)D};
    p.seg[1] = Seg{g_ws + D, 2 * D, 1, w_lv, K, 0, (int)D};
    p.M = (int)M;
    p.N = (int)K;
    p.C = dh;
    p.ldc = lddh;
    p.splitk = 1;
    e = launch_gemm(p, FHVAE_F32, st);
    if (e) return e;
  }
  if (dw_mu || dw_lv) {  // [dw_mu; dw_lv][2D,K] += g^T . h: one contraction over the M rows, output rows split at D
    FH_CHECK_PTR(h);
    FH_CHECK_PTR(dw_mu);
    FH_CHECK_PTR(dw_lv);
    GemmParams p = {};
    p.seg[0] = Seg{g_ws, 2 * D, 0, h, ldh, 0, (int)M};
    p.M = (int)(2 * D);
    p.N = (int)K;
    p.C = dw_mu;
    p.C2 = dw_lv;
    p.c_split = (int)D;
    p.ldc = K;
    p.mode = 1;
    p.splitk = 0;
    e = launch_gemm(p, FHVAE_F32, st);
    if (e) return e;
  }
  if (db_mu || db_lv) return launch_colsum(g_ws, FHVAE_F32, 2 * D, db_mu, db_lv, M, 2 * D, st, D);
  return FHVAE_OK;
}

