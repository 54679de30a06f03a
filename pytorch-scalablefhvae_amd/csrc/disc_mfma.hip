// disc_mfma.hip -- K5 on the matrix cores (large tables; simple_fhvae.py:119-122).
//
//   logit[b,s] = -c |q_b - t_s|^2 = 2c (q_b . t_s) - c |q_b|^2 - c |t_s|^2
//
// The cross term q.t is a (B x S x D) contraction: it runs on exact-f32 MFMA (v_mfma_f32_16x16x4_f32: every product
// and every accumulation step is an f32 fma, so the expanded form costs only the cancellation of the norms, ~1e-5
// absolute on logits of 10..1000), and the exp / online log-sum-exp runs on the VALU under it (separate pipes).
// The VALU kernels in loss.hip spend 2*D lane-instructions per (query,row) pair on the distance alone.
//
// One kernel template, three uses.  A workgroup keeps 256 STATIONARY vectors X (64 per wave, as MFMA B-operand
// fragments in registers) and streams the other set Y through LDS in tiles of 64:
//   MODE 0  forward   X = queries,    Y = table rows : per-query online (max, sumexp) partials per row chunk
//   MODE 1  dq        X = queries,    Y = table rows : G_x = sum_y w[y,x] Y_y  on MFMA again -> dq = 2c (G - q W)
//   MODE 1  dtable    X = table rows, Y = queries    : same formula gives dt = 2c (G - t W)
// with w = g (softmax - onehot) recomputed from the per-query (max, sumexp) of the forward.  The logit tile comes out
// of the MFMA with the stationary index on the lanes (col = lane&15) and the streamed index in the 4 accumulator
// registers (row = 4*(lane>>4)+r): exactly the B-operand layout of the second product, so w never leaves registers.
#include <algorithm>

#include "disc_mfma.h"

#include <cstdlib>
#include <type_traits>

namespace fh {


template <int D>
__device__ __forceinline__ int yoff(int row, int ch) {  // byte offset of 16-byte chunk ch of LDS row `row`
  constexpr int CHN = D / 4;
  return row * (D * 4) + ((ch ^ (row & (CHN >= 8 ? 7 : CHN - 1))) << 4);
}

// The (query, own table row) pairs are NOT computed here.  The expanded form's absolute error ~1e-7 * 2c (|q|^2 + |t|^2) is
// harmless on far rows (their softmax weight is 0 either way) but it is the whole signal on the pair training drives together
// (q -> table[idx]).  That one logit per query is therefore masked out of these kernels (logit = -inf: no contribution to the
// log-sum-exp, zero weight in both backward passes) and taken in the DIRECT form -c |q - t|^2 by the callers: the forward's
// combine kernel merges exp(target - max) into the row sum, the backward adds the pair's gradient in disc_own_bwd_kernel
// (loss.hip).  CE -> log(1 + sum_others) and p_target - 1 -> -sum_others then come out cleanly however large the norms are.
template <int D, int MODE>
__global__ __launch_bounds__(256, 2) void disc_mfma_kernel(DiscMfmaArgs a) {
  constexpr int CHN = D / 4;   // 16-byte chunks per vector
  constexpr int NJ = D / 16;   // 16-k groups (also 16-wide d blocks)
  constexpr int YT = 64;       // streamed vectors per LDS tile
  __shared__ __attribute__((aligned(16))) char ytile[YT * D * 4];
  __shared__ __attribute__((aligned(16))) float yn[YT];
  __shared__ float ymax[YT], yinv[YT];
  __shared__ int ytgt[YT];
  __shared__ int yown[YT / 16];  // streamed queries: block b holds one whose own row is among this workgroup's stationary rows
  // MODE 2 (queries stationary only) = MODE 1 plus the STREAMED side's gradient from the same weights (disc_lp.hip has the
  // bf16 form and the reasoning): G2[y][d] = sum_x w[y,x] X[x][d] contracts over x, which sits on the lanes of the logit tile, so
  // each wave passes its weights through a private [16 y][64 x] f32 LDS image (4 ds_write_b32 per tile, one ds_read_b128 back:
  // lane (g, i) gets w[y = i][x = 16t + 4g .. + 3], the A operands of the 4 MFMAs of tile t); B = X[x0 + 16t + 4g + q][16dj + i].  WY[y] = sum_x w[y,x] is summed on the VALU from the same transposed registers.  Per-wave LDS slots,
  // partial buffers and the two reduce kernels below instead of atomics.
  constexpr bool BW = MODE >= 1, BOTH = MODE == 2;
  constexpr int kWLd = 68;        // row stride of the weight image (floats): the 4 lane groups write different banks
  constexpr int kDtLd = D + 4;    // row stride of a slot
  __shared__ __attribute__((aligned(16))) float wimg[BOTH ? 4 : 1][BOTH ? 16 * kWLd : 4];
  __shared__ float wy_lds[BOTH ? 4 : 1][BOTH ? YT : 1];
  constexpr int kRed = !BW ? 1 : (BOTH && 4 * YT * kDtLd > 256 * (D + 1) ? 4 * YT * kDtLd : 256 * (D + 1));
  __shared__ __attribute__((aligned(16))) float red[kRed];  // epilogue: tr[256][D + 1]; MODE 2, in the loop: 4 slots [64 y][kDtLd]
  float (*tr)[D + 1] = (float (*)[D + 1]) red;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, i = lane & 15;
  // blockIdx.x = chunk of the STREAMED set: round-robin XCD placement then gives every XCD (private L2) 1/8 of the
  // streamed vectors (the table, 128 MB at 1M rows) instead of all of them
  const int x0 = blockIdx.y * 256 + wave * 64;
  const int y_begin = blockIdx.x * a.chunk;
  const int y_end = min(a.NY, y_begin + a.chunk);
  const float gscale = BW ? (*a.gsc) * a.gmul : 0.f;

  // ---- stationary fragments: lane (g,i) of tile t holds X[x0+16t+i][4g+16jj .. +3]
  uint4 xf[4][NJ];
  float xn[4], xmax[4], xinv[4];
  int xtgt[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int x = x0 + t * 16 + i;
    const bool ok = x < a.NX;
    float nrm = 0.f;
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      uint4 u = make_uint4(0, 0, 0, 0);
      if (ok) u = *(const uint4*)(a.X + (int64_t)x * D + 4 * g + 16 * jj);
      xf[t][jj] = u;
      const float f0 = __uint_as_float(u.x), f1 = __uint_as_float(u.y), f2 = __uint_as_float(u.z), f3 = __uint_as_float(u.w);
      nrm += f0 * f0 + f1 * f1 + f2 * f2 + f3 * f3;
    }
    nrm += __shfl_xor(nrm, 16, 64);
    nrm += __shfl_xor(nrm, 32, 64);
    xn[t] = nrm;
    xmax[t] = 0.f;
    xinv[t] = 0.f;
    xtgt[t] = -1;
    if (a.x_is_query) {
      if (ok) {
        const int64_t tg = a.idx[x] - a.row0;
        xtgt[t] = (tg >= 0 && tg < a.NY) ? (int)tg : -1;
        if (BW) {
          xmax[t] = a.rmax[x];
          xinv[t] = gscale / a.rsum[x];  // (the upstream scale rides on the normaliser)
        }
      }
    } else {
      xtgt[t] = ok ? x : -2;  // table row index: a streamed query hits it when its target == x
    }
  }
  float m[4], ssum[4], wsum[4];
  f32x4 gacc[4][NJ];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    m[t] = -INFINITY;
    ssum[t] = 0.f;
    wsum[t] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) gacc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- stream Y in tiles of 64 vectors
  constexpr int LOADS = YT * CHN / 256;  // 16-byte chunks per thread per tile (2 for D=32)
  uint4 st[LOADS];
  auto issue = [&](int y0) {
#pragma unroll
    for (int p = 0; p < LOADS; ++p) {
      const int id = tid + p * 256;
      const int row = id / CHN, ch = id % CHN;
      const int y = y0 + row;
      st[p] = (y < y_end) ? *(const uint4*)(a.Y + (int64_t)y * D + ch * 4) : make_uint4(0, 0, 0, 0);
    }
  };
  if (y_begin < y_end) issue(y_begin);
  for (int y0 = y_begin; y0 < y_end; y0 += YT) {
#pragma unroll
    for (int p = 0; p < LOADS; ++p) {
      const int id = tid + p * 256;
      *(uint4*)(ytile + yoff<D>(id / CHN, id % CHN)) = st[p];
    }
    __syncthreads();
    if (y0 + YT < y_end) issue(y0 + YT);
    // per-y scalars: norm (4 lanes per vector), and for streamed queries their (max, 1/sum, target)
    {
      const int row = tid >> 2, part = tid & 3;  // 64 rows x 4 lanes
      float nrm = 0.f;
#pragma unroll
      for (int c4 = part; c4 < CHN; c4 += 4) {
        const uint4 u = *(const uint4*)(ytile + yoff<D>(row, c4));
        const float f0 = __uint_as_float(u.x), f1 = __uint_as_float(u.y), f2 = __uint_as_float(u.z), f3 = __uint_as_float(u.w);
        nrm += f0 * f0 + f1 * f1 + f2 * f2 + f3 * f3;
      }
      nrm += __shfl_xor(nrm, 1, 64);
      nrm += __shfl_xor(nrm, 2, 64);
      bool mine = false;
      if (part == 0) {
        yn[row] = nrm;
        const int y = y0 + row;
        if (!a.x_is_query) {
          const bool ok = y < y_end;
          ymax[row] = ok && MODE == 1 ? a.rmax[y] : 0.f;
          yinv[row] = ok && MODE == 1 ? gscale / a.rsum[y] : 0.f;  // (the upstream scale rides on the normaliser)
          int tg = -3;
          if (ok) {
            const int64_t v = a.idx[y] - a.row0;
            tg = (v >= 0 && v < a.NX) ? (int)v : -3;
          }
          ytgt[row] = tg;
          mine = tg >= (int)blockIdx.y * 256 && tg < (int)blockIdx.y * 256 + 256;
        }
      }
      {  // wave w holds the 16 rows of block w
        const bool any = __any(mine);
        if (lane == 0) yown[wave] = any ? 1 : 0;
      }
    }
    __syncthreads();

#pragma unroll 1
    for (int yb = 0; yb < YT / 16; ++yb) {
      if (y0 + yb * 16 >= y_end) break;
      // A fragments of the logit product: Y[yb*16+i][4g+16jj .. +3]
      uint4 af[NJ];
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) af[jj] = *(const uint4*)(ytile + yoff<D>(yb * 16 + i, g + 4 * jj));
      const float4 ynv = *(const float4*)(yn + yb * 16 + 4 * g);
      const float ynr[4] = {ynv.x, ynv.y, ynv.z, ynv.w};
      float ymx[4], yiv[4];
      int ytg[4];
      if (MODE == 1 && !a.x_is_query) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ymx[r] = ymax[yb * 16 + 4 * g + r];
          yiv[r] = yinv[yb * 16 + 4 * g + r];
          ytg[r] = ytgt[yb * 16 + 4 * g + r];
        }
      }
      // interior blocks without a (query, own row) pair take the body without the validity / own-row selects (disc_lp.hip)
      const int ybase = y0 + yb * 16;
      const bool whole = ybase + 16 <= y_end && x0 + 64 <= a.NX;
      const bool own_blk = !a.x_is_query && yown[yb] != 0;
      const float c2 = 2.f * a.c;
      f32x4 oacc[BOTH ? NJ : 1];
      float wyp = 0.f;
      if constexpr (BOTH) {
#pragma unroll
        for (int dj = 0; dj < NJ; ++dj) oacc[dj] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // the logit product of tile t: 8 dependent MFMAs
      auto logits = [&](int t) __attribute__((always_inline)) -> f32x4 {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
          const uint4 ua = af[jj], ub = xf[t][jj];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ua.x), __uint_as_float(ub.x), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ua.y), __uint_as_float(ub.y), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ua.z), __uint_as_float(ub.z), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ua.w), __uint_as_float(ub.w), acc, 0, 0, 0);
        }
        return acc;
      };
      auto tile = [&](int t, auto masked_c, const f32x4& acc) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_c)::value;
        // MODE 2: B operands of the streamed side's product, xb[q][dj] = X[x0 + 16t + 4g + q][16dj + i], fetched per tile (L1 /
        // L2 hits) rather than held: 32 more stationary registers would halve the occupancy.  Vectors past NX are clamped: their
        // weights are zero.
        const bool xok = x0 + t * 16 + i < a.NX;
        const float cxn = a.c * xn[t];
        float lg[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          lg[r] = c2 * acc[r] - (a.c * ynr[r] + cxn);
          if constexpr (MASKED) {
            const int y = ybase + 4 * g + r;
            if (!(xok && y < y_end)) lg[r] = -INFINITY;
            // the query's own row is handled exactly by the callers (see the note above the kernel)
            const bool own = (MODE == 1 && !a.x_is_query) ? ytg[r] == xtgt[t] : xtgt[t] == y;
            if (own) lg[r] = -INFINITY;
          }
        }
        if constexpr (MODE == 0) {
          const float gm = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
          if (gm > m[t]) {
            ssum[t] *= __expf(m[t] - gm);
            m[t] = gm;
          }
          if (!MASKED || m[t] > -INFINITY) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ssum[t] += __expf(lg[r] - m[t]);
          }
        } else {
          float w[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p;
            if (a.x_is_query)
              p = __expf(lg[r] - xmax[t]) * xinv[t];
            else
              p = __expf(lg[r] - ymx[r]) * yiv[r];
            w[r] = (!MASKED || lg[r] > -INFINITY) ? p : 0.f;  // (own pairs: masked above, added by disc_own_bwd_kernel)
            wsum[t] += w[r];
          }
          float xb[BOTH ? 4 : 1][NJ];  // MODE 2: requested behind the exp arithmetic; their latency hides under the G product's MFMAs
          if constexpr (BOTH) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int x = min(x0 + 16 * t + 4 * g + q, a.NX - 1);
#pragma unroll
              for (int dj = 0; dj < NJ; ++dj) xb[q][dj] = a.X[(int64_t)x * D + 16 * dj + i];
            }
          }
          // G^T[d][x] += sum_y Y[y][d] * w[y][x]: A = Y^T from LDS (lane: d = 16*dj + i, y = 4g + r), B = w[r]
#pragma unroll
          for (int dj = 0; dj < NJ; ++dj) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = yb * 16 + 4 * g + r, d = dj * 16 + i;
              const float av = *(const float*)(ytile + yoff<D>(row, d >> 2) + (d & 3) * 4);
              gacc[t][dj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w[r], gacc[t][dj], 0, 0, 0);
            }
          }
          if constexpr (BOTH) {
            float* wi = wimg[wave];
#pragma unroll
            for (int r = 0; r < 4; ++r) wi[(4 * g + r) * kWLd + 16 * t + i] = w[r];
            asm volatile("" ::: "memory");  // (LDS operations of one wave complete in order; only the compiler must keep it)
            const float4 wt = *(const float4*)(wi + i * kWLd + 16 * t + 4 * g);  // w[y = i][x = 16t + 4g + q]
            asm volatile("" ::: "memory");
            const float wq[4] = {wt.x, wt.y, wt.z, wt.w};
            wyp += (wq[0] + wq[1]) + (wq[2] + wq[3]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int dj = 0; dj < NJ; ++dj) oacc[dj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[q], xb[q][dj], oacc[dj], 0, 0, 0);
          }
        }
      };
      // Interior blocks with no masked tile (all but a few per cent): ONE basic block for the four tiles, tile t + 1's logit MFMAs
      // issued before tile t's exp / weight arithmetic, so that the matrix pipe works under the VALU of the same wave (with a
      // branch per tile the chains ran one after the other: MfmaUtil 46 % forward; S = 1M forward 1.81 -> 1.65 ms).
      bool any_masked = !whole || own_blk;
      if (a.x_is_query) {
#pragma unroll
        for (int t = 0; t < 4; ++t) any_masked = any_masked || __any((unsigned)(xtgt[t] - ybase) < 16u);
      }
      if (!BOTH && !any_masked) {  // (the one-pass backward has no registers for a second accumulator in flight: 101 spills)
        f32x4 nxt = logits(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f32x4 cur = nxt;
          if (t + 1 < 4) nxt = logits(t + 1);
          tile(t, std::false_type{}, cur);
        }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          bool masked = !whole || own_blk;
          if (a.x_is_query) masked = masked || __any((unsigned)(xtgt[t] - ybase) < 16u);
          const f32x4 acc = logits(t);
          if (masked)
            tile(t, std::true_type{}, acc);
          else
            tile(t, std::false_type{}, acc);
        }
      }
      if constexpr (BOTH) {  // lane holds out[y = 4g + r][d = 16dj + i] -> the wave's slot; wyp: the partial row sum of y = i
        float* slot = red + wave * (YT * kDtLd);
#pragma unroll
        for (int dj = 0; dj < NJ; ++dj)
#pragma unroll
          for (int r = 0; r < 4; ++r) slot[(yb * 16 + 4 * g + r) * kDtLd + 16 * dj + i] = oacc[dj][r];
        wyp += __shfl_xor(wyp, 16, 64);
        wyp += __shfl_xor(wyp, 32, 64);
        if (g == 0) wy_lds[wave][yb * 16 + i] = wyp;
      }
    }
    __syncthreads();
    if constexpr (BOTH) {  // the tile's four slots summed into this x-tile's slice of the partial buffers (plain stores)
      for (int e = tid; e < YT * D; e += 256) {
        const int row = e / D, d = e % D, o = row * kDtLd + d;
        if (y0 + row < y_end)
          a.G2[((int64_t)blockIdx.y * a.NY + y0 + row) * D + d] =
              (red[o] + red[YT * kDtLd + o]) + (red[2 * YT * kDtLd + o] + red[3 * YT * kDtLd + o]);
      }
      if (tid < YT && y0 + tid < y_end)
        a.WY[(int64_t)blockIdx.y * a.NY + y0 + tid] = (wy_lds[0][tid] + wy_lds[1][tid]) + (wy_lds[2][tid] + wy_lds[3][tid]);
    }
  }

  if constexpr (MODE == 0) {
    // merge the 4 lane groups that share a stationary vector, then one partial per (chunk, x)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float mm = m[t], ss = ssum[t];
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {
        const float om = __shfl_xor(mm, o, 64), os = __shfl_xor(ss, o, 64);
        const float nm = fmaxf(mm, om);
        ss = (nm == -INFINITY) ? 0.f : ss * __expf(mm - nm) + os * __expf(om - nm);
        mm = nm;
      }
      const int x = x0 + t * 16 + i;
      if (g == 0 && x < a.NX) a.part[(int64_t)blockIdx.x * a.NX + x] = make_float2(mm, ss);
    }
  } else {
    if constexpr (BOTH) __syncthreads();  // (tr shares its memory with the slots the last tile's sums were read from)
    // grad_x = 2c (G - X W); lane holds G[x = 16t+i][d = 16dj + 4g + reg]; transpose through LDS -> row-contiguous atomics
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float ws = wsum[t];
      ws += __shfl_xor(ws, 16, 64);
      ws += __shfl_xor(ws, 32, 64);
#pragma unroll
      for (int dj = 0; dj < NJ; ++dj) {
        const int x = x0 + t * 16 + i;
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x < a.NX) xv = *(const float4*)(a.X + (int64_t)x * D + dj * 16 + 4 * g);
        const float xr[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          tr[wave * 64 + t * 16 + i][dj * 16 + 4 * g + r] = 2.f * a.c * (gacc[t][dj][r] - xr[r] * ws);
      }
    }
    __syncthreads();
    for (int e = tid; e < 256 * D; e += 256) {
      const int rr = e / D, d = e % D;
      const int x = blockIdx.y * 256 + rr;
      if (x < a.NX) {
        if constexpr (BOTH)
          a.G[((int64_t)blockIdx.x * a.NX + x) * D + d] = tr[rr][d];  // this chunk's slice of the partial buffer
        else
          atomicAdd(a.G + (int64_t)x * D + d, tr[rr][d]);
      }
    }
  }
}

// the reductions of the one-pass backward's partials (disc_lp.hip, MODE 2)
// dY[y][d] += 2c (sum_xt G2[xt][y][d] - Y[y][d] sum_xt WY[xt][y])
__global__ void disc_dt_finish_kernel(float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ g2,
                                      const float* __restrict__ wy, int nxt, float c2, int64_t NY, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NY * D) return;
  const int64_t row = i / D;
  float sg = 0.f, sw = 0.f;
  for (int t = 0; t < nxt; ++t) {
    sg += g2[(int64_t)t * NY * D + i];
    sw += wy[(int64_t)t * NY + row];
  }
  dy[i] += c2 * (sg - y[i] * sw);
}
// dX[i] = sum_chunks G[chunk][i]
__global__ void disc_dq_reduce_kernel(float* __restrict__ dx, const float* __restrict__ g, int nchunks, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int c = 0;
  for (; c + 4 <= nchunks; c += 4) {
    a0 += g[(int64_t)c * n + i], a1 += g[(int64_t)(c + 1) * n + i], a2 += g[(int64_t)(c + 2) * n + i], a3 += g[(int64_t)(c + 3) * n + i];
  }
  for (; c < nchunks; ++c) a0 += g[(int64_t)c * n + i];
  dx[i] = (a0 + a1) + (a2 + a3);
}

// streamed vectors per workgroup for about `target` workgroups.  Forward (one partial per (chunk, x)): 1024.  Backward: every
// workgroup adds its whole 256 x D partial gradient with atomics, so fewer, longer chunks pay (c2, S = 4600: 0.103 -> 0.078 ms per
// step with 512; 384 and fewer lose on the large tables: S = 1M backward 6.4 ms with 512, 7.5 ms with 384)
int mfma_chunk(int64_t nx, int64_t ny, int target) {
  const int64_t xt = fh_cdiv(nx, 256);
  int64_t want = fh_cdiv(target, xt);
  int64_t chunk = fh_cdiv(fh_cdiv(ny, want), 64) * 64;
  if (chunk < 64) chunk = 64;
  return (int)chunk;
}

// host-side entry points used by loss.hip
bool disc_mfma_supported(int64_t B, int64_t S, int64_t D) { return (D == 32 || D == 16) && B * S >= (int64_t)1 << 16; }

int64_t disc_mfma_ws_bytes(int64_t B, int64_t S) {
  const int chunk = mfma_chunk(B, S);
  return fh_cdiv(S, chunk) * B * (int64_t)sizeof(float2);
}

// one-pass backward: the queries go in groups of `tiles` 256-query tiles; a group needs its chunks' partials of dq
// (nchunks x rows x D floats) and tiles x S x (D + 1) floats of the streamed side's partial sums
static int64_t onepass_group_bytes(int64_t tiles, int64_t B, int64_t S, int64_t D) {
  const int64_t rows = std::min<int64_t>(B, tiles * 256);
  const int64_t nchunks = fh_cdiv(S, mfma_chunk(rows, S, 512));
  return (nchunks * rows * D + tiles * S * (D + 1)) * (int64_t)sizeof(float);
}
// the most tiles per group (<= all of them) whose partials fit `bytes`; 0: not even one
static int64_t onepass_group_tiles(int64_t bytes, int64_t B, int64_t S, int64_t D) {
  const int64_t nxt = fh_cdiv(B, 256);
  int64_t lo = 0, hi = nxt;  // (the size grows with the tile count)
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) / 2;
    if (onepass_group_bytes(mid, B, S, D) <= bytes) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}
int64_t disc_onepass_ws_bytes(int64_t B, int64_t S, int64_t D) {
  const int64_t t = onepass_group_tiles(kOnePassWsCap, B, S, D);
  return t > 0 ? onepass_group_bytes(t, B, S, D) : 0;
}

int disc_mfma_fwd(const float* q, const float* table, const int64_t* idx, int64_t row0, float c, float2* part, int* nchunks,
                  int64_t B, int64_t S, int64_t D, int lp, hipStream_t st) {
  DiscMfmaArgs a = {};
  a.X = q;
  a.Y = table;
  a.NX = (int)B;
  a.NY = (int)S;
  a.c = c;
  a.x_is_query = 1;
  a.idx = idx;
  a.row0 = row0;
  a.part = part;
  a.chunk = mfma_chunk(B, S);
  *nchunks = (int)fh_cdiv(S, a.chunk);
  dim3 grid((unsigned)*nchunks, (unsigned)fh_cdiv(B, 256));
  if (lp && D == 32)
    disc_lp_launch(a, 0, grid, st);
  else if (D == 32)
    hipLaunchKernelGGL((disc_mfma_kernel<32, 0>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((disc_mfma_kernel<16, 0>), grid, dim3(256), 0, st, a);
  return fh_launch_status();
}

int disc_mfma_bwd(const float* q, const float* table, const int64_t* idx, int64_t row0, float c, const float* rmax,
                  const float* rsum, const float* gsc, float gmul, float* dq, float* dtable, float* ws, int64_t ws_bytes, int64_t B,
                  int64_t S, int64_t D, int lp, hipStream_t st) {
  DiscMfmaArgs a = {};
  a.c = c;
  a.idx = idx;
  a.row0 = row0;
  a.rmax = rmax;
  a.rsum = rsum;
  a.gsc = gsc;
  a.gmul = gmul;
  const int64_t gtiles = (dq && dtable && ws) ? onepass_group_tiles(ws_bytes, B, S, D) : 0;
  if (gtiles > 0) {
    // one pass: stationary = queries, streamed = table rows; dq as in the two-pass form, dtable from the same weights.  Query
    // groups of gtiles tiles, one after the other on the same workspace (dtable accumulates over the groups)
    for (int64_t x0 = 0; x0 < B; x0 += gtiles * 256) {
      const int64_t nb = std::min<int64_t>(B - x0, gtiles * 256), nxt = fh_cdiv(nb, 256);
      a.X = q + x0 * D;
      a.Y = table;
      a.NX = (int)nb;
      a.NY = (int)S;
      a.x_is_query = 1;
      a.idx = idx + x0;
      a.rmax = rmax + x0;
      a.rsum = rsum + x0;
      a.chunk = mfma_chunk(nb, S, 512);
      const int64_t nchunks = fh_cdiv(S, a.chunk);
      a.G = ws;                       // [nchunks][nb, D]
      a.G2 = a.G + nchunks * nb * D;  // [nxt][S, D]
      a.WY = a.G2 + nxt * S * D;      // [nxt][S]
      dim3 grid((unsigned)nchunks, (unsigned)nxt);
      if (lp && D == 32)
        disc_lp_launch(a, 2, grid, st);
      else if (D == 32)
        hipLaunchKernelGGL((disc_mfma_kernel<32, 2>), grid, dim3(256), 0, st, a);
      else
        hipLaunchKernelGGL((disc_mfma_kernel<16, 2>), grid, dim3(256), 0, st, a);
      int e = fh_launch_status();
      if (e) return e;
      hipLaunchKernelGGL(disc_dq_reduce_kernel, dim3((unsigned)fh_cdiv(nb * D, 256)), dim3(256), 0, st, dq + x0 * D, a.G, (int)nchunks, nb * D);
      hipLaunchKernelGGL(disc_dt_finish_kernel, dim3((unsigned)fh_cdiv(S * D, 256)), dim3(256), 0, st, dtable, table, a.G2, a.WY, (int)nxt,
                         2.f * c, S, (int)D);
      e = fh_launch_status();
      if (e) return e;
    }
    return FHVAE_OK;
  }
  if (dq) {  // stationary = queries, streamed = table rows; the workgroups ADD their partial gradients: zero first
    hipError_t he = hipMemsetAsync(dq, 0, (size_t)(B * D) * sizeof(float), st);
    if (he != hipSuccess) return (int)he;
    a.X = q;
    a.Y = table;
    a.NX = (int)B;
    a.NY = (int)S;
    a.x_is_query = 1;
    a.G = dq;
    a.chunk = mfma_chunk(B, S, 512);
    dim3 grid((unsigned)fh_cdiv(S, a.chunk), (unsigned)fh_cdiv(B, 256));
    if (lp && D == 32)
      disc_lp_launch(a, 1, grid, st);
    else if (D == 32)
      hipLaunchKernelGGL((disc_mfma_kernel<32, 1>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((disc_mfma_kernel<16, 1>), grid, dim3(256), 0, st, a);
    int e = fh_launch_status();
    if (e) return e;
  }
  if (dtable) {  // stationary = table rows, streamed = queries
    a.X = table;
    a.Y = q;
    a.NX = (int)S;
    a.NY = (int)B;
    a.x_is_query = 0;
    a.G = dtable;
    a.chunk = mfma_chunk(S, B, 512);
    dim3 grid((unsigned)fh_cdiv(B, a.chunk), (unsigned)fh_cdiv(S, 256));
    if (lp && D == 32)
      disc_lp_launch(a, 1, grid, st);
    else if (D == 32)
      hipLaunchKernelGGL((disc_mfma_kernel<32, 1>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((disc_mfma_kernel<16, 1>), grid, dim3(256), 0, st, a);
    int e = fh_launch_status();
    if (e) return e;
  }
  return FHVAE_OK;
}

}  // namespace fh
