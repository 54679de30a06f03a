// disc_lp.hip -- K5 (simple_fhvae.py:119-122) for the bf16 compute mode: the structure of disc_mfma.hip with both products on
// v_mfma_f32_16x16x32_bf16 and SPLIT operands.
//
//   q.t = q_hi.t_hi + q_hi.t_lo + q_lo.t_hi   (v = v_hi + v_lo, v_hi = bf16(v), v_lo = bf16(v - v_hi): 16 mantissa bits each)
//
// One 16x16x32 MFMA covers the whole D = 32 contraction of a 16 x 16 logit tile in 16 cycles: 3 of them (48 cycles) replace the
// 8 v_mfma_f32_16x16x4_f32 (256 cycles) of the f32 kernel, at an error of ~2^-16 |q||t| on the cross term (the norms are exact
// f32), i.e. ~1e-3 absolute on a logit at N(0,1)-scale vectors: below the bf16 tolerance of everything upstream (z2_mu itself
// comes out of bf16 LSTM nets), and the one logit that matters when training converges -- the query's own row -- is not
// computed here at all (masked; the callers take it in the direct f32 form, see disc_mfma.hip).
// The second product G = sum_y w[y,x] Y[y] (both backward passes) takes w in bf16 and Y split: G = Y_hi^T.w + Y_lo^T.w, with the
// weight sum W accumulated from the SAME rounded w, so that the gradient 2c (G - x W) = 2c sum w' (y - x) keeps the small
// differences (y - x) of near rows (a W from unrounded w would leave sum (w' - w) y, ~0.4 % of |y|, in it).
// Y^T comes out of the [y][d] LDS image through ds_read_b64_tr_b16; the k order of that read (rows 4g..4g+3 and 16+4g..16+4g+3
// of a 32-row block for lane group g) is exactly where the two 16-row logit tiles of the block leave their weights in the
// accumulator registers, so w never leaves registers here either.
// MFMA cycles per (32 streamed x 16 stationary) block: 6 + 4 = 10 x 16 against 2 x (8 + 8) x 32 in the f32 kernel: the kernel
// is bound by the exp / weight arithmetic on the VALU instead.
#include "disc_mfma.h"

#include <type_traits>

namespace fh {

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int kD = 32, kYT = 64;

__device__ __forceinline__ void split2(float v, u16& hi, u16& lo) {
  hi = f2bf(v);
  lo = f2bf(v - bf2f(hi));
}
// 8 consecutive f32 -> the hi and lo bf16x8 fragments
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  union {
    u16 h[8];
    bf16x8 v;
  } a, b;
#pragma unroll
  for (int k = 0; k < 8; ++k) split2(v[k], a.h[k], b.h[k]);
  hi = a.v;
  lo = b.v;
}
// byte offset of 16-byte chunk ch (0..3) of row `row` of a [64][32] bf16 image (64-byte rows)
__device__ __forceinline__ int img_off(int row, int ch) { return row * 64 + ((ch ^ ((row >> 2) & 3)) << 4); }

// 8 k-values (rows 4g..4g+3 and 16+4g..16+4g+3 of the 32-row block at row0) of column d = 16*dj + i
__device__ __forceinline__ bf16x8 frag_t(const char* img, int row0, int dj, int g, int i) {
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const int ra = row0 + 4 * g + (i >> 2);
  const int cb = 32 * dj + 8 * (i & 3);  // byte offset of the 4 columns inside the row
  union {
    struct {
      s16x4 lo, hi;
    } s;
    bf16x8 v;
  } u;
  u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + img_off(ra, cb >> 4) + (cb & 15)));
  u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + img_off(ra + 16, cb >> 4) + (cb & 15)));
  return u.v;
}

// XQ: the stationary set is the queries (forward, dq); else the table rows (dtable).
// MODE 2 (XQ only) = MODE 1 plus the STREAMED side's gradient from the same weights: both backward products from one
// recomputation of the logits and one exp per pair.  G2[y][d] += 2c sum_x w[y,x] X[x][d] needs w with x on the contraction
// index, i.e. transposed: each wave leaves its bf16 weights of a 32-row block in a private [x][y] LDS image (ds_write_b64: the 4
// accumulator registers of a tile are 4 consecutive y) and reads them back as A fragments with ds_read_b64_tr_b16 (frag_t:
// the image is the [k][m] layout that read expects).  B = the wave's own 64 stationary vectors as [k = x][n = d] fragments in
// the k order of frag_t (hi / lo split), kept in registers; a third B operand of ones gives WY[y] = sum_x w[y,x] from the same
// rounded weights.  Each wave leaves its partial tiles in an LDS slot of its own; the four slots are summed after each 64-row tile.
// No global atomics in this mode (the chip retires ~70 G f32 atomics/s: (B/256) S D of them would cost more than the second
// pass they replace): both sides leave PARTIALS with plain stores -- G[chunk][x][d] and G2[x-tile][y][d], WY[x-tile][y] -- and
// two small kernels (disc_mfma.hip) reduce them: dq = sum over chunks; dtable += 2c (sum G2 - t_y sum WY).
constexpr int kDtLd = 36;  // row stride of the LDS accumulator (floats): rows 4 apart fall into different banks
template <int MODE, bool XQ>
__global__ __launch_bounds__(256, 2) void disc_lp_kernel(DiscMfmaArgs a) {
  static_assert(MODE != 2 || XQ, "the one-pass backward keeps the queries stationary");
  constexpr bool BW = MODE >= 1;     // backward: weights + the stationary side's product
  constexpr bool BOTH = MODE == 2;   // ... and the streamed side's
  __shared__ __attribute__((aligned(16))) char wimg[BOTH ? 4 : 1][BOTH ? 64 * 64 : 16];  // per wave: [x = 64][y = 32] bf16
  __shared__ float wy_lds[BOTH ? 4 : 1][BOTH ? kYT : 1];  // per wave
  __shared__ __attribute__((aligned(16))) char yhi[kYT * kD * 2];
  __shared__ __attribute__((aligned(16))) char ylo[kYT * kD * 2];
  __shared__ __attribute__((aligned(16))) float yn[kYT];
  __shared__ float ymax[kYT], yinv[kYT];
  __shared__ int ytgt[kYT];
  __shared__ int yown[kYT / 32][4];  // !XQ: wave w staged a query of block b whose target is one of this workgroup's rows
  // backward epilogue: the transpose buffer tr[256][D + 1]; MODE 2, inside the loop: one [64 y][kDtLd] slot per wave for its
  // partial of the streamed side's product (plain writes; LDS float atomics from 8 waves cost more than the second pass did)
  constexpr int kRed = !BW ? 1 : (BOTH && 4 * kYT * kDtLd > 256 * (kD + 1) ? 4 * kYT * kDtLd : 256 * (kD + 1));
  __shared__ __attribute__((aligned(16))) float red[kRed];
  float (*tr)[kD + 1] = (float (*)[kD + 1]) red;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, i = lane & 15;
  const int x0 = blockIdx.y * 256 + wave * 64;
  const int y_begin = blockIdx.x * a.chunk;
  const int y_end = min(a.NY, y_begin + a.chunk);
  const float gscale = BW ? (*a.gsc) * a.gmul : 0.f;

  // ---- stationary fragments (B operand of the logit product): lane (g,i) of tile t holds X[x0+16t+i][8g .. 8g+7], split
  bf16x8 xh[4], xl[4];
  float xn[4], xmax[4], xinv[4];
  int xtgt[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int x = x0 + t * 16 + i;
    const bool ok = x < a.NX;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = 0.f;
    if (ok) {
      const float4 u0 = *(const float4*)(a.X + (int64_t)x * kD + 8 * g), u1 = *(const float4*)(a.X + (int64_t)x * kD + 8 * g + 4);
      v[0] = u0.x, v[1] = u0.y, v[2] = u0.z, v[3] = u0.w, v[4] = u1.x, v[5] = u1.y, v[6] = u1.z, v[7] = u1.w;
    }
    float nrm = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) nrm = fmaf(v[k], v[k], nrm);
    nrm += __shfl_xor(nrm, 16, 64);
    nrm += __shfl_xor(nrm, 32, 64);
    xn[t] = nrm;
    split8(v, xh[t], xl[t]);
    xmax[t] = 0.f;
    xinv[t] = 0.f;
    xtgt[t] = -1;
    if (XQ) {
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

  // MODE 2: B fragments of the streamed side's product: k = x in frag_t's order (rows 4g..4g+3 and 16+4g..16+4g+3 of the
  // 32-vector block kb), n = d = 16 dj + i
  bf16x8 xbh[BOTH ? 2 : 1][2], xbl[BOTH ? 2 : 1][2], ones;
  if constexpr (BOTH) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int dj = 0; dj < 2; ++dj) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int x = x0 + 32 * kb + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
          v[j] = x < a.NX ? a.X[(int64_t)x * kD + 16 * dj + i] : 0.f;
        }
        split8(v, xbh[kb][dj], xbl[kb][dj]);
      }
    union {
      u16 h[8];
      bf16x8 v;
    } o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.h[j] = 0x3F80;  // bf16(1.0)
    ones = o.v;
  }

  float m[4], ssum[4], wsum[4];
  f32x4 gacc[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    m[t] = -INFINITY;
    ssum[t] = 0.f;
    wsum[t] = 0.f;
    gacc[t][0] = gacc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- stream Y in tiles of 64 vectors: thread -> (row = id / 8, 4 floats id % 8), two ids per thread
  float4 st[2];
  auto issue = [&](int y0) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int id = tid + p * 256;
      const int y = y0 + (id >> 3);
      st[p] = (y < y_end) ? *(const float4*)(a.Y + (int64_t)y * kD + (id & 7) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (y_begin < y_end) issue(y_begin);
  for (int y0 = y_begin; y0 < y_end; y0 += kYT) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int id = tid + p * 256;
      const int row = id >> 3, c4 = id & 7;
      const float v[4] = {st[p].x, st[p].y, st[p].z, st[p].w};
      u16 h[4], l[4];
      float nrm = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        split2(v[k], h[k], l[k]);
        nrm = fmaf(v[k], v[k], nrm);
      }
      const int off = img_off(row, c4 >> 1) + (c4 & 1) * 8;
      *(uint2*)(yhi + off) = uint2{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16)};
      *(uint2*)(ylo + off) = uint2{(uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16)};
      // the row's 8 threads are 8 consecutive lanes
      nrm += __shfl_xor(nrm, 1, 64);
      nrm += __shfl_xor(nrm, 2, 64);
      nrm += __shfl_xor(nrm, 4, 64);
      bool mine = false;  // (!XQ) this query's own row is among the workgroup's stationary rows
      if (c4 == 0) {
        yn[row] = nrm;
        if (!XQ) {  // streamed queries: their (max, scale/sum, target)
          const int y = y0 + row;
          const bool ok = y < y_end;
          ymax[row] = ok && MODE == 1 ? a.rmax[y] : 0.f;
          yinv[row] = ok && MODE == 1 ? gscale / a.rsum[y] : 0.f;
          int tg = -3;
          if (ok) {
            const int64_t vv = a.idx[y] - a.row0;
            tg = (vv >= 0 && vv < a.NX) ? (int)vv : -3;
          }
          ytgt[row] = tg;
          mine = tg >= (int)blockIdx.y * 256 && tg < (int)blockIdx.y * 256 + 256;
        }
      }
      if (!XQ) {  // pass p stages the 32 rows of block p, 8 per wave
        const bool any = __any(mine);
        if (lane == 0) yown[p][wave] = any ? 1 : 0;
      }
    }
    __syncthreads();
    if (y0 + kYT < y_end) issue(y0 + kYT);

#pragma unroll 1
    for (int yp = 0; yp < kYT / 32; ++yp) {  // 32-row blocks: two 16-row logit tiles each
      if (y0 + yp * 32 >= y_end) break;
      // A fragments of the logit product: Y[32 yp + 16 h + i][8g .. 8g+7], hi and lo
      bf16x8 ah[2], al[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = yp * 32 + h * 16 + i;
        ah[h] = __builtin_bit_cast(bf16x8, *(const uint4*)(yhi + img_off(row, g)));
        al[h] = __builtin_bit_cast(bf16x8, *(const uint4*)(ylo + img_off(row, g)));
      }
      float ynr[2][4], ymx[2][4], yiv[2][4];
      int ytg[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 ynv = *(const float4*)(yn + yp * 32 + h * 16 + 4 * g);
        ynr[h][0] = ynv.x, ynr[h][1] = ynv.y, ynr[h][2] = ynv.z, ynr[h][3] = ynv.w;
        if (MODE == 1 && !XQ) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            ymx[h][r] = ymax[yp * 32 + h * 16 + 4 * g + r];
            yiv[h][r] = yinv[yp * 32 + h * 16 + 4 * g + r];
            ytg[h][r] = ytgt[yp * 32 + h * 16 + 4 * g + r];
          }
        }
      }
      // A fragments of the second product: Y^T (d on the rows), k = the block's 32 streamed rows
      bf16x8 th[2], tl[2];
      if constexpr (BW) {
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
          th[dj] = frag_t(yhi, yp * 32, dj, g, i);
          tl[dj] = frag_t(ylo, yp * 32, dj, g, i);
        }
      }
      // Interior blocks with no (query, own row) pair -- all but ~2 % of them -- take the unmasked body: validity and own-row
      // selects were a third of the loop's VALU work (58 v_cndmask + 59 compares + 64 s_and per 32-row block), and the loop is
      // VALU-bound.  The test is wave-uniform: the block and this wave's 64 stationary vectors are whole, and none of the
      // vectors' targets falls into the block (XQ) / no streamed query of the block has its target among the workgroup's rows.
      const int yb = y0 + yp * 32;
      const bool whole = yb + 32 <= y_end && x0 + 64 <= a.NX;
      bool own_blk = false;
      if (!XQ) own_blk = (yown[yp][0] | yown[yp][1] | yown[yp][2] | yown[yp][3]) != 0;
      const float c2 = 2.f * a.c;
      auto tile = [&](int t, auto masked_c) {
        constexpr bool MASKED = decltype(masked_c)::value;
        const bool xok = x0 + t * 16 + i < a.NX;
        const float cxn = a.c * xn[t];
        float w8[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[h], xh[t], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[h], xl[t], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[h], xh[t], acc, 0, 0, 0);
          float lg[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            lg[r] = c2 * acc[r] - (a.c * ynr[h][r] + cxn);
            if constexpr (MASKED) {
              const int y = yb + h * 16 + 4 * g + r;
              if (!(xok && y < y_end)) lg[r] = -INFINITY;
              // the query's own row is handled exactly by the callers (disc_mfma.hip)
              const bool own = (MODE == 1 && !XQ) ? ytg[h][r] == xtgt[t] : xtgt[t] == y;
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
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float p;
              if (XQ)
                p = __expf(lg[r] - xmax[t]) * xinv[t];
              else
                p = __expf(lg[r] - ymx[h][r]) * yiv[h][r];
              w8[h * 4 + r] = (!MASKED || lg[r] > -INFINITY) ? p : 0.f;
            }
          }
        }
        if constexpr (BW) {
          // w in bf16 (k-slot 8g + j <-> row 4g + j of the first, 16 + 4g + (j - 4) of the second tile: the order of frag_t);
          // the weight sum takes the ROUNDED values (see the header)
          union {
            u16 h[8];
            bf16x8 v;
          } wb;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            wb.h[k] = f2bf(w8[k]);
            wsum[t] += bf2f(wb.h[k]);
          }
#pragma unroll
          for (int dj = 0; dj < 2; ++dj) {
            gacc[t][dj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th[dj], wb.v, gacc[t][dj], 0, 0, 0);
            gacc[t][dj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tl[dj], wb.v, gacc[t][dj], 0, 0, 0);
          }
          if constexpr (BOTH) {  // row x = 16t + i of the wave's [x][y] image: y = 16h + 4g .. + 3 are the 4 registers of tile h
            char* wi = wimg[wave];
            const int row = 16 * t + i;
#pragma unroll
            for (int h = 0; h < 2; ++h)
              *(uint2*)(wi + img_off(row, 2 * h + (g >> 1)) + (g & 1) * 8) =
                  uint2{(uint32_t)wb.h[4 * h] | ((uint32_t)wb.h[4 * h + 1] << 16), (uint32_t)wb.h[4 * h + 2] | ((uint32_t)wb.h[4 * h + 3] << 16)};
          }
        }
      };
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        bool masked = !whole || own_blk;
        if (XQ) masked = masked || __any((unsigned)(xtgt[t] - yb) < 32u);
        if (masked)
          tile(t, std::true_type{});
        else
          tile(t, std::false_type{});
      }
      if constexpr (BOTH) {
        // the streamed side's product over the wave's 64 stationary vectors (LDS operations of one wave complete in order: the
        // reads below see the writes of the four tiles above)
        // (a compiler-only ordering: a fence instruction would also wait for the prefetch of the next tile)
        asm volatile("" ::: "memory");
        const char* wi = wimg[wave];
        f32x4 oacc[2][2], wya[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          oacc[h][0] = oacc[h][1] = wya[h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            const bf16x8 wa = frag_t(wi, 32 * kb, h, g, i);  // A[m = y = 16h + i][k = x of block kb]
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) {
              oacc[h][dj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xbh[kb][dj], oacc[h][dj], 0, 0, 0);
              oacc[h][dj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xbl[kb][dj], oacc[h][dj], 0, 0, 0);
            }
            wya[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, ones, wya[h], 0, 0, 0);
          }
        }
        asm volatile("" ::: "memory");  // (the next block's writes stay behind these reads)
        // lane holds out[y = 16h + 4g + r][d = 16dj + i] -> the wave's slot; WY is in every column: lane i == 0 stores it
        float* slot = red + wave * (kYT * kDtLd);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int yr = yp * 32 + 16 * h + 4 * g + r;
            slot[yr * kDtLd + i] = oacc[h][0][r];
            slot[yr * kDtLd + 16 + i] = oacc[h][1][r];
            if (i == 0) wy_lds[wave][yr] = wya[h][r];
          }
      }
    }
    __syncthreads();
    if constexpr (BOTH) {  // the tile's four slots summed: plain stores into this x-tile's slice of the partial buffer (every
                           // (x-tile, y) is written exactly once); the next tile's slot writes are behind its staging barrier
      for (int e = tid; e < kYT * kD; e += 256) {
        const int row = e / kD, d = e % kD, o = row * kDtLd + d;
        if (y0 + row < y_end)
          a.G2[((int64_t)blockIdx.y * a.NY + y0 + row) * kD + d] =
              (red[o] + red[kYT * kDtLd + o]) + (red[2 * kYT * kDtLd + o] + red[3 * kYT * kDtLd + o]);
      }
      if (tid < kYT && y0 + tid < y_end)
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
      for (int dj = 0; dj < 2; ++dj) {
        const int x = x0 + t * 16 + i;
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x < a.NX) xv = *(const float4*)(a.X + (int64_t)x * kD + dj * 16 + 4 * g);
        const float xr[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) tr[wave * 64 + t * 16 + i][dj * 16 + 4 * g + r] = 2.f * a.c * (gacc[t][dj][r] - xr[r] * ws);
      }
    }
    __syncthreads();
    for (int e = tid; e < 256 * kD; e += 256) {
      const int rr = e / kD, d = e % kD;
      const int x = blockIdx.y * 256 + rr;
      if (x < a.NX) {
        if constexpr (BOTH)
          a.G[((int64_t)blockIdx.x * a.NX + x) * kD + d] = tr[rr][d];  // this chunk's slice of the partial buffer
        else
          atomicAdd(a.G + (int64_t)x * kD + d, tr[rr][d]);
      }
    }
  }
}

}  // namespace

void disc_lp_launch(const DiscMfmaArgs& a, int mode, dim3 grid, hipStream_t st) {
  if (mode == 0)
    hipLaunchKernelGGL((disc_lp_kernel<0, true>), grid, dim3(256), 0, st, a);
  else if (mode == 2)
    hipLaunchKernelGGL((disc_lp_kernel<2, true>), grid, dim3(256), 0, st, a);
  else if (a.x_is_query)
    hipLaunchKernelGGL((disc_lp_kernel<1, true>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((disc_lp_kernel<1, false>), grid, dim3(256), 0, st, a);
}

}  // namespace fh
