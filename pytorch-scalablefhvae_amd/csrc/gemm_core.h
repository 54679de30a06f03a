// gemm_core.h -- the one MFMA contraction engine every GEMM-shaped kernel of the hot path uses
// (linear layers, Gaussian heads, LSTM step cells forward and backward, weight gradients).
//
//   acc[m][n] += sum over K-segments  sum_k  A(m,k) * B(n,k)
//
// Design for gfx950 (MI355X_MICROARCH.md / cdna_hip_programming.md):
//   * 256-thread workgroup = 4 waves of 64; each wave owns TM x TN tiles of 16x16 accumulators.
//     f32 operands -> v_mfma_f32_16x16x4_f32 (exact f32, the parity mode);
//     bf16 operands -> v_mfma_f32_16x16x32_bf16 (f32 accumulate).
//   * operand tiles are staged global -> registers -> LDS with 16-byte accesses; the next K-block's
//     global loads are issued before the current block's MFMAs (register prefetch, guide T14).
//   * "KC" operands (contraction index contiguous in memory) live in LDS as 128-byte rows with the
//     16-byte chunk index XOR-swizzled by (row & 7) -> the ds_read_b128 fragment reads are
//     bank-conflict-free (guide T2).  One 16-byte chunk holds 4 consecutive k (f32) and feeds 4
//     MFMA k-steps with the k order permuted identically for A and B (sum is order-free per step).
//   * "KM" operands (f32 only: contraction index is the slow dimension, e.g. W in dh = dg . W or both
//     operands of dW = dg^T . h) are staged untransposed as [32 k][R+4] floats and read with
//     ds_read_b32; the +4 row pad makes the 4 k-rows of a 32-lane group hit disjoint banks.
//   * up to two K-segments per tile let a cell consume [h^{l-1}_t ; h^l_{t-1}] without a concat.
#pragma once
#include "common.h"

namespace fh {

constexpr int kThreads = 256;

struct Seg {
  const void* A;
  int64_t lda;
  int a_kc;  // 1: A(m,k) = A[m*lda + k]   0: A(m,k) = A[k*lda + m]
  const void* B;
  int64_t ldb;
  int b_kc;  // 1: B(n,k) = B[n*ldb + k]   0: B(n,k) = B[k*ldb + n]
  int K;     // contraction length (0 = segment unused)
  int a_rmod;  // > 0: A row m is read from physical row (m % a_rmod) (time-constant inputs broadcast over t)
};

template <typename T>
struct Op;
template <>
struct Op<float> {
  static constexpr int EPC = 4;   // elements per 16-byte chunk
  static constexpr int BK = 32;   // k per staged block (128-byte KC rows)
};
template <>
struct Op<u16> {  // bf16 bit patterns
  static constexpr int EPC = 8;
  static constexpr int BK = 64;
};

struct RowIdent {
  int X;  // number of valid rows
  __device__ __forceinline__ int64_t operator()(int r) const { return r < X ? (int64_t)r : -1; }
};

// LDS bytes for one operand tile of R rows (max of the KC and KM images)
template <int R>
constexpr int tile_bytes() {
  return 32 * (R + 4) * 4;
}

template <typename T>
__device__ __forceinline__ uint4 load_elems(const T* p, int nvalid) {
  // slow path: up to EPC elements, zero filled
  constexpr int EPC = Op<T>::EPC;
  union {
    uint4 u;
    T e[EPC];
  } r;
  r.u = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < EPC; ++i)
    if (i < nvalid) r.e[i] = p[i];
  return r.u;
}

// global -> registers for one operand tile (R rows x BK k) of K-block starting at k0
template <typename T, int R, class RowMap>
__device__ __forceinline__ void load_tile(uint4 (&v)[R * 8 / kThreads], const void* base_, int64_t ld, int kc,
                                          bool vec_ok, int x0, int X, int k0, int K, const RowMap& rm, int rmod,
                                          int tid) {
  constexpr int EPC = Op<T>::EPC;
  constexpr int NCH = R * 8 / kThreads;
  const T* base = (const T*)base_;
  if (kc) {
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      int idx = tid + p * kThreads;
      int row = idx >> 3, ch = idx & 7;
      int64_t grow = rm(x0 + row);
      if (rmod > 0 && grow >= 0) grow %= rmod;
      int k = k0 + ch * EPC;
      if (grow >= 0 && k < K) {
        const T* src = base + grow * ld + k;
        if (vec_ok && k + EPC <= K)
          v[p] = *(const uint4*)src;
        else
          v[p] = load_elems<T>(src, K - k);
      } else {
        v[p] = make_uint4(0, 0, 0, 0);
      }
    }
  } else {
    // KM image (f32 only): 32 k-rows of R floats
    constexpr int CPR = R / 4;  // 16-byte chunks per k-row
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      int idx = tid + p * kThreads;
      int krow = idx / CPR, c4 = idx % CPR;
      int k = k0 + krow, x = x0 + c4 * 4;
      if (k < K && x < X) {
        const T* src = base + (int64_t)k * ld + x;
        if (vec_ok && x + 4 <= X)
          v[p] = *(const uint4*)src;
        else
          v[p] = load_elems<T>(src, X - x);
      } else {
        v[p] = make_uint4(0, 0, 0, 0);
      }
    }
  }
}

template <typename T, int R>
__device__ __forceinline__ void store_tile(const uint4 (&v)[R * 8 / kThreads], char* lds, int kc, int tid) {
  constexpr int NCH = R * 8 / kThreads;
  if (kc) {
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      int idx = tid + p * kThreads;
      int row = idx >> 3, ch = idx & 7;
      *(uint4*)(lds + row * 128 + ((ch ^ (row & 7)) << 4)) = v[p];
    }
  } else {
    constexpr int CPR = R / 4;
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      int idx = tid + p * kThreads;
      int krow = idx / CPR, c4 = idx % CPR;
      *(uint4*)(lds + (krow * (R + 4) + c4 * 4) * 4) = v[p];
    }
  }
}

// 16-byte fragment chunk (4j+q) of LDS row `row` of a KC image
__device__ __forceinline__ uint4 frag_kc(const char* lds, int row, int j, int q) {
  return *(const uint4*)(lds + row * 128 + ((((j << 2) | q) ^ (row & 7)) << 4));
}

template <typename T, int BM, int BN, int WM, int WN>
struct Tile {
  static constexpr int TM = BM / WM / 16;
  static constexpr int TN = BN / WN / 16;
  static constexpr int A_BYTES = tile_bytes<BM>();
  static constexpr int B_BYTES = tile_bytes<BN>();
  static constexpr int SMEM = A_BYTES + B_BYTES;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "tile/wave shape");
  static_assert((BM * 8) % kThreads == 0 && (BN * 8) % kThreads == 0, "staging shape");
};

// The main loop.  K-blocks of both segments are numbered consecutively; [it_begin, it_end) selects
// a sub-range (split-K).  smem must hold Tile::SMEM bytes (16-byte aligned).
template <typename T, int BM, int BN, int WM, int WN, class ARowMap, class BRowMap>
__device__ __forceinline__ void mainloop(f32x4 (&acc)[BM / WM / 16][BN / WN / 16], const Seg (&segs)[2], int m0, int M,
                                         int n0, int N, const ARowMap& arm, const BRowMap& brm, int it_begin,
                                         int it_end, char* smem) {
  using TL = Tile<T, BM, BN, WM, WN>;
  constexpr int TM = TL::TM, TN = TL::TN;
  constexpr int BK = Op<T>::BK;
  constexpr int EPC = Op<T>::EPC;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 15, q = lane >> 4;
  char* As = smem;
  char* Bs = smem + TL::A_BYTES;

  const int nkb0 = (segs[0].K + BK - 1) / BK;

  uint4 ra[BM * 8 / kThreads], rb[BN * 8 / kThreads];

  auto seg_of = [&](int it, int& k0) -> int {
    if (it < nkb0) {
      k0 = it * BK;
      return 0;
    }
    k0 = (it - nkb0) * BK;
    return 1;
  };
  auto vec_ok = [&](const void* p, int64_t ld, int kc) -> bool {
    return (((uintptr_t)p) & 15) == 0 && (ld % (kc ? EPC : 4)) == 0;
  };
  auto issue = [&](int it) {
    int k0;
    const Seg& s = segs[seg_of(it, k0)];
    load_tile<T, BM>(ra, s.A, s.lda, s.a_kc, vec_ok(s.A, s.lda, s.a_kc), m0, M, k0, s.K, arm, s.a_rmod, tid);
    load_tile<T, BN>(rb, s.B, s.ldb, s.b_kc, vec_ok(s.B, s.ldb, s.b_kc), n0, N, k0, s.K, brm, 0, tid);
  };

  if (it_begin < it_end) issue(it_begin);
  for (int it = it_begin; it < it_end; ++it) {
    int k0_unused;
    const int si = seg_of(it, k0_unused);
    const int a_kc = segs[si].a_kc, b_kc = segs[si].b_kc;
    store_tile<T, BM>(ra, As, a_kc, tid);
    store_tile<T, BN>(rb, Bs, b_kc, tid);
    __syncthreads();
    if (it + 1 < it_end) issue(it + 1);  // next block's loads fly under this block's MFMAs
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (sizeof(T) == 4) {
        float a[TM][4], b[TN][4];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const int row = wm * (TM * 16) + tm * 16 + r;
          if (a_kc) {
            uint4 u = frag_kc(As, row, j, q);
            a[tm][0] = __uint_as_float(u.x);
            a[tm][1] = __uint_as_float(u.y);
            a[tm][2] = __uint_as_float(u.z);
            a[tm][3] = __uint_as_float(u.w);
          } else {
            const float* Af = (const float*)As;
#pragma unroll
            for (int s = 0; s < 4; ++s) a[tm][s] = Af[(16 * j + 4 * q + s) * (BM + 4) + row];
          }
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int row = wn * (TN * 16) + tn * 16 + r;
          if (b_kc) {
            uint4 u = frag_kc(Bs, row, j, q);
            b[tn][0] = __uint_as_float(u.x);
            b[tn][1] = __uint_as_float(u.y);
            b[tn][2] = __uint_as_float(u.z);
            b[tn][3] = __uint_as_float(u.w);
          } else {
            const float* Bf = (const float*)Bs;
#pragma unroll
            for (int s = 0; s < 4; ++s) b[tn][s] = Bf[(16 * j + 4 * q + s) * (BN + 4) + row];
          }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm][s], b[tn][s], acc[tm][tn], 0, 0, 0);
      } else {
        // bf16: KC images only (callers guarantee); chunk (4j+q) = k 32j+8q .. +7 of this block
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          uint4 u = frag_kc(As, wm * (TM * 16) + tm * 16 + r, j, q);
          a[tm] = __builtin_bit_cast(bf16x8, u);
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          uint4 u = frag_kc(Bs, wn * (TN * 16) + tn * 16 + r, j, q);
          b[tn] = __builtin_bit_cast(bf16x8, u);
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
      }
    }
    __syncthreads();
  }
}

template <int TM, int TN>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[TM][TN]) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// total number of K-blocks of a segment pair
template <typename T>
__host__ __device__ inline int num_kblocks(const Seg (&segs)[2]) {
  constexpr int BK = Op<T>::BK;
  return (segs[0].K + BK - 1) / BK + (segs[1].K + BK - 1) / BK;
}

}  // namespace fh
