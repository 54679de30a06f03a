// gemm_core.h -- the one MFMA contraction engine every GEMM-shaped kernel of the hot path uses
// (linear layers, Gaussian heads, LSTM step cells forward and backward, weight gradients).
//
//   acc[m][n] += sum over K-segments  sum_k  A(m,k) * B(n,k)
//
// Design for gfx950 (MI355X_MICROARCH.md / cdna_hip_programming.md):
//   * 256-thread workgroup = 4 waves of 64; each wave owns TM x TN tiles of 16x16 accumulators.
//     f32 operands -> v_mfma_f32_16x16x4_f32 (exact f32, the parity mode);
//     bf16 operands -> v_mfma_f32_16x16x32_bf16 (f32 accumulate).
//   * the contraction is consumed in PANELS of CH 16-byte chunks per row (CH = 8: 128-byte rows for
//     short contractions; CH = 32: 512-byte rows).  The step-cell GEMMs are latency-bound (one
//     workgroup per CU, everything L2-resident), so a panel issues all its global loads at once
//     (up to 16 x 16 B per thread in flight) and the next panel's loads are issued before the current
//     panel's MFMAs (register prefetch, guide T14): the load latency is paid once per panel, not
//     once per 32 k.
//   * "KC" operands (contraction index contiguous in memory) live in LDS as rows of CH chunks with the
//     chunk index XOR-swizzled by the row -> the ds_read_b128 fragment reads are bank-conflict-free
//     (guide T2).  One 16-byte chunk holds 4 consecutive k (f32: feeds 4 MFMA k-steps with the k order
//     permuted identically for A and B) or 8 consecutive k (bf16: one lane's share of a 16x16x32).
//   * "KM" operands (contraction index is the slow dimension: W in dh = dg . W, both operands of
//     dW = dg^T . h) are staged untransposed.  f32: [k][R+4] floats read with ds_read_b32 (the pad
//     puts the 4 k-rows of a 32-lane group on disjoint banks).  bf16: [k][R+16] read with
//     ds_read_b64_tr_b16 (hardware transpose, guide T10): lane group q takes k-rows 4q..4q+3 and
//     16+4q..16+4q+3 of each 32-k block, so a 32-lane half touches 8 consecutive rows, conflict-free
//     at the 160-byte row stride.  (That k order differs from the KC order, so in bf16 both operands
//     of a segment must be KC or both KM.)
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
  int K;       // contraction length (0 = segment unused)
  int a_rmod;  // > 0: A row m is read from physical row (m % a_rmod) (time-constant inputs broadcast over t)
};

template <typename T>
struct Op;
template <>
struct Op<float> {
  static constexpr int EPC = 4;  // elements per 16-byte chunk
  static constexpr int KMPAD = 4;
};
template <>
struct Op<u16> {  // bf16 bit patterns
  static constexpr int EPC = 8;
  static constexpr int KMPAD = 16;
};

struct RowIdent {
  int X;  // number of valid rows
  __device__ __forceinline__ int64_t operator()(int r) const { return r < X ? (int64_t)r : -1; }
  __device__ __forceinline__ bool all_valid(int r0, int n) const { return r0 + n <= X; }
};

// LDS bytes for one operand tile of R rows and CH chunks (max of the KC and KM images)
template <typename T, int R, int CH>
constexpr int tile_bytes() {
  constexpr int kc = R * CH * 16;
  constexpr int km = CH * Op<T>::EPC * (R + Op<T>::KMPAD) * (int)sizeof(T);
  return kc > km ? kc : km;
}

// global -> registers for one operand panel (R rows x CH chunks) starting at contraction index k0.
// Preconditions (checked on the host, see seg_fast_ok): base 16-byte aligned, ld % EPC == 0 and the chunked
// extent (K for KC images, the row count X for KM images) a multiple of EPC -> every 16-byte chunk is either
// fully inside or fully outside, so the loads are UNCONDITIONAL from a clamped address with the result
// zeroed by a select.  No per-chunk branches: all loads of the panel issue back to back (a branchy version
// made hipcc wait vmcnt(0) after every single load: 8 serial L2 round trips per panel).  Shapes that break
// the preconditions go to the scalar fallback kernel in gemm.hip.
template <typename T, int R, int CH, bool KC, class RowMap>
__device__ __forceinline__ void load_tile(uint4 (&v)[R * CH / kThreads], uint32_t& okbits, const void* base_, int64_t ld,
                                          int x0, int X, int k0, int K, const RowMap& rm, int rmod, int tid) {
  // okbits: bit p set <=> chunk p holds real data; the zeroing of the others is applied by store_tile, AFTER the
  // MFMAs of the current panel (masking here would consume the loads and stall right behind their issue)
  okbits = 0;
  constexpr int EPC = Op<T>::EPC;
  constexpr int NCH = R * CH / kThreads;
  const T* base = (const T*)base_;
  if constexpr (KC) {
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      const int idx = tid + p * kThreads;
      const int row = idx / CH, ch = idx % CH;
      const int64_t grow0 = rm(x0 + row);
      const int k = k0 + ch * EPC;
      const bool ok = grow0 >= 0 && k < K;
      uint32_t g32 = ok ? (uint32_t)grow0 : 0u;  // physical rows are < 2^31 (host-checked)
      if (rmod > 0) g32 %= (uint32_t)rmod;
      const int64_t off = ok ? (int64_t)g32 * ld + k : 0;
      v[p] = *(const uint4*)(base + off);
      okbits |= ok ? (1u << p) : 0u;
    }
  } else {
    // KM image: CH*EPC k-rows of R elements
    constexpr int CPR = R / EPC;  // 16-byte chunks per k-row
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      const int idx = tid + p * kThreads;
      const int krow = idx / CPR, cx = idx % CPR;
      const int k = k0 + krow, x = x0 + cx * EPC;
      const bool ok = k < K && x < X;
      const int64_t off = ok ? (int64_t)k * ld + x : 0;
      v[p] = *(const uint4*)(base + off);
      okbits |= ok ? (1u << p) : 0u;
    }
  }
}

template <int CH>
__device__ __forceinline__ int kc_off(int row, int ch) {
  constexpr int SW = CH >= 16 ? 15 : CH - 1;
  return row * (CH * 16) + ((ch ^ (row & SW)) << 4);
}

__device__ __forceinline__ uint4 mask4(uint4 t, uint32_t okbits, int p) {
  const uint32_t msk = 0u - ((okbits >> p) & 1u);  // all ones / all zeros (a select between objects made hipcc spill)
  t.x &= msk;
  t.y &= msk;
  t.z &= msk;
  t.w &= msk;
  return t;
}

template <typename T, int R, int CH, bool KC>
__device__ __forceinline__ void store_tile(const uint4 (&v)[R * CH / kThreads], uint32_t okbits, char* lds, int tid) {
  constexpr int EPC = Op<T>::EPC;
  constexpr int NCH = R * CH / kThreads;
  if constexpr (KC) {
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      int idx = tid + p * kThreads;
      *(uint4*)(lds + kc_off<CH>(idx / CH, idx % CH)) = mask4(v[p], okbits, p);
    }
  } else {
    constexpr int CPR = R / EPC;
    constexpr int LD = R + Op<T>::KMPAD;
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      int idx = tid + p * kThreads;
      int krow = idx / CPR, cx = idx % CPR;
      *(uint4*)(lds + (krow * LD + cx * EPC) * (int)sizeof(T)) = mask4(v[p], okbits, p);
    }
  }
}

template <typename T, int BM, int BN, int WM, int WN, int CH>
struct Tile {
  static constexpr int TM = BM / WM / 16;
  static constexpr int TN = BN / WN / 16;
  static constexpr int BK = CH * Op<T>::EPC;  // contraction elements per panel
  static constexpr int A_BYTES = tile_bytes<T, BM, CH>();
  static constexpr int B_BYTES = tile_bytes<T, BN, CH>();
  static constexpr int SMEM = A_BYTES + B_BYTES;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "tile/wave shape");
  static_assert((BM * CH) % kThreads == 0 && (BN * CH) % kThreads == 0, "staging shape");
  static_assert(CH == 8 || CH == 16 || CH == 32, "panel width");
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

// bf16 KM fragment: 8 k-values of column `col` for lane group q of 32-k block j (two transposed reads)
template <int LD>
__device__ __forceinline__ bf16x8 frag_km_bf16(const char* lds, int col0, int j, int lane) {
  const int q = lane >> 4, i = lane & 15;
  const int qq = i >> 2, p = i & 3;
  const int ka = 32 * j + 4 * q;
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const char* a0 = lds + ((ka + qq) * LD + col0 + 4 * p) * 2;
  const char* a1 = a0 + 16 * LD * 2;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a1));
  union {
    struct {
      s16x4 lo, hi;
    } s;
    bf16x8 v;
  } u;
  u.s.lo = lo;
  u.s.hi = hi;
  return u.v;
}

// host-side precondition of the branch-free staging for one segment
template <typename T>
__host__ inline bool seg_fast_ok(const Seg& s, int M, int N) {
  constexpr int EPC = Op<T>::EPC;
  if (s.K <= 0) return true;
  auto ok = [&](const void* p, int64_t ld, int ext) { return (((uintptr_t)p) & 15) == 0 && (ld % EPC) == 0 && (ext % EPC) == 0; };
  return ok(s.A, s.lda, s.a_kc ? s.K : M) && ok(s.B, s.ldb, s.b_kc ? s.K : N);
}

// The main loop.  Panels of both segments are numbered consecutively; [it_begin, it_end) selects a
// sub-range (split-K).  smem must hold Tile::SMEM bytes (16-byte aligned).  AKC/BKC: operand orientation
// (compile time; both segments share it).  bf16 supports KC/KC and KM/KM only.
// SWAP: the B fragment is the first MFMA operand, so a lane holds 4 consecutive COLUMNS (n) of one row (m = lane & 15) instead
// of 4 consecutive rows of one column: row-major outputs then leave as 16-byte stores (gemm_tile's epilogue)
template <typename T, int BM, int BN, int WM, int WN, int CH, bool AKC, bool BKC, bool SWAP, class ARowMap, class BRowMap>
__device__ __forceinline__ void mainloop(f32x4 (&acc)[BM / WM / 16][BN / WN / 16], const Seg (&segs)[2], int m0, int M,
                                         int n0, int N, const ARowMap& arm, const BRowMap& brm, int it_begin,
                                         int it_end, char* smem) {
  using TL = Tile<T, BM, BN, WM, WN, CH>;
  constexpr int TM = TL::TM, TN = TL::TN;
  constexpr int BK = TL::BK;
  constexpr int EPC = Op<T>::EPC;
  constexpr int KSTEP = 4 * EPC;  // contraction elements per j-block (16 for f32, 32 for bf16)
  static_assert(sizeof(T) == 4 || AKC == BKC, "bf16: KC/KC or KM/KM");
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 15, q = lane >> 4;
  char* As = smem;
  char* Bs = smem + TL::A_BYTES;

  const int nkb0 = (segs[0].K + BK - 1) / BK;

  uint4 ra[BM * CH / kThreads], rb[BN * CH / kThreads];
  uint32_t oka = 0, okb = 0;

  auto seg_of = [&](int it, int& k0) -> int {
    if (it < nkb0) {
      k0 = it * BK;
      return 0;
    }
    k0 = (it - nkb0) * BK;
    return 1;
  };
  auto issue = [&](int it) {
    int k0;
    const Seg& s = segs[seg_of(it, k0)];
    load_tile<T, BM, CH, AKC>(ra, oka, s.A, s.lda, m0, M, k0, s.K, arm, s.a_rmod, tid);
    load_tile<T, BN, CH, BKC>(rb, okb, s.B, s.ldb, n0, N, k0, s.K, brm, 0, tid);
  };

  if (it_begin < it_end) issue(it_begin);
  for (int it = it_begin; it < it_end; ++it) {
    int k0;
    const int si = seg_of(it, k0);
    const int kleft = segs[si].K - k0;
    const int nj = kleft >= BK ? CH / 4 : (kleft + KSTEP - 1) / KSTEP;  // j-blocks that hold data
    store_tile<T, BM, CH, AKC>(ra, oka, As, tid);
    store_tile<T, BN, CH, BKC>(rb, okb, Bs, tid);
    __syncthreads();
    if (it + 1 < it_end) issue(it + 1);  // next panel's loads fly under this panel's MFMAs
#pragma unroll 2
    for (int j = 0; j < nj; ++j) {
      if constexpr (sizeof(T) == 4) {
        float a[TM][4], b[TN][4];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const int row = wm * (TM * 16) + tm * 16 + r;
          if constexpr (AKC) {
            uint4 u = *(const uint4*)(As + kc_off<CH>(row, (j << 2) | q));
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
          if constexpr (BKC) {
            uint4 u = *(const uint4*)(Bs + kc_off<CH>(row, (j << 2) | q));
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
              acc[tm][tn] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x4f32(b[tn][s], a[tm][s], acc[tm][tn], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm][s], b[tn][s], acc[tm][tn], 0, 0, 0);
      } else {
        bf16x8 a[TM], b[TN];
        if constexpr (AKC) {
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) {
            uint4 u = *(const uint4*)(As + kc_off<CH>(wm * (TM * 16) + tm * 16 + r, (j << 2) | q));
            a[tm] = __builtin_bit_cast(bf16x8, u);
          }
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            uint4 u = *(const uint4*)(Bs + kc_off<CH>(wn * (TN * 16) + tn * 16 + r, (j << 2) | q));
            b[tn] = __builtin_bit_cast(bf16x8, u);
          }
        } else {  // both KM: hardware-transposed reads
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) a[tm] = frag_km_bf16<BM + 16>(As, wm * (TM * 16) + tm * 16, j, lane);
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) b[tn] = frag_km_bf16<BN + 16>(Bs, wn * (TN * 16) + tn * 16, j, lane);
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[tn], a[tm], acc[tm][tn], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Direct-to-LDS staging (LDS-DMA, global_load_lds_dwordx4) for INTERIOR KC/KC tiles: no staging registers, the
// next panel lands in the other LDS buffer while this one is multiplied, one barrier per panel instead of two.
// The LDS destination of a wave-instruction is linear (wave-uniform base + lane*16 = 64/CH whole rows), so the XOR
// swizzle is applied to the per-lane SOURCE chunk (guide rule 21): LDS position `pos` of row r receives global chunk
// pos ^ (r & SW), the same involution the fragment reads use.
// ---------------------------------------------------------------------------------------------
// AUX: cache-policy bits of the load (16 = sc1: served by L2, bypasses this CU's vector L1).
template <typename T, int R, int CH, int AUX = 0, class RowMap>
__device__ __forceinline__ void glds_tile(char* lds_buf, const void* base_, int64_t ld, int x0, int k0, const RowMap& rm,
                                          int rmod, int tid) {
  constexpr int EPC = Op<T>::EPC;
  constexpr int RPI = 64 / CH;        // rows written by one wave-instruction
  constexpr int NI = R / (4 * RPI);   // instructions per wave
  constexpr int SW = CH >= 16 ? 15 : CH - 1;
  static_assert(R % (4 * RPI) == 0, "glds tile shape");
  const int lane = tid & 63, wave = tid >> 6;
  const T* base = (const T*)base_;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = (wave * NI + j) * RPI + lane / CH;
    const int c = (lane % CH) ^ (row & SW);
    uint32_t g32 = (uint32_t)rm(x0 + row);  // interior tile: every row valid
    if (rmod > 0) g32 %= (uint32_t)rmod;
    const T* src = base + (int64_t)g32 * ld + k0 + c * EPC;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)(lds_buf + (wave * NI + j) * 1024), 16, 0, AUX);
  }
}

template <typename T>
__device__ __forceinline__ bool seg_glds_ok(const Seg& s, int BK) {
  constexpr int EPC = Op<T>::EPC;
  return s.K == 0 || (s.a_kc && s.b_kc && (s.K % BK) == 0 && (s.lda % EPC) == 0 && (s.ldb % EPC) == 0 &&
                      ((((uintptr_t)s.A) | ((uintptr_t)s.B)) & 15) == 0);
}

// NBUF = 2: the next panel lands in the other buffer under this panel's MFMAs (one barrier per panel; few, large
// workgroups).  NBUF = 1: one buffer, two barriers per panel, half the LDS: more workgroups per CU hide the latency
// instead (large batches).
template <typename T, int BM, int BN, int WM, int WN, int CH, int NBUF = 2>
struct GldsTile {
  static constexpr int A_BYTES = BM * CH * 16, B_BYTES = BN * CH * 16;
  static constexpr int SMEM = NBUF * (A_BYTES + B_BYTES);
};

// whole contraction (no split-K) of an interior KC/KC tile
template <typename T, int BM, int BN, int WM, int WN, int CH, int NBUF, bool SWAP, class ARowMap, class BRowMap>
__device__ __forceinline__ void mainloop_glds(f32x4 (&acc)[BM / WM / 16][BN / WN / 16], const Seg (&segs)[2], int m0, int n0,
                                              const ARowMap& arm, const BRowMap& brm, char* smem) {
  using GT = GldsTile<T, BM, BN, WM, WN, CH, NBUF>;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int BK = CH * Op<T>::EPC;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 15, q = lane >> 4;
  const int nkb0 = segs[0].K / BK, nkb = nkb0 + segs[1].K / BK;
  auto issue = [&](int p) {
    const int si = p < nkb0 ? 0 : 1;
    const int k0 = (p < nkb0 ? p : p - nkb0) * BK;
    const Seg& s = segs[si];
    char* buf = smem + (NBUF == 2 ? (p & 1) : 0) * (GT::A_BYTES + GT::B_BYTES);
    glds_tile<T, BM, CH>(buf, s.A, s.lda, m0, k0, arm, s.a_rmod, tid);
    glds_tile<T, BN, CH>(buf + GT::A_BYTES, s.B, s.ldb, n0, k0, brm, 0, tid);
  };
  if (nkb > 0) issue(0);
  for (int p = 0; p < nkb; ++p) {
    __syncthreads();  // hipcc drains the LDS-DMA (vmcnt(0)) in front of the barrier: panel p has landed, and every
                      // wave is done with the buffer panel p+1 is about to overwrite
    if (NBUF == 2 && p + 1 < nkb) issue(p + 1);
    const char* As = smem + (NBUF == 2 ? (p & 1) : 0) * (GT::A_BYTES + GT::B_BYTES);
    const char* Bs = As + GT::A_BYTES;
#pragma unroll 2
    for (int j = 0; j < CH / 4; ++j) {
      if constexpr (sizeof(T) == 4) {
        float a[TM][4], b[TN][4];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const uint4 u = *(const uint4*)(As + kc_off<CH>(wm * (TM * 16) + tm * 16 + r, (j << 2) | q));
          a[tm][0] = __uint_as_float(u.x), a[tm][1] = __uint_as_float(u.y), a[tm][2] = __uint_as_float(u.z), a[tm][3] = __uint_as_float(u.w);
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const uint4 u = *(const uint4*)(Bs + kc_off<CH>(wn * (TN * 16) + tn * 16 + r, (j << 2) | q));
          b[tn][0] = __uint_as_float(u.x), b[tn][1] = __uint_as_float(u.y), b[tn][2] = __uint_as_float(u.z), b[tn][3] = __uint_as_float(u.w);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
              acc[tm][tn] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x4f32(b[tn][s], a[tm][s], acc[tm][tn], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm][s], b[tn][s], acc[tm][tn], 0, 0, 0);
      } else {
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          a[tm] = __builtin_bit_cast(bf16x8, *(const uint4*)(As + kc_off<CH>(wm * (TM * 16) + tm * 16 + r, (j << 2) | q)));
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          b[tn] = __builtin_bit_cast(bf16x8, *(const uint4*)(Bs + kc_off<CH>(wn * (TN * 16) + tn * 16 + r, (j << 2) | q)));
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[tn], a[tm], acc[tm][tn], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
      }
    }
    if (NBUF == 1 && p + 1 < nkb) {
      __syncthreads();  // every wave has read the single buffer
      issue(p + 1);
    }
  }
  __syncthreads();  // the epilogue may reuse LDS / the kernel may fall through to another phase
}

template <int TM, int TN>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[TM][TN]) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// total number of panels of a segment pair
template <typename T, int CH>
__host__ __device__ inline int num_kblocks(const Seg (&segs)[2]) {
  constexpr int BK = CH * Op<T>::EPC;
  return (segs[0].K + BK - 1) / BK + (segs[1].K + BK - 1) / BK;
}

}  // namespace fh
