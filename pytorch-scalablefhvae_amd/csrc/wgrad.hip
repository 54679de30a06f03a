// wgrad.hip -- the weight gradients of the LSTM nets: C[M,N] += A[K,M]^T . B[K,N] with K = T*B (40,960 at the bench shape),
// M = 4H gate columns, N = H or I input columns; bf16 operands, f32 accumulation.  (No reference counterpart: the reference's
// model is FC and its autograd computes `dy^T x` with ATen, simple_fhvae.py:127-134.)
//
// Why a kernel of its own: both operands are K-MAJOR (the contraction index is the row of dgates / of the saved states), the
// outputs are small (1 MB) and the contraction is long, so the GEMM is bound by how many operand bytes a CU pulls from L2 per
// FLOP and by the split-K partial sums.  The generic engine (gemm_core.h: 4 waves, 128x64 tiles, register staging, padded LDS
// image) ran it at 445 TFLOP/s with 512 workgroups x 32 KB of f32 atomics per GEMM, one launch per weight matrix.  Here:
//   * 256 x 256 (or 256 x 128) output tile per 512-thread workgroup: 8 waves as 2 (m) x 4 (n), 128 x 64 per wave,
//     v_mfma_f32_16x16x32_bf16 -- a quarter of the L2 -> LDS bytes per FLOP of the 128x64 tiles;
//   * operands go global -> LDS by LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction = 2 k-rows of 256 columns), two
//     64-k stages; the loads of stage s+1 stay in flight under the MFMAs of stage s behind a COUNTED s_waitcnt vmcnt(n) and
//     raw s_barriers (guide: "Pipelining across barriers"); buffer range checking zero-fills the k rows past K, so any K works;
//   * the LDS image keeps the memory layout [k][column] (the DMA cannot transpose); fragments are read with
//     ds_read_b64_tr_b16 (hardware transpose, guide T10).  Rows are 512 B = two bank rows, so the 32-byte segment index of a
//     row is XOR-ed with (k & 7) -- applied to the per-lane SOURCE address of the DMA and to the read address (guide rule 21):
//     the 8 k-rows a 32-lane half touches land on 8 different 32-byte slots of the 256-byte bank row: conflict-free;
//   * ALL weight matrices of ALL nets of a step go out as ONE launch (fhvae_lstm_param_grads_multi): 36 tiles x split-K 7 = 252
//     workgroups at the bench shape instead of 12 launches x 512 workgroups; the f32 atomics (the chip adds ~1.3 TB/s) drop
//     from 12 x 16.8 MB to 63 MB per step.
#include "wgrad.h"

#include <algorithm>
#include <cstdlib>

#include "gemm_core.h"

namespace fh {

constexpr int kWgThreads = 512;
constexpr int kWgBM = 256, kWgBK = 64;

template <int W>  // operand tile width in elements
struct WgImg {
  static constexpr int RB = W * 2;              // bytes per k-row of the image
  static constexpr int BYTES = kWgBK * RB;      // one stage
  static constexpr int CPR = RB / 16;           // 16-byte chunks per row
  static constexpr int RPI = 1024 / RB;         // k-rows written by one wave-instruction
  static constexpr int NI = BYTES / 1024 / 8;   // wave-instructions per wave and stage
};

typedef void __attribute__((address_space(3))) * lds_void_p;

// per-lane byte offsets (relative to the stage's first k-row) of this wave's DMA pieces: row * ld + swizzled chunk
template <int W>
__device__ __forceinline__ void wg_dma_offsets(unsigned (&voff)[WgImg<W>::NI], unsigned ld_bytes, int wave, int lane) {
  using I = WgImg<W>;
#pragma unroll
  for (int q = 0; q < I::NI; ++q) {
    const int row = (wave * I::NI + q) * I::RPI + lane / I::CPR;
    const int pc = lane % I::CPR;                                  // physical 16-byte chunk of the LDS row this lane fills
    const int c = ((((pc >> 1) ^ (row & 7)) << 1) | (pc & 1));     // ... with the bytes of this logical chunk
    voff[q] = (unsigned)row * ld_bytes + (unsigned)c * 16u;
  }
}

template <int W>
__device__ __forceinline__ void wg_issue(char* stage, __amdgpu_buffer_rsrc_t rs, const unsigned (&voff)[WgImg<W>::NI], unsigned kbase,
                                         int wave) {
  using I = WgImg<W>;
#pragma unroll
  for (int q = 0; q < I::NI; ++q)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_p)(stage + (wave * I::NI + q) * 1024), 16, voff[q] + kbase, 0, 0, 0);
}

// 8 k-values of one column for the 16x16x32 operand: k rows 4g..4g+3 and 16+4g..16+4g+3 of 32-k block j (the same
// permutation of k for A and B)
template <int RB>
__device__ __forceinline__ bf16x8 wg_frag(const char* img, int off, int j) {
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const char* a0 = img + off + j * 32 * RB;
  union {
    struct {
      s16x4 lo, hi;
    } s;
    bf16x8 v;
  } u;
  u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0));
  u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0 + 16 * RB));
  return u.v;
}

// The DMA pieces of the next stage are issued in two halves behind each 32-k block's fragment reads (their issue cost then
// overlaps the LDS latency), plain vmcnt(0) at the top of the next step (they have had a whole step to land).  Measured at
// 4096^3 (64 steps per workgroup + 67 MB of epilogue): 159-164 us; all pieces at the top of a step behind a counted wait: 173 us;
// no DMA in the loop at all: 141 us; DMA only: 98 us -- the fragment-read + MFMA phases between the two barriers of a step bound
// the loop (~970 TFLOP/s with no DMA); a software pipeline over the (32-k block, m-tile) groups pinned with sched_group_barrier
// was no faster (173 us).
template <int BN>
__global__ __launch_bounds__(kWgThreads) void wgrad_kernel(WgGroup g) {
  using IA = WgImg<kWgBM>;
  using IB = WgImg<BN>;
  constexpr int STAGE = IA::BYTES + IB::BYTES;
  constexpr int TM = 8, TN = BN / 64;
  constexpr int NLOAD = IA::NI + IB::NI;  // DMA pieces per wave and stage
  // TWO LDS objects, one per stage, and the K loop written out for both: hipcc then knows (alias scopes of the two
  // variables) that the fragment reads of one stage cannot alias the DMA in flight into the other and emits a COUNTED
  // s_waitcnt vmcnt(n) in front of them; with one array (or a runtime stage index) it drains every LDS-DMA (vmcnt(0)) before
  // the first ds_read of each step and the loads never overlap the MFMAs
  __shared__ __attribute__((aligned(1024))) char stage0[STAGE];
  __shared__ __attribute__((aligned(1024))) char stage1[STAGE];

  // XCD-aware order (guide T1, bijective form): each XCD gets a contiguous range of logical workgroups; the m-tiles of one
  // (problem, K slice, n-tile) are adjacent, so the B panel they share is fetched into one L2 (speed only)
  int wg = blockIdx.x;
  {
    const int nb = gridDim.x;
    if (nb >= 16) {
      const int q = nb >> 3, r = nb & 7, xcd = wg & 7;
      wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
  }
  int pi = 0;
  while (pi + 1 < g.n && wg >= g.base[pi + 1]) ++pi;
  const WgProblem& p = g.p[pi];
  const int local = wg - g.base[pi];
  const int mt = local % p.m_tiles, nt = (local / p.m_tiles) % p.n_tiles, sp = local / (p.m_tiles * p.n_tiles);
  const int ks_total = (p.K + kWgBK - 1) / kWgBK;
  const int ks0 = sp * p.ksteps_per, ks1 = min(ks_total, ks0 + p.ksteps_per);
  if (ks0 >= ks1) return;
  const int m0 = mt * kWgBM, n0 = nt * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int gq = lane >> 4, i = lane & 15;

  // buffer descriptors from the tile's first column: offsets past the last valid k-row read as zero (K tail); columns past
  // M / N inside a row read the neighbouring bytes (in bounds) and only feed output columns that are never stored
  const unsigned lda_b = (unsigned)p.lda * 2u, ldb_b = (unsigned)p.ldb * 2u;
  const __amdgpu_buffer_rsrc_t rsa =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.A + m0), 0, (int)(((int64_t)p.K * p.lda - p.a_col0 - m0) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.B + n0), 0, (int)(((int64_t)p.K * p.ldb - n0) * 2), 0x00020000);
  unsigned va[IA::NI], vb[IB::NI];
  wg_dma_offsets<kWgBM>(va, lda_b, wave, lane);
  wg_dma_offsets<BN>(vb, ldb_b, wave, lane);
  auto a_step = [&](int ks) -> unsigned { return (unsigned)(ks * kWgBK) * lda_b; };  // byte offset of k-step ks (uniform)

  // fragment read offsets: k-row 4g + (i >> 2) of the 32-k block, 32-byte segment (col0 / 16) ^ (k & 7), 8 bytes per lane
  const int kr = 4 * gq + (i >> 2), x = kr & 7;
  int offa[TM], offb[TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) offa[tm] = kr * IA::RB + ((((wm * 8 + tm)) ^ x) << 5) + 8 * (i & 3);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) offb[tn] = kr * IB::RB + ((((wn * TN + tn)) ^ x) << 5) + 8 * (i & 3);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};

  // k-steps past this workgroup's slice load from an offset beyond the descriptors' range: zeros (see the loop below)
  constexpr unsigned kOob = 0x40000000u;  // >= num_records (wgrad_eligible: operands < 2^30 bytes)
  auto issue = [&](char* st, int ks) {
    const bool in = ks < ks1;
    wg_issue<kWgBM>(st, rsa, va, in ? a_step(ks) : kOob, wave);
    wg_issue<BN>(st + IA::BYTES, rsb, vb, in ? (unsigned)(ks * kWgBK) * ldb_b : kOob, wave);
  };
  auto compute = [&](const char* As, char* nxt, int ks_next) {
    const char* Bs = As + IA::BYTES;
#pragma unroll
    for (int j = 0; j < kWgBK / 32; ++j) {
      bf16x8 a[TM], b[TN];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[tn] = wg_frag<IB::RB>(Bs, offb[tn], j);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) a[tm] = wg_frag<IA::RB>(As, offa[tm], j);
      {  // this half of the next stage's DMA pieces: issued while the fragment reads are in flight
        const bool in = ks_next < ks1;
        if (j == 0)
          wg_issue<kWgBM>(nxt, rsa, va, in ? a_step(ks_next) : kOob, wave);
        else
          wg_issue<BN>(nxt + IA::BYTES, rsb, vb, in ? (unsigned)(ks_next * kWgBK) * ldb_b : kOob, wave);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  };
  // One K-step: the next stage's DMA stays in flight under this stage's MFMAs (its buffer was released by the barrier that ended
  // the previous step).  The loop body is branch-free and handles two steps (one per LDS object): an odd slice gets one padding
  // step whose operands are the zeros of out-of-range loads, and the look-ahead DMA of the last step is such a zero fill too.
  auto step = [&](const char* cur, char* nxt, int ks_next) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's pieces of the current stage have landed
    compute(cur, nxt, ks_next);
    __builtin_amdgcn_s_barrier();  // every wave is done reading it: the next step may refill it
  };
  issue(stage0, ks0);
  for (int ks = ks0; ks < ks1; ks += 2) {
    step(stage0, stage1, ks + 1);
    step(stage1, stage0, ks + 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last look-ahead DMA (zeros) must land before the LDS is released

  // split-K partial tile -> f32 atomics: a 16-lane group adds 64 contiguous bytes of one output row per instruction
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n = n0 + wn * (BN / 4) + tn * 16 + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 128 + tm * 16 + 4 * gq + r;
        if (m < p.M && n < p.N) {
          float* c = p.C + (int64_t)m * p.ldc + n;
          if (p.splitk == 1 && !p.shared_c)
            *c += acc[tm][tn][r];  // the only workgroup on this tile: plain read-modify-write (the atomic path adds ~1.3 TB/s)
          else
            atomicAdd(c, acc[tm][tn][r]);
        }
      }
    }
}

template __global__ void wgrad_kernel<256>(WgGroup);
template __global__ void wgrad_kernel<128>(WgGroup);

bool wgrad_eligible(const WgProblem& p) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || !p.A || !p.B || !p.C) return false;
  if ((((uintptr_t)p.A) | ((uintptr_t)p.B)) & 15) return false;
  if ((p.lda % 8) || (p.ldb % 8) || p.lda < p.a_col0 + p.M || p.ldb < p.N || p.a_col0 < 0) return false;
  // 32-bit buffer offsets / num_records
  if ((int64_t)p.K * p.lda * 2 >= (1LL << 30) || (int64_t)p.K * p.ldb * 2 >= (1LL << 30)) return false;
  return true;
}

template <int BN>
static int launch_class(const WgProblem* ps, const int* which, int n, hipStream_t st) {
  // one launch of about one workgroup per CU: the K slices are what is left after the tiles
  for (int at = 0; at < n; at += kMaxWgProblems) {
    const int cnt = n - at < kMaxWgProblems ? n - at : kMaxWgProblems;
    int64_t tiles = 0;
    for (int k = 0; k < cnt; ++k) {
      const WgProblem& p = ps[which[at + k]];
      tiles += fh_cdiv(p.M, kWgBM) * fh_cdiv(p.N, BN);
    }
    // K slices: every extra slice adds a tile of f32 atomics per output tile (the chip adds ~1.3 TB/s, guide: global float
    // atomics) and shortens the slices; pick the count that minimises  waves x steps x t_step + atomic bytes / rate
    // (t_step: one 64-k step of a workgroup at ~1 PFLOP/s over 256 CUs)
    int64_t ks_max = 1;
    for (int k = 0; k < cnt; ++k) ks_max = std::max<int64_t>(ks_max, fh_cdiv(ps[which[at + k]].K, kWgBK));
    const double t_step = 2.0 * kWgBM * BN * kWgBK / (1.0e15 / 256), tile_bytes = 4.0 * kWgBM * BN;
    int64_t sk = 1;
    double best = 1e30;
    for (int64_t c = 1; c <= 64 && c * 2 <= ks_max; ++c) {
      const double waves = (double)fh_cdiv(tiles * c, 256), steps = (double)fh_cdiv(ks_max, c) + 2.0;
      const double t = waves * steps * t_step + (c > 1 ? tiles * c * tile_bytes / 1.3e12 : 0.0);
      if (t < best) best = t, sk = c;
    }
    WgGroup g = {};
    g.n = cnt;
    for (int k = 0; k < cnt; ++k) {
      WgProblem p = ps[which[at + k]];
      p.m_tiles = (int)fh_cdiv(p.M, kWgBM);
      p.n_tiles = (int)fh_cdiv(p.N, BN);
      const int64_t ks_total = fh_cdiv(p.K, kWgBK);
      int64_t s = sk;
      if (s > ks_total / 2) s = ks_total / 2;
      if (s < 1) s = 1;
      p.ksteps_per = (int)fh_cdiv(ks_total, s);
      p.splitk = (int)fh_cdiv(ks_total, p.ksteps_per);
      g.p[k] = p;
      g.base[k + 1] = g.base[k] + p.m_tiles * p.n_tiles * p.splitk;
    }
    // two problems of one launch that accumulate into the same matrix (a net queued twice: gradient accumulation over two
    // backward passes before one optimizer step) must not take the plain read-modify-write path
    for (int a = 0; a < cnt; ++a)
      for (int b = 0; b < cnt; ++b)
        if (a != b && g.p[a].C == g.p[b].C) g.p[a].shared_c = 1;
    const dim3 grid((unsigned)g.base[cnt]), block(kWgThreads);
    hipLaunchKernelGGL((wgrad_kernel<BN>), grid, block, 0, st, g);
    const int e = fh_launch_status();
    if (e) return e;
  }
  return FHVAE_OK;
}

int launch_wgrad(const WgProblem* ps, int n, hipStream_t st) {
  if (n <= 0) return FHVAE_OK;
  int wide[256], narrow[256], nw = 0, nn = 0;
  if (n > 256) return FHVAE_ERR_LIMIT;
  for (int k = 0; k < n; ++k) {
    if (!wgrad_eligible(ps[k])) return FHVAE_ERR_ALIGN;
    if (ps[k].N > 128)
      wide[nw++] = k;
    else
      narrow[nn++] = k;
  }
  int e = launch_class<256>(ps, wide, nw, st);
  if (e) return e;
  return launch_class<128>(ps, narrow, nn, st);
}

}  // namespace fh

using namespace fh;

// C[M,N] (f32, ldc) += A[K,M]^T . B[K,N]: bf16 operands with the contraction index as the ROW of both (lda, ldb in elements,
// multiples of 8; 16-byte aligned bases).  The weight-gradient contraction of a linear / LSTM layer over K = batch x time rows
// (dW += dY^T X; nn.Linear's backward at simple_fhvae.py:127-134, torch.nn.LSTM's for the stub fhvae.py:14).
extern "C" int fhvae_wgrad_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, float* c, int64_t ldc, int64_t M, int64_t N,
                                int64_t K, void* stream) {
  FH_CHECK_PTR(a);
  FH_CHECK_PTR(b);
  FH_CHECK_PTR(c);
  FH_CHECK_POS(M);
  FH_CHECK_POS(N);
  FH_CHECK_POS(K);
  FH_CHECK_I32(M);
  FH_CHECK_I32(N);
  FH_CHECK_I32(K);
  WgProblem p = {};
  p.A = (const u16*)a, p.B = (const u16*)b, p.C = c;
  p.lda = lda, p.ldb = ldb, p.ldc = ldc;
  p.M = (int)M, p.N = (int)N, p.K = (int)K;
  if (!wgrad_eligible(p)) return FHVAE_ERR_ALIGN;
  return launch_wgrad(&p, 1, (hipStream_t)stream);
}
