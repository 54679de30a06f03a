// wgrad_f32.hip -- the weight gradients of the LSTM nets in the f32 parity mode (configs[4]): C[M,N] += A[K,M]^T . B[K,N] with
// K = T*B (81,920 at configs[4]), f32 operands, exact-f32 MFMA (v_mfma_f32_16x16x4_f32: every product and every accumulation step
// an f32 fma, like autograd's dy^T x that it replaces, train_model.py:452).  The f32 form of wgrad.hip ("an f32
// long-K weight-gradient kernel"): the same tiling and pipeline where the element size allows --
//   * one 512-thread workgroup = a 256 x 256 (256 x 128 for N <= 128) output tile, 8 waves as 2 (m) x 4 (n), 128 x 64 per wave:
//     128 accumulator registers; per 4-k step a wave reads 8 + 4 operand scalars (ds_read_b32) for 32 MFMAs of 32 cycles, so the
//     LDS is idle next to the matrix pipe (the generic engine's 64 x 64 tiles read one scalar per MFMA);
//   * operands go global -> LDS by LDS-DMA (16 B per lane: a k-row of 256 floats = one wave-instruction), two 32-k stages in
//     two separate __shared__ objects (counted waits, see wgrad.hip), the next stage's DMA issued under this stage's MFMAs;
//   * the image keeps the memory layout [k][column]; an MFMA operand is ONE float per lane (A[m = lane & 15][k = lane >> 4]),
//     i.e. 16 consecutive floats of each of 4 consecutive k-rows: the 64-byte group index of a row is XOR-ed with (k & 3) --
//     on the per-lane SOURCE address of the DMA and on the read address -- so the four rows fall into four different 16-bank
//     ranges (conflict-free);
//   * all weight matrices of all queued nets go out as ONE grouped launch with split-K sized to fill the chip once
//     (fhvae_lstm_param_grads_multi), f32 atomics for the partial tiles.
#include "wgrad_f32.h"

#include <algorithm>
#include <cstdlib>

#include "gemm_core.h"

namespace fh {

namespace {

constexpr int kFThreads = 512, kFBM = 256, kFBK = 32;
typedef void __attribute__((address_space(3))) * lds_void_p32;

// per-lane byte offsets (relative to the stage's first k-row) of a wave's NI DMA pieces of a W-column operand: piece = one k-row
// (W = 256) or two (W = 128); the lane fills physical 16-byte chunk pc with the bytes of logical chunk ((pc >> 2) ^ (k & 3)) << 2 | pc & 3
template <int W>
struct FImg {
  static constexpr int RB = W * 4;             // bytes per k-row
  static constexpr int BYTES = kFBK * RB;      // one stage
  static constexpr int CPR = RB / 16;          // 16-byte chunks per row
  static constexpr int RPI = 1024 / RB;        // k-rows per wave-instruction
  static constexpr int NI = BYTES / 1024 / 8;  // wave-instructions per wave and stage
};
template <int W>
__device__ __forceinline__ void f_dma_offsets(unsigned (&voff)[FImg<W>::NI], unsigned ld_bytes, int wave, int lane) {
  using I = FImg<W>;
#pragma unroll
  for (int q = 0; q < I::NI; ++q) {
    const int row = (wave * I::NI + q) * I::RPI + lane / I::CPR;
    const int pc = lane % I::CPR;
    const int c = (((pc >> 2) ^ (row & 3)) << 2) | (pc & 3);
    voff[q] = (unsigned)row * ld_bytes + (unsigned)c * 16u;
  }
}
template <int W>
__device__ __forceinline__ void f_issue(char* stage, __amdgpu_buffer_rsrc_t rs, const unsigned (&voff)[FImg<W>::NI], unsigned kbase, int wave) {
  using I = FImg<W>;
#pragma unroll
  for (int q = 0; q < I::NI; ++q)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_p32)(stage + (wave * I::NI + q) * 1024), 16, voff[q] + kbase, 0, 0, 0);
}

struct FGroup {
  int n;
  int base[kMaxWg32Problems + 1];
  WgProblem32 p[kMaxWg32Problems];
};

template <int BN>
__global__ __launch_bounds__(kFThreads) void wgrad_f32_kernel(FGroup g) {
  using IA = FImg<kFBM>;
  using IB = FImg<BN>;
  constexpr int STAGE = IA::BYTES + IB::BYTES;
  constexpr int TM = 8, TN = BN / 64;
  __shared__ __attribute__((aligned(1024))) char stage0[STAGE];
  __shared__ __attribute__((aligned(1024))) char stage1[STAGE];

  int wg = blockIdx.x;
  {  // XCD-aware order (wgrad.hip): each XCD gets a contiguous range of logical workgroups
    const int nb = gridDim.x;
    if (nb >= 16) {
      const int q = nb >> 3, r = nb & 7, xcd = wg & 7;
      wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
  }
  int pi = 0;
  while (pi + 1 < g.n && wg >= g.base[pi + 1]) ++pi;
  const WgProblem32& p = g.p[pi];
  const int local = wg - g.base[pi];
  const int mt = local % p.m_tiles, nt = (local / p.m_tiles) % p.n_tiles, sp = local / (p.m_tiles * p.n_tiles);
  const int ks_total = (p.K + kFBK - 1) / kFBK;
  const int ks0 = sp * p.ksteps_per, ks1 = min(ks_total, ks0 + p.ksteps_per);
  if (ks0 >= ks1) return;
  const int m0 = mt * kFBM, n0 = nt * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int gq = lane >> 4, i = lane & 15;

  // buffer descriptors from the tile's first column: k-rows past K read as zero (range check); columns past M / N inside a row read
  // neighbouring bytes (in bounds) and only feed output columns that are never stored
  const unsigned lda_b = (unsigned)p.lda * 4u, ldb_b = (unsigned)p.ldb * 4u;
  const __amdgpu_buffer_rsrc_t rsa =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A + m0), 0, (int)(((int64_t)p.K * p.lda - m0) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B + n0), 0, (int)(((int64_t)p.K * p.ldb - n0) * 4), 0x00020000);
  unsigned va[IA::NI], vb[IB::NI];
  f_dma_offsets<kFBM>(va, lda_b, wave, lane);
  f_dma_offsets<BN>(vb, ldb_b, wave, lane);

  // operand reads: k-row gq of the 4-k block (k & 3 == gq: the blocks start at multiples of 4), 16-float group = the 16-wide tile
  int offa[TM], offb[TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) offa[tm] = gq * IA::RB + (((wm * 8 + tm) ^ gq) << 6) + i * 4;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) offb[tn] = gq * IB::RB + (((wn * TN + tn) ^ gq) << 6) + i * 4;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr unsigned kOob = 0x7ffffff0u;  // >= num_records (wgrad32_eligible: operands < 2^31 bytes)
  // (always_inline: the body of `compute` is 256 MFMAs; left as a call, its closure -- and with it the accumulators -- lived in scratch)
  auto kofs = [&](int ks, unsigned ld_b) __attribute__((always_inline)) -> unsigned { return (unsigned)(ks * kFBK) * ld_b; };
  auto compute = [&](const char* As, char* nxt, int ks_next) __attribute__((always_inline)) {
    const char* Bs = As + IA::BYTES;
#pragma unroll
    for (int j = 0; j < kFBK / 4; ++j) {
      float a[TM], b[TN];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[tn] = *(const float*)(Bs + offb[tn] + j * 4 * IB::RB);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) a[tm] = *(const float*)(As + offa[tm] + j * 4 * IA::RB);
      if (j == 0 || j == 4) {  // the next stage's DMA pieces, in two halves under the MFMAs
        const bool in = ks_next < ks1;
        if (j == 0)
          f_issue<kFBM>(nxt, rsa, va, in ? kofs(ks_next, lda_b) : kOob, wave);
        else
          f_issue<BN>(nxt + IA::BYTES, rsb, vb, in ? kofs(ks_next, ldb_b) : kOob, wave);
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    }
  };
  auto step = [&](const char* cur, char* nxt, int ks_next) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's pieces of the current stage have landed
    compute(cur, nxt, ks_next);
    __builtin_amdgcn_s_barrier();  // every wave is done reading it: the next step may refill it
  };
  f_issue<kFBM>(stage0, rsa, va, kofs(ks0, lda_b), wave);
  f_issue<BN>(stage0 + IA::BYTES, rsb, vb, kofs(ks0, ldb_b), wave);
  for (int ks = ks0; ks < ks1; ks += 2) {  // two steps per iteration (one per LDS object); an odd slice gets one step of zeros
    step(stage0, stage1, ks + 1);
    step(stage1, stage0, ks + 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last look-ahead DMA (zeros) must land before the LDS is released

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
            *c += acc[tm][tn][r];
          else
            atomicAdd(c, acc[tm][tn][r]);
        }
      }
    }
}

template __global__ void wgrad_f32_kernel<256>(FGroup);
template __global__ void wgrad_f32_kernel<128>(FGroup);

template <int BN>
int launch_class32(const WgProblem32* ps, const int* which, int n, hipStream_t st) {
  for (int at = 0; at < n; at += kMaxWg32Problems) {
    const int cnt = n - at < kMaxWg32Problems ? n - at : kMaxWg32Problems;
    int64_t tiles = 0, ks_max = 1;
    for (int k = 0; k < cnt; ++k) {
      const WgProblem32& p = ps[which[at + k]];
      tiles += fh_cdiv(p.M, kFBM) * fh_cdiv(p.N, BN);
      ks_max = std::max<int64_t>(ks_max, fh_cdiv(p.K, kFBK));
    }
    // K slices: waves x steps x t_step + atomic bytes / rate (wgrad.hip); a 32-k step of a workgroup at the f32 MFMA rate
    const double t_step = 2.0 * kFBM * BN * kFBK / (140.0e12 / 256), tile_bytes = 4.0 * kFBM * BN;
    int64_t sk = 1;
    double best = 1e30;
    for (int64_t c = 1; c <= 64 && c * 2 <= ks_max; ++c) {
      const double waves = (double)fh_cdiv(tiles * c, 256), steps = (double)fh_cdiv(ks_max, c) + 2.0;
      const double t = waves * steps * t_step + (c > 1 ? tiles * c * tile_bytes / 1.3e12 : 0.0);
      if (t < best) best = t, sk = c;
    }
    FGroup g = {};
    g.n = cnt;
    for (int k = 0; k < cnt; ++k) {
      WgProblem32 p = ps[which[at + k]];
      p.m_tiles = (int)fh_cdiv(p.M, kFBM);
      p.n_tiles = (int)fh_cdiv(p.N, BN);
      const int64_t ks_total = fh_cdiv(p.K, kFBK);
      int64_t s = sk;
      if (s > ks_total / 2) s = ks_total / 2;
      if (s < 1) s = 1;
      p.ksteps_per = (int)fh_cdiv(ks_total, s);
      p.splitk = (int)fh_cdiv(ks_total, p.ksteps_per);
      g.p[k] = p;
      g.base[k + 1] = g.base[k] + p.m_tiles * p.n_tiles * p.splitk;
    }
    for (int a = 0; a < cnt; ++a)  // the same matrix twice in one launch (a net queued twice): atomics, no plain read-modify-write
      for (int b = 0; b < cnt; ++b)
        if (a != b && g.p[a].C == g.p[b].C) g.p[a].shared_c = 1;
    hipLaunchKernelGGL((wgrad_f32_kernel<BN>), dim3((unsigned)g.base[cnt]), dim3(kFThreads), 0, st, g);
    const int e = fh_launch_status();
    if (e) return e;
  }
  return FHVAE_OK;
}

}  // namespace

bool wgrad32_eligible(const WgProblem32& p) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || !p.A || !p.B || !p.C) return false;
  if ((((uintptr_t)p.A) | ((uintptr_t)p.B)) & 15) return false;
  if ((p.lda % 4) || (p.ldb % 4) || p.lda < p.M || p.ldb < p.N) return false;
  // 32-bit buffer offsets / num_records
  if ((int64_t)p.K * p.lda * 4 >= 0x7ffffff0LL || (int64_t)p.K * p.ldb * 4 >= 0x7ffffff0LL) return false;
  return true;
}

int launch_wgrad32(const WgProblem32* ps, int n, hipStream_t st) {
  if (n <= 0) return FHVAE_OK;
  int wide[256], narrow[256], nw = 0, nn = 0;
  if (n > 256) return FHVAE_ERR_LIMIT;
  for (int k = 0; k < n; ++k) {
    if (!wgrad32_eligible(ps[k])) return FHVAE_ERR_ALIGN;
    if (ps[k].N > 128)
      wide[nw++] = k;
    else
      narrow[nn++] = k;
  }
  int e = launch_class32<256>(ps, wide, nw, st);
  if (e) return e;
  return launch_class32<128>(ps, narrow, nn, st);
}

}  // namespace fh

using namespace fh;

// C[M,N] (f32, ldc) += A[K,M]^T . B[K,N]: f32 operands with the contraction index as the ROW of both (lda, ldb in elements,
// multiples of 4; 16-byte aligned bases), exact-f32 MFMA.  dW += dY^T X of a linear / LSTM layer over K = batch x time rows
// (autograd's at train_model.py:452; nn.Linear's backward at simple_fhvae.py:127-134).
extern "C" int fhvae_wgrad_f32(const float* a, int64_t lda, const float* b, int64_t ldb, float* c, int64_t ldc, int64_t M, int64_t N,
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
  WgProblem32 p = {};
  p.A = a, p.B = b, p.C = c;
  p.lda = lda, p.ldb = ldb, p.ldc = ldc;
  p.M = (int)M, p.N = (int)N, p.K = (int)K;
  if (!wgrad32_eligible(p)) return FHVAE_ERR_ALIGN;
  return launch_wgrad32(&p, 1, (hipStream_t)stream);
}
