// proj.hip -- C[M,N] = A[M,K] . B[N,K]^T (+ bias): bf16 operands with the contraction index CONTIGUOUS in both (activations
// times a weight matrix as torch.nn.Linear / torch.nn.LSTM store it), f32 accumulate, f32 output.  M = T*B rows (40,960 at
// the bench shape), N and K a few hundred: the projections around the recurrences --
//   * the from-above term of the LSTM backward, dh^{l}_t += dg^{l+1}_t . W_ih[l+1] for all t at once (M = T*B, K = 4H, N = H;
//     operand B = the transposed bf16 weight copy [H,4H] of the workspace), between the two per-layer launches of a net
//     (lstm_bwd_rs.hip; the body the reference never wrote, fhvae.py:14);
// Why a kernel of its own: the generic engine (gemm_core.h: 4 waves, 64x64 tiles, one workgroup-wide barrier pair per 32 k)
// ran this shape at 325 TFLOP/s (66 us); a prologue of the persistent launch that computed it per cluster member re-read the
// operand four times and took 70 us.  Here:
//   * ONE tile per CU: the row tile BM is chosen so that ceil(M / BM) is about the CU count (BM = 160 at M = 40,960: 256 tiles;
//     256-row tiles would leave 96 CUs idle), all N columns per tile (BN = 256 or 128): A is read exactly once;
//   * 512 threads = 8 waves as 2 (m) x 4 (n); operands go global -> LDS by LDS-DMA (8 rows x 128 B per wave-instruction, whole
//     lines), two 64-k stages in two LDS objects (counted waits, raw barriers: the structure of wgrad.hip), the pieces of the
//     next stage issued behind the fragment reads of the current one;
//   * LDS image [row][8 chunks] with the 16-byte chunk XOR-ed by (row & 7) on the DMA's per-lane source and on the ds_read_b128
//     (conflict-free for 128-byte rows: the two rows of a bank row take opposite halves);
//   * MFMA roles swapped (weight fragment first): a lane holds 4 consecutive columns of one row -> 16-byte f32 stores.
#include "proj.h"

#include <cstdlib>

#include "gemm_core.h"

namespace fh {

constexpr int kPjThreads = 512;
typedef void __attribute__((address_space(3))) * lds_void_p;

struct ProjArgs {
  const u16* A;
  const u16* B;
  float* C;
  const float* bias;   // [N] or NULL
  const float* bias2;  // columns >= bias_split take bias2[n - bias_split] (two heads side by side); NULL: one vector
  int bias_split;
  int64_t lda, ldb, ldc;
  int M, N, K;
};

// NP 1-KB pieces (8 rows x 128 B) of one operand stage, piece wave + 8 i by this wave.  (A __device__ function, not a lambda of
// the kernel: with the address-space cast inside a lambda the HOST pass silently dropped the kernel's instantiation.)
template <int NP>
__device__ __forceinline__ void pj_issue(char* st, __amdgpu_buffer_rsrc_t rs, const unsigned (&voff)[(NP + 7) / 8], unsigned kbytes, int wave) {
#pragma unroll
  for (int i = 0; i < (NP + 7) / 8; ++i)
    if (wave + 8 * i < NP)  // (uniform)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_p)(st + (wave + 8 * i) * 1024), 16, voff[i] + kbytes, 0, 0, 0);
}

template <int BM, int BN>
__global__ __launch_bounds__(kPjThreads) void proj_kernel(ProjArgs p) {
  constexpr int TM = BM / 32, TN = BN / 64;       // 16x16 tiles per wave: (BM/2 rows) x (BN/4 columns)
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int NA = BM / 8, NB = BN / 8;         // 1-KB DMA pieces (8 rows x 128 B) per stage
  constexpr int PA = (NA + 7) / 8, PB = NB / 8;   // per wave
  static_assert(BM % 32 == 0 && BN % 64 == 0, "tile shape");
  __shared__ __attribute__((aligned(1024))) char stage0[STAGE];
  __shared__ __attribute__((aligned(1024))) char stage1[STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  // rows past M / N read as zeros (buffer range check); K % 64 == 0 (host-checked), so no chunk straddles a row end
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.A + (int64_t)m0 * p.lda), 0,
                                                                     (int)(((int64_t)(p.M - m0) * p.lda) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.B + (int64_t)n0 * p.ldb), 0,
                                                                     (int)(((int64_t)(p.N - n0) * p.ldb) * 2), 0x00020000);
  constexpr unsigned kOob = 0x7f000000u;
  unsigned va[PA], vb[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int piece = wave + 8 * i, row = piece * 8 + (lane >> 3);
    va[i] = piece < NA ? (unsigned)row * (unsigned)(p.lda * 2) + (unsigned)(((lane & 7) ^ (row & 7)) << 4) : kOob;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int piece = wave + 8 * i, row = piece * 8 + (lane >> 3);
    vb[i] = (unsigned)row * (unsigned)(p.ldb * 2) + (unsigned)(((lane & 7) ^ (row & 7)) << 4);
  }
  auto issue_a = [&](char* st, int k0) { pj_issue<NA>(st, rsa, va, (unsigned)k0 * 2u, wave); };
  auto issue_b = [&](char* st, int k0) { pj_issue<NB>(st + A_BYTES, rsb, vb, (unsigned)k0 * 2u, wave); };
  // fragment reads: row (tile * 16 + r), chunk (4 j + q) ^ (r & 7): two bases (j), the tile is an immediate
  int offj[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) offj[j] = r * 128 + (((j * 4 + q) ^ (r & 7)) << 4);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / 64;
  auto compute = [&](const char* cur, char* nxt, int ks_next) {
    const char* As = cur + wm * (BM / 2) * 128;
    const char* Bs = cur + A_BYTES + wn * (BN / 4) * 128;
    const int k0n = ks_next < nk ? ks_next * 64 : (int)(kOob >> 1);  // past the end: out of range = zeros, never read
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16x8 a[TM], b[TN];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[tn] = __builtin_bit_cast(bf16x8, *(const uint4*)(Bs + offj[j] + tn * 2048));
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) a[tm] = __builtin_bit_cast(bf16x8, *(const uint4*)(As + offj[j] + tm * 2048));
      if (j == 0)
        issue_a(nxt, k0n);
      else
        issue_b(nxt, k0n);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[tn], a[tm], acc[tm][tn], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  };
  auto step = [&](const char* cur, char* nxt, int ks_next) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the current stage (issued a whole step ago)
    __builtin_amdgcn_s_barrier();                     // ... and every other wave's
    compute(cur, nxt, ks_next);
    __builtin_amdgcn_s_barrier();                     // every wave is done reading it: the next step may refill it
  };
  issue_a(stage0, 0);
  issue_b(stage0, 0);
  for (int ks = 0; ks < nk; ks += 2) {
    step(stage0, stage1, ks + 1);
    if (ks + 1 < nk) step(stage1, stage0, ks + 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last look-ahead DMA (zeros) must land before the LDS is released

#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn * (BN / 4) + tn * 16 + 4 * q;
    if (n >= p.N) continue;  // (N % 4 == 0, host-checked)
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = (p.bias2 && n >= p.bias_split) ? *(const f32x4*)(p.bias2 + (n - p.bias_split)) : *(const f32x4*)(p.bias + n);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int m = m0 + wm * (BM / 2) + tm * 16 + r;
      if (m < p.M) *(f32x4*)(p.C + (int64_t)m * p.ldc + n) = acc[tm][tn] + bv;
    }
  }
}

// (explicit instantiations: left implicit, the host stubs of kernels with static LDS objects were not emitted)
template __global__ void proj_kernel<64, 256>(ProjArgs);
template __global__ void proj_kernel<96, 256>(ProjArgs);
template __global__ void proj_kernel<128, 256>(ProjArgs);
template __global__ void proj_kernel<160, 256>(ProjArgs);
template __global__ void proj_kernel<192, 256>(ProjArgs);
template __global__ void proj_kernel<224, 256>(ProjArgs);
template __global__ void proj_kernel<256, 256>(ProjArgs);
template __global__ void proj_kernel<64, 128>(ProjArgs);
template __global__ void proj_kernel<96, 128>(ProjArgs);
template __global__ void proj_kernel<128, 128>(ProjArgs);
template __global__ void proj_kernel<160, 128>(ProjArgs);
template __global__ void proj_kernel<192, 128>(ProjArgs);
template __global__ void proj_kernel<224, 128>(ProjArgs);
template __global__ void proj_kernel<256, 128>(ProjArgs);

template <int BM, int BN>
static int launch_one(const ProjArgs& a, hipStream_t st) {
  const dim3 grid((unsigned)fh_cdiv(a.M, BM), (unsigned)fh_cdiv(a.N, BN));
  hipLaunchKernelGGL((proj_kernel<BM, BN>), grid, dim3(kPjThreads), 0, st, a);
  return fh_launch_status();
}

template <int BN>
static int launch_bn(const ProjArgs& a, int bm, hipStream_t st) {
  switch (bm) {
    case 64: return launch_one<64, BN>(a, st);
    case 96: return launch_one<96, BN>(a, st);
    case 128: return launch_one<128, BN>(a, st);
    case 160: return launch_one<160, BN>(a, st);
    case 192: return launch_one<192, BN>(a, st);
    case 224: return launch_one<224, BN>(a, st);
    default: return launch_one<256, BN>(a, st);
  }
}

bool proj_eligible(const void* a, int64_t lda, const void* b, int64_t ldb, const float* c, int64_t ldc, int64_t M, int64_t N,
                   int64_t K) {
  if (!a || !b || !c || M <= 0 || N <= 0 || K <= 0) return false;
  if ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) return false;
  if ((K % 64) || (N % 4) || (lda % 8) || (ldb % 8) || (ldc % 4) || lda < K || ldb < K || ldc < N) return false;
  if (M * lda * 2 >= (1LL << 31) || N * ldb * 2 >= (1LL << 30)) return false;  // 32-bit buffer offsets / num_records
  return true;
}

int launch_proj(const void* a, int64_t lda, const void* b, int64_t ldb, float* c, int64_t ldc, const float* bias, int64_t M,
                int64_t N, int64_t K, hipStream_t st, const float* bias2, int bias_split) {
  if (!proj_eligible(a, lda, b, ldb, c, ldc, M, N, K) || (bias2 && (bias_split % 4))) return FHVAE_ERR_ALIGN;
  ProjArgs p = {(const u16*)a, (const u16*)b, c, bias, bias2, bias_split, lda, ldb, ldc, (int)M, (int)N, (int)K};
  const int BN = N > 128 ? 256 : 128;
  const int64_t ncol = fh_cdiv(N, BN);
  // the row tile that fills the chip's 256 CUs in the fewest rounds with the least padding: one round if it can
  int best = 256;
  double best_t = 1e30;
  for (int bm = 64; bm <= 256; bm += 32) {
    const int64_t tiles = fh_cdiv(M, bm) * ncol;
    const double t = (double)fh_cdiv(tiles, 256) * (bm + 24);  // per-tile time ~ rows + a fixed cost (weights, fill, drain)
    if (t < best_t) best_t = t, best = bm;
  }
  return BN == 256 ? launch_bn<256>(p, best, st) : launch_bn<128>(p, best, st);
}

}  // namespace fh

using namespace fh;

extern "C" int fhvae_proj_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, const float* bias, float* c, int64_t ldc,
                               int64_t M, int64_t N, int64_t K, void* stream) {
  FH_CHECK_PTR(a);
  FH_CHECK_PTR(b);
  FH_CHECK_PTR(c);
  FH_CHECK_POS(M);
  FH_CHECK_POS(N);
  FH_CHECK_POS(K);
  FH_CHECK_I32(M);
  return launch_proj(a, lda, b, ldb, c, ldc, bias, M, N, K, (hipStream_t)stream, nullptr, 0);
}
