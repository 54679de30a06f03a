// lstm_stream.hip -- the large-tile bf16 LSTM cells of lstm_cell.hip as ONE persistent launch per direction (H up to 512:
// configs[3]), instead of one launch per wavefront step.
//
// Why: a step launch of those cells is the sum of two phases that load different parts of the chip -- the contraction (bound by
// what a CU takes in from L2, ~70 GB/s) and the epilogue (bound by HBM: gates, cell state, saved activations) -- plus a
// dependent-launch gap; measured per step 12.6 + 9 + ~4 us forward, 21 + 11 + ~3 us backward.  The recurrence only couples the
// hidden units of one batch row, so the workgroups that share a ROW TILE (all unit tiles of both layers) form a cluster that
// synchronises through flags in its XCD's L2 and never with another cluster: the clusters drift apart, one cluster's epilogue
// runs beside another's contraction on the same CU (two workgroups per CU), the gaps disappear, and the cell state c / dc
// stays in registers for the whole sequence.
//
// Unlike the H <= 256 cluster kernels (lstm_cluster.hip) the weights do NOT fit the LDS (a row tile's W slice is 128 x 1024
// bf16 per layer and unit tile): they are streamed from L2 every step exactly as in the step cells (same K loop, same LDS ring,
// lstm_cell_dev.h).  Hand-off protocol = lstm_cluster.hip's (measured there at 1.1-1.4 us per step): plain stores ->
// s_waitcnt vmcnt(0) -> workgroup barrier -> one agent-scope flag store; consumers poll with L1-bypassing loads and read the
// exchanged operand (h / dgates) with sc1 LDS-DMA.  A cluster lives on one XCD: a workgroup reads HW_REG_XCC_ID and takes a
// slot by an atomic ticket on that XCD's counter (64 tickets per XCD and launch; ticket / 64 numbers the launch on the sync
// block, which the operand cast of every forward re-arms).  Every spin is bounded and watches the status word.
// Semantics: lstm_cell.hip / lstm.hip (torch.nn.LSTM gate order; stands where the FC layers of simple_fhvae.py:160-164,
// :186-190, :240-244 stand).
#include "lstm_stream.h"

#include <cstdlib>
#include <cstring>

#include "lstm_cell_dev.h"
#include "lstm_cluster.h"
#include "trace.h"

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

namespace fh {

constexpr unsigned kStreamSpinLimit = 1u << 20;
constexpr int kStreamGrid = 512;  // two workgroups per CU, 64 per XCD
constexpr int kStreamSc1 = 16;    // aux bits of an L1-bypassing load

struct StreamSeg {  // one K segment of a layer's cell
  const u16* A;     // operand rows of step t: A + t * a_tstride (elements), rows of this launch's first batch row
  int64_t a_tstride;
  int64_t lda;
  const u16* W;  // [4H, ldw] (forward: gate rows) / [H, ldw] (backward: the transposed copy)
  int64_t ldw;
  int K;
  int t_first, t_last;  // the segment exists for t_first <= t <= t_last
};

struct FwdStreamLayer {
  StreamSeg seg[3];
  const float* bias_a;
  const float* bias_b;
  float* cs;      // + t * B * H
  u16* hs;        // + t * B * H
  u16* gates;     // + t * B * 4H
  float* hs_f32;  // + t * B * H, or NULL
  float* hn;      // this layer's slot of the (B, L*H) final-state buffer, or NULL
  int64_t hn_ld;
};
struct FwdStream {
  unsigned* sync;
  int B, H, T, L;
  int nrows;  // rows of this launch (<= 2048, a multiple of 128); every row pointer already points at its first row
  int stagger;  // the second half of an XCD's workgroups starts this many s_sleep(127) (~3.4 us) later (see the kernel)
  unsigned long long* tlog;  // FHVAE_CLUSTER_TLOG: phase clocks of row tile 0, unit tile 0 (tools/prof_stream.py)
  FwdStreamLayer layer[2];
};

struct BwdStreamLayer {
  StreamSeg seg[2];  // [0] dg^l_{t+1} . W_hh[l] (t <= T-2), [1] dg^{l+1}_t . W_ih[l+1]
  const float* ext;  // d_hs_top + t * B * H (top layer) or NULL
  const float* ext2; // slot of d_hn (t == T-1) or NULL
  int64_t ext2_ld;
  const u16* gates;  // + t * B * 4H
  const float* cs;   // + t * B * H
  u16* dg;           // + t * B * 4H
};
struct BwdStream {
  unsigned* sync;
  int B, H, T, L;
  int nrows;
  int stagger;
  unsigned long long* tlog;
  BwdStreamLayer layer[2];
};

__device__ __forceinline__ unsigned stream_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__device__ __forceinline__ void stream_give_up(unsigned* sync, unsigned code) {
  __hip_atomic_fetch_or(sync + kSyncStatus, code, RLX_AGENT);
  const unsigned long long a = (unsigned long long)__hip_atomic_load(sync + kSyncSticky, RLX_AGENT) |
                               ((unsigned long long)__hip_atomic_load(sync + kSyncSticky + 1, RLX_AGENT) << 32);
  if (a) __hip_atomic_fetch_or((unsigned*)a, code, RLX_AGENT);
}

// Placement.  A cluster must live on one XCD (its exchange stays in that L2), and the two workgroups of a CU should belong to
// DIFFERENT clusters that run half a step apart: one's epilogue and flag wait then fall into the other's contraction, which gets
// the CU's whole L2 -> LDS intake (~70 GB/s; two contractions side by side get half each).  The dispatcher gives no such
// guarantee (measured: the co-resident pairs are mostly neighbours in dispatch order), so a workgroup takes its role from where it
// finds itself: the XCD (HW_REG_XCC_ID), then the CU inside it (HW_REG_HW_ID: se | sh | cu) -- the first arrival on a CU joins
// the early half of the XCD's clusters, the second the late half -- then a role ticket inside that half.  Counters in the sync
// block (re-armed by every forward's operand cast); each launch adds 2 per CU and 32 per half, so the counts also number the
// launch.  A launch whose workgroups are not all resident at once ends through the bounded spins (status word).
constexpr int kStreamCuCnt = kSyncFlags + 512;          // + xcd * 256 + (HW_ID >> 8 & 255)
constexpr int kStreamRoleCnt = kStreamCuCnt + 8 * 256;  // + xcd * 2 + half
static_assert(kStreamRoleCnt + 16 <= kSyncWordsUsed, "sync block layout");

// (launch << 16) | (xcd << 6) | (half << 5) | role, or -1
__device__ __forceinline__ int stream_join(unsigned* sync, int* s_word) {
  if (threadIdx.x == 0) {
    const unsigned x = stream_xcc_id();
    int v = -1;
    if (x < 8) {
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      const unsigned arrival = __hip_atomic_fetch_add(sync + kStreamCuCnt + x * 256 + ((hw >> 8) & 255u), 1u, RLX_AGENT);
      const unsigned half = arrival & 1u;
      const unsigned ticket = __hip_atomic_fetch_add(sync + kStreamRoleCnt + x * 2 + half, 1u, RLX_AGENT);
      v = (int)(((ticket >> 5) << 16) | (x << 6) | (half << 5) | (ticket & 31u));
    }
    if (v < 0) stream_give_up(sync, 2u);
    s_word[0] = v;
  }
  __syncthreads();
  return s_word[0];
}

// Every wave polls for itself: lanes [0, n_a) watch fa[] for need_a, lanes [16, 16 + n_b) watch fb[] for need_b (0 = nothing to
// wait for).  false = abort.
__device__ __forceinline__ bool stream_wait(unsigned* sync, const unsigned* fa, int n_a, unsigned need_a, const unsigned* fb, int n_b,
                                            unsigned need_b) {
  const int lane = threadIdx.x & 63;
  const unsigned* fp = nullptr;
  unsigned need = 0;
  if (lane < n_a && need_a) fp = fa + lane, need = need_a;
  if (lane >= 16 && lane < 16 + n_b && need_b) fp = fb + (lane - 16), need = need_b;
  for (unsigned spins = 0;; ++spins) {
    unsigned v = need, st = 0;
    if (fp) v = __hip_atomic_load(fp, RLX_AGENT);
    if (lane == 63) st = __hip_atomic_load(sync + kSyncStatus, RLX_AGENT);
    if (__any(st != 0)) return false;
    if (__all(v >= need)) break;
    if (spins > kStreamSpinLimit) {
      if (lane == 0) stream_give_up(sync, 1u);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);  // the other workgroup of this CU is computing
  }
  asm volatile("" ::: "memory");
  return true;
}

// all stores of this workgroup have reached the XCD's L2, then ONE lane raises the flag
__device__ __forceinline__ void stream_publish(unsigned* flag, unsigned epoch) {
  cell_wait_vmcnt<0>();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, RLX_AGENT);
}

__device__ __forceinline__ int seg_steps(const StreamSeg& S, int t) {
  return (t >= S.t_first && t <= S.t_last) ? (S.K + kCellBK - 1) / kCellBK : 0;
}

// ---------------------------------------------------------------------------------------------
// forward: workgroup = (row tile of 128, layer, 32 units x 4 gates) for all T steps
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kCellThreads, 2) void cell_fwd_stream_kernel(FwdStream p) {
  constexpr int BM = 128, RB = 128, UN = 32, NS = 2;
  constexpr int STAGE = (BM + RB) * 128;
  __shared__ __attribute__((aligned(1024))) char st0[STAGE];
  __shared__ __attribute__((aligned(1024))) char st1[STAGE];
  __shared__ __attribute__((aligned(1024))) char st2[16];
  __shared__ __attribute__((aligned(1024))) char st3[16];
  __shared__ int s_word[4];
  const int id = stream_join(p.sync, s_word);
  if (id < 0) return;
  const int H = p.H, T = p.T;
  const int NY = H / UN, R = p.nrows / BM;
  const int xs = id & 0xffff, xcd = xs >> 6, slot = xs & 63;
  const int m = xcd + 8 * (slot >> 5), l = (slot >> 4) & 1, y = slot & 15;
  if (m >= R || l >= p.L || y >= NY) return;
  const unsigned ep0 = (unsigned)(id >> 16) * (unsigned)kSeqEpochs;
  unsigned* f_own = p.sync + kSyncFlags + (m * 2 + l) * 16;
  const unsigned* f_below = p.sync + kSyncFlags + (m * 2 + (l > 0 ? l - 1 : 0)) * 16;
  const FwdStreamLayer& Lr = p.layer[l];
  const int m0 = m * BM, u0 = y * UN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 15, gq = lane >> 4;
  const unsigned uH = (unsigned)H;
  const int64_t BH = (int64_t)p.B * H;

  const unsigned c8 = (unsigned)((lane & 7) ^ (lane >> 3));
  const unsigned rowa0 = (unsigned)(wave * 32 + (lane >> 3));
  const unsigned rowb0 = (unsigned)((wave & 1) * 2 * H + u0 + (wave >> 1) * 16 + (lane >> 3));
  auto swz = [](int row) { return (row & 1) ^ (((row >> 2) & 1) << 2) ^ ((((row >> 1) ^ (row >> 3)) & 1) << 3); };

  float bsum[4];
  {
    const unsigned unit = u0 + wn * 16 + i;
#pragma unroll
    for (int g = 0; g < 4; ++g) bsum[g] = Lr.bias_a[g * uH + unit] + Lr.bias_b[g * uH + unit];
  }
  const int lr = threadIdx.x >> 2, chunk = threadIdx.x & 3;  // epilogue items: (row lr of half k, 8 units)
  const unsigned u = u0 + chunk * 8;
  float creg[2][8];  // the cell state of this workgroup's tile, in the epilogue's item layout
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) creg[k][e] = 0.f;

  // the late half (stream_join) starts about half a step after the early one
  if (slot >= 32)
    for (int k = 0; k < p.stagger; ++k) __builtin_amdgcn_s_sleep(127);

  unsigned long long* tl = (p.tlog && m == 0 && y == 0 && threadIdx.x == 0) ? p.tlog + l * 256 : nullptr;
#define ST_TLOG(slot_)                  \
  do {                                  \
    if (tl) tl[(slot_)] = wall_clock64(); \
  } while (0)
  for (int t = 0; t < T; ++t) {
    ST_TLOG(t * 8 + 0);
    // h^l_{t-1} of every unit tile of this row tile (t > 0) and h^{l-1}_t (l > 0)
    if (t > 0 || l > 0) {
      if (!stream_wait(p.sync, f_own, NY, t > 0 ? ep0 + (unsigned)t : 0u, f_below, NY, l > 0 ? ep0 + (unsigned)t + 1u : 0u)) return;
    }
    ST_TLOG(t * 8 + 1);
    const int end0 = seg_steps(Lr.seg[0], t), end1 = end0 + seg_steps(Lr.seg[1], t), end2 = end1 + seg_steps(Lr.seg[2], t);
    const int nsteps = end2;
    f32x4 acc[4][4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto issue = [&](char* stg, int ks, int part) {
      const int s = (ks >= end0) + (ks >= end1);  // uniform; the segment is read from the kernel arguments by index (no branch)
      int start = ks >= end0 ? end0 : 0;
      start = ks >= end1 ? end1 : start;
      const int kl = ks - start;
      const StreamSeg& S = Lr.seg[s];
      const unsigned la = (unsigned)(S.lda * 2), lb = (unsigned)(S.ldw * 2);
      const __amdgpu_buffer_rsrc_t a =
          __builtin_amdgcn_make_buffer_rsrc((u16*)S.A + (int64_t)t * S.a_tstride + (int64_t)m0 * S.lda, 0, (int)(BM * la), 0x00020000);
      const __amdgpu_buffer_rsrc_t b = __builtin_amdgcn_make_buffer_rsrc((u16*)S.W, 0, (int)(4 * H * lb), 0x00020000);
      const int segK = S.K;
      const unsigned oob = (unsigned)((int)(ks >= nsteps) | (int)(kl * kCellBK + (int)c8 * 8 >= segK)) << 30;
      const unsigned kb = ((unsigned)(kl * (kCellBK * 2)) + c8 * 16u) | oob;
      unsigned xa[4], xb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xa[q] = (rowa0 + (unsigned)(q * 8)) * la + kb;
        xb[q] = (rowb0 + (unsigned)((q >> 1) * H + (q & 1) * 8)) * lb + kb;
      }
      if (part != 1) cell_issue<4, kStreamSc1>(stg, a, xa, 0u, wave);  // h of this launch (and x, xc): past the L1
      if (part != 0) cell_issue<4>(stg + BM * 128, b, xb, 0u, wave);
    };
    cell_mainloop<u16, BM, RB, NS>(acc, nsteps, issue, st0, st1, st2, st3);
    ST_TLOG(t * 8 + 2);

    // epilogue through LDS (lstm_cell.hip): X[row][gate][32 units] f32, then (row, 8 units) items with 16-byte accesses
    __syncthreads();
    {
      char* xw = wm ? st1 : st0;
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int xr = tm * 16 + gq * 4 + r;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int slot_ = g * 8 + wn * 4 + (i >> 2);
            *(cell_lds_f)(xw + xr * 512 + ((slot_ ^ swz(xr)) << 4) + (i & 3) * 4) = acc[tm][g][r] + bsum[g];
          }
        }
    }
    __syncthreads();
    float* c_out = Lr.cs + (int64_t)t * BH;
    u16* h_out = Lr.hs + (int64_t)t * BH;
    u16* g_out = Lr.gates + (int64_t)t * BH * 4;
    float* hf = Lr.hs_f32 ? Lr.hs_f32 + (int64_t)t * BH : nullptr;
    float* hn = (t == T - 1) ? Lr.hn : nullptr;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const char* xr = (k ? st1 : st0) + lr * 512;
      const unsigned row = m0 + k * 64 + lr;
      float x[4][8];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const f32x4v v = *(cell_lds_f4)(xr + (((g * 8 + chunk * 2 + hh) ^ swz(lr)) << 4));
#pragma unroll
          for (int e = 0; e < 4; ++e) x[g][hh * 4 + e] = v[e];
        }
      float ig[8], fg[8], gg[8], og[8], h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ig[e] = sigmoidf_(x[0][e]);
        fg[e] = sigmoidf_(x[1][e]);
        gg[e] = tanhf_(x[2][e]);
        og[e] = sigmoidf_(x[3][e]);
        creg[k][e] = __builtin_fmaf(fg[e], creg[k][e], ig[e] * gg[e]);
        h[e] = og[e] * tanhf_(creg[k][e]);
      }
      const unsigned o = row * uH + u;
      st8(c_out + o, creg[k]);
      st8_bf(h_out + o, h);
      if (hf) st8(hf + o, h);
      u16* go = g_out + row * 4u * uH + u;
      st8_bf(go, ig);
      st8_bf(go + uH, fg);
      st8_bf(go + 2 * uH, gg);
      st8_bf(go + 3 * uH, og);
      if (hn) st8(hn + row * (unsigned)Lr.hn_ld + u, h);
    }
    ST_TLOG(t * 8 + 3);
    stream_publish(f_own + y, ep0 + (unsigned)t + 1u);  // (its barrier also orders the X reads before the next step's DMA)
    ST_TLOG(t * 8 + 4);
  }
}

// ---------------------------------------------------------------------------------------------
// backward: workgroup = (row tile of 64, layer, 64 units) for all T steps, t = T-1 .. 0
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kCellThreads, 2) void cell_bwd_stream_kernel(BwdStream p) {
  constexpr int BM = 64, BN = 64, NS = 4, TM = BM / 32, NIA = BM / 32;
  constexpr int STAGE = (BM + BN) * 128;
  __shared__ __attribute__((aligned(1024))) char st0[STAGE];
  __shared__ __attribute__((aligned(1024))) char st1[STAGE];
  __shared__ __attribute__((aligned(1024))) char st2[STAGE];
  __shared__ __attribute__((aligned(1024))) char st3[STAGE];
  __shared__ int s_word[4];
  const int id = stream_join(p.sync, s_word);
  if (id < 0) return;
  const int H = p.H, T = p.T;
  const int NY = H / BN, R = p.nrows / BM;
  const int xs = id & 0xffff, xcd = xs >> 6, slot = xs & 63;
  const int m = xcd + 8 * (slot >> 4), l = (slot >> 3) & 1, y = slot & 7;  // slot >> 5 = half: row tiles {0, 1} early, {2, 3} late
  if (m >= R || l >= p.L || y >= NY) return;
  const unsigned ep0 = (unsigned)(id >> 16) * (unsigned)kSeqEpochs;
  unsigned* f_own = p.sync + kSyncFlags + (m * 2 + l) * 8;
  const unsigned* f_above = p.sync + kSyncFlags + (m * 2 + (l + 1 < p.L ? l + 1 : l)) * 8;
  const BwdStreamLayer& Lr = p.layer[l];
  const bool has_above = l + 1 < p.L;
  const int m0 = m * BM, n0 = y * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 15, gq = lane >> 4;
  const unsigned uH = (unsigned)H;
  const int64_t BH = (int64_t)p.B * H;

  const unsigned c8 = (unsigned)((lane & 7) ^ (lane >> 3));
  const unsigned rowa0 = (unsigned)(wave * NIA * 8 + (lane >> 3));
  const unsigned rowb0 = (unsigned)(n0 + wave * 16 + (lane >> 3));
  auto swz = [](int row) { return (((row >> 2) & 3) << 2) ^ ((row >> 1) & 1); };

  const int chunk = threadIdx.x & 7, lr = threadIdx.x >> 3;  // epilogue items: (row lr of wave row k, 8 units)
  const unsigned u = n0 + chunk * 8;
  float dcreg[2][8];  // running dL/dc of this workgroup's tile
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) dcreg[k][e] = 0.f;

  if (slot >= 32)  // (see the forward kernel)
    for (int k = 0; k < p.stagger; ++k) __builtin_amdgcn_s_sleep(127);

  unsigned long long* tl = (p.tlog && m == 0 && y == 0 && threadIdx.x == 0) ? p.tlog + l * 256 : nullptr;
  for (int s_ = 0; s_ < T; ++s_) {
    const int t = T - 1 - s_;
    ST_TLOG(s_ * 8 + 0);
    // dg^l_{t+1} of every unit tile of this row tile (published as step s_) and dg^{l+1}_t (the layer above's step s_ + 1)
    if (s_ > 0 || has_above) {
      if (!stream_wait(p.sync, f_own, NY, s_ > 0 ? ep0 + (unsigned)s_ : 0u, f_above, NY, has_above ? ep0 + (unsigned)s_ + 1u : 0u)) return;
    }
    ST_TLOG(s_ * 8 + 1);
    const int end0 = seg_steps(Lr.seg[0], t), end1 = end0 + seg_steps(Lr.seg[1], t);
    const int nsteps = end1;
    f32x4 acc[TM][2];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto issue = [&](char* stg, int ks, int part) {
      const int s = (int)(ks >= end0);
      const int kl = ks - (ks >= end0 ? end0 : 0);
      const StreamSeg& S = Lr.seg[s];
      const unsigned la = (unsigned)(S.lda * 2), lb = (unsigned)(S.ldw * 2);
      const __amdgpu_buffer_rsrc_t a =
          __builtin_amdgcn_make_buffer_rsrc((u16*)S.A + (int64_t)t * S.a_tstride + (int64_t)m0 * S.lda, 0, (int)(BM * la), 0x00020000);
      const __amdgpu_buffer_rsrc_t b = __builtin_amdgcn_make_buffer_rsrc((u16*)S.W, 0, (int)(H * lb), 0x00020000);
      const unsigned oob = (unsigned)((int)(ks >= nsteps) | (int)(kl * kCellBK >= S.K)) << 30;
      const unsigned kb = ((unsigned)(kl * (kCellBK * 2)) + c8 * 16u) | oob;
      unsigned xa[NIA], xb[2];
#pragma unroll
      for (int q = 0; q < NIA; ++q) xa[q] = (rowa0 + (unsigned)(q * 8)) * la + kb;
#pragma unroll
      for (int q = 0; q < 2; ++q) xb[q] = (rowb0 + (unsigned)(q * 8)) * lb + kb;
      if (part != 1) cell_issue<NIA, kStreamSc1>(stg, a, xa, 0u, wave);  // dgates of this launch: past the L1
      if (part != 0) cell_issue<2>(stg + BM * 128, b, xb, 0u, wave);
    };
    cell_mainloop<u16, BM, BN, NS>(acc, nsteps, issue, st0, st1, st2, st3);
    ST_TLOG(s_ * 8 + 2);

    __syncthreads();
    {
      char* xw = wm ? st1 : st0;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int xr = tm * 16 + gq * 4 + r;
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            const int slot_ = wn * 8 + tn * 4 + (i >> 2);
            *(cell_lds_f)(xw + xr * 256 + ((slot_ ^ swz(xr)) << 4) + (i & 3) * 4) = acc[tm][tn][r];
          }
        }
    }
    __syncthreads();
    const u16* gates = Lr.gates + (int64_t)t * BH * 4;
    const float* c_cur = Lr.cs + (int64_t)t * BH;
    const bool has_cp = t > 0, has_e1 = Lr.ext != nullptr, has_e2 = Lr.ext2 != nullptr && t == T - 1;
    const float* cprev = has_cp ? c_cur - BH : c_cur;  // stand-ins keep the loads unconditional (masked below)
    const float* e1p = has_e1 ? Lr.ext + (int64_t)t * BH : c_cur;
    const float* e2p = has_e2 ? Lr.ext2 : c_cur;
    const unsigned e2ld = has_e2 ? (unsigned)Lr.ext2_ld : uH;
    u16* dg_out = Lr.dg + (int64_t)t * BH * 4;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const char* xr = (k ? st1 : st0) + lr * 256;
      const unsigned row = m0 + k * 32 + lr;
      const unsigned o = row * uH + u, o4 = row * 4u * uH + u;
      float dh[8], ig[8], fg[8], gg[8], og[8], cp[8], cc[8], e1[8], e2[8];
      unpack_bf8(*(const u32x4v*)(gates + o4), ig);
      unpack_bf8(*(const u32x4v*)(gates + o4 + uH), fg);
      unpack_bf8(*(const u32x4v*)(gates + o4 + 2 * uH), gg);
      unpack_bf8(*(const u32x4v*)(gates + o4 + 3 * uH), og);
      ld8(cprev + o, cp);
      ld8(c_cur + o, cc);
      ld8(e1p + o, e1);
      ld8(e2p + row * e2ld + u, e2);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const f32x4v v = *(cell_lds_f4)(xr + (((chunk * 2 + hh) ^ swz(lr)) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) dh[hh * 4 + e] = v[e];
      }
      float dp[4][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float d = dh[e];
        d += has_e1 ? e1[e] : 0.f;
        d += has_e2 ? e2[e] : 0.f;
        const float cpv = has_cp ? cp[e] : 0.f;
        const float tc = tanhf_(cc[e]);
        const float dc = d * og[e] * (1.f - tc * tc) + dcreg[k][e];
        const float d_o = d * tc;
        const float d_i = dc * gg[e], d_f = dc * cpv, d_g = dc * ig[e];
        dcreg[k][e] = dc * fg[e];
        dp[0][e] = d_i * ig[e] * (1.f - ig[e]);
        dp[1][e] = d_f * fg[e] * (1.f - fg[e]);
        dp[2][e] = d_g * (1.f - gg[e] * gg[e]);
        dp[3][e] = d_o * og[e] * (1.f - og[e]);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) st8_bf(dg_out + o4 + g * uH, dp[g]);
    }
    ST_TLOG(s_ * 8 + 3);
    stream_publish(f_own + y, ep0 + (unsigned)s_ + 1u);
    ST_TLOG(s_ * 8 + 4);
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static bool stream_device_ok() {
  static int ok = -1;
  if (ok < 0) {
    int dev = 0;
    hipDeviceProp_t pr;
    ok = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess)
      ok = (strncmp(pr.gcnArchName, "gfx950", 6) == 0 && pr.multiProcessorCount == 256) ? 1 : 0;
  }
  return ok == 1;
}

static bool mis16(const void* p) { return (((uintptr_t)p) & 15) != 0; }

bool stream_eligible(const fhvae_lstm_desc* d) {
  // Opt-in (FHVAE_STREAM=1).  Measured at configs[3] (B = 2048, H = 512, T = 20; tools/prof_stream.py): a forward step costs
  // 24 us (flag wait 4-6, contraction 15 = 16 k-steps of one 32-KB stage in flight at the ~0.9 us LDS-DMA latency, epilogue 3,
  // publish 0.5) against 25 us for a step LAUNCH of the same cells; the backward 42 us on the lower layer (64 k-steps of 16 KB:
  // 30 us) against 33 us per launch, because the top layer's workgroups run ahead and leave the lower layer's alone at the
  // per-workgroup rate.  Placing the two workgroups of a CU in different clusters and starting one half a step late (so that one
  // contraction at a time has the CU's intake) changed neither: a workgroup's contraction is bound by its own bytes in flight
  // (LDS: 64 KB per workgroup at two per CU), not by its neighbour.  The launch-per-step cells stay the default.
  const char* on = getenv("FHVAE_STREAM");
  if (!on || atoi(on) == 0) return false;
  if (d->dtype != FHVAE_BF16 || !d->lp) return false;
  if (d->L < 1 || d->L > 2 || d->H % 64 || d->H > 512 || d->H < 64) return false;
  if (d->B % 128 || d->I % 8 || d->Ic % 8 || d->T + 2 >= kSeqEpochs) return false;
  // 32-bit element / byte offsets inside one (l, t) slab and inside one operand tile
  if (d->B * 4 * d->H * 4 >= (1LL << 31) || 4 * d->H * (d->I + d->Ic + d->H) * 2 >= (1LL << 30)) return false;
  if (mis16(d->hs) || mis16(d->cs) || mis16(d->gates) || mis16(d->hn) || mis16(d->hs_top_f32) || mis16(d->lp)) return false;
  return stream_device_ok();
}

int stream_fwd(const fhvae_lstm_desc* d, const StreamWeights& w, hipStream_t st) {
  const int64_t B = d->B, T = d->T, I = d->I, Ic = d->Ic, H = d->H, K0 = I + Ic;
  const int L = d->L;
  if ((I > 0 && mis16(w.x)) || (Ic > 0 && mis16(w.xc))) return FHVAE_ERR_ALIGN;
  u16* hs = (u16*)d->hs;
  u16* gates = (u16*)d->gates;
  const double fl = 2.0 * T * 4 * H * ((double)K0 + H + (L > 1 ? 2.0 * H * (L - 1) : 0.0));  // per batch row
  for (int64_t r0 = 0; r0 < B; r0 += 2048) {
    const int64_t nrows = B - r0 < 2048 ? B - r0 : 2048;
    FwdStream p = {};
    p.sync = (unsigned*)d->lp;
    p.B = (int)B, p.H = (int)H, p.T = (int)T, p.L = L, p.nrows = (int)nrows;
    p.stagger = getenv("FHVAE_STREAM_STAGGER_F") ? atoi(getenv("FHVAE_STREAM_STAGGER_F")) : 0;
    p.tlog = getenv("FHVAE_CLUSTER_TLOG") ? (unsigned long long*)((char*)d->lp + FHVAE_LSTM_SYNC_BYTES * 3 / 4) : nullptr;
    for (int l = 0; l < L; ++l) {
      FwdStreamLayer& Y = p.layer[l];
      u16* hs_l = hs + ((int64_t)l * T * B + r0) * H;
      int s = 0;
      if (l == 0) {
        if (I > 0) Y.seg[s++] = StreamSeg{w.x + r0 * I, B * I, I, w.w_ih[0], K0, (int)I, 0, (int)T};
        if (Ic > 0) Y.seg[s++] = StreamSeg{w.xc + r0 * Ic, 0, Ic, w.w_ih[0] + I, K0, (int)Ic, 0, (int)T};
      } else {
        Y.seg[s++] = StreamSeg{hs + ((int64_t)(l - 1) * T * B + r0) * H, B * H, H, w.w_ih[l], H, (int)H, 0, (int)T};
      }
      Y.seg[s++] = StreamSeg{hs_l - B * H, B * H, H, w.w_hh[l], H, (int)H, 1, (int)T};
      for (; s < 3; ++s) Y.seg[s] = StreamSeg{nullptr, 0, 8, nullptr, 8, 0, 1, 0};  // never active
      Y.bias_a = d->b_ih[l];
      Y.bias_b = d->b_hh[l];
      Y.cs = d->cs + ((int64_t)l * T * B + r0) * H;
      Y.hs = hs_l;
      Y.gates = gates + ((int64_t)l * T * B + r0) * 4 * H;
      Y.hs_f32 = (l == L - 1 && d->hs_top_f32) ? d->hs_top_f32 + r0 * H : nullptr;
      Y.hn = d->hn ? d->hn + r0 * L * H + (int64_t)l * H : nullptr;
      Y.hn_ld = (int64_t)L * H;
    }
    const int ts = trace_begin(st, kTraceFwdCell, fl * nrows);
    hipLaunchKernelGGL(cell_fwd_stream_kernel, dim3(kStreamGrid), dim3(kCellThreads), 0, st, p);
    trace_end(st, ts);
    const int e = fh_launch_status();
    if (e) return e;
  }
  return FHVAE_OK;
}

int stream_bwd(const fhvae_lstm_bwd_desc* bd, const StreamWeights& w, hipStream_t st) {
  const fhvae_lstm_desc* d = &bd->f;
  const int64_t B = d->B, T = d->T, H = d->H;
  const int L = d->L;
  if (mis16(bd->dgates) || mis16(bd->d_hs_top) || mis16(bd->d_hn)) return FHVAE_ERR_ALIGN;
  u16* dg = (u16*)bd->dgates;
  const u16* gates = (const u16*)d->gates;
  const double fl = 2.0 * H * 4 * H * ((double)(T - 1) * L + (double)T * (L - 1));  // per batch row
  for (int64_t r0 = 0; r0 < B; r0 += 2048) {
    const int64_t nrows = B - r0 < 2048 ? B - r0 : 2048;
    BwdStream p = {};
    p.sync = (unsigned*)d->lp;
    p.B = (int)B, p.H = (int)H, p.T = (int)T, p.L = L, p.nrows = (int)nrows;
    p.stagger = getenv("FHVAE_STREAM_STAGGER_B") ? atoi(getenv("FHVAE_STREAM_STAGGER_B")) : 0;
    p.tlog = getenv("FHVAE_CLUSTER_TLOG") ? (unsigned long long*)((char*)d->lp + FHVAE_LSTM_SYNC_BYTES * 3 / 4) : nullptr;
    for (int l = 0; l < L; ++l) {
      BwdStreamLayer& Y = p.layer[l];
      u16* dg_l = dg + ((int64_t)l * T * B + r0) * 4 * H;
      Y.seg[0] = StreamSeg{dg_l + B * 4 * H, B * 4 * H, 4 * H, w.w_hh_t[l], 4 * H, (int)(4 * H), 0, (int)T - 2};
      if (l + 1 < L)
        Y.seg[1] = StreamSeg{dg + ((int64_t)(l + 1) * T * B + r0) * 4 * H, B * 4 * H, 4 * H, w.w_ih_t[l + 1], 4 * H, (int)(4 * H), 0, (int)T};
      else
        Y.seg[1] = StreamSeg{nullptr, 0, 8, nullptr, 8, 0, 1, 0};
      Y.ext = (l == L - 1 && bd->d_hs_top) ? bd->d_hs_top + r0 * H : nullptr;
      Y.ext2 = bd->d_hn ? bd->d_hn + r0 * L * H + (int64_t)l * H : nullptr;
      Y.ext2_ld = (int64_t)L * H;
      Y.gates = gates + ((int64_t)l * T * B + r0) * 4 * H;
      Y.cs = d->cs + ((int64_t)l * T * B + r0) * H;
      Y.dg = dg_l;
    }
    const int ts = trace_begin(st, kTraceBwdCell, fl * nrows);
    hipLaunchKernelGGL(cell_bwd_stream_kernel, dim3(kStreamGrid), dim3(kCellThreads), 0, st, p);
    trace_end(st, ts);
    const int e = fh_launch_status();
    if (e) return e;
  }
  return FHVAE_OK;
}

}  // namespace fh
