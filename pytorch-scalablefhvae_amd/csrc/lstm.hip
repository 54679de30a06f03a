// lstm.hip -- K1: multi-layer LSTM over a whole segment with step-fused cells.
//
// No reference body exists (fhvae.py:14 raises NotImplementedError); semantics = torch.nn.LSTM CPU
// (gate order i,f,g,o).  These kernels stand where the FC pre-encoders / pre-decoder stand in
// simple_fhvae.py:160-164, :186-190, :240-244.
//
// Schedule (see DESIGN.md "K1"): the recurrence needs every hidden unit of step t-1 before step t,
// i.e. an all-to-all between workgroups per step.  On gfx950 a dependent kernel boundary (~1.5 us)
// is cheaper than any in-launch grid barrier (4-7 us, MI355X_MICROARCH.md price list), so each
// WAVEFRONT step (layer l at time t together with layer l-1 at time t+1) is one launch of
// (B/64) x (H/16) x jobs workgroups, and the whole sequence is meant to be replayed from a hipGraph.
// Inside a launch one workgroup computes the four gate pre-activations of 64 segments x 16 hidden
// units on MFMA (virtual column order: gate-major inside the tile, so one lane owns i,f,g,o of the
// same (segment, unit)) and applies sigmoid/tanh + the cell update in registers: the (B,4H)
// pre-activations never touch HBM.
#include "gemm_launch.h"
#include "lstm_cell.h"
#include "lstm_cluster.h"
#include "wgrad.h"
#include "wgrad_f32.h"
#include <algorithm>
#include <vector>
#include "trace.h"
#include <cstdlib>

namespace fh {

constexpr int kCH = 32;  // 512-byte panels: the step GEMMs are latency-bound, pay the latency once per panel

// virtual column n of the gate matrix -> physical weight row: tiles of 64 = 4 gates x 16 units
struct GateRowMap {
  int H;
  __device__ __forceinline__ int64_t operator()(int n) const {
    int unit = (n >> 6) * 16 + (n & 15);
    int gate = (n & 63) >> 4;
    return unit < H ? (int64_t)gate * H + unit : -1;
  }
  __device__ __forceinline__ bool all_valid(int n0, int n) const { return ((n0 + n) >> 6) * 16 <= H; }
};

template <typename T>
__device__ __forceinline__ void store_h(T* p, float v);
template <>
__device__ __forceinline__ void store_h<float>(float* p, float v) {
  *p = v;
}
template <>
__device__ __forceinline__ void store_h<u16>(u16* p, float v) {
  *p = f2bf(v);
}

template <typename T>
__device__ __forceinline__ float load_h(const T* p);
template <>
__device__ __forceinline__ float load_h<float>(const float* p) {
  return *p;
}
template <>
__device__ __forceinline__ float load_h<u16>(const u16* p) {
  return bf2f(*p);
}

// Tile shapes: <64,64,4,1> (more workgroups, epilogue operands prefetched) and <128,128,2,2> (very large
// batches only: measured at B = 2048 the 64x64 tiles on 4x more workgroups are 1.2-1.5x faster).
// In both, one wave owns whole 64-column groups (4 gates x 16 units), so i,f,g,o of a (row, unit) sit in
// the 4 accumulators acc[tm][0..3] of one lane.
template <typename T, int BM, int BN, int WM, int WN, int CH>
__global__ __launch_bounds__(kThreads) void lstm_fwd_step_kernel(FwdJobs<T> jobs) {
  using TL = Tile<T, BM, BN, WM, WN, CH>;
  constexpr int TM = TL::TM;
  static_assert(TL::TN == 4, "one wave = one 64-column gate group");
  constexpr bool kPrefetch = TM == 1;
  constexpr int NBUF = CH >= 32 ? 2 : 1;  // wide panels = few large workgroups: double buffer; narrow: occupancy
  using GT = GldsTile<T, BM, BN, WM, WN, CH, NBUF>;
  __shared__ __attribute__((aligned(16))) char smem[TL::SMEM > GT::SMEM ? TL::SMEM : GT::SMEM];
  const FwdJob<T>& J = jobs.job[blockIdx.z];
  const int B = jobs.B, H = jobs.H;
  // blockIdx.x walks the ROW tiles: workgroups are dealt round-robin over the 8 XCDs by linear id, so every XCD
  // (private 4 MB L2) sees 1/8 of the activations and all of the (small) weight slice, instead of all activations
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  f32x4 acc[TM][4];
  zero_acc(acc);
  RowIdent arm{B};
  GateRowMap brm{H};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int unit = (n0 / 64 + wn) * 16 + (lane & 15);
  const bool uok = unit < H;
  auto row_of = [&](int tm, int r) { return m0 + wm * (TM * 16) + tm * 16 + (lane >> 4) * 4 + r; };
  auto fetch_add = [&](int row, int g) -> float {
    float v = 0.f;
    if (J.pre) v = J.pre[(int64_t)row * J.pre_ld + g * H + unit];
    if (J.bias_a) v += J.bias_a[g * H + unit] + J.bias_b[g * H + unit];
    return v;
  };
  // small tile: epilogue operands are fetched BEFORE the contraction so their latency hides under it
  float padd[kPrefetch ? 4 : 1][4], cprev[kPrefetch ? 4 : 1];
  if constexpr (kPrefetch) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row_of(0, r);
      const bool ok = uok && row < B;
#pragma unroll
      for (int g = 0; g < 4; ++g) padd[r][g] = ok ? fetch_add(row, g) : 0.f;
      cprev[r] = (ok && J.c_prev) ? J.c_prev[(int64_t)row * H + unit] : 0.f;
    }
  }
  const int nkb = num_kblocks<T, CH>(J.seg);
  // interior tiles with panel-aligned K take the LDS-DMA path; edges / odd shapes the register-staged one
  const bool dma = jobs.glds && m0 + BM <= B && brm.all_valid(n0, BN) && seg_glds_ok<T>(J.seg[0], TL::BK) &&
                   seg_glds_ok<T>(J.seg[1], TL::BK);
  if (dma)
    mainloop_glds<T, BM, BN, WM, WN, CH, NBUF, false>(acc, J.seg, m0, n0, arm, brm, smem);
  else
    mainloop<T, BM, BN, WM, WN, CH, true, true, false>(acc, J.seg, m0, B, n0, (int)gridDim.y * BN, arm, brm, 0, nkb, smem);

  if (!uok) return;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row_of(tm, r);
      if (row >= B) continue;
      float pa[4], cp;
      if constexpr (kPrefetch) {
#pragma unroll
        for (int g = 0; g < 4; ++g) pa[g] = padd[r][g];
        cp = cprev[r];
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) pa[g] = fetch_add(row, g);
        cp = J.c_prev ? J.c_prev[(int64_t)row * H + unit] : 0.f;
      }
      const float ig = sigmoidf_(acc[tm][0][r] + pa[0]), fg = sigmoidf_(acc[tm][1][r] + pa[1]);
      const float gg = tanhf_(acc[tm][2][r] + pa[2]), og = sigmoidf_(acc[tm][3][r] + pa[3]);
      const float c = __builtin_fmaf(fg, cp, ig * gg);
      const float h = og * tanhf_(c);
      J.c_out[(int64_t)row * H + unit] = c;
      store_h<T>(J.h_out + (int64_t)row * H + unit, h);
      if (J.h_out_f32) J.h_out_f32[(int64_t)row * H + unit] = h;
      T* go = J.gates_out + (int64_t)row * 4 * H + unit;
      store_h<T>(go, ig);
      store_h<T>(go + H, fg);
      store_h<T>(go + 2 * H, gg);
      store_h<T>(go + 3 * H, og);
      if (J.hn_out) J.hn_out[(int64_t)row * J.hn_ld + unit] = h;
    }
}

// Tile shapes: <32,32,2,2> and <128,64,4,1> (very large batches only, as for the forward cell).
template <typename T, int BM, int BN, int WM, int WN, int CH>
__global__ __launch_bounds__(kThreads) void lstm_bwd_step_kernel(BwdJobs<T> jobs) {
  using TL = Tile<T, BM, BN, WM, WN, CH>;
  constexpr int TM = TL::TM, TN = TL::TN;
  constexpr int NBUF = CH >= 32 ? 2 : 1;
  using GT = GldsTile<T, BM, BN, WM, WN, CH, NBUF>;
  __shared__ __attribute__((aligned(16))) char smem[GT::SMEM > TL::SMEM ? GT::SMEM : TL::SMEM];
  const BwdJob<T>& J = jobs.job[blockIdx.z];
  const int B = jobs.B, H = jobs.H;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;  // row tiles on x: XCD-local activations (see the forward cell)
  f32x4 acc[TM][TN];
  zero_acc(acc);
  RowIdent arm{B}, brm{H};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int nkb = num_kblocks<T, CH>(J.seg);
  // W: the transposed copy [H,4H] (KC operand: bf16 always, f32 when the caller gave the forward a workspace) -> LDS-DMA path;
  // else (f32) the untransposed master weight as a KM operand
  const bool wkc = (J.seg[0].K == 0 || J.seg[0].b_kc) && (J.seg[1].K == 0 || J.seg[1].b_kc);
  const bool dma = wkc && jobs.glds && m0 + BM <= B && n0 + BN <= H && seg_glds_ok<T>(J.seg[0], TL::BK) && seg_glds_ok<T>(J.seg[1], TL::BK);
  if (dma) {
    mainloop_glds<T, BM, BN, WM, WN, CH, NBUF, false>(acc, J.seg, m0, n0, arm, brm, smem);
  } else if (wkc) {
    mainloop<T, BM, BN, WM, WN, CH, true, true, false>(acc, J.seg, m0, B, n0, H, arm, brm, 0, nkb, smem);
  } else {
    if constexpr (sizeof(T) == 4) mainloop<T, BM, BN, WM, WN, CH, true, false, false>(acc, J.seg, m0, B, n0, H, arm, brm, 0, nkb, smem);
  }

#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int unit = n0 + wn * (TN * 16) + tn * 16 + (lane & 15);
    if (unit >= H) continue;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * (TM * 16) + tm * 16 + (lane >> 4) * 4 + r;
        if (row >= B) continue;
        const int64_t o = (int64_t)row * H + unit;
        float dh = acc[tm][tn][r];
        if (J.ext) dh += J.ext[(int64_t)row * J.ext_ld + unit];
        if (J.ext2) dh += J.ext2[(int64_t)row * J.ext2_ld + unit];
        const T* gp = J.gates + (int64_t)row * 4 * H + unit;
        const float ig = load_h<T>(gp), fg = load_h<T>(gp + H), gg = load_h<T>(gp + 2 * H), og = load_h<T>(gp + 3 * H);
        const float cp = J.c_prev ? J.c_prev[o] : 0.f;
        const float tc = tanhf_(J.c_cur[o]);
        float dc = dh * og * (1.f - tc * tc);
        if (!J.first) dc += J.dc[o];
        const float d_o = dh * tc;
        const float d_i = dc * gg, d_f = dc * cp, d_g = dc * ig;
        J.dc[o] = dc * fg;
        float dp[4];
        dp[0] = d_i * ig * (1.f - ig);
        dp[1] = d_f * fg * (1.f - fg);
        dp[2] = d_g * (1.f - gg * gg);
        dp[3] = d_o * og * (1.f - og);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int64_t go = (int64_t)row * 4 * H + g * H + unit;
          store_h<T>(J.dg_out + go, dp[g]);
          if (J.dgsum) J.dgsum[go] = J.first ? dp[g] : J.dgsum[go] + dp[g];
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------
// batched f32 -> bf16 cast (+ optional transposed copy): one launch for all operands of a net
// ---------------------------------------------------------------------------------------------
struct CastItem {
  const float* src;
  u16* dst;    // [R,C] or NULL
  u16* dst_t;  // [C,R] (row stride ld_t) or NULL
  int64_t R, C;
  int64_t ld_t;  // row stride of dst_t (0: R)
  int64_t pad_t; // columns [R, R + pad_t) of every dst_t row are zeroed (the head's [W_mu^T | W_lv^T | 0])
};
struct CastBatch {
  unsigned* sync;  // the workspace's sync block (lstm_cluster.h): cleared here, once per forward
  int* sticky;     // fhvae_lstm_desc.sticky_status: its address is left in the block for cluster_give_up
  int n;
  CastItem it[4 * FHVAE_MAX_LAYERS + 4];
};
// 32x32 tiles: coalesced f32 reads, coalesced bf16 writes of the straight copy, and the transposed copy through an LDS tile
// (the element-wise version wrote the transpose as 2-byte scatters: 13-15 us per net, now ~4)
__global__ __launch_bounds__(256) void cast_batch_kernel(CastBatch cb) {
  __shared__ u16 tile[32][33];
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    // words kSyncSticky, +1 carry the address of the caller's sticky status word: a launch that gives up ORs its code into it
    // as well (cluster_give_up), so the failure stays visible after this block is re-armed by the next forward
    for (int i = threadIdx.x; i < kSyncWordsUsed; i += blockDim.x)
      cb.sync[i] = i == kSyncSticky ? (unsigned)((uintptr_t)cb.sticky & 0xffffffffu) : i == kSyncSticky + 1 ? (unsigned)((uintptr_t)cb.sticky >> 32) : 0u;
  }
  const CastItem& c = cb.it[blockIdx.y];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t tr = (c.R + 31) / 32, tc = (c.C + 31) / 32;
  for (int64_t t = blockIdx.x; t < tr * tc; t += gridDim.x) {
    const int64_t r0 = (t / tc) * 32, c0 = (t % tc) * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t r = r0 + ty + 8 * k, cc = c0 + tx;
      u16 v = 0;
      if (r < c.R && cc < c.C) {
        v = f2bf(c.src[r * c.C + cc]);
        if (c.dst) c.dst[r * c.C + cc] = v;
      }
      tile[ty + 8 * k][tx] = v;
    }
    if (c.dst_t) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t cc = c0 + ty + 8 * k, r = r0 + tx;
        if (cc < c.C && r < c.R) c.dst_t[cc * (c.ld_t ? c.ld_t : c.R) + r] = tile[tx][ty + 8 * k];
        if (c.pad_t > 0 && r0 == 0 && cc < c.C)
          for (int64_t j = tx; j < c.pad_t; j += 32) c.dst_t[cc * c.ld_t + c.R + j] = 0;
      }
      __syncthreads();
    }
  }
}

// f32 [R,C] -> f32 [C,R] through a 32x32 LDS tile (the transposed weights of the f32 backward cells), one item per blockIdx.y
struct TransItem {
  const float* src;
  float* dst_t;
  int64_t R, C;
};
struct TransBatch {
  int n;
  TransItem it[2 * FHVAE_MAX_LAYERS];
};
__global__ __launch_bounds__(256) void transpose_f32_kernel(TransBatch tb) {
  __shared__ float tile[32][33];
  const TransItem& c = tb.it[blockIdx.y];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t tr = (c.R + 31) / 32, tc = (c.C + 31) / 32;
  for (int64_t t = blockIdx.x; t < tr * tc; t += gridDim.x) {
    const int64_t r0 = (t / tc) * 32, c0 = (t % tc) * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t r = r0 + ty + 8 * k, cc = c0 + tx;
      tile[ty + 8 * k][tx] = (r < c.R && cc < c.C) ? c.src[r * c.C + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t cc = c0 + ty + 8 * k, r = r0 + tx;
      if (cc < c.C && r < c.R) c.dst_t[cc * c.R + r] = tile[tx][ty + 8 * k];
    }
    __syncthreads();
  }
}

__global__ void zero_f32_kernel(float* p, int64_t n) {
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4)
    for (int k = 0; k < 4 && i + k < n; ++k) p[i + k] = 0.f;
}

}  // namespace fh

using namespace fh;

static int check_desc(const fhvae_lstm_desc* d) {
  FH_CHECK_PTR(d);
  if (d->dtype != FHVAE_F32 && d->dtype != FHVAE_BF16) return FHVAE_ERR_DTYPE;
  if (d->L < 1 || d->L > FHVAE_MAX_LAYERS) return FHVAE_ERR_SHAPE;
  FH_CHECK_POS(d->B);
  FH_CHECK_POS(d->T);
  FH_CHECK_POS(d->H);
  if (d->I < 0 || d->Ic < 0 || d->I + d->Ic <= 0) return FHVAE_ERR_SHAPE;
  FH_CHECK_I32(d->B * d->T);
  FH_CHECK_I32(4 * d->H);
  FH_CHECK_I32(d->I + d->Ic);
  if (d->I > 0) FH_CHECK_PTR(d->x);
  if (d->Ic > 0) FH_CHECK_PTR(d->xc);
  for (int l = 0; l < d->L; ++l) {
    FH_CHECK_PTR(d->w_ih[l]);
    FH_CHECK_PTR(d->w_hh[l]);
    FH_CHECK_PTR(d->b_ih[l]);
    FH_CHECK_PTR(d->b_hh[l]);
  }
  FH_CHECK_PTR(d->hs);
  FH_CHECK_PTR(d->cs);
  FH_CHECK_PTR(d->gates);
  FH_CHECK_PTR(d->pre);
  // the step cells stage h / W rows in 16-byte chunks: H must be a multiple of 4 (f32) / 8 (bf16)
  if (d->dtype == FHVAE_BF16) {
    FH_CHECK_PTR(d->lp);
    if (d->H % 8) return FHVAE_ERR_ALIGN;
  } else {
    if (d->H % 4) return FHVAE_ERR_ALIGN;
    if (d->hs_top_f32) return FHVAE_ERR_SHAPE;
  }
  if ((((uintptr_t)d->hs) | ((uintptr_t)d->cs) | ((uintptr_t)d->gates)) & 15) return FHVAE_ERR_ALIGN;
  for (int l = 0; l < d->L; ++l)
    if ((((uintptr_t)d->w_ih[l]) | ((uintptr_t)d->w_hh[l])) & 15) return FHVAE_ERR_ALIGN;
  if (d->head_w_mu) {  // the head's stacked operands ride in the operand-cast launch: bf16 mode only
    if (d->dtype != FHVAE_BF16) return FHVAE_ERR_DTYPE;
    FH_CHECK_PTR(d->head_w_lv);
    if (!d->head_wl && !d->head_wt) return FHVAE_ERR_NULL;
    FH_CHECK_POS(d->head_D);
    FH_CHECK_POS(d->head_K);
    if (d->head_wt && d->head_ldt < 2 * d->head_D) return FHVAE_ERR_SHAPE;
  }
  return FHVAE_OK;
}

// layout of the bf16 workspace (element offsets, every block a multiple of 8 elements)
struct LpLayout {
  int64_t x, xc, w_ih[FHVAE_MAX_LAYERS], w_hh[FHVAE_MAX_LAYERS], w_ih_t[FHVAE_MAX_LAYERS], w_hh_t[FHVAE_MAX_LAYERS], xch, total;
};
static LpLayout lp_layout(const fhvae_lstm_desc* d) {
  LpLayout o;
  int64_t n = FHVAE_LSTM_SYNC_BYTES / 2;  // the cluster kernels' sync block comes first (lstm_cluster.h)
  auto take = [&](int64_t cnt) {
    int64_t at = n;
    n += (cnt + 7) / 8 * 8;
    return at;
  };
  o.x = take(d->T * d->B * d->I);
  o.xc = take(d->B * d->Ic);
  for (int l = 0; l < d->L; ++l) {
    const int64_t kin = l == 0 ? d->I + d->Ic : d->H;
    o.w_ih[l] = take(4 * d->H * kin);
    o.w_hh[l] = take(4 * d->H * d->H);
    o.w_ih_t[l] = take(l == 0 ? 0 : 4 * d->H * kin);
    o.w_hh_t[l] = take(4 * d->H * d->H);
  }
  {  // the persistent kernels' exchange buffer: blocked copies of h / dg (lstm_cluster.hip, xch_off) or the partial-dh slots of
     // the per-layer backward at H = 256 (lstm_bwd_rs.hip: a fixed kRsXchElems whatever the batch)
    int64_t cnt = 2 * (int64_t)d->L * d->B * 4 * d->H;
    if (d->H == 256 && cnt < fh::kRsXchElems) cnt = fh::kRsXchElems;
    o.xch = take(cnt);
  }
  o.total = n;
  return o;
}

// f32 mode: the optional workspace holds the TRANSPOSED f32 weights [H,4H] the backward cells multiply by (float offsets; the
// sync block's bytes are skipped so that both modes keep the same head)
struct Lp32Layout {
  int64_t w_ih_t[FHVAE_MAX_LAYERS], w_hh_t[FHVAE_MAX_LAYERS], total;
};
static Lp32Layout lp32_layout(const fhvae_lstm_desc* d) {
  Lp32Layout o;
  int64_t n = FHVAE_LSTM_SYNC_BYTES / 4;
  auto take = [&](int64_t cnt) {
    int64_t at = n;
    n += (cnt + 3) / 4 * 4;
    return at;
  };
  for (int l = 0; l < d->L; ++l) {
    o.w_ih_t[l] = take(l == 0 ? 0 : 4 * d->H * d->H);
    o.w_hh_t[l] = take(4 * d->H * d->H);
  }
  o.total = n;
  return o;
}

extern "C" int64_t fhvae_lstm_lp_bytes(const fhvae_lstm_desc* d) {
  if (!d || d->L < 1 || d->L > FHVAE_MAX_LAYERS) return 0;
  if (d->dtype == FHVAE_F32) return lp32_layout(d).total * 4;
  if (d->dtype != FHVAE_BF16) return 0;
  return lp_layout(d).total * 2;
}

// operand views for one dtype
template <typename T>
struct Ops {
  const T* x;
  const T* xc;
  const T* w_ih[FHVAE_MAX_LAYERS];
  const T* w_hh[FHVAE_MAX_LAYERS];
  const T* w_ih_t[FHVAE_MAX_LAYERS];  // bf16 only
  const T* w_hh_t[FHVAE_MAX_LAYERS];
};
static Ops<float> ops_f32(const fhvae_lstm_desc* d) {
  Ops<float> o = {};
  o.x = d->x;
  o.xc = d->xc;
  for (int l = 0; l < d->L; ++l) {
    o.w_ih[l] = d->w_ih[l];
    o.w_hh[l] = d->w_hh[l];
  }
  if (d->lp) {  // transposed copies (filled by the forward): the backward cells then take the LDS-DMA KC/KC path
    const Lp32Layout Y = lp32_layout(d);
    const float* base = (const float*)d->lp;
    for (int l = 0; l < d->L; ++l) {
      o.w_ih_t[l] = l > 0 ? base + Y.w_ih_t[l] : nullptr;
      o.w_hh_t[l] = base + Y.w_hh_t[l];
    }
  }
  return o;
}
static Ops<u16> ops_bf16(const fhvae_lstm_desc* d) {
  Ops<u16> o = {};
  LpLayout L = lp_layout(d);
  const u16* base = (const u16*)d->lp;
  o.x = d->x_lp ? (const u16*)d->x_lp : base + L.x;
  o.xc = base + L.xc;
  for (int l = 0; l < d->L; ++l) {
    o.w_ih[l] = base + L.w_ih[l];
    o.w_hh[l] = base + L.w_hh[l];
    o.w_ih_t[l] = base + L.w_ih_t[l];
    o.w_hh_t[l] = base + L.w_hh_t[l];
  }
  return o;
}

static int cast_operands(const fhvae_lstm_desc* d, hipStream_t st) {
  LpLayout L = lp_layout(d);
  u16* base = (u16*)d->lp;
  CastBatch cb = {};
  cb.sync = (unsigned*)d->lp;
  cb.sticky = d->sticky_status;
  auto add = [&](const float* s, u16* dst, u16* dst_t, int64_t R, int64_t C) {
    if (R * C > 0) cb.it[cb.n++] = CastItem{s, dst, dst_t, R, C, 0, 0};
  };
  if (!d->x_lp) add(d->x, base + L.x, nullptr, d->T * d->B, d->I);
  add(d->xc, base + L.xc, nullptr, d->B, d->Ic);
  for (int l = 0; l < d->L; ++l) {
    const int64_t kin = l == 0 ? d->I + d->Ic : d->H;
    add(d->w_ih[l], base + L.w_ih[l], l == 0 ? nullptr : base + L.w_ih_t[l], 4 * d->H, kin);
    add(d->w_hh[l], base + L.w_hh[l], base + L.w_hh_t[l], 4 * d->H, d->H);
  }
  if (d->head_w_mu) {  // the stacked operands of the Gaussian head behind this net (fhvae_lstm_desc.head_*)
    const int64_t D = d->head_D, K = d->head_K, ldt = d->head_ldt;
    u16* wl = (u16*)d->head_wl;
    u16* wt = (u16*)d->head_wt;
    cb.it[cb.n++] = CastItem{d->head_w_mu, wl, wt, D, K, ldt, 0};
    cb.it[cb.n++] = CastItem{d->head_w_lv, wl ? wl + D * K : nullptr, wt ? wt + D : nullptr, D, K, ldt, wt ? ldt - 2 * D : 0};
  }
  hipLaunchKernelGGL(cast_batch_kernel, dim3(256, (unsigned)cb.n), dim3(256), 0, st, cb);
  return fh_launch_status();
}

// the large-tile cells (lstm_cell.hip) take a wavefront step once it offers them about a workgroup per CU (f32, whose
// generic cells are further from their roofline: from 256 tiles per launch); FHVAE_BIG_CELLS=0/1 overrides
static bool big_cells(int64_t B, int64_t H, int dtype = FHVAE_BF16) {
  const char* ev = getenv("FHVAE_BIG_CELLS");  // read per call: the tests flip it
  const int env = ev ? atoi(ev) : -1;
  if (env >= 0) return env != 0;
  return (B / 128) * (H / 64) >= (dtype == FHVAE_F32 ? 64 : 96);
}

// the forward jobs of wavefront step w; `big`: for the large-tile cells, which multiply layer 0's input themselves (no `pre`)
template <typename T>
static FwdJobs<T> fwd_jobs(const fhvae_lstm_desc* d, const Ops<T>& op, int64_t w, bool big, int& nj) {
  const int64_t B = d->B, T_ = d->T, I = d->I, Ic = d->Ic, H = d->H, K0 = I + Ic;
  const int L = d->L;
  const int64_t pre_tstride = I > 0 ? B * 4 * H : 0;
  T* hs = (T*)d->hs;
  const T* w0 = op.w_ih[0];
  FwdJobs<T> jobs = {};
  jobs.B = (int)B;
  jobs.H = (int)H;
  jobs.glds = 1;
  nj = 0;
  for (int l = 0; l < L; ++l) {
    const int64_t t = w - l;
    if (t < 0 || t >= T_) continue;
    FwdJob<T>& J = jobs.job[nj++];
    const int64_t lt = (int64_t)l * T_ + t;
    if (l > 0) J.seg[0] = Seg{hs + ((int64_t)(l - 1) * T_ + t) * B * H, H, 1, op.w_ih[l], H, 1, (int)H, 0};
    if (t > 0) J.seg[1] = Seg{hs + (lt - 1) * B * H, H, 1, op.w_hh[l], H, 1, (int)H, 0};
    if (l == 0 && !big) {
      J.pre = d->pre + t * pre_tstride;
      J.pre_ld = 4 * H;
    } else {
      J.bias_a = d->b_ih[l];
      J.bias_b = d->b_hh[l];
    }
    if (l == 0 && big) {
      if (I > 0) J.xseg[0] = Seg{op.x + t * B * I, I, 1, w0, K0, 1, (int)I, 0};
      if (Ic > 0) J.xseg[1] = Seg{op.xc, Ic, 1, w0 + I, K0, 1, (int)Ic, 0};
    }
    J.c_prev = t > 0 ? d->cs + (lt - 1) * B * H : nullptr;
    J.c_out = d->cs + lt * B * H;
    J.h_out = hs + lt * B * H;
    if (l == L - 1 && d->hs_top_f32) J.h_out_f32 = d->hs_top_f32 + t * B * H;
    J.gates_out = (T*)d->gates + lt * B * 4 * H;
    if (d->hn && t == T_ - 1) {
      J.hn_out = d->hn + (int64_t)l * H;
      J.hn_ld = (int64_t)L * H;
    }
  }
  return jobs;
}

// every wavefront step of the sequence meets the large-tile cells' preconditions
template <typename T>
static bool cell_fwd_plan_ok(const fhvae_lstm_desc* d, const Ops<T>& op) {
  for (int64_t w = 0; w < d->T + d->L - 1; ++w) {
    int nj = 0;
    const FwdJobs<T> jobs = fwd_jobs<T>(d, op, w, true, nj);
    if (!cell_fwd_big_ok(jobs, nj)) return false;
  }
  return true;
}

// the shape part of the large-tile cells' preconditions (what remains is 16-byte alignment of the caller's buffers)
static bool big_shape_ok(const fhvae_lstm_desc* d) {
  const int es = d->dtype == FHVAE_BF16 ? 2 : 4, epc = 16 / es;
  if (d->B % 128 || d->H % 64 || d->I % epc || d->Ic % epc) return false;
  if (d->B * 4 * d->H * 4 >= (1LL << 31) || (int64_t)d->L * d->H * d->B * 4 >= (1LL << 31)) return false;
  return 4 * d->H * (d->I + d->Ic > d->H ? d->I + d->Ic : d->H) * es < (1LL << 30);
}

template <typename T>
static int lstm_fwd_impl(const fhvae_lstm_desc* d, const Ops<T>& op, hipStream_t st) {
  const int64_t B = d->B, T_ = d->T, I = d->I, Ic = d->Ic, H = d->H;
  const int L = d->L;
  const int64_t K0 = I + Ic;
  const T* w0 = op.w_ih[0];
  // ---- layer-0 input projection (+ both biases): pre = [x_t || xc] . W_ih0^T + b_ih0 + b_hh0.  The persistent kernels
  //      multiply x_t themselves when they can (fold): then only the time-constant part is left for this GEMM
  bool cluster = false, fold = false, xc_in = false;
  if constexpr (sizeof(T) == 2) {
    cluster = cluster_eligible(d);
    fold = cluster && cluster_can_fold(d);
    xc_in = cluster && cluster_xc_in_kernel(d);
  }
  // large-tile step cells (lstm_cell.hip; all steps of the sequence or none): they multiply layer 0's input themselves
  const bool cell_big = !cluster && big_cells(B, H, d->dtype) && cell_fwd_plan_ok(d, op);
  // fhvae_lstm_pre_elems has promised the caller that `pre` is not needed for this shape
  if (!cluster && !cell_big && big_cells(B, H, d->dtype) && big_shape_ok(d)) return FHVAE_ERR_ALIGN;
  if (!cell_big && !(fold && Ic == 0) && !xc_in) {
    GemmParams p = {};
    int s = 0;
    if (I > 0 && !fold) p.seg[s++] = Seg{op.x, I, 1, w0, K0, 1, (int)I, 0};
    if (Ic > 0) p.seg[s++] = Seg{op.xc, Ic, 1, w0 + I, K0, 1, (int)Ic, (I > 0 && !fold) ? (int)B : 0};
    p.M = (int)((I > 0 && !fold) ? T_ * B : B);
    p.N = (int)(4 * H);
    p.C = d->pre;
    p.ldc = 4 * H;
    p.bias = d->b_ih[0];
    p.bias2 = d->b_hh[0];
    p.splitk = 1;
    int e = launch_gemm(p, d->dtype, st);
    if (e) return e;
  }
  if constexpr (sizeof(T) == 2) {
    if (cluster) {  // persistent form: the whole recurrence in one launch (lstm_cluster.hip)
      ClusterWeights cw = {};
      for (int l = 0; l < L; ++l) cw.w_ih[l] = op.w_ih[l], cw.w_hh[l] = op.w_hh[l], cw.w_ih_t[l] = op.w_ih_t[l], cw.w_hh_t[l] = op.w_hh_t[l];
      cw.xch = (u16*)d->lp + lp_layout(d).xch;
      cw.x_fold = fold ? (const u16*)op.x : nullptr;
      cw.xc_fold = xc_in ? (const u16*)op.xc : nullptr;
      return cluster_fwd(d, cw, st);
    }
  }
  if constexpr (sizeof(T) == 2) {
  }
  {
    if (cell_big) {  // large-tile cells (lstm_cell.hip)
      for (int64_t w = 0; w < T_ + L - 1; ++w) {
        int nj = 0;
        const FwdJobs<T> jobs = fwd_jobs(d, op, w, true, nj);
        double fl = 0;
        for (int j = 0; j < nj; ++j)
          fl += 2.0 * B * 4 * H * (jobs.job[j].seg[0].K + jobs.job[j].seg[1].K + jobs.job[j].xseg[0].K + jobs.job[j].xseg[1].K);
        const int ts = trace_begin(st, kTraceFwdCell, fl);
        const int e = launch_cell_fwd_big(jobs, nj, st);
        trace_end(st, ts);
        if (e) return e;
      }
      return FHVAE_OK;
    }
  }
  // ---- wavefront over (layer, time)
  for (int64_t w = 0; w < T_ + L - 1; ++w) {
    int nj = 0;
    const FwdJobs<T> jobs = fwd_jobs(d, op, w, false, nj);
    double fl = 0;
    for (int j = 0; j < nj; ++j) fl += 2.0 * B * 4 * H * (jobs.job[j].seg[0].K + jobs.job[j].seg[1].K);
    const int ts = trace_begin(st, kTraceFwdCell, fl);
    // large tiles (half the L2 -> LDS operand bytes per FLOP) only pay once they still give >= 2 workgroups per CU (see
    // gemm.hip): B >= 16384 at H = 256, B >= 2048 at H = 512 (configs[3]: 2048 workgroups of 64x64 pulled 14 TB/s from L2)
    // (measured at B = 2048, H = 512, bf16: 128x128 tiles 1.3-1.8 ms per net forward against 1.0-1.3 ms with 64x64: the
    //  heuristic stays "B >= 16384")
    const bool big_fwd = B >= 16384;
    if (big_fwd) {
      dim3 grid((unsigned)fh_cdiv(B, 128), (unsigned)fh_cdiv(H, 32), (unsigned)nj);
      hipLaunchKernelGGL((lstm_fwd_step_kernel<T, 128, 128, 2, 2, 16>), grid, dim3(kThreads), 0, st, jobs);
    } else if (B >= 1024) {
      // many workgroups per CU: 256-byte panels (32 KB LDS) so 2 workgroups per CU keep twice the bytes in flight
      dim3 grid((unsigned)fh_cdiv(B, 64), (unsigned)fh_cdiv(H, 16), (unsigned)nj);
      hipLaunchKernelGGL((lstm_fwd_step_kernel<T, 64, 64, 4, 1, 16>), grid, dim3(kThreads), 0, st, jobs);
    } else {
      dim3 grid((unsigned)fh_cdiv(B, 64), (unsigned)fh_cdiv(H, 16), (unsigned)nj);
      hipLaunchKernelGGL((lstm_fwd_step_kernel<T, 64, 64, 4, 1, kCH>), grid, dim3(kThreads), 0, st, jobs);
    }
    trace_end(st, ts);
    int e = fh_launch_status();
    if (e) return e;
  }
  return FHVAE_OK;
}

extern "C" int fhvae_lstm_form(const fhvae_lstm_desc* d) {
  if (!d || check_desc(d) != FHVAE_OK || !cluster_eligible(d)) return 0;
  return cluster_form(d);
}

// What decides the layouts of the tensors a forward saves for its backward (gates, the schedule-specific workspaces): the schedule
// (per-step cells, large-tile cells, persistent rows / contraction-split form) and its variants (register-stationary forward:
// unit-major gates).  The schedule is re-derived per call from the descriptor and the environment: a caller keeps the forward's
// value and compares it before the backward (hip_binding does; a mismatch would otherwise be silently wrong gradients).
extern "C" int fhvae_lstm_layout_id(const fhvae_lstm_desc* d) {
  if (!d || check_desc(d) != FHVAE_OK) return -1;
  if (d->dtype == FHVAE_BF16 && cluster_eligible(d)) return 16 + cluster_form(d) * 2 + (cluster_fwd_wr_ok(d) ? 1 : 0);
  return big_cells(d->B, d->H, d->dtype) && big_shape_ok(d) ? 1 : 0;
}

extern "C" int64_t fhvae_lstm_pre_elems(const fhvae_lstm_desc* d) {
  if (!d || d->L < 1 || d->L > FHVAE_MAX_LAYERS || d->B <= 0 || d->T <= 0 || d->H <= 0) return 0;
  const int64_t full = (d->I > 0 ? d->T : 1) * d->B * 4 * d->H;
  if (d->dtype == FHVAE_BF16 && cluster_eligible(d)) return (d->I == 0 || cluster_can_fold(d)) ? d->B * 4 * d->H : full;
  if (big_cells(d->B, d->H, d->dtype) && big_shape_ok(d)) return 1;  // the cells multiply layer 0's input themselves
  return full;
}

extern "C" int64_t fhvae_lstm_ws_below_elems(const fhvae_lstm_desc* d) {
  if (!d || d->dtype != FHVAE_BF16 || d->L < 2) return 0;
  return cluster_needs_ws_below(d) ? d->T * d->B * d->H : 0;
}

__global__ void cast_hn_kernel(const float* __restrict__ s, u16* __restrict__ d, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = f2bf(s[i]);
}

extern "C" int fhvae_lstm_seq_fwd(const fhvae_lstm_desc* d, void* stream) {
  int e = check_desc(d);
  if (e) return e;
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == FHVAE_F32) {
    if (d->lp) {
      const Lp32Layout Y = lp32_layout(d);
      float* base = (float*)d->lp;
      TransBatch tb = {};
      for (int l = 0; l < d->L; ++l) {
        if (l > 0) tb.it[tb.n++] = TransItem{d->w_ih[l], base + Y.w_ih_t[l], 4 * d->H, d->H};
        tb.it[tb.n++] = TransItem{d->w_hh[l], base + Y.w_hh_t[l], 4 * d->H, d->H};
      }
      hipLaunchKernelGGL(transpose_f32_kernel, dim3(256, (unsigned)tb.n), dim3(256), 0, (hipStream_t)stream, tb);
      e = fh_launch_status();
      if (e) return e;
    }
    return lstm_fwd_impl<float>(d, ops_f32(d), st);
  }
  e = cast_operands(d, st);
  if (e) return e;
  e = lstm_fwd_impl<u16>(d, ops_bf16(d), st);
  if (e || !d->hn_lp) return e;
  if (!d->hn) return FHVAE_ERR_NULL;
  if (cluster_eligible(d)) return FHVAE_OK;  // the persistent forward kernels stored the bf16 copy beside hn
  const int64_t n = d->B * d->L * d->H;
  hipLaunchKernelGGL(cast_hn_kernel, dim3((unsigned)fh_cdiv(n, 256)), dim3(256), 0, st, d->hn, (u16*)d->hn_lp, n);
  return fh_launch_status();
}

// the backward jobs of wavefront step w (layer l at time T-1-(w-(L-1-l)))
template <typename T>
static BwdJobs<T> bwd_jobs(const fhvae_lstm_bwd_desc* bd, const Ops<T>& op, int64_t w, int& nj) {
  const fhvae_lstm_desc* d = &bd->f;
  const int64_t B = d->B, T_ = d->T, Ic = d->Ic, H = d->H;
  const int L = d->L;
  constexpr bool kF32 = sizeof(T) == 4;
  T* dg = (T*)bd->dgates;
  BwdJobs<T> jobs = {};
  jobs.B = (int)B;
  jobs.H = (int)H;
  jobs.glds = 1;
  nj = 0;
  // f32: the weights are read untransposed as KM operands; bf16: the transposed copies [H,4H] are KC operands.
  for (int l = L - 1; l >= 0; --l) {
    const int64_t u = w - (L - 1 - l);
    if (u < 0 || u >= T_) continue;
    const int64_t t = T_ - 1 - u;
    BwdJob<T>& J = jobs.job[nj++];
    const int64_t lt = (int64_t)l * T_ + t;
    if (t < T_ - 1) {
      const T* a = dg + (lt + 1) * B * 4 * H;
      J.seg[0] = (kF32 && !op.w_hh_t[l]) ? Seg{a, 4 * H, 1, op.w_hh[l], H, 0, (int)(4 * H), 0}
                                         : Seg{a, 4 * H, 1, op.w_hh_t[l], 4 * H, 1, (int)(4 * H), 0};
    }
    if (l < L - 1) {
      const T* a = dg + ((int64_t)(l + 1) * T_ + t) * B * 4 * H;
      J.seg[1] = (kF32 && !op.w_ih_t[l + 1]) ? Seg{a, 4 * H, 1, op.w_ih[l + 1], H, 0, (int)(4 * H), 0}
                                             : Seg{a, 4 * H, 1, op.w_ih_t[l + 1], 4 * H, 1, (int)(4 * H), 0};
    }
    if (l == L - 1 && bd->d_hs_top) {
      J.ext = bd->d_hs_top + t * B * H;
      J.ext_ld = H;
    }
    if (t == T_ - 1 && bd->d_hn) {
      J.ext2 = bd->d_hn + (int64_t)l * H;
      J.ext2_ld = (int64_t)L * H;
    }
    J.gates = (const T*)d->gates + lt * B * 4 * H;
    J.c_prev = t > 0 ? d->cs + (lt - 1) * B * H : nullptr;
    J.c_cur = d->cs + lt * B * H;
    J.dc = bd->dc + (int64_t)l * B * H;
    J.first = t == T_ - 1;
    J.dg_out = dg + lt * B * 4 * H;
    J.dgsum = (l == 0 && Ic > 0) ? bd->dgsum : nullptr;
  }
  return jobs;
}

template <typename T>
static bool cell_bwd_plan_ok(const fhvae_lstm_bwd_desc* bd, const Ops<T>& op) {
  for (int l = 0; l < bd->f.L; ++l)
    if (!op.w_hh_t[l] || (l > 0 && !op.w_ih_t[l])) return false;  // (f32 without the workspace: no transposed weights)
  for (int64_t w = 0; w < bd->f.T + bd->f.L - 1; ++w) {
    int nj = 0;
    const BwdJobs<T> jobs = bwd_jobs<T>(bd, op, w, nj);
    if (!cell_bwd_big_ok(jobs, nj)) return false;
  }
  return true;
}

template <typename T>
static int lstm_bwd_impl(const fhvae_lstm_bwd_desc* bd, const Ops<T>& op, hipStream_t st) {
  const fhvae_lstm_desc* d = &bd->f;
  const int64_t B = d->B, T_ = d->T, Ic = d->Ic, H = d->H;
  const int L = d->L;
  constexpr bool kF32 = sizeof(T) == 4;
  if constexpr (!kF32) {
    if (cluster_eligible(d)) {  // the forward on this workspace took the persistent form too (same predicate)
      ClusterWeights cw = {};
      for (int l = 0; l < L; ++l) cw.w_ih[l] = op.w_ih[l], cw.w_hh[l] = op.w_hh[l], cw.w_ih_t[l] = op.w_ih_t[l], cw.w_hh_t[l] = op.w_hh_t[l];
      cw.xch = (u16*)d->lp + lp_layout(d).xch;
      return cluster_bwd(bd, cw, st);
    }
  }
  if (big_cells(B, H, d->dtype) && cell_bwd_plan_ok(bd, op)) {
    // large-tile cells (lstm_cell.hip); they leave the time sum of layer 0's gate gradients to one pass over the saved
    // dgates.  (Tried: the batch rows as 2 / 4 independent launch chains on side streams, so that one chain's
    // HBM-bound epilogue would run beside the other's contraction: 348 k / 312 k segments/s against 361 k for one chain.)
    for (int64_t w = 0; w < T_ + L - 1; ++w) {
      int nj = 0;
      const BwdJobs<T> jobs = bwd_jobs<T>(bd, op, w, nj);
      double fl = 0;
      for (int j = 0; j < nj; ++j) fl += 2.0 * B * H * (jobs.job[j].seg[0].K + jobs.job[j].seg[1].K);
      const int ts = trace_begin(st, kTraceBwdCell, fl);
      const int e = launch_cell_bwd_big(jobs, nj, st);
      trace_end(st, ts);
      if (e) return e;
    }
    if (Ic > 0) return launch_cell_dgsum((const T*)bd->dgates, bd->dgsum, (int)T_, B * 4 * H, st);
    return FHVAE_OK;
  }
  for (int64_t w = 0; w < T_ + L - 1; ++w) {
    int nj = 0;
    const BwdJobs<T> jobs = bwd_jobs<T>(bd, op, w, nj);
    double fl = 0;
    for (int j = 0; j < nj; ++j) fl += 2.0 * B * H * (jobs.job[j].seg[0].K + jobs.job[j].seg[1].K);
    const int ts = trace_begin(st, kTraceBwdCell, fl);
    // tile by how many workgroups the shape offers (>= 2 per CU wanted): 128x64 from B = 16384 at H = 256, 64x64 from
    // B = 2048 at H = 512 (32x32 tiles there: 2048 workgroups re-reading 1 GB of operand panels per launch from L2)
    // (measured at B = 2048, H = 512, bf16: 64x64 tiles 2.4 ms per net backward, 32x32 2.2-2.3 ms: no gain from larger tiles)
    const int bt = B >= 16384 ? 128 : 32;
    if (bt == 128) {
      dim3 grid((unsigned)fh_cdiv(B, 128), (unsigned)fh_cdiv(H, 64), (unsigned)nj);
      hipLaunchKernelGGL((lstm_bwd_step_kernel<T, 128, 64, 4, 1, 16>), grid, dim3(kThreads), 0, st, jobs);
    } else if (bt == 64) {
      dim3 grid((unsigned)fh_cdiv(B, 64), (unsigned)fh_cdiv(H, 64), (unsigned)nj);
      hipLaunchKernelGGL((lstm_bwd_step_kernel<T, 64, 64, 2, 2, 16>), grid, dim3(kThreads), 0, st, jobs);
    } else if (B >= 1024) {
      dim3 grid((unsigned)fh_cdiv(B, 32), (unsigned)fh_cdiv(H, 32), (unsigned)nj);
      hipLaunchKernelGGL((lstm_bwd_step_kernel<T, 32, 32, 2, 2, 16>), grid, dim3(kThreads), 0, st, jobs);
    } else {
      dim3 grid((unsigned)fh_cdiv(B, 32), (unsigned)fh_cdiv(H, 32), (unsigned)nj);
      hipLaunchKernelGGL((lstm_bwd_step_kernel<T, 32, 32, 2, 2, kCH>), grid, dim3(kThreads), 0, st, jobs);
    }
    trace_end(st, ts);
    int e = fh_launch_status();
    if (e) return e;
  }
  return FHVAE_OK;
}

// weight / bias / input gradients after the recurrence.  The long contractions (over T*B rows) run in
// the operand dtype (KM/KM: bf16 uses transposed LDS reads); the time-constant-input part (B rows) and
// d_xc stay f32 (f32 dgsum).
constexpr int64_t kWgradMinK = 1024;  // shorter contractions stay on the generic engine (grouped 64x64 tiles)

// `wq` (bf16 only): long contractions that meet wgrad.hip's preconditions are appended to it instead of being launched; the
// caller launches everything it has collected (possibly from several nets) as one grouped launch (launch_wgrad).
template <typename T>
static int lstm_param_grads(const fhvae_lstm_bwd_desc* bd, const Ops<T>& op, hipStream_t st, std::vector<WgProblem>* wq = nullptr,
                            std::vector<GemmParams>* fq = nullptr, std::vector<WgProblem32>* wq32 = nullptr) {
  const fhvae_lstm_desc* d = &bd->f;
  const int64_t B = d->B, T_ = d->T, I = d->I, Ic = d->Ic, H = d->H;
  const int L = d->L;
  const int64_t K0 = I + Ic, G = 4 * H;
  const T* dg = (const T*)bd->dgates;
  const T* hs = (const T*)d->hs;
  // c[G, Ncols] += a[Kc, G]^T . b[Kc, Ncols]; the operand-dtype contractions of the whole net go out as ONE grouped launch
  // (at small batches each is a latency-bound launch of a few workgroups)
  GemmParams grp[2 * FHVAE_MAX_LAYERS];
  int ng = 0;
  // -> true: taken by the dedicated long-K kernel (queued in wq)
  auto wgrad_long = [&](const void* a, int64_t lda, const void* b, int64_t ldb, int64_t Kc, float* c, int64_t ldc, int64_t Ncols) {
    if (sizeof(T) == 4) {  // f32 mode: the exact-f32 form of the long-K kernel (wgrad_f32.hip)
      if (!wq32 || Kc < kWgradMinK) return false;
      WgProblem32 w = {};
      w.A = (const float*)a, w.B = (const float*)b, w.C = c;
      w.lda = lda, w.ldb = ldb, w.ldc = ldc;
      w.M = (int)G, w.N = (int)Ncols, w.K = (int)Kc;
      if (!wgrad32_eligible(w)) return false;
      wq32->push_back(w);
      return true;
    }
    if (!wq || sizeof(T) != 2 || Kc < kWgradMinK) return false;
    WgProblem w = {};
    w.A = (const u16*)a, w.B = (const u16*)b, w.C = c;
    w.lda = lda, w.ldb = ldb, w.ldc = ldc;
    w.M = (int)G, w.N = (int)Ncols, w.K = (int)Kc;
    if (!wgrad_eligible(w)) return false;
    wq->push_back(w);
    return true;
  };
  auto wgrad = [&](const void* a, int64_t lda, const void* b, int64_t ldb, int64_t Kc, float* c, int64_t ldc, int64_t Ncols) {
    GemmParams p = {};
    p.seg[0] = Seg{a, lda, 0, b, ldb, 0, (int)Kc, 0};
    p.M = (int)G;
    p.N = (int)Ncols;
    p.C = c;
    p.ldc = ldc;
    p.splitk = 0;  // auto (atomics when split)
    p.mode = 1;
    return p;
  };
  auto flush = [&]() -> int {
    int e = FHVAE_OK;
    for (int i = 0; i < ng && !e; i += kMaxGroup) e = launch_gemm_group(grp + i, ng - i < kMaxGroup ? ng - i : kMaxGroup, d->dtype, st);
    ng = 0;
    return e;
  };
  for (int l = 0; l < L; ++l) {
    const T* dgl = dg + (int64_t)l * T_ * B * G;
    const T* hl = hs + (int64_t)l * T_ * B * H;
    int e;
    if (bd->dw_hh[l] && T_ > 1 && !wgrad_long(dgl + B * G, G, hl, H, (T_ - 1) * B, bd->dw_hh[l], H, H))
      grp[ng++] = wgrad(dgl + B * G, G, hl, H, (T_ - 1) * B, bd->dw_hh[l], H, H);
    if (bd->dw_ih[l]) {
      if (l > 0) {
        if (!wgrad_long(dgl, G, hs + (int64_t)(l - 1) * T_ * B * H, H, T_ * B, bd->dw_ih[l], H, H))
          grp[ng++] = wgrad(dgl, G, hs + (int64_t)(l - 1) * T_ * B * H, H, T_ * B, bd->dw_ih[l], H, H);
      } else {
        if (I > 0 && !wgrad_long(dgl, G, op.x, I, T_ * B, bd->dw_ih[0], K0, I))
          grp[ng++] = wgrad(dgl, G, op.x, I, T_ * B, bd->dw_ih[0], K0, I);
        if (Ic > 0) {  // the time-constant input's part: f32 running sum over t of dg (B rows)
          GemmParams p = wgrad(bd->dgsum, G, d->xc, Ic, B, bd->dw_ih[0] + I, K0, Ic);
          if (fq) {  // several nets at once: these small f32 contractions go out as one grouped launch too
            fq->push_back(p);
          } else {
            e = launch_gemm(p, FHVAE_F32, st);
            if (e) return e;
          }
        }
      }
    }
    if (!(sizeof(T) == 2 && cluster_eligible(d))) {  // (the persistent backward kernels sum the bias gradients themselves)
      // layer 0 with a time-constant input: the sum over t already exists (dgsum, f32 [B,4H]): B rows instead of T*B
      if (l == 0 && Ic > 0 && bd->dgsum)
        e = launch_colsum(bd->dgsum, FHVAE_F32, G, bd->db_ih[l], bd->db_hh[l], B, G, st);
      else
        e = launch_colsum(dgl, d->dtype, G, bd->db_ih[l], bd->db_hh[l], T_ * B, G, st);
      if (e) return e;
    }
  }
  return flush();
}

// d_xc[B,Ic] = dgsum[B,4H] . W_ih0[:, I:]   (f32 master weight as KM operand: B(n, k) = W[k*K0 + I + n])
static int lstm_dxc(const fhvae_lstm_bwd_desc* bd, hipStream_t st, bool zeroed = false) {
  const fhvae_lstm_desc* d = &bd->f;
  if (!bd->d_xc || d->Ic <= 0) return FHVAE_OK;
  const int64_t G = 4 * d->H, K0 = d->I + d->Ic;
  GemmParams p = {};
  p.seg[0] = Seg{bd->dgsum, G, 1, d->w_ih[0] + d->I, K0, 0, (int)G, 0};
  p.M = (int)d->B;
  p.N = (int)d->Ic;
  p.C = bd->d_xc;
  p.ldc = d->Ic;
  // few output tiles (B/64), a 4H-long contraction: split K over the workgroups (zero + f32 atomics) instead of walking
  // it serially in B/64 of them (33 -> 10 us at B = 256 .. 2048)
  const int64_t tiles = fh_cdiv(d->B, 64) * fh_cdiv(d->Ic, 64);
  if (tiles <= 64 && G >= 512) {
    const int64_t n = d->B * d->Ic;
    if (!zeroed) hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)fh_cdiv(n, 1024)), dim3(256), 0, st, bd->d_xc, n);
    p.mode = 2;
    // about one workgroup per CU, slices of at least 128 (4 at B = 2048 left half the chip idle: 13.4 us per launch)
    int64_t sk = 256 / tiles;
    while (sk > 1 && G / sk < 128) sk >>= 1;
    p.splitk = (int)(sk < 4 ? 4 : sk > 16 ? 16 : sk);
  } else {
    p.splitk = 1;
  }
  return launch_gemm(p, FHVAE_F32, st);
}

extern "C" int fhvae_lstm_seq_bwd(const fhvae_lstm_bwd_desc* bd, void* stream) {
  FH_CHECK_PTR(bd);
  const fhvae_lstm_desc* d = &bd->f;
  int e = check_desc(d);
  if (e) return e;
  FH_CHECK_PTR(bd->dgates);
  FH_CHECK_PTR(bd->dc);
  if (d->Ic > 0) FH_CHECK_PTR(bd->dgsum);
  if (bd->phase < 0 || bd->phase > 2) return FHVAE_ERR_SHAPE;
  const bool rec = bd->phase != 2, par = bd->phase != 1;
  if (rec && !bd->d_hs_top && !bd->d_hn) return FHVAE_ERR_NULL;
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == FHVAE_F32) {
    Ops<float> op = ops_f32(d);
    if (rec) {
      e = lstm_bwd_impl<float>(bd, op, st);
      if (e) return e;
      e = lstm_dxc(bd, st);
      if (e) return e;
    }
    if (!par) return FHVAE_OK;
    std::vector<WgProblem32> wq32;
    e = lstm_param_grads<float>(bd, op, st, nullptr, nullptr, getenv("FHVAE_NO_WGRAD") ? nullptr : &wq32);
    if (e) return e;
    return launch_wgrad32(wq32.data(), (int)wq32.size(), st);
  }
  Ops<u16> op = ops_bf16(d);  // filled by the forward
  if (rec) {
    e = lstm_bwd_impl<u16>(bd, op, st);
    if (e) return e;
    e = lstm_dxc(bd, st, cluster_eligible(d) && cluster_bwd_zeroes_dxc(d));
    if (e) return e;
  }
  if (!par) return FHVAE_OK;
  std::vector<WgProblem> wq;
  e = lstm_param_grads<u16>(bd, op, st, getenv("FHVAE_NO_WGRAD") ? nullptr : &wq);
  if (e) return e;
  return launch_wgrad(wq.data(), (int)wq.size(), st);
}

// Phase 2 (parameter gradients) of n backward passes whose recurrences (phase 1) have run: the long weight-gradient
// contractions of ALL of them go out as one grouped launch of wgrad.hip's kernel (one workgroup per CU, the K slices being
// what is left after the tiles: 3 nets = 36 tiles x 7 slices instead of 12 launches x 512 workgroups of split-K atomics).
static bool wg_from_desc(const fhvae_wgrad_desc* x, WgProblem& p) {
  if (!x || x->M <= 0 || x->N <= 0 || x->K <= 0 || x->M > INT32_MAX || x->N > INT32_MAX || x->K > INT32_MAX || x->a_col0 < 0 ||
      x->a_col0 > INT32_MAX)
    return false;
  p = WgProblem{};
  p.A = (const u16*)x->a, p.B = (const u16*)x->b, p.C = x->c;
  p.lda = x->lda, p.ldb = x->ldb, p.ldc = x->ldc;
  p.M = (int)x->M, p.N = (int)x->N, p.K = (int)x->K;
  p.a_col0 = (int)x->a_col0;
  return wgrad_eligible(p) && x->ldc >= x->N;
}

extern "C" int fhvae_wgrad_desc_ok(const fhvae_wgrad_desc* x) {
  WgProblem p;
  return wg_from_desc(x, p) ? 1 : 0;
}

extern "C" int fhvae_lstm_param_grads_multi(const fhvae_lstm_bwd_desc* const* bds, int n, const fhvae_wgrad_desc* extra, int n_extra,
                                            void* stream) {
  if (n < 0 || n_extra < 0) return FHVAE_ERR_SHAPE;
  if (n > 0) FH_CHECK_PTR(bds);
  if (n_extra > 0) FH_CHECK_PTR(extra);
  hipStream_t st = (hipStream_t)stream;
  std::vector<WgProblem> wq;
  for (int i = 0; i < n_extra; ++i) {
    WgProblem p;
    if (!wg_from_desc(extra + i, p)) return FHVAE_ERR_ALIGN;
    wq.push_back(p);
  }
  std::vector<GemmParams> fq;  // the f32 (B-row) contractions of the time-constant inputs
  std::vector<WgProblem32> wq32;  // f32 mode: the long contractions of every queued net, one grouped launch (wgrad_f32.hip)
  const bool use_wq = !getenv("FHVAE_NO_WGRAD");
  for (int i = 0; i < n; ++i) {
    const fhvae_lstm_bwd_desc* bd = bds[i];
    FH_CHECK_PTR(bd);
    const fhvae_lstm_desc* d = &bd->f;
    int e = check_desc(d);
    if (e) return e;
    FH_CHECK_PTR(bd->dgates);
    if (d->Ic > 0) FH_CHECK_PTR(bd->dgsum);
    if (d->dtype == FHVAE_F32) {
      e = lstm_param_grads<float>(bd, ops_f32(d), st, nullptr, nullptr, use_wq ? &wq32 : nullptr);
    } else {
      e = lstm_param_grads<u16>(bd, ops_bf16(d), st, use_wq ? &wq : nullptr, &fq);
    }
    if (e) return e;
  }
  // one grouped launch per batch of problems with DISTINCT outputs: a net queued twice (two backward passes before one flush)
  // accumulates into the same matrix twice, and two workgroups of one launch must not read-modify-write the same tile
  for (size_t i = 0; i < fq.size();) {
    size_t j = i;
    while (j < fq.size() && j - i < (size_t)kMaxGroup) {
      bool dup = false;
      for (size_t k = i; k < j; ++k) dup = dup || fq[k].C == fq[j].C;
      if (dup) break;
      ++j;
    }
    const int e = launch_gemm_group(fq.data() + i, (int)(j - i), FHVAE_F32, st);
    if (e) return e;
    i = j;
  }
  {
    const int e = launch_wgrad32(wq32.data(), (int)wq32.size(), st);
    if (e) return e;
  }
  return launch_wgrad(wq.data(), (int)wq.size(), st);
}
