// lstm_cluster.hip -- K1 in its persistent form: the bf16 recurrence of a net (all T steps) in ONE launch of 256
// workgroups, one per CU, instead of one launch per wavefront step.
//
// Why: one launch per wavefront step costs a dependent kernel boundary plus a cold start of every tile (27 us per
// step at B = 2048, 11 us at B = 256).  The recurrence only couples the HIDDEN UNITS of one batch row, so the batch is
// cut into CLUSTERS of workgroups that never talk to another cluster; a cluster is formed from workgroups of ONE XCD
// (HW_REG_XCC_ID), so everything its members exchange stays in that XCD's L2:
//   * a member owns a slice of the hidden units: its gate columns of W_hh (and W_ih of the layer above) live in LDS for the
//     whole launch, the cell state c (forward) / dc and the bias-gradient sums (backward) live in registers;
//   * per step the only shared data is h (forward) / dg (backward) of the previous step, bf16, written once at a fresh
//     address (they are the saved-for-backward / weight-gradient operands anyway);
//   * hand-off per step: plain stores -> s_waitcnt vmcnt(0) (the write-through L1 has delivered them to the XCD's L2)
//     -> workgroup barrier -> one agent-scope flag store; every wave polls the flags of its cluster with L1-bypassing
//     loads, then reads the operand with sc1 loads.  No agent-scope fence: an L2 write-back / invalidate costs 17+ us per
//     step (tools/exp/xcd_barrier.hip: 1.1-1.4 us per step for this form, 17-27 us with fences) and is not needed inside
//     one XCD.  The same-XCD premise is not assumed from blockIdx: a workgroup reads its XCD from the hardware
//     register and takes a slot by an atomic ticket on that XCD's counter (32 tickets per XCD and launch: ticket / 32 is also
//     the launch's number on the sync block); a workgroup that found its XCD full would hold a ticket of the next launch, wait
//     for flags nobody raises and end the launch through the bounded spin (status word): it never computes from a stale line;
//   * every spin is bounded and watches the abort word: a lost workgroup ends the launch, it cannot hang the GPU.
// Kernels (H in {128, 256}, L <= 2; chosen by rows per cluster, cluster_form):
//   lstm_fwd_cluster_kernel   forward, both layers as one wavefront, 16 units per member, 64/128 rows per cluster; every
//                             wave stages its own rows with LDS-DMA into a private ring (no barrier in the contraction);
//                             the layer-0 input projection folded in (W_ih[0] fragments in registers)
//   lstm_bwd_layer_kernel     backward of ONE layer per launch, 32 units per member (a quarter of the wavefront kernel's
//                             exchange bytes per step), a helper wave fetching the epilogue operands; the from-above term
//                             is a GEMM between the two launches (cluster_bwd_layers)
//   lstm_fwd/bwd_ksplit_kernel  <= 32 rows per cluster (B <= 512): the waves split the CONTRACTION, operands go
//                             global -> registers from a blocked exchange buffer, partial tiles are summed through LDS
//
// Semantics are those of lstm.hip's step kernels (same accumulation order per k is NOT promised: parity is the bf16
// tolerance of the model tests, and tests/test_lstm_cluster_gpu.py compares the two schedules directly).  f32 (parity
// mode) stays on the per-step kernels.
#include "lstm_cluster_dev.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "gemm_core.h"
#include "gemm_launch.h"
#include "proj.h"
#include "trace.h"


namespace fh {

// Wave-private operand staging.  A wave multiplies only its own TM*16 batch rows, so it stages them itself (LDS-DMA,
// L1-bypassing) into its own ring of kRing 4-KB panels and orders itself with s_waitcnt vmcnt(n) alone: no workgroup
// barrier inside the contraction, kRing-1 panels (12 KB per wave, 48 KB per CU) in flight under the MFMAs.
constexpr int kRing = 4;
constexpr int kPanel = 4096;

// one panel = ROWS rows x CH chunks (ROWS * CH * 16 = 4096): rows row_base.. of the source, contraction offset k0
template <int ROWS, int CH, class RowMap>
__device__ __forceinline__ void glds_wave_panel(char* lds, const u16* base, int64_t ld, int k0, const RowMap& rm, int row_base,
                                                int lane) {
  constexpr int RPI = 64 / CH;  // rows per wave-instruction
  constexpr int SW = CH >= 16 ? 15 : CH - 1;
  static_assert(ROWS * CH * 16 == kPanel && ROWS % RPI == 0, "panel shape");
#pragma unroll
  for (int j = 0; j < ROWS / RPI; ++j) {
    const int row = j * RPI + lane / CH;
    const int c = (lane % CH) ^ (row & SW);
    const u16* src = base + (int64_t)(uint32_t)rm(row_base + row) * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)(lds + j * 1024), 16, 0, kSc1);
  }
}
// The EXCHANGE buffer of the contraction-split kernels.  What the members hand each other (h forward, dg backward) is ALSO
// written in a blocked layout [parity][layer][k-step (32 contraction elements)][batch row][32] bf16, next to the row-major
// tensors the later consumers read: an MFMA operand fragment (16 rows x 32 k) is then 1 KB of CONTIGUOUS memory instead of
// 16 pieces of 64 B at the row stride -- measured (tools/exp/l2_read.hip, 16 readers per block, sc1 loads, one 64-KB burst
// per CU) 57-74 GB/s per CU against 31 GB/s.  (The rows-form kernels stage 128-byte row segments and stream 512 KB per CU
// and step: they sit at the ~10 TB/s aggregate L2 rate in either layout, 433 us against 398 with the extra stores.)
// Two parities: step s writes parity s&1 and reads (s-1)&1; a member enters step s only after every member has finished its
// reads of step s-2's parity.
__device__ __forceinline__ int64_t xch_off(int par, int l, int L, int KS, int ks, int64_t B, int64_t row) {
  return ((((int64_t)par * L + l) * KS + ks) * B + row) * 32;
}


// wait until at most `younger` panels (4 DMA instructions each) issued after the needed one are still in flight
__device__ __forceinline__ void wait_panels(int younger) {
  if (younger >= 2)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (younger == 1)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int H, int L, int RB>
struct ClFwdCfg {
  static constexpr int HC = H / 8;                        // 16-byte chunks per h row
  static constexpr int TM = RB >= 64 ? RB / 64 : 1;       // 16-row tiles per wave
  static constexpr int WR = TM * 16;                      // rows of one wave
  static constexpr int PCH = kPanel / (WR * 16);          // chunks per row in a 4-KB panel (TM = 2: 8, TM = 1: 16)
  static constexpr int NPP = HC / PCH;                    // panels per source
  static constexpr int W_BYTES = 64 * HC * 16;
  static constexpr int NW = 2 * L - 1;
  static constexpr int SMEM = NW * W_BYTES + 4 * kRing * kPanel;
  static_assert(PCH >= 8 && HC % PCH == 0, "panel shape");
};

template <int H, int L, int RB>
__global__ __launch_bounds__(kThreads) void lstm_fwd_cluster_kernel(ClFwd p) {
  using CF = ClFwdCfg<H, L, RB>;
  constexpr int HC = CF::HC, TM = CF::TM, PCH = CF::PCH, NPP = CF::NPP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Wl = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* ring = smem + CF::NW * CF::W_BYTES + wave * (kRing * kPanel);  // this wave's staging ring
  const int r = lane & 15, q = lane >> 4;

  const int joined = cluster_join(p.sync, (int*)(smem + CF::NW * CF::W_BYTES));  // (the rings are idle until step 1)
  if (joined < 0) return;
  const int info = joined & 255;                                 // XCD * 32 + slot
  const unsigned ep0 = (unsigned)(joined >> 8) * kSeqEpochs;     // this launch's number on the sync block
  const int NU = p.NU;
  const int cluster = (info >> 5) * (32 / NU) + (info & 31) / NU, me = (info & 31) % NU;
  const int r0 = p.row0 + cluster * p.Mc;
  const int rend = min(p.row0 + p.nrows, r0 + p.Mc);
  if (r0 >= rend) return;  // the whole cluster leaves: nobody waits for it
  unsigned* flags = p.sync + kSyncFlags + cluster * 32;
  const int u0 = me * 16;
  const int B = p.B, T = p.T;

  // ---- weights of this workgroup's 64 gate columns -> LDS, once
  {
    ClGateMap gm{H, u0};
#pragma unroll
    for (int l = 0; l < L; ++l) {
      glds_tile<u16, 64, HC>(Wl + (2 * l) * CF::W_BYTES, p.w_hh[l], H, 0, 0, gm, 0, tid);
      if (l > 0) glds_tile<u16, 64, HC>(Wl + (2 * l - 1) * CF::W_BYTES, p.w_ih[l], H, 0, 0, gm, 0, tid);
    }
  }
  // MFMA roles: the WEIGHT fragment is the first operand, the h fragment the second, so the 16x16 result has the
  // hidden unit on (lane>>4)*4 + reg and the batch row on lane&15: a lane owns i,f,g,o of FOUR CONSECUTIVE units of one
  // row -> every f32 output leaves as one 16-byte store, every bf16 output as one 8-byte store
  const int uq = u0 + q * 4;  // first of this lane's 4 units
  f32x4 bias[L][4];
#pragma unroll
  for (int l = 0; l < L; ++l)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bias[l][g] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (l > 0) bias[l][g] = *(const f32x4*)(p.b_ih[l] + g * H + uq) + *(const f32x4*)(p.b_hh[l] + g * H + uq);
    }

  if (!p.pre) {  // no time-constant input: layer 0's additive term is its two biases
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[0][g] = *(const f32x4*)(p.b_ih[0] + g * H + uq) + *(const f32x4*)(p.b_hh[0] + g * H + uq);
  }
  const int wrow0 = wave * (TM * 16);       // first cluster row of this wave
  const bool wact = wrow0 < RB;             // waves beyond the tile only help staging
  f32x4 creg[L][TM];
#pragma unroll
  for (int l = 0; l < L; ++l)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) creg[l][tm] = f32x4{0.f, 0.f, 0.f, 0.f};
  ClRowMap arm{r0, rend - 1};
  auto pack4 = [](const f32x4& v) -> uint2 {
    return uint2{(uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16), (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16)};
  };

  unsigned long long* tl = (p.tlog && cluster == 0 && me == 0) ? p.tlog : nullptr;
  // what only the backward / the caller reads (c, the activated gates, the f32 copy of the top h): stored after the publish.
  // (Deferring these stores into the next step's contraction was measured slower: they queue in front of its operand
  // loads, 246 -> 269 us per net at B = 2048.)
  uint2 gpk[L][TM][4];  // activated gates, packed bf16
  f32x4 hreg[L][TM];
  auto tail_stores = [&](int sp) {
    if (!wact) return;
#pragma unroll
    for (int ll = 0; ll < L; ++ll) {
      const int t = sp - ll;
      if (t < 0 || t >= T) continue;
      const int64_t lt = (int64_t)ll * T + t;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int row = r0 + wrow0 + tm * 16 + r;
        if (row >= rend) continue;
        *(f32x4*)(p.cs + (lt * B + row) * H + uq) = creg[ll][tm];
        cl_store_gates(p.gates + (lt * B + row) * (4 * H), uq, gpk[ll][tm]);
        if (ll == L - 1 && p.hs_top_f32) *(f32x4*)(p.hs_top_f32 + ((int64_t)t * B + row) * H + uq) = hreg[ll][tm];
        if (p.hn && t == T - 1) {
          *(f32x4*)(p.hn + (int64_t)row * (L * H) + ll * H + uq) = hreg[ll][tm];
          if (p.hn_lp) *(uint2*)(p.hn_lp + (int64_t)row * (L * H) + ll * H + uq) = pack4(hreg[ll][tm]);  // (fhvae_lstm_desc.hn_lp)
        }
      }
    }
  };
  f32x4 pnext[TM][4];
  const bool pvar = p.pre && p.pre_tstride != 0;  // a different additive term every step (unfolded input projection)
  auto load_pre = [&](int t) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int row = r0 + wrow0 + tm * 16 + r;
      const float* pp = p.pre + (int64_t)t * p.pre_tstride + (int64_t)(row < rend ? row : rend - 1) * (4 * H) + uq;
#pragma unroll
      for (int g = 0; g < 4; ++g) pnext[tm][g] = (wact && p.pre) ? *(const f32x4*)(pp + g * H) : bias[0][g];
    }
  };
  load_pre(0);
  if (p.xcv) {  // the time-constant input's projection, once: same fragment roles as the folded x projection below
    const int nkc = (p.Ic + 31) / 32, nchc = p.Ic / 8;
    f32x4 accx[TM][4];
    zero_acc(accx);
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // Ic <= 128
      if (j >= nkc) break;
      const int c = j * 4 + q;
      const bool ok = c < nchc;
      uint4 xf[TM];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int row = r0 + wrow0 + tm * 16 + r;
        xf[tm] = uint4{0u, 0u, 0u, 0u};
        if (ok && wact) xf[tm] = *(const uint4*)(p.xcv + (int64_t)(row < rend ? row : rend - 1) * p.Ic + c * 8);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint4 wf = uint4{0u, 0u, 0u, 0u};
        if (ok) wf = *(const uint4*)(p.w_ih0 + (int64_t)(g * H + u0 + r) * p.K0 + p.I + c * 8);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          accx[tm][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xf[tm]), accx[tm][g], 0, 0, 0);
      }
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int g = 0; g < 4; ++g) pnext[tm][g] += accx[tm][g];
  }
  // folded input projection: this member's W_ih[0] fragments (gate g, k-step j) stay in registers for the whole launch;
  // the x fragments of step s+1 are fetched under the epilogue of step s (x comes from HBM, like `pre`)
  constexpr int KSX = 4;  // up to 128 input features
  const bool fold = p.x != nullptr;
  const int nkx = fold ? (p.I + 31) / 32 : 0, nchx = p.I / 8;
  uint4 wx[4][KSX], xn[TM][KSX];
#pragma unroll
  for (int j = 0; j < KSX; ++j) {
    const int c = j * 4 + q;
    const bool ok = fold && c < nchx;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint4 v = uint4{0u, 0u, 0u, 0u};
      if (ok) v = *(const uint4*)(p.w_ih0 + (int64_t)(g * H + u0 + r) * p.K0 + c * 8);
      wx[g][j] = v;
    }
  }
  auto load_x = [&](int t) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int row = r0 + wrow0 + tm * 16 + r;
      const u16* xp = p.x + ((int64_t)t * B + (row < rend ? row : rend - 1)) * p.I;
#pragma unroll
      for (int j = 0; j < KSX; ++j) {
        const int c = j * 4 + q;
        uint4 v = uint4{0u, 0u, 0u, 0u};
        if (fold && wact && c < nchx) v = *(const uint4*)(xp + c * 8);
        xn[tm][j] = v;
      }
    }
  };
  load_x(0);
  const int nsteps = T + L - 1;
  for (int s = 0; s < nsteps; ++s) {
    CL_TLOG(s * 8 + 0);
    // (1) layer 0's additive term for t = s was fetched during step s-1 (it comes from HBM: loads return in order, so
    //     fetching it here would put its latency in front of the flag poll)
    f32x4 padd[TM][4];
    uint4 xc[TM][KSX];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int g = 0; g < 4; ++g) padd[tm][g] = pnext[tm][g];
#pragma unroll
      for (int j = 0; j < KSX; ++j) xc[tm][j] = xn[tm][j];
    }
    // (2) h of step s-1 from every member
    if (s > 0 && !cluster_wait(p.sync, flags, NU, ep0 + (unsigned)s)) return;

    CL_TLOG(s * 8 + 1);
    // (3) contraction: sources l with tau = s-l-1 in [0,T): h^l_tau feeds layer l (recurrent) and layer l+1 (input)
    f32x4 acc[L][TM][4];
#pragma unroll
    for (int l = 0; l < L; ++l) zero_acc(acc[l]);
    f32x4 g0v[TM][4];   // layer 0's pre-activations, then its activated gates (interleaved form)
    bool did0 = false;  // layer 0's gate math already ran inside the contraction
    const int lo = s - T > 0 ? s - T : 0;
    const int hi = s - 1 < L - 1 ? s - 1 : L - 1;
    const int npan = hi >= lo ? (hi - lo + 1) * NPP : 0;
    auto issue = [&](int n) {
      const int l = lo + n / NPP, pp = n % NPP;
      const u16* src = p.hs + ((int64_t)(l * T + (s - l - 1)) * B) * H;
      glds_wave_panel<CF::WR, PCH>(ring + (n % kRing) * kPanel, src, H, pp * PCH * 8, arm, wrow0, lane);
    };
    if (wact) {
      for (int n = 0; n < kRing - 1 && n < npan; ++n) issue(n);
      if (fold && s < T) {  // layer 0's input projection for t = s, while the first panels are landing
#pragma unroll
        for (int j = 0; j < KSX; ++j) {
          if (j >= nkx) break;
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
              acc[0][tm][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wx[g][j]), __builtin_bit_cast(bf16x8, xc[tm][j]),
                                                                      acc[0][tm][g], 0, 0, 0);
        }
      }
      // L = 2, both sources present (every step from s = 2 on): the h^1 panels are multiplied by an unrolled loop below that
      // carries layer 0's gate math between its MFMAs (layer 0's accumulators are final once the h^0 panels are done)
      const bool two = L == 2 && hi > lo && p.il;
      const int npan_a = two ? NPP : npan;
      for (int n = 0; n < npan_a; ++n) {
        wait_panels(npan - 1 - n < kRing - 2 ? npan - 1 - n : kRing - 2);  // panel n has landed
        if (n + kRing - 1 < npan) issue(n + kRing - 1);  // into the slot consumed one iteration ago
        const int l = lo + n / NPP, pp = n % NPP;
        const char* As = ring + (n % kRing) * kPanel;
        {
          // one panel of source ll: every fragment of a k-step (TM of h, 4 of W_hh[ll], 4 of W_ih[ll+1]) is requested before
          // its MFMAs -- ONE LDS round trip per k-step (a read per MFMA pair costs 8: a wave is alone on its SIMD, nothing
          // hides that latency); REC = layer ll itself is active at this step
          auto panel = [&](auto ll_c, auto rec_c) {
            constexpr int ll = decltype(ll_c)::value;
            constexpr bool REC = decltype(rec_c)::value;
            constexpr bool UP = ll + 1 < L;
            constexpr int lu = UP ? ll + 1 : L - 1;
            const char* Whh = Wl + (2 * ll) * CF::W_BYTES;
            const char* Wih = Wl + (2 * ll + 1) * CF::W_BYTES;  // of layer ll+1 (exists when UP)
            constexpr int NJ = PCH / 4, NB = (REC ? 4 : 0) + (UP ? 4 : 0);
            bf16x8 a[2][TM], bh[2][4], bu[2][4];
            auto frags = [&](int j, int buf) {
              const int kc = pp * PCH + ((j << 2) | q);
#pragma unroll
              for (int tm = 0; tm < TM; ++tm)
                a[buf][tm] = __builtin_bit_cast(bf16x8, *(const uint4*)(As + kc_off<PCH>(tm * 16 + r, (j << 2) | q)));
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                if constexpr (REC) bh[buf][g] = __builtin_bit_cast(bf16x8, *(const uint4*)(Whh + kc_off<HC>(g * 16 + r, kc)));
                if constexpr (UP) bu[buf][g] = __builtin_bit_cast(bf16x8, *(const uint4*)(Wih + kc_off<HC>(g * 16 + r, kc)));
              }
            };
            frags(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TM + NB, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              if (j + 1 < NJ) frags(j + 1, (j + 1) & 1);  // the next k-step's fragments fly under this one's MFMAs
#pragma unroll
              for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                  if constexpr (REC) acc[ll][tm][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j & 1][g], a[j & 1][tm], acc[ll][tm][g], 0, 0, 0);
                  if constexpr (UP) acc[lu][tm][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bu[j & 1][g], a[j & 1][tm], acc[lu][tm][g], 0, 0, 0);
                }
              if (j + 1 < NJ) {
#pragma unroll
                for (int i = 0; i < TM + NB; ++i) {
                  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, TM * NB - (TM + NB), 0);
              } else {
                __builtin_amdgcn_sched_group_barrier(0x008, TM * NB, 0);
              }
            }
          };
          if (l == 0) {
            if (s < T)
              panel(std::integral_constant<int, 0>{}, std::true_type{});
            else
              panel(std::integral_constant<int, 0>{}, std::false_type{});
          }
          if constexpr (L > 1) {
            if (l == 1) {
              if (s - 1 < T)
                panel(std::integral_constant<int, 1>{}, std::true_type{});
              else
                panel(std::integral_constant<int, 1>{}, std::false_type{});
            }
          }
        }
      }
      if constexpr (L == 2) {
        if (two) {
          if (s < T) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
              for (int g = 0; g < 4; ++g) g0v[tm][g] = acc[0][tm][g] + padd[tm][g];
          }
          // chunk c = (tile c / 4, element c % 4) of layer 0's gate math: 10 transcendentals + the cell update of 1 (row, unit)
          auto g0_chunk = [&](int c) {
            const int tm = c >> 2, i = c & 3;
            const float ig = sigmoidf_(g0v[tm][0][i]), fg = sigmoidf_(g0v[tm][1][i]), gg = tanhf_(g0v[tm][2][i]), og = sigmoidf_(g0v[tm][3][i]);
            const float cn = __builtin_fmaf(fg, creg[0][tm][i], ig * gg);
            creg[0][tm][i] = cn;
            hreg[0][tm][i] = og * tanhf_(cn);
            g0v[tm][0][i] = ig, g0v[tm][1][i] = fg, g0v[tm][2][i] = gg, g0v[tm][3][i] = og;
          };
          auto second = [&](auto with_gates) {
            constexpr bool WG = decltype(with_gates)::value;
            constexpr int KS1 = NPP * (PCH / 4), NCH = TM * 4;  // k-steps of the h^1 source, gate chunks of layer 0
            const char* Whh = Wl + 2 * CF::W_BYTES;
#pragma unroll
            for (int pp = 0; pp < NPP; ++pp) {
              const int n = NPP + pp;
              wait_panels(2 * NPP - 1 - n < kRing - 2 ? 2 * NPP - 1 - n : kRing - 2);
              if (n + kRing - 1 < 2 * NPP) issue(n + kRing - 1);
              const char* As = ring + (n % kRing) * kPanel;
#pragma unroll
              for (int j = 0; j < PCH / 4; ++j) {
                bf16x8 a[TM], b[4];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                  a[tm] = __builtin_bit_cast(bf16x8, *(const uint4*)(As + kc_off<PCH>(tm * 16 + r, (j << 2) | q)));
                const int kc = pp * PCH + ((j << 2) | q);
#pragma unroll
                for (int g = 0; g < 4; ++g) b[g] = __builtin_bit_cast(bf16x8, *(const uint4*)(Whh + kc_off<HC>(g * 16 + r, kc)));
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                  for (int tm = 0; tm < TM; ++tm)
                    acc[1][tm][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[g], a[tm], acc[1][tm][g], 0, 0, 0);
                if constexpr (WG) {
                  const int kk = pp * (PCH / 4) + j;
#pragma unroll
                  for (int c = 0; c < NCH; ++c)
                    if (c >= kk * NCH / KS1 && c < (kk + 1) * NCH / KS1) g0_chunk(c);
                }
              }
            }
          };
          if (s < T) {
            second(std::true_type{});
            did0 = true;
          } else {
            second(std::false_type{});
          }
        }
      }
    }

    CL_TLOG(s * 8 + 2);
    if (s + 1 < T) {  // fly under the epilogue
      if (pvar) load_pre(s + 1);
      if (fold) load_x(s + 1);
    }
    // (4) gates + cell update for the active layers; h leaves first (it is what the other members wait for)
    if (wact) {
#pragma unroll
      for (int ll = 0; ll < L; ++ll) {
        const int t = s - ll;
        if (t < 0 || t >= T) continue;
        const int64_t lt = (int64_t)ll * T + t;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const int row = r0 + wrow0 + tm * 16 + r;
          if (ll == 0 && did0) {  // computed between the MFMAs of the h^1 panels
#pragma unroll
            for (int g = 0; g < 4; ++g) gpk[0][tm][g] = pack4(g0v[tm][g]);
            if (row < rend) *(uint2*)(p.hs + (lt * B + row) * H + uq) = pack4(hreg[0][tm]);
            continue;
          }
          f32x4 gv[4], c, h;
#pragma unroll
          for (int g = 0; g < 4; ++g) gv[g] = acc[ll][tm][g] + (ll == 0 ? padd[tm][g] : bias[ll][g]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float ig = sigmoidf_(gv[0][i]), fg = sigmoidf_(gv[1][i]), gg = tanhf_(gv[2][i]), og = sigmoidf_(gv[3][i]);
            c[i] = __builtin_fmaf(fg, creg[ll][tm][i], ig * gg);  // (explicit: see tanhf_)
            h[i] = og * tanhf_(c[i]);
            gv[0][i] = ig, gv[1][i] = fg, gv[2][i] = gg, gv[3][i] = og;
          }
          creg[ll][tm] = c;
          hreg[ll][tm] = h;
#pragma unroll
          for (int g = 0; g < 4; ++g) gpk[ll][tm][g] = pack4(gv[g]);
          if (row < rend) *(uint2*)(p.hs + (lt * B + row) * H + uq) = pack4(h);
        }
      }
    }
    CL_TLOG(s * 8 + 3);
    // (5) publish step s (also the barrier that frees the staging buffers for the next step)
    if (s + 1 < nsteps) cluster_publish(flags, me, ep0 + (unsigned)(s + 1));
    CL_TLOG(s * 8 + 4);
    // (6) what only the backward reads.  Not free: 20 partial-line store instructions per wave and step cost the wave ~130 ns
    // each wherever they are issued -- as this burst (2.6 us before the next flag poll), all behind the next step's first
    // panel issues (228 us per launch against 204), or two or three behind every panel issue of the next step (226): the
    // panels queue behind them; or four behind each of the next step's last three panel waits, i.e. behind its last panel
    // issue and ~1.5 us ahead of the h stores (201 against 190, with the cl_goff layout): a wave whose store queue is full
    // stalls its MFMAs too.  Fewer, fuller stores are what helps (the gate layout, cl_goff).
    tail_stores(s);
  }
}

// ---------------------------------------------------------------------------------------------
// backward, ONE LAYER per launch (rows form, L = 2 nets run layer 1, then the from-above contraction as a GEMM, then layer 0).
// In the wavefront kernel above every member reads dg of BOTH layers (rows x 4H x 2 B each) every step: 512 KB per CU and
// step at B = 2048, which is what bounds it (~10 TB/s of L2 reads over the chip).  A single layer only needs its own
// W_hh^T (HU = 32 units per member: 64 KB of LDS instead of 96 KB for 16), so clusters shrink to H/32 members:
// half as many readers per dg row and half as many rows' worth of K per step -> 128 KB per CU and step.  The from-above
// term dg^{l+1} . W_ih^{l+1} is not recurrent: it is one (T*B x 4H x H) contraction between the two launches, handed to the
// lower layer as its external gradient.  2T steps instead of T+1, each a quarter of the exchange.
// ---------------------------------------------------------------------------------------------
template <int H, int RB, int HU>
struct ClLayerCfg {
  static constexpr int G = 4 * H, GC = G / 8, KB = GC / 64, UT = HU / 16;
  static constexpr int TM = RB >= 64 ? RB / 64 : 1;
  static constexpr int WR = TM * 16;
  static constexpr int PCH = kPanel / (WR * 16);
  static constexpr int NPP = GC / PCH;
  static constexpr int W_BYTES = HU * GC * 16;
  static constexpr int SMEM = W_BYTES + 4 * kRing * kPanel;
  static constexpr int SMEM_HW = SMEM + 4 * 8192;  // + the helper wave's operand buffer
  static_assert(GC % 64 == 0 && GC % PCH == 0 && PCH >= 8 && HU % 16 == 0, "shape");
};

// HW (RB <= 64): a fifth wave does nothing but fetch the epilogue operands of step s+1 (saved gates, c_{t-1}, the external
// gradient: HBM) into LDS with LDS-DMA while the four compute waves work on step s.  Loads return in order per wave: issued
// by the compute waves themselves these loads sit in front of the flag poll (2.1 us of every 6.5-us step), and fetching them
// a step ahead from the same waves only moved the stall (the dg stores then queue behind them).
template <int H, int RB, int HU, bool HW>
__global__ __launch_bounds__(HW ? kThreads + 64 : kThreads) void lstm_bwd_layer_kernel(ClBwd p) {
  using CF = ClLayerCfg<H, RB, HU>;
  constexpr int G = CF::G, KB = CF::KB, UT = CF::UT, TM = CF::TM, PCH = CF::PCH, NPP = CF::NPP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Wl = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* ring = smem + CF::W_BYTES + wave * (kRing * kPanel);
  const int r = lane & 15, q = lane >> 4;

  const int joined = cluster_join(p.sync, (int*)(smem + CF::W_BYTES));
  if (joined < 0) return;
  const int info = joined & 255;                                 // XCD * 32 + slot
  const unsigned ep0 = (unsigned)(joined >> 8) * kSeqEpochs;     // this launch's number on the sync block
  const int NU = p.NU;
  const int cluster = (info >> 5) * (32 / NU) + (info & 31) / NU, me = (info & 31) % NU;
  const int r0 = p.row0 + cluster * p.Mc;
  const int rend = min(p.row0 + p.nrows, r0 + p.Mc);
  if (r0 >= rend) return;
  unsigned* flags = p.sync + kSyncFlags + cluster * 32;
  const int u0 = me * HU;
  const int B = p.B, T = p.T;
  {
    ClUnitMap um{u0};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
      if (tid < kThreads) glds_tile<u16, HU, 64>(Wl + kb * (HU * 1024), p.w_hh_t[0], G, 0, kb * 512, um, 0, tid);
  }
  static_assert(!HW || TM == 1, "the helper wave serves 16-row tiles");
  const int wrow0 = wave * (TM * 16);
  const bool helper = HW && wave == 4;
  const bool wact = wrow0 < RB && !helper;
  char* ops = smem + CF::W_BYTES + 4 * kRing * kPanel;  // HW: [tile][gates 4 KB | c_prev 2 KB | ext 2 KB]
  // helper: operands of step sn for every 16-row tile of the workgroup, one contiguous KB per DMA instruction
  auto fetch = [&](int sn) {
    const int t = T - 1 - sn;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w * 16 >= RB) break;
      char* op = ops + w * 8192;
      {
        const int rl = lane >> 2, piece = lane & 3;
        const int row0_ = r0 + w * 16 + rl;
        const int64_t row = row0_ < rend ? row0_ : rend - 1;
        // the member's 32 units = two 128-byte blocks of the row (cl_goff): image [16 rows][256 B], four rows per instruction
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row4_ = r0 + w * 16 + g * 4 + (lane >> 4);
          const int64_t row4 = row4_ < rend ? row4_ : rend - 1;
          const u16* src = p.gates + ((int64_t)t * B + row4) * G + (u0 >> 4) * 64 + (lane & 15) * 8;
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                           (void __attribute__((address_space(3)))*)(op + g * 1024), 16, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rl = i * 8 + (lane >> 3), piece = lane & 7;
        const int row0_ = r0 + w * 16 + rl;
        const int64_t row = row0_ < rend ? row0_ : rend - 1;
        if (t > 0) {
          const float* src = p.cs + ((int64_t)(t - 1) * B + row) * H + u0 + piece * 4;
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                           (void __attribute__((address_space(3)))*)(op + 4096 + i * 1024), 16, 0, 0);
        }
        if (p.d_hs_top) {
          const float* src = p.d_hs_top + ((int64_t)t * B + row) * H + u0 + piece * 4;
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                           (void __attribute__((address_space(3)))*)(op + 6144 + i * 1024), 16, 0, 0);
        }
      }
    }
  };
  f32x4 dcreg[TM][UT], ccur[TM][UT], dgs[TM][UT][4];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int ut = 0; ut < UT; ++ut) {
      dcreg[tm][ut] = ccur[tm][ut] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 4; ++g) dgs[tm][ut][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  ClRowMap arm{r0, rend - 1};
  auto pack4 = [](const f32x4& v) -> uint2 {
    return uint2{(uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16), (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16)};
  };
  unsigned long long* tl = (p.tlog && cluster == 0 && me == 0) ? p.tlog : nullptr;
  if constexpr (HW) {
    if (helper) fetch(0);
    __syncthreads();  // weights and the operands of step 0 have landed
  }

  for (int s = 0; s < T; ++s) {
    CL_TLOG(s * 8 + 0);
    const int t = T - 1 - s;
    // (1) saved activations, cell states, external gradient of time t: independent of the exchange
    uint2 gk[TM][UT][4];
    f32x4 cprev[TM][UT], ext[TM][UT];
    if constexpr (HW) {
      if (wact) {
        const char* op = ops + wave * 8192;
        const int row0_ = r0 + wrow0 + r;
        const int64_t row = row0_ < rend ? row0_ : rend - 1;
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
          const int ul = ut * 16 + q * 4, uq = u0 + ul;
          {
            const uint4 ga = *(const uint4*)(op + r * 256 + ut * 128 + q * 16), gb = *(const uint4*)(op + r * 256 + ut * 128 + 64 + q * 16);
            gk[0][ut][0] = uint2{ga.x, ga.y}, gk[0][ut][1] = uint2{ga.z, ga.w}, gk[0][ut][2] = uint2{gb.x, gb.y}, gk[0][ut][3] = uint2{gb.z, gb.w};
          }
          cprev[0][ut] = t > 0 ? *(const f32x4*)(op + 4096 + r * 128 + ul * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
          f32x4 e = p.d_hs_top ? *(const f32x4*)(op + 6144 + r * 128 + ul * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
          if (s == 0) {
            ccur[0][ut] = *(const f32x4*)(p.cs + ((int64_t)t * B + row) * H + uq);
            if (p.d_hn) e += *(const f32x4*)(p.d_hn + row * p.hn_ld + uq);
          }
          ext[0][ut] = e;
        }
      }
      __syncthreads();                      // every compute wave holds its operands: the helper may refill the buffer
      if (helper && s + 1 < T) fetch(s + 1);  // (its vmcnt is drained in front of the publish barrier below)
    } else
    if (wact) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int row0_ = r0 + wrow0 + tm * 16 + r;
        const int64_t row = row0_ < rend ? row0_ : rend - 1;
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
          const int uq = u0 + ut * 16 + q * 4;
          cl_load_gates(p.gates + ((int64_t)t * B + row) * G, uq, gk[tm][ut]);
          if (s == 0) ccur[tm][ut] = *(const f32x4*)(p.cs + ((int64_t)t * B + row) * H + uq);
          cprev[tm][ut] = t > 0 ? *(const f32x4*)(p.cs + ((int64_t)(t - 1) * B + row) * H + uq) : f32x4{0.f, 0.f, 0.f, 0.f};
          f32x4 e = f32x4{0.f, 0.f, 0.f, 0.f};
          if (p.d_hs_top) e = *(const f32x4*)(p.d_hs_top + ((int64_t)t * B + row) * H + uq);
          if (s == 0 && p.d_hn) e += *(const f32x4*)(p.d_hn + row * p.hn_ld + uq);
          ext[tm][ut] = e;
        }
      }
    }
    // (2) dg of time t+1 from every member
    if (s > 0 && !helper && !cluster_wait(p.sync, flags, NU, ep0 + (unsigned)s)) return;
    CL_TLOG(s * 8 + 1);

    // (3) dh = dg_{t+1} . W_hh (this member's HU units)
    f32x4 acc[TM][UT];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int ut = 0; ut < UT; ++ut) acc[tm][ut] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int npan = s > 0 ? NPP : 0;
    auto issue = [&](int n) {
      const u16* src = p.dg + ((int64_t)(t + 1) * B) * G;
      glds_wave_panel<CF::WR, PCH>(ring + (n % kRing) * kPanel, src, G, n * PCH * 8, arm, wrow0, lane);
    };
    if (wact) {
      for (int n = 0; n < kRing - 1 && n < npan; ++n) issue(n);
      for (int n = 0; n < npan; ++n) {
        wait_panels(npan - 1 - n < kRing - 2 ? npan - 1 - n : kRing - 2);
        if (n + kRing - 1 < npan) issue(n + kRing - 1);
        const char* As = ring + (n % kRing) * kPanel;
        // every fragment of a k-step is requested before its MFMAs, the next k-step's fly under them (one exposed LDS round
        // trip per panel: a wave is alone on its SIMD, nothing else hides that latency)
        constexpr int NJ = PCH / 4;
        bf16x8 a[2][TM], b[2][UT];
        auto frags = [&](int j, int buf) {
          const int kc = n * PCH + ((j << 2) | q);
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
            a[buf][tm] = __builtin_bit_cast(bf16x8, *(const uint4*)(As + kc_off<PCH>(tm * 16 + r, (j << 2) | q)));
#pragma unroll
          for (int ut = 0; ut < UT; ++ut)
            b[buf][ut] = __builtin_bit_cast(bf16x8, *(const uint4*)(Wl + (kc >> 6) * (HU * 1024) + kc_off<64>(ut * 16 + r, kc & 63)));
        };
        frags(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, TM + UT, 0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (j + 1 < NJ) frags(j + 1, (j + 1) & 1);
#pragma unroll
          for (int ut = 0; ut < UT; ++ut)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) acc[tm][ut] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j & 1][ut], a[j & 1][tm], acc[tm][ut], 0, 0, 0);
          if (j + 1 < NJ) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TM + UT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * UT - 1, 0);
          } else {
            __builtin_amdgcn_sched_group_barrier(0x008, TM * UT, 0);
          }
        }
      }
    }
    CL_TLOG(s * 8 + 2);

    // (4) elementwise LSTM backward -> dg_t
    if (wact) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int row = r0 + wrow0 + tm * 16 + r;
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
          const int uq = u0 + ut * 16 + q * 4;
          const f32x4 ig = unpack4(gk[tm][ut][0]), fg = unpack4(gk[tm][ut][1]), gg = unpack4(gk[tm][ut][2]), og = unpack4(gk[tm][ut][3]);
          const f32x4 dh = acc[tm][ut] + ext[tm][ut];
          f32x4 dp[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float tc = tanhf_(ccur[tm][ut][i]);
            float dc = dh[i] * og[i] * (1.f - tc * tc);
            if (s > 0) dc += dcreg[tm][ut][i];
            const float d_o = dh[i] * tc;
            const float d_i = dc * gg[i], d_f = dc * cprev[tm][ut][i], d_g = dc * ig[i];
            dcreg[tm][ut][i] = dc * fg[i];
            dp[0][i] = d_i * ig[i] * (1.f - ig[i]);
            dp[1][i] = d_f * fg[i] * (1.f - fg[i]);
            dp[2][i] = d_g * (1.f - gg[i] * gg[i]);
            dp[3][i] = d_o * og[i] * (1.f - og[i]);
          }
          ccur[tm][ut] = cprev[tm][ut];
          if (row < rend) {
            u16* go = p.dg + ((int64_t)t * B + row) * G + uq;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              dgs[tm][ut][g] += dp[g];
              *(uint2*)(go + g * H) = pack4(dp[g]);
            }
          }
        }
      }
    }
    CL_TLOG(s * 8 + 3);
    if (s + 1 < T) cluster_publish(flags, me, ep0 + (unsigned)(s + 1));
    CL_TLOG(s * 8 + 4);
  }
  if (wact) {
#pragma unroll
    for (int ut = 0; ut < UT; ++ut) {
      const int uq = u0 + ut * 16 + q * 4;
      if (p.dgsum) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const int row = r0 + wrow0 + tm * 16 + r;
          if (row >= rend) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g) *(f32x4*)(p.dgsum + (int64_t)row * G + g * H + uq) = dgs[tm][ut][g];
        }
      }
      if (p.db_ih[0] || p.db_hh[0]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v = dgs[0][ut][g];
#pragma unroll
          for (int tm = 1; tm < TM; ++tm) v += dgs[tm][ut][g];
          db_reduce_add(v, p.db_ih[0], p.db_hh[0], g * H + uq, lane);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// SMALL batches (at most 32 rows per cluster, B <= 512): the step is pure latency, so the four waves split the
// CONTRACTION instead of the rows.  Wave w = (row tile w / KSP, k-part w % KSP) loads its k-steps of the exchanged
// operand straight into registers (L1-bypassing buffer loads: a lane's 16 bytes are its MFMA fragment), multiplies
// them against the stationary weights, the partial tiles are summed through LDS and wave (rt, l) finishes layer l.
// One L2 round trip for the operand, a quarter of the MFMA / LDS work per wave.
// ---------------------------------------------------------------------------------------------
template <int H, int L, int RB>
__global__ __launch_bounds__(kThreads) void lstm_fwd_ksplit_kernel(ClFwd p) {
  constexpr int HC = H / 8, KS = H / 32;
  constexpr int KSP = 64 / RB;    // waves per row tile
  constexpr int KPW = KS / KSP;   // k32-steps of each source per wave
  constexpr int W_BYTES = 64 * HC * 16;
  constexpr int NW = 2 * L - 1;
  static_assert(KSP >= L && KS % KSP == 0, "k split");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Wl = smem;
  char* Part = smem + NW * W_BYTES;             // [wave][L][4] tiles of 1 KB
  int* s_word = (int*)(Part + 4 * L * 4 * 1024);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int rt = wave / KSP, kp = wave % KSP;

  const int joined = cluster_join(p.sync, s_word);
  if (joined < 0) return;
  const int info = joined & 255;                                 // XCD * 32 + slot
  const unsigned ep0 = (unsigned)(joined >> 8) * kSeqEpochs;     // this launch's number on the sync block
  const int NU = p.NU;
  const int cluster = (info >> 5) * (32 / NU) + (info & 31) / NU, me = (info & 31) % NU;
  const int r0 = p.row0 + cluster * p.Mc;
  const int rend = min(p.row0 + p.nrows, r0 + p.Mc);
  if (r0 >= rend) return;
  unsigned* flags = p.sync + kSyncFlags + cluster * 32;
  const int u0 = me * 16, uq = u0 + q * 4;
  const int B = p.B, T = p.T;
  {
    ClGateMap gm{H, u0};
#pragma unroll
    for (int l = 0; l < L; ++l) {
      glds_tile<u16, 64, HC>(Wl + (2 * l) * W_BYTES, p.w_hh[l], H, 0, 0, gm, 0, tid);
      if (l > 0) glds_tile<u16, 64, HC>(Wl + (2 * l - 1) * W_BYTES, p.w_ih[l], H, 0, 0, gm, 0, tid);
    }
  }
  const int row = r0 + rt * 16 + r;
  const int64_t rowc = row < rend ? row : rend - 1;
  const bool epi = kp < L;  // this wave finishes layer kp of its row tile
  f32x4 bias[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bias[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (epi && (kp > 0 || !p.pre)) bias[g] = *(const f32x4*)(p.b_ih[kp] + g * H + uq) + *(const f32x4*)(p.b_hh[kp] + g * H + uq);
  }
  // the time-constant input's projection (p.xcv): once, by the wave that finishes layer 0, into a constant added to its gates
  f32x4 pxc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) pxc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.xcv && kp == 0) {
    const int nkc = (p.Ic + 31) / 32, nchc = p.Ic / 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // Ic <= 128
      if (j >= nkc) break;
      const int c = j * 4 + q;
      uint4 xf = uint4{0u, 0u, 0u, 0u};
      if (c < nchc) xf = *(const uint4*)(p.xcv + rowc * p.Ic + c * 8);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint4 wf = uint4{0u, 0u, 0u, 0u};
        if (c < nchc) wf = *(const uint4*)(p.w_ih0 + (int64_t)(g * H + u0 + r) * p.K0 + p.I + c * 8);
        pxc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xf), pxc[g], 0, 0, 0);
      }
    }
  }
  f32x4 creg = f32x4{0.f, 0.f, 0.f, 0.f};
  auto pack4 = [](const f32x4& v) -> uint2 {
    return uint2{(uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16), (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16)};
  };
  // folded input projection: W_ih[0] fragments of this wave's k-steps (j = kp, kp + KSP, ...) stay in registers
  constexpr int KSXW = (4 + KSP - 1) / KSP;  // up to 128 input features
  const bool fold = p.x != nullptr;
  const int nchx = p.I / 8;
  uint4 wx[4][KSXW];
#pragma unroll
  for (int jj = 0; jj < KSXW; ++jj) {
    const int c = (kp + jj * KSP) * 4 + q;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint4 v = uint4{0u, 0u, 0u, 0u};
      if (fold && c < nchx) v = *(const uint4*)(p.w_ih0 + (int64_t)(g * H + u0 + r) * p.K0 + c * 8);
      wx[g][jj] = v;
    }
  }
  uint4 xn[KSXW];
  auto load_x = [&](int t) {
#pragma unroll
    for (int jj = 0; jj < KSXW; ++jj) {
      const int c = (kp + jj * KSP) * 4 + q;
      uint4 v = uint4{0u, 0u, 0u, 0u};
      if (fold && c < nchx) v = *(const uint4*)(p.x + ((int64_t)t * B + rowc) * p.I + c * 8);
      xn[jj] = v;
    }
  };
  load_x(0);
  const __amdgpu_buffer_rsrc_t hs_rs = make_rsrc(p.xch);
  unsigned long long* tl = (p.tlog && cluster == 0 && me == 0) ? p.tlog : nullptr;
  __syncthreads();  // weights have landed

  const int nsteps = T + L - 1;
  for (int s = 0; s < nsteps; ++s) {
    CL_TLOG(s * 8 + 0);
    f32x4 padd[4];
    if (kp == 0 && s < T && p.pre) {
      const float* pp = p.pre + (int64_t)s * p.pre_tstride + rowc * (4 * H) + uq;
#pragma unroll
      for (int g = 0; g < 4; ++g) padd[g] = *(const f32x4*)(pp + g * H);
    }
    uint4 xc[KSXW];  // this wave's k-steps of x_s (folded input projection), fetched during step s-1
#pragma unroll
    for (int jj = 0; jj < KSXW; ++jj) xc[jj] = xn[jj];
    if (s > 0 && !cluster_wait(p.sync, flags, NU, ep0 + (unsigned)s)) return;
    CL_TLOG(s * 8 + 1);

    uint4 a[L][KPW];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int tau = s - l - 1;
      if (tau < 0 || tau >= T) continue;
      const int64_t base = (xch_off((s - 1) & 1, l, L, KS, kp * KPW, B, rowc) + q * 8) * 2;
#pragma unroll
      for (int j = 0; j < KPW; ++j) a[l][j] = load_sc1(hs_rs, base + j * (B * 64));
    }
    f32x4 acc[L][4];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[l][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the weight fragments do not depend on the exchange: all those of a source are requested together while the exchange
    // loads are in flight (a read in front of every MFMA exposed an LDS round trip per MFMA: a wave is alone on its SIMD)
    bf16x8 wh[L][KPW][4], wu[L][KPW][4];
    auto wfrags = [&](int l) {
      const char* Whh = Wl + (2 * l) * W_BYTES;
      const char* Wih = Wl + (2 * l + 1) * W_BYTES;
#pragma unroll
      for (int j = 0; j < KPW; ++j) {
        const int kc = ((kp * KPW + j) << 2) | q;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          wh[l][j][g] = __builtin_bit_cast(bf16x8, *(const uint4*)(Whh + kc_off<HC>(g * 16 + r, kc)));
          if (l + 1 < L) wu[l][j][g] = __builtin_bit_cast(bf16x8, *(const uint4*)(Wih + kc_off<HC>(g * 16 + r, kc)));
        }
      }
    };
    wfrags(0);
    __builtin_amdgcn_sched_barrier(0);
    if (fold && s < T) {
#pragma unroll
      for (int jj = 0; jj < KSXW; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[0][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wx[g][jj]), __builtin_bit_cast(bf16x8, xc[jj]), acc[0][g], 0,
                                                              0, 0);
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
      if (l + 1 < L) wfrags(l + 1);  // the next source's fragments fly under this one's MFMAs
      const int tau = s - l - 1;
      if (tau < 0 || tau >= T) continue;
      const bool rec = s - l < T;
#pragma unroll
      for (int j = 0; j < KPW; ++j) {
        const bf16x8 av = __builtin_bit_cast(bf16x8, a[l][j]);
        if (rec) {
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[l][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[l][j][g], av, acc[l][g], 0, 0, 0);
        }
        if (l + 1 < L) {
          constexpr int kTop = L - 1;
          const int lu = l + 1 < L ? l + 1 : kTop;
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[lu][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wu[l][j][g], av, acc[lu][g], 0, 0, 0);
        }
      }
    }
    if (s + 1 < T) load_x(s + 1);  // (HBM: must not sit in front of the next flag poll)
    // partial tiles -> LDS, then wave (rt, l) sums the KSP parts of layer l
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
      for (int g = 0; g < 4; ++g) *(f32x4*)(Part + ((wave * L + l) * 4 + g) * 1024 + lane * 16) = acc[l][g];
    __syncthreads();
    CL_TLOG(s * 8 + 2);
    const int t = s - kp;
    const bool act = epi && t >= 0 && t < T;
    uint2 gpk[4];
    f32x4 hreg;
    if (act) {
      f32x4 gv[4], c;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        gv[g] = ((kp == 0 && p.pre) ? padd[g] : bias[g]) + pxc[g];
#pragma unroll
        for (int k = 0; k < KSP; ++k) gv[g] += *(const f32x4*)(Part + (((rt * KSP + k) * L + kp) * 4 + g) * 1024 + lane * 16);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float ig = sigmoidf_(gv[0][i]), fg = sigmoidf_(gv[1][i]), gg = tanhf_(gv[2][i]), og = sigmoidf_(gv[3][i]);
        c[i] = __builtin_fmaf(fg, creg[i], ig * gg);
        hreg[i] = og * tanhf_(c[i]);
        gv[0][i] = ig, gv[1][i] = fg, gv[2][i] = gg, gv[3][i] = og;
      }
      creg = c;
#pragma unroll
      for (int g = 0; g < 4; ++g) gpk[g] = pack4(gv[g]);
      if (row < rend) {
        const uint2 hp = pack4(hreg);
        *(uint2*)(p.xch + xch_off(s & 1, kp, L, KS, uq >> 5, B, row) + (uq & 31)) = hp;  // what the members wait for
        *(uint2*)(p.hs + (((int64_t)kp * T + t) * B + row) * H + uq) = hp;
      }
    }
    CL_TLOG(s * 8 + 3);
    if (s + 1 < nsteps) cluster_publish(flags, me, ep0 + (unsigned)(s + 1));  // (its barrier also frees Part)
    CL_TLOG(s * 8 + 4);
    if (act && row < rend) {
      const int64_t lt = (int64_t)kp * T + t;
      *(f32x4*)(p.cs + (lt * B + row) * H + uq) = creg;
      cl_store_gates(p.gates + (lt * B + row) * (4 * H), uq, gpk);
      if (kp == L - 1 && p.hs_top_f32) *(f32x4*)(p.hs_top_f32 + ((int64_t)t * B + row) * H + uq) = hreg;
      if (p.hn && t == T - 1) {
        *(f32x4*)(p.hn + (int64_t)row * (L * H) + kp * H + uq) = hreg;
        if (p.hn_lp) *(uint2*)(p.hn_lp + (int64_t)row * (L * H) + kp * H + uq) = pack4(hreg);  // (fhvae_lstm_desc.hn_lp)
      }
    }
  }
}

template <int H, int L, int RB>
__global__ __launch_bounds__(kThreads) void lstm_bwd_ksplit_kernel(ClBwd p) {
  constexpr int G = 4 * H, GC = G / 8, KB = GC / 64, KS = G / 32;
  constexpr int KSP = 64 / RB;
  constexpr int KPW = KS / KSP;  // k32-steps of each source per wave (H = 256: 8 or 16)
  constexpr int W_BYTES = 16 * GC * 16;
  constexpr int NW = 2 * L - 1;
  static_assert(KSP >= L && KS % KSP == 0, "k split");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Wl = smem;
  char* Part = smem + NW * W_BYTES;  // [wave][L] tiles of 1 KB
  int* s_word = (int*)(Part + 4 * L * 1024);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int rt = wave / KSP, kp = wave % KSP;

  const int joined = cluster_join(p.sync, s_word);
  if (joined < 0) return;
  const int info = joined & 255;                                 // XCD * 32 + slot
  const unsigned ep0 = (unsigned)(joined >> 8) * kSeqEpochs;     // this launch's number on the sync block
  const int NU = p.NU;
  const int cluster = (info >> 5) * (32 / NU) + (info & 31) / NU, me = (info & 31) % NU;
  const int r0 = p.row0 + cluster * p.Mc;
  const int rend = min(p.row0 + p.nrows, r0 + p.Mc);
  if (r0 >= rend) return;
  unsigned* flags = p.sync + kSyncFlags + cluster * 32;
  const int u0 = me * 16, uq = u0 + q * 4;
  const int B = p.B, T = p.T;
  {
    ClUnitMap um{u0};
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        glds_tile<u16, 16, 64>(Wl + (2 * l) * W_BYTES + kb * 16384, p.w_hh_t[l], G, 0, kb * 512, um, 0, tid);
        if (l > 0) glds_tile<u16, 16, 64>(Wl + (2 * l - 1) * W_BYTES + kb * 16384, p.w_ih_t[l], G, 0, kb * 512, um, 0, tid);
      }
  }
  const int row = r0 + rt * 16 + r;
  const int64_t rowc = row < rend ? row : rend - 1;
  const bool epi = kp < L;  // this wave finishes layer kp of its row tile
  f32x4 dcreg = f32x4{0.f, 0.f, 0.f, 0.f}, ccur = dcreg, dgs[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) dgs[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto pack4 = [](const f32x4& v) -> uint2 {
    return uint2{(uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16), (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16)};
  };
  const __amdgpu_buffer_rsrc_t dg_rs = make_rsrc(p.xch);
  unsigned long long* tl = (p.tlog && cluster == 0 && me == 0) ? p.tlog : nullptr;
  __syncthreads();

  uint2 gkn[4];
  f32x4 cprevn, extn, ccurn;
  auto load_epi = [&](int sn) {
    const int t = T - 1 - (sn - (L - 1 - kp));
    if (!(epi && t >= 0 && t < T)) return;
    const int64_t lt = (int64_t)kp * T + t;
    cl_load_gates(p.gates + (lt * B + rowc) * G, uq, gkn);
    if (t == T - 1) ccurn = *(const f32x4*)(p.cs + (lt * B + rowc) * H + uq);
    cprevn = t > 0 ? *(const f32x4*)(p.cs + ((lt - 1) * B + rowc) * H + uq) : f32x4{0.f, 0.f, 0.f, 0.f};
    extn = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kp == L - 1 && p.d_hs_top) extn = *(const f32x4*)(p.d_hs_top + ((int64_t)t * B + rowc) * H + uq);
    if (t == T - 1 && p.d_hn) extn += *(const f32x4*)(p.d_hn + rowc * p.hn_ld + kp * H + uq);
  };
  load_epi(0);
  const int nsteps = T + L - 1;
  for (int s = 0; s < nsteps; ++s) {
    CL_TLOG(s * 8 + 0);
    const int t = T - 1 - (s - (L - 1 - kp));  // the time layer kp handles at this step
    const bool act = epi && t >= 0 && t < T;
    // the saved activations / cell state / upstream gradient of this step were fetched during step s-1 (they come from
    // HBM and loads return in order: fetched here they would sit in front of the flag poll)
    uint2 gk[4], dpk[4];
    f32x4 cprev = cprevn, ext = extn;
#pragma unroll
    for (int g = 0; g < 4; ++g) gk[g] = gkn[g];
    if (act && t == T - 1) ccur = ccurn;
    if (s > 0 && !cluster_wait(p.sync, flags, NU, ep0 + (unsigned)s)) return;
    CL_TLOG(s * 8 + 1);

    f32x4 acc[L];
#pragma unroll
    for (int l = 0; l < L; ++l) acc[l] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 a[L][KPW];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int tau = T - s + (L - 1 - l);
      if (s == 0 || tau < 0 || tau > T - 1) continue;
      const int64_t base = (xch_off((s - 1) & 1, l, L, KS, kp * KPW, B, rowc) + q * 8) * 2;
#pragma unroll
      for (int j = 0; j < KPW; ++j) a[l][j] = load_sc1(dg_rs, base + j * (B * 64));
    }
    // the weight fragments do not depend on the exchange: those of a source are requested together while the exchange loads
    // are in flight, the next source's under this one's MFMAs (a read in front of every MFMA exposed an LDS round trip each)
    bf16x8 wh[L][KPW], wu[L][KPW];
    auto wfrags = [&](int l) {
      const char* Whh = Wl + (2 * l) * W_BYTES;
      const char* Wih = Wl + (l > 0 ? 2 * l - 1 : 0) * W_BYTES;
#pragma unroll
      for (int j = 0; j < KPW; ++j) {
        const int kc = ((kp * KPW + j) << 2) | q;
        const int woff = (kc >> 6) * 16384 + kc_off<64>(r, kc & 63);
        wh[l][j] = __builtin_bit_cast(bf16x8, *(const uint4*)(Whh + woff));
        if (l > 0) wu[l][j] = __builtin_bit_cast(bf16x8, *(const uint4*)(Wih + woff));
      }
    };
    if (s > 0) wfrags(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int l = 0; l < L; ++l) {
      if (l + 1 < L && s > 0) wfrags(l + 1);
      const int tau = T - s + (L - 1 - l);
      if (s == 0 || tau < 0 || tau > T - 1) continue;
      const bool rec = tau - 1 >= 0;
#pragma unroll
      for (int j = 0; j < KPW; ++j) {
        const bf16x8 av = __builtin_bit_cast(bf16x8, a[l][j]);
        if (rec) acc[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[l][j], av, acc[l], 0, 0, 0);
        if (l > 0) {
          const int ld = l > 0 ? l - 1 : 0;
          acc[ld] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wu[l][j], av, acc[ld], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int l = 0; l < L; ++l) *(f32x4*)(Part + (wave * L + l) * 1024 + lane * 16) = acc[l];
    if (s + 1 < nsteps) load_epi(s + 1);
    __syncthreads();
    CL_TLOG(s * 8 + 2);
    if (act) {
      f32x4 dh = ext;
#pragma unroll
      for (int k = 0; k < KSP; ++k) dh += *(const f32x4*)(Part + ((rt * KSP + k) * L + kp) * 1024 + lane * 16);
      const f32x4 ig = unpack4(gk[0]), fg = unpack4(gk[1]), gg = unpack4(gk[2]), og = unpack4(gk[3]);
      f32x4 dp[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float tc = tanhf_(ccur[i]);
        float dc = dh[i] * og[i] * (1.f - tc * tc);
        if (t != T - 1) dc += dcreg[i];
        const float d_o = dh[i] * tc;
        const float d_i = dc * gg[i], d_f = dc * cprev[i], d_g = dc * ig[i];
        dcreg[i] = dc * fg[i];
        dp[0][i] = d_i * ig[i] * (1.f - ig[i]);
        dp[1][i] = d_f * fg[i] * (1.f - fg[i]);
        dp[2][i] = d_g * (1.f - gg[i] * gg[i]);
        dp[3][i] = d_o * og[i] * (1.f - og[i]);
      }
      ccur = cprev;
      if (row < rend) {
#pragma unroll
        for (int g = 0; g < 4; ++g) dgs[g] += dp[g];
      }
      if (row < rend) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          dpk[g] = pack4(dp[g]);
          *(uint2*)(p.xch + xch_off(s & 1, kp, L, KS, (g * H + uq) >> 5, B, row) + (uq & 31)) = dpk[g];  // what the members wait for
        }
      }
    }
    CL_TLOG(s * 8 + 3);
    if (s + 1 < nsteps) cluster_publish(flags, me, ep0 + (unsigned)(s + 1));
    CL_TLOG(s * 8 + 4);
    if (act && row < rend) {  // the row-major copy the weight-gradient contractions read: after the publish
      u16* go = p.dg + (((int64_t)kp * T + t) * B + row) * G + uq;
#pragma unroll
      for (int g = 0; g < 4; ++g) *(uint2*)(go + g * H) = dpk[g];
    }
  }
  if (p.dgsum && kp == 0 && row < rend) {
#pragma unroll
    for (int g = 0; g < 4; ++g) *(f32x4*)(p.dgsum + (int64_t)row * G + g * H + uq) = dgs[g];
    // one lane per row clears the row of d_xc for the split-K contraction that follows (lstm.hip: no zeroing launch)
    if (p.d_xc_zero && uq == 0)
      for (int c = 0; c < p.Ic; ++c) p.d_xc_zero[(int64_t)row * p.Ic + c] = 0.f;
  }
  if (epi && (p.db_ih[kp] || p.db_hh[kp])) {
#pragma unroll
    for (int g = 0; g < 4; ++g) db_reduce_add(dgs[g], p.db_ih[kp], p.db_hh[kp], g * H + uq, lane);
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static bool device_ok() {
  static int ok = -1;
  if (ok < 0) {
    int dev = 0;
    hipDeviceProp_t pr;
    ok = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess)
      ok = (strncmp(pr.gcnArchName, "gfx950", 6) == 0 && pr.multiProcessorCount == kGrid) ? 1 : 0;
  }
  return ok == 1;
}

bool cluster_eligible(const fhvae_lstm_desc* d) {
  if (getenv("FHVAE_NO_CLUSTER")) return false;
  if (d->dtype != FHVAE_BF16 || !d->lp) return false;
  if (d->H != 256 && d->H != 128) return false;
  if (d->L > 2 || d->T + d->L >= kSeqEpochs) return false;
  if (d->B > 131072) return false;  // 32-bit byte offsets into the exchange buffer (buffer loads)
  return device_ok();
}

template <int H, int L, int RB>
static int launch_fwd_rb(const ClFwd& p, hipStream_t st) {
  using CF = ClFwdCfg<H, L, RB>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)lstm_fwd_cluster_kernel<H, L, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL((lstm_fwd_cluster_kernel<H, L, RB>), dim3(kGrid), dim3(kThreads), CF::SMEM, st, p);
  return fh_launch_status();
}

template <int H, int L, int RB>
static int launch_fwd_ks(const ClFwd& p, hipStream_t st) {
  constexpr int SMEM = (2 * L - 1) * 64 * (H / 8) * 16 + 4 * L * 4 * 1024 + 16;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)lstm_fwd_ksplit_kernel<H, L, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL((lstm_fwd_ksplit_kernel<H, L, RB>), dim3(kGrid), dim3(kThreads), SMEM, st, p);
  return fh_launch_status();
}

template <int H, int L>
static int launch_fwd(const ClFwd& p, int RB, hipStream_t st) {
  switch (RB) {
    case 16: return launch_fwd_ks<H, L, 16>(p, st);
    case 32: return launch_fwd_ks<H, L, 32>(p, st);
    case 64: return launch_fwd_rb<H, L, 64>(p, st);
    default: return launch_fwd_rb<H, L, 128>(p, st);
  }
}

// rows per cluster and the tile that holds them
static void cluster_rows(int64_t nrows, int NC, int* Mc, int* RB) {
  int64_t m = (nrows + NC - 1) / NC;
  m = (m + 15) / 16 * 16;
  *Mc = (int)m;
  *RB = m <= 16 ? 16 : m <= 32 ? 32 : m <= 64 ? 64 : 128;
}

// the forward kernels project the time-constant input themselves (no GEMM, no (B,4H) f32 round trip)
bool cluster_xc_in_kernel(const fhvae_lstm_desc* d) {
  if (!(d->Ic > 0 && d->Ic % 8 == 0 && d->Ic <= 128 && d->I % 8 == 0 && (d->I == 0 || cluster_can_fold(d))) || getenv("FHVAE_NO_XC_FOLD"))
    return false;
  return true;
}

bool cluster_can_fold(const fhvae_lstm_desc* d) {
  return d->I > 0 && d->I % 8 == 0 && d->I <= 128 && (d->I + d->Ic) % 8 == 0 && !getenv("FHVAE_NO_FOLD");
}

int cluster_form(const fhvae_lstm_desc* d) {
  const int NC = kGrid / ((int)d->H / 16);
  const int64_t chunk = (int64_t)NC * 128;
  int Mc, RB;
  cluster_rows(d->B < chunk ? d->B : chunk, NC, &Mc, &RB);
  return RB <= 32 ? 2 : 1;
}

static bool cluster_bwd_rs(const fhvae_lstm_desc* d);

// H = 256, two layers, rows form, the inputs foldable: the forward with register-stationary weights (lstm_fwd_wr.hip).  It saves the
// activated gates unit-major, which only the partial-dh backward reads (ClBwd::gates_um): both follow from this one predicate.
bool cluster_fwd_wr_ok(const fhvae_lstm_desc* d) {
  if (!cluster_bwd_rs(d) || d->L != 2 || getenv("FHVAE_NO_FWD_WR")) return false;
  if (d->I > 0 && !cluster_can_fold(d)) return false;
  if (d->Ic > 0 && !cluster_xc_in_kernel(d)) return false;
  if ((int64_t)2 * d->T * d->B * d->H * 2 >= (1LL << 31) || (int64_t)d->T * d->B * d->I * 2 >= (1LL << 31)) return false;  // 32-bit buffer offsets
  return d->I > 0 || d->Ic > 0;
}

int cluster_fwd(const fhvae_lstm_desc* d, const ClusterWeights& w, hipStream_t st) {
  const int H = (int)d->H, L = d->L;
  const bool wr = cluster_fwd_wr_ok(d);
  const int NU = wr ? 8 : H / 16, NC = kGrid / NU;
  const int64_t chunk = (int64_t)NC * (wr ? 64 : 128);
  for (int64_t row0 = 0; row0 < d->B; row0 += chunk) {
    const int64_t nrows = d->B - row0 < chunk ? d->B - row0 : chunk;
    ClFwd p = {};
    int RB;
    cluster_rows(nrows, NC, &p.Mc, &RB);
    p.B = (int)d->B;
    p.T = (int)d->T;
    p.NU = NU;
    p.row0 = (int)row0;
    p.nrows = (int)nrows;
    for (int l = 0; l < L; ++l) {
      p.w_ih[l] = w.w_ih[l];
      p.w_hh[l] = w.w_hh[l];
      p.b_ih[l] = d->b_ih[l];
      p.b_hh[l] = d->b_hh[l];
    }
    if (w.x_fold) {  // lstm.hip has left only the time-constant part (or nothing) in d->pre
      p.x = w.x_fold;
      p.w_ih0 = w.w_ih[0];
      p.I = (int)d->I;
      p.K0 = (int)(d->I + d->Ic);
      p.pre = d->Ic > 0 ? d->pre : nullptr;
      p.pre_tstride = 0;
    } else {
      p.pre = d->pre;
      p.pre_tstride = d->I > 0 ? d->B * 4 * d->H : 0;
    }
    if (w.xc_fold) {  // (cluster_xc_in_kernel) nothing was left in d->pre: the kernel projects xc itself
      p.xcv = w.xc_fold;
      p.Ic = (int)d->Ic;
      p.w_ih0 = w.w_ih[0];
      p.I = (int)d->I;
      p.K0 = (int)(d->I + d->Ic);
      p.pre = nullptr;
      p.pre_tstride = 0;
    }
    p.hs = (u16*)d->hs;
    p.cs = d->cs;
    p.gates = (u16*)d->gates;
    p.hs_top_f32 = d->hs_top_f32;
    p.hn = d->hn;
    p.hn_lp = (u16*)d->hn_lp;  // every persistent forward stores the bf16 copy beside hn
    p.sync = (unsigned*)d->lp;
    p.xch = w.xch;
    p.tlog = getenv("FHVAE_CLUSTER_TLOG") ? (unsigned long long*)((char*)d->lp + FHVAE_LSTM_SYNC_BYTES * 3 / 4) : nullptr;
    p.il = 1;
    double fl = 0;
    for (int l = 0; l < L; ++l) fl += 2.0 * nrows * 4 * H * ((l > 0 ? d->T * H : 0) + (d->T - 1) * (double)H);
    const int ts = trace_begin(st, kTraceFwdCell, fl);
    int e;
    if (wr) {
      p.gates_um = 1;
      e = cluster_fwd_wr(p, st);
    } else if (H == 256)
      e = L == 1 ? launch_fwd<256, 1>(p, RB, st) : launch_fwd<256, 2>(p, RB, st);
    else
      e = L == 1 ? launch_fwd<128, 1>(p, RB, st) : launch_fwd<128, 2>(p, RB, st);
    trace_end(st, ts);
    if (e) return e;
  }
  return FHVAE_OK;
}

template <int H, int L, int RB>
static int launch_bwd_ks(const ClBwd& p, hipStream_t st) {
  constexpr int SMEM = (2 * L - 1) * 16 * (4 * H / 8) * 16 + 4 * L * 1024 + 16;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)lstm_bwd_ksplit_kernel<H, L, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL((lstm_bwd_ksplit_kernel<H, L, RB>), dim3(kGrid), dim3(kThreads), SMEM, st, p);
  return fh_launch_status();
}

template <int H, int L>
static int launch_bwd(const ClBwd& p, int RB, hipStream_t st) {
  return RB <= 16 ? launch_bwd_ks<H, L, 16>(p, st) : launch_bwd_ks<H, L, 32>(p, st);
}

template <int H, int RB>
static int launch_bwd_layer(const ClBwd& p, hipStream_t st) {
  using CF = ClLayerCfg<H, RB, 32>;
  constexpr bool HW = RB <= 64 && CF::SMEM_HW <= 163840;  // helper wave where its operand buffer fits
  constexpr int SMEM = HW ? CF::SMEM_HW : CF::SMEM;
  const bool hw = HW;
  static bool attr[2] = {false, false};
  if (!attr[hw]) {
    hipError_t e = hw ? hipFuncSetAttribute((const void*)lstm_bwd_layer_kernel<H, RB, 32, HW>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM)
                      : hipFuncSetAttribute((const void*)lstm_bwd_layer_kernel<H, RB, 32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
    if (e != hipSuccess) return (int)e;
    attr[hw] = true;
  }
  if (hw)
    hipLaunchKernelGGL((lstm_bwd_layer_kernel<H, RB, 32, HW>), dim3(kGrid), dim3(kThreads + 64), SMEM, st, p);
  else
    hipLaunchKernelGGL((lstm_bwd_layer_kernel<H, RB, 32, false>), dim3(kGrid), dim3(kThreads), CF::SMEM, st, p);
  return fh_launch_status();
}
template <int H>
static int launch_bwd_layer_rb(const ClBwd& p, int RB, hipStream_t st) {
  switch (RB) {
    case 16: return launch_bwd_layer<H, 16>(p, st);
    case 32: return launch_bwd_layer<H, 32>(p, st);
    case 64: return launch_bwd_layer<H, 64>(p, st);
    default: return launch_bwd_layer<H, 128>(p, st);
  }
}


// H = 256, rows form: the per-layer backward exchanges partial dh (lstm_bwd_rs.hip)
static bool cluster_bwd_rs(const fhvae_lstm_desc* d) {
  return cluster_eligible(d) && cluster_form(d) == 1 && d->H == 256 && !getenv("FHVAE_NO_RS");
}

// rows form, layer by layer: the top layer's recurrence as one persistent launch, then for every layer below the from-above term
// dg^{l+1} . W_ih[l+1] of ALL steps as one GEMM into bd->ws_below (it is not recurrent), then that layer's launch with it as the
// external gradient.  H = 256: 64 units per member, partial-dh exchange (lstm_bwd_rs.hip, 2048 rows per launch, larger batches
// as consecutive launches) and the projection kernel (proj.hip); else 32 units per member, dg exchanged, generic engine.
static int cluster_bwd_layers(const fhvae_lstm_bwd_desc* bd, const ClusterWeights& w, hipStream_t st) {
  const fhvae_lstm_desc* d = &bd->f;
  const int H = (int)d->H, L = d->L;
  const bool rs = cluster_bwd_rs(d);
  const int HU = rs ? 64 : 32;
  const int NU = H / HU, NC = kGrid / NU;
  const int64_t B = d->B, T = d->T, G = 4 * H;
  const int64_t chunk = (int64_t)NC * (rs ? 32 : 128);
  for (int l = L - 1; l >= 0; --l) {
    if (l < L - 1) {  // ws_below[T*B, H] = dg^{l+1} [T*B, 4H] . W_ih[l+1]  (its transposed bf16 copy [H,4H] is the K-contiguous operand)
      const u16* dg_up = (const u16*)bd->dgates + (int64_t)(l + 1) * T * B * G;
      int e;
      if (rs && proj_eligible(dg_up, G, w.w_ih_t[l + 1], G, bd->ws_below, H, T * B, H, G)) {
        e = launch_proj(dg_up, G, w.w_ih_t[l + 1], G, bd->ws_below, H, nullptr, T * B, H, G, st);
      } else {
        GemmParams g = {};
        g.seg[0] = Seg{dg_up, G, 1, w.w_ih_t[l + 1], G, 1, (int)G, 0};
        g.M = (int)(T * B);
        g.N = H;
        g.C = bd->ws_below;
        g.ldc = H;
        g.splitk = 1;
        e = launch_gemm(g, FHVAE_BF16, st);
      }
      if (e) return e;
    }
    for (int64_t row0 = 0; row0 < B; row0 += chunk) {
      const int64_t nrows = B - row0 < chunk ? B - row0 : chunk;
      ClBwd p = {};
      int RB;
      cluster_rows(nrows, NC, &p.Mc, &RB);
      p.B = (int)B;
      p.T = (int)T;
      p.NU = NU;
      p.row0 = (int)row0;
      p.nrows = (int)nrows;
      p.w_hh_t[0] = w.w_hh_t[l];
      p.gates = (const u16*)d->gates + (int64_t)l * T * B * G;
      p.cs = d->cs + (int64_t)l * T * B * H;
      p.d_hs_top = l == L - 1 ? bd->d_hs_top : bd->ws_below;
      p.d_hn = bd->d_hn ? bd->d_hn + (int64_t)l * H : nullptr;
      p.hn_ld = L * H;
      p.dg = (u16*)bd->dgates + (int64_t)l * T * B * G;
      p.dgsum = (l == 0 && d->Ic > 0) ? bd->dgsum : nullptr;
      p.db_ih[0] = bd->db_ih[l];
      p.db_hh[0] = bd->db_hh[l];
      p.sync = (unsigned*)d->lp;
      p.xch = w.xch;
      p.tlog = getenv("FHVAE_CLUSTER_TLOG") ? (unsigned long long*)((char*)d->lp + FHVAE_LSTM_SYNC_BYTES * 3 / 4) : nullptr;
      p.tlog_slot = l == L - 1;
      p.nt = 1;  // streaming hints on its once-read operands: HBM reads 202 -> 175 MB per launch (1.04x algorithmic), 0.5 % of a step
      p.gates_um = cluster_fwd_wr_ok(d) ? 1 : 0;  // the forward on this workspace saved the gates unit-major (same predicate)
      if (rs && l == 0 && bd->d_xc && d->Ic > 0) p.d_xc_zero = bd->d_xc, p.Ic = (int)d->Ic;  // (cluster_bwd_zeroes_dxc)
      const int ts = trace_begin(st, kTraceBwdCell, 2.0 * nrows * H * (T - 1) * 4.0 * H);
      const int e = rs ? cluster_bwd_layer_rs(p, st) : (H == 256 ? launch_bwd_layer_rb<256>(p, RB, st) : launch_bwd_layer_rb<128>(p, RB, st));
      trace_end(st, ts);
      if (e) return e;
    }
  }
  return FHVAE_OK;
}

// the partial-dh backward and the contraction-split backward clear d_xc themselves (their layer-0 epilogue covers every row once)
bool cluster_bwd_zeroes_dxc(const fhvae_lstm_desc* d) { return cluster_bwd_rs(d) || cluster_form(d) == 2; }

// the layer-by-layer backward (rows form, two layers or more) hands the from-above gradient to the lower layer through
// bd->ws_below (T,B,H) f32
bool cluster_needs_ws_below(const fhvae_lstm_desc* d) { return cluster_eligible(d) && cluster_form(d) == 1 && d->L >= 2; }

int cluster_bwd(const fhvae_lstm_bwd_desc* bd, const ClusterWeights& w, hipStream_t st) {
  const fhvae_lstm_desc* d = &bd->f;
  const int H = (int)d->H, L = d->L;
  if (cluster_form(d) == 1) {
    if (cluster_needs_ws_below(d) && !bd->ws_below) return FHVAE_ERR_NULL;  // (fhvae_lstm_ws_below_elems says when)
    return cluster_bwd_layers(bd, w, st);
  }

  const int NU = H / 16, NC = kGrid / NU;
  const int64_t chunk = (int64_t)NC * 128;
  for (int64_t row0 = 0; row0 < d->B; row0 += chunk) {
    const int64_t nrows = d->B - row0 < chunk ? d->B - row0 : chunk;
    ClBwd p = {};
    int RB;
    cluster_rows(nrows, NC, &p.Mc, &RB);
    p.B = (int)d->B;
    p.T = (int)d->T;
    p.NU = NU;
    p.row0 = (int)row0;
    p.nrows = (int)nrows;
    for (int l = 0; l < L; ++l) {
      p.w_ih_t[l] = w.w_ih_t[l];
      p.w_hh_t[l] = w.w_hh_t[l];
    }
    p.gates = (const u16*)d->gates;
    p.cs = d->cs;
    p.d_hs_top = bd->d_hs_top;
    p.d_hn = bd->d_hn;
    p.hn_ld = L * H;
    p.dg = (u16*)bd->dgates;
    p.dgsum = d->Ic > 0 ? bd->dgsum : nullptr;
    if (bd->d_xc && d->Ic > 0) p.d_xc_zero = bd->d_xc, p.Ic = (int)d->Ic;  // (cluster_bwd_zeroes_dxc)
    for (int l = 0; l < L; ++l) p.db_ih[l] = bd->db_ih[l], p.db_hh[l] = bd->db_hh[l];
    p.sync = (unsigned*)d->lp;
    p.xch = w.xch;
    p.tlog = getenv("FHVAE_CLUSTER_TLOG") ? (unsigned long long*)((char*)d->lp + FHVAE_LSTM_SYNC_BYTES * 3 / 4) : nullptr;
    double fl = 0;
    for (int l = 0; l < L; ++l) fl += 2.0 * nrows * H * ((l < L - 1 ? d->T * 4.0 * H : 0) + (d->T - 1) * 4.0 * H);
    const int ts = trace_begin(st, kTraceBwdCell, fl);
    int e;
    if (H == 256)
      e = L == 1 ? launch_bwd<256, 1>(p, RB, st) : launch_bwd<256, 2>(p, RB, st);
    else
      e = L == 1 ? launch_bwd<128, 1>(p, RB, st) : launch_bwd<128, 2>(p, RB, st);
    trace_end(st, ts);
    if (e) return e;
  }
  return FHVAE_OK;
}

}  // namespace fh
