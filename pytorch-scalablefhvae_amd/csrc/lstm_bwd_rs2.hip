// lstm_bwd_rs2.hip -- K1 backward of a TWO-layer net, H = 256, rows form, BOTH layers in flight in ONE persistent launch
// (VERDICT r03 #1).  lstm_bwd_rs.hip runs a layer per launch: top layer (20 dependent steps) -> proj.hip (the from-above term
// dg^1 . W_ih^1 of all steps as one GEMM, a (T,B,H) f32 round trip) -> lower layer (20 more dependent steps).  Here the lower
// layer runs ONE step behind the top layer inside the same launch, T + 1 wavefront steps in all, and the from-above term never
// leaves the chip as a tensor:
//
//   super-cluster = 8 workgroups of ONE XCD, <= 64 batch rows: members 0..3 own layer 1 (the top), members 4..7 layer 0, member
//   m of either layer the hidden units [64 m, 64 m + 64).  32 super-clusters = 2048 rows per launch.
//   * both layers: the partial-dh recurrence of lstm_bwd_rs.hip unchanged (own K-slice of dg_t times the register-stationary
//     rows of W_hh^T, three bf16 partials out, three in per step) -- at up to four row tiles per wave instead of two;
//   * a layer-1 member ALSO holds its 256 rows of W_ih^1 (128 KB bf16, LDS, per-wave fragment order) and multiplies the dg^1_t
//     image it just produced by them: (rows x 256 units) PARTIAL from-above sums over its own 256 gate columns, handed to the
//     four layer-0 members as bf16 fragments in the accumulator layout of the destination wave, exactly like the recurrent
//     partials.  The product is issued BEHIND the recurrent publish, so it runs while the other layer-1 members' partials
//     are still on their way;
//   * a layer-0 member at its step s adds 4 from-above partials (published by layer 1 during ITS step s, one wavefront step
//     earlier) + 3 recurrent partials + its own.
// Flags (one 128-byte line per super-cluster, every wave polls it with one load): words 0..3 layer-1 recurrence, 4..7 layer-0
// recurrence, 8..11 from-above.  The from-above slots are a ring of kNS steps; before a layer-1 member starts step s it also
// waits for the layer-0 members to have finished their step s - kNS (back-pressure; layer 0 is the lighter role and is normally
// a whole step ahead of that bound).  Layer 1 never waits for anything else of layer 0, so there is no cycle; every spin is
// bounded and watches the abort word (lstm_cluster_dev.h).
// Body of the reference's stub (fhvae.py:14); arithmetic = torch.nn.LSTM backward.
#include <cstdlib>

#include "lstm_cluster_dev.h"
#include "trace.h"

namespace fh {

namespace {

constexpr int kH = 256, kG = 4 * kH, kHU = 64, kNU = 4;
constexpr int kKS = 8;    // k-steps (32 gate columns) of a member's own 256 gate columns
constexpr int kNS = 3;    // ring depth of the from-above slots
constexpr int kSC = 32;   // super-clusters per launch
constexpr int kWlBytes = kKS * kNU * 4 * 1024;  // W_ih^1 rows of a layer-1 member as per-wave fragments: 128 KB

// byte offsets into the exchange buffer.  Recurrent partials: [parity][super-cluster][layer][source][slot j = (dst - src) & 3
// in 1..3][wave][row tile][512 B]; from-above partials behind them: [ring slot][super-cluster][source][destination][wave][row tile][512 B]
constexpr int kRecBytes = 2 * kSC * 2 * kNU * 3 * 4 * 4 * 512;  // sized for four row tiles whatever RT is
__device__ __forceinline__ int rec_off(int par, int sc, int lay, int src, int j, int wave, int rt) {
  return (((((((par * kSC + sc) * 2 + lay) * kNU + src) * 3 + (j - 1)) * 4 + wave) * 4 + rt) << 9);
}
__device__ __forceinline__ int ab_off(int slot, int sc, int src, int dst, int wave, int rt) {
  return kRecBytes + (((((((slot * kSC + sc) * kNU + src) * kNU + dst) * 4 + wave) * 4 + rt)) << 9);
}
static_assert((int64_t)kRecBytes + (int64_t)kNS * kSC * kNU * kNU * 4 * 4 * 512 == kRs2XchElems * 2, "exchange buffer size (lstm_cluster.h)");

// every wave polls the super-cluster's flag line: lane i < 12 compares word i with its own target (0 = nothing to wait for)
__device__ __forceinline__ bool pair_wait(unsigned* sync, const unsigned* flags, unsigned target) {
  const int lane = threadIdx.x & 63;
  for (unsigned spins = 0;; ++spins) {
    unsigned v = 0xffffffffu, st = 0;
    if (lane < 12) v = __hip_atomic_load(flags + lane, RLX_AGENT);
    if (lane == 63) st = __hip_atomic_load(sync + kSyncStatus, RLX_AGENT);
    if (__any(st != 0)) return false;
    if (__all(v >= target)) break;
    if (spins > kSpinLimit) {
      if (lane == 0) cluster_give_up(sync, 1u);
      return false;
    }
  }
  asm volatile("" ::: "memory");
  return true;
}

// aux bit of a buffer access: non-temporal (streaming) -- on the once-read epilogue operands and the once-written row-major dg, so
// that they do not displace the exchanged partial lines from the XCD's L2 (lstm_bwd_rs.hip measured it: ClBwd::nt)
constexpr int kNt = 2;

template <int AUX>
__device__ __forceinline__ uint4 bld16(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, AUX);
  return uint4{v.x, v.y, v.z, v.w};
}
template <int AUX>
__device__ __forceinline__ f32x4 bldf4(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, AUX));
}

// one member's whole launch; TOP: a layer-1 member (the two roles are separate instantiations so that neither carries the other's
// registers: the from-above operands and the external gradient on one side, the per-row gate-gradient sums on the other).
// Every global access of the loop is a buffer instruction = (descriptor) + (one step-invariant per-lane VGPR offset) + (a scalar
// offset computed per step): written with pointers, hipcc hoisted ~100 loop-invariant 64-bit addresses into VGPRs and the
// four-row-tile form spilled 160 registers.
template <int RT, bool UM, bool TOP>
__device__ __forceinline__ void pair_member(const ClBwd& p, char* smem, int info, unsigned ep0) {
  constexpr int H = kH, G = kG;
  constexpr bool top = TOP;
  constexpr int lay = TOP ? 1 : 0;
  constexpr int DR = TOP ? 1 : RT;     // rows of gate-gradient sums a lane keeps: per row only where dgsum may be asked for
  char* Img = smem;                    // dg_t image: [RT*16 rows][32 chunks], XOR-swizzled (kc_off<32>)
  char* Wl = smem + RT * 8192;         // layer-1 members: W_ih^1 fragments [wave][ks][dst][lane] x 16 B
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int sc = (info >> 5) * 4 + ((info & 31) >> 3);           // super-cluster: 4 per XCD
  const int me = info & 3;
  const int r0 = p.row0 + sc * p.Mc;
  const int rend = min(p.row0 + p.nrows, r0 + p.Mc);
  if (r0 >= rend) return;  // the whole super-cluster leaves: nobody waits for it (every 'return' below is an abort)
  unsigned* flags = p.sync + kSyncFlags + sc * 32;
  const int u0 = me * kHU;
  const int uq = u0 + wave * 16 + q * 4;  // wave w finishes unit tile w of every row tile; a lane owns 4 units of one row
  const int B = p.B, T = p.T;
  const int64_t TB = (int64_t)T * B;

  const __amdgpu_buffer_rsrc_t x_rs = make_rsrc(p.xch);
  const __amdgpu_buffer_rsrc_t g_rs = make_rsrc(p.gates + (int64_t)lay * TB * G);
  const __amdgpu_buffer_rsrc_t c_rs = make_rsrc(p.cs + (int64_t)lay * TB * H);
  const __amdgpu_buffer_rsrc_t d_rs = make_rsrc(p.dg + (int64_t)lay * TB * G);
  // (T,B,H) f32 gradient of the top layer's outputs; NULL: a descriptor of zero records, every load of it returns zeros (no branch
  // around a load anywhere in the loop: behind a branch hipcc waits for the loaded registers at the join, i.e. right after the issue)
  const __amdgpu_buffer_rsrc_t e_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.d_hs_top), 0, p.d_hs_top ? 0x7fffffff : 0, 0x00020000);
  const float* d_hn = p.d_hn ? p.d_hn + lay * H : nullptr;
  float* dgsum = top ? nullptr : p.dgsum;

  // step-invariant per-lane byte offsets: saved gates / c (and the external gradient: same shape as c) of this lane's rows
  int vg[RT], vc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int rw = r0 + rt * 16 + r;
    const int rc = rw < rend ? rw : rend - 1;
    vg[rt] = (rc * G + (UM ? uq * 4 : cl_goff(uq))) * 2;
    vc[rt] = (rc * H + uq) * 4;
  }
  const int lane8 = lane * 8;
  // the row-major dg copy: chunk (lane & 31) of image row 2 * (wave * RT * 2 + i) + (lane >> 5)
  const int cc = lane & 31;
  const int vd = ((lane >> 5) * G + (cc >> 3) * H + u0 + (cc & 7) * 8) * 2;

  f32x4 dcreg[RT], ccur[RT], dgs[DR][4], own[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) dcreg[rt] = own[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int rt = 0; rt < DR; ++rt)
#pragma unroll
    for (int g = 0; g < 4; ++g) dgs[rt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  // LDS addressing as (base register) + (immediate), see lstm_bwd_rs.hip
  int ibase[4], wbase[2];
#pragma unroll
  for (int k3 = 0; k3 < 4; ++k3) ibase[k3] = r * 512 + (((k3 * 4 + q) ^ r) << 4);
#pragma unroll
  for (int g1 = 0; g1 < 2; ++g1) wbase[g1] = r * 512 + (((g1 * 8 + 2 * wave + (q >> 1)) ^ r) << 4) + (q & 1) * 8;
  const int wlbase = wave * (kKS * kNU * 1024) + lane * 16;
  unsigned long long* tl = (p.tlog && sc == 0 && me == 0) ? p.tlog + (top ? 256 : 0) : nullptr;  // (tools/prof_rs.py)
  CL_TLOG(6);

  // epilogue operands (saved gates, c_{t-1}, the external gradient: HBM) of the NEXT step, requested right behind the recurrent
  // publish (not earlier: with four row tiles per wave a second set of them beside the live one does not fit the registers)
  uint2 gk[RT][4];
  f32x4 cprev[RT], ext[RT];
  auto load_epi = [&](int sn) {
    const int t = T - 1 - sn;
    const int sg = t * B * G * 2, sh = t * B * H * 4;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      // unit-major (lstm_fwd_wr.hip): [unit][i,f,g,o] -> this lane's 4 units are 32 contiguous bytes; else the cl_goff layout
      const uint4 a = bld16<kNt>(g_rs, vg[rt], sg), b = bld16<kNt>(g_rs, vg[rt] + (UM ? 16 : 64), sg);
      gk[rt][0] = uint2{a.x, a.y}, gk[rt][1] = uint2{a.z, a.w}, gk[rt][2] = uint2{b.x, b.y}, gk[rt][3] = uint2{b.z, b.w};
      cprev[rt] = bldf4<kNt>(c_rs, vc[rt], t > 0 ? sh - B * H * 4 : 0);  // t = 0: any valid address, zeroed where it is used
      if constexpr (TOP) ext[rt] = bldf4<kNt>(e_rs, vc[rt], sh);
    }
  };
  __syncthreads();  // every thread has read the join word out of the image
  // W_hh^T fragments of this member's layer, stationary in registers: slot j multiplies towards destination member (me + j) & 3,
  // unit tile `wave` of that member; k-step ks covers gate ks >> 1, units [32 (ks & 1), +32) of this member's 64
  bf16x8 wreg[kKS][kNU];
  {
    const u16* whh = p.w_hh_t[lay];
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks)
#pragma unroll
      for (int j = 0; j < kNU; ++j) {
        const int unit = ((me + j) & 3) * kHU + wave * 16 + r;
        const int col = (ks >> 1) * H + u0 + (ks & 1) * 32 + q * 8;
        wreg[ks][j] = __builtin_bit_cast(bf16x8, *(const uint4*)(whh + (int64_t)unit * G + col));
      }
  }
  if constexpr (TOP) {  // W_ih^1 rows of the same K-slice for ALL 256 layer-0 units: slot j = destination member j, unit tile `wave`
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) {
      uint4 w4[kNU];
#pragma unroll
      for (int j = 0; j < kNU; ++j) {
        const int unit = j * kHU + wave * 16 + r;
        const int col = (ks >> 1) * H + u0 + (ks & 1) * 32 + q * 8;
        w4[j] = *(const uint4*)(p.w_ih_t[1] + (int64_t)unit * G + col);
      }
#pragma unroll
      for (int j = 0; j < kNU; ++j) *(uint4*)(Wl + wlbase + (ks * kNU + j) * 1024) = w4[j];
    }
  }
  load_epi(0);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) ccur[rt] = bldf4<0>(c_rs, vc[rt], (T - 1) * B * H * 4);
  if (d_hn) {  // the gradient of this layer's final state joins the first step's dh
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int rw = r0 + rt * 16 + r;
      const f32x4 e = *(const f32x4*)(d_hn + (int64_t)(rw < rend ? rw : rend - 1) * p.hn_ld + uq);
      if constexpr (TOP) ext[rt] += e;
      else own[rt] = e;
    }
  }
  CL_TLOG(7);
  for (int s = 0; s < T; ++s) {
    CL_TLOG(s * 8 + 0);
    const int t = T - 1 - s;
    f32x4 dh[RT];
    // ---- wait: the recurrent partials of dh_t (own layer, epoch s); layer 1: the from-above slot it will overwrite at the end of
    // this step has been read (layer-0 step s - kNS done); layer 0: layer 1's from-above partials of time t (its step s)
    {
      unsigned target = 0;
      if (top) {
        if (lane < 4 && s > 0) target = ep0 + (unsigned)s;
        if (lane >= 4 && lane < 8 && s >= kNS) target = ep0 + (unsigned)(s - kNS + 1);
      } else {
        if (lane >= 4 && lane < 8 && s > 0) target = ep0 + (unsigned)s;
        if (lane >= 8 && lane < 12) target = ep0 + (unsigned)(s + 1);
      }
      if (s > 0 || !top) {
        if (!pair_wait(p.sync, flags, target)) return;
      }
    }
    CL_TLOG(s * 8 + 2);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) dh[rt] = TOP ? ext[rt] + own[rt] : own[rt];
    if (s > 0) {
      uint2 pin[RT][3];
#pragma unroll
      for (int jj = 1; jj < kNU; ++jj) {
        const int so = rec_off((s - 1) & 1, sc, lay, (me - jj) & 3, jj, wave, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(x_rs, lane8 + rt * 512, so, kSc1);
          pin[rt][jj - 1] = uint2{v.x, v.y};
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < 3; ++j) dh[rt] += unpack4(pin[rt][j]);
    }
    if constexpr (!TOP) {
      uint2 ain[RT][kNU];
      const int slot = s % kNS;
#pragma unroll
      for (int src = 0; src < kNU; ++src) {
        const int so = ab_off(slot, sc, src, me, wave, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(x_rs, lane8 + rt * 512, so, kSc1);
          ain[rt][src] = uint2{v.x, v.y};
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < kNU; ++j) dh[rt] += unpack4(ain[rt][j]);
    }
    // ---- elementwise LSTM backward -> dg_t
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x4 ig, fg, gg, og;
      if constexpr (UM) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ig[i] = bf2f((u16)(gk[rt][i].x & 0xffff)), fg[i] = bf2f((u16)(gk[rt][i].x >> 16));
          gg[i] = bf2f((u16)(gk[rt][i].y & 0xffff)), og[i] = bf2f((u16)(gk[rt][i].y >> 16));
        }
      } else {
        ig = unpack4(gk[rt][0]), fg = unpack4(gk[rt][1]), gg = unpack4(gk[rt][2]), og = unpack4(gk[rt][3]);
      }
      f32x4 dp[4];
      const f32x4 cp = t > 0 ? cprev[rt] : f32x4{0.f, 0.f, 0.f, 0.f};  // c_{-1} = 0
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float tc = tanhf_(ccur[rt][i]);
        float dc = dh[rt][i] * og[i] * (1.f - tc * tc);
        if (s > 0) dc += dcreg[rt][i];
        const float d_o = dh[rt][i] * tc;
        const float d_i = dc * gg[i], d_f = dc * cp[i], d_g = dc * ig[i];
        dcreg[rt][i] = dc * fg[i];
        dp[0][i] = d_i * ig[i] * (1.f - ig[i]);
        dp[1][i] = d_f * fg[i] * (1.f - fg[i]);
        dp[2][i] = d_g * (1.f - gg[i] * gg[i]);
        dp[3][i] = d_o * og[i] * (1.f - og[i]);
      }
      ccur[rt] = cp;
      const bool live = r0 + rt * 16 + r < rend;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (live) dgs[TOP ? 0 : rt][g] += dp[g];
        *(uint2*)(Img + wbase[g & 1] + (g >> 1) * 256 + rt * 8192) = pack4(dp[g]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the image is complete
    CL_TLOG(s * 8 + 3);
    // the products run over TWO row tiles at a time (8 accumulators, 2 x 2 image fragments in flight): with all four at once the
    // MFMA operands alone (weights 128 + accumulators 64 + fragments 64 registers) left hipcc spilling stationary weights to scratch
    constexpr int RH = TOP ? 2 : RT;  // (measured in spills: layer 1 31 against 55, layer 0 0 against 24)
    f32x4 acc[RH][kNU];
    bf16x8 bfrag[2][RH];
    auto ifrag = [&](int ks, int buf, int rbase) {
#pragma unroll
      for (int rt = 0; rt < RH; ++rt)
        bfrag[buf][rt] = __builtin_bit_cast(bf16x8, *(const uint4*)(Img + ibase[ks & 3] + (ks >> 2) * 256 + (rbase + rt) * 8192));
    };
    // ---- this member's K-slice of dh_{t-1} for all 256 units of its layer: 8 RT fragment reads, 32 RT MFMAs per wave
    if (s + 1 < T) {
#pragma unroll
      for (int rb = 0; rb < RT; rb += RH) {
#pragma unroll
        for (int rt = 0; rt < RH; ++rt)
#pragma unroll
          for (int j = 0; j < kNU; ++j) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        ifrag(0, 0, rb);
        __builtin_amdgcn_sched_group_barrier(0x100, RH, 0);
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
          if (ks + 1 < kKS) ifrag(ks + 1, (ks + 1) & 1, rb);  // the next k-step's fragments fly under this one's MFMAs
#pragma unroll
          for (int j = 0; j < kNU; ++j)
#pragma unroll
            for (int rt = 0; rt < RH; ++rt) acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks][j], bfrag[ks & 1][rt], acc[rt][j], 0, 0, 0);
          if (ks + 1 < kKS) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, RH, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, RH * kNU - 1, 0);
          } else {
            __builtin_amdgcn_sched_group_barrier(0x008, RH * kNU, 0);
          }
        }
#pragma unroll
        for (int j = 1; j < kNU; ++j) {
          const int so = rec_off(s & 1, sc, lay, me, j, wave, 0);
#pragma unroll
          for (int rt = 0; rt < RH; ++rt) {  // what the members wait for
            const uint2 v = pack4(acc[rt][j]);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{v.x, v.y}, x_rs, lane8 + (rb + rt) * 512, so, 0);
          }
        }
#pragma unroll
        for (int rt = 0; rt < RH; ++rt) own[rb + rt] = acc[rt][0];
        __builtin_amdgcn_sched_barrier(0);  // one half after the other (interleaved by the scheduler they need both sets of registers)
      }
    }
    // ---- the step's bulk traffic, issued BEHIND the partial stores and allowed to stay in flight across the publish: a wave's
    // memory operations complete in order, so `vmcnt(kBulk)` below = "everything up to and including the partial stores has
    // landed", while these kBulk younger instructions -- the next step's epilogue operands from HBM and the row-major dg_t copy
    // to HBM, 80-96 KB per CU and step, 22 MB over the chip -- spread over the publish, the from-above product and the next flag
    // wait instead of standing in front of the flag (the first form of this kernel published with vmcnt(0) behind them: 3-6 us
    // per step at the HBM's rate).  NOTHING else may issue a vector memory instruction between the partial stores and the wait
    // (no phase clock there; the dg copy masks rows by an out-of-range offset instead of a branch, so the count is exact).
    constexpr int kBulk = (TOP ? 4 : 3) * RT + 2 * RT;
    load_epi(s + 1 < T ? s + 1 : s);  // (unconditional: behind a branch the loaded registers are waited for at the join)
    // the row-major copy of dg_t (what the weight-gradient contractions read): whole 128-byte lines out of the image
#pragma unroll
    for (int i0 = 0; i0 < RT * 2; i0 += 4) {
      uint4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *(const uint4*)(Img + kc_off<32>(2 * (wave * RT * 2 + i0 + i) + (lane >> 5), cc));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int irow = 2 * (wave * RT * 2 + i0 + i) + (lane >> 5);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{v[i].x, v[i].y, v[i].z, v[i].w}, d_rs, r0 + irow < rend ? vd : 0x7ffffff0,
                                               ((t * B + r0 + 2 * (wave * RT * 2 + i0 + i)) * G) * 2, kNt);
      }
    }
    // publish: the partial stores have reached the XCD's L2 (the bulk may still be in flight), every wave is past its image reads
    // (layer 0: the next step may overwrite the image; layer 1 reads it once more for the from-above product below), one flag store
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kBulk) : "memory");
    CL_TLOG(s * 8 + 4);
    if (s + 1 < T && tid == 0) __hip_atomic_store(flags + (TOP ? me : 4 + me), ep0 + (unsigned)(s + 1), RLX_AGENT);
    CL_TLOG(s * 8 + 5);
    if constexpr (TOP) {
      // ---- from-above partials for layer 0 at time t: the same image times this member's rows of W_ih^1 (LDS), behind the
      // recurrent publish -- the other layer-1 members' partials are in flight meanwhile
      bf16x8 wf[2][kNU];
      auto wfrag = [&](int ks, int buf) {
#pragma unroll
        for (int j = 0; j < kNU; ++j) wf[buf][j] = __builtin_bit_cast(bf16x8, *(const uint4*)(Wl + wlbase + (ks * kNU + j) * 1024));
      };
      const int slot = s % kNS;
#pragma unroll
      for (int rb = 0; rb < RT; rb += RH) {
#pragma unroll
        for (int rt = 0; rt < RH; ++rt)
#pragma unroll
          for (int j = 0; j < kNU; ++j) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        ifrag(0, 0, rb);
        wfrag(0, 0);
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
          if (ks + 1 < kKS) {
            ifrag(ks + 1, (ks + 1) & 1, rb);
            wfrag(ks + 1, (ks + 1) & 1);
          }
#pragma unroll
          for (int j = 0; j < kNU; ++j)
#pragma unroll
            for (int rt = 0; rt < RH; ++rt) acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][j], bfrag[ks & 1][rt], acc[rt][j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < kNU; ++j) {
          const int so = ab_off(slot, sc, me, j, wave, 0);
#pragma unroll
          for (int rt = 0; rt < RH; ++rt) {
            const uint2 v = pack4(acc[rt][j]);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{v.x, v.y}, x_rs, lane8 + (rb + rt) * 512, so, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // the barrier inside also keeps the next step's image writes behind this step's last fragment reads
      cluster_publish(flags, 8 + me, ep0 + (unsigned)(s + 1));
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 vs = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < DR; ++rt) {
      vs += dgs[rt][g];
      if (!TOP && dgsum && r0 + rt * 16 + r < rend) *(f32x4*)(dgsum + (int64_t)(r0 + rt * 16 + r) * G + g * H + uq) = dgs[rt][g];
    }
    if (p.db_ih[lay] || p.db_hh[lay]) db_reduce_add(vs, p.db_ih[lay], p.db_hh[lay], g * H + uq, lane);
  }
  CL_TLOG((T - 1) * 8 + 7);
}

template <int RT, bool UM>
__global__ __launch_bounds__(kThreads) void lstm_bwd_pair_rs_kernel(ClBwd p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int joined = cluster_join(p.sync, (int*)smem);
  if (joined < 0) return;
 const int info = __builtin_amdgcn_readfirstlane(joined & 255);  // XCD * 32 + slot (uniform: scalar offsets follow from it)
  const unsigned ep0 = (unsigned)__builtin_amdgcn_readfirstlane(joined >> 8) * kSeqEpochs;  // this launch's number on the sync block
  if (((info >> 2) & 1) == 0)                                    // members 0..3 of a super-cluster's eight: layer 1
    pair_member<RT, UM, true>(p, smem, info, ep0);
  else
    pair_member<RT, UM, false>(p, smem, info, ep0);
}

template <int RT>
int launch_pair(const ClBwd& p, hipStream_t st) {
  constexpr int SMEM = RT * 8192 + kWlBytes;
  static_assert(SMEM <= 163840, "LDS of a CU");
  static bool attr[2] = {false, false};
  const int um = p.gates_um ? 1 : 0;
  if (!attr[um]) {
    const hipError_t e = um ? hipFuncSetAttribute((const void*)lstm_bwd_pair_rs_kernel<RT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM)
                            : hipFuncSetAttribute((const void*)lstm_bwd_pair_rs_kernel<RT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return (int)e;
    attr[um] = true;
  }
  if (um)
    hipLaunchKernelGGL((lstm_bwd_pair_rs_kernel<RT, true>), dim3(kGrid), dim3(kThreads), SMEM, st, p);
  else
    hipLaunchKernelGGL((lstm_bwd_pair_rs_kernel<RT, false>), dim3(kGrid), dim3(kThreads), SMEM, st, p);
  return fh_launch_status();
}

}  // namespace

int cluster_bwd_pair_rs(const ClBwd& p, hipStream_t st) {
  if (p.NU != kNU || p.Mc > 64 || p.Mc % 16 != 0 || !p.w_ih_t[1] || !p.w_hh_t[0] || !p.w_hh_t[1]) return FHVAE_ERR_SHAPE;
  return p.Mc <= 32 ? launch_pair<2>(p, st) : launch_pair<4>(p, st);
}

}  // namespace fh
