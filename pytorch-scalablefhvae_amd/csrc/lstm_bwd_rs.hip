// lstm_bwd_rs.hip -- K1 backward, one layer per persistent launch, H = 256, rows form: the members of a cluster exchange
// PARTIAL dh (a reduce-scatter over the hidden units) instead of the whole gate-gradient matrix dg (an all-gather).
//
//   dh_{t-1}[row, u] = sum over the 4H gate columns c of dg_t[row, c] * W_hh[c, u]        (torch.nn.LSTM backward; the body the
//                                                                                         reference never wrote: fhvae.py:14)
// A cluster = 4 workgroups of ONE XCD (lstm_cluster_dev.h), 32 batch rows; member m owns hidden units [64m, 64m+64) and therefore
// PRODUCES dg_t for the 256 gate columns {g*H + 64m + j}.  lstm_bwd_layer_ks_kernel (round 2) made every member read the
// cluster's whole dg_t (64 KB per CU and step) before it could multiply.  Here a member multiplies the K-slice it already holds
// -- its own 256 gate columns, straight out of its epilogue -- by its rows of W_hh for ALL 256 units, and hands each other
// member the (32 rows x 64 units) partial of that member's units (bf16): 12 KB out and 12 KB in per CU and step, six 8-byte loads
// per lane instead of sixteen 16-byte ones, no K-split reduction through LDS, and the weights (32 fragments per wave) stay in registers:
//   wave w multiplies unit tile w (16 units) of every destination member: its own-destination tile is already in the epilogue's
//   lane layout (unit quad on lane>>4, row on lane&15), the three others leave as whole 512-byte fragments that the destination's
//   wave w adds to its accumulator as they are.
// Per step (t = T-1-s):  wait for the flags -> 6 partial loads -> dh = external + from-above + own + 3 partials -> elementwise
// LSTM backward -> dg_t (bf16) into a 16-KB LDS image [row][gate][64 units] -> barrier -> 16 B-fragment reads + 64 MFMAs per wave
// -> 6 partial stores -> publish.  The image also feeds the row-major dg copy the weight-gradient contractions read (whole
// 128-byte lines, stored behind the publish).
// A layer below the top: its from-above term dg^{l+1}_t . W_ih[l+1] does not depend on the recurrence; it is ONE projection GEMM
// over all T*B rows between the two launches of a net (proj.hip: every operand row read once, one tile per CU) whose f32 result
// (fhvae_lstm_bwd_desc.ws_below) the lower layer's launch reads as its external gradient, like the top layer reads d_hs_top.
// (Tried and dropped: both weight sets register-stationary in one launch, the term computed per step in the flag wait -- 256 +
// ~250 registers, hipcc spilled the stationary fragments inside the MFMA loop; the term as a prologue of the launch, wave w
// multiplying its unit tile over K = 4H from an LDS-DMA double buffer -- 70 us for 20 steps: each of a cluster's 4 members
// streams the same operand rows.)
// Hand-off protocol, XCD placement, bounded spins: lstm_cluster_dev.h / lstm_cluster.hip.
#include <cstdlib>

#include "lstm_cluster_dev.h"
#include "trace.h"

namespace fh {

namespace {

constexpr int kH = 256, kG = 4 * kH, kHU = 64, kNU = 4;
constexpr int kKS = 8;     // k-steps (32 gate columns) of a member's own 256 gate columns

// byte offset of the 512-byte fragment (parity, cluster, source member, slot j = (dst - src) & 3 in 1..3, wave, row tile): the
// partials travel in bf16 (8 bytes per lane: 4 units of one row) -- half the bytes through the L2 and to memory (VERDICT r02 #1:
// the f32 partials were 126 MB of the launch's 207 MB of HBM writes), one more rounding on three of the four terms of dh
template <int RT>
__device__ __forceinline__ int rs_xoff(int par, int cluster, int src, int j, int wave, int rt) {
  return ((((((par * 64 + cluster) * kNU + src) * 3 + (j - 1)) * 4 + wave) * RT + rt) << 9);
}

template <int RT>
struct RsCfg {
  static constexpr int SMEM = RT * 16 * 512;  // dg_t image: [RT*16 rows][32 chunks]
};

template <int RT, bool UM>
__global__ __launch_bounds__(kThreads) void lstm_bwd_layer_rs_kernel(ClBwd p) {
  constexpr int H = kH, G = kG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Img = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;

  const int joined = cluster_join(p.sync, (int*)Img);
  if (joined < 0) return;
  const int info = joined & 255;                                 // XCD * 32 + slot
  const unsigned ep0 = (unsigned)(joined >> 8) * kSeqEpochs;     // this launch's number on the sync block
  const int cluster = (info >> 5) * (32 / kNU) + (info & 31) / kNU, me = (info & 31) % kNU;
  const int r0 = p.row0 + cluster * p.Mc;
  const int rend = min(p.row0 + p.nrows, r0 + p.Mc);
  if (r0 >= rend) return;  // the whole cluster leaves: nobody waits for it
  unsigned* flags = p.sync + kSyncFlags + cluster * 32;
  const int u0 = me * kHU;
  const int uq = u0 + wave * 16 + q * 4;  // epilogue: wave w finishes unit tile w of every row tile; a lane owns 4 units of one row
  const int B = p.B, T = p.T;

  int row[RT];
  int64_t rowc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    row[rt] = r0 + rt * 16 + r;
    rowc[rt] = row[rt] < rend ? row[rt] : rend - 1;
  }
  f32x4 dcreg[RT], ccur[RT], dgs[RT][4], own[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    dcreg[rt] = ccur[rt] = own[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) dgs[rt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const __amdgpu_buffer_rsrc_t x_rs = make_rsrc(p.xch);
  const float* ext_src = p.d_hs_top;  // (T,B,H) f32 external gradient of this layer's h (a lower layer: its from-above term)
  // LDS addressing as (base register) + (immediate): the XOR swizzle of 16-byte chunk c of row rw is c ^ (rw & 15) = c ^ r, and
  // only c's low four bits take part, so the chunks c = 4 ks + q of all k-steps share FOUR bases (ks & 3); the rest -- (ks >> 2)
  // * 256, the row tile -- is an instruction offset.  (Written as one expression per access, hipcc hoisted 64 + 16 + 8 loop-
  // invariant addresses into registers and spilled.)
  int ibase[4], wbase[2];
#pragma unroll
  for (int k3 = 0; k3 < 4; ++k3) {
    ibase[k3] = r * 512 + (((k3 * 4 + q) ^ r) << 4);    // image fragment reads (kc_off<32>)
  }
#pragma unroll
  for (int g1 = 0; g1 < 2; ++g1)  // image writes: chunk g * 8 + 2 wave + (q >> 1), 8 bytes at (q & 1) * 8
    wbase[g1] = r * 512 + (((g1 * 8 + 2 * wave + (q >> 1)) ^ r) << 4) + (q & 1) * 8;
  unsigned long long* tl = (p.tlog && cluster == 0 && me == 0) ? p.tlog + (p.tlog_slot ? 256 : 0) : nullptr;  // (tools/prof_rs.py)
  CL_TLOG(6);

  // epilogue operands (saved gates, c_{t-1}, the external gradient: HBM), fetched one step ahead
  uint2 gkn[RT][4];
  f32x4 cprevn[RT], extn[RT], ccurn[RT];
  // the epilogue operands (gates, c, the external gradient) are read once and the row-major dg is written once: streaming hints,
  // so that they do not displace the partial-dh lines the members exchange from the XCD's L2 (ClBwd::nt)
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  const bool nt = p.nt != 0;
  auto ld16 = [&](const void* q) -> uint4 {
    if (nt) {
      const u32x4v v = __builtin_nontemporal_load((const u32x4v*)q);
      return uint4{v.x, v.y, v.z, v.w};
    }
    return *(const uint4*)q;
  };
  auto ldf4 = [&](const float* q) -> f32x4 { return nt ? __builtin_nontemporal_load((const f32x4*)q) : *(const f32x4*)q; };
  auto load_epi = [&](int sn) {
    const int t = T - 1 - sn;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if constexpr (UM) {  // unit-major (lstm_fwd_wr.hip): [unit][i,f,g,o] -> this lane's 4 units are 32 contiguous bytes
        const u16* gp = p.gates + ((int64_t)t * B + rowc[rt]) * G + uq * 4;
        const uint4 a = ld16(gp), b = ld16(gp + 8);
        gkn[rt][0] = uint2{a.x, a.y}, gkn[rt][1] = uint2{a.z, a.w}, gkn[rt][2] = uint2{b.x, b.y}, gkn[rt][3] = uint2{b.z, b.w};
      } else {
        cl_load_gates(p.gates + ((int64_t)t * B + rowc[rt]) * G, uq, gkn[rt]);
      }
      if (sn == 0) ccurn[rt] = *(const f32x4*)(p.cs + ((int64_t)t * B + rowc[rt]) * H + uq);
      cprevn[rt] = t > 0 ? ldf4(p.cs + ((int64_t)(t - 1) * B + rowc[rt]) * H + uq) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 e = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ext_src) e = ldf4(ext_src + ((int64_t)t * B + rowc[rt]) * H + uq);
      if (sn == 0 && p.d_hn) e += *(const f32x4*)(p.d_hn + rowc[rt] * p.hn_ld + uq);
      extn[rt] = e;
    }
  };
  __syncthreads();  // every thread has read the join word out of the image
  // W_hh^T fragments, stationary: slot j multiplies towards destination member (me + j) & 3 (j = 0: this member itself), unit
  // tile `wave` of that member; k-step ks covers gate ks >> 1, units [32 (ks & 1), +32) of this member's 64
  bf16x8 wreg[kKS][kNU];
#pragma unroll
  for (int ks = 0; ks < kKS; ++ks)
#pragma unroll
    for (int j = 0; j < kNU; ++j) {
      const int unit = ((me + j) & 3) * kHU + wave * 16 + r;
      const int col = (ks >> 1) * H + u0 + (ks & 1) * 32 + q * 8;
      wreg[ks][j] = __builtin_bit_cast(bf16x8, *(const uint4*)(p.w_hh_t[0] + (int64_t)unit * G + col));
    }
  load_epi(0);
  CL_TLOG(7);
  for (int s = 0; s < T; ++s) {
    CL_TLOG(s * 8 + 0);
    const int t = T - 1 - s;
    f32x4 dh[RT];
    CL_TLOG(s * 8 + 1);
    // ---- the other members' partials of dh_t (published as epoch s)
    uint2 pin[RT][3];
    if (s > 0) {
      if (!cluster_wait(p.sync, flags, kNU, ep0 + (unsigned)s)) return;
      CL_TLOG(s * 8 + 2);
#pragma unroll
      for (int jj = 1; jj < kNU; ++jj)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(x_rs, rs_xoff<RT>((s - 1) & 1, cluster, (me - jj) & 3, jj, wave, rt) + lane * 8, 0, kSc1);
          pin[rt][jj - 1] = uint2{v.x, v.y};
        }
    }
    uint2 gk[RT][4];
    f32x4 cprev[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) gk[rt][g] = gkn[rt][g];
      cprev[rt] = cprevn[rt];
      dh[rt] = extn[rt] + own[rt];
      if (s == 0) ccur[rt] = ccurn[rt];
    }
    if (s > 0) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < 3; ++j) dh[rt] += unpack4(pin[rt][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < T) load_epi(s + 1);  // behind the partial loads: lands under the epilogue and the MFMAs
    // ---- elementwise LSTM backward -> dg_t
    uint2 dpk[RT][4];
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
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float tc = tanhf_(ccur[rt][i]);
        float dc = dh[rt][i] * og[i] * (1.f - tc * tc);
        if (s > 0) dc += dcreg[rt][i];
        const float d_o = dh[rt][i] * tc;
        const float d_i = dc * gg[i], d_f = dc * cprev[rt][i], d_g = dc * ig[i];
        dcreg[rt][i] = dc * fg[i];
        dp[0][i] = d_i * ig[i] * (1.f - ig[i]);
        dp[1][i] = d_f * fg[i] * (1.f - fg[i]);
        dp[2][i] = d_g * (1.f - gg[i] * gg[i]);
        dp[3][i] = d_o * og[i] * (1.f - og[i]);
      }
      ccur[rt] = cprev[rt];
      const bool live = row[rt] < rend;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (live) dgs[rt][g] += dp[g];
        dpk[rt][g] = pack4(dp[g]);
        *(uint2*)(Img + wbase[g & 1] + (g >> 1) * 256 + rt * 8192) = dpk[rt][g];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the image is complete (no vmcnt: prefetches stay in flight)
    CL_TLOG(s * 8 + 3);
    // ---- this member's K-slice of dh_{t-1} for all 256 units: 16 fragment reads, 64 MFMAs per wave
    if (s + 1 < T) {
      f32x4 acc[RT][kNU];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < kNU; ++j) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x8 bfrag[2][RT];
      auto ifrag = [&](int ks, int buf) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          bfrag[buf][rt] = __builtin_bit_cast(bf16x8, *(const uint4*)(Img + ibase[ks & 3] + (ks >> 2) * 256 + rt * 8192));
      };
      ifrag(0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, RT, 0);
#pragma unroll
      for (int ks = 0; ks < kKS; ++ks) {
        if (ks + 1 < kKS) ifrag(ks + 1, (ks + 1) & 1);  // the next k-step's fragments fly under this one's MFMAs
#pragma unroll
        for (int j = 0; j < kNU; ++j)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks][j], bfrag[ks & 1][rt], acc[rt][j], 0, 0, 0);
        if (ks + 1 < kKS) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, RT, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, RT * kNU - 1, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, RT * kNU, 0);
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        own[rt] = acc[rt][0];
#pragma unroll
        for (int j = 1; j < kNU; ++j)
          *(uint2*)((char*)p.xch + rs_xoff<RT>(s & 1, cluster, me, j, wave, rt) + lane * 8) = pack4(acc[rt][j]);  // what the members wait for
      }
    }
    // the row-major copy of dg_t (what the weight-gradient contractions and the layer below read): whole 128-byte lines out of
    // the image; read before the publish barrier (the next epilogue rewrites the image), stored behind it
    // (named values, not an array: across the publish's asm / branches hipcc kept an array in scratch)
    auto img_chunk = [&](int i) -> uint4 {
      const int c = (wave * RT * 2 + i) * 64 + lane;  // chunk c of the image: row c >> 5, chunk c & 31
      return *(const uint4*)(Img + kc_off<32>(c >> 5, c & 31));
    };
    const uint4 v0 = img_chunk(0), v1 = img_chunk(1), v2 = RT > 1 ? img_chunk(2) : uint4{0u, 0u, 0u, 0u},
                v3 = RT > 1 ? img_chunk(3) : uint4{0u, 0u, 0u, 0u};
    CL_TLOG(s * 8 + 4);
    if (s + 1 < T) cluster_publish(flags, me, ep0 + (unsigned)(s + 1));
    else __syncthreads();
    CL_TLOG(s * 8 + 5);
    auto dg_store = [&](int i, const uint4& v) {
      const int c = (wave * RT * 2 + i) * 64 + lane;
      const int rw = r0 + (c >> 5), cc = c & 31;
      if (rw < rend) {
        u16* dst = p.dg + ((int64_t)t * B + rw) * G + (cc >> 3) * H + u0 + (cc & 7) * 8;
        if (nt)
          __builtin_nontemporal_store(u32x4v{v.x, v.y, v.z, v.w}, (u32x4v*)dst);
        else
          *(uint4*)dst = v;
      }
    };
    dg_store(0, v0);
    dg_store(1, v1);
    if constexpr (RT > 1) {
      dg_store(2, v2);
      dg_store(3, v3);
    }
  }
  if (p.d_xc_zero && me == 0 && wave == 0) {  // one wave per cluster clears its rows of d_xc for the contraction that follows
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      if (row[rt] < rend)
        for (int c = q; c < p.Ic; c += 4) p.d_xc_zero[(int64_t)row[rt] * p.Ic + c] = 0.f;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 vs = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      vs += dgs[rt][g];
      if (p.dgsum && row[rt] < rend) *(f32x4*)(p.dgsum + (int64_t)row[rt] * G + g * H + uq) = dgs[rt][g];
    }
    if (p.db_ih[0] || p.db_hh[0]) db_reduce_add(vs, p.db_ih[0], p.db_hh[0], g * H + uq, lane);
  }
  CL_TLOG((T - 1) * 8 + 7);
}

template <int RT>
int launch_rs(const ClBwd& p, hipStream_t st) {
  if (p.gates_um)
    hipLaunchKernelGGL((lstm_bwd_layer_rs_kernel<RT, true>), dim3(kGrid), dim3(kThreads), RsCfg<RT>::SMEM, st, p);
  else
    hipLaunchKernelGGL((lstm_bwd_layer_rs_kernel<RT, false>), dim3(kGrid), dim3(kThreads), RsCfg<RT>::SMEM, st, p);
  return fh_launch_status();
}

}  // namespace

int cluster_bwd_layer_rs(const ClBwd& p, hipStream_t st) {
  if (p.NU != kNU || p.Mc > 32 || p.Mc % 16 != 0) return FHVAE_ERR_SHAPE;
  return p.Mc <= 16 ? launch_rs<1>(p, st) : launch_rs<2>(p, st);
}

}  // namespace fh
