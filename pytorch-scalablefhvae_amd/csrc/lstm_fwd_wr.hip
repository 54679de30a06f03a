// lstm_fwd_wr.hip -- K1 forward, two-layer H = 256 net, rows form, ONE persistent launch for all T + 1 wavefront steps, with the
// WEIGHTS REGISTER-STATIONARY (the lesson of lstm_bwd_rs.hip applied to the forward; the body the reference never wrote:
// fhvae.py:14, semantics of torch.nn.LSTM).
//
// lstm_fwd_cluster_kernel (rounds 1-2) keeps a member's 64 gate columns of W_hh^0, W_ih^1, W_hh^1 in LDS (96 KB) and lets each of
// its four waves multiply its own 32 batch rows: per k-step a wave re-reads 8 weight fragments for 16 MFMAs -- 900 KB of LDS reads
// per CU and step, 5.1 us of contraction for 1.7 us of MFMA issue -- and a wave alone on its SIMD exposes every latency.  Here:
//   * a cluster = 8 workgroups of ONE XCD (lstm_cluster_dev.h), 64 batch rows; member m owns hidden units [32m, 32m + 32) of both
//     layers; a workgroup = 512 threads = EIGHT waves, two per SIMD (256 registers each): wave w owns 4 units = one 16-column tile of
//     [4 units][i,f,g,o], for ALL 64 rows.  Its fragments of the three recurrent / inter-layer matrices (24) and of W_ih[0] (4) stay
//     in registers for the whole launch: the LDS holds activations only, and the SIMD's other wave issues MFMAs while this one
//     does gate math or waits for an LDS read;
//   * what the members exchange per step is h^0_t and h^1_{t-1} (bf16, the saved-for-backward tensors themselves), fetched by LDS-DMA
//     (L1-bypassing, whole 512-byte rows) into two LDS images; layer 0's input row [x_t | xc] (the time-constant input rides along,
//     so there is no per-row additive term) goes into a third.  Images are separate LDS objects, waits are counted;
//   * two chains with their own flags -- A: layer 0 (h^0_t needs h^0_{t-1} only), B: layer 1 -- so that a step's waits hide behind
//     the other chain's work (the step loop below); VMEM operations of a wave complete in order, so polls, image requests and stores
//     are dealt to different waves: waves 0-3 poll A and request the h^0 / x images, waves 4-7 poll B and request the h^1 image;
//   * MFMA roles swapped (weight fragment first) with the tile's 16 columns ordered [unit][gate]: a lane holds i,f,g,o of ONE unit of
//     ONE row per tile -- the gate math never leaves registers; its results are staged in LDS in the layout they leave in and
//     stored by the whole workgroup as 16-byte pieces of whole lines: h (bf16, what the other members wait for) at once, c (f32),
//     the activated gates (UNIT-MAJOR: [row][unit][i,f,g,o] bf16; ClFwd::gates_um / ClBwd::gates_um) and the f32 copies of h a
//     step later, when their acknowledgements can no longer hold up an image wait.
// Hand-off protocol, XCD placement, bounded spins: lstm_cluster_dev.h / lstm_cluster.hip.
#include <cstdlib>
#include <type_traits>

#include "lstm_cluster_dev.h"

#include "trace.h"

namespace fh {

constexpr int kFwH = 256, kFwG = 4 * kFwH, kFwHU = 32, kFwNU = 8, kFwThreads = 512;
typedef void __attribute__((address_space(3))) * fw_lds_p;

// LDS-DMA of an activation image, NP consecutive 1-KB pieces per requesting wave: buffer form, so that the per-lane part of the address
// is a 32-bit offset that does NOT change from step to step (one register per piece, computed once) while the step's slab is the
// scalar offset.  (With 64-bit per-lane pointers hipcc hoisted the step-invariant halves of 20 addresses out of the step loop,
// spilled them, and reloaded each behind an s_waitcnt vmcnt(0) between two DMA instructions.)
//   h image: [rows][32 chunks]: one piece = 2 rows x 512 B; the per-lane SOURCE chunk carries the XOR swizzle (chunk ^ (row & 15)) the
//   fragment reads undo (guide rule 21); rows past `rlast` are clamped (their results are never stored)
template <int NP>
__device__ __forceinline__ void fw_h_offsets(unsigned (&voff)[NP], int r0, int rlast, int piece0, int lane) {
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int rw = (piece0 + i) * 2 + (lane >> 5), rg = r0 + rw;
    voff[i] = (unsigned)(rg < rlast ? rg : rlast) * (kFwH * 2) + (unsigned)(((lane & 31) ^ (rw & 15)) << 4);
  }
}
// L1-bypassing (sc1): the rows were written by other CUs of this XCD
template <int NP>
__device__ __forceinline__ void fw_dma_h(char* img, __amdgpu_buffer_rsrc_t rs, const unsigned (&voff)[NP], unsigned slab_bytes, int piece0) {
#pragma unroll
  for (int i = 0; i < NP; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (fw_lds_p)(img + (piece0 + i) * 1024), 16, voff[i], slab_bytes, 0, kSc1);
}
//   layer 0's input image: row = [x_t (I) | xc (Ic) | padding] bf16 in 256 bytes (16 chunks), one piece = 4 rows.  The x_t chunks are
//   rewritten every step (lanes of other chunks are masked off: an LDS-DMA lane writes its own 16 bytes); the time-constant xc chunks
//   and the padding (finite data against zero weight fragments) are written once (fw_dma_xc)
template <int NP>
__device__ __forceinline__ void fw_x_offsets(unsigned (&voff)[NP], int I, int r0, int rlast, int piece0, int lane) {
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int rw = (piece0 + i) * 4 + (lane >> 4), rg = r0 + rw;
    voff[i] = (unsigned)(rg < rlast ? rg : rlast) * (unsigned)(I * 2) + (unsigned)(((lane & 15) ^ (rw & 15)) << 4);
  }
}
template <int NP>
__device__ __forceinline__ void fw_dma_x(char* img, __amdgpu_buffer_rsrc_t rs, const unsigned (&voff)[NP], unsigned slab_bytes, int nchx, int piece0,
                                         int lane) {
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int rw = (piece0 + i) * 4 + (lane >> 4);
    if (((lane & 15) ^ (rw & 15)) < nchx)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (fw_lds_p)(img + (piece0 + i) * 1024), 16, voff[i], slab_bytes, 0, 0);
  }
}
template <int NP>
__device__ __forceinline__ void fw_dma_xc(char* img, const u16* xc, int Ic, const u16* pad, int nchx, int r0, int rlast, int piece0, int lane) {
  const int nchc = Ic >> 3;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int rw = (piece0 + i) * 4 + (lane >> 4), rg = r0 + rw;
    const int64_t rc = rg < rlast ? rg : rlast;
    const int c = (lane & 15) ^ (rw & 15);
    if (c >= nchx) {
      const u16* src = c < nchx + nchc ? xc + rc * Ic + (c - nchx) * 8 : pad;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (fw_lds_p)(img + (piece0 + i) * 1024), 16, 0, 0);
    }
  }
}

template <int RT, int PDT = 4>
__global__ __launch_bounds__(kFwThreads) void lstm_fwd_wr_kernel(ClFwd p) {
  constexpr int H = kFwH, G = kFwG;
  constexpr int ROWS = RT * 16;
  // separate LDS objects: hipcc then knows that the fragment reads of one image cannot alias the DMA in flight into another
  // (counted waits instead of vmcnt(0) in front of every read: guide, "three .s-level traps")
  __shared__ __attribute__((aligned(1024))) char img_h0[ROWS * 512];
  __shared__ __attribute__((aligned(1024))) char img_x[ROWS * 256];
  __shared__ __attribute__((aligned(1024))) char img_h1[ROWS * 512];
  // the cells' outputs, staged at gate-math time in the layout they leave in: per layer [gates: 256 B per row | c: 128 B | f32 h: 128 B]
  // for the member's 32 units, rows padded by 16 B (the 16 rows a wave writes at once then fall on different banks), + the bf16 h
  // of the layer being published (64 B per row)
  constexpr int kGS = 272, kCS = 144, kHS = 80;  // row strides
  constexpr int kStageL = ROWS * (kGS + 2 * kCS);
  __shared__ __attribute__((aligned(16))) char stage_all[2 * kStageL];
  __shared__ __attribute__((aligned(16))) char h16_all[ROWS * kHS];
  __shared__ int misc[2];  // [0]: the join word; [1]: a polling wave gave up -- every wave leaves behind the next barrier
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;

  const int joined = cluster_join(p.sync, misc);
  if (joined < 0) return;
  const int info = joined & 255;                                 // XCD * 32 + slot
  const unsigned ep0 = (unsigned)(joined >> 8) * kSeqEpochs;     // this launch's number on the sync block
  const int cluster = (info >> 5) * (32 / kFwNU) + (info & 31) / kFwNU, me = (info & 31) % kFwNU;
  const int r0 = p.row0 + cluster * p.Mc;
  const int rend = min(p.row0 + p.nrows, r0 + p.Mc);
  if (r0 >= rend) return;  // the whole cluster leaves: nobody waits for it
  unsigned* flags = p.sync + kSyncFlags + cluster * 32;
  unsigned* flagsB = flags + 8;
  const int B = p.B, T = p.T;
  const int um = me * kFwHU;       // first unit of this member
  const int ul = wave * 4 + q;     // this lane's unit within the member
  if (tid == 0) misc[1] = 0;
  const int wv = __builtin_amdgcn_readfirstlane(wave);  // (provably wave-uniform: scalar branches around the waits)
  __syncthreads();                                      // every thread has read the join word

  // ---- stationary weight fragments: A-operand row i = lane & 15 -> (unit i >> 2 of the wave's four, gate i & 3)
  const int wrow = (r & 3) * H + um + wave * 4 + (r >> 2);  // physical weight row of this lane's A row
  bf16x8 w_hh0[8], w_ih1[8], w_hh1[8], w_x[4];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const int64_t o = (int64_t)wrow * H + ks * 32 + q * 8;
    w_hh0[ks] = __builtin_bit_cast(bf16x8, *(const uint4*)(p.w_hh[0] + o));
    w_ih1[ks] = __builtin_bit_cast(bf16x8, *(const uint4*)(p.w_ih[1] + o));
    w_hh1[ks] = __builtin_bit_cast(bf16x8, *(const uint4*)(p.w_hh[1] + o));
  }
  const int nch0 = p.K0 / 8, nkx = (p.K0 + 31) / 32;  // 16-byte chunks / k-steps of layer 0's input [x_t | xc]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = j * 4 + q;
    uint4 v = uint4{0u, 0u, 0u, 0u};
    if (c < nch0) v = *(const uint4*)(p.w_ih0 + (int64_t)wrow * p.K0 + c * 8);
    w_x[j] = __builtin_bit_cast(bf16x8, v);
  }
  // additive terms = the two biases of each layer; a lane's tile cell = (row rt*16 + r, unit um + ul), its 4 accumulator registers
  // are the gates i, f, g, o
  f32x4 add0, add1;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    add0[g] = p.b_ih[0][g * H + um + ul] + p.b_hh[0][g * H + um + ul];
    add1[g] = p.b_ih[1][g * H + um + ul] + p.b_hh[1][g * H + um + ul];
  }
  // LDS addressing as (base) + (immediate): chunk (4 ks + q) ^ r shares four bases (ks & 3), see lstm_bwd_rs.hip
  int hbase[4], xbase[4];
#pragma unroll
  for (int k3 = 0; k3 < 4; ++k3) {
    hbase[k3] = r * 512 + (((k3 * 4 + q) ^ r) << 4);
    xbase[k3] = r * 256 + (((k3 * 4 + q) ^ r) << 4);
  }
  float creg[2][RT];
#pragma unroll
  for (int l = 0; l < 2; ++l)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) creg[l][rt] = 0.f;
  unsigned long long* tl = (p.tlog && cluster == 0 && me == 0) ? p.tlog : nullptr;
  const bool xvar = p.I > 0;  // the input image changes per step (else: the time-constant input only, fetched once)
  constexpr int NPX = RT * 4 / 4, NPH = RT * 8 / 4;  // pieces per requesting wave: x (RT*4 pieces, waves 0-3), h (RT*8 pieces, 4 waves)
  const int nchx = p.I >> 3;
  const __amdgpu_buffer_rsrc_t hs_rs = make_rsrc(p.hs);
  const __amdgpu_buffer_rsrc_t x_rs = make_rsrc(xvar ? p.x : p.hs);
  unsigned voff_h[NPH], voff_x[NPX];  // step-invariant per-lane offsets of this wave's pieces (waves 0-3: h^0 and x, waves 4-7: h^1)
  fw_h_offsets<NPH>(voff_h, r0, rend - 1, (wv & 3) * NPH, lane);
  fw_x_offsets<NPX>(voff_x, p.I, r0, rend - 1, (wv & 3) * NPX, lane);
  const unsigned slab_h = (unsigned)B * (kFwH * 2), slab_x = (unsigned)B * (unsigned)(p.I * 2);  // bytes per time step
  if (wv < 4) {
    fw_dma_xc<NPX>(img_x, p.xcv, p.Ic, xvar ? p.x : p.xcv, nchx, r0, rend - 1, wv * NPX, lane);
    if (xvar) fw_dma_x<NPX>(img_x, x_rs, voff_x, 0u, nchx, wv * NPX, lane);
  }

  // the saved-for-backward stores of layer l of step sp (c, the activated gates unit-major) and the f32 copies of h, out of the
  // staging area: one half of the workgroup (waves 0-3 or 4-7), 16 bytes per thread and piece, 16 / 8 consecutive lanes per row.
  // Round 4: the two layers go out at different points of the step and from different waves -- layer 0 (ready after P4) from
  // waves 4-7 during P7, where they have nothing else to do and nothing latency-critical queued behind the stores (a wave's vector
  // memory operations complete in order); layer 1 (ready after P8) from waves 0-3 in P2 of the next step as before.  All of it in
  // P2 meant 56 KB per CU, 14 MB over the chip, in one burst at the HBM's rate with waves 0-3 blocked on the issue while the
  // other four waited for them at the next barrier (P2 + P3 2.1 us of an 8.4-us step).
  auto tail_stores = [&](int sp, int l, int wbase) {
    if (wv < wbase || wv >= wbase + 4) return;
    const int tq = tid - wbase * 64;
    const int t = sp - l;
    if (t < 0 || t >= T) return;
    const int64_t lt = (int64_t)l * T + t;
    const char* st = stage_all + l * kStageL;
#pragma unroll
    for (int i = 0; i < ROWS * 16 / 256; ++i) {  // gates: 16 chunks per row
      const int c = i * 256 + tq, rw = c >> 4, part = c & 15;
      const uint4 v = *(const uint4*)(st + rw * kGS + part * 16);
      if (r0 + rw < rend) *(uint4*)(p.gates + (lt * B + r0 + rw) * G + um * 4 + part * 8) = v;
    }
    const bool top = l == 1 && p.hs_top_f32, last = p.hn && t == T - 1;
#pragma unroll
    for (int i = 0; i < ROWS * 8 / 256; ++i) {  // c (f32), f32 h: 8 chunks per row each
      const int c = i * 256 + tq, rw = c >> 3, part = c & 7;
      const uint4 v = *(const uint4*)(st + ROWS * kGS + rw * kCS + part * 16);
      if (r0 + rw < rend) *(uint4*)(p.cs + (lt * B + r0 + rw) * H + um + part * 4) = v;
      if (top || last) {
        const uint4 hv = *(const uint4*)(st + ROWS * (kGS + kCS) + rw * kCS + part * 16);
        if (r0 + rw < rend) {
          if (top) *(uint4*)(p.hs_top_f32 + ((int64_t)t * B + r0 + rw) * H + um + part * 4) = hv;
          if (last) {
            *(uint4*)(p.hn + (int64_t)(r0 + rw) * (2 * H) + l * H + um + part * 4) = hv;
            if (p.hn_lp) {  // the latent head's bf16 operand (fhvae_lstm_desc.hn_lp)
              const float4 hf = __builtin_bit_cast(float4, hv);
              *(uint2*)(p.hn_lp + (int64_t)(r0 + rw) * (2 * H) + l * H + um + part * 4) = pack4(f32x4{hf.x, hf.y, hf.z, hf.w});
            }
          }
        }
      }
    }
  };
  // this member's 32 units x ROWS rows of h (bf16) out of the h16 image: 4 chunks per row, then publish on `fl`
  auto h_out = [&](int l, int t, unsigned* fl, unsigned epoch, bool publish) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave's cells are in the image
    if (tid < ROWS * 4) {
      const int rw = tid >> 2, part = tid & 3;
      const uint4 v = *(const uint4*)(h16_all + rw * kHS + part * 16);
      if (r0 + rw < rend) *(uint4*)(p.hs + (((int64_t)l * T + t) * B + r0 + rw) * H + um + part * 8) = v;
    }
    if (publish) {  // (only the waves that stored h wait for their acknowledgements: what the other four have in flight -- bulk
      if (wv < 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stores, an image request, a flag read -- is nothing the flag promises)
      __syncthreads();
      if (tid == 0) __hip_atomic_store(fl + me, epoch, RLX_AGENT);
    } else {
      __syncthreads();
    }
  };
  // A poll is an L2 round trip (~1 us) even when every flag has long been raised, and the waves that polled arrived late at the
  // products behind it (P2 + P3 and P7 + P8 were each ~1 us longer than their work).  The flag line is now READ AHEAD -- the load
  // is issued ~0.5-2 us before its value is looked at, with products in between -- and the image request follows the look; only if
  // a flag is still missing then (a partner more than the look-ahead behind), the blocking poll runs.  Round 4, same-box A/B over
  // 200 calls: 212 -> 196 us per forward call with the look-ahead, -> 184 with the h^0 request moved behind P8's products.
  auto flags_peek = [&](const unsigned* fl) -> unsigned {
    unsigned v = 0xffffffffu;
    if (lane < kFwNU) v = __hip_atomic_load(fl + lane, RLX_AGENT);
    return v;
  };
  auto flags_ready = [&](unsigned peeked, const unsigned* fl, unsigned epoch) -> bool {
    if (__all(peeked >= epoch)) return true;
    return cluster_wait(p.sync, fl, kFwNU, epoch);
  };
  unsigned peekA = 0u, peekB = 0u;

  // ---- the step loop.  Per step s (layer 0 at t = s, layer 1 at t = s - 1); what a step waits for is requested while the OTHER
  // chain still has work:
  //   P1  waves 0-3: the h^0_{s-1} image and the input image have landed (requested in P8 of step s-1)
  //   P2  waves 4-7: read the flag line B ahead (published at the end of step s-1); waves 0-3: layer 1's saved-for-backward stores
  //       of step s-1 (not behind its publish: VMEM operations of a wave complete in order, so stores in front of an image request
  //       hold the image wait until their acknowledgements -- measured 0.6-1.0 us per step)
  //   P3  acc0 = [x_s | xc] . W_ih0 + h^0_{s-1} . W_hh0; then waves 4-7: look at B, request the h^1_{s-2} image (its flight: P4, P5)
  //   P4  layer 0's gate math, h^0_s out, publish A
  //   P5  acc1 = h^0_{s-1} . W_ih1; waves 0-3 read the flag line A ahead behind their last MFMA
  //   P6  barrier: every wave is done with the h^0 image; the h^1 image has landed
  //   P7  waves 4-7: layer 0's saved-for-backward stores
  //   P8  acc1 += h^1_{s-2} . W_hh1; waves 0-3: look at A, request the h^0_s image and the next input image; layer 1's gate math,
  //       h^1_{s-1} out, publish B
  constexpr int NI = 8 * RT;  // (k-step, row tile) items of one source
  for (int s = 0; s <= T; ++s) {
    CL_TLOG(s * 8 + 0);
    const bool act0 = s < T, act1 = s >= 1;
    f32x4 acc[2][RT];
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[l][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- P1
    if (wv < 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (misc[1]) return;
    CL_TLOG(s * 8 + 1);
    // ---- P2
    if (wv >= 4 && s >= 2) peekB = flags_peek(flagsB);  // (B: published at the end of the previous step; looked at behind P3's products)
    if (s > 0) tail_stores(s - 1, 1, 0);  // layer 1 of the previous step (layer 0 left in its P7)
    // ---- P3
    auto mm = [&](auto l_c, const char* img, const bf16x8 (&w)[8], bool peek_a = false) {  // acc[l] += image . w: the fragment of item
      constexpr int l = decltype(l_c)::value;                                              // i + 2 is requested before the MFMA of item i
      // Fragments in flight ahead of the MFMA that needs them, and the order PINNED (sched_barrier around every read and every
      // MFMA).  Written as a plain software pipeline of depth 2, hipcc -- at 237 of the 256 registers two waves per SIMD leave --
      // read every fragment into ONE register set and waited for it in front of its MFMA: a full LDS latency (~100 clk) per
      // 16-clk MFMA, 2 us for layer 0's 48 products where the LDS array needs 0.6.  Found only in the ISA: with the
      // saved-for-backward stores compiled out P2 + P3 kept its 2.0 us.  Same-box A/B (tools/exp/ab_fwd.py), depth 3 / 4 / 5 / the
      // unpinned form: 162.0 / 161.2 / 164.5 / 185.4 us per forward call.
      constexpr int PD = PDT;
      bf16x8 fb[PD + 1];
      auto frag = [&](int i) {
        const int ks = i / RT, rt = i % RT;
        fb[i % (PD + 1)] = __builtin_bit_cast(bf16x8, *(const uint4*)(img + hbase[ks & 3] + (ks >> 2) * 256 + rt * 8192));
      };
#pragma unroll
      for (int i = 0; i < PD; ++i) frag(i);
      if (PD > 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int ks = i / RT, rt = i % RT;
        if (i + PD < NI) frag(i + PD);
        // pin the order (one fragment read, one MFMA): left to itself hipcc reads every fragment into ONE register set and waits
        // for it in front of its MFMA -- a full LDS latency per MFMA
        if (PD > 2) __builtin_amdgcn_sched_barrier(0);
        acc[l][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ks], fb[i % (PD + 1)], acc[l][rt], 0, 0, 0);
        if (PD > 2) __builtin_amdgcn_sched_barrier(0);
        if (i == NI - 1 && peek_a) peekA = flags_peek(flags);  // (A: published a P5 ago; looked at behind P8's products)
      }
    };
    if (act0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j >= nkx) break;
        bf16x8 b[RT];  // (all reads of the k-step in flight before its first MFMA)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) b[rt] = __builtin_bit_cast(bf16x8, *(const uint4*)(img_x + xbase[j] + rt * 4096));
        __builtin_amdgcn_sched_barrier(0);  // (pinned like mm below: hipcc serialised the later k-steps read by read)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[0][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_x[j], b[rt], acc[0][rt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (s > 0) mm(std::integral_constant<int, 0>{}, img_h0, w_hh0);
    }
    if (wv >= 4 && s >= 2) {  // the h^1_{s-2} image (consumed in P8; every wave left P8 of the previous step long ago)
      if (!flags_ready(peekB, flagsB, ep0 + (unsigned)(s - 1))) {
        if (lane == 0) misc[1] = 1;
      } else {
        fw_dma_h<NPH>(img_h1, hs_rs, voff_h, (unsigned)(T + s - 2) * slab_h, (wv - 4) * NPH);
      }
    }
    CL_TLOG(s * 8 + 2);
    // ---- P4
    auto cell = [&](auto l_c, int rt) {
      constexpr int l = decltype(l_c)::value;
      const f32x4 v = acc[l][rt] + (l == 0 ? add0 : add1);
      const float ig = sigmoidf_(v[0]), fg = sigmoidf_(v[1]), gg = tanhf_(v[2]), og = sigmoidf_(v[3]);
      const float c = __builtin_fmaf(fg, creg[l][rt], ig * gg);  // (explicit: see tanhf_)
      const float h = og * tanhf_(c);
      creg[l][rt] = c;
      char* st = stage_all + l * kStageL;
      const int row = rt * 16 + r;
      *(uint2*)(st + row * kGS + ul * 8) = pack4(f32x4{ig, fg, gg, og});
      *(float*)(st + ROWS * kGS + row * kCS + ul * 4) = c;
      *(float*)(st + ROWS * (kGS + kCS) + row * kCS + ul * 4) = h;
      *(u16*)(h16_all + row * kHS + ul * 2) = f2bf(h);
    };
    if (s > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave has read the staging area (P2)
    if (act0) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) cell(std::integral_constant<int, 0>{}, rt);
      CL_TLOG(s * 8 + 3);
      h_out(0, s, flags, ep0 + (unsigned)(s + 1), true);
    }
    CL_TLOG(s * 8 + 4);
    if (act1) {
      // ---- P5
      mm(std::integral_constant<int, 1>{}, img_h0, w_ih1, wv < 4 && act0);
      // ---- P6
      if (wv >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    CL_TLOG(s * 8 + 5);
    // ---- P7
    auto request_a = [&]() {
      if (!flags_ready(act1 ? peekA : 0u, flags, ep0 + (unsigned)(s + 1))) {
        if (lane == 0) misc[1] = 1;
      } else {
        fw_dma_h<NPH>(img_h0, hs_rs, voff_h, (unsigned)s * slab_h, wv * NPH);
        // (x_s was consumed in P3, every wave has passed a barrier since)
        if (xvar && s + 1 < T) fw_dma_x<NPX>(img_x, x_rs, voff_x, (unsigned)(s + 1) * slab_x, nchx, wv * NPX, lane);
      }
    };
    if (wv < 4 && act0 && !act1) request_a();  // (the first step has no P8)
    if (act0) tail_stores(s, 0, 4);  // waves 4-7: layer 0 of this step (its staging area is rewritten in P4 of the next step, behind a barrier)
    CL_TLOG(s * 8 + 6);
    // ---- P8
    if (act1) {
      if (s > 1) mm(std::integral_constant<int, 1>{}, img_h1, w_hh1);
      if (wv < 4 && act0) request_a();  // (the image's flight: layer 1's gate math and hand-over, P1 of the next step)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) cell(std::integral_constant<int, 1>{}, rt);
      h_out(1, s - 1, flagsB, ep0 + (unsigned)s, s < T);
    }
    CL_TLOG(s * 8 + 7);
  }
  __syncthreads();
  tail_stores(T, 1, 0);
}

template __global__ void lstm_fwd_wr_kernel<2>(ClFwd);
template __global__ void lstm_fwd_wr_kernel<4>(ClFwd);

int cluster_fwd_wr(const ClFwd& p, hipStream_t st) {
  if ((int64_t)2 * p.T * p.B * kFwH * 2 >= (1LL << 31) || (int64_t)p.T * p.B * p.I * 2 >= (1LL << 31)) return FHVAE_ERR_LIMIT;  // 32-bit buffer offsets
  if (p.NU != kFwNU || p.Mc > 64 || p.Mc % 16 != 0 || p.pre || p.K0 != p.I + p.Ic || p.K0 <= 0 || p.K0 > 128 || (p.I % 8) || (p.Ic % 8) ||
      (p.I > 0 && !p.x) || (p.Ic > 0 && !p.xcv))
    return FHVAE_ERR_SHAPE;
  if (p.Mc <= 32)
    hipLaunchKernelGGL((lstm_fwd_wr_kernel<2>), dim3(kGrid), dim3(kFwThreads), 0, st, p);
  else
    hipLaunchKernelGGL((lstm_fwd_wr_kernel<4>), dim3(kGrid), dim3(kFwThreads), 0, st, p);
  return fh_launch_status();
}

}  // namespace fh
