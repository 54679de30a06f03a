// lstm_cluster_dev.h -- device-side helpers shared by the persistent recurrence kernels (lstm_cluster.hip, lstm_bwd_rs.hip):
// XCD placement, the per-step flag hand-off, row / unit maps, the saved-gate layout, L1-bypassing loads, the backward
// descriptor.  See the header comment of lstm_cluster.hip for the protocol.
#pragma once
#include "lstm_cluster.h"

#include "gemm_core.h"

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

namespace fh {

constexpr unsigned kSpinLimit = 1u << 21;  // polls of >= one L2 round trip each: gives up after about a second
constexpr int kGrid = 256;                  // one workgroup per CU, 32 per XCD
constexpr int kSc1 = 16;                    // aux bits of an L1-bypassing load

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

// a launch gives up: the code goes into the block's status word (the other workgroups watch it and leave) and into the
// caller's sticky word (fhvae_lstm_desc.sticky_status), which no later forward clears
__device__ __forceinline__ void cluster_give_up(unsigned* sync, unsigned code) {
  __hip_atomic_fetch_or(sync + kSyncStatus, code, RLX_AGENT);
  const unsigned long long a = (unsigned long long)__hip_atomic_load(sync + kSyncSticky, RLX_AGENT) |
                               ((unsigned long long)__hip_atomic_load(sync + kSyncSticky + 1, RLX_AGENT) << 32);
  if (a) __hip_atomic_fetch_or((unsigned*)a, code, RLX_AGENT);
}

// Slot of this workgroup on its XCD and the number of this launch on the sync block: (launch << 8) | (x * 32 + slot), or -1
// (abort).  An XCD's counter hands out 32 tickets per launch (256 workgroups, 32 per XCD), so ticket / 32 IS the launch number:
// the host does not have to count launches, and the backward needs no re-arming of the block after the forward.  (If an XCD ever
// received a 33rd workgroup of one launch it would take a ticket of the next launch and wait for flags nobody raises: the bounded
// spin ends the launch with the status word set.)
__device__ __forceinline__ int cluster_join(unsigned* sync, int* s_word) {
  if (threadIdx.x == 0) {
    const unsigned x = xcc_id();
    int v = -1;
    if (x < 8) {
      const unsigned ticket = __hip_atomic_fetch_add(sync + kSyncXcdCnt + x, 1u, RLX_AGENT);
      v = (int)(((ticket >> 5) << 8) | (x * 32 + (ticket & 31u)));
    }
    if (v < 0) cluster_give_up(sync, 2u);
    s_word[0] = v;
  }
  __syncthreads();
  return s_word[0];
}

// every member of the cluster has published `epoch`; false = abort.  EVERY wave polls for itself (one 64-byte line per
// poll): the waves stage and multiply their own rows, so nothing has to re-converge here.
__device__ __forceinline__ bool cluster_wait(unsigned* sync, const unsigned* flags, int nu, unsigned epoch) {
  const int lane = threadIdx.x & 63;
  for (unsigned spins = 0;; ++spins) {
    unsigned v = epoch, st = 0;
    if (lane < nu) v = __hip_atomic_load(flags + lane, RLX_AGENT);
    if (lane == 63) st = __hip_atomic_load(sync + kSyncStatus, RLX_AGENT);
    if (__any(st != 0)) return false;
    if (__all(v >= epoch)) break;
    if (spins > kSpinLimit) {
      if (lane == 0) cluster_give_up(sync, 1u);
      return false;
    }
  }
  asm volatile("" ::: "memory");
  return true;
}

// all stores of this workgroup have reached the XCD's L2, then ONE lane raises the flag
__device__ __forceinline__ void cluster_publish(unsigned* flags, int me, unsigned epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flags + me, epoch, RLX_AGENT);
}

// row r of the cluster -> physical batch row, clamped to the launch's last row (loads only; stores are masked)
struct ClRowMap {
  int r0, rlast;
  __device__ __forceinline__ int64_t operator()(int r) const {
    const int x = r0 + r;
    return x < rlast ? x : rlast;
  }
};
// virtual gate column n (0..63: gate-major, 16 units) of this workgroup -> physical weight row
struct ClGateMap {
  int H, u0;
  __device__ __forceinline__ int64_t operator()(int n) const { return (int64_t)(n >> 4) * H + u0 + (n & 15); }
};
struct ClUnitMap {
  int u0;
  __device__ __forceinline__ int64_t operator()(int n) const { return u0 + n; }
};


// Layout of the saved activated gates in the persistent schedules (fhvae_lstm_desc.gates is a workspace: forward and backward of
// a net always take the same schedule; the per-step cells keep [row][gate][H]).  Within a row, the 16 units of block u >> 4
// occupy 64 elements: [gates 0,1 | gates 2,3][unit quad][gate of the pair][4 units].  A lane's 4 units x 4 gates are then two
// 16-byte pieces (they were four 8-byte pieces at the stride H), and the 16 units x 4 gates of a forward member are ONE 128-byte
// line per row: the forward's saved-for-backward stores (20 partial-line instructions per wave and step, ~130 ns each: the
// 2.6-us tail of every step) become 12 that fill whole lines.
__device__ __forceinline__ int cl_goff(int uq) { return (uq >> 4) * 64 + ((uq >> 2) & 3) * 8; }
__device__ __forceinline__ void cl_load_gates(const u16* row_base, int uq, uint2 (&g)[4]) {
  const uint4 a = *(const uint4*)(row_base + cl_goff(uq)), b = *(const uint4*)(row_base + cl_goff(uq) + 32);
  g[0] = uint2{a.x, a.y}, g[1] = uint2{a.z, a.w}, g[2] = uint2{b.x, b.y}, g[3] = uint2{b.z, b.w};
}
__device__ __forceinline__ void cl_store_gates(u16* row_base, int uq, const uint2 (&g)[4]) {
  *(uint4*)(row_base + cl_goff(uq)) = uint4{g[0].x, g[0].y, g[1].x, g[1].y};
  *(uint4*)(row_base + cl_goff(uq) + 32) = uint4{g[2].x, g[2].y, g[3].x, g[3].y};
}

// phase clock (100 MHz) of one workgroup into the log, when asked for
#define CL_TLOG(slot)                                                                                      \
  do {                                                                                                     \
    if (tl && tid == 0) tl[(slot)] = wall_clock64();                                                        \
  } while (0)

// L1-bypassing 16-byte loads of exchanged data into registers (the compiler tracks their vmcnt)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// (2 GB of records: for loads / stores whose offsets are always valid.  Do NOT mask a lane by an "out-of-range" offset against this
//  descriptor: the range check is offset < num_records, so 0x7ffffff0 is IN range -- a masked store needs the tensor's real size
//  as num_records and an offset >= it, e.g. 0xfffffff0; found the hard way in round 4, DESIGN 9.
//  Second trap, same place: a 128-bit buffer STORE with an SGPR soffset still reads its data registers for a cycle or two after
//  issue, and hipcc's hazard recogniser only pads the form without an SGPR offset -- a VALU write of the first data register right
//  behind the store (e.g. the next LDS address) reached memory instead of the data, now and then.  Put `s_nop 1` behind such a store.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ uint4 load_sc1(__amdgpu_buffer_rsrc_t rs, int64_t byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, kSc1);
  return uint4{v.x, v.y, v.z, v.w};
}


// column sums for the bias gradients: v holds this lane's partial sums for 4 consecutive units (rows = lane & 15 of its
// 16-row tiles); add up the 16 row lanes, then one lane per unit quad adds into both bias gradients
__device__ __forceinline__ void db_reduce_add(f32x4 v, float* db_a, float* db_b, int col, int lane) {
#pragma unroll
  for (int off = 1; off < 16; off <<= 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += __shfl_xor(v[i], off, 64);
  // every row lane now holds the four sums: row lane i (< 4) adds unit i, so a wave adds its 16 units with ONE atomic instruction
  // per array (16 active lanes, four 16-byte runs) instead of four instructions of 4 lanes (the launch's tail: ~130 k scalar
  // atomics into the same 8 KB from all 256 CUs)
  const int i = lane & 15;
  if (i < 4) {
    const float s = i == 0 ? v[0] : i == 1 ? v[1] : i == 2 ? v[2] : v[3];
    if (db_a) atomicAdd(db_a + col + i, s);
    if (db_b) atomicAdd(db_b + col + i, s);
  }
}

struct ClBwd {
  int B, T, NU, Mc;
  int row0, nrows;
  const u16* w_ih_t[2];  // [H,4H] bf16 (l = 1)
  const u16* w_hh_t[2];  // [H,4H]
  const u16* gates;      // (L,T,B,4H) saved activations
  const float* cs;       // (L,T,B,H)
  const float* d_hs_top; // (T,B,H) or NULL
  const float* d_hn;     // (B,L*H) or NULL
  int hn_ld;             // row stride of d_hn (the layer kernel gets the slot of its layer pre-offset)
  u16* dg;               // (L,T,B,4H) out
  float* dgsum;          // (B,4H) out: sum over t of layer 0's dg, or NULL
  float* db_ih[2];       // [4H] bias gradients (accumulated with atomics: += sum over t and rows of dg^l), may be NULL
  float* db_hh[2];
  u16* xch;  // exchange buffer (blocked copy of dg; contraction-split form)
  unsigned* sync;
  unsigned long long* tlog;
  // contraction-split per-layer kernel, layer below the top: the from-above term dg^{l+1}_t . W_ih[l+1] is computed by the launch
  // itself from the finished layer above (row-major dg, (T,B,4H)) instead of being handed in through d_hs_top
  const u16* dg_above;
  const u16* w_above_t;  // W_ih[l+1]^T, [H,4H] bf16
  int gates_um;          // the forward saved the gates unit-major (lstm_fwd_wr.hip): [row][unit][i,f,g,o] bf16
  int tlog_slot;         // lstm_bwd_rs.hip: which half of the phase-clock log this launch writes (tools/prof_rs.py)
  int nt;                // lstm_bwd_rs.hip: streaming (non-temporal) hints on the once-read operands and the once-written dg
  float* d_xc_zero;      // lstm_bwd_rs.hip, layer 0: (B,Ic) f32 to ZERO for the rows of this launch (the split-K contraction that
  int Ic;                // follows adds into it: the zeroing launch in front of it is gone), or NULL
};

__device__ __forceinline__ f32x4 unpack4(uint2 v) {
  return f32x4{bf2f((u16)(v.x & 0xffff)), bf2f((u16)(v.x >> 16)), bf2f((u16)(v.y & 0xffff)), bf2f((u16)(v.y >> 16))};
}
__device__ __forceinline__ uint2 pack4(const f32x4& v) {
  return uint2{(uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16), (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16)};
}

struct ClFwd {
  int B, T, NU, Mc;    // B: row count of the (T,B,.) buffers; NU workgroups per cluster; Mc rows per cluster
  int row0, nrows;     // rows handled by this launch
  const u16* w_ih[2];  // [4H,H] bf16 (l = 1)
  const u16* w_hh[2];
  const float* b_ih[2];
  const float* b_hh[2];
  const float* pre;  // layer-0 additive term incl. biases: (T,B,4H), or (B,4H) with pre_tstride = 0, or NULL (= biases)
  int64_t pre_tstride;
  // folded input projection (x != NULL): the kernels multiply x_t (T,B,I) bf16 by this member's rows of W_ih[0][:, :I]
  // themselves (fragments stationary in registers); `pre` then only holds the time-constant part
  const u16* x;
  const u16* w_ih0;  // [4H,K0] bf16
  int I, K0;
  u16* hs;     // (L,T,B,H)
  float* cs;   // (L,T,B,H)
  u16* gates;  // (L,T,B,4H)
  float* hs_top_f32;  // optional (T,B,H)
  float* hn;          // optional (B, L*H)
  u16* hn_lp;         // optional (B, L*H) bf16 copy of hn (lstm_fwd_wr.hip)
  u16* xch;  // exchange buffer (blocked copy of h; contraction-split form), see xch_off
  unsigned* sync;
  unsigned long long* tlog;  // optional phase clock log of cluster 0 / member 0 (tools/prof_cluster.py)
  int il;                    // rows form, L = 2: layer 0's gate math between the MFMAs of the h^1 panels
  // the time-constant input xc (B,Ic) bf16 is projected by the kernel itself, once, into layer 0's additive term
  // (W_ih[0][:, I:I+Ic] from w_ih0); `pre` is then NULL and no GEMM runs before the launch
  const u16* xcv;
  int Ic;
  int gates_um;  // lstm_fwd_wr.hip: the saved gates are unit-major ([row][unit][i,f,g,o]); the backward launches get the same flag
};

// lstm_fwd_wr.hip: forward of a two-layer H = 256 net, weights register-stationary (p.NU = 8, p.Mc <= 64, x / xc folded)
int cluster_fwd_wr(const ClFwd& p, hipStream_t st);

// lstm_bwd_rs.hip: backward of ONE layer, H = 256, partial-dh exchange (p.NU = 4, p.Mc <= 32); p.xch holds kRsXchElems bf16 elements
int cluster_bwd_layer_rs(const ClBwd& p, hipStream_t st);

}  // namespace fh
