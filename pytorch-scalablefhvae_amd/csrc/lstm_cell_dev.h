// lstm_cell_dev.h -- device-side pieces shared by the large-tile step cells (lstm_cell.hip) and their persistent streaming form
// (lstm_stream.hip): LDS-DMA staging, the K loop over a ring of stages, 16-byte epilogue accessors.
#pragma once
#include "lstm_cell.h"

namespace fh {

constexpr int kCellThreads = 256;
constexpr int kCellBK = 64;  // bf16 elements per stage row (128 B); f32: CellOp<float>::BK = 32

// operand type of the cells: bf16 (v_mfma_f32_16x16x32_bf16: a 16-byte chunk = a lane's 8 k of one MFMA) or f32
// (v_mfma_f32_16x16x4_f32, exact: a 16-byte chunk = 4 k, one per MFMA, the k order permuted identically for both operands)
template <typename T>
struct CellOp;
template <>
struct CellOp<u16> {
  using Frag = bf16x8;
  static constexpr int EPC = 8, BK = 64;
  static __device__ __forceinline__ void mma(f32x4& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
};
template <>
struct CellOp<float> {
  using Frag = f32x4;
  static constexpr int EPC = 4, BK = 32;
  static __device__ __forceinline__ void mma(f32x4& acc, const Frag& a, const Frag& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
  }
};
constexpr unsigned kCellOob = 0x40000000u;  // beyond every descriptor's num_records: the load returns zeros

typedef void __attribute__((address_space(3))) * cell_lds_p;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t cell_rsrc(const void* p, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? (int)bytes : 0, 0x00020000);
}

// s_waitcnt vmcnt(N) as the builtin (gfx9 encoding: vmcnt in bits [3:0] and [15:14], expcnt [6:4] and lgkmcnt [11:8] left at
// "no wait").  Unlike an asm statement the backend's waitcnt pass sees it and keeps its scoreboard exact across it -- with asm
// waits it assumed older global stores might still be pending next to the LDS-DMA loads ("mixed events": out-of-order return)
// and put a vmcnt(0) in front of the first fragment read of a K loop that follows an epilogue (lstm_stream.hip)
template <int N>
__device__ __forceinline__ void cell_wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 0xf) | ((N >> 4) << 14) | (0x7 << 4) | (0xf << 8));
  asm volatile("" ::: "memory");
}

// one stage of one operand: NI wave-instructions of 1 KiB (8 image rows x 128 B) per wave.  AUX = 16 (sc1): L1-bypassing
// loads, for operands other workgroups of the same launch have written (lstm_stream.hip)
template <int NI, int AUX = 0>
__device__ __forceinline__ void cell_issue(char* img, __amdgpu_buffer_rsrc_t rs, const unsigned (&voff)[NI], unsigned kbytes, int wave) {
#pragma unroll
  for (int q = 0; q < NI; ++q)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (cell_lds_p)(img + (wave * NI + q) * 1024), 16, voff[q] + kbytes, 0, 0, AUX);
}

template <typename F>
__device__ __forceinline__ F cell_frag(const char* img, int off) {
  typedef F __attribute__((address_space(3))) * lp;
  return *(lp)(img + off);
}

// The K loop shared by both cells.  RA / RB: image rows of the A (batch rows) and B (weight rows) operands; a wave owns
// TM x TN 16x16 tiles at A rows wm * RA/2 ..., B rows wn * RB/2 ....  `issue(stage, ks, part)` starts the DMA of k-step ks (zeros
// past the last one): part 0 = the A pieces, 1 = the B pieces, 2 = both.  Step s: wait for this wave's pieces of stage s, barrier (all pieces landed; everybody is done with
// stage s-1), refill stage s-1's buffer with step s+NS-1, multiply stage s.
template <typename T, int RA, int RB, int NS, typename Issue>
__device__ __forceinline__ void cell_mainloop(f32x4 (&acc)[RA / 32][RB / 32], int nsteps, Issue&& issue, char* s0, char* s1, char* s2, char* s3) {
  constexpr int TM = RA / 32, TN = RB / 32;
  constexpr int NLOAD = RA / 32 + RB / 32;
  constexpr int ABYTES = RA * 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 15, gq = lane >> 4;
  const int offa = (wm * (RA / 2) + i) * 128, offb = ABYTES + (wn * (RB / 2) + i) * 128;
  int cj[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) cj[j] = ((j * 4 + gq) ^ (i & 7)) << 4;
  char* bufs[4] = {s0, s1, s2, s3};
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(bufs[s], s, 2);
  // (Tried: the refill issued in two halves, A pieces behind the first 32-k block's fragment reads and B pieces behind the
  //  second's, as in wgrad.hip: the contraction of the forward cell went from 15 to 17 us per step.)
  auto step = [&](const char* cur, char* nxt, int ks_next) {
    cell_wait_vmcnt<(NS - 2) * NLOAD>();
    __builtin_amdgcn_s_barrier();
    issue(nxt, ks_next, 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      using Frag = typename CellOp<T>::Frag;
      Frag a[TM], b[TN];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[tn] = cell_frag<Frag>(cur, offb + tn * 2048 + cj[j]);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) a[tm] = cell_frag<Frag>(cur, offa + tm * 2048 + cj[j]);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) CellOp<T>::mma(acc[tm][tn], a[tm], b[tn]);
      __builtin_amdgcn_s_setprio(0);
    }
  };
  // branch-free body (a wait whose count depends on a branch becomes vmcnt(0)): NS steps per trip, the steps past nsteps
  // multiply the zeros of out-of-range loads
  for (int ks = 0; ks < nsteps; ks += NS) {
#pragma unroll
    for (int u = 0; u < NS; ++u) step(bufs[u], bufs[(u + NS - 1) % NS], ks + u + NS - 1);
  }
  cell_wait_vmcnt<0>();  // the look-ahead pieces (zeros) land before the LDS goes back
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef float __attribute__((address_space(3))) * cell_lds_f;
typedef f32x4v __attribute__((address_space(3))) * cell_lds_f4;

__device__ __forceinline__ unsigned pack_bf2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }
__device__ __forceinline__ void unpack_bf8(const u32x4v v, float (&o)[8]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[2 * k] = __builtin_bit_cast(float, v[k] << 16);
    o[2 * k + 1] = __builtin_bit_cast(float, v[k] & 0xffff0000u);
  }
}
__device__ __forceinline__ void ld8(const float* p, float (&o)[8]) {
  const f32x4v a = *(const f32x4v*)p, b = *(const f32x4v*)(p + 4);
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = a[k], o[4 + k] = b[k];
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
  *(f32x4v*)p = f32x4v{v[0], v[1], v[2], v[3]};
  *(f32x4v*)(p + 4) = f32x4v{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void st8_bf(u16* p, const float (&v)[8]) {
  *(u32x4v*)p = u32x4v{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
}

// 8 consecutive elements of the operand dtype <-> f32
__device__ __forceinline__ void ld8t(const u16* p, float (&o)[8]) { unpack_bf8(*(const u32x4v*)p, o); }
__device__ __forceinline__ void ld8t(const float* p, float (&o)[8]) { ld8(p, o); }
__device__ __forceinline__ void st8t(u16* p, const float (&v)[8]) { st8_bf(p, v); }
__device__ __forceinline__ void st8t(float* p, const float (&v)[8]) { st8(p, v); }

}  // namespace fh
