// lstm_cell.hip -- large-tile bf16 LSTM step cells for the shapes the persistent kernels do not take (H = 512: configs[3]).
//
// The generic step kernels (lstm.hip on gemm_core.h) tile a wavefront step into 64x64 (forward) / 32x32 (backward) outputs so
// that a launch has thousands of workgroups; at B = 2048, H = 512 that costs 0.4-1.0 GB of L2 -> LDS operand traffic per launch
// (every panel of h / dgates re-read by 32 column tiles) and the launch runs at the L2's bandwidth: 36 us forward, 64 us
// backward for 12.9 GFLOP.  Here a 256-thread workgroup owns a 128-row tile (128 virtual gate columns forward = 32 units x 4
// gates; 64 units backward), 2 x 2 waves of 64 x 64 / 64 x 32, v_mfma_f32_16x16x32_bf16; the operands (both K-contiguous: h and
// W rows forward, dgates and the transposed W copy backward) go global -> LDS by LDS-DMA into a ring of NS 64-k stages that are
// separate __shared__ objects (so hipcc emits counted vmcnt waits, see wgrad.hip), one barrier per stage; images are
// [row][128 B] with the 16-byte chunk index XOR-ed by (row & 7) on the DMA source side and on the ds_read_b128 side
// (conflict-free, guide T2).  Semantics of the cells: lstm.hip (torch.nn.LSTM gate order i,f,g,o; stands where the FC layers of
// simple_fhvae.py:160-164, :186-190, :240-244 stand).
#include "lstm_cell.h"

#include <cstdlib>

namespace fh {

constexpr int kCellThreads = 256;
constexpr int kCellBK = 64;
constexpr unsigned kCellOob = 0x40000000u;  // beyond every descriptor's num_records: the load returns zeros

typedef void __attribute__((address_space(3))) * cell_lds_p;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t cell_rsrc(const void* p, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? (int)bytes : 0, 0x00020000);
}

// one stage of one operand: NI wave-instructions of 1 KiB (8 image rows x 128 B) per wave
template <int NI>
__device__ __forceinline__ void cell_issue(char* img, __amdgpu_buffer_rsrc_t rs, const unsigned (&voff)[NI], unsigned kbytes, int wave) {
#pragma unroll
  for (int q = 0; q < NI; ++q)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (cell_lds_p)(img + (wave * NI + q) * 1024), 16, voff[q] + kbytes, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 cell_frag(const char* img, int off) {
  typedef bf16x8 __attribute__((address_space(3))) * lp;
  return *(lp)(img + off);
}

// The K loop shared by both cells.  RA / RB: image rows of the A (batch rows) and B (weight rows) operands; a wave owns
// TM x TN 16x16 tiles at A rows wm * RA/2 ..., B rows wn * RB/2 ....  `issue(stage, ks)` starts the DMA of k-step ks (zeros
// past the last one).  Step s: wait for this wave's pieces of stage s, barrier (all pieces landed; everybody is done with
// stage s-1), refill stage s-1's buffer with step s+NS-1, multiply stage s.
template <int RA, int RB, int NS, typename Issue>
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
  for (int s = 0; s < NS - 1; ++s) issue(bufs[s], s);
  auto step = [&](const char* cur, char* nxt, int ks_next) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * NLOAD) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(nxt, ks_next);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16x8 a[TM], b[TN];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[tn] = cell_frag(cur, offb + tn * 2048 + cj[j]);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) a[tm] = cell_frag(cur, offa + tm * 2048 + cj[j]);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  };
  // branch-free body (a wait whose count depends on a branch becomes vmcnt(0)): NS steps per trip, the steps past nsteps
  // multiply the zeros of out-of-range loads
  for (int ks = 0; ks < nsteps; ks += NS) {
#pragma unroll
    for (int u = 0; u < NS; ++u) step(bufs[u], bufs[(u + NS - 1) % NS], ks + u + NS - 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the look-ahead pieces (zeros) land before the LDS goes back
}

// k-step -> (segment, byte offset inside the row)
struct CellSegs {
  __amdgpu_buffer_rsrc_t a[2], b[2];
  int n0, n;  // steps of segment 0, total
};

// ---------------------------------------------------------------------------------------------
// forward cell: 128 rows x 32 units (x 4 gates)
// ---------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(kCellThreads, NS <= 2 ? 2 : 1) void cell_fwd_kernel(FwdJobs<u16> jobs) {
  constexpr int BM = 128, RB = 128, UN = 32;
  constexpr int STAGE = (BM + RB) * 128;
  __shared__ __attribute__((aligned(1024))) char st0[STAGE];
  __shared__ __attribute__((aligned(1024))) char st1[STAGE];
  __shared__ __attribute__((aligned(1024))) char st2[NS > 2 ? STAGE : 16];
  __shared__ __attribute__((aligned(1024))) char st3[NS > 3 ? STAGE : 16];
  const FwdJob<u16>& J = jobs.job[blockIdx.z];
  const int H = jobs.H;
  const int m0 = blockIdx.x * BM, u0 = blockIdx.y * UN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 15, gq = lane >> 4;

  CellSegs sg;
  unsigned va[2][4], vb[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const Seg& S = J.seg[s];
    const bool on = S.K > 0;
    sg.a[s] = cell_rsrc(on ? (const u16*)S.A + (int64_t)m0 * S.lda : nullptr, (int64_t)BM * S.lda * 2);
    sg.b[s] = cell_rsrc(on ? S.B : nullptr, (int64_t)4 * H * S.ldb * 2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = (wave * 4 + q) * 8 + (lane >> 3);  // image row
      const int c = (lane & 7) ^ (row & 7);              // logical chunk that lands in physical chunk lane & 7
      va[s][q] = (unsigned)row * (unsigned)(S.lda * 2) + (unsigned)c * 16u;
      // image row j = wn' * 64 + g * 16 + i'  <->  weight row g * H + u0 + wn' * 16 + i'
      const int wrow = ((row >> 4) & 3) * H + u0 + (row >> 6) * 16 + (row & 15);
      vb[s][q] = (unsigned)wrow * (unsigned)(S.ldb * 2) + (unsigned)c * 16u;
    }
  }
  sg.n0 = J.seg[0].K / kCellBK;
  sg.n = sg.n0 + J.seg[1].K / kCellBK;

  f32x4 acc[4][4];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](char* stg, int ks) {
    const bool s1 = ks >= sg.n0;
    const unsigned kb = ks < sg.n ? (unsigned)((s1 ? ks - sg.n0 : ks) * (kCellBK * 2)) : kCellOob;
    const __amdgpu_buffer_rsrc_t ra = s1 ? sg.a[1] : sg.a[0], rb = s1 ? sg.b[1] : sg.b[0];
    unsigned xa[4], xb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) xa[q] = s1 ? va[1][q] : va[0][q], xb[q] = s1 ? vb[1][q] : vb[0][q];
    cell_issue<4>(stg, ra, xa, kb, wave);
    cell_issue<4>(stg + BM * 128, rb, xb, kb, wave);
  };
  cell_mainloop<BM, RB, NS>(acc, sg.n, issue, st0, st1, st2, st3);

  // element offsets as 32-bit unsigned from uniform bases (64-bit per-element address arithmetic tripled the epilogue's VALU work)
  const unsigned unit = u0 + wn * 16 + i;
  const unsigned uH = (unsigned)H;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  if (J.bias_a) {
#pragma unroll
    for (int g = 0; g < 4; ++g) bsum[g] = J.bias_a[g * uH + unit] + J.bias_b[g * uH + unit];
  }
  const bool has_pre = J.pre != nullptr, has_cp = J.c_prev != nullptr;
  const float* prep = has_pre ? J.pre : J.c_out;  // stand-ins keep the loads unconditional (masked below)
  const unsigned pld = has_pre ? (unsigned)J.pre_ld : 0u;
  const float* cprev = has_cp ? J.c_prev : J.c_out;
  const unsigned row0 = m0 + wm * 64 + gq * 4;
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    float pa[4][4], cp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned row = row0 + tm * 16 + r;
#pragma unroll
      for (int g = 0; g < 4; ++g) pa[r][g] = prep[row * pld + (has_pre ? g * uH + unit : 0u)];
      cp[r] = cprev[row * uH + unit];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned row = row0 + tm * 16 + r;
      float x[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) x[g] = acc[tm][g][r] + (has_pre ? pa[r][g] : 0.f) + bsum[g];
      const float ig = sigmoidf_(x[0]), fg = sigmoidf_(x[1]);
      const float gg = tanhf_(x[2]), og = sigmoidf_(x[3]);
      const float c = __builtin_fmaf(fg, has_cp ? cp[r] : 0.f, ig * gg);
      const float h = og * tanhf_(c);
      const unsigned o = row * uH + unit;
      J.c_out[o] = c;
      J.h_out[o] = f2bf(h);
      if (J.h_out_f32) J.h_out_f32[o] = h;
      u16* go = J.gates_out;
      const unsigned og0 = row * 4u * uH + unit;
      go[og0] = f2bf(ig);
      go[og0 + uH] = f2bf(fg);
      go[og0 + 2 * uH] = f2bf(gg);
      go[og0 + 3 * uH] = f2bf(og);
      if (J.hn_out) J.hn_out[row * (unsigned)J.hn_ld + unit] = h;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward cell: 128 rows x 64 units
// ---------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(kCellThreads, NS <= 3 ? 2 : 1) void cell_bwd_kernel(BwdJobs<u16> jobs) {
  constexpr int BM = 128, BN = 64;
  constexpr int STAGE = (BM + BN) * 128;
  __shared__ __attribute__((aligned(1024))) char st0[STAGE];
  __shared__ __attribute__((aligned(1024))) char st1[STAGE];
  __shared__ __attribute__((aligned(1024))) char st2[NS > 2 ? STAGE : 16];
  __shared__ __attribute__((aligned(1024))) char st3[NS > 3 ? STAGE : 16];
  const BwdJob<u16>& J = jobs.job[blockIdx.z];
  const int H = jobs.H;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 15, gq = lane >> 4;

  CellSegs sg;
  unsigned va[2][4], vb[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const Seg& S = J.seg[s];
    const bool on = S.K > 0;
    sg.a[s] = cell_rsrc(on ? (const u16*)S.A + (int64_t)m0 * S.lda : nullptr, (int64_t)BM * S.lda * 2);
    sg.b[s] = cell_rsrc(on ? (const u16*)S.B + (int64_t)n0 * S.ldb : nullptr, (int64_t)BN * S.ldb * 2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = (wave * 4 + q) * 8 + (lane >> 3);
      va[s][q] = (unsigned)row * (unsigned)(S.lda * 2) + (unsigned)((lane & 7) ^ (row & 7)) * 16u;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = (wave * 2 + q) * 8 + (lane >> 3);
      vb[s][q] = (unsigned)row * (unsigned)(S.ldb * 2) + (unsigned)((lane & 7) ^ (row & 7)) * 16u;
    }
  }
  sg.n0 = J.seg[0].K / kCellBK;
  sg.n = sg.n0 + J.seg[1].K / kCellBK;

  f32x4 acc[4][2];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](char* stg, int ks) {
    const bool s1 = ks >= sg.n0;
    const unsigned kb = ks < sg.n ? (unsigned)((s1 ? ks - sg.n0 : ks) * (kCellBK * 2)) : kCellOob;
    const __amdgpu_buffer_rsrc_t ra = s1 ? sg.a[1] : sg.a[0], rb = s1 ? sg.b[1] : sg.b[0];
    unsigned xa[4], xb[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) xa[q] = s1 ? va[1][q] : va[0][q];
#pragma unroll
    for (int q = 0; q < 2; ++q) xb[q] = s1 ? vb[1][q] : vb[0][q];
    cell_issue<4>(stg, ra, xa, kb, wave);
    cell_issue<2>(stg + BM * 128, rb, xb, kb, wave);
  };
  cell_mainloop<BM, BN, NS>(acc, sg.n, issue, st0, st1, st2, st3);

  // Branch-free epilogue: an absent optional input is read from a valid stand-in (c_cur) and masked by a uniform select --
  // with uniform branches per optional pointer the unrolled epilogue became a CFG of hundreds of blocks and spilled 270 VGPRs.
  // Element offsets are 32-bit unsigned from uniform bases.
  const bool has_cp = J.c_prev != nullptr, has_e1 = J.ext != nullptr, has_e2 = J.ext2 != nullptr, first = J.first != 0;
  const unsigned uH = (unsigned)H;
  const float* cprev = has_cp ? J.c_prev : J.c_cur;
  const float* e1p = has_e1 ? J.ext : J.c_cur;
  const unsigned e1ld = has_e1 ? (unsigned)J.ext_ld : uH;
  const float* e2p = has_e2 ? J.ext2 : J.c_cur;
  const unsigned e2ld = has_e2 ? (unsigned)J.ext2_ld : uH;
  const unsigned row0 = m0 + wm * 64 + gq * 4;
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const unsigned unit = n0 + wn * 32 + tn * 16 + i;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
      float ig[4], fg[4], gg[4], og[4], cp[4], cc[4], dcin[4], e1[4], e2[4], dp[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned row = row0 + tm * 16 + r;
        const unsigned o = row * uH + unit, o4 = row * 4u * uH + unit;
        ig[r] = bf2f(J.gates[o4]), fg[r] = bf2f(J.gates[o4 + uH]), gg[r] = bf2f(J.gates[o4 + 2 * uH]), og[r] = bf2f(J.gates[o4 + 3 * uH]);
        cp[r] = cprev[o];
        cc[r] = J.c_cur[o];
        dcin[r] = J.dc[o];
        e1[r] = e1p[row * e1ld + unit];
        e2[r] = e2p[row * e2ld + unit];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned row = row0 + tm * 16 + r;
        const unsigned o = row * uH + unit, o4 = row * 4u * uH + unit;
        // same order of operations as lstm_bwd_step_kernel (lstm.hip): ext, then ext2
        float dh = acc[tm][tn][r];
        dh += has_e1 ? e1[r] : 0.f;
        dh += has_e2 ? e2[r] : 0.f;
        const float cpv = has_cp ? cp[r] : 0.f;
        const float tc = tanhf_(cc[r]);
        float dc = dh * og[r] * (1.f - tc * tc);
        dc += first ? 0.f : dcin[r];
        const float d_o = dh * tc;
        const float d_i = dc * gg[r], d_f = dc * cpv, d_g = dc * ig[r];
        J.dc[o] = dc * fg[r];
        dp[r][0] = d_i * ig[r] * (1.f - ig[r]);
        dp[r][1] = d_f * fg[r] * (1.f - fg[r]);
        dp[r][2] = d_g * (1.f - gg[r] * gg[r]);
        dp[r][3] = d_o * og[r] * (1.f - og[r]);
#pragma unroll
        for (int g = 0; g < 4; ++g) J.dg_out[o4 + g * uH] = f2bf(dp[r][g]);
      }
      if (J.dgsum) {
        float old[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int g = 0; g < 4; ++g) old[r][g] = J.dgsum[(row0 + tm * 16 + r) * 4u * uH + g * uH + unit];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int g = 0; g < 4; ++g) J.dgsum[(row0 + tm * 16 + r) * 4u * uH + g * uH + unit] = (first ? 0.f : old[r][g]) + dp[r][g];
      }
    }
  }
}

template __global__ void cell_fwd_kernel<2>(FwdJobs<u16>);
template __global__ void cell_fwd_kernel<3>(FwdJobs<u16>);
template __global__ void cell_bwd_kernel<2>(BwdJobs<u16>);
template __global__ void cell_bwd_kernel<3>(BwdJobs<u16>);
template __global__ void cell_bwd_kernel<4>(BwdJobs<u16>);

static bool cell_seg_ok(const Seg& s, int64_t rows_a, int64_t rows_b) {
  if (s.K == 0) return true;
  if (!s.a_kc || !s.b_kc || s.a_rmod || (s.K % kCellBK) || (s.lda % 8) || (s.ldb % 8)) return false;
  if ((((uintptr_t)s.A) | ((uintptr_t)s.B)) & 15) return false;
  // 32-bit buffer offsets below kCellOob
  return rows_a * s.lda * 2 < (int64_t)kCellOob && rows_b * s.ldb * 2 < (int64_t)kCellOob;
}

bool cell_fwd_big_ok(const FwdJobs<u16>& jobs, int nj) {
  if (jobs.B % 128 || jobs.H % 32 || (int64_t)jobs.B * 4 * jobs.H * 4 >= (1LL << 31)) return false;
  for (int j = 0; j < nj; ++j) {
    const FwdJob<u16>& J = jobs.job[j];
    if ((J.pre && J.pre_ld * jobs.B * 4 >= (1LL << 31)) || (J.hn_out && J.hn_ld * jobs.B * 4 >= (1LL << 31))) return false;
  }
  for (int j = 0; j < nj; ++j)
    for (int s = 0; s < 2; ++s)
      if (!cell_seg_ok(jobs.job[j].seg[s], 128, 4 * (int64_t)jobs.H)) return false;
  return true;
}

bool cell_bwd_big_ok(const BwdJobs<u16>& jobs, int nj) {
  if (jobs.B % 128 || jobs.H % 64 || (int64_t)jobs.B * 4 * jobs.H * 4 >= (1LL << 31)) return false;
  for (int j = 0; j < nj; ++j) {
    const BwdJob<u16>& J = jobs.job[j];
    if ((J.ext && J.ext_ld * jobs.B * 4 >= (1LL << 31)) || (J.ext2 && J.ext2_ld * jobs.B * 4 >= (1LL << 31))) return false;
  }
  for (int j = 0; j < nj; ++j)
    for (int s = 0; s < 2; ++s)
      if (!cell_seg_ok(jobs.job[j].seg[s], 128, 64)) return false;
  return true;
}

int launch_cell_fwd_big(const FwdJobs<u16>& jobs, int nj, hipStream_t st) {
  static const int ns = getenv("FHVAE_CELL_FWD_NS") ? atoi(getenv("FHVAE_CELL_FWD_NS")) : 2;
  const dim3 grid((unsigned)(jobs.B / 128), (unsigned)(jobs.H / 32), (unsigned)nj), block(kCellThreads);
  if (ns == 3)
    hipLaunchKernelGGL((cell_fwd_kernel<3>), grid, block, 0, st, jobs);
  else
    hipLaunchKernelGGL((cell_fwd_kernel<2>), grid, block, 0, st, jobs);
  return fh_launch_status();
}

int launch_cell_bwd_big(const BwdJobs<u16>& jobs, int nj, hipStream_t st) {
  static const int ns = getenv("FHVAE_CELL_BWD_NS") ? atoi(getenv("FHVAE_CELL_BWD_NS")) : 3;
  const dim3 grid((unsigned)(jobs.B / 128), (unsigned)(jobs.H / 64), (unsigned)nj), block(kCellThreads);
  if (ns == 2)
    hipLaunchKernelGGL((cell_bwd_kernel<2>), grid, block, 0, st, jobs);
  else if (ns == 4)
    hipLaunchKernelGGL((cell_bwd_kernel<4>), grid, block, 0, st, jobs);
  else
    hipLaunchKernelGGL((cell_bwd_kernel<3>), grid, block, 0, st, jobs);
  return fh_launch_status();
}

}  // namespace fh
