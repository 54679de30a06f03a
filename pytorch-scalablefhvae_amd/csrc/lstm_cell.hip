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
#include "lstm_cell_dev.h"

#include <cstddef>
#include <cstdlib>

namespace fh {

// k-step -> (segment, byte offset inside the row)
struct CellSegs {
  __amdgpu_buffer_rsrc_t a[2], b[2];
  int n0, n;  // steps of segment 0, total
};

// ---------------------------------------------------------------------------------------------
// forward cell: 128 rows x 32 units (x 4 gates)
// ---------------------------------------------------------------------------------------------
template <typename T, int NS>
__global__ __launch_bounds__(kCellThreads, NS <= 2 ? 2 : 1) void cell_fwd_kernel(FwdJobs<T> jobs) {
  constexpr int BK = CellOp<T>::BK, EPC = CellOp<T>::EPC, ES = (int)sizeof(T);
  constexpr int BM = 128, RB = 128, UN = 32;
  constexpr int STAGE = (BM + RB) * 128;
  __shared__ __attribute__((aligned(1024))) char st0[STAGE];
  __shared__ __attribute__((aligned(1024))) char st1[STAGE];
  __shared__ __attribute__((aligned(1024))) char st2[NS > 2 ? STAGE : 16];
  __shared__ __attribute__((aligned(1024))) char st3[NS > 3 ? STAGE : 16];
  const FwdJob<T>& J = jobs.job[blockIdx.z];
  const int H = jobs.H;
  const int m0 = blockIdx.x * BM, u0 = blockIdx.y * UN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 15, gq = lane >> 4;

  // Up to four K segments: h^{l-1}_t . W_ih[l], h^l_{t-1} . W_hh[l], and for layer 0 x_t . W_ih[0][:, :I], xc . W_ih[0][:, I:]
  // (any K that is a multiple of 8: the 16-byte chunks past K are loaded from an out-of-range offset = zeros).  The segment
  // of a k-step is read from the kernel arguments by a dynamic (uniform) index: a select chain over four preloaded descriptors
  // became branches in the loop, and a branch there turns the counted vmcnt waits into vmcnt(0).
  static_assert(offsetof(FwdJob<T>, xseg) == offsetof(FwdJob<T>, seg) + 2 * sizeof(Seg), "seg[] and xseg[] form one array of 4");
  const Seg* segs = &J.seg[0];
  int end0, end1, end2, end3;  // first k-step after each segment
  end0 = (segs[0].K + BK - 1) / BK;
  end1 = end0 + (segs[1].K + BK - 1) / BK;
  end2 = end1 + (segs[2].K + BK - 1) / BK;
  end3 = end2 + (segs[3].K + BK - 1) / BK;
  const int nsteps = end3;
  // image row of this lane's piece q: (wave * 4 + q) * 8 + (lane >> 3); logical chunk c8 lands in physical chunk lane & 7.
  // Weight rows: image row j = wn' * 64 + g * 16 + i'  <->  row g * H + u0 + wn' * 16 + i' of W, i.e. piece q adds
  // (q >> 1) * H + (q & 1) * 8 rows to piece 0's
  const unsigned c8 = (unsigned)((lane & 7) ^ (lane >> 3));
  const unsigned rowa0 = (unsigned)(wave * 32 + (lane >> 3));
  const unsigned rowb0 = (unsigned)((wave & 1) * 2 * H + u0 + (wave >> 1) * 16 + (lane >> 3));

  f32x4 acc[4][4];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](char* stg, int ks, int part) {
    int s = (ks >= end0) + (ks >= end1) + (ks >= end2);  // uniform
    int start = ks >= end0 ? end0 : 0;
    start = ks >= end1 ? end1 : start;
    start = ks >= end2 ? end2 : start;
    const int kl = ks - start;
    const Seg& S = segs[s];
    const unsigned la = (unsigned)(S.lda * ES), lb = (unsigned)(S.ldb * ES);
    const __amdgpu_buffer_rsrc_t a = __builtin_amdgcn_make_buffer_rsrc((T*)S.A + (int64_t)m0 * S.lda, 0, (int)(BM * la), 0x00020000);
    const __amdgpu_buffer_rsrc_t b = __builtin_amdgcn_make_buffer_rsrc((T*)S.B, 0, (int)(4 * H * lb), 0x00020000);
    // past the segment's K (or past the last step): bit 30 set = beyond num_records, the load returns zeros.  Plain ALU on
    // purpose: selects here came back as exec-masked branches inside the loop
    const int segK = S.K;
    const unsigned oob = (unsigned)((int)(ks >= nsteps) | (int)(kl * BK + (int)c8 * EPC >= segK)) << 30;
    const unsigned kb = ((unsigned)(kl * 128) + c8 * 16u) | oob;
    unsigned xa[4], xb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      xa[q] = (rowa0 + (unsigned)(q * 8)) * la + kb;
      xb[q] = (rowb0 + (unsigned)((q >> 1) * H + (q & 1) * 8)) * lb + kb;
    }
    if (part != 1) cell_issue<4>(stg, a, xa, 0u, wave);
    if (part != 0) cell_issue<4>(stg + BM * 128, b, xb, 0u, wave);
  };
  cell_mainloop<T, BM, RB, NS>(acc, nsteps, issue, st0, st1, st2, st3);

  // Epilogue through LDS: the accumulators (one lane = i,f,g,o of a (row, unit): 16 lanes x 4 B runs) go to an f32 image
  // X[row][gate][32 units] (512 B per row; rows 0..63 in st0, 64..127 in st1), then every lane takes (row, 8 consecutive
  // units) items: 16-byte global loads / stores, whole 64- / 128-byte runs per row (the per-lane form issued 12 two- and
  // four-byte accesses per element).  16-byte slot s of a row sits at s ^ swz(row): conflict-free for the 4-byte writes
  // (the four 4-row groups of a wave land on the four 64-byte quarters) and for the 16-byte reads.
  auto swz = [](int row) { return (row & 1) ^ (((row >> 2) & 1) << 2) ^ ((((row >> 1) ^ (row >> 3)) & 1) << 3); };
  const unsigned uH = (unsigned)H;
  {
    const unsigned unit = u0 + wn * 16 + i;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    if (J.bias_a) {
#pragma unroll
      for (int g = 0; g < 4; ++g) bsum[g] = J.bias_a[g * uH + unit] + J.bias_b[g * uH + unit];
    }
    __syncthreads();  // every wave has read its last stage
    char* xw = wm ? st1 : st0;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lr = tm * 16 + gq * 4 + r;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int slot = g * 8 + wn * 4 + (i >> 2);
          *(cell_lds_f)(xw + lr * 512 + ((slot ^ swz(lr)) << 4) + (i & 3) * 4) = acc[tm][g][r] + bsum[g];
        }
      }
    __syncthreads();
  }
  const bool has_pre = J.pre != nullptr, has_cp = J.c_prev != nullptr;
  const float* prep = has_pre ? J.pre : J.c_out;  // stand-ins keep the loads unconditional (masked below)
  const unsigned pld = has_pre ? (unsigned)J.pre_ld : uH;
  const float* cprev = has_cp ? J.c_prev : J.c_out;
  const int lr = threadIdx.x >> 2, chunk = threadIdx.x & 3;
  const unsigned u = u0 + chunk * 8;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const char* xr = (k ? st1 : st0) + lr * 512;
    const unsigned row = m0 + k * 64 + lr;
    float x[4][8], pa[4][8], cp[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      ld8(prep + row * pld + (has_pre ? g * uH + u : 0u), pa[g]);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const f32x4v v = *(cell_lds_f4)(xr + (((g * 8 + chunk * 2 + hh) ^ swz(lr)) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) x[g][hh * 4 + e] = v[e];
      }
    }
    ld8(cprev + row * uH + u, cp);
    float ig[8], fg[8], gg[8], og[8], c[8], h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ig[e] = sigmoidf_(x[0][e] + (has_pre ? pa[0][e] : 0.f));
      fg[e] = sigmoidf_(x[1][e] + (has_pre ? pa[1][e] : 0.f));
      gg[e] = tanhf_(x[2][e] + (has_pre ? pa[2][e] : 0.f));
      og[e] = sigmoidf_(x[3][e] + (has_pre ? pa[3][e] : 0.f));
      c[e] = __builtin_fmaf(fg[e], has_cp ? cp[e] : 0.f, ig[e] * gg[e]);
      h[e] = og[e] * tanhf_(c[e]);
    }
    const unsigned o = row * uH + u;
    st8(J.c_out + o, c);
    st8t(J.h_out + o, h);
    if (J.h_out_f32) st8(J.h_out_f32 + o, h);
    T* go = J.gates_out + row * 4u * uH + u;
    st8t(go, ig);
    st8t(go + uH, fg);
    st8t(go + 2 * uH, gg);
    st8t(go + 3 * uH, og);
    if (J.hn_out) st8(J.hn_out + row * (unsigned)J.hn_ld + u, h);
  }
}

// ---------------------------------------------------------------------------------------------
// backward cell: BM (128 or 64) rows x 64 units
// ---------------------------------------------------------------------------------------------
template <typename T, int BM, int NS>
__global__ __launch_bounds__(kCellThreads, (BM + 64) * 128 * NS <= 80 * 1024 ? 2 : 1) void cell_bwd_kernel(BwdJobs<T> jobs) {
  constexpr int BN = 64, TM = BM / 32, NIA = BM / 32;
  constexpr int BK = CellOp<T>::BK, ES = (int)sizeof(T);
  constexpr int STAGE = (BM + BN) * 128;
  __shared__ __attribute__((aligned(1024))) char st0[STAGE];
  __shared__ __attribute__((aligned(1024))) char st1[STAGE];
  __shared__ __attribute__((aligned(1024))) char st2[NS > 2 ? STAGE : 16];
  __shared__ __attribute__((aligned(1024))) char st3[NS > 3 ? STAGE : 16];
  const BwdJob<T>& J = jobs.job[blockIdx.z];
  const int H = jobs.H;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 15, gq = lane >> 4;

  CellSegs sg;
  unsigned va[2][NIA], vb[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const Seg& S = J.seg[s];
    const bool on = S.K > 0;
    sg.a[s] = cell_rsrc(on ? (const T*)S.A + (int64_t)((jobs.glds & 8) ? 0 : m0) * S.lda : nullptr, (int64_t)BM * S.lda * ES);
    sg.b[s] = cell_rsrc(on ? (const T*)S.B + (int64_t)((jobs.glds & 16) ? 0 : n0) * S.ldb : nullptr, (int64_t)BN * S.ldb * ES);
#pragma unroll
    for (int q = 0; q < NIA; ++q) {
      const int row = (wave * NIA + q) * 8 + (lane >> 3);
      va[s][q] = (unsigned)row * (unsigned)(S.lda * ES) + (unsigned)((lane & 7) ^ (row & 7)) * 16u;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = (wave * 2 + q) * 8 + (lane >> 3);
      vb[s][q] = (unsigned)row * (unsigned)(S.ldb * ES) + (unsigned)((lane & 7) ^ (row & 7)) * 16u;
    }
  }
  sg.n0 = J.seg[0].K / BK;
  sg.n = sg.n0 + J.seg[1].K / BK;

  f32x4 acc[TM][2];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](char* stg, int ks, int part) {
    const bool s1 = ks >= sg.n0;
    const unsigned kb = ks < sg.n ? (unsigned)((s1 ? ks - sg.n0 : ks) * 128) : kCellOob;
    const __amdgpu_buffer_rsrc_t ra = s1 ? sg.a[1] : sg.a[0], rb = s1 ? sg.b[1] : sg.b[0];
    unsigned xa[NIA], xb[2];
#pragma unroll
    for (int q = 0; q < NIA; ++q) xa[q] = s1 ? va[1][q] : va[0][q];
#pragma unroll
    for (int q = 0; q < 2; ++q) xb[q] = s1 ? vb[1][q] : vb[0][q];
    if (part != 1) cell_issue<NIA>(stg, ra, xa, kb, wave);
    if (part != 0) cell_issue<2>(stg + BM * 128, rb, xb, kb, wave);
  };
  cell_mainloop<T, BM, BN, NS>(acc, sg.n, issue, st0, st1, st2, st3);

  // Epilogue through LDS (see the forward cell): dh -> X[row][64 units] f32 (256 B per row; the rows of wave row wm in
  // st<wm>), slot s of a row at s ^ swz(row); then (row, 8 units) items per lane with 16-byte global accesses.  Absent optional
  // inputs are read from a valid stand-in (c_cur) and masked by a uniform select: no branches in the unrolled body.
  auto swz = [](int row) { return (((row >> 2) & 3) << 2) ^ ((row >> 1) & 1); };
  __syncthreads();
  {
    char* xw = wm ? st1 : st0;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lr = tm * 16 + gq * 4 + r;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          const int slot = wn * 8 + tn * 4 + (i >> 2);
          *(cell_lds_f)(xw + lr * 256 + ((slot ^ swz(lr)) << 4) + (i & 3) * 4) = acc[tm][tn][r];
        }
      }
  }
  __syncthreads();
  const bool has_cp = J.c_prev != nullptr, has_e1 = J.ext != nullptr, has_e2 = J.ext2 != nullptr, first = J.first != 0;
  const unsigned uH = (unsigned)H;
  const float* cprev = has_cp ? J.c_prev : J.c_cur;
  const float* e1p = has_e1 ? J.ext : J.c_cur;
  const unsigned e1ld = has_e1 ? (unsigned)J.ext_ld : uH;
  const float* e2p = has_e2 ? J.ext2 : J.c_cur;
  const unsigned e2ld = has_e2 ? (unsigned)J.ext2_ld : uH;
  const int chunk = threadIdx.x & 7;
  const unsigned u = n0 + chunk * 8;
#pragma unroll
  for (int k = 0; k < BM / 32; ++k) {  // 32 rows x 8 chunks per pass
    const int wmk = k / (BM / 64), lr = (threadIdx.x >> 3) + 32 * (k % (BM / 64));
    const char* xr = (wmk ? st1 : st0) + lr * 256;
    const unsigned row = m0 + wmk * (BM / 2) + lr;
    const unsigned o = row * uH + u, o4 = row * 4u * uH + u;
    float dh[8], ig[8], fg[8], gg[8], og[8], cp[8], cc[8], dcin[8], e1[8], e2[8];
    ld8t(J.gates + o4, ig);
    ld8t(J.gates + o4 + uH, fg);
    ld8t(J.gates + o4 + 2 * uH, gg);
    ld8t(J.gates + o4 + 3 * uH, og);
    ld8(cprev + o, cp);
    ld8(J.c_cur + o, cc);
    ld8(J.dc + o, dcin);
    ld8(e1p + row * e1ld + u, e1);
    ld8(e2p + row * e2ld + u, e2);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const f32x4v v = *(cell_lds_f4)(xr + (((chunk * 2 + hh) ^ swz(lr)) << 4));
#pragma unroll
      for (int e = 0; e < 4; ++e) dh[hh * 4 + e] = v[e];
    }
    float dcn[8], dp[4][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // same order of operations as lstm_bwd_step_kernel (lstm.hip): ext, then ext2
      float d = dh[e];
      d += has_e1 ? e1[e] : 0.f;
      d += has_e2 ? e2[e] : 0.f;
      const float cpv = has_cp ? cp[e] : 0.f;
      const float tc = tanhf_(cc[e]);
      float dc = d * og[e] * (1.f - tc * tc);
      dc += first ? 0.f : dcin[e];
      const float d_o = d * tc;
      const float d_i = dc * gg[e], d_f = dc * cpv, d_g = dc * ig[e];
      dcn[e] = dc * fg[e];
      dp[0][e] = d_i * ig[e] * (1.f - ig[e]);
      dp[1][e] = d_f * fg[e] * (1.f - fg[e]);
      dp[2][e] = d_g * (1.f - gg[e] * gg[e]);
      dp[3][e] = d_o * og[e] * (1.f - og[e]);
    }
    st8(J.dc + o, dcn);
#pragma unroll
    for (int g = 0; g < 4; ++g) st8t(J.dg_out + o4 + g * uH, dp[g]);
  }
}

// sum over t of the saved gate gradients of layer 0 -> f32 [B,4H]: what the time-constant input's gradients contract with
// (the generic cells keep this sum as a read-modify-write of 32 MB per step; one pass over the saved dgates is cheaper)
template <typename T>
__global__ __launch_bounds__(256) void cell_dgsum_kernel(const T* __restrict__ dg, float* __restrict__ out, int T_, int64_t n) {
  const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (e >= n) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int t = 0; t < T_; ++t) {
    float v[8];
    ld8t(dg + (int64_t)t * n + e, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += v[k];
  }
  st8(out + e, acc);
}

static bool cell_misaligned(const void* p) { return (((uintptr_t)p) & 15) != 0; }

// es: bytes per operand element
static bool cell_seg_ok(const Seg& s, int64_t rows_a, int64_t rows_b, int kmult, int es) {
  if (s.K == 0) return true;
  const int epc = 16 / es;
  if (!s.a_kc || !s.b_kc || s.a_rmod || (s.K % kmult) || (s.lda % epc) || (s.ldb % epc)) return false;
  if ((((uintptr_t)s.A) | ((uintptr_t)s.B)) & 15) return false;
  // 32-bit buffer offsets below kCellOob
  return rows_a * s.lda * es < (int64_t)kCellOob && rows_b * s.ldb * es < (int64_t)kCellOob;
}

template <typename T>
bool cell_fwd_big_ok(const FwdJobs<T>& jobs, int nj) {
  if (jobs.B % 128 || jobs.H % 32 || (int64_t)jobs.B * 4 * jobs.H * 4 >= (1LL << 31)) return false;
  for (int j = 0; j < nj; ++j) {
    const FwdJob<T>& J = jobs.job[j];
    if ((J.pre && J.pre_ld * jobs.B * 4 >= (1LL << 31)) || (J.hn_out && J.hn_ld * jobs.B * 4 >= (1LL << 31))) return false;
    // 16-byte epilogue accesses
    if ((J.pre && (J.pre_ld % 4)) || (J.hn_out && (J.hn_ld % 4))) return false;
    if (cell_misaligned(J.pre) || cell_misaligned(J.c_prev) || cell_misaligned(J.c_out) || cell_misaligned(J.h_out) ||
        cell_misaligned(J.h_out_f32) || cell_misaligned(J.gates_out) || cell_misaligned(J.hn_out))
      return false;
  }
  for (int j = 0; j < nj; ++j)
    for (int s = 0; s < 2; ++s) {
      if (!cell_seg_ok(jobs.job[j].seg[s], 128, 4 * (int64_t)jobs.H, CellOp<T>::BK, sizeof(T))) return false;
      if (!cell_seg_ok(jobs.job[j].xseg[s], 128, 4 * (int64_t)jobs.H, CellOp<T>::EPC, sizeof(T))) return false;
    }
  return true;
}

template <typename T>
bool cell_bwd_big_ok(const BwdJobs<T>& jobs, int nj) {
  if (jobs.B % 128 || jobs.H % 64 || (int64_t)jobs.B * 4 * jobs.H * 4 >= (1LL << 31)) return false;
  for (int j = 0; j < nj; ++j) {
    const BwdJob<T>& J = jobs.job[j];
    if ((J.ext && J.ext_ld * jobs.B * 4 >= (1LL << 31)) || (J.ext2 && J.ext2_ld * jobs.B * 4 >= (1LL << 31))) return false;
    if ((J.ext && (J.ext_ld % 4)) || (J.ext2 && (J.ext2_ld % 4))) return false;
    if (cell_misaligned(J.ext) || cell_misaligned(J.ext2) || cell_misaligned(J.gates) || cell_misaligned(J.c_prev) ||
        cell_misaligned(J.c_cur) || cell_misaligned(J.dc) || cell_misaligned(J.dg_out))
      return false;
  }
  for (int j = 0; j < nj; ++j)
    for (int s = 0; s < 2; ++s)
      if (!cell_seg_ok(jobs.job[j].seg[s], 128, 64, CellOp<T>::BK, sizeof(T))) return false;
  return true;
}

template <typename T>
int launch_cell_fwd_big(const FwdJobs<T>& jobs, int nj, hipStream_t st) {
  FwdJobs<T> jd = jobs;
  jd.glds = 0;
  const dim3 grid((unsigned)(jobs.B / 128), (unsigned)(jobs.H / 32), (unsigned)nj), block(kCellThreads);
  hipLaunchKernelGGL((cell_fwd_kernel<T, 2>), grid, block, 0, st, jd);  // (two ring stages; three measured no faster)
  return fh_launch_status();
}

template <typename T>
int launch_cell_bwd_big(const BwdJobs<T>& jobs, int nj, hipStream_t st) {
  BwdJobs<T> jd = jobs;
  jd.glds = 0;
  const dim3 grid((unsigned)(jobs.B / 64), (unsigned)(jobs.H / 64), (unsigned)nj), block(kCellThreads);
  hipLaunchKernelGGL((cell_bwd_kernel<T, 64, 4>), grid, block, 0, st, jd);  // (64-row tiles, four stages; 128 rows / three stages no faster)
  return fh_launch_status();
}

template <typename T>
int launch_cell_dgsum(const T* dg, float* out, int T_, int64_t n, hipStream_t st) {
  if ((n % 8) || cell_misaligned(dg) || cell_misaligned(out)) return FHVAE_ERR_ALIGN;
  hipLaunchKernelGGL(cell_dgsum_kernel<T>, dim3((unsigned)fh_cdiv(n / 8, 256)), dim3(256), 0, st, dg, out, T_, n);
  return fh_launch_status();
}

// explicit instantiations (the kernels' device stubs and the host entry points of both operand types)
#define FH_CELL_INST(T)                                                    \
  template __global__ void cell_fwd_kernel<T, 2>(FwdJobs<T>);             \
  template __global__ void cell_bwd_kernel<T, 64, 4>(BwdJobs<T>);         \
  template __global__ void cell_dgsum_kernel<T>(const T*, float*, int, int64_t); \
  template bool cell_fwd_big_ok<T>(const FwdJobs<T>&, int);               \
  template bool cell_bwd_big_ok<T>(const BwdJobs<T>&, int);               \
  template int launch_cell_fwd_big<T>(const FwdJobs<T>&, int, hipStream_t); \
  template int launch_cell_bwd_big<T>(const BwdJobs<T>&, int, hipStream_t); \
  template int launch_cell_dgsum<T>(const T*, float*, int, int64_t, hipStream_t);
FH_CELL_INST(u16)
FH_CELL_INST(float)

}  // namespace fh
