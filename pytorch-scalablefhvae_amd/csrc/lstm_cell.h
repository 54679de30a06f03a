// lstm_cell.h -- the per-wavefront-step LSTM cells: job descriptors shared by the generic step kernels (lstm.hip) and the
// large-tile bf16 cells (lstm_cell.hip).
#pragma once
#include "gemm_core.h"

namespace fh {

// ---------------------------------------------------------------------------------------------
// forward cell
// ---------------------------------------------------------------------------------------------
template <typename T>
struct FwdJob {
  Seg seg[2];           // seg0: layer input x W_ih (l >= 1), seg1: h_{t-1} x W_hh (t >= 1); K = 0 when absent
  Seg xseg[2];          // large-tile cells only (layer 0), directly behind seg[]: x_t x W_ih[0][:, :I] and xc x W_ih[0][:, I:], K any
                        // multiple of 8 -- the cell multiplies the layer's input itself instead of reading a precomputed `pre`
  const float* pre;     // [B,4H] additive term incl. biases (l == 0) or NULL
  int64_t pre_ld;
  const float* bias_a;  // [4H] (l >= 1) or NULL
  const float* bias_b;
  const float* c_prev;  // [B,H] or NULL (t == 0)
  float* c_out;         // [B,H]
  T* h_out;             // [B,H] operand dtype
  float* h_out_f32;     // optional f32 copy of h (top layer in bf16 mode: feeds the f32 Gaussian head)
  T* gates_out;         // [B,4H] activated i,f,g,o (operand dtype: bf16 halves the cell's dominant HBM write)
  float* hn_out;        // optional slot in the (B, L*H) final-state buffer (t == T-1)
  int64_t hn_ld;
};
template <typename T>
struct FwdJobs {
  int B, H;
  int glds;  // use the LDS-DMA main loop where the tile allows it
  FwdJob<T> job[FHVAE_MAX_LAYERS];
};

// ---------------------------------------------------------------------------------------------
// backward cell: dh_t = dg^l_{t+1} . W_hh[l] + dg^{l+1}_t . W_ih[l+1] (+ external), then the
// elementwise LSTM backward in the epilogue -> dg^l_t (pre-activation gate gradients)
// ---------------------------------------------------------------------------------------------
template <typename T>
struct BwdJob {
  Seg seg[2];
  const float* ext;   // [B,H] external dh (top layer: d_hs_top[t]) or NULL
  int64_t ext_ld;
  const float* ext2;  // [B,H] slot of d_hn (t == T-1) or NULL
  int64_t ext2_ld;
  const T* gates;       // [B,4H] saved activations (operand dtype)
  const float* c_prev;  // [B,H] or NULL
  const float* c_cur;   // [B,H]
  float* dc;            // [B,H] running dL/dc (already multiplied by f of the later step)
  int first;            // 1 at t == T-1: dc input is zero
  T* dg_out;            // [B,4H]
  float* dgsum;         // optional [B,4H] running f32 sum over t (layer 0 with a time-constant input)
};
template <typename T>
struct BwdJobs {
  int B, H;
  int glds;
  BwdJob<T> job[FHVAE_MAX_LAYERS];
};

// Large-tile cells (lstm_cell.hip; T = u16: bf16 operands, T = float: exact-f32 MFMA): every job's operands K-contiguous
// (a_kc = b_kc = 1), 16-byte aligned, K a multiple of 64 (bf16) / 32 (f32) -- layer-0 input segments: of 8 / 4 --, B a multiple
// of 128, H of 64.  The predicates check one launch's jobs.
template <typename T>
bool cell_fwd_big_ok(const FwdJobs<T>& jobs, int nj);
template <typename T>
bool cell_bwd_big_ok(const BwdJobs<T>& jobs, int nj);
template <typename T>
int launch_cell_fwd_big(const FwdJobs<T>& jobs, int nj, hipStream_t st);
template <typename T>
int launch_cell_bwd_big(const BwdJobs<T>& jobs, int nj, hipStream_t st);  // ignores BwdJob::dgsum: launch_cell_dgsum after the loop
// out[n] (f32) = sum over t of dg[t][n], n = B * 4H a multiple of 8
template <typename T>
int launch_cell_dgsum(const T* dg, float* out, int T_, int64_t n, hipStream_t st);

}  // namespace fh
