/*
 * fhvae_hip.h -- C ABI of libfhvae_hip.so: the MI355X (gfx950) hot path of the ScalableFHVAE
 * training step.  The reference (BurnhamG/PyTorch-ScalableFHVAE) is pure Python on stock ATen
 * CPU ops and has no FFI; every entry point below names the reference lines it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; all buffers are caller-owned DEVICE memory, dense row-major
 *     unless a leading dimension (ld*, in ELEMENTS) is given; kernels never allocate or free.
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream); no internal synchronisation, no global state, re-entrant -> graph-capturable.
 *   - return value: 0 on success; FHVAE_ERR_* (<0) for argument errors detected on the host
 *     before anything is launched; a positive hipError_t if a launch failed.
 *   - dtype: FHVAE_F32 computes with exact-f32 MFMA (v_mfma_f32_16x16x4_f32) -- the parity mode;
 *     FHVAE_BF16 uses bf16 MFMA operands with f32 accumulation and f32 cell state.
 *   - "time-major" activations are (T, B, X): row index t*B + b.
 */
#ifndef FHVAE_HIP_H
#define FHVAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FHVAE_ABI_VERSION 11

enum { FHVAE_F32 = 0, FHVAE_BF16 = 1 };

enum {
  FHVAE_OK = 0,
  FHVAE_ERR_NULL = -1,   /* required pointer is NULL */
  FHVAE_ERR_SHAPE = -2,  /* non-positive / inconsistent dimension */
  FHVAE_ERR_DTYPE = -3,  /* unknown dtype code */
  FHVAE_ERR_ALIGN = -4,  /* pointer / leading dimension not aligned as required */
  FHVAE_ERR_LIMIT = -5   /* size beyond what the kernels index (int32 rows/cols) */
};

#define FHVAE_MAX_LAYERS 4
/* BF16 mode: the first bytes of fhvae_lstm_desc.lp are the sync block of the persistent (cluster) recurrence kernels;
   u32 word 0 = status of the last such launch on that workspace: 0 ok, non-zero = the launch gave up (outputs invalid). */
#define FHVAE_LSTM_SYNC_BYTES 16384

int fhvae_abi_version(void);
/* human-readable text for a return code (static storage) */
const char* fhvae_strerror(int code);

/* ------------------------------------------------------------------------------------------
 * Linear layer:  y[M,N] = act(x[M,K] . w[N,K]^T + b[N])        (act = ReLU if relu != 0)
 * replaces nn.Linear / VariableLinearLayer, simple_fhvae.py:127-134, :208-213.
 * dtype selects the MFMA operand type of x and w; y (and b) are f32; y_lp (optional, may be
 * NULL) receives a copy of y in the operand dtype (for chaining bf16 layers).
 * ------------------------------------------------------------------------------------------ */
int fhvae_linear_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* b,
                     float* y, int64_t ldy, void* y_lp, int64_t M, int64_t K, int64_t N, int relu,
                     int dtype, void* stream);

/* Backward of the above.  dy[M,N] f32 is the upstream gradient; if relu != 0, `y` (the forward
 * OUTPUT) masks it (dy *= y > 0) and the masked gradient is written to `dy_masked` (M*N f32
 * workspace, required iff relu).  Produces dx[M,K] f32 (may be NULL; += if dx_accumulate), and ACCUMULATES
 * dw[N,K] += dy^T x, db[N] += colsum(dy) (either may be NULL).  x, w are f32 here (the weight
 * gradient contraction runs on exact-f32 MFMA in both modes for round 1). */
int fhvae_linear_bwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* y,
                     int64_t ldy, const float* dy, int64_t lddy, float* dy_masked, float* dx,
                     int64_t lddx, float* dw, int64_t lddw, float* db, int64_t M, int64_t K,
                     int64_t N, int relu, int dx_accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Gaussian head + reparameterisation (K2):  mu = h.Wmu^T + bmu, logvar = h.Wlv^T + blv,
 * sample = mu + eps * exp(0.5*logvar)   -- GaussianLayer.forward, simple_fhvae.py:211-216.
 * eps is supplied by the caller (the reference draws it with randn_like, :215); eps == NULL
 * skips the sample (decoder head: x_sample is computed and never used, simple_fhvae.py:102).
 * h[M,K]; mu, logvar, sample, eps: [M,D] f32 dense.
 * ------------------------------------------------------------------------------------------ */
int fhvae_gauss_head_reparam_fwd(const void* h, int64_t ldh, const void* w_mu, const void* w_lv,
                                 const float* b_mu, const float* b_lv, const float* eps, float* mu,
                                 float* logvar, float* sample, int64_t M, int64_t K, int64_t D,
                                 int dtype, void* stream);
/* mu | logvar of a Gaussian head WITHOUT sampling (the decoder's per-frame head, simple_fhvae.py:98-103, :211-213) as ONE
 * projection over the M = T*B rows: h_lp [M,K] bf16, w_pair_lp = the two bf16 weight matrices stacked [2D,K]; out[M, 2D] f32 (row
 * stride ldo): mu in columns [0,D), logvar in [D,2D). */
int fhvae_gauss_head_pair_fwd(const void* h_lp, int64_t ldh, const void* w_pair_lp, const float* b_mu, const float* b_lv,
                              float* out, int64_t ldo, int64_t M, int64_t K, int64_t D, void* stream);
/* The stacked bf16 operands of a Gaussian head from its f32 master weights w_mu, w_lv [D,K] (nn.Linear layout,
 * simple_fhvae.py:197-198), one launch: wl_pair [2D,K] (forward: fhvae_gauss_head_pair_fwd) and wt_pair [K,ldt] =
 * [w_mu^T | w_lv^T | 0] (backward: fhvae_gauss_head_bwd_pair), ldt >= 2D. */
int fhvae_head_pair_weights(const float* w_mu, const float* w_lv, void* wl_pair, void* wt_pair, int64_t ldt,
                            int64_t D, int64_t K, void* stream);
/* sample[m,c] = out[m,c] + eps[m,c] * exp(0.5 * out[m,D+c]) on the side-by-side (mu | logvar) buffer of
 * fhvae_gauss_head_pair_fwd (simple_fhvae.py:214-216); mu / logvar (may be NULL): contiguous [M,D] copies of the two halves. */
int fhvae_gauss_reparam_pair_fwd(const float* out, int64_t ldo, const float* eps, float* sample, float* mu, float* logvar,
                                 int64_t M, int64_t D, void* stream);
/* g_lp[m, 0..D) = bf16(d_mu + d_sample), [D..2D) = bf16(d_logvar + d_sample * eps * 0.5 * exp(0.5 logvar)), [2D..ldg) = 0:
 * the upstream gradient of both linear layers of a head as one bf16 operand (any of d_mu / d_logvar / d_sample may be NULL;
 * d_sample needs eps and logvar; d_sample has row stride ld_s -- a column slice of the following net's input gradient is taken
 * as it is --, logvar has row stride ld_lv).  db_mu / db_lv [D] (may be NULL): the bias gradients, += the
 * column sums of the two halves of g_lp (nn.Linear's bias backward) from the same launch where a workgroup covers whole rows. */
int fhvae_gauss_reparam_bwd_pair(const float* d_mu, const float* d_logvar, const float* d_sample, int64_t ld_s, const float* eps,
                                 const float* logvar, int64_t ld_lv, void* g_lp, int64_t ldg, float* db_mu, float* db_lv,
                                 int64_t M, int64_t D, void* stream);
/* Backward of both linear layers of a head (nn.Linear backward at simple_fhvae.py:197-198,:210-211) from the bf16 operand
 * g_lp [M,ldg] (fhvae_gauss_reparam_bwd_pair or fhvae_elbo_bwd's d_x_pair_lp): dh[M,K] = g . [W_mu; W_lv] (OVERWRITTEN, may be
 * NULL), dw_mu / dw_lv [D,K] += g^T . h_lp (skipped when both are NULL), db_mu / db_lv [D] += column sums of g (the sum of the
 * col_sum_rows partial rows col_sum[.][2D] when given, else reduced here).  wt_pair [K,ldt] from fhvae_head_pair_weights. */
int fhvae_gauss_head_bwd_pair(const void* h_lp, int64_t ldh, const void* wt_pair, int64_t ldt, const void* g_lp, int64_t ldg,
                              const float* col_sum, int64_t col_sum_rows, float* dh, int64_t lddh, float* dw_mu, float* dw_lv, float* db_mu,
                              float* db_lv, int64_t M, int64_t K, int64_t D, void* stream);

/* Elementwise part of the head's backward: given upstream d_mu, d_logvar, d_sample (any may be
 * NULL = zero) produce the gradients w.r.t. the two linear outputs:
 *   g_mu = d_mu + d_sample ; g_lv = d_logvar + d_sample * eps * 0.5 * exp(0.5*logvar)
 * (the two linear layers' own backward is fhvae_linear_bwd). n = M*D elements. */
int fhvae_gauss_reparam_bwd(const float* d_mu, const float* d_logvar, const float* d_sample,
                            const float* eps, const float* logvar, float* g_mu, float* g_lv,
                            int64_t n, void* stream);
/* The whole backward of fhvae_gauss_head_reparam_fwd (f32 h) in four launches: g_ws[M,2D] = [g_mu | g_lv] (workspace,
 * as fhvae_gauss_reparam_bwd computes them), dh[M,K] = g_mu.W_mu + g_lv.W_lv (one two-segment contraction; NULL to
 * skip), dw_mu/dw_lv[D,K] += g^T.h (one contraction, output split), db_mu/db_lv[D] += column sums of g.
 * Replaces reparam_bwd + 2 x fhvae_linear_bwd (7 launches) -- GaussianLayer backward, simple_fhvae.py:193-216. */
int fhvae_gauss_head_bwd(const float* h, int64_t ldh, const float* w_mu, const float* w_lv,
                         const float* d_mu, const float* d_logvar, const float* d_sample,
                         const float* eps, const float* logvar, float* g_ws, float* dh, int64_t lddh,
                         float* dw_mu, float* dw_lv, float* db_mu, float* db_lv, int64_t M, int64_t K,
                         int64_t D, void* stream);
/* ------------------------------------------------------------------------------------------
 * Multi-layer LSTM over a whole segment (K1), step-fused cells: one launch per wavefront step
 * computes the 4-gate contraction on MFMA and applies sigmoid/tanh + the cell update in the
 * epilogue.  There is NO reference body (fhvae.py:14 raises NotImplementedError); semantics are
 * torch.nn.LSTM(batch_first) CPU: gate order i,f,g,o; gates = W_ih x + b_ih + W_hh h + b_hh.
 * It stands where the FC pre-encoders/decoder stand in simple_fhvae.py:160-164,186-190,240-244.
 *
 * Layer-0 input at step t is [x_t (I cols) || xc (Ic cols)]: x is time-major (T,B,I) (may be
 * NULL with I = 0), xc (B,Ic) is constant over time (may be NULL with Ic = 0); w_ih[0] is
 * [4H, I+Ic].  Layers l >= 1 take h^{l-1}_t (H cols).  All layers share H.
 * ------------------------------------------------------------------------------------------ */
typedef struct fhvae_lstm_desc {
  int32_t dtype;  /* FHVAE_F32 | FHVAE_BF16: MFMA operand type used inside */
  int32_t L;      /* layers, 1..FHVAE_MAX_LAYERS */
  int64_t B, T, I, Ic, H;
  /* inputs and parameters are ALWAYS f32 (master copies) */
  const float* x;   /* (T,B,I) time-major */
  const void* x_lp; /* BF16 mode, optional: the same x already in bf16 (e.g. from fhvae_to_time_major); then the
                       forward does not cast x again (two encoders share one input) */
  const float* xc;  /* (B,Ic) */
  const float* w_ih[FHVAE_MAX_LAYERS]; /* [4H, I+Ic] (l=0) / [4H, H] */
  const float* w_hh[FHVAE_MAX_LAYERS]; /* [4H, H] */
  const float* b_ih[FHVAE_MAX_LAYERS]; /* [4H] */
  const float* b_hh[FHVAE_MAX_LAYERS];
  /* forward outputs, saved for backward */
  void* hs;      /* (L,T,B,H) in the operand dtype (f32 or bf16): h^l_t */
  float* cs;     /* (L,T,B,H) f32: c^l_t */
  void* gates;   /* (L,T,B,4H) in the operand dtype: activated i,f,g,o.  A workspace between a forward and ITS backward: the
                    per-step cells keep column blocks of H per gate, the persistent schedules keep the four gates of a unit
                    quad together (csrc/lstm_cluster.hip, cl_goff); do not read it from outside */
  float* hn;     /* (B, L*H) f32: final hidden state of every layer, concatenated (may be NULL) */
  float* hs_top_f32; /* (T,B,H) f32 copy of the top layer's h_t (BF16 mode, may be NULL; in F32 mode
                        the top layer is hs + (L-1)*T*B*H and this must be NULL) */
  float* pre;    /* workspace (T,B,4H) f32 (I > 0) or (B,4H) (I == 0): layer-0 input projection (the persistent
                    schedules only use its first (B,4H): they multiply x_t by W_ih[0] inside the kernel; the large-tile
                    bf16 step cells of csrc/lstm_cell.hip multiply the whole layer-0 input themselves and leave it unused) */
  void* lp;      /* F32 mode: optional workspace of fhvae_lstm_lp_bytes() bytes (may be NULL): the forward leaves the
                    transposed f32 weights in it and the backward cells then stage both operands by LDS-DMA (64 -> 40 us per
                    launch at B = 2048, H = 256); without it they read the master weights as K-major operands.
                    BF16 mode: workspace of fhvae_lstm_lp_bytes() bytes; the forward fills it with bf16
                    copies of x, xc, the weights and the transposed weights, the backward reuses it.  It also
                    holds the persistent schedules' sync block (first FHVAE_LSTM_SYNC_BYTES) and their exchange
                    buffer (2*L*B*4H bf16): keep it alive and untouched between the forward and its backward */
  void* hn_lp;   /* BF16 mode, optional (may be NULL): (B, L*H) bf16 copy of hn -- the operand of a bf16 Gaussian head
                    (simple_fhvae.py:193-216) straight from the kernel that produced the final states, instead of a cast launch
                    per head and step.  Always filled when set (the persistent kernels store it with hn; the per-step schedules cast hn at the end). */
  /* BF16 mode, optional (head_w_mu == NULL: none): the Gaussian head that consumes this net's states (GaussianLayer,
     simple_fhvae.py:193-216; f32 master weights head_w_mu / head_w_lv [head_D, head_K]).  The forward's operand-cast launch then
     also writes the head's stacked bf16 operands -- head_wl [2 head_D, head_K] = [W_mu; W_lv] and head_wt [head_K, head_ldt] =
     [W_mu^T | W_lv^T | 0] (head_ldt >= 2 head_D), exactly what fhvae_head_pair_weights produces -- instead of one more launch per
     head and step. */
  const float* head_w_mu;
  const float* head_w_lv;
  void* head_wl;
  void* head_wt;
  int64_t head_D, head_K, head_ldt;
  int32_t* sticky_status; /* BF16 mode, optional (may be NULL): int32 device word that the library never clears.  A persistent
                    launch that gives up ORs its status code into it as well as into the workspace's status word (which
                    the next forward on that workspace re-arms): the failure stays visible however late the host looks. */
} fhvae_lstm_desc;

int64_t fhvae_lstm_lp_bytes(const fhvae_lstm_desc* d);
/* Floats the `pre` workspace must hold for this descriptor on the current device (evaluate with `lp` set, as for
   fhvae_lstm_form): (T,B,4H) for the per-step cells of gemm_core.h with a per-frame input, (B,4H) for the persistent
   schedules (they multiply x_t inside the kernel), 1 for the large-tile bf16 cells (csrc/lstm_cell.hip: the whole layer-0
   input projection is theirs; they then REQUIRE 16-byte aligned buffers: the forward returns FHVAE_ERR_ALIGN instead of
   falling back to a schedule that would need the full buffer).  `pre` must still be non-NULL. */
int64_t fhvae_lstm_pre_elems(const fhvae_lstm_desc* d);
/* Identifies the layout of what fhvae_lstm_seq_fwd saves for the backward (gates, schedule workspaces): it follows from the
 * schedule the library picks for this descriptor AND the FHVAE_* environment switches at call time.  A caller that may change
 * either between a forward and its backward keeps the forward's value and checks it before fhvae_lstm_seq_bwd (< 0: bad
 * descriptor). */
int fhvae_lstm_layout_id(const fhvae_lstm_desc* d);

/* Floats fhvae_lstm_bwd_desc.ws_below must hold (0: may be NULL). */
int64_t fhvae_lstm_ws_below_elems(const fhvae_lstm_desc* d);
/* Which schedule fhvae_lstm_seq_fwd/_bwd take for this descriptor on the current device: 0 = one launch per wavefront
   step; 1 = persistent cluster kernel, waves split the batch rows; 2 = persistent cluster kernel, waves split the
   contraction (small batches).  1 and 2 need the GPU to themselves while they run (256 co-resident workgroups);
   FHVAE_NO_CLUSTER=1 in the environment forces 0. */
int fhvae_lstm_form(const fhvae_lstm_desc* d);
int fhvae_lstm_seq_fwd(const fhvae_lstm_desc* d, void* stream);

typedef struct fhvae_lstm_bwd_desc {
  fhvae_lstm_desc f;      /* the forward descriptor (same buffers, already filled by fwd) */
  const float* d_hs_top;  /* (T,B,H) f32 gradient w.r.t. the top layer's h_t (may be NULL) */
  const float* d_hn;      /* (B, L*H) f32 gradient w.r.t. hn (may be NULL) */
  /* workspaces */
  void* dgates;   /* (L,T,B,4H) operand dtype: gradient w.r.t. pre-activation gates */
  float* dgsum;   /* (B,4H) f32: sum_t dgates of layer 0 (required iff Ic > 0) */
  float* dc;      /* (L,B,H) f32: running cell-state gradient */
  /* outputs, f32 (ACCUMULATED: += ; any may be NULL) */
  float* dw_ih[FHVAE_MAX_LAYERS];
  float* dw_hh[FHVAE_MAX_LAYERS];
  float* db_ih[FHVAE_MAX_LAYERS];
  float* db_hh[FHVAE_MAX_LAYERS];
  float* d_xc;    /* (B,Ic) f32, OVERWRITTEN (may be NULL) */
  int32_t phase;  /* 0: everything; 1: the recurrence (dgates, dgsum, d_xc) only; 2: the weight/bias gradient
                     contractions only (reads what phase 1 left in dgates/dgsum) -- lets the host put phase 2 on a
                     second stream, under the next net's latency-bound recurrence */
  float* ws_below; /* (T,B,H) f32 workspace, required when fhvae_lstm_ws_below_elems(&f) > 0 (persistent backward, layer by
                      layer, H != 256: the from-above gradient dg^{l+1}.W_ih^{l+1} reaches the lower layer through it;
                      at H = 256 the lower layer's launch computes that term itself) */
} fhvae_lstm_bwd_desc;

int fhvae_lstm_seq_bwd(const fhvae_lstm_bwd_desc* d, void* stream);
/* Phase 2 of n backward passes at once (their phase 1 must have been enqueued on `stream` before): the weight / bias
 * gradients of several nets.  In BF16 mode the long contractions dW[4H,.] += dgates^T . [x | h] over the T*B rows of ALL the
 * descriptors run as ONE grouped launch (csrc/wgrad.hip); the host defers them to the end of the backward pass, where the
 * three nets of the model together fill the chip with whole tiles and few K slices. */
/* `extra` (may be NULL with n_extra = 0): further contractions C[M,N] += A[K,M]^T . B[K,N] of the same kind (the heads' weight
 * gradients) that ride in the same grouped launch; each must satisfy fhvae_wgrad_desc_ok. */
typedef struct fhvae_wgrad_desc {
  const void* a; int64_t lda;  /* [K, lda] bf16: the contraction index is the ROW */
  int64_t a_col0;              /* `a` points a_col0 columns into the rows of its buffer (a column slice) */
  const void* b; int64_t ldb;  /* [K, ldb] bf16 */
  float* c; int64_t ldc;       /* [M, ldc] f32, accumulated */
  int64_t M, N, K;
} fhvae_wgrad_desc;
int fhvae_wgrad_desc_ok(const fhvae_wgrad_desc* p); /* 1 = the grouped kernel takes it (alignment, ranges), else 0 */
int fhvae_lstm_param_grads_multi(const fhvae_lstm_bwd_desc* const* descs, int n, const fhvae_wgrad_desc* extra, int n_extra,
                                 void* stream);
/* The contraction itself: C[M,N] (f32, ldc) += A[K,M]^T . B[K,N], bf16 operands whose ROW index is the contraction index
 * (lda, ldb in elements, multiples of 8; 16-byte aligned bases; K*ld*2 < 2^30) -- dW += dY^T X of a linear / LSTM layer over
 * K = batch x time rows (autograd of nn.Linear, simple_fhvae.py:127-134; of the LSTM body missing at fhvae.py:14).
 * FHVAE_ERR_ALIGN when the preconditions do not hold. */
int fhvae_wgrad_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, float* c, int64_t ldc, int64_t M,
                     int64_t N, int64_t K, void* stream);
/* The same contraction with f32 operands on exact-f32 MFMA (the parity mode's weight gradients; lda, ldb multiples of 4,
 * K*ld*4 < 2^31). */
int fhvae_wgrad_f32(const float* a, int64_t lda, const float* b, int64_t ldb, float* c, int64_t ldc, int64_t M, int64_t N,
                    int64_t K, void* stream);

/* c[M,N] (f32, ldc) = a[M,K] . b[N,K]^T (+ bias[N], may be NULL): bf16 operands with the contraction index CONTIGUOUS in both
 * (lda, ldb in elements, multiples of 8; K % 64 == 0, N % 4 == 0; 16-byte aligned bases; M*lda*2 < 2^31) -- an activation matrix
 * over M = batch x time rows times a weight matrix as nn.Linear / nn.LSTM store it (y = x W^T, simple_fhvae.py:130; for the LSTM
 * body missing at fhvae.py:14: the from-above term dh^l += dgates^{l+1} . W_ih^{l+1} of the backward, for all time steps at
 * once).  One tile per CU, every row of `a` read once (csrc/proj.hip).  FHVAE_ERR_ALIGN when the preconditions do not hold. */
int fhvae_proj_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, const float* bias, float* c, int64_t ldc,
                    int64_t M, int64_t N, int64_t K, void* stream);

/* ------------------------------------------------------------------------------------------
 * mu2 gather (K4): mu2[b,:] = table[idx[b],:]  -- torch.gather, simple_fhvae.py:53.
 * bwd: dtable[idx[b],:] += scale * dmu2[b,:] (float atomics; duplicate indices accumulate; rows
 * outside [0,S) are skipped, which is how a row shard ignores other shards' rows).
 * fwd/bwd take idx RELATIVE to the table passed (a shard passes idx - row0 via `idx_offset`).
 * idx is int64 (DataLoader collate, train_model.py:445); rows outside [0,S) -> error flag:
 * the kernel writes zeros for them and sets *oob_flag (int32 device word, may be NULL).
 * ------------------------------------------------------------------------------------------ */
int fhvae_mu2_gather_fwd(const float* table, const int64_t* idx, int64_t idx_offset, float* mu2,
                         int64_t B, int64_t S, int64_t D, int32_t* oob_flag, void* stream);
int fhvae_mu2_gather_bwd(const float* dmu2, const int64_t* idx, int64_t idx_offset, float* dtable,
                         int64_t B, int64_t S, int64_t D, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused variational lower bound (K3) -- simple_fhvae.py:105-116 with log_gauss :56-60, kld :62-69.
 *   log_pmu2   = sum_d logN(mu2; 0, 1)
 *   neg_kld_z2 = -sum_d KL(N(z2_mu, e^z2_lv) || N(mu2, 0.25))
 *   neg_kld_z1 = -sum_d KL(N(z1_mu, e^z1_lv) || N(0, 1))
 *   log_px_z   = sum_{t,f} logN(x; x_mu, e^x_lv)
 *   lower_bound = log_px_z + neg_kld_z1 + neg_kld_z2 + log_pmu2 / num_segs
 * x, x_mu, x_lv are addressed as base + b*sb + t*st + f (f contiguous) so that both the
 * batch-major (B,T,F) layout of the reference and the time-major (T,B,F) layout of the LSTM
 * decoder are read in place.  num_segs is int64 (B,) or NULL with nsegs_scalar used instead.
 * ------------------------------------------------------------------------------------------ */
typedef struct fhvae_elbo_desc {
  int64_t B, T, F, D1, D2;
  const float* x;    int64_t x_sb, x_st;
  const float* x_mu; const float* x_lv; int64_t xo_sb, xo_st; /* shared strides of x_mu/x_lv */
  const float* z1_mu; const float* z1_lv;  /* (B,D1) */
  const float* z2_mu; const float* z2_lv;  /* (B,D2) */
  const float* mu2;                        /* (B,D2) */
  const int64_t* num_segs; double nsegs_scalar;
  /* outputs (B,) f32 */
  float* lower_bound; float* log_px_z; float* neg_kld_z1; float* neg_kld_z2; float* log_pmu2;
} fhvae_elbo_desc;

int fhvae_elbo_fwd(const fhvae_elbo_desc* d, void* stream);

typedef struct fhvae_elbo_bwd_desc {
  fhvae_elbo_desc f;
  /* upstream gradients, (B,) f32, any may be NULL (= 0): */
  const float* g_lower_bound; const float* g_log_px_z; const float* g_neg_kld_z1;
  const float* g_neg_kld_z2; const float* g_log_pmu2;
  int32_t reference_detach;  /* 1: no gradient into x_mu/x_lv and none from log_pmu2 into mu2
                                (the .detach() calls at simple_fhvae.py:107,114) */
  /* outputs, OVERWRITTEN; same layouts as the forward inputs; d_x_mu/d_x_lv may be NULL */
  float* d_x_mu; float* d_x_lv;
  float* d_z1_mu; float* d_z1_lv; float* d_z2_mu; float* d_z2_lv; float* d_mu2;
  /* optional (NULL = off; needs d_x_mu / d_x_lv, F % 4 == 0, F <= 256, 16-byte aligned buffers and time-major rows
     r = t*B + b): the decoder-output gradients once more as ONE bf16 matrix, the operand of the per-frame head's backward
     contractions (fhvae_gauss_head_bwd_pair): d_x_pair_lp[r*ld_pair + 0..F) = d_x_mu, [F..2F) = d_x_lv, [2F..ld_pair) = 0,
     and partial column sums of them, d_x_colsum[fhvae_elbo_colsum_rows(B)][2F] (OVERWRITTEN; the sum over its rows = the
     head's bias gradients) */
  void* d_x_pair_lp; int64_t ld_pair; float* d_x_colsum;
} fhvae_elbo_bwd_desc;

int fhvae_elbo_bwd(const fhvae_elbo_bwd_desc* d, void* stream);
int64_t fhvae_elbo_colsum_rows(int64_t B);

/* ------------------------------------------------------------------------------------------
 * Discriminative loss (K5): logits[b,s] = -sum_d (q[b,d]-table[s,d])^2 * inv_two_var,
 * log-sum-exp over s, cross-entropy against idx  -- simple_fhvae.py:119-122 (the reference
 * materialises (B,S,D) three times; here the table is streamed once per 256-query tile with an
 * online max-subtracted log-sum-exp and nothing of size B*S is ever written).
 *
 * fwd writes per-query partials so that a row-sharded table can be combined across GPUs:
 *   row_max[b], row_sumexp[b] over THIS table's rows, tgt_logit[b] = logit at row idx[b]-row0
 *   (0 if that row is not in [row0, row0+S)), and, if ce_mean != NULL, the single-shard scalar
 *   ce_mean = ce_scale * mean_b( (row_max - tgt_logit) + log(row_sumexp) )  (ce_scale = 1: the reference's log_qy,
 *   simple_fhvae.py:122; -1: the intended objective's -CE without a negation launch each way).
 * ws: workspace of fhvae_disc_lse_ws_bytes(B,S) bytes.
 * dtype (fwd and bwd; q, table and every output stay f32): FHVAE_F32 = the logits in exact f32 (direct form on the VALU, or
 * the expanded form on exact-f32 MFMA for D = 32 and B*S >= 65536) -- the parity mode; FHVAE_BF16 (D = 32 large problems
 * only, otherwise as F32) = the bf16 compute mode: cross terms on bf16 MFMA with hi/lo-split operands (~2^-16 relative on
 * q.t).  In both MFMA forms the query's own row (idx) is taken in the direct f32 form.
 * ------------------------------------------------------------------------------------------ */
int64_t fhvae_disc_lse_ws_bytes(int64_t B, int64_t S);
int fhvae_disc_lse_fwd(const float* q, const float* table, const int64_t* idx, int64_t row0,
                       float inv_two_var, float* row_max, float* row_sumexp, float* tgt_logit,
                       float* ce_mean, float ce_scale, void* ws, int64_t B, int64_t S, int64_t D, int dtype, void* stream);
/* Helpers of the row-sharded table's exchange (the reference has no distributed code): one launch each.
 * pack / unpack: [q[b, 0..D) | int32 bits of idx[b]] rows of D+1 floats -- queries and row indices travel in one all-gather.
 * merge_partials: the W ranks' K5 partials, parts[w] = [row_max | row_sumexp | tgt_logit] of N queries each, merged into the
 *   global (row_max, row_sumexp, tgt_logit): max, sum of sumexp_w * exp(max_w - max), sum (log-sum-exp of simple_fhvae.py:122).
 * bwd_pack / bwd_unpack: the backward's single all-reduce buffer, N x 2D: [dq_all * dq_scale | dmu2 of the n_own local queries
 *   from row own0 on, zeros elsewhere]; unpack returns the local queries' dq and every query's dmu2 (either may be NULL). */
int fhvae_shard_pack(const float* q, const int64_t* idx, float* out, int64_t B, int64_t D, void* stream);
int fhvae_shard_unpack(const float* packed, float* q, int64_t* idx, int64_t N, int64_t D, void* stream);
int fhvae_disc_merge_partials(const float* parts, float* row_max, float* row_sumexp, float* tgt_logit, int64_t W, int64_t N,
                              void* stream);
int fhvae_shard_bwd_pack(const float* dq_all, float dq_scale, const float* dmu2_local, int64_t own0, int64_t n_own, float* out,
                         int64_t N, int64_t D, void* stream);
int fhvae_shard_bwd_unpack(const float* buf, int64_t own0, int64_t n_own, float* dq_local, float* dmu2_all, int64_t N, int64_t D,
                           void* stream);
int fhvae_disc_ce_mean(const float* row_max, const float* row_sumexp, const float* tgt_logit,
                       float* ce_mean, float ce_scale, int64_t B, void* stream);

/* bwd: given the GLOBAL (all shards combined) row_max[b] and row_sumexp[b] and the scalar scale
 * g = (*g_scale) * g_mul  (= dL/d(ce_mean) / B_total), computes  p[b,s] = exp(logit - row_max[b]) / row_sumexp[b]
 * (kept as max and sum, not as one log-sum-exp: |max| ~ 1e3 would put its ulp into every p),
 * w = g*(p - [s == idx[b]-row0])
 *   dq[b,:]     = sum_s w * (-2 c)(q[b]-t[s])      OVERWRITTEN  (partial over this shard's rows)
 *   dtable[s,:] += sum_b w * (+2 c)(q[b]-t[s])     ACCUMULATED
 * g is read from device memory (g_scale, one f32) so the call stays graph-capturable.
 * ws / ws_bytes: workspace (16-byte aligned) or NULL / 0.  With it (and dq and dtable both wanted) the kernels that have a one-pass
 * form take both gradients from ONE recomputation of the logits; without it, one pass per gradient.  The one-pass form keeps
 * (B/256) x S x (D+1) floats of partial sums: fhvae_disc_lse_bwd_ws_bytes(B,S,D) is the RECOMMENDED size, capped at 1.5 GiB (the
 * partials of B = 16384 queries against 10^6 rows would be 8.4 GB); with fewer bytes than the whole problem needs the queries
 * are processed in groups of as many 256-query tiles as fit (any size from one tile's partials up works; below that the
 * call takes the two-pass form).  0 from the size function = no one-pass form for this shape. */
int64_t fhvae_disc_lse_bwd_ws_bytes(int64_t B, int64_t S, int64_t D);
int fhvae_disc_lse_bwd(const float* q, const float* table, const int64_t* idx, int64_t row0,
                       float inv_two_var, const float* row_max, const float* row_sumexp,
                       const float* g_scale, float g_mul, float* dq, float* dtable, void* ws, int64_t ws_bytes,
                       int64_t B, int64_t S, int64_t D, int dtype, void* stream);

/* The discriminative segment variational lower bound, train_model.py:243-251:
 *   loss = -mean_b(lower_bound[b] + alpha * log_qy) = -(mean(lower_bound) + alpha * log_qy)   (log_qy one f32 on the device)
 * and its backward d_lower_bound[b] = -g/B, d_log_qy = -alpha*g (g = *g_loss, NULL = 1): one launch each instead of the
 * ~10 elementwise/reduction launches of the expression.  nan_flag (optional int32 device word, never cleared here): bit 0 is
 * set when mean(lower_bound) is NaN -- the loop's divergence test `torch.isnan(lower_bound).any()` (train_model.py:464-466)
 * without a host synchronisation per batch; the host reads it once per epoch. */
int fhvae_loss_fwd(const float* lower_bound, const float* log_qy, float alpha, float* loss, int64_t B,
                   int32_t* nan_flag, void* stream);
int fhvae_loss_bwd(const float* g_loss, float alpha, float* d_lower_bound, float* d_log_qy, int64_t B,
                   void* stream);

/* ------------------------------------------------------------------------------------------
 * Adam (train_model.py:409-411: torch.optim.Adam(lr, betas=(beta_one, beta_two)), eps 1e-8, no
 * weight decay) over one flat f32 buffer; step_count is read from device memory (int32; already
 * incremented by the caller unless FHVAE_ADAM_ADVANCE) so a captured graph advances the bias correction.  p_lp (optional)
 * receives the updated parameters in bf16.  grad_scale multiplies g first (1/world for DP).
 * ------------------------------------------------------------------------------------------ */
int fhvae_adam_step(float* p, float* g, float* m, float* v, void* p_lp, int64_t n, float lr,
                    float beta1, float beta2, float eps, float grad_scale, int flags, int32_t* step_count,
                    void* stream);
#define FHVAE_ADAM_ZERO_GRAD 1 /* g is cleared behind its use: the next backward accumulates into zeros, no memset launch */
#define FHVAE_ADAM_ADVANCE 2   /* the launch counts the step itself: it uses step_count[0] + 1 and stores it; the words behind
                                  it are its arrival counters (one per 128-byte line, zero between launches): step_count is
                                  int32[FHVAE_ADAM_STEP_WORDS], 128-byte aligned -- no increment launch in front of it */
#define FHVAE_ADAM_STEP_WORDS (65 * 32)

/* ------------------------------------------------------------------------------------------
 * Measurement aid (bench.py roofline leg; no reference counterpart): while enabled, every step-cell launch
 * of fhvae_lstm_seq_fwd/bwd is bracketed by two HIP events on its stream and tagged with its kind
 * (0 = forward cell, 1 = backward cell) and its algorithmic FLOPs.  fhvae_trace_collect synchronises on
 * the recorded events, returns the number of records copied (host arrays, any may be NULL) and clears the
 * trace.  Off by default; the only process-global state of the library; not for use under graph capture.
 * ------------------------------------------------------------------------------------------ */
int fhvae_trace_enable(int on);
int64_t fhvae_trace_collect(float* ms, int32_t* kind, double* flops, int64_t cap);

/* ------------------------------------------------------------------------------------------
 * SURVEY 8f "next" #1 -- segment sampler over an utterance pool resident in HBM.
 * pool: (pool_frames, F) f32, all utterances of a split concatenated (288 GB of HBM hold ~900 M frames of
 * 80-bin features); start[b]: absolute first frame of segment b in the pool (= utterance offset + seg.start,
 * datasets.py:155-185).  Writes the (B,T,F) batch-major batch the reference's DataLoader would collate
 * (datasets.py:214-223) and/or its time-major (T,B,F) form, with (x - mean) * inv_std applied per feature
 * when mean/inv_std are given (apply_mvn, datasets.py:100-105).  Frames outside the pool read as 0 and set
 * *oob_flag (int32 device word, may be NULL).
 * ------------------------------------------------------------------------------------------ */
int fhvae_segment_gather(const float* pool, int64_t pool_frames, const int64_t* start, const float* mean,
                         const float* inv_std, float* out_btf, float* out_tbf, int64_t B, int64_t T,
                         int64_t F, int32_t* oob_flag, void* stream);

/* ------------------------------------------------------------------------------------------
 * SURVEY 8f "next" #2 -- closed-form mu2 estimate (utils.estimate_mu2_dict, utils.py:45-60):
 *   mu2[y] = sum_{segments n with idx[n] == y} z2_mu[n] / (count[y] + ratio),  ratio = exp(pz2_logvar)/exp(pmu2_logvar)
 * accumulate: zsum[idx[n],:] += z2_mu[n,:], count[idx[n]] += 1 (float atomics; call once per batch, buffers
 * zeroed by the caller); finalize: mu2 = zsum / (count + ratio), 0 for sequences without segments.
 * ------------------------------------------------------------------------------------------ */
int fhvae_mu2_accumulate(const float* z2_mu, const int64_t* idx, float* zsum, float* count, int64_t N,
                         int64_t S, int64_t D, void* stream);
int fhvae_mu2_finalize(const float* zsum, const float* count, float* mu2, int64_t S, int64_t D,
                       float ratio, void* stream);

/* small utilities used by the host side */
/* (B,T,F) batch-major f32 -> (T,B,F) time-major in operand dtype `dtype` (and optionally f32) */
int fhvae_to_time_major(const float* x_btf, void* x_tbf, float* x_tbf_f32, int64_t B, int64_t T,
                        int64_t F, int dtype, void* stream);
/* f32 -> bf16 cast, optional transposed copy: src [R,C] -> dst [R,C] and dst_t [C,R] (either NULL) */
int fhvae_cast_bf16(const float* src, void* dst, void* dst_t, int64_t R, int64_t C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FHVAE_HIP_H */
