// lstm_cluster.h -- persistent "cluster" form of the bf16 LSTM recurrence (lstm_cluster.hip), used by lstm.hip.
#pragma once
#include "common.h"

namespace fh {

// The first FHVAE_LSTM_SYNC_BYTES of the bf16 workspace (fhvae_lstm_desc.lp) are the kernels' sync block (u32 words):
constexpr int kSyncStatus = 0;    // 0 = ok; bit 0: a bounded spin gave up, bit 1: a workgroup could not read its XCD
constexpr int kSyncSticky = 2;    // two words: address of fhvae_lstm_desc.sticky_status (0 = none), written when the block is armed
constexpr int kSyncXcdCnt = 16;   // 8 arrival counters (one per XCD): ticket & 31 = a workgroup's slot on its XCD, ticket >> 5 = the launch
constexpr int kSyncFlags = 64;    // + cluster * 32: one word per workgroup of the cluster = the last step it has published
constexpr int kSyncWordsUsed = 3072;  // the words the operand cast re-arms: the cluster kernels' flags end at kSyncFlags + 64 * 32; the last
                                      // quarter of the block (from byte 12288) is the phase-clock log of the profiling tools
// The block is zeroed once per forward (by the operand-cast launch that precedes every bf16 forward); the launches that then
// share it -- forward chunks, later the backward's, however often it runs -- number themselves: every launch takes 32 tickets
// of each XCD counter, so launch n holds the tickets [32 n, 32 n + 32) and uses the flag epochs (n * kSeqEpochs, (n + 1) *
// kSeqEpochs].  Nothing is re-armed between the launches (and nothing depends on the host counting them: graph replay repeats
// the same sequence from the same zeroed block).
constexpr int kSeqEpochs = 4096;

// bf16 elements of the exchange buffer the partial-dh backward (lstm_bwd_rs.hip) needs: 2 parities x 64 clusters x 4 sources x
// 3 destinations x 4 waves x 2 row tiles x 512 B (bf16 partials)
constexpr int64_t kRsXchElems = 2LL * 64 * 4 * 3 * 4 * 2 * 256;

struct ClusterWeights {  // bf16 operand copies in the workspace
  const u16* w_ih[FHVAE_MAX_LAYERS];    // [4H, H]   (l >= 1)
  const u16* w_hh[FHVAE_MAX_LAYERS];    // [4H, H]
  const u16* w_ih_t[FHVAE_MAX_LAYERS];  // [H, 4H]   (l >= 1)
  const u16* w_hh_t[FHVAE_MAX_LAYERS];  // [H, 4H]
  u16* xch;                             // exchange buffer: 2 * L * B * 4H bf16 (lstm_cluster.hip, xch_off)
  const u16* x_fold;                    // (T,B,I) bf16 when the forward kernels do the layer-0 input projection themselves
  const u16* xc_fold;                   // (B,Ic) bf16 when the rows-form forward kernel projects the time-constant input itself
};

// whether this device / shape can run the cluster kernels (gfx950 with 256 CUs, bf16, H in {128, 256}, L <= 2, ...)
bool cluster_eligible(const fhvae_lstm_desc* d);
// 1: the waves of a workgroup split the cluster's rows, 2: they split the contraction (<= 32 rows per cluster)
int cluster_form(const fhvae_lstm_desc* d);
// the forward kernels can multiply x_t by W_ih[0][:, :I] themselves (I a multiple of 8, at most 128, rows 16-byte aligned)
bool cluster_can_fold(const fhvae_lstm_desc* d);
bool cluster_xc_in_kernel(const fhvae_lstm_desc* d);
// the forward runs with register-stationary weights and saves the gates unit-major (lstm_fwd_wr.hip)
bool cluster_fwd_wr_ok(const fhvae_lstm_desc* d);
// the recurrence of fhvae_lstm_seq_fwd after the layer-0 input projection (d->pre filled): all T steps, all layers
int cluster_fwd(const fhvae_lstm_desc* d, const ClusterWeights& w, hipStream_t st);
// the backward needs fhvae_lstm_bwd_desc.ws_below
bool cluster_needs_ws_below(const fhvae_lstm_desc* d);
// the backward recurrence leaves bd->d_xc zeroed (lstm_bwd_rs.hip): lstm.hip's split-K contraction into it needs no zeroing launch
bool cluster_bwd_zeroes_dxc(const fhvae_lstm_desc* d);
// the recurrence of fhvae_lstm_seq_bwd: fills dgates (and dgsum when Ic > 0)
int cluster_bwd(const fhvae_lstm_bwd_desc* bd, const ClusterWeights& w, hipStream_t st);

}  // namespace fh
