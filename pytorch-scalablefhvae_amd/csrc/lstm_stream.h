// lstm_stream.h -- the large-tile bf16 LSTM cells (lstm_cell.hip) as ONE persistent launch per direction (lstm_stream.hip).
#pragma once
#include "common.h"

namespace fh {

struct StreamWeights {  // bf16 operand copies in the workspace (lstm.hip, lp_layout)
  const u16* x;         // (T,B,I) or NULL
  const u16* xc;        // (B,Ic) or NULL
  const u16* w_ih[FHVAE_MAX_LAYERS];    // [4H, K_l]
  const u16* w_hh[FHVAE_MAX_LAYERS];    // [4H, H]
  const u16* w_ih_t[FHVAE_MAX_LAYERS];  // [H, 4H] (l >= 1)
  const u16* w_hh_t[FHVAE_MAX_LAYERS];  // [H, 4H]
};

// gfx950 with 256 CUs (8 XCDs x 32), bf16, L <= 2, H a multiple of 64 up to 512, B a multiple of 128, I and Ic multiples of 8,
// 16-byte aligned buffers
bool stream_eligible(const fhvae_lstm_desc* d);
int stream_fwd(const fhvae_lstm_desc* d, const StreamWeights& w, hipStream_t st);
// fills bd->dgates (row-major (L,T,B,4H) bf16); bd->dgsum is NOT written (launch_cell_dgsum)
int stream_bwd(const fhvae_lstm_bwd_desc* bd, const StreamWeights& w, hipStream_t st);

}  // namespace fh
