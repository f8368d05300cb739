// Dispatcher of the shape-specialised split kernels, and their H = 128 instantiations (reserve-saving forward at 2 or 4 waves
// per 16-sequence group for the layer shapes of the FN-SSL training step); H = 256: lstm_split_static_h256.hip.
#include "lstm_split_static.h"

namespace fnssl_lstm {

int launch_split_static_h256(const LstmParams& p, int H, int nw, int split, int mode, int max_chq, int nwg, hipStream_t st);

// (H, NW, M, SPLIT, NV0, NS0, NS2, CHQ, PAD): c0 = 16 NV0 + 4 NS0, c2 = 4 NS2; kSave always, kHas2 with NS2
#define TRYS(H_, NW_, M_, S_, NV0_, NS0_, NS2_, CHQ_, PAD_)                                                        \
  if (H == H_ && nw == NW_ && split == S_ && p.c0 == 16 * NV0_ + 4 * NS0_ && p.c2 == 4 * NS2_ && CHQ_ <= max_chq &&  \
      mode == (kSave | (NS2_ ? kHas2 : 0)))                                                                          \
    return launch_split_static_k<H_, NW_, M_, S_, NV0_, NS0_, NS2_, CHQ_, PAD_, kSave | (NS2_ ? kHas2 : 0)>(p, nwg, st);

int launch_split_static_h128(const LstmParams& p, int H, int nw, int split, int mode, int max_chq, int nwg, hipStream_t st) {
  if (max_chq <= 0) max_chq = 1 << 20;
  // H = 128: full-band (25 quads; block 1: 10) and the offline narrow-band layers (25 / 26)
  TRYS(128, 4, 4, 2, 16, 0, 0, 2, 1)
  TRYS(128, 4, 4, 2, 16, 0, 1, 2, 0)
  TRYS(128, 4, 4, 2, 0, 1, 0, 2, 0)
  TRYS(128, 8, 8, 4, 16, 0, 0, 4, 3)
  TRYS(128, 8, 8, 4, 16, 0, 1, 4, 2)
  TRYS(128, 8, 8, 4, 0, 1, 0, 4, 2)
  return kNoStatic;
}

int launch_split_static(const LstmParams& p, int H, int nw, int split, int mode, int max_chq, int nwg, hipStream_t st) {
  if (H == 128) return launch_split_static_h128(p, H, nw, split, mode, max_chq, nwg, st);
  if (H == 256) return launch_split_static_h256(p, H, nw, split, mode, max_chq, nwg, st);
  return kNoStatic;
}

}  // namespace fnssl_lstm
