// (H = 256) Instantiations of lstm_split_static_kernel for the layer shapes of the FN-SSL training step (reserve-saving
// forward at 2 or 4 waves per 16-sequence group).
#include "lstm_split_static.h"
#include "tuning.h"

namespace fnssl_lstm {

// (H, NW, M, SPLIT, NV0, NS0, NS2, CHQ, PAD): c0 = 16 NV0 + 4 NS0, c2 = 4 NS2; kSave always, kHas2 with NS2
#define TRYS(H_, NW_, M_, S_, NV0_, NS0_, NS2_, CHQ_, PAD_)                                                        \
  if (H == H_ && nw == NW_ && split == S_ && p.c0 == 16 * NV0_ + 4 * NS0_ && p.c2 == 4 * NS2_ && CHQ_ <= max_chq &&  \
      mode == (kSave | (NS2_ ? kHas2 : 0)))                                                                          \
    return launch_split_static_k<H_, NW_, M_, S_, NV0_, NS0_, NS2_, CHQ_, PAD_, kSave | (NS2_ ? kHas2 : 0)>(p, nwg, st);

// ring-free variants (4 waves per group): weights straight from L2 through the register pipeline
#define TRYD(H_, NW_, S_, NV0_, NS0_, NS2_)                                                                        \
  if (H == H_ && nw == NW_ && split == S_ && p.c0 == 16 * NV0_ + 4 * NS0_ && p.c2 == 4 * NS2_ &&                     \
      mode == (kSave | (NS2_ ? kHas2 : 0)))                                                                          \
    return launch_split_static_k<H_, NW_, 4, S_, NV0_, NS0_, NS2_, 1, 0, kSave | (NS2_ ? kHas2 : 0), true>(p, nwg, st);

// inference (fused forward: residual output for the next layer), 4 waves per group, ring-free
#define TRYI(NS2_, MODE_)                                                                                   \
  if (H == 256 && nw == 8 && split == 4 && p.c0 == 256 && p.c2 == 4 * NS2_ && mode == (MODE_))              \
    return launch_split_static_k<256, 8, 4, 4, 16, 0, NS2_, 1, 0, MODE_, true>(p, nwg, st);

int launch_split_static_h256(const LstmParams& p, int H, int nw, int split, int mode, int max_chq, int nwg, hipStream_t st) {
  if (max_chq <= 0) max_chq = 1 << 20;
  if (!fnssl::tune(FNSSL_TUNE_FWD_RING)) {
    TRYI(0, kSum)
    TRYI(1, kHas2 | kSum)
    TRYI(0, 0)
    TRYI(1, kHas2)
    // IPDnet narrow-band: [256 | 16-channel concatenated block]
    if (H == 256 && nw == 8 && split == 4 && p.c0 == 256 && p.c2 == 16 && mode == kHas2)
      return launch_split_static_k<256, 8, 4, 4, 16, 0, 0, 1, 0, kHas2, true, 1>(p, nwg, st);
    TRYD(256, 8, 4, 16, 0, 0)
    TRYD(256, 8, 4, 16, 0, 1)
  }
  // narrow-band H = 256: 33 quads per slice (34 with the 4 data channels of block 1)
  TRYS(256, 8, 8, 4, 16, 0, 0, 4, 3)
  TRYS(256, 8, 8, 4, 16, 0, 1, 4, 2)
  TRYS(256, 8, 8, 4, 16, 0, 0, 2, 1)
  TRYS(256, 8, 8, 4, 16, 0, 1, 2, 0)
  TRYS(256, 4, 4, 2, 16, 0, 0, 2, 1)
  TRYS(256, 4, 4, 2, 16, 0, 1, 2, 0)
  // H = 128: full-band (25 quads; block 1: 10) and the offline narrow-band layers (25 / 26)
  return kNoStatic;
}

}  // namespace fnssl_lstm
