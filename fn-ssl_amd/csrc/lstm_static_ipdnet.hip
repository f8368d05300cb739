// Shape-specialised LSTM kernels for the IPDnet layers (concatenated 16-channel input skip, 8 mics):
// narrow-band H = 256 reading [full-band out (256) | input (16)], full-band H = 128 reading
// [narrow-band out (256) | input (16)].  See lstm_static.h.  At the batch sizes IPDnet runs with
// (64 utterances -> 1024 narrow-band wave tasks = ONE wave per SIMD) nothing hides latency behind
// other waves, so the 4-wave entry uses a deep weight ring (12 staging registers per wave, 10-quad
// chunks) and an 8-block operand prefetch ring.
#include "lstm_static.h"

namespace fnssl_lstm {

// c0 = 16 * NV0 (summed input), c2 = 16 * NV2 (concatenated input, packed as 16-channel blocks)
#define TRYH(H_, NW_, M_, NV0_, NV2_, CHQ_, PAD_, XD_)                                                   \
  if (H == H_ && NW == NW_ && p.c0 == 16 * NV0_ && p.c2 == 16 * NV2_ && mode == kHas2)                   \
    return launch_static_k<H_, NW_, M_, NV0_, 0, 0, CHQ_, PAD_, kHas2, false, XD_, NV2_>(p, nwg, st);

int launch_static_ipdnet(const LstmParams& p, int mode, int H, int NW, int nwg, hipStream_t st) {
  if (p.ablate) return kNoStatic;
  // narrow-band: 1 + 16 + 1 + 16 = 34 quads per slice
  TRYH(256, 4, 12, 16, 1, 12, 2, 8)
  TRYH(256, 8, 4, 16, 1, 7, 1, 4)
  TRYH(256, 12, 4, 16, 1, 12, 2, 4)
  // full-band of block 2: 1 + 16 + 1 + 8 = 26 quads per slice
  TRYH(128, 12, 4, 16, 1, 9, 1, 4)
  TRYH(128, 16, 4, 16, 1, 13, 0, 4)
  return kNoStatic;
}

}  // namespace fnssl_lstm
