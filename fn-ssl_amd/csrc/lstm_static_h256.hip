// Shape-specialised LSTM kernels, hidden size 256 (online narrow-band LSTM).  See lstm_static.h.
#include "lstm_static.h"

namespace fnssl_lstm {

#define TRY(NW_, M_, NV0_, NS0_, NS2_, CHQ_, PAD_, MODE_)                                              \
  if (NW == NW_ && p.c0 == 16 * NV0_ + 4 * NS0_ && p.c2 == 4 * NS2_ && NS2_ < 4 && mode == (MODE_))                  \
    return launch_static_k<256, NW_, M_, NV0_, NS0_, NS2_, CHQ_, PAD_, MODE_>(p, nwg, st);

int launch_static_h256(const LstmParams& p, int mode, int NW, int nwg, hipStream_t st) {
  // block 1: 256 + 4 channels, 34 quads (+2 pad = 3 x 12)
  TRY(12, 4, 16, 0, 1, 12, 2, kHas2 | kSum)
  // blocks 2/3: 256 channels, 33 quads = 3 x 11
#ifdef FNSSL_BUILD_ABLATE   // timing-ablation twin (wrong results by construction): only in `make ABLATE=1` builds
  if (p.ablate && NW == 12 && p.c0 == 256 && p.c2 == 0 && mode == kSum)
    return launch_static_k<256, 12, 4, 16, 0, 0, 11, 0, kSum, true>(p, nwg, st);
#endif
  TRY(12, 4, 16, 0, 0, 11, 0, kSum)
  TRY(12, 4, 16, 0, 0, 11, 0, 0)
  // smaller launches: 8 / 4 waves per workgroup
  TRY(8, 4, 16, 0, 1, 7, 1, kHas2 | kSum)
  TRY(8, 4, 16, 0, 0, 7, 2, kSum)
  TRY(8, 4, 16, 0, 0, 7, 2, 0)
  TRY(4, 4, 16, 0, 1, 4, 2, kHas2 | kSum)
  TRY(4, 4, 16, 0, 0, 3, 0, kSum)
  TRY(4, 4, 16, 0, 0, 3, 0, 0)
  return kNoStatic;
}

}  // namespace fnssl_lstm
