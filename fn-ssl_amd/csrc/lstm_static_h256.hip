// Shape-specialised LSTM kernels, hidden size 256 (online narrow-band LSTM).  See lstm_static.h.
#include "lstm_static.h"

namespace fnssl_lstm {

#define TRY(NW_, M_, NV0_, NS0_, NS2_, CHQ_, PAD_, MODE_)                                              \
  if (NW == NW_ && p.c0 == 16 * NV0_ + 4 * NS0_ && p.c2 == 4 * NS2_ && NS2_ < 4 && mode == (MODE_))                  \
    return launch_static_k<256, NW_, M_, NV0_, NS0_, NS2_, CHQ_, PAD_, MODE_>(p, nwg, st);

#define TRYX(NW_, M_, NV0_, NS0_, NS2_, CHQ_, PAD_, MODE_, XD_)                                        \
  if (NW == NW_ && p.c0 == 16 * NV0_ + 4 * NS0_ && p.c2 == 4 * NS2_ && NS2_ < 4 && mode == (MODE_))                \
    return launch_static_k<256, NW_, M_, NV0_, NS0_, NS2_, CHQ_, PAD_, MODE_, false, XD_>(p, nwg, st);

#define TRYS(NW_, M_, NV0_, NS0_, NS2_, CHQ_, PAD_, MODE_)                                             \
  if (NW == NW_ && p.c0 == 16 * NV0_ + 4 * NS0_ && p.c2 == 4 * NS2_ && NS2_ < 4 && mode == (MODE_))                \
    return launch_static_k<256, NW_, M_, NV0_, NS0_, NS2_, CHQ_, PAD_, MODE_, false, 4, true>(p, nwg, st);

int launch_static_h256(const LstmParams& p, int mode, int NW, int nwg, hipStream_t st) {
  if (!p.ablate && env_int("FNSSL_STATIC_XD8", 1, 1)) {   // experiment: 8-deep x ring (no gain measured)
    TRYX(12, 4, 16, 0, 0, 11, 0, kSum, 8)
    TRYX(12, 4, 16, 0, 0, 11, 0, 0, 8)
  }
  if (!p.ablate && env_int("FNSSL_STATIC_STAG", 1, 1)) {   // experiment (slower, r01): two wave groups one chunk apart
    TRYS(12, 4, 16, 0, 1, 12, 2, kHas2 | kSum)
    TRYS(12, 4, 16, 0, 0, 11, 0, kSum)
    TRYS(12, 4, 16, 0, 0, 11, 0, 0)
  }
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
