// Shape-specialised LSTM kernels, hidden size 128 (full-band BiLSTM of every block and the
// offline narrow-band BiLSTM).  See lstm_static.h.
#include "lstm_static.h"

namespace fnssl_lstm {

#define TRY(NW_, M_, NV0_, NS0_, NS2_, CHQ_, PAD_, MODE_)                                              \
  if (NW == NW_ && p.c0 == 16 * NV0_ + 4 * NS0_ && p.c2 == 4 * NS2_ && NS2_ < 4 && mode == (MODE_))                  \
    return launch_static_k<128, NW_, M_, NV0_, NS0_, NS2_, CHQ_, PAD_, MODE_>(p, nwg, st);

int launch_static_h128(const LstmParams& p, int mode, int NW, int nwg, hipStream_t st) {
  // block 1 full-band: 4 input channels, 10 quads per slice
  TRY(16, 4, 0, 1, 0, 10, 0, 0)
  TRY(15, 4, 0, 1, 0, 10, 0, 0)
  TRY(14, 4, 0, 1, 0, 10, 0, 0)
  TRY(13, 4, 0, 1, 0, 10, 0, 0)
  // blocks 2/3 full-band (and offline narrow-band 2/3): 256 channels, 25 quads (+3 pad = 4 x 7)
#ifdef FNSSL_BUILD_ABLATE   // timing-ablation twin (wrong results by construction): only in `make ABLATE=1` builds
  if (p.ablate && NW == 16 && p.c0 == 256 && p.c2 == 0 && mode == kSum)
    return launch_static_k<128, 16, 2, 16, 0, 0, 7, 3, kSum, true>(p, nwg, st);
#endif
  // 26 virtual quads = 2 chunks of 13
  TRY(16, 4, 16, 0, 0, 13, 1, kSum)
  TRY(15, 4, 16, 0, 0, 13, 1, kSum)
  TRY(14, 4, 16, 0, 0, 13, 1, kSum)
  TRY(16, 4, 16, 0, 0, 13, 1, 0)
  TRY(15, 4, 16, 0, 0, 13, 1, 0)
  TRY(14, 4, 16, 0, 0, 13, 1, 0)
  // offline narrow-band block 1: 256 + 4 channels, 26 quads (+2 pad = 4 x 7)
  TRY(16, 2, 16, 0, 1, 7, 2, kHas2 | kSum)
  TRY(12, 4, 16, 0, 1, 9, 1, kHas2 | kSum)
  TRY(8, 4, 16, 0, 1, 7, 2, kHas2 | kSum)
  // smaller launches (fewer than 14 waves per CU): 13 / 12 / 8 / 4 waves per workgroup.  12 waves: 27 virtual quads = 3
  // chunks of 9 (one full round of 3 000 groups: 36.8 ms, against 37.1 ms with 5 chunks of 5 — and 48.9 ms for a round
  // of 16-wave workgroups: a round is paced by its fullest SIMD, 3 waves against 4)
  TRY(13, 4, 16, 0, 0, 13, 1, kSum)
  TRY(13, 4, 16, 0, 0, 13, 1, 0)
  TRY(12, 4, 16, 0, 0, 9, 2, kSum)
  TRY(12, 4, 16, 0, 0, 9, 2, 0)
  TRY(12, 4, 0, 1, 0, 10, 0, 0)
  TRY(8, 4, 16, 0, 0, 7, 3, kSum)
  TRY(8, 4, 16, 0, 0, 7, 3, 0)
  TRY(8, 4, 0, 1, 0, 5, 0, 0)
  TRY(4, 4, 16, 0, 0, 4, 3, kSum)
  TRY(4, 4, 16, 0, 0, 4, 3, 0)
  TRY(4, 4, 0, 1, 0, 4, 2, 0)
  return kNoStatic;
}

}  // namespace fnssl_lstm
