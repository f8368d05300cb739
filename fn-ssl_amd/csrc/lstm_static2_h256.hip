// Two-slices-per-pass narrow-band kernels (lstm_static2.h): H = 256, 12 waves per workgroup, 256 input channels.
#include "lstm_static2.h"

namespace fnssl_lstm {

int launch_static2_h256(const LstmParams& p, int mode, int nwg, hipStream_t st) {
  // 33 pair-quads per slice pair + 3 padding = 6 chunks of 6 (48 ring barriers per step, as with single slices)
  // (a 2-deep x ring frees 8 registers but makes the allocator spill 30 in the kSum instantiation; 4-deep: 2 / 1 spills)
#ifdef FNSSL_BUILD_ABLATE   // timing-ablation twin (wrong results by construction): only in `make ABLATE=1` builds
  if (p.ablate && p.c0 == 256 && p.c2 == 0 && mode == kSum) return launch_static2_k<256, 12, 4, 16, 0, 6, 3, kSum, 4, true>(p, nwg, st);
#endif
  if (p.c0 == 256 && p.c2 == 0 && mode == kSum) return launch_static2_k<256, 12, 4, 16, 0, 6, 3, kSum>(p, nwg, st);
  if (p.c0 == 256 && p.c2 == 0 && mode == 0) return launch_static2_k<256, 12, 4, 16, 0, 6, 3, 0>(p, nwg, st);
  return kNoStatic;
}

}  // namespace fnssl_lstm
