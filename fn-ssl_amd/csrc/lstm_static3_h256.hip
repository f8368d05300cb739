// Operand-ring narrow-band kernels (lstm_static3.h): H = 256, 12 waves per workgroup, 256 (+ 4) input channels.
#include "lstm_static3.h"

namespace fnssl_lstm {

int launch_static3_h256(const LstmParams& p_in, int mode, int nwg, hipStream_t st) {
  LstmParams p = p_in;
#ifdef FNSSL_BUILD_ABLATE
  p.ablate = env_int("FNSSL_STATIC3_ABL", 1, 3);
#endif
  // Ring geometry of lstm_static2_kernel: 6-quad chunks, 4 staging registers per wave, 48 barriers per step.  (9-quad
  // chunks — 6 staging registers, 32 barriers per step — were built and measured: 111.1 against 110.3 ms with the fused
  // residual, 108.2 against 107.5 without, 112.1 against 110.7 for block 1's layer: fewer barriers, longer skew; not kept.)
  if (p.c0 == 256 && p.c2 == 0 && mode == kSum) return launch_static3_k<256, 12, 4, 16, 0, 6, 3, kSum>(p, nwg, st);
  if (p.c0 == 256 && p.c2 == 0 && mode == 0) return launch_static3_k<256, 12, 4, 16, 0, 6, 3, 0>(p, nwg, st);
  if (p.c0 == 256 && p.c2 == 4 && mode == (kHas2 | kSum))   // block 1: 256 + 4 channels, 34 pair-quads (+ 2 padding)
    return launch_static3_k<256, 12, 4, 16, 1, 6, 2, kHas2 | kSum>(p, nwg, st);
  return kNoStatic;
}

}  // namespace fnssl_lstm
