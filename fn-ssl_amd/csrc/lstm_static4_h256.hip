// Operand-ring narrow-band kernels, four hidden slices per pass (lstm_static4.h): H = 256, 12 waves per workgroup, 256 (+ 4)
// input channels.  Ring: 3-quad chunks of 48 records (48 KiB per slot, 96 KiB of LDS like lstm_static3_kernel's), four DMA
// instructions per wave and chunk; blocks 2 - 3: 33 quads per slice = 11 chunks; block 1: 34 quads + 2 of padding = 12.
#include "lstm_static4.h"

namespace fnssl_lstm {

int launch_static4_h256(const LstmParams& p_in, int mode, int nwg, hipStream_t st) {
  LstmParams p = p_in;
#ifdef FNSSL_BUILD_ABLATE
  p.ablate = env_int("FNSSL_STATIC3_ABL", 1, 3);
#endif
  // PF4 (cell state / residual operand of all four slices requested at the start of the pass's last chunk): 109.4 -> 108.7 ms per
  // launch at config 2 (profiles/r06/); block 1's variant has no registers for it (166 of 168)
  if (p.c0 == 256 && p.c2 == 0 && mode == kSum) return launch_static4_k<256, 12, 16, 0, 3, 0, kSum, 4, true>(p, nwg, st);
  if (p.c0 == 256 && p.c2 == 0 && mode == 0) return launch_static4_k<256, 12, 16, 0, 3, 0, 0, 4, true>(p, nwg, st);
  if (p.c0 == 256 && p.c2 == 4 && mode == (kHas2 | kSum))   // block 1: 256 + 4 channels
    return launch_static4_k<256, 12, 16, 1, 3, 2, kHas2 | kSum>(p, nwg, st);
  return kNoStatic;
}

}  // namespace fnssl_lstm
