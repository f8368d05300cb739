// Dispatcher of the shape-specialised split kernels (instantiations: lstm_split_static_h128.hip / _h256.hip).
#include "lstm_split_static.h"

namespace fnssl_lstm {

int launch_split_static_h128(const LstmParams& p, int H, int nw, int split, int mode, int max_chq, int nwg, hipStream_t st);
int launch_split_static_h256(const LstmParams& p, int H, int nw, int split, int mode, int max_chq, int nwg, hipStream_t st);

int launch_split_static(const LstmParams& p, int H, int nw, int split, int mode, int max_chq, int nwg, hipStream_t st) {
  if (H == 128) return launch_split_static_h128(p, H, nw, split, mode, max_chq, nwg, st);
  if (H == 256) return launch_split_static_h256(p, H, nw, split, mode, max_chq, nwg, st);
  return kNoStatic;
}

}  // namespace fnssl_lstm
