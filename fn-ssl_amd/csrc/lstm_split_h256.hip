// Split-geometry LSTM kernels (few sequences: several waves per 16-sequence group) for hidden size 256;
// see lstm_kernel.h (SPLIT).
#include "lstm_kernel.h"

namespace fnssl_lstm {
template int launch_split_h<256>(int, const LstmParams&, int, int, hipStream_t);
}  // namespace fnssl_lstm
