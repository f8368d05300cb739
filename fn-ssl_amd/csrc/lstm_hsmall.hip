// LSTM recurrence kernels for the small hidden sizes 16 / 32 / 64 (IPDnet with two microphones, tests of every size;
// explicit instantiations, see lstm_kernel.h).  One translation unit since round 5: they are generic-kernel
// instantiations only and together compile in the time lstm_h256.hip takes.
#include "lstm_kernel.h"

namespace fnssl_lstm {
template int launch_h<16>(int, const LstmParams&, int, int, hipStream_t);
template int launch_h<32>(int, const LstmParams&, int, int, hipStream_t);
template int launch_h<64>(int, const LstmParams&, int, int, hipStream_t);
}  // namespace fnssl_lstm
