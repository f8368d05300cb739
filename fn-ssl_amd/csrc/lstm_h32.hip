// LSTM recurrence kernels for hidden size 32 (explicit instantiation; see lstm_kernel.h).
#include "lstm_kernel.h"

namespace fnssl_lstm {
template int launch_h<32>(int, const LstmParams&, int, int, hipStream_t);
}  // namespace fnssl_lstm
