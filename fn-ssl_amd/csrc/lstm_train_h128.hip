// Training LSTM kernels (forward with reserve, BPTT) for hidden size 128; see lstm_train.h.
#include "lstm_train.h"

namespace fnssl_lstm {
template int launch_bwd<128>(int, int, const BwdParams&, int, hipStream_t);
template int launch_save<128>(int, int, const LstmParams&, int, int, hipStream_t);
}  // namespace fnssl_lstm
