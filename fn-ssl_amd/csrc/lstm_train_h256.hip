// Training LSTM kernels (forward with reserve, BPTT) for hidden size 256; see lstm_train.h.
#include "lstm_train.h"
#include "lstm_bwd2.h"
#include "lstm_fwd2.h"

namespace fnssl_lstm {
template int launch_bwd<256>(int, int, const BwdParams&, int, hipStream_t);
template int launch_save<256>(int, int, const LstmParams&, int, int, hipStream_t);
template int launch_bwd2_k<256>(const BwdParams&, int, hipStream_t);
template int launch_fwd2_k<256, 16, 0>(const LstmParams&, int, hipStream_t);
template int launch_fwd2_k<256, 16, 1>(const LstmParams&, int, hipStream_t);
}  // namespace fnssl_lstm
