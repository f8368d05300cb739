// Split-geometry LSTM kernels (few sequences: several waves per 16-sequence group) for hidden sizes 128 and 256;
// see lstm_kernel.h (SPLIT).  Since round 5 the guarded fallback of the cluster-resident kernel at small sizes and the
// path of layouts / operand forms it does not take.
#include "lstm_kernel.h"

namespace fnssl_lstm {
template int launch_split_h<128>(int, const LstmParams&, int, int, hipStream_t);
template int launch_split_h<256>(int, const LstmParams&, int, int, hipStream_t);
}  // namespace fnssl_lstm
