// Cluster-resident bf16 LSTM kernel (lstm_bf16c.h): instantiation for IPDnet's narrow-band layer shape and its launcher.
#include <cstdlib>

#include "lstm_bf16c.h"

namespace fnssl_lstm {

// true when the cluster kernel is built for the shape (and not switched off: FNSSL_NO_CLUSTER=1 keeps the pair-split
// kernels of lstm_bf16p.h, which give the same bits — A/B and fallback for shapes not built here)
bool bf16c_handles(const LstmParams& p, int H, int flags) {
  if (getenv("FNSSL_NO_CLUSTER")) return false;
  return H == 256 && p.c0 == 256 && p.c2 == 16 && flags == kW_F2 && p.nseq >= kClusterSeqs;
}

// Launches of at most CUs / 8 clusters (one workgroup per CU: every member of every cluster of a launch is resident).
int forward_bf16c(LstmParams p, int H, int flags, hipStream_t st) {
  (void)H;
  (void)flags;
  const int ncu = fnssl::device_cus();
  const int per_launch = ncu / kClusterMembers;
  FNSSL_REQUIRE(per_launch >= 1, "lstm_forward: the cluster kernel needs at least 8 CUs");
  const int cl_per_dir = (p.nseq + kClusterSeqs - 1) / kClusterSeqs;
  const int ncl = cl_per_dir * p.ndir;
  ClusterParams cp;
  cp.status = reinterpret_cast<unsigned*>(p.cluster_ws);
  cp.tags = reinterpret_cast<unsigned*>(p.cluster_ws + 256);
  cp.hx = p.cluster_ws + 256 + (size_t)ncl * (kClusterTagWords * 4);
  cp.parity_stride = (unsigned)((size_t)ncl * kClusterParityBytes);
  cp.cl_per_dir = cl_per_dir;
  FNSSL_REQUIRE((size_t)ncl * kClusterHxBytes < 0xf0000000ull, "lstm_forward: too many sequences for one call of the cluster kernel");
  // status word + tags, and the parity-1 operand records (step 0 reads h_{-1} = 0 from them)
  FNSSL_HIP(hipMemsetAsync(p.cluster_ws, 0, 256 + (size_t)ncl * (kClusterTagWords * 4), st));
  FNSSL_HIP(hipMemsetAsync(cp.hx + cp.parity_stride, 0, cp.parity_stride, st));
  for (int c0 = 0; c0 < ncl; c0 += per_launch) {
    cp.cl0 = c0;
    cp.ncl = ncl - c0 < per_launch ? ncl - c0 : per_launch;
    const int rc = launch_bf16c_k<256, 16, 1, kW_F2>(p, cp, st);
    if (rc != FNSSL_OK) return rc;
  }
  return FNSSL_OK;
}

}  // namespace fnssl_lstm
