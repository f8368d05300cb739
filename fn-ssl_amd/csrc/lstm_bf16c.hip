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
  cp.stagger = env_int("FNSSL_CLUSTER_STAGGER", 1, 1000);
  for (int c0 = 0; c0 < ncl; c0 += per_launch) {
    cp.cl0 = c0;
    cp.ncl = ncl - c0 < per_launch ? ncl - c0 : per_launch;
    int rc = FNSSL_OK;
#ifdef FNSSL_BUILD_ABLATE   // timing ablations (wrong results): make ABLATE=1 only
    switch (env_int("FNSSL_CLUSTER_ABL", 1, 4095)) {
      case 1: rc = launch_bf16c_k<256, 16, 1, kW_F2, 1>(p, cp, st); break;
      case 2: rc = launch_bf16c_k<256, 16, 1, kW_F2, 2>(p, cp, st); break;
      case 4: rc = launch_bf16c_k<256, 16, 1, kW_F2, 4>(p, cp, st); break;
      case 6: rc = launch_bf16c_k<256, 16, 1, kW_F2, 6>(p, cp, st); break;
      case 9: rc = launch_bf16c_k<256, 16, 1, kW_F2, 9>(p, cp, st); break;
      case 25: rc = launch_bf16c_k<256, 16, 1, kW_F2, 25>(p, cp, st); break;
      case 128: rc = launch_bf16c_k<256, 16, 1, kW_F2, 128>(p, cp, st); break;
      case 256: rc = launch_bf16c_k<256, 16, 1, kW_F2, 256>(p, cp, st); break;
      case 512: rc = launch_bf16c_k<256, 16, 1, kW_F2, 512>(p, cp, st); break;
      case 1024: rc = launch_bf16c_k<256, 16, 1, kW_F2, 1024>(p, cp, st); break;
      case 2048: rc = launch_bf16c_k<256, 16, 1, kW_F2, 2048>(p, cp, st); break;
      case 32: rc = launch_bf16c_k<256, 16, 1, kW_F2, 32>(p, cp, st); break;
      case 64: rc = launch_bf16c_k<256, 16, 1, kW_F2, 64>(p, cp, st); break;
      case 65: rc = launch_bf16c_k<256, 16, 1, kW_F2, 65>(p, cp, st); break;
      case 97: rc = launch_bf16c_k<256, 16, 1, kW_F2, 97>(p, cp, st); break;
      case 57: rc = launch_bf16c_k<256, 16, 1, kW_F2, 57>(p, cp, st); break;
      case 89: rc = launch_bf16c_k<256, 16, 1, kW_F2, 89>(p, cp, st); break;
      case 121: rc = launch_bf16c_k<256, 16, 1, kW_F2, 121>(p, cp, st); break;
      case 123: rc = launch_bf16c_k<256, 16, 1, kW_F2, 123>(p, cp, st); break;
      default: rc = launch_bf16c_k<256, 16, 1, kW_F2>(p, cp, st); break;
    }
#else
    rc = launch_bf16c_k<256, 16, 1, kW_F2>(p, cp, st);
#endif
    if (rc != FNSSL_OK) return rc;
  }
  return FNSSL_OK;
}

}  // namespace fnssl_lstm
