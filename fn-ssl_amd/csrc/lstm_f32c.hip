// Cluster-resident fp32 LSTM kernel (lstm_f32c.h): instantiations for FN-SSL's H = 128 full-band layers and the launcher.
#include <cstdlib>

#include "lstm_f32c.h"
#include "tuning.h"

namespace fnssl_lstm {

// Full-chip launches of the two full-band shapes of the network (block 1: 4 input channels; blocks 2-3: 256, fused
// residual output).  FNSSL_NO_F32_CLUSTER=1 keeps lstm_static_kernel's rounds (same bits): A/B.
bool f32c_handles(const LstmParams& p, int H, int mode) {
  if (fnssl::tune(FNSSL_TUNE_NO_F32_CLUSTER)) return false;
  if (H != 128 || p.carry || p.c2 != 0 || p.ablate) return false;
  if (p.reserve && mode != 0) return false;   // training forward: no fused residual
  // (block 1's layer, 4 input channels = 144 MFMAs per group-step, gains little — 35.7 against 36.2 ms;
  //  FNSSL_NO_F32C_B1=1 keeps it on the rounds: A/B)
  if (!(p.c0 == 256 && (mode == kSum || mode == 0)) && !(p.c0 == 4 && mode == 0 && !fnssl::tune(FNSSL_TUNE_NO_F32C_B1))) return false;
  // the kernel's addressing takes a group's first sequence as its lowest address
  auto grows = [&](long long so, long long si) { return si >= 0 && so >= (long long)(p.q_inner - 1) * si; };
  if (p.q_inner < 16 || !grows(p.src0.so, p.src0.si) || !grows(p.out_so, p.out_si) ||
      ((mode & kSum) && !grows(p.skip.so, p.skip.si)))
    return false;
  const int ncu = fnssl::device_cus();
  if (ncu < 8 * p.ndir) return false;
  // inference: at least what one full round of three waves per SIMD would cover (below that the planner's rounds / split
  // kernels are tuned); training forward (reserve): from two groups per wave of every cluster — config 4's shard has 1200
  // groups on 1024 SIMDs, which the 2-waves-per-group kernels run as 3 wave-times for 2.34 (0.52 of the roof), while here
  // the groups beyond two per wave rotate over the waves step by step (lstm_f32c.h): balanced over the launch
  const long long groups = (long long)p.ntasks * p.ndir;
  if (p.reserve) return !fnssl::tune(FNSSL_TUNE_TRAIN_NO_F32_CLUSTER) && groups >= 2LL * kF32cWaves * (ncu / 8);
  return groups >= 12LL * ncu;
}

int forward_f32c(LstmParams p, int mode, hipStream_t st) {
  const int ncu = fnssl::device_cus();
  F32ClusterParams cp;
  cp.clusters_per_dir = (ncu / 8) / p.ndir;
  cp.groups_per_cluster = (p.ntasks + cp.clusters_per_dir - 1) / cp.clusters_per_dir;
  cp.status = reinterpret_cast<unsigned*>(p.cluster_ws);
  cp.tags = reinterpret_cast<unsigned*>(p.cluster_ws + 256);
  cp.spin_limit = cluster_spin_limit();
  cp.stall_member = cluster_test_stall();
  cp.rotate = (kF32cWaves & (kF32cWaves - 1)) == 0 && !fnssl::tune(FNSSL_TUNE_F32C_NO_ROTATE);   // A/B knob, same bits
  cp.prio_mode = fnssl::tune(FNSSL_TUNE_F32C_PRIO, 9, 9) ? 0 : 2;   // see F32ClusterParams
  const size_t tag_bytes = (size_t)p.ndir * cp.clusters_per_dir * cp.groups_per_cluster * 8 * sizeof(unsigned);
  if (!p.dry) FNSSL_HIP(hipMemsetAsync(p.cluster_ws, 0, 256 + tag_bytes, st));
#ifdef FNSSL_BUILD_ABLATE   // timing ablations (wrong results): make ABLATE=1 only
  if (const int abl = env_int("FNSSL_F32C_ABL", 1, 1023)) {
    p.ablate = abl;
    if (mode == kSum) return launch_f32c_k<16, 0, kSum, true>(p, cp, st);
    if (p.c0 == 4) return launch_f32c_k<0, 1, 0, true>(p, cp, st);
  }
  p.ablate = 0;
#endif
  if (p.reserve) return p.c0 == 4 ? launch_f32c_k<0, 1, kSave>(p, cp, st) : launch_f32c_k<16, 0, kSave>(p, cp, st);
  if (p.c0 == 4) return launch_f32c_k<0, 1, 0>(p, cp, st);
  // (drift bounds of 1 / 3 / 4 group-steps and 12 waves per member were measured in round 3: +0.5 / 0.0 / +1.2 ms, +1.5 ms)
  if (mode == kSum) return launch_f32c_k<16, 0, kSum>(p, cp, st);
  return launch_f32c_k<16, 0, 0>(p, cp, st);
}

}  // namespace fnssl_lstm
