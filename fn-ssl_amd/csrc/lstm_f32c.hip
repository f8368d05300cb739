// Cluster-resident fp32 LSTM kernel (lstm_f32c.h): instantiations for FN-SSL's layer shapes and the launcher.
#include <cstdlib>

#include "lstm_f32c.h"
#include "tuning.h"

namespace fnssl_lstm {

constexpr int kF32cWavesH256 = 16;       // H = 256, a wave owns its groups: h_{t-1} streamed through the operand ring, four waves per SIMD
constexpr int kF32cWavesH256Split = 8;   // H = 256, gate split: the row of h_{t-1} held (64 registers), two waves per SIMD = two slots

// Which calls the cluster-resident kernel takes (NO_F32_CLUSTER keeps the per-wave rounds / split kernels, same bits: A/B).
//   H = 128: the full-band layers (block 1: 4 input channels; blocks 2-3: 256, optionally with the fused residual output)
//            and the offline model's narrow-band layers of blocks 2-3 (same shape) — at ANY size since round 5: with a
//            handful of groups (one utterance, a streaming chunk) the several-waves-per-group kernels stream the whole
//            0.8 MB matrix from L2 per group and step, here a step costs one member's matrix work plus a hand-off
//            (one 4-mic utterance: 25.1 -> 11.7 ms for the three layers; 4 utterances: 40.3 -> 30.1; from 16 utterances on
//            it was the cluster kernel already: tools/latency_bench.py, profiles/r05/);
//   H = 256: the online model's narrow-band layers (clusters of 16) below the full-chip size; full-chip launches stay on
//            lstm_static3_kernel's rounds (0.89 of the roof: a wave that owns its groups for the whole recurrence needs no
//            hand-off at all).
// every tensor of the call has its sequences evenly spaced: sequence q at q * si
static bool f32c_uniform(const LstmParams& p, int mode) {
  auto even = [&](long long so, long long si) { return si >= 0 && so == (long long)p.q_inner * si; };
  return even(p.src0.so, p.src0.si) && even(p.out_so, p.out_si) && (!(mode & kSum) || even(p.skip.so, p.skip.si)) &&
         (!(mode & kHas2) || even(p.src2.so, p.src2.si));
}

bool f32c_handles(const LstmParams& p, int H, int mode) {
  if (fnssl::tune(FNSSL_TUNE_NO_F32_CLUSTER)) return false;
  if ((H != 128 && H != 256) || p.ablate) return false;
  if (H == 128) {
    if (p.c2 != 0 || p.carry) return false;
    if (p.reserve && mode != 0) return false;   // training forward: no fused residual
    // (block 1's layer, 4 input channels = 144 MFMAs per group-step, gains little at full-chip size — 35.7 against 36.2 ms;
    //  NO_F32C_B1 keeps it on the rounds: A/B)
    if (!(p.c0 == 256 && (mode == kSum || mode == 0)) && !(p.c0 == 4 && mode == 0 && !fnssl::tune(FNSSL_TUNE_NO_F32C_B1))) return false;
  } else {
    if (p.ndir != 1 || p.c0 != 256 || fnssl::tune(FNSSL_TUNE_NO_F32_SMALL)) return false;
    const bool plain = p.c2 == 0 && (mode == 0 || mode == kSum), cat = p.c2 == 4 && (mode == kHas2 || mode == (kHas2 | kSum));
    if (!plain && !cat) return false;
    if (p.reserve && ((mode & kSum) || p.carry)) return false;   // training forward: no fused residual, no streaming
  }
  // the kernel's addressing takes a group's first sequence as its lowest address and lets a group cross ONE outer index
  // (q_inner >= 16) — or the sequences are evenly spaced in every tensor (so == q_inner * si: a 12-frame streaming chunk's
  // full-band layers, q_inner = 12), which forward_f32c() runs as one outer index
  auto grows = [&](long long so, long long si) { return si >= 0 && so >= (long long)(p.q_inner - 1) * si; };
  if (!f32c_uniform(p, mode) &&
      (p.q_inner < 16 || !grows(p.src0.so, p.src0.si) || !grows(p.out_so, p.out_si) ||
       ((mode & kSum) && !grows(p.skip.so, p.skip.si)) || ((mode & kHas2) && !grows(p.src2.so, p.src2.si))))
    return false;
  const int ncu = cluster_cus();
  if (ncu < (H / 16) * p.ndir) return false;
  const long long groups = (long long)p.ntasks * p.ndir;
  if (const int mg = fnssl::tune(FNSSL_TUNE_F32C_MIN_GROUPS, 1, 1 << 30)) {
    if (groups < mg) return false;
  }
  // training forward (reserve).  H = 128: config 4's shard has 1200 groups on 1024 SIMDs, which the 2-waves-per-group kernels
  // run as 3 wave-times for 2.34 (0.52 of the roof), while here the groups beyond two per wave rotate over the waves step by
  // step (lstm_f32c.h): balanced over the launch — and, round 5, every SMALLER shard too: 16 utterances per GPU took 38.3 ms
  // on the split kernels, as long as 32 (37.1).  H = 256 (round 5): the narrow-band layers of shards BELOW config 4's (fewer
  // than two groups per CU), which the 4-waves-per-group kernels ran at 80 - 86 ms per step whatever the batch (16
  // utterances: 85.9 ms against 59.9 for 32; 24 utterances: 111 ms); from two groups per CU on, lstm_fwd2_kernel (0.82) stays.
  if (p.reserve) {
    if (fnssl::tune(FNSSL_TUNE_TRAIN_NO_F32_CLUSTER)) return false;
    // (H = 256, end of round 5: with the streamed row and 16 waves per member the cluster kernel also takes config 4's shard and
    //  anything larger — 57.8 against lstm_fwd2_kernel's 59.6 ms for the three layers; lstm_fwd2_kernel stays its guarded fallback)
    return true;
  }
  // H = 256: everything below the full-chip launch (12 groups per CU: lstm_static3_kernel's one round of 12 waves per CU, 0.89
  // of the roof) — in between the rounds are paced by their fullest SIMD (9 groups per CU = 3, 2, 2, 2 waves per SIMD), the
  // cluster's groups are work items that balance (16 utterances: 202 -> 176 ms for the three layers, 0.84 of the roof)
  if (H == 256) {
    // ... and ABOVE it whenever the groups do not fill whole rounds of 12 waves per CU (48 utterances = 18 groups per CU: two
    // rounds of 9 = 3, 2, 2, 2 waves per SIMD, i.e. two full round-times for 1.5 rounds of work): the streamed-row form keeps
    // its 0.86 - 0.89 at any group count, the rounds lose what their last round idles
    const long long full = 12LL * ncu, rounds = (groups + full - 1) / full;
    return groups < full || groups * 100 < rounds * full * 97;
  }
  return !fnssl::tune(FNSSL_TUNE_NO_F32_SMALL) || groups >= 12LL * ncu;
}

int forward_f32c(LstmParams p, int H, int mode, hipStream_t st) {
  if (f32c_uniform(p, mode)) p.q_inner = p.nseq;   // one outer index: no group ever crosses (any q_inner, e.g. 12 frames)
  const int ncu = cluster_cus();
  const int members = H / 16;
  F32ClusterParams cp;
  cp.clusters_per_dir = (ncu / members) / p.ndir;
  cp.groups_per_cluster = (p.ntasks + cp.clusters_per_dir - 1) / cp.clusters_per_dir;
  cp.status = reinterpret_cast<unsigned*>(p.cluster_ws);
  cp.tags = reinterpret_cast<unsigned*>(p.cluster_ws + 256);
  cp.spin_limit = cluster_spin_limit();
  cp.stall_member = cluster_test_stall();
  cp.rotate = !fnssl::tune(FNSSL_TUNE_F32C_NO_ROTATE);   // A/B knob, same bits (wave counts are powers of two)
  cp.prio_mode = fnssl::tune(FNSSL_TUNE_F32C_PRIO, 9, 9) ? 0 : 2;   // see F32ClusterParams
  const size_t tag_bytes = (size_t)p.ndir * cp.clusters_per_dir * cp.groups_per_cluster * members * sizeof(unsigned);
  if (!p.dry) FNSSL_HIP(hipMemsetAsync(p.cluster_ws, 0, 256 + tag_bytes, st));
  // Gate split (lstm_f32c.h): only with ONE group per cluster (one 2-mic utterance, the full-band layers of a streaming
  // chunk) — there a step is the group's own matrix work plus its hand-off, and four SIMDs share the former.  With more
  // groups the waves that each own a group hide one another's hand-off latency, which two (four) slots working through
  // their groups one by one do not: measured at 6 - 8 groups per cluster 42.0 against 26.8 ms per 4-mic utterance, at 1 - 2
  // groups (2-mic, 300 frames) the H = 128 layers 5.9 against 5.4 ms; at one group 9.75 against 12.7 ms per utterance and
  // 3.7 against 5.2 ms per 12-frame chunk (profiles/r05/).
  const int gs_knob = fnssl::tune(FNSSL_TUNE_F32C_GATE_SPLIT, 1, 4);
  const bool gsplit = gs_knob ? gs_knob == 4 : cp.groups_per_cluster <= 1;
  if (H == 256) {
    constexpr int W = kF32cWavesH256;
    if (p.reserve)
      return p.c2 == 4 ? launch_f32c_k<256, 16, 1, kHas2 | kSave, false, 2, W>(p, cp, st) : launch_f32c_k<256, 16, 0, kSave, false, 2, W>(p, cp, st);
    if (gsplit) {
      if (mode == (kHas2 | kSum)) return launch_f32c_k<256, 16, 1, kHas2 | kSum, false, 2, kF32cWavesH256Split, 4>(p, cp, st);
      if (mode == kHas2) return launch_f32c_k<256, 16, 1, kHas2, false, 2, kF32cWavesH256Split, 4>(p, cp, st);
      if (mode == kSum) return launch_f32c_k<256, 16, 0, kSum, false, 2, kF32cWavesH256Split, 4>(p, cp, st);
      return launch_f32c_k<256, 16, 0, 0, false, 2, kF32cWavesH256Split, 4>(p, cp, st);
    }
    // Up to one group per wave of the held-row form (8 waves): a step is one group's chain per wave, and the held row's loads all
    // go out at once where the ring (depth 4) requests them block by block — one 4-mic utterance 15.4 against 16.1 ms (l_*, n_*)
    if (cp.groups_per_cluster <= kF32cWavesH256Split) {
      constexpr int W8 = kF32cWavesH256Split;
      if (mode == (kHas2 | kSum)) return launch_f32c_k<256, 16, 1, kHas2 | kSum, false, 2, W8>(p, cp, st);
      if (mode == kHas2) return launch_f32c_k<256, 16, 1, kHas2, false, 2, W8>(p, cp, st);
      if (mode == kSum) return launch_f32c_k<256, 16, 0, kSum, false, 2, W8>(p, cp, st);
      return launch_f32c_k<256, 16, 0, 0, false, 2, W8>(p, cp, st);
    }
    if (mode == (kHas2 | kSum)) return launch_f32c_k<256, 16, 1, kHas2 | kSum, false, 2, W>(p, cp, st);
    if (mode == kHas2) return launch_f32c_k<256, 16, 1, kHas2, false, 2, W>(p, cp, st);
    // (drift bounds of 1 / 3 / 4 / none measured at the 'M'-pairing size, 96 groups per cluster: 60.5 / 58.3 / 58.4 / 58.6 ms per
    //  layer against 58.5 with the default 2 — profiles/r05/)
    if (mode == kSum) return launch_f32c_k<256, 16, 0, kSum, false, 2, W>(p, cp, st);
    return launch_f32c_k<256, 16, 0, 0, false, 2, W>(p, cp, st);
  }
  if (gsplit && !p.reserve) {
    if (p.c0 == 4) return launch_f32c_k<128, 0, 1, 0, false, 2, kF32cWaves, 4>(p, cp, st);
    if (mode == kSum) return launch_f32c_k<128, 16, 0, kSum, false, 2, kF32cWaves, 4>(p, cp, st);
    return launch_f32c_k<128, 16, 0, 0, false, 2, kF32cWaves, 4>(p, cp, st);
  }
#ifdef FNSSL_BUILD_ABLATE   // timing ablations (wrong results): make ABLATE=1 only
  if (const int abl = env_int("FNSSL_F32C_ABL", 1, 1023)) {
    p.ablate = abl;
    if (mode == kSum) return launch_f32c_k<128, 16, 0, kSum, true>(p, cp, st);
    if (p.c0 == 4) return launch_f32c_k<128, 0, 1, 0, true>(p, cp, st);
  }
  p.ablate = 0;
#endif
  if (p.reserve) return p.c0 == 4 ? launch_f32c_k<128, 0, 1, kSave>(p, cp, st) : launch_f32c_k<128, 16, 0, kSave>(p, cp, st);
  if (p.c0 == 4) return launch_f32c_k<128, 0, 1, 0>(p, cp, st);
  // (drift bounds of 1 / 3 / 4 group-steps and 12 waves per member were measured in round 3: +0.5 / 0.0 / +1.2 ms, +1.5 ms)
  if (mode == kSum) return launch_f32c_k<128, 16, 0, kSum>(p, cp, st);
  return launch_f32c_k<128, 16, 0, 0>(p, cp, st);
}

}  // namespace fnssl_lstm
