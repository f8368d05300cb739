// Wide bf16-MFMA LSTM kernels (lstm_bf16w.h): instantiations for the IPDnet (hidden 256) layer shapes, the packer of
// the tile-ordered bf16 weight stream and the launch planner.
#include <cstdlib>
#include <cstring>

#include "lstm_bf16p.h"
#include "tuning.h"

namespace fnssl_lstm {

bool bf16c_handles(const LstmParams& p, int H, int flags);           // lstm_bf16c.hip
int forward_bf16c(LstmParams p, int H, int flags, hipStream_t st);

// NW = 2: two consumer waves + two loader waves on the SIMDs the launch leaves idle; NW = 4: four consumers that
// fetch their own stream (H = 128 only: at H = 256 the ring plus four h staging areas exceed the LDS)
#define TRYW(H_, NB0_, NB2_, FL_)                                                                                \
  if (H == H_ && p.c0 == 16 * NB0_ && p.c2 == 16 * NB2_ && flags == (FL_)) {                                    \
    if (NW == 2) return launch_bf16w_k<H_, 2, NB0_, NB2_, FL_, 3, 2>(p, nwg, st);                                \
    if (NW == 4 && H_ < 256) return launch_bf16w_k<H_ < 256 ? H_ : 128, 4, NB0_, NB2_, FL_, 3, 0>(p, nwg, st);   \
  }

int launch_bf16w(const LstmParams& p, int H, int NW, int flags, int nwg, hipStream_t st) {
  // IPDnet, hidden 256: full-band 128 <- 16 fp32 feature channels; narrow-band 256 <- [256 bf16 | 16 fp32];
  // full-band 128 <- [256 bf16 | 16 fp32]; outputs bf16 (the conv head reads bf16: fnssl_conv3x3_causal_bf16a).
  // (fp32 main input / output at H = 256 would need > 512 registers per lane: convert outside instead.)
  TRYW(128, 1, 0, kW_F0)
  TRYW(256, 16, 1, kW_F2)
  TRYW(128, 16, 1, kW_F2)
  return kNoStatic;
}

// NG_ = the largest groups-per-workgroup instantiation built for the shape (lstm_bf16p_kernel's NG; 2 = the pair of
// groups every shape runs with — larger workgroups were measured and do not pay, see forward_bf16w)
#define TRYP(H_, NB0_, NB2_, FL_)                                                    \
  if (H == H_ && p.c0 == 16 * NB0_ && p.c2 == 16 * NB2_ && flags == (FL_))          \
    return drain ? launch_bf16p_k<H_, NB0_, NB2_, FL_, 0, 2, true>(p, nwg, st) : launch_bf16p_k<H_, NB0_, NB2_, FL_>(p, nwg, st);

// pair-split kernels (lstm_bf16p.h): workgroup = ng groups of 32 sequences x 2 roles (ng = 2, or 5 for the H = 128 shapes)
int launch_bf16p(const LstmParams& p, int H, int flags, int nwg, hipStream_t st) {
#ifdef FNSSL_BUILD_ABLATE   // timing ablations of the config-3 narrow-band kernel (wrong results): make ABLATE=1 only
  if (const char* e = getenv("FNSSL_BF16W_ABL")) {
    if (H == 256 && p.c0 == 256 && p.c2 == 16 && flags == kW_F2) {
      switch (atoi(e)) {
        case 1: return launch_bf16p_k<256, 16, 1, kW_F2, 1>(p, nwg, st);
        case 4: return launch_bf16p_k<256, 16, 1, kW_F2, 4>(p, nwg, st);
        case 5: return launch_bf16p_k<256, 16, 1, kW_F2, 5>(p, nwg, st);
        default: break;
      }
    }
  }
#endif
  const bool drain = fnssl::tune(FNSSL_TUNE_BF16P_DRAIN) != 0;   // vmcnt(0) at every ring barrier, all shapes
  TRYP(128, 1, 0, kW_F0)
  TRYP(256, 16, 1, kW_F2)
  TRYP(128, 16, 1, kW_F2)
  return kNoStatic;
}

// One launch: every 32-sequence group of every direction.  Default: the pair-split kernels; FNSSL_BF16W_SOLO=1 keeps
// the one-wave-per-group kernels of lstm_bf16w.h (A/B).
int forward_bf16w(LstmParams p, int H, int flags, hipStream_t st, int* family) {
  const int ncu = fnssl::device_cus();
  const int groups = (p.nseq + 31) / 32;
  const long long total = (long long)groups * p.ndir;
  p.task0 = 0;
  p.task1 = groups;
  // cluster-resident kernel (lstm_bf16c.h) for the shape it is built for: weights stay in LDS, h_t is exchanged through L2
  // — followed by the pair-split launch below as its GUARDED fallback (include/fnssl.h, fnssl_lstm_forward)
  bool guarded = false;
  if (bf16c_handles(p, H, flags)) {
    const int rc = forward_bf16c(p, H, flags, st);
    if (rc == FNSSL_OK) {
      if (family) *family = FNSSL_LSTM_FAMILY_BF16_CLUSTER;
      if (p.dry) return FNSSL_OK;
      guarded = true;
      p.guard = reinterpret_cast<const unsigned*>(p.cluster_ws);
    } else if (rc != kNoCluster) {
      return rc;
    }
  }
  int rc;
  if (!fnssl::tune(FNSSL_TUNE_BF16W_SOLO) || guarded) {
    if (family && !guarded) *family = FNSSL_LSTM_FAMILY_BF16_PAIR;
    // Two 32-sequence groups per workgroup.  lstm_bf16p_kernel takes NG groups (template), and 3 / 4 / 5 were built and
    // measured at config 3 (profiles/r03/h_bf16p_groups_per_workgroup.txt) in the hope of turning the full-band layers'
    // 600 workgroups (2.34 rounds run as 3) into 400 / 300 / 240: bit-identical, never faster (full-band layers 13.1 ms
    // per step at 2 groups, 13.0 / 15.2 / 14.0 at 3 / 4 / 5) — a round's time grows with its groups, because every wave
    // reads each 1-KiB weight record of its role from LDS once per 32-cycle MFMA: four waves already draw the LDS's
    // 128 B/clk, so the formulation is bound by LDS bandwidth, not by rounds.
    const int ng = 2;
    p.wgs_per_dir = (groups + ng - 1) / ng;
    rc = launch_bf16p(p, H, flags, p.wgs_per_dir * p.ndir, st);
  } else {
    if (family) *family = FNSSL_LSTM_FAMILY_BF16_SOLO;
    const int nw = (total <= 2ll * ncu || H >= 256) ? 2 : 4;
    p.wgs_per_dir = (groups + nw - 1) / nw;
    rc = launch_bf16w(p, H, nw, flags, p.wgs_per_dir * p.ndir, st);
  }
  if (rc == kNoStatic) {
    fnssl::set_error("lstm_forward: the wide bf16 path is not built for hidden %d, inputs (%d, %d), element mask %d", H,
                     p.c0, p.c2, flags);
    return FNSSL_E_INVALID;
  }
  return rc;
}

static unsigned short to_bf16w(float f) {   // round to nearest even
  unsigned u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float from_bf16w(unsigned short h) {
  const unsigned u = (unsigned)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

}  // namespace fnssl_lstm

using namespace fnssl_lstm;

extern "C" {

size_t fnssl_lstm_packed_floats_bf16w(int c0, int c2, int hidden) {
  if (hidden <= 0 || hidden % 16 || c0 < 0 || c2 < 0 || (c0 & 15) || (c2 & 15) || c0 + c2 == 0) return 0;
  return (size_t)(hidden / 8) * bf16w_records_per_tile(c0, c2, hidden) * 256;
}

int fnssl_lstm_pack_bf16w(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int c0, int c2,
                          int H, float* packed) {
  FNSSL_REQUIRE(w_ih && w_hh && b_ih && b_hh && packed, "lstm_pack_bf16w: null pointer");
  const size_t total = fnssl_lstm_packed_floats_bf16w(c0, c2, H);
  FNSSL_REQUIRE(total > 0, "lstm_pack_bf16w: unsupported sizes (c0 %d, c2 %d, hidden %d: multiples of 16)", c0, c2, H);
  std::memset(packed, 0, total * sizeof(float));
  const int I = c0 + c2, NT = H / 8, NKX = I / 16, NKH = H / 16;
  unsigned short* rec = reinterpret_cast<unsigned short*>(packed);   // 512 bf16 per record: [lane 64][i 8]
  for (int T = 0; T < NT; ++T) {
    auto row_of = [&](int l) { return ((l & 31) >> 3) * H + 8 * T + (l & 7); };   // tile row rho = [gate][unit]
    // record 0: bias = hi + mid + lo in k-slots (hb 0, i 0..2)
    for (int l = 0; l < 32; ++l) {
      const float b = b_ih[row_of(l)] + b_hh[row_of(l)];
      const unsigned short hi = to_bf16w(b);
      const float r1 = b - from_bf16w(hi);
      const unsigned short mid = to_bf16w(r1);
      const unsigned short lo = to_bf16w(r1 - from_bf16w(mid));
      rec[l * 8 + 0] = hi;
      rec[l * 8 + 1] = mid;
      rec[l * 8 + 2] = lo;
    }
    rec += 512;
    for (int b = 0; b < NKX; ++b, rec += 512)
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 8; ++i) rec[l * 8 + i] = to_bf16w(w_ih[(size_t)row_of(l) * I + 16 * b + 8 * (l >> 5) + i]);
    for (int s = 0; s < NKH; ++s, rec += 512)
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 8; ++i) {
          const int unit = 16 * s + 8 * (i >> 2) + 4 * (l >> 5) + (i & 3);
          rec[l * 8 + i] = to_bf16w(w_hh[(size_t)row_of(l) * H + unit]);
        }
  }
  if ((size_t)(reinterpret_cast<float*>(rec) - packed) != total) {
    fnssl::set_error("lstm_pack_bf16w: internal size mismatch");
    return FNSSL_E_INVALID;
  }
  return FNSSL_OK;
}

}  // extern "C"
