// bf16-MFMA LSTM kernels (lstm_bf16.h): instantiations for the IPDnet layer shapes, the packer of the bf16
// weight stream and the launch planner.
#include <cstring>

#include "lstm_bf16.h"

namespace fnssl_lstm {

#define TRYB(H_, NW_, M_, NV0_, NV2_, CHQ_, PAD_)                                          \
  if (H == H_ && NW == NW_ && p.c0 == 16 * NV0_ && p.c2 == 16 * NV2_)                       \
    return launch_bf16_k<H_, NW_, M_, NV0_, NV2_, CHQ_, PAD_>(p, nwg, st);

int launch_bf16(const LstmParams& p, int H, int NW, int nwg, hipStream_t st) {
  // IPDnet, hidden 256 (more than two microphones): narrow-band 256 <- [256 | 16]
  TRYB(256, 4, 12, 16, 1, 9, 0)
  // full-band 128 <- [256 | 16] (block 2; also the offline narrow-band layers) and 128 <- 16 (block 1)
  TRYB(128, 4, 12, 16, 1, 7, 0)
  TRYB(128, 8, 4, 16, 1, 7, 0)
  TRYB(128, 4, 12, 1, 0, 6, 0)
  TRYB(128, 8, 4, 1, 0, 6, 0)
  // FN-SSL (no concatenated segment in blocks 2 / 3; block 1 uses the [256 | 16] and 16-channel shapes above with the
  // 4 input channels zero-padded): narrow-band 256 <- 256, full-band / offline narrow-band 128 <- 256
  TRYB(256, 4, 12, 16, 0, 9, 1)
  TRYB(128, 4, 12, 16, 0, 7, 1)
  TRYB(128, 8, 4, 16, 0, 7, 1)
  // IPDnet, hidden 128 (two microphones, input zero-padded to 16 channels): 128 <- [128 | 16], 64 <- [128 | 16], 64 <- 16
  TRYB(128, 4, 12, 8, 1, 10, 0)
  TRYB(128, 8, 4, 8, 1, 5, 0)
  TRYB(64, 4, 12, 8, 1, 8, 0)
  TRYB(64, 8, 4, 8, 1, 8, 0)
  TRYB(64, 4, 12, 1, 0, 4, 0)
  TRYB(64, 8, 4, 1, 0, 4, 0)
  return kNoStatic;
}

static unsigned short to_bf16(float f) {   // round to nearest even
  unsigned u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// Launch plan of the bf16 path: 4 or 8 waves per workgroup, rounds of 8 waves per CU.
int forward_bf16(LstmParams p, int H, hipStream_t st) {
  const int ncu = fnssl::device_cus();
  const int tasks = p.ntasks;
  const long long total = (long long)tasks * p.ndir;
  const int W = (int)((total + ncu - 1) / ncu);
  const int wcap = H > 128 ? 4 : 8;
  const int rounds = (W + wcap - 1) / wcap;
  const int wgs_per_dir_round = ncu / p.ndir > 0 ? ncu / p.ndir : 1;
  int t0 = 0;
  for (int r = 0; r < rounds && t0 < tasks; ++r) {
    const long long left_total = (long long)(tasks - t0) * p.ndir;
    const int want = (int)(((left_total + ncu - 1) / ncu + (rounds - r) - 1) / (rounds - r));
    const int nw = (want <= 4 || H > 128) ? 4 : 8;   // H = 256 keeps 64 + 64 state registers: 4-wave workgroups only
    int t1 = r + 1 == rounds ? tasks : t0 + wgs_per_dir_round * nw;
    if (t1 > tasks) t1 = tasks;
    p.task0 = t0;
    p.task1 = t1;
    p.wgs_per_dir = (t1 - t0 + nw - 1) / nw;
    const int rc = launch_bf16(p, H, nw, p.wgs_per_dir * p.ndir, st);
    if (rc == kNoStatic) {
      fnssl::set_error("lstm_forward: the bf16 path is not built for hidden %d with inputs (%d, %d)", H, p.c0, p.c2);
      return FNSSL_E_INVALID;
    }
    if (rc != FNSSL_OK) return rc;
    t0 = t1;
  }
  return FNSSL_OK;
}

}  // namespace fnssl_lstm

using namespace fnssl_lstm;

extern "C" {

size_t fnssl_lstm_packed_floats_bf16(int c0, int c2, int hidden) {
  if (hidden <= 0 || hidden % 32 || c0 < 0 || c2 < 0 || (c0 & 15) || (c2 & 15) || c0 + c2 == 0) return 0;
  return (size_t)(hidden / 16) * bf16_quads_per_slice(c0, c2, hidden) * 4 * 256;
}

int fnssl_lstm_pack_bf16(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int c0, int c2,
                         int H, float* packed) {
  FNSSL_REQUIRE(w_ih && w_hh && b_ih && b_hh && packed, "lstm_pack_bf16: null pointer");
  const size_t total = fnssl_lstm_packed_floats_bf16(c0, c2, H);
  FNSSL_REQUIRE(total > 0, "lstm_pack_bf16: unsupported sizes (c0 %d, c2 %d multiples of 16; hidden %d of 32)", c0, c2, H);
  std::memset(packed, 0, total * sizeof(float));
  const int I = c0 + c2, NS = H / 16;
  float* recf = packed;
  for (int s = 0; s < NS; ++s) {
    for (int qg = 0; qg < 4; ++qg, recf += 256)   // bias quad, fp32
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
          const int unit = 16 * s + 4 * (l >> 4) + r;
          recf[l * 4 + r] = b_ih[qg * H + unit] + b_hh[qg * H + unit];
        }
    // one quad per pair of 16-column blocks of `mat` (row stride ld); blk >= nblk is zero padding
    auto pack_pairs = [&](const float* mat, int ld, int col0, int nblk) {
      for (int pi = 0; pi < (nblk + 1) / 2; ++pi)
        for (int qg = 0; qg < 4; ++qg, recf += 256) {
          unsigned short* rec = reinterpret_cast<unsigned short*>(recf);
          for (int l = 0; l < 64; ++l)
            for (int half = 0; half < 2; ++half)
              for (int j = 0; j < 4; ++j) {
                const int blk = 2 * pi + half;
                const float v = blk < nblk ? mat[(size_t)(qg * H + 16 * s + (l & 15)) * ld + col0 + 16 * blk + 4 * (l >> 4) + j]
                                           : 0.f;
                rec[l * 8 + half * 4 + j] = to_bf16(v);
              }
        }
    };
    pack_pairs(w_ih, I, 0, c0 >> 4);
    pack_pairs(w_ih, I, c0, c2 >> 4);
    pack_pairs(w_hh, H, 0, NS);
  }
  if ((size_t)(recf - packed) != total) {
    fnssl::set_error("lstm_pack_bf16: internal size mismatch");
    return FNSSL_E_INVALID;
  }
  return FNSSL_OK;
}

}  // extern "C"
