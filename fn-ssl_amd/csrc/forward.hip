// FN_SSL.forward on device (reference FN-SSL/Model.py:72-90 and FNblock.forward
// :31-50): three full-band / narrow-band blocks and the DP-IPD head, with every
// permute / reshape / cat of the reference folded into the strided operand views
// of the LSTM kernels, and every residual add (x + fb_skip, x + nb_skip,
// Model.py:36-37,44-45) folded into the PRODUCING kernel's epilogue: each LSTM
// kernel stores its raw output h and, next to it, h + skip = the next layer's
// input, so every layer streams ONE input tensor.  Full-band tensors live in
// [pairs, nt, nf, 256], narrow-band tensors in [pairs, nf, nt, 256] — the layout
// in which the writing kernel's 16 sequences per wave are adjacent.
#include <algorithm>

#include "common.h"

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Plan {
  int chunk;
  size_t act_floats;   // one [chunk, nt, nf, 256] activation
  size_t lstm_ws;      // bytes
  size_t head_floats;  // [chunk, nt2, 2nf] staging for the DOA layer
  size_t total;        // bytes
};

Plan make_plan(int nb, int nf, int nt, int is_online, int chunk_pairs) {
  Plan p;
  p.chunk = (chunk_pairs <= 0 || chunk_pairs > nb) ? nb : chunk_pairs;
  p.act_floats = (size_t)p.chunk * nt * nf * 256;
  const size_t ws_full = fnssl_lstm_workspace_bytes_ex(p.chunk * nt, 128, 2, FNSSL_PRECISION_FP32);
  const size_t ws_narr = is_online ? fnssl_lstm_workspace_bytes_ex(p.chunk * nf, 256, 1, FNSSL_PRECISION_FP32)
                                   : fnssl_lstm_workspace_bytes_ex(p.chunk * nf, 128, 2, FNSSL_PRECISION_FP32);
  p.lstm_ws = align_up(std::max(ws_full, ws_narr), 256);
  p.head_floats = (size_t)p.chunk * (nt / FNSSL_SEG_FRAMES) * 2 * nf;
  p.total = 6 * align_up(p.act_floats * 4, 256) + p.lstm_ws + align_up(p.head_floats * 4, 256) + 256;
  return p;
}

}  // namespace

extern "C" {

size_t fnssl_forward_workspace_bytes(int nb, int nf, int nt, int is_online, int chunk_pairs) {
  if (nb <= 0 || nf <= 0 || nt <= 0) return 0;
  return make_plan(nb, nf, nt, is_online, chunk_pairs).total;
}

int fnssl_forward(const fnssl_net* net, const float* x0, int nb, int nf, int nt, float* out, void* workspace,
                  size_t workspace_bytes, int chunk_pairs, void* stream) {
  FNSSL_REQUIRE(net && x0, "forward: null pointer");
  FNSSL_REQUIRE(nb > 0 && nf > 0 && nt > 0, "forward: empty problem (nb %d, nf %d, nt %d)", nb, nf, nt);
  if (nt / FNSSL_SEG_FRAMES == 0) return FNSSL_OK;   // AvgPool2d((12,1)) floors: the output is empty
  FNSSL_REQUIRE(out, "forward: null output pointer");
  FNSSL_REQUIRE(net->input_size > 0 && net->input_size % 4 == 0, "forward: input_size %d must be a multiple of 4",
                net->input_size);
  FNSSL_REQUIRE(!net->doa_wt || nf == 256, "forward: the DOA layer needs nf = 256 (Linear(512, 180))");
  const Plan pl = make_plan(nb, nf, nt, net->is_online, chunk_pairs);
  if (!workspace || workspace_bytes < pl.total) {
    fnssl::set_error("forward: workspace %zu < %zu bytes", workspace_bytes, pl.total);
    return FNSSL_E_WORKSPACE;
  }
  char* wsb = static_cast<char*>(workspace);
  wsb = reinterpret_cast<char*>(align_up(reinterpret_cast<uintptr_t>(wsb), 256));
  const size_t act_bytes = align_up(pl.act_floats * 4, 256);
  float* F[2] = {reinterpret_cast<float*>(wsb), reinterpret_cast<float*>(wsb + act_bytes)};
  float* N[2] = {reinterpret_cast<float*>(wsb + 2 * act_bytes), reinterpret_cast<float*>(wsb + 3 * act_bytes)};
  float* Sf = reinterpret_cast<float*>(wsb + 4 * act_bytes);   // f_k + n_{k-1}: narrow-band input, full layout
  float* Sn = reinterpret_cast<float*>(wsb + 5 * act_bytes);   // n_k + f_k    : full-band input, narrow layout
  float* lws = reinterpret_cast<float*>(wsb + 6 * act_bytes);
  float* head_tmp = reinterpret_cast<float*>(wsb + 6 * act_bytes + pl.lstm_ws);

  const int cin = net->input_size;
  const int nt2 = nt / FNSSL_SEG_FRAMES;
  const int narr_h = net->is_online ? 256 : 128;
  const int narr_dirs = net->is_online ? 1 : 2;
  const long long C = 256;
  // strides of the two activation layouts, as (sequence-outer, sequence-inner, step)
  const long long PS = (long long)nt * nf * C;                         // one pair
  // full layout [pair, t, f, C]  : seen by the full-band kernel (seq = (pair, t), step = f) ...
  const fnssl_view fl_by_full = {nullptr, PS, (long long)nf * C, C};
  // ... and by the narrow-band kernel (seq = (pair, f), step = t)
  const fnssl_view fl_by_narr = {nullptr, PS, C, (long long)nf * C};
  // narrow layout [pair, f, t, C]
  const fnssl_view nl_by_full = {nullptr, PS, C, (long long)nt * C};
  const fnssl_view nl_by_narr = {nullptr, PS, (long long)nt * C, C};
  auto with = [](fnssl_view v, const float* p) {
    v.p = p;
    return v;
  };

  for (int p0 = 0; p0 < nb; p0 += pl.chunk) {
    const int G = std::min(pl.chunk, nb - p0);
    const float* xc = x0 + (size_t)p0 * nt * nf * cin;
    for (int blk = 0; blk < 3; ++blk) {
      float* Fo = F[blk & 1];
      float* No = N[blk & 1];
      const float* Np = N[(blk + 1) & 1];   // previous block's narrow-band output (raw)
      // ---- full-band BiLSTM over frequency, one sequence per (pair, frame) ----
      fnssl_lstm_desc d = {};
      d.hidden = 128;
      d.ndir = 2;
      d.nseq = G * nt;
      d.q_inner = nt;
      d.nsteps = nf;
      if (blk == 0) {
        d.src0 = fnssl_view{xc, (long long)nt * nf * cin, (long long)nf * cin, cin};
        d.c0 = cin;
      } else {
        d.src0 = with(nl_by_full, Sn);      // x + fb_skip (Model.py:35-37), summed by the previous narrow kernel
        d.c0 = 256;
      }
      d.out = Fo;
      d.out_so = fl_by_full.so;
      d.out_si = fl_by_full.si;
      d.out_st = fl_by_full.st;
      if (blk > 0) {                        // dropout(f) + nb_skip (Model.py:44-45) for this block's narrow LSTM
        d.skip = with(nl_by_full, Np);
        d.out_sum = Sf;
      }
      d.wpack[0] = net->wpack[blk][0][0];
      d.wpack[1] = net->wpack[blk][0][1];
      d.workspace = lws;
      d.workspace_bytes = pl.lstm_ws;
      d.fallback_count = net->fallback_count;
      d.tuning = net->tuning;
      int rc = fnssl_lstm_forward(&d, stream);
      if (rc != FNSSL_OK) return rc;
      // ---- narrow-band LSTM over time, one sequence per (pair, bin) ------------
      fnssl_lstm_desc e = {};
      e.hidden = narr_h;
      e.ndir = narr_dirs;
      e.nseq = G * nf;
      e.q_inner = nf;
      e.nsteps = nt;
      e.c0 = 256;
      if (blk == 0) {
        e.src0 = with(fl_by_narr, Fo);                                                     // dropout(f) (:40-41)
        e.src2 = fnssl_view{xc, (long long)nt * nf * cin, cin, (long long)nf * cin};       // cat nb_skip (:42-43)
        e.c2 = cin;
      } else {
        e.src0 = with(fl_by_narr, Sf);
      }
      e.out = No;
      e.out_so = nl_by_narr.so;
      e.out_si = nl_by_narr.si;
      e.out_st = nl_by_narr.st;
      if (blk < 2) {                        // next block's full-band input: n + f  (x + fb_skip)
        e.skip = with(fl_by_narr, Fo);
        e.out_sum = Sn;
      }
      e.wpack[0] = net->wpack[blk][1][0];
      e.wpack[1] = net->wpack[blk][1][1];
      e.workspace = lws;
      e.workspace_bytes = pl.lstm_ws;
      e.fallback_count = net->fallback_count;
      e.tuning = net->tuning;
      rc = fnssl_lstm_forward(&e, stream);
      if (rc != FNSSL_OK) return rc;
    }
    // ---- head on block 3's narrow-band output (N[0]) ----------------------------
    if (nt2 > 0) {
      if (net->doa_wt) {
        int rc = fnssl_head(N[0], G, nf, nt, net->emb_w, net->emb_b, head_tmp, stream);
        if (rc != FNSSL_OK) return rc;
        rc = fnssl_linear(head_tmp, G * nt2, 2 * nf, net->doa_wt, net->doa_b, 180,
                          out + (size_t)p0 * nt2 * 180, stream);
        if (rc != FNSSL_OK) return rc;
      } else {
        int rc = fnssl_head(N[0], G, nf, nt, net->emb_w, net->emb_b, out + (size_t)p0 * nt2 * 2 * nf, stream);
        if (rc != FNSSL_OK) return rc;
      }
    }
  }
  return FNSSL_OK;
}

}  // extern "C"
