// Whole-network training backward / step behind the C ABI ("later fnssl_backward(...)"; reference
// FN-SSL/Lightning/main.py:149-157 training_step, :191-198 cal_loss, :269-271 Adam).
//
// fnssl_train_backward runs, for one chunk of microphone pairs, the train-mode forward of FN_SSL (the LSTM kernels
// that also save their gate activations, dropout + residual adds as fnssl_train_combine), the MSE loss, and the
// backward pass (head backward, one BPTT kernel per LSTM layer, the weight gradients, dropout backward), accumulating
// into the caller's flat gradient vector — what fnssl/train.py::TrainEngine._chunk orchestrates from Python, with no
// Python and no torch in it.  fnssl_train_step = zero the gradient + backward + fnssl_adam_step (single process); a
// multi-GPU caller puts its sum all-reduce of the flat gradient between fnssl_train_backward and fnssl_adam_step.
//
// The weight gradients  dW = dA^T [x | h_prev],  db = sum dA  (a 2.5 M-row reduction) are ONE launch of the library's own
// split-K fp32-MFMA kernel per layer (fnssl_lstm_weight_grads, csrc/wgrad.hip) — the same entry point the Python engine
// calls; h_prev is an index shift inside its loader, never materialised.  (Rounds 1-2 called rocBLAS here through
// dlopen; nothing in this library calls a vendor GEMM any more.)
#include <cstring>
#include <vector>

#include "common.h"

namespace {

constexpr int kHFull = 128, kCh = 256;

struct Layer {
  const char* name;
  bool full;        // full-band (sequences = (b, t), steps along f) or narrow-band
  int hidden, ndir, c0, c2, c0g;
  long long off_wih[2], off_whh[2], off_bih[2], off_bhh[2];   // floats into the flat vectors, per direction
  size_t fw_floats, bw_floats;                                 // packed stream sizes per direction
  size_t map_fw_a[2], map_fw_b[2], map_bw[2];                  // offsets (ints) into the device map buffer
};

}  // namespace

struct fnssl_train {
  int is_online;
  Layer L[6];
  long long off_emb_w, off_emb_b, nparam;   // flat length = 1 + nparam (element 0 is a constant 0)
  std::vector<int> maps;                    // host copy of all gather maps
  const int* dmaps = nullptr;               // device copy (caller-owned buffer)
};

namespace {

// ---- small kernels ------------------------------------------------------------------------------------------
__global__ void gather_kernel(const float* __restrict__ theta, const int* __restrict__ ia, const int* __restrict__ ib,
                              long long n, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = theta[ia[i]] + (ib ? theta[ib[i]] : 0.f);
}

// ---- logical [b, t, f, C] tensors -----------------------------------------------------------------------------
struct T4 {
  float* p;
  long long sb, st, sf;
  int c;
};
T4 make_F(float* p, int nt, int nf, int c) { return T4{p, (long long)nt * nf * c, (long long)nf * c, c, c}; }   // [b,t,f,C]
T4 make_N(float* p, int nt, int nf, int c) { return T4{p, (long long)nf * nt * c, c, (long long)nt * c, c}; }   // [b,f,t,C]
T4 chan(const T4& t, int c0, int c) { return T4{t.p + c0, t.sb, t.st, t.sf, c}; }
fnssl_btf_view btf(const T4& t) { return fnssl_btf_view{t.p, t.sb, t.st, t.sf}; }
fnssl_view seqview(const T4& t, bool full) {
  return full ? fnssl_view{t.p, t.sb, t.st, t.sf} : fnssl_view{t.p, t.sb, t.sf, t.st};
}

struct Ctx {
  hipStream_t st;
  int nbp, nt, nf;
  void* wgrad_ws;
  size_t wgrad_ws_bytes;
  float* lstm_ws;
  size_t lstm_ws_bytes;
  void* bwd_ws;
  size_t bwd_ws_bytes;
};

#define TRY(call)                  \
  do {                             \
    const int rc__ = (call);       \
    if (rc__ != FNSSL_OK) return rc__; \
  } while (0)

int combine(const Ctx& c, const T4& out, std::initializer_list<T4> masked, std::initializer_list<T4> plain, bool use_mask,
            unsigned seed, long long b0) {
  fnssl_btf_view mv[3], pv[3];
  int nm = 0, np = 0;
  for (const T4& t : masked) mv[nm++] = btf(t);
  for (const T4& t : plain) pv[np++] = btf(t);
  return fnssl_train_combine(out.p, out.sb, out.st, out.sf, c.nbp, c.nt, c.nf, out.c, mv, nm, pv, np, use_mask ? 1 : 0, seed,
                             b0, c.st);
}

int lstm_fwd(const Ctx& c, const Layer& L, const T4& x0, const T4* x2, const float* const* wp, const T4& out,
             float* reserve, size_t reserve_bytes) {
  fnssl_lstm_desc d;
  std::memset(&d, 0, sizeof(d));
  d.src0 = seqview(x0, L.full);
  d.c0 = L.c0;
  if (x2) {
    d.src2 = seqview(*x2, L.full);
    d.c2 = L.c2;
  }
  const fnssl_view ov = seqview(out, L.full);
  d.out = const_cast<float*>(ov.p);
  d.out_so = ov.so;
  d.out_si = ov.si;
  d.out_st = ov.st;
  d.hidden = L.hidden;
  d.ndir = L.ndir;
  d.nseq = c.nbp * (L.full ? c.nt : c.nf);
  d.q_inner = L.full ? c.nt : c.nf;
  d.nsteps = L.full ? c.nf : c.nt;
  d.wpack[0] = wp[0];
  d.wpack[1] = L.ndir == 2 ? wp[1] : nullptr;
  d.workspace = c.lstm_ws;
  d.workspace_bytes = c.lstm_ws_bytes;
  d.reserve = reserve;
  d.reserve_bytes = reserve_bytes;
  return fnssl_lstm_forward(&d, c.st);
}

int lstm_bwd(const Ctx& c, const Layer& L, const float* reserve, const T4& dh, const T4& da, const T4* dx,
             const float* const* wbw) {
  fnssl_lstm_bwd_desc d;
  std::memset(&d, 0, sizeof(d));
  d.reserve = reserve;
  d.dh = seqview(dh, L.full);
  const fnssl_view av = seqview(da, L.full);
  d.da = const_cast<float*>(av.p);
  d.da_so = av.so;
  d.da_si = av.si;
  d.da_st = av.st;
  if (dx) {
    const fnssl_view xv = seqview(*dx, L.full);
    d.dx = const_cast<float*>(xv.p);
    d.dx_so = xv.so;
    d.dx_si = xv.si;
    d.dx_st = xv.st;
  }
  d.c0g = L.c0g;
  d.hidden = L.hidden;
  d.ndir = L.ndir;
  d.nseq = c.nbp * (L.full ? c.nt : c.nf);
  d.q_inner = L.full ? c.nt : c.nf;
  d.nsteps = L.full ? c.nf : c.nt;
  d.wpack_bwd[0] = wbw[0];
  d.wpack_bwd[1] = L.ndir == 2 ? wbw[1] : nullptr;
  d.workspace = c.bwd_ws;
  d.workspace_bytes = c.bwd_ws_bytes;
  return fnssl_lstm_backward(&d, c.st);
}

// dW_ih = dA^T [x0 | x2], dW_hh = dA^T h_prev, db = sum dA, accumulated into the flat gradient.  Every operand is
// stored in the layer's natural layout, i.e. as a [rows = seq * step, C] matrix.
int weight_grads(const Ctx& c, const Layer& L, float* grad, const float* da, const float* x0, const float* x2,
                 const float* hout) {
  fnssl_wgrad_desc d;
  std::memset(&d, 0, sizeof(d));
  d.da = da;
  d.lda = (long long)L.ndir * 4 * L.hidden;
  d.x0 = L.c0 ? x0 : nullptr;
  d.ldx0 = L.c0;
  d.c0 = L.c0;
  d.x2 = L.c2 ? x2 : nullptr;
  d.ldx2 = L.c2;
  d.c2 = L.c2;
  d.h = hout;
  d.ldh = (long long)L.ndir * L.hidden;
  d.nseq = (long long)c.nbp * (L.full ? c.nt : c.nf);
  d.nsteps = L.full ? c.nf : c.nt;
  d.hidden = L.hidden;
  d.ndir = L.ndir;
  for (int di = 0; di < L.ndir; ++di) {
    d.g_wih[di] = grad + L.off_wih[di];
    d.g_whh[di] = grad + L.off_whh[di];
    d.g_bih[di] = grad + L.off_bih[di];
    d.g_bhh[di] = grad + L.off_bhh[di];
  }
  d.workspace = c.wgrad_ws;
  d.workspace_bytes = c.wgrad_ws_bytes;
  return fnssl_lstm_weight_grads(&d, c.st);
}

unsigned fmix32(unsigned h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
unsigned layer_seed(unsigned seed, int layer) { return fmix32(seed + 0x9E3779B9u * (unsigned)(layer + 1)); }

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

struct Plan {   // byte offsets into the workspace
  size_t xf, xn, f[3], u[3], v[3], n[3], x, res[6], pred, dpred, g, dn, da, dv[2], df, du[2], s, small, wgrad, lstm_ws,
      bwd_ws, fw[6][2], bw[6][2], total;
  size_t res_bytes[6], lstm_ws_bytes, bwd_ws_bytes, wgrad_bytes;
};

Plan make_plan(const fnssl_train* t, int nbp, int nf, int nt) {
  Plan pl;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += align256(bytes);
    return o;
  };
  const size_t P = (size_t)nbp * nt * nf * sizeof(float);
  const int nho = t->L[1].ndir * t->L[1].hidden;   // narrow-band output channels (256 either way)
  pl.xf = take(P * 4);
  pl.xn = take(P * 4);
  for (int k = 0; k < 3; ++k) {
    pl.f[k] = take(P * 2 * kHFull);
    pl.u[k] = take(k ? P * kCh : 0);
    pl.v[k] = take(P * kCh);
    pl.n[k] = take(P * nho);
  }
  pl.x = take(P * kCh);
  pl.lstm_ws_bytes = 0;
  pl.bwd_ws_bytes = 0;
  pl.wgrad_bytes = 0;
  for (int l = 0; l < 6; ++l) {
    const Layer& L = t->L[l];
    const int nseq = nbp * (L.full ? nt : nf), nsteps = L.full ? nf : nt;
    const size_t w3 = fnssl_lstm_weight_grads_workspace_bytes((long long)nseq * nsteps, L.hidden, L.ndir, L.c0, L.c2);
    if (w3 > pl.wgrad_bytes) pl.wgrad_bytes = w3;
    pl.res_bytes[l] = fnssl_lstm_reserve_bytes(nseq, L.hidden, L.ndir, nsteps);
    pl.res[l] = take(pl.res_bytes[l]);
    const size_t w1 = fnssl_lstm_workspace_bytes_ex(nseq, L.hidden, L.ndir, FNSSL_PRECISION_FP32), w2 = fnssl_lstm_bwd_workspace_bytes(nseq, L.hidden, L.ndir);
    if (w1 > pl.lstm_ws_bytes) pl.lstm_ws_bytes = w1;
    if (w2 > pl.bwd_ws_bytes) pl.bwd_ws_bytes = w2;
    for (int d = 0; d < L.ndir; ++d) {
      pl.fw[l][d] = take(L.fw_floats * sizeof(float));
      pl.bw[l][d] = take(L.bw_floats * sizeof(float));
    }
  }
  const size_t npred = (size_t)nbp * (nt / 12) * 2 * nf * sizeof(float);
  pl.pred = take(npred);
  pl.dpred = take(npred);
  pl.g = take(P * kCh);
  pl.dn = take(P * nho);
  pl.da = take(P * 1024);
  pl.dv[0] = take(P * t->L[1].ndir * kCh);
  pl.dv[1] = take(P * t->L[1].ndir * kCh);
  pl.df = take(P * 2 * kHFull);
  pl.du[0] = take(P * 2 * kCh);
  pl.du[1] = take(P * 2 * kCh);
  pl.s = take(t->is_online ? 0 : P * kCh);
  const size_t small = fnssl_head_backward_workspace_bytes();
  pl.small = take(small > 4096 ? small : 4096);
  pl.wgrad = take(pl.wgrad_bytes);
  pl.lstm_ws = take(pl.lstm_ws_bytes);
  pl.bwd_ws = take(pl.bwd_ws_bytes);
  pl.total = off + 256;
  return pl;
}

}  // namespace

extern "C" {

int fnssl_train_create(int is_online, fnssl_train** out) {
  FNSSL_REQUIRE(out, "train_create: null pointer");
  fnssl_train* t = new fnssl_train();
  t->is_online = is_online ? 1 : 0;
  const int nh = is_online ? 256 : 128, nd = is_online ? 1 : 2;
  static const char* names[6] = {"block_1.fullLstm", "block_1.narrLstm", "block_2.fullLstm",
                                 "block_2.narrLstm", "block_3.fullLstm", "block_3.narrLstm"};
  long long off = 1;   // element 0 of the flat vectors is a constant 0 (gather maps point "nowhere" there)
  for (int l = 0; l < 6; ++l) {
    Layer& L = t->L[l];
    const bool first = l < 2, full = (l % 2) == 0;
    L.name = names[l];
    L.full = full;
    if (full) {
      L.hidden = kHFull, L.ndir = 2, L.c0 = first ? 4 : kCh, L.c2 = 0, L.c0g = first ? 0 : kCh;
    } else {
      L.hidden = nh, L.ndir = nd, L.c0 = kCh, L.c2 = first ? 4 : 0, L.c0g = kCh;
    }
    const int I = L.c0 + L.c2, H = L.hidden;
    for (int d = 0; d < L.ndir; ++d) {   // nn.LSTM parameter order: w_ih, w_hh, b_ih, b_hh, then the _reverse set
      L.off_wih[d] = off;
      off += 4LL * H * I;
      L.off_whh[d] = off;
      off += 4LL * H * H;
      L.off_bih[d] = off;
      off += 4 * H;
      L.off_bhh[d] = off;
      off += 4 * H;
    }
    L.fw_floats = fnssl_lstm_packed_floats(L.c0, L.c2, H);
    L.bw_floats = fnssl_lstm_bwd_packed_floats(L.c0g, H);
  }
  t->off_emb_w = off;
  off += 2 * 256;
  t->off_emb_b = off;
  off += 2;
  t->nparam = off - 1;
  // gather maps: run the host packers on index-valued weights (indices < 2^24 are exact in fp32)
  for (int l = 0; l < 6; ++l) {
    Layer& L = t->L[l];
    const int I = L.c0 + L.c2, H = L.hidden;
    std::vector<float> wi((size_t)4 * H * I), wh((size_t)4 * H * H), bi(4 * H), bh(4 * H), z(4 * H, 0.f), zi(wi.size(), 0.f),
        zh(wh.size(), 0.f), pa(L.fw_floats), pb(L.fw_floats), pw(L.bw_floats);
    for (int d = 0; d < L.ndir; ++d) {
      for (size_t i = 0; i < wi.size(); ++i) wi[i] = (float)(L.off_wih[d] + (long long)i);
      for (size_t i = 0; i < wh.size(); ++i) wh[i] = (float)(L.off_whh[d] + (long long)i);
      for (int i = 0; i < 4 * H; ++i) bi[i] = (float)(L.off_bih[d] + i), bh[i] = (float)(L.off_bhh[d] + i);
      int rc = fnssl_lstm_pack(wi.data(), wh.data(), bi.data(), z.data(), L.c0, L.c2, H, pa.data());
      if (rc == FNSSL_OK) rc = fnssl_lstm_pack(zi.data(), zh.data(), z.data(), bh.data(), L.c0, L.c2, H, pb.data());
      if (rc == FNSSL_OK) rc = fnssl_lstm_pack_bwd(wi.data(), wh.data(), I, L.c0g, H, pw.data());
      if (rc != FNSSL_OK) {
        delete t;
        return rc;
      }
      L.map_fw_a[d] = t->maps.size();
      for (float v : pa) t->maps.push_back((int)v);
      L.map_fw_b[d] = t->maps.size();
      for (float v : pb) t->maps.push_back((int)v);
      L.map_bw[d] = t->maps.size();
      for (float v : pw) t->maps.push_back((int)v);
    }
  }
  *out = t;
  return FNSSL_OK;
}

void fnssl_train_destroy(fnssl_train* t) { delete t; }

long long fnssl_train_param_floats(const fnssl_train* t) { return t ? 1 + t->nparam : 0; }

/* what = 0 weight_ih, 1 weight_hh, 2 bias_ih, 3 bias_hh of LSTM `layer` (0..5), direction dir; layer 6: emb2ipd weight
 * (what 0) / bias (what 1).  Returns the offset in floats into the flat vectors, or -1. */
long long fnssl_train_param_offset(const fnssl_train* t, int layer, int dir, int what) {
  if (!t) return -1;
  if (layer == 6) return what == 0 ? t->off_emb_w : (what == 1 ? t->off_emb_b : -1);
  if (layer < 0 || layer > 5 || dir < 0 || dir >= t->L[layer].ndir) return -1;
  const Layer& L = t->L[layer];
  return what == 0 ? L.off_wih[dir] : what == 1 ? L.off_whh[dir] : what == 2 ? L.off_bih[dir] : what == 3 ? L.off_bhh[dir] : -1;
}

size_t fnssl_train_map_bytes(const fnssl_train* t) { return t ? t->maps.size() * sizeof(int) : 0; }

int fnssl_train_upload_maps(fnssl_train* t, void* dev_buf, void* stream) {
  FNSSL_REQUIRE(t && dev_buf, "train_upload_maps: null pointer");
  FNSSL_HIP(hipMemcpyAsync(dev_buf, t->maps.data(), t->maps.size() * sizeof(int), hipMemcpyHostToDevice, fnssl::as_stream(stream)));
  FNSSL_HIP(hipStreamSynchronize(fnssl::as_stream(stream)));   // the host vector may not outlive an async copy
  t->dmaps = static_cast<const int*>(dev_buf);
  return FNSSL_OK;
}

size_t fnssl_train_workspace_bytes(const fnssl_train* t, int nbp, int nf, int nt) {
  if (!t || nbp <= 0 || nf <= 0 || nt < 12) return 0;
  return make_plan(t, nbp, nf, nt).total;
}

int fnssl_train_backward(fnssl_train* t, const float* theta, float* grad, const float* x, const float* gt, int nb,
                         int np, int nf, int nt, unsigned seed_base, long long pair0, long long n_total, float* loss,
                         void* workspace, size_t workspace_bytes, void* stream) {
  FNSSL_REQUIRE(t && theta && grad && x && gt && loss, "train_backward: null pointer");
  FNSSL_REQUIRE(t->dmaps, "train_backward: call fnssl_train_upload_maps first");
  FNSSL_REQUIRE(nb > 0 && np > 0 && nf > 0 && nt >= 12, "train_backward: empty problem");
  const int nbp = nb * np, nt2 = nt / 12;
  const Plan pl = make_plan(t, nbp, nf, nt);
  if (!workspace || workspace_bytes < pl.total) {
    fnssl::set_error("train_backward: workspace %zu < %zu bytes", workspace_bytes, pl.total);
    return FNSSL_E_WORKSPACE;
  }
  hipStream_t st = fnssl::as_stream(stream);
  char* base = reinterpret_cast<char*>((reinterpret_cast<size_t>(workspace) + 255) / 256 * 256);
  auto F = [&](size_t o) { return reinterpret_cast<float*>(base + o); };
  Ctx c{st, nbp, nt, nf, F(pl.wgrad), pl.wgrad_bytes, F(pl.lstm_ws), pl.lstm_ws_bytes, F(pl.bwd_ws), pl.bwd_ws_bytes};
  unsigned seeds[6];
  for (int l = 0; l < 6; ++l) seeds[l] = layer_seed(seed_base, l);

  // weight streams from the current flat parameters (two gathers per forward stream)
  const float* fw[6][2] = {};
  const float* bw[6][2] = {};
  for (int l = 0; l < 6; ++l)
    for (int d = 0; d < t->L[l].ndir; ++d) {
      const Layer& L = t->L[l];
      hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((L.fw_floats + 255) / 256)), dim3(256), 0, st, theta,
                         t->dmaps + L.map_fw_a[d], t->dmaps + L.map_fw_b[d], (long long)L.fw_floats, F(pl.fw[l][d]));
      hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((L.bw_floats + 255) / 256)), dim3(256), 0, st, theta,
                         t->dmaps + L.map_bw[d], (const int*)nullptr, (long long)L.bw_floats, F(pl.bw[l][d]));
      FNSSL_CHECK_LAUNCH("gather_kernel");
      fw[l][d] = F(pl.fw[l][d]);
      bw[l][d] = F(pl.bw[l][d]);
    }

  const int nho = t->L[1].ndir * t->L[1].hidden, ndn = t->L[1].ndir;
  const T4 XF = make_F(F(pl.xf), nt, nf, 4), XN = make_N(F(pl.xn), nt, nf, 4);
  TRY(fnssl_nchw_to_seq(x, nbp, 4, nf, nt, XF.p, st));
  TRY(combine(c, XN, {}, {XF}, false, 0, 0));   // the same numbers stored [b, f, t, 4]
  T4 Fk[3], Uk[3], Vk[3], Nk[3];
  const T4 X = make_N(F(pl.x), nt, nf, kCh);
  // ---------------- forward (train mode) ----------------
  for (int k = 0; k < 3; ++k) {
    const Layer &lf = t->L[2 * k], &ln = t->L[2 * k + 1];
    Fk[k] = make_F(F(pl.f[k]), nt, nf, 2 * kHFull);
    Vk[k] = make_N(F(pl.v[k]), nt, nf, kCh);
    Nk[k] = make_N(F(pl.n[k]), nt, nf, nho);
    if (k == 0) {
      TRY(lstm_fwd(c, lf, XF, nullptr, fw[0], Fk[0], F(pl.res[0]), pl.res_bytes[0]));
      TRY(combine(c, Vk[0], {Fk[0]}, {}, true, seeds[0], pair0));                        // dropout_full (:40)
    } else {
      Uk[k] = make_F(F(pl.u[k]), nt, nf, kCh);
      TRY(combine(c, Uk[k], {}, {X, Fk[k - 1]}, false, 0, 0));                            // x + fb_skip (:36-37)
      TRY(lstm_fwd(c, lf, Uk[k], nullptr, fw[2 * k], Fk[k], F(pl.res[2 * k]), pl.res_bytes[2 * k]));
      TRY(combine(c, Vk[k], {Fk[k]}, {X}, true, seeds[2 * k], pair0));                    // + nb_skip (:44-45)
    }
    TRY(lstm_fwd(c, ln, Vk[k], k == 0 ? &XN : nullptr, fw[2 * k + 1], Nk[k], F(pl.res[2 * k + 1]), pl.res_bytes[2 * k + 1]));
    TRY(combine(c, X, {Nk[k]}, {}, true, seeds[2 * k + 1], pair0));                       // dropout_narr (:48)
  }
  const float* emb_w = theta + t->off_emb_w;
  const float* emb_b = theta + t->off_emb_b;
  float* pred = F(pl.pred);
  float* dpred = F(pl.dpred);
  TRY(fnssl_head(X.p, nbp, nf, nt, emb_w, emb_b, pred, st));
  TRY(fnssl_mse_loss(pred, gt, nb, np, nt2, 2 * nf, n_total, dpred, loss, 1, F(pl.small), 4096, st));
  // ---------------- backward ----------------
  const T4 G = make_N(F(pl.g), nt, nf, kCh);
  TRY(fnssl_head_backward(X.p, emb_w, pred, dpred, nbp, nf, nt, G.p, grad + t->off_emb_w, grad + t->off_emb_b, 1, F(pl.small),
                          fnssl_head_backward_workspace_bytes(), st));
  T4 gx[3] = {G};
  int ngx = 1;
  T4 dfb[2];
  int ndfb = 0;
  for (int k = 2; k >= 0; --k) {
    const Layer &lf = t->L[2 * k], &ln = t->L[2 * k + 1];
    const T4 DN = make_N(F(pl.dn), nt, nf, nho);
    {   // dropout_narr backward: DN = mask * sum(gx)
      fnssl_btf_view mv[3];
      for (int i = 0; i < ngx; ++i) mv[i] = btf(gx[i]);
      TRY(fnssl_train_combine(DN.p, DN.sb, DN.st, DN.sf, nbp, nt, nf, nho, mv, ngx, nullptr, 0, 1, seeds[2 * k + 1], pair0, st));
    }
    const T4 dA_n = make_N(F(pl.da), nt, nf, ndn * 4 * ln.hidden);
    const T4 DV = make_N(F(pl.dv[k & 1]), nt, nf, ndn * kCh);
    TRY(lstm_bwd(c, ln, F(pl.res[2 * k + 1]), DN, dA_n, &DV, bw[2 * k + 1]));
    TRY(weight_grads(c, ln, grad, dA_n.p, Vk[k].p, k == 0 ? XN.p : nullptr, Nk[k].p));
    const T4 DF = make_F(F(pl.df), nt, nf, 2 * kHFull);
    {   // dropout_full backward + fb_skip: DF = mask * sum(dv slabs) + sum(dfb)
      fnssl_btf_view mv[2], pv[2];
      for (int d = 0; d < ndn; ++d) mv[d] = btf(chan(DV, d * kCh, kCh));
      for (int i = 0; i < ndfb; ++i) pv[i] = btf(dfb[i]);
      TRY(fnssl_train_combine(DF.p, DF.sb, DF.st, DF.sf, nbp, nt, nf, 2 * kHFull, mv, ndn, pv, ndfb, 1, seeds[2 * k], pair0, st));
    }
    const T4 dA_f = make_F(F(pl.da), nt, nf, 2 * 4 * kHFull);
    if (k > 0) {
      const T4 DU = make_F(F(pl.du[k & 1]), nt, nf, 2 * kCh);
      TRY(lstm_bwd(c, lf, F(pl.res[2 * k]), DF, dA_f, &DU, bw[2 * k]));
      TRY(weight_grads(c, lf, grad, dA_f.p, Uk[k].p, nullptr, Fk[k].p));
      T4 du[2] = {chan(DU, 0, kCh), chan(DU, kCh, kCh)};
      int ndu = 2;
      if (ndn + 2 > 3) {   // offline narrow-band: 2 + 2 operands -> fold the full-band pair first
        const T4 S = make_F(F(pl.s), nt, nf, kCh);
        TRY(combine(c, S, {}, {du[0], du[1]}, false, 0, 0));
        du[0] = S;
        ndu = 1;
      }
      ngx = 0;   // dL/dX_k = dV_k + dU_k (both uses of x: nb_skip and the full-band input)
      for (int d = 0; d < ndn; ++d) gx[ngx++] = chan(DV, d * kCh, kCh);
      for (int i = 0; i < ndu; ++i) gx[ngx++] = du[i];
      ndfb = ndu;   // dL/dF_{k-1} through fb_skip
      for (int i = 0; i < ndu; ++i) dfb[i] = du[i];
    } else {
      TRY(lstm_bwd(c, lf, F(pl.res[0]), DF, dA_f, nullptr, bw[0]));
      TRY(weight_grads(c, lf, grad, dA_f.p, XF.p, nullptr, Fk[0].p));
    }
  }
  return FNSSL_OK;
}

int fnssl_train_step(fnssl_train* t, float* theta, float* grad, float* exp_avg, float* exp_avg_sq, const float* x,
                     const float* gt, int nb, int np, int nf, int nt, unsigned seed_base, float lr, float beta1,
                     float beta2, float eps, int step, float* loss, void* workspace, size_t workspace_bytes,
                     void* stream) {
  FNSSL_REQUIRE(t && theta && grad && exp_avg && exp_avg_sq && loss, "train_step: null pointer");
  const long long n = 1 + t->nparam;
  FNSSL_HIP(hipMemsetAsync(grad, 0, n * sizeof(float), fnssl::as_stream(stream)));
  FNSSL_HIP(hipMemsetAsync(loss, 0, sizeof(float), fnssl::as_stream(stream)));
  const long long n_total = (long long)nb * np * (nt / 12) * 2 * nf;
  TRY(fnssl_train_backward(t, theta, grad, x, gt, nb, np, nf, nt, seed_base, 0, n_total, loss, workspace, workspace_bytes, stream));
  TRY(fnssl_adam_step(theta, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, 1.f, stream));
  FNSSL_HIP(hipMemsetAsync(theta, 0, sizeof(float), fnssl::as_stream(stream)));   // element 0 stays the constant 0
  return FNSSL_OK;
}

}  // extern "C"
