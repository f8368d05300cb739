// Element-wise / reduction kernels of the training step (reference
// FN-SSL/Lightning/main.py:149-157 training_step, :191-198 cal_loss, :269-271 Adam; the dropouts and
// residual adds of FNblock.forward, FN-SSL/Model.py:36-48).  All HBM-bound: one pass over their operands,
// float4 accesses, deterministic reductions (two stages, no float atomics).
//
//   keep_scale         dropout keep-scale {0, 1.25} as a pure function of (layer seed, logical element index)
//   combine_kernel     out = keep_scale ⊙ (a_0 + a_1 + a_2) + (b_0 + b_1): every dropout + residual add of
//                      the forward, every gradient accumulation + dropout-backward of the backward, with
//                      each operand read through its own (batch, frame, bin) strides, so the full-band <->
//                      narrow-band layout change rides along
//   head_bwd           tanh' + Linear(256, 2)^T + AvgPool(12)^T, and the per-block partial dW / db
//   mse_kernel         loss and d loss / d pred of the re-batched prediction
//   adam_kernel        torch.optim.Adam (no amsgrad / weight decay) on the flat parameter vector
#include "common.h"

namespace {

__device__ __forceinline__ unsigned fmix32(unsigned h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

// keep probability 0.8 (p = 0.2, Model.py:9), scale 1 / 0.8; idx = ((b*nt + t)*nf + f)*C + c
__device__ __forceinline__ float keep_scale(unsigned seed, unsigned long long idx) {
  const unsigned lo = (unsigned)idx, hi = (unsigned)(idx >> 32);
  unsigned h = fmix32((lo * 0xCC9E2D51u) ^ seed);
  h = fmix32(h + hi * 0x1B873593u + seed * 0x85EBCA6Bu);
  return (h >> 8) < 13421773u ? 1.25f : 0.0f;
}

struct BtfView {
  const float* p;
  long long sb, st, sf;
};

struct CombineParams {
  float* out;
  long long o_sb, o_st, o_sf;
  BtfView m[3];   // masked operands
  BtfView b[2];   // plain operands
  int nm, np;
  int nb, nt, nf, c4;   // c4 = C / 4
  int use_mask;
  unsigned seed;
  long long b0;
};

__global__ void __launch_bounds__(256) combine_kernel(const CombineParams p) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)p.nb * p.nt * p.nf * p.c4;
  if (gid >= total) return;
  const int c = (int)(gid % p.c4);
  long long r = gid / p.c4;
  const int f = (int)(r % p.nf);
  r /= p.nf;
  const int t = (int)(r % p.nt);
  const int b = (int)(r / p.nt);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = 0; i < p.nm; ++i) {
    const float4 v = *reinterpret_cast<const float4*>(p.m[i].p + b * p.m[i].sb + t * p.m[i].st + f * p.m[i].sf + 4 * c);
    acc.x += v.x;
    acc.y += v.y;
    acc.z += v.z;
    acc.w += v.w;
  }
  if (p.use_mask) {
    const unsigned long long idx = ((((unsigned long long)(b + p.b0) * p.nt + t) * p.nf + f) * p.c4 + c) * 4ull;
    acc.x *= keep_scale(p.seed, idx);
    acc.y *= keep_scale(p.seed, idx + 1);
    acc.z *= keep_scale(p.seed, idx + 2);
    acc.w *= keep_scale(p.seed, idx + 3);
  }
  for (int i = 0; i < p.np; ++i) {
    const float4 v = *reinterpret_cast<const float4*>(p.b[i].p + b * p.b[i].sb + t * p.b[i].st + f * p.b[i].sf + 4 * c);
    acc.x += v.x;
    acc.y += v.y;
    acc.z += v.z;
    acc.w += v.w;
  }
  *reinterpret_cast<float4*>(p.out + b * p.o_sb + t * p.o_st + f * p.o_sf + 4 * c) = acc;
}

constexpr int kSeg = FNSSL_SEG_FRAMES;   // 12
constexpr int kHeadBlocks = 1024;

// One (utterance-pair, bin, segment) item per loop iteration, one channel per thread (256 channels).
//   dz_o = dpred[b, s, f + o*nf] * (1 - pred^2);  dx[b, f, 12s + k, c] = (dz_0 W[0,c] + dz_1 W[1,c]) / 12
//   partial dW[o, c] += dz_o * mean_k x[b, f, 12s + k, c];  partial db[o] += dz_o
__global__ void __launch_bounds__(256)
head_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ pred,
                const float* __restrict__ dpred, int nb, int nf, int nt, int nt2, float* __restrict__ dx,
                float* __restrict__ part) {
  const int c = threadIdx.x;
  const float w0 = w[c], w1 = w[256 + c];
  float dw0 = 0.f, dw1 = 0.f, db0 = 0.f, db1 = 0.f;
  const long long items = (long long)nb * nf * nt2;
  for (long long it = blockIdx.x; it < items; it += gridDim.x) {
    const int s = (int)(it % nt2);
    const long long bf = it / nt2;
    const int f = (int)(bf % nf);
    const long long b = bf / nf;
    const long long po = (b * nt2 + s) * (2ll * nf) + f;
    const float p0 = pred[po], p1 = pred[po + nf];
    const float dz0 = dpred[po] * (1.f - p0 * p0), dz1 = dpred[po + nf] * (1.f - p1 * p1);
    const float g = (dz0 * w0 + dz1 * w1) * (1.0f / kSeg);
    const float* xr = x + ((b * nf + f) * nt + (long long)s * kSeg) * 256 + c;
    float* dr = dx + ((b * nf + f) * nt + (long long)s * kSeg) * 256 + c;
    float pool = 0.f;
#pragma unroll
    for (int k = 0; k < kSeg; ++k) {
      pool += xr[k * 256];
      dr[k * 256] = g;
    }
    pool *= (1.0f / kSeg);
    dw0 += dz0 * pool;
    dw1 += dz1 * pool;
    db0 += dz0;
    db1 += dz1;
  }
  float* pp = part + (long long)blockIdx.x * 520;
  pp[c] = dw0;
  pp[256 + c] = dw1;
  if (c == 0) {
    pp[512] = db0;
    pp[513] = db1;
  }
}

// frames beyond 12 * nt2 never reach the head: zero gradient
__global__ void __launch_bounds__(256)
head_tail_zero_kernel(float* __restrict__ dx, long long rows, int nt, int nt2) {
  const int tail = nt - nt2 * kSeg;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;   // over rows * tail * 64 float4
  if (gid >= rows * tail * 64) return;
  const int c4 = (int)(gid & 63);
  const long long r = gid >> 6;
  const int k = (int)(r % tail);
  const long long row = r / tail;
  reinterpret_cast<float4*>(dx)[(row * nt + nt2 * kSeg + k) * 64 + c4] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// dst[j] (+)= sum_i part[i * stride + j]   (fixed order: deterministic)
__global__ void __launch_bounds__(256)
reduce_parts_kernel(const float* __restrict__ part, int nparts, int stride, int n, float* __restrict__ dst,
                    int accumulate, float scale) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  float s = 0.f;
  for (int i = 0; i < nparts; ++i) s += part[(long long)i * stride + j];
  s *= scale;
  dst[j] = accumulate ? dst[j] + s : s;
}

constexpr int kMseBlocks = 256;

// pred [nb*np, nt2, nf2]; gt [nb, nt2, nf2, np]; dpred = gscale * (pred - gt); partial sums of squares
__global__ void __launch_bounds__(256)
mse_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int nb, int np, int nt2, int nf2,
           float gscale, float* __restrict__ dpred, float* __restrict__ part) {
  __shared__ float red[256];
  const long long total = (long long)nb * np * nt2 * nf2;
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k = (int)(i % nf2);
    long long r = i / nf2;
    const int seg = (int)(r % nt2);
    r /= nt2;
    const int pi = (int)(r % np);
    const long long b = r / np;
    const float d = pred[i] - gt[((b * nt2 + seg) * nf2 + k) * np + pi];
    dpred[i] = gscale * d;
    s += d * d;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
            long long n, float gscale, float beta1, float beta2, float step_size, float inv_sqrt_bc2, float eps) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gr = g[i] * gscale;
  const float mi = m[i] + (gr - m[i]) * (1.f - beta1);          // exp_avg.lerp_(grad, 1 - beta1)
  const float vi = v[i] * beta2 + (1.f - beta2) * gr * gr;      // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
  p[i] = p[i] - step_size * (mi / denom);
}

__global__ void __launch_bounds__(256)
dropout_scale_kernel(float* __restrict__ out, long long n, unsigned seed, long long offset) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = keep_scale(seed, (unsigned long long)(i + offset));
}

}  // namespace

extern "C" {

int fnssl_train_combine(float* out, long long o_sb, long long o_st, long long o_sf, int nb, int nt, int nf, int c,
                        const fnssl_btf_view* masked, int n_masked, const fnssl_btf_view* plain, int n_plain,
                        int use_mask, unsigned seed32, long long b0, void* stream) {
  FNSSL_REQUIRE(out && nb > 0 && nt > 0 && nf > 0 && c > 0 && c % 4 == 0, "train_combine: bad shape");
  FNSSL_REQUIRE(n_masked >= 0 && n_masked <= 3 && n_plain >= 0 && n_plain <= 2 && n_masked + n_plain > 0 &&
                    (n_masked == 0 || masked) && (n_plain == 0 || plain),
                "train_combine: up to 3 masked and 2 plain operands");
  CombineParams p;
  p.out = out;
  p.o_sb = o_sb;
  p.o_st = o_st;
  p.o_sf = o_sf;
  auto ok = [](const void* q, long long a, long long b, long long cc) {
    return q && (reinterpret_cast<uintptr_t>(q) & 15) == 0 && !(a & 3) && !(b & 3) && !(cc & 3);
  };
  FNSSL_REQUIRE(ok(out, o_sb, o_st, o_sf), "train_combine: output must be 16-byte aligned, strides multiples of 4");
  for (int i = 0; i < n_masked; ++i) {
    FNSSL_REQUIRE(ok(masked[i].p, masked[i].sb, masked[i].st, masked[i].sf), "train_combine: bad masked operand %d", i);
    p.m[i] = BtfView{masked[i].p, masked[i].sb, masked[i].st, masked[i].sf};
  }
  for (int i = 0; i < n_plain; ++i) {
    FNSSL_REQUIRE(ok(plain[i].p, plain[i].sb, plain[i].st, plain[i].sf), "train_combine: bad plain operand %d", i);
    p.b[i] = BtfView{plain[i].p, plain[i].sb, plain[i].st, plain[i].sf};
  }
  p.nm = n_masked;
  p.np = n_plain;
  p.nb = nb;
  p.nt = nt;
  p.nf = nf;
  p.c4 = c / 4;
  p.use_mask = use_mask;
  p.seed = seed32;
  p.b0 = b0;
  const long long total = (long long)nb * nt * nf * p.c4;
  FNSSL_REQUIRE((total + 255) / 256 < (1ll << 31), "train_combine: too large");
  fnssl::TimedLaunch tl("train_combine", fnssl::as_stream(stream));
  hipLaunchKernelGGL(combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, fnssl::as_stream(stream), p);
  FNSSL_CHECK_LAUNCH("combine_kernel");
  return FNSSL_OK;
}

int fnssl_dropout_scale(float* out, long long n, unsigned seed32, long long offset, void* stream) {
  FNSSL_REQUIRE(out && n > 0 && offset >= 0, "dropout_scale: bad arguments");
  hipLaunchKernelGGL(dropout_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, fnssl::as_stream(stream),
                     out, n, seed32, offset);
  FNSSL_CHECK_LAUNCH("dropout_scale_kernel");
  return FNSSL_OK;
}

size_t fnssl_head_backward_workspace_bytes(void) { return (size_t)kHeadBlocks * 520 * sizeof(float); }

int fnssl_head_backward(const float* x, const float* w, const float* pred, const float* dpred, int nb, int nf, int nt,
                        float* dx, float* dw, float* db, int accumulate, void* workspace, size_t workspace_bytes,
                        void* stream) {
  FNSSL_REQUIRE(x && w && pred && dpred && dx && dw && db, "head_backward: null pointer");
  FNSSL_REQUIRE(nb > 0 && nf > 0 && nt > 0, "head_backward: empty problem");
  FNSSL_REQUIRE(workspace && workspace_bytes >= fnssl_head_backward_workspace_bytes(), "head_backward: workspace too small");
  const int nt2 = nt / kSeg;
  hipStream_t st = fnssl::as_stream(stream);
  fnssl::TimedLaunch tl("head_backward", st);
  float* part = static_cast<float*>(workspace);
  const long long items = (long long)nb * nf * nt2;
  const int nblk = (int)(items < kHeadBlocks ? (items > 0 ? items : 1) : kHeadBlocks);
  if (items > 0) {
    hipLaunchKernelGGL(head_bwd_kernel, dim3(nblk), dim3(256), 0, st, x, w, pred, dpred, nb, nf, nt, nt2, dx, part);
    FNSSL_CHECK_LAUNCH("head_bwd_kernel");
  } else {
    FNSSL_HIP(hipMemsetAsync(part, 0, 520 * sizeof(float), st));
  }
  if (nt > nt2 * kSeg) {
    const long long n = (long long)nb * nf * (nt - nt2 * kSeg) * 64;
    hipLaunchKernelGGL(head_tail_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dx,
                       (long long)nb * nf, nt, nt2);
    FNSSL_CHECK_LAUNCH("head_tail_zero_kernel");
  }
  hipLaunchKernelGGL(reduce_parts_kernel, dim3(2), dim3(256), 0, st, part, nblk, 520, 512, dw, accumulate, 1.0f);
  hipLaunchKernelGGL(reduce_parts_kernel, dim3(1), dim3(256), 0, st, part + 512, nblk, 520, 2, db, accumulate, 1.0f);
  FNSSL_CHECK_LAUNCH("reduce_parts_kernel");
  return FNSSL_OK;
}

int fnssl_mse_loss(const float* pred, const float* gt, int nb, int np, int nt2, int nf2, long long n_total,
                   float* dpred, float* loss, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  FNSSL_REQUIRE(pred && gt && dpred && loss, "mse_loss: null pointer");
  FNSSL_REQUIRE(nb > 0 && np > 0 && nt2 > 0 && nf2 > 0 && n_total > 0, "mse_loss: empty problem");
  FNSSL_REQUIRE(workspace && workspace_bytes >= kMseBlocks * sizeof(float), "mse_loss: workspace too small");
  hipStream_t st = fnssl::as_stream(stream);
  fnssl::TimedLaunch tl("mse_loss", st);
  float* part = static_cast<float*>(workspace);
  const long long total = (long long)nb * np * nt2 * nf2;
  const int nblk = (int)((total + 255) / 256 < kMseBlocks ? (total + 255) / 256 : kMseBlocks);
  hipLaunchKernelGGL(mse_kernel, dim3(nblk), dim3(256), 0, st, pred, gt, nb, np, nt2, nf2, 2.0f / (float)n_total, dpred,
                     part);
  hipLaunchKernelGGL(reduce_parts_kernel, dim3(1), dim3(256), 0, st, part, nblk, 1, 1, loss, accumulate,
                     1.0f / (float)n_total);
  FNSSL_CHECK_LAUNCH("mse_kernel");
  return FNSSL_OK;
}

int fnssl_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                    float beta1, float beta2, float eps, int step, float grad_scale, void* stream) {
  FNSSL_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_step: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  fnssl::TimedLaunch tl("adam", fnssl::as_stream(stream));
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, fnssl::as_stream(stream), param, grad,
                     exp_avg, exp_avg_sq, n, grad_scale, beta1, beta2, (float)((double)lr / bc1),
                     (float)(1.0 / sqrt(bc2)), eps);
  FNSSL_CHECK_LAUNCH("adam_kernel");
  return FNSSL_OK;
}

}  // extern "C"
