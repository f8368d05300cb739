/*
 * fnssl.h — C ABI of libfnssl_hip.so: the MI355X (gfx950) implementation of the
 * FN-SSL DP-IPD forward path.
 *
 * The reference (Audio-WestlakeU/FN-SSL) is pure PyTorch and has no FFI of its
 * own; every entry point below replaces the PyTorch call sequence named in its
 * comment (paths relative to the reference root).  INTEGRATION.md shows the
 * ctypes stub a maintainer adds on the reference side.
 *
 * Conventions (all entry points):
 *   - plain C types only: raw DEVICE pointers (fp32 unless stated), ints, a
 *     `void* stream` that is a hipStream_t (0 = the null stream);
 *   - the caller owns every buffer; the library never allocates or frees
 *     device memory and never synchronises the device: all work is enqueued on
 *     `stream` and the call returns immediately;
 *   - return 0 on success, <0 on error (FNSSL_E_*); `fnssl_last_error()` gives
 *     a thread-local message.  Nothing throws across the ABI;
 *   - thread-safe for concurrent calls on distinct streams; one process per GPU.
 *   - the library reads NO environment variable: kernel-family overrides, A/B knobs and the fault-injection
 *     hook travel in a caller-owned `fnssl_tuning` (below) — the process default set through
 *     fnssl_tuning_set(), or per call through the `tuning` field of the LSTM / network descriptors.
 *   - "pack" functions are host-only (no GPU needed) and write HOST memory.
 */
#ifndef FNSSL_H_
#define FNSSL_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FNSSL_ABI_VERSION 19

#define FNSSL_OK 0
#define FNSSL_E_INVALID (-1)     /* bad argument / unsupported shape          */
#define FNSSL_E_HIP (-2)         /* a HIP runtime call or kernel launch failed */
#define FNSSL_E_WORKSPACE (-3)   /* workspace too small                       */

#define FNSSL_CH_MODE_M 0        /* pairs (0, j)            Module.py:387-393 */
#define FNSSL_CH_MODE_MM 1       /* pairs (i, j), i < j     Module.py:397-402 */

#define FNSSL_WIN_LEN 512
#define FNSSL_HOP 256
#define FNSSL_NBIN 257           /* rFFT bins incl. DC                        */
#define FNSSL_NF 256             /* bins kept by the network (1..256)         */
#define FNSSL_SEG_FRAMES 12      /* AvgPool2d((12,1))       Model.py:68       */

int fnssl_abi_version(void);
const char* fnssl_last_error(void);

/* ------------------------------------------------------------------------- */
/* Tuning: explicit and caller-owned (the library reads no environment)       */
/* ------------------------------------------------------------------------- */

/* Knob indices.  0 is every knob's default; all of them select between kernels that produce the SAME bits
 * (A/B references, launch geometries) except the fault-injection hook, which the fallback tests use. */
#define FNSSL_TUNE_LSTM_NO_STATIC 0           /* per-wave LSTM rounds: the generic run-time-loop kernels instead of the shape-specialised ones */
#define FNSSL_TUNE_NO_STATIC3 1               /* H = 256 narrow-band layers at full-chip size: the one-slice kernel (lstm_static.h) instead of the operand-ring kernel (lstm_static3.h) */
#define FNSSL_TUNE_NO_STATIC2 2               /* (unused since round 5: lstm_static2.h was removed; the slot keeps the indices stable) */
#define FNSSL_TUNE_NO_STATIC_IPDNET 3         /* IPDnet layer shapes on the generic kernels */
#define FNSSL_TUNE_LSTM_SPLIT 4               /* 1 / 2 / 4: force the waves-per-group split of small launches */
#define FNSSL_TUNE_SPLIT4_MAX_H256 5          /* largest groups-per-CU count that still takes 4 waves per group at H = 256 (default 6) */
#define FNSSL_TUNE_LSTM_VARIANT_H128 6        /* force a launch geometry (kernel variants) for H = 128 */
#define FNSSL_TUNE_LSTM_VARIANT_H256 7        /* ... for H = 256 */
#define FNSSL_TUNE_LSTM_CHQ 8                 /* force the weight-ring chunk (quads) */
#define FNSSL_TUNE_NO_F32_CLUSTER 9           /* H = 128 full-band layers on the per-wave rounds instead of the cluster-resident kernel (lstm_f32c.h) */
#define FNSSL_TUNE_NO_F32C_B1 10              /* ... block 1's layer only */
#define FNSSL_TUNE_TRAIN_NO_F32_CLUSTER 11    /* ... the training forward only */
#define FNSSL_TUNE_F32C_NO_ROTATE 12          /* cluster kernel: leftover groups stay with one wave */
#define FNSSL_TUNE_F32C_PRIO 13               /* 9 = no issue priorities in the cluster kernel */
#define FNSSL_TUNE_NO_CLUSTER 14              /* bf16 IPDnet layers on the pair-split kernels instead of the cluster-resident ones (lstm_bf16c.h) */
#define FNSSL_TUNE_NO_CLUSTER_B1 15           /* ... block 1's narrow-band layer only */
#define FNSSL_TUNE_NO_CLUSTER_H128 16         /* ... the H = 128 layers only */
#define FNSSL_TUNE_CLUSTER_SPREAD 17          /* members of a bf16 cluster on different XCDs (placement is speed only) */
#define FNSSL_TUNE_BF16P_DRAIN 18             /* pair-split bf16 kernels: drained ring barriers on every shape */
#define FNSSL_TUNE_BF16W_SOLO 19              /* wide bf16 layers: one wave per group instead of the pair split */
#define FNSSL_TUNE_BWD_NO_CLUSTER 20          /* H = 128 BPTT on the split kernels instead of the cluster-resident kernel (lstm_bwdc.h) */
#define FNSSL_TUNE_BWD_CLUSTER_MIN_GROUPS 21  /* smallest groups-per-cluster count the cluster BPTT takes (default 1; round 4: 8) */
#define FNSSL_TUNE_BWD_CLUSTER_NO_ROTATE 22   /* cluster BPTT: leftover groups stay with one wave */
#define FNSSL_TUNE_BWDC_NO_PREFETCH 23        /* cluster BPTT: phase-A operands requested in their own group-step */
#define FNSSL_TUNE_BWDC_NO_TOKEN 24           /* cluster BPTT: no per-SIMD matrix-phase token */
#define FNSSL_TUNE_BWDC_WAVES16 25            /* cluster BPTT: the 16-wave / 4-deep-ring shape */
#define FNSSL_TUNE_FWD_RING 26                /* training forward, 4 waves per group: weights through the LDS ring instead of direct streams */
#define FNSSL_TUNE_BWD_RING 27                /* BPTT, 4 waves per group: the same */
#define FNSSL_TUNE_TRAIN_SPLIT 28             /* 1 / 2 / 4: force the waves-per-group split of the training kernels */
#define FNSSL_TUNE_TRAIN_NO_STATIC 29         /* training forward on the generic kernels */
#define FNSSL_TUNE_NO_FWD2 30                 /* narrow-band training forward: 4 waves per group instead of two groups per wave set (lstm_fwd2.h) */
#define FNSSL_TUNE_NO_BWD2 31                 /* narrow-band BPTT: the same (lstm_bwd2.h) */
#define FNSSL_TUNE_SN_SCALAR 32               /* IPDnet2 fp32: scalar-operand kernels instead of the matrix-pipe ones */
#define FNSSL_TUNE_STFT_PER_FRAME 33          /* front end: one wave per frame instead of the persistent row kernel */
#define FNSSL_TUNE_CLUSTER_SPIN_LIMIT 34      /* bounded-wait length of the cluster kernels' hand-offs (default 2^20 spins, about 1.5 s) */
#define FNSSL_TUNE_CLUSTER_TEST_STALL 35      /* FAULT INJECTION: m + 1 = member m of cluster 0 never shows up (the give-up / guarded-fallback tests) */
#define FNSSL_TUNE_RESERVED_CUS 36            /* compute units the caller keeps busy with other work (RCCL's all-reduce kernels under an overlapped backward): the cluster-resident kernels size their co-resident grids for device CUs minus this */
#define FNSSL_TUNE_NO_F32_SMALL 37            /* few-sequence fp32 launches (one utterance, a streaming chunk) on the split kernels instead of the slice-resident cluster kernel (lstm_f32s.h) */
#define FNSSL_TUNE_F32C_MIN_GROUPS 38        /* smallest (groups x directions) count the fp32 cluster kernels take for inference (default: see lstm_f32c.hip) */
#define FNSSL_TUNE_F32C_GATE_SPLIT 39        /* fp32 cluster kernels: 1 = every wave owns a group, 4 = the four waves of a slot share one (one gate each); default: by groups per cluster */
#define FNSSL_TUNE_CLUSTER_FULL_TILES 40    /* bf16 cluster kernels, H = 128: full clusters of 24 tiles (rounds 3 - 5) instead of more clusters of 17 - 20 */
#define FNSSL_TUNE_NO_STATIC4 41             /* H = 256 narrow-band layers at full-chip size: two hidden slices per pass (lstm_static3.h) instead of four (lstm_static4.h) */
#define FNSSL_TUNE_COUNT 48               /* room for more without changing the struct */

typedef struct fnssl_tuning {
  unsigned struct_bytes;                 /* sizeof(fnssl_tuning) of the caller's build */
  int knob[FNSSL_TUNE_COUNT];
} fnssl_tuning;

/* The process DEFAULT tuning (copied; NULL = all defaults): used by every call whose descriptor carries none.  It is
 * process-wide so that worker threads (PyTorch's autograd engine runs backward() on its own thread) see what the main
 * thread configured; set it while no call is in flight on another thread — the per-call `tuning` field of the LSTM /
 * network descriptors is the form that is safe to vary between concurrent calls. */
int fnssl_tuning_set(const fnssl_tuning* t);
int fnssl_tuning_get(fnssl_tuning* t);
/* Name of knob `index` ("NO_STATIC3", ...; NULL past the last): lets a host binding map its own configuration
 * source onto the indices (fnssl/_lib.py maps FNSSL_<name> environment variables, on the Python side). */
const char* fnssl_tuning_name(int index);

/* Diagnostic (tests of the cluster kernels' co-residency): `nblocks` workgroups of 64 threads, each claiming `lds_bytes`
 * of LDS (160 KiB = a whole CU), idle on `stream` until stop[0] is non-zero or `max_ms` have passed — what RCCL's
 * persistent all-reduce kernels do to the CUs while the backward pass they overlap with runs.  `stop` (may be NULL):
 * 1 + nblocks DEVICE-visible words (e.g. pinned host memory), zeroed by the caller; workgroup b stores 1 to stop[1 + b]
 * when it has become resident.  Returns at once; never synchronises. */
int fnssl_occupy_cus(int nblocks, int lds_bytes, unsigned* stop, int max_ms, void* stream);

/* ------------------------------------------------------------------------- */
/* Front end                                                                 */
/* ------------------------------------------------------------------------- */

/* Number of STFT frames: floor((ns - 512) / 256 + 1).  Module.py:56. */
int fnssl_num_frames(int ns);

/* Number of mic pairs per utterance for a channel mode.  Module.py:388,396. */
int fnssl_num_pairs(int nch, int ch_mode);

/*
 * Replaces STFT.forward (FN-SSL/Module.py:48-68): per channel Hann-512
 * (periodic) windowed 512-point rFFT, hop 256, center=False, unnormalised.
 *   sig     logical [nb, ns, nch] (reference layout, main.py:185) addressed as
 *           sig[b*sb + n*sn + c*sc] (floats) — so the [nb, nch, ns] batch the
 *           dataloader yields is read in place (sb = nch*ns, sn = 1, sc = ns)
 *   spec    [nb, nch, nt, 257] interleaved (re, im) pairs  (k fastest)
 *   magsum  [nb, nch, nt]  = sum_k |X[k]| over all 257 bins (feeds the
 *           recursive normalisation; may be NULL)
 */
int fnssl_stft(const float* sig, int nb, int ns, int nch,
               long long sb, long long sn, long long sc,
               float* spec, float* magsum, void* stream);

/*
 * The same transform with the hop and the framing of IPDnet2's front end (IPDnet2/Module.py:47-64:
 * torch.stft(n_fft 512, hop_length int(512 * 0.625) = 320, win_length 512, hann, center=True)):
 *   hop     1..512 samples between frames
 *   center  0: frames start at t*hop (as fnssl_stft);  1: frame t is centred on sample t*hop of the signal
 *           extended by REFLECTION of 256 samples at both ends (torch.stft's default pad_mode) — the extension
 *           is an index fold inside the kernel, never materialised;  nt = fnssl_num_frames_ex(ns, hop, center)
 *           = ns / hop + 1 (Module.py:55), needs ns > 256.
 * spec / magsum as fnssl_stft.  fnssl_stft(...) == fnssl_stft_ex(..., 256, 0, ...).
 */
int fnssl_num_frames_ex(int ns, int hop, int center);
int fnssl_stft_ex(const float* sig, int nb, int ns, int nch,
                  long long sb, long long sn, long long sc, int hop, int center,
                  float* spec, float* magsum, void* stream);

/*
 * Host helper: per-frame coefficients (a_t, b_t) of
 *   mu_t = a_t * mu_{t-1} + b_t * mean_t
 * with the reference's float32 rounding (FN-SSL/utils.py:26-44).
 * a, b: HOST arrays of nt floats.
 */
int fnssl_forgetting_coefs(int nt, int sample_length, float* a, float* b);

/*
 * Replaces AddChToBatch + torch.abs + forgetting_norm + real/imag normalise +
 * cat + DC-drop (FN-SSL/Lightning/main.py:207-225, Module.py:383-404,
 * utils.py:9-55).
 *   spec, magsum  outputs of fnssl_stft
 *   coef_a/b      DEVICE arrays [nt] from fnssl_forgetting_coefs
 *   mu            [nb*np, nt] (written; the recursive mean, = forgetting_norm's output)
 *   x             features, channels [Re i, Re j, Im i, Im j], bins 1..256:
 *                 layout 0: [nb*np, nt, 256, 4]  (what the LSTM kernels read)
 *                 layout 1: [nb*np, 4, 256, nt]  (the tensor data_preprocess returns)
 */
int fnssl_pair_features(const float* spec, const float* magsum,
                        const float* coef_a, const float* coef_b,
                        int nb, int nch, int nt, int ch_mode, float eps,
                        float* mu, float* x, int layout, void* stream);

/* x [n, c, nf, nt] -> y [n, nt, nf, c]  (the permute at FN-SSL/Model.py:73). */
int fnssl_nchw_to_seq(const float* x, int n, int c, int nf, int nt, float* y, void* stream);

/* ------------------------------------------------------------------------- */
/* LSTM recurrence (replaces nn.LSTM, FN-SSL/Model.py:25-29,38,46)            */
/* ------------------------------------------------------------------------- */

/* A strided view of sequences: element (q, step, ch) lives at
 *   p[(q / q_inner) * so + (q % q_inner) * si + step * st + ch]   (floats)
 * with q_inner taken from the descriptor.  Channels are contiguous. */
typedef struct {
  const float* p;
  long long so, si, st;
} fnssl_view;

typedef struct {
  /* input = [ (src0 + src1)[0:c0] | src2[0:c2] ] along channels
   * (the residual adds / the concat of FNblock.forward, Model.py:36-37,42-45).
   * src1.p and src2.p may be NULL (c2 = 0).  c0 % 4 == 0, c2 % 4 == 0.       */
  fnssl_view src0, src1, src2;
  int c0, c2;
  /* output h, written at channel offset dir * hidden (bi-dir: [fwd || bwd]) */
  float* out;
  long long out_so, out_si, out_st;
  /* optional fused residual for the NEXT layer: out_sum = h + skip, same strides as
   * out; skip is read at the output's channels (Model.py:36-37,44-45 done one kernel
   * early).  Both NULL to disable; not combinable with src1.                      */
  fnssl_view skip;
  float* out_sum;
  int hidden;          /* H: 16, 32, 64, 128 or 256                           */
  int ndir;            /* 1 = forward only, 2 = bidirectional                 */
  int nseq;            /* number of sequences                                 */
  int q_inner;         /* see fnssl_view                                      */
  int nsteps;          /* sequence length                                     */
  const float* wpack[2];   /* DEVICE packed weights per direction (fnssl_lstm_pack) */
  float* workspace;    /* DEVICE, >= fnssl_lstm_workspace_bytes()             */
  size_t workspace_bytes;
  int variant;         /* 0 = default; see DESIGN.md (kernel variants)        */
  /* training forward (autograd's "save for backward"): when non-NULL the kernel also stores the gate
   * activations and cell state, >= fnssl_lstm_reserve_bytes(); hidden 128/256, no src1 / out_sum.  */
  float* reserve;
  size_t reserve_bytes;
  /* streaming (uni-directional layers only): 1 = continue a sequence.  h_{-1} is read from the row
   * one step BEFORE `out` (out - out_st: the caller's buffer must hold the last h of the previous call
   * there) and c_{-1} from `workspace`, where every call leaves its final cell state (the layout only
   * depends on nseq / hidden, so keep one workspace per layer between calls).                      */
  int carry_state;
  /* arithmetic of the matrix product: FNSSL_PRECISION_FP32 (default: exact fp32 MFMA), or
   * FNSSL_PRECISION_BF16: weights (wpack from fnssl_lstm_pack_bf16) and the [x | h] operands are rounded
   * to bf16, accumulation / gates / cell state / all tensors stay fp32 (BASELINE config 3).  Built for
   * the IPDnet layer shapes (c0, c2 multiples of 16, no src1 / out_sum).                              */
  int precision;
  /* FNSSL_PRECISION_BF16W only: element types of the activation tensors.  Bit 0 / 1 / 2 set = src0 / src2 / out hold
   * fp32, clear = bf16 (then the pointer addresses 2-byte elements; view strides always count ELEMENTS and must be
   * multiples of 8 for bf16 tensors).                                                                  */
  int f32_mask;
  /* optional DEVICE counter (caller-owned, caller-zeroed; NULL to disable): incremented once per layer whose
   * cluster-resident kernel gave up on a hand-off and was re-run by the per-wave / pair-split kernels (see
   * fnssl_lstm_forward).  Read it asynchronously whenever convenient: results are correct either way.      */
  unsigned* fallback_count;
  /* optional per-call tuning (NULL: the calling thread's, fnssl_tuning_set) */
  const fnssl_tuning* tuning;
} fnssl_lstm_desc;

#define FNSSL_PRECISION_FP32 0
#define FNSSL_PRECISION_BF16 1
/* "wide" bf16 kernels (lstm_bf16w.h): same arithmetic contract as FNSSL_PRECISION_BF16 — weights and the [x | h]
 * operands rounded to bf16, fp32 accumulation, gates and cell state — on v_mfma_f32_32x32x16_bf16 with 32 sequences
 * per wave, an LDS-DMA weight ring and bf16 activation tensors between layers (f32_mask).  wpack from
 * fnssl_lstm_pack_bf16w.  Built for the IPDnet hidden-256 layer shapes; no src1 / out_sum / reserve / carry. */
#define FNSSL_PRECISION_BF16W 2

size_t fnssl_lstm_packed_floats_bf16w(int c0, int c2, int hidden);
int fnssl_lstm_pack_bf16w(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                          int c0, int c2, int hidden, float* packed);

/* bf16 weight stream (bias stays fp32): size in floats, and the host-only packer. */
size_t fnssl_lstm_packed_floats_bf16(int c0, int c2, int hidden);
int fnssl_lstm_pack_bf16(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                         int c0, int c2, int hidden, float* packed);

/* Floats in one direction's packed weight stream for (input_size = c0 + c2, hidden). */
size_t fnssl_lstm_packed_floats(int c0, int c2, int hidden);

/*
 * Host-only: pack one direction's PyTorch LSTM parameters
 *   w_ih [4H, c0+c2], w_hh [4H, H], b_ih [4H], b_hh [4H]   (gate order i,f,g,o)
 * into the MFMA-operand stream the kernel consumes (layout: DESIGN.md §3).
 * packed: HOST buffer of fnssl_lstm_packed_floats() floats.
 */
int fnssl_lstm_pack(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                    int c0, int c2, int hidden, float* packed);

/* Scratch a call of fnssl_lstm_forward needs: cell state of a carried (streaming) call, the re-ordered weight stream of the
 * two-slices-per-pass kernel (hidden 256) and — hidden 128 / 256 — the hand-off area of the cluster-resident bf16 kernels
 * (status word, tags, two parities of h_t operand records: 0.5 / 0.4 MiB per 512 / 768 sequences).  One workspace per stream:
 * two calls in flight must not share it. */
size_t fnssl_lstm_workspace_bytes(int nseq, int hidden, int ndir);
/* The same for one `precision` (fnssl_lstm_desc.precision) — what fnssl_lstm_forward actually checks: fp32 calls need only
 * the status word and 8 tag words per 16-sequence group of the cluster-resident fp32 kernel (~2 B per sequence) where the
 * bf16 "wide" calls need ~513 B per sequence; training / streaming / per-stream caches should size with this one. */
size_t fnssl_lstm_workspace_bytes_ex(int nseq, int hidden, int ndir, int precision);

/* Host-only query of the launch planner of fnssl_lstm_forward (full-chip fp32 launches, hidden 128 / 256): the rounds
 * (one launch each, one workgroup of waves_per_wg[i] waves per CU) it runs for nseq sequences per direction on ncu CUs.
 * Returns the number of rounds (waves_per_wg gets min(rounds, cap) entries) or a negative status. */
int fnssl_lstm_plan_rounds(int hidden, int nseq, int ndir, int ncu, int* waves_per_wg, int cap);

/*
 * The layer.  Never traps and never leaves a wrong result behind: the cluster-resident kernels (lstm_f32c.h,
 * lstm_bf16c.h) need every member workgroup of a cluster resident at the same time, which a shared, partitioned or
 * otherwise busy device may not grant; a member that waits ~1.5 s (FNSSL_CLUSTER_SPIN_LIMIT spins) for a hand-off
 * records a code in the call's status word, every workgroup of the launch drains and exits, and the SAME call has
 * already enqueued the per-wave rounds / pair-split kernels behind it, guarded by that word: they return at once when
 * it is 0 and otherwise recompute the whole layer (same bits as the cluster kernel).  Before launching, the grid is
 * checked against the occupancy the device reports; a cluster kernel that cannot be co-resident is not launched.
 * Two cluster launches on different streams of one process take turns on the device by themselves (measured: two
 * concurrent one-utterance forwards = 2 x one, no fallback; tools/two_stream_probe.py); should the dispatcher ever split
 * the CUs between them, the bounded waits and the guarded fallback above are what ends it.
 * Returns FNSSL_OK or a negative status (invalid descriptor, workspace, HIP launch error) with fnssl_last_error().
 */
int fnssl_lstm_forward(const fnssl_lstm_desc* d, void* stream);

/* Which kernel family fnssl_lstm_forward(d) takes on the current device with the current environment (host-only,
 * launches nothing).  Tests assert it, so that "cluster kernel == rounds" comparisons cannot silently compare the
 * rounds with the rounds.  *rounds (optional) = number of launches of the family (planner rounds; 1 otherwise). */
#define FNSSL_LSTM_FAMILY_GENERIC 1         /* lstm_rec_kernel rounds (any shape)                               */
#define FNSSL_LSTM_FAMILY_STATIC 2          /* shape-specialised rounds (lstm_static.h)                         */
#define FNSSL_LSTM_FAMILY_STATIC2 3         /* (retired in round 5: lstm_static2.h removed; never reported)      */
#define FNSSL_LSTM_FAMILY_SPLIT 4           /* several waves per 16-sequence group, generic (small batches)     */
#define FNSSL_LSTM_FAMILY_SPLIT_STATIC 5    /* the same, shape-specialised and ring-free                        */
#define FNSSL_LSTM_FAMILY_F32_CLUSTER 6     /* cluster-resident fp32 kernel (lstm_f32c.h) + guarded fallback    */
#define FNSSL_LSTM_FAMILY_STATIC3 7         /* operand-ring rounds, hidden 256: x_t AND h_{t-1} streamed (lstm_static3.h) */
#define FNSSL_LSTM_FAMILY_BF16 8            /* 16-sequence bf16 kernels (lstm_bf16.h)                           */
#define FNSSL_LSTM_FAMILY_BF16_SOLO 9       /* one-wave-per-group wide bf16 kernels (lstm_bf16w.h)              */
#define FNSSL_LSTM_FAMILY_BF16_PAIR 10      /* pair-split wide bf16 kernels (lstm_bf16p.h)                      */
#define FNSSL_LSTM_FAMILY_BF16_CLUSTER 11   /* cluster-resident bf16 kernels (lstm_bf16c.h) + guarded fallback  */
#define FNSSL_LSTM_FAMILY_TRAIN 12          /* reserve-saving training forward (lstm_train.hip)                 */
#define FNSSL_LSTM_FAMILY_BWD 13            /* fnssl_lstm_backward_plan: per-wave / split BPTT kernels (lstm_train.h) */
#define FNSSL_LSTM_FAMILY_BWD_CLUSTER 14    /* ... cluster-resident BPTT kernel (lstm_bwdc.h) + guarded fallback */
int fnssl_lstm_plan(const fnssl_lstm_desc* d, int* family, int* rounds);

/* Status word the cluster-resident kernels left in a workspace that fnssl_lstm_forward has used with the same
 * (nseq, hidden, ndir): 0 = every hand-off arrived; 0xKnnnn = a wave of cluster nnnn gave up after the spin limit
 * (K = 1 bf16 tag, 3 fp32 tag, 4 fp32 drift bound, 5 aborted because another wave had) — the layer was then re-run by
 * the guarded fallback kernels of the same call, so this is a diagnostic, not an error.  Synchronises with `stream`. */
int fnssl_lstm_cluster_status(const void* workspace, size_t workspace_bytes, int nseq, int hidden, int ndir, void* stream,
                              unsigned* status);

/* ---- training (next row 8f-1): back-propagation through time ---------------------------------- */

size_t fnssl_lstm_reserve_bytes(int nseq, int hidden, int ndir, int nsteps);

/* Floats of one direction's transposed weight stream [W_ih[:, :c0g] | W_hh]^T (c0g = leading input
 * channels whose gradient is wanted, a multiple of 16; 0 for a layer fed by data only). */
size_t fnssl_lstm_bwd_packed_floats(int c0g, int hidden);

/* Host-only: pack w_ih [4H, c_in], w_hh [4H, H] for fnssl_lstm_backward. */
int fnssl_lstm_pack_bwd(const float* w_ih, const float* w_hh, int c_in, int c0g, int hidden, float* packed);

size_t fnssl_lstm_bwd_workspace_bytes(int nseq, int hidden, int ndir);

/*
 * Backward of one (bi)LSTM layer through time (what autograd does for nn.LSTM in
 * training_step, FN-SSL/Lightning/main.py:149-157): from the upstream gradient dh of the layer output
 * and the forward's reserve it writes
 *   da  the pre-activation gate gradients, [.., dir * 4H + (i | f | g | o)]: the weight gradients are
 *       the plain GEMMs  dW_ih = da^T x,  dW_hh = da^T h_prev,  db = sum(da)  (host side, rocBLAS);
 *   dx  the gradient of the first c0g input channels, one slab per direction [.., dir * c0g + ch]
 *       (the caller adds the two directions).
 * Views are addressed like fnssl_lstm_desc's (q / q_inner, q % q_inner, step).
 */
typedef struct {
  const float* reserve;
  fnssl_view dh;
  float* da;
  long long da_so, da_si, da_st;
  float* dx;
  long long dx_so, dx_si, dx_st;
  int c0g;
  int hidden, ndir, nseq, q_inner, nsteps;
  const float* wpack_bwd[2];
  void* workspace;
  size_t workspace_bytes;
  /* optional DEVICE counter (caller-owned, caller-zeroed; NULL to disable): incremented once per call whose
   * cluster-resident BPTT kernel gave up on a hand-off and was re-run by the split kernels of the same call */
  unsigned* fallback_count;
  /* optional per-call tuning (NULL: the calling thread's, fnssl_tuning_set) */
  const fnssl_tuning* tuning;
} fnssl_lstm_bwd_desc;

int fnssl_lstm_backward(const fnssl_lstm_bwd_desc* d, void* stream);

/* Same contract as fnssl_lstm_forward's cluster kernels (above): for the H = 128 full-band layers of a large shard the
 * call launches the cluster-resident BPTT kernel (output slices of [W_ih | W_hh]^T over clusters of co_pad / 64 CUs,
 * hand-offs through memory with bounded, cooperative waits — never a trap) and, behind it in the same call, the per-wave /
 * split kernels as its guarded fallback, which recompute da and dx from the reserve if — and only if — the cluster kernel
 * recorded a hand-off it gave up on.  Results are bit-identical either way.
 *   fnssl_lstm_backward_plan    host only, launches nothing: the FNSSL_LSTM_FAMILY_BWD* code the call would take;
 *   fnssl_lstm_backward_status  the status word the last call left in `workspace` (0: the cluster kernel completed or was
 *                               not used; else (reason << 16 | cluster) of the first wave that gave up, the fallback has
 *                               run); synchronises `stream`.                                                          */
int fnssl_lstm_backward_plan(const fnssl_lstm_bwd_desc* d, int* family);
int fnssl_lstm_backward_status(const void* workspace, size_t workspace_bytes, int nseq, int hidden, int ndir, void* stream,
                               unsigned* status);

/*
 * Weight gradients of one LSTM layer (what autograd accumulates into nn.LSTM's parameters; reference
 * FN-SSL/Lightning/main.py:149-157 loss.backward()):
 *     g_wih[dir] [4H, c0 + c2] += dA_dir^T [x0 | x2]      g_whh[dir] [4H, H] += dA_dir^T h_prev_dir
 *     g_bih[dir], g_bhh[dir] [4H] += sum_r dA_dir[r]
 * Every operand is a row-major [rows = nseq * nsteps, C] matrix (the layer's natural layout: sequence-major, then
 * step), addressed in place:
 *   da   [rows, ndir * 4H] (row stride lda)  gate pre-activation gradients left by fnssl_lstm_backward
 *   x0   [rows, c0] (ldx0), x2 [rows, c2] (ldx2)  the layer's input segments (either may be absent: c = 0)
 *   h    [rows, ndir * H] (ldh)  the layer's OUTPUT; h_prev is taken from it with a one-step shift inside each
 *        sequence (zero at the sequence start / end), never materialised
 * One split-K fp32-MFMA product per layer (csrc/wgrad.hip), partial tiles reduced in a fixed order (deterministic).
 * c0, c2, H multiples of 4; 4H a multiple of 256; strides multiples of 4 floats; pointers 16-byte aligned.
 */
typedef struct {
  const float* da;
  long long lda;
  const float* x0;
  long long ldx0;
  int c0;
  const float* x2;
  long long ldx2;
  int c2;
  const float* h;
  long long ldh;
  long long nseq;
  int nsteps, hidden, ndir;
  float* g_wih[2];
  float* g_whh[2];
  float* g_bih[2];
  float* g_bhh[2];
  void* workspace;
  size_t workspace_bytes;
} fnssl_wgrad_desc;

size_t fnssl_lstm_weight_grads_workspace_bytes(long long rows, int hidden, int ndir, int c0, int c2);
int fnssl_lstm_weight_grads(const fnssl_wgrad_desc* d, void* stream);

/* A logical [nb, nt, nf, C] activation in any memory layout: element (b, t, f, c) at p[b*sb + t*st + f*sf + c]. */
typedef struct {
  const float* p;
  long long sb, st, sf;
} fnssl_btf_view;

/*
 * out = keep_scale(seed32) * (masked[0] + .. ) + (plain[0] + ..): the dropouts and residual adds of
 * FNblock.forward in train mode (FN-SSL/Model.py:36-48) and their backward (gradient accumulation +
 * dropout backward), including the full-band <-> narrow-band layout change (every operand has its own
 * strides).  keep_scale is 0 or 1/(1-p) = 1.25 (p = 0.2), a pure function of seed32 and the logical
 * element index ((b0 + b)*nt + t)*nf + f)*C + c, so forward and backward regenerate the same mask and
 * no mask tensor exists.  use_mask = 0: plain sum.  C % 4 == 0.
 */
int fnssl_train_combine(float* out, long long o_sb, long long o_st, long long o_sf,
                        int nb, int nt, int nf, int c,
                        const fnssl_btf_view* masked, int n_masked,
                        const fnssl_btf_view* plain, int n_plain,
                        int use_mask, unsigned seed32, long long b0, void* stream);

/* out[i] = keep_scale(seed32, offset + i): the mask itself (tests / debugging). */
int fnssl_dropout_scale(float* out, long long n, unsigned seed32, long long offset, void* stream);

size_t fnssl_head_backward_workspace_bytes(void);

/*
 * Backward of fnssl_head (pooling + emb2ipd + tanh, FN-SSL/Model.py:79-87):
 *   x [nb, nf, nt, 256] the head input, pred / dpred [nb, nt//12, 2*nf]
 *   dx [nb, nf, nt, 256] (written);  dw [2, 256], db [2] (written, or += when accumulate != 0)
 */
int fnssl_head_backward(const float* x, const float* w, const float* pred, const float* dpred,
                        int nb, int nf, int nt, float* dx, float* dw, float* db, int accumulate,
                        void* workspace, size_t workspace_bytes, void* stream);

/*
 * cal_loss (FN-SSL/Lightning/main.py:191-198): MSE between the re-batched prediction and the target.
 *   pred [nb*np, nt2, nf2], gt [nb, nt2, nf2, np];  n_total = element count of the WHOLE batch the mean
 *   runs over (>= nb*np*nt2*nf2 when the batch is processed in chunks)
 *   dpred = 2 (pred - gt) / n_total;  *loss (DEVICE) = or += sum((pred - gt)^2) / n_total
 *   workspace >= 1 KiB.
 */
int fnssl_mse_loss(const float* pred, const float* gt, int nb, int np, int nt2, int nf2, long long n_total,
                   float* dpred, float* loss, int accumulate, void* workspace, size_t workspace_bytes,
                   void* stream);

/* torch.optim.Adam (main.py:269-271; no amsgrad, no weight decay) on a flat vector; the gradient is
 * multiplied by grad_scale first (1 / world_size after the sum all-reduce).  step counts from 1. */
int fnssl_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                    float lr, float beta1, float beta2, float eps, int step, float grad_scale, void* stream);

/* ---- whole-network training backward / step (SURVEY.md 8b "later fnssl_backward"; FN-SSL/Lightning/main.py:149-157,
 * 191-198, 269-271) ------------------------------------------------------------------------------------------
 * The parameters, gradient and Adam moments of FN_SSL live in flat DEVICE vectors of fnssl_train_param_floats()
 * floats: element 0 is a constant 0, then the tensors in the reference's named_parameters() order
 * (block_k.fullLstm.{weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0, *_reverse}, block_k.narrLstm..., emb2ipd.weight,
 * emb2ipd.bias; fnssl_train_param_offset gives each offset).  The handle is a HOST object holding the layer table and
 * the gather maps flat vector -> packed weight streams; the maps are uploaded once into a caller-owned device buffer. */
typedef struct fnssl_train fnssl_train;
int fnssl_train_create(int is_online, fnssl_train** out);
void fnssl_train_destroy(fnssl_train* t);
long long fnssl_train_param_floats(const fnssl_train* t);
long long fnssl_train_param_offset(const fnssl_train* t, int layer, int dir, int what);
size_t fnssl_train_map_bytes(const fnssl_train* t);
int fnssl_train_upload_maps(fnssl_train* t, void* dev_buf, void* stream);   /* synchronises `stream` once */
size_t fnssl_train_workspace_bytes(const fnssl_train* t, int nbp, int nf, int nt);

/*
 * One chunk of nb utterances x np pairs: train-mode forward (dropout keyed on seed_base and the GLOBAL pair index
 * pair0 + local pair), MSE against gt, backward; ACCUMULATES into grad and *loss (both DEVICE; the caller zeroes them
 * once per step).
 *   x   [nb*np, 4, nf, nt]  features (layout 1 of fnssl_pair_features)      gt  [nb, nt/12, 2*nf, np]
 *   n_total = element count of the whole batch the loss mean runs over (= nb*np*(nt/12)*2*nf when this is the only chunk)
 * The weight-gradient GEMMs call rocBLAS (resolved with dlopen at first use).  A multi-GPU caller sum-all-reduces grad
 * between this call and fnssl_adam_step (grad_scale = 1 / world).
 */
int fnssl_train_backward(fnssl_train* t, const float* theta, float* grad, const float* x, const float* gt,
                         int nb, int np, int nf, int nt, unsigned seed_base, long long pair0, long long n_total,
                         float* loss, void* workspace, size_t workspace_bytes, void* stream);

/* Single-process optimisation step: zero grad / loss, fnssl_train_backward on the whole batch, fnssl_adam_step. */
int fnssl_train_step(fnssl_train* t, float* theta, float* grad, float* exp_avg, float* exp_avg_sq,
                     const float* x, const float* gt, int nb, int np, int nf, int nt, unsigned seed_base,
                     float lr, float beta1, float beta2, float eps, int step, float* loss,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* Head                                                                      */
/* ------------------------------------------------------------------------- */

/*
 * Replaces pooling + emb2ipd + tanh + re|im packing (FN-SSL/Model.py:79-87).
 *   x    narrow-band output in [nb, nf, nt, 256] layout
 *   w    [2, 256], b [2]      (emb2ipd)
 *   out  [nb, nt/12, 2*nf]
 */
int fnssl_head(const float* x, int nb, int nf, int nt, const float* w, const float* b,
               float* out, void* stream);

/* y[m, n_out] = x[m, k] @ wt[k, n_out] + b   (ipd2doa, Model.py:88-89; wt = weight^T). */
int fnssl_linear(const float* x, int m, int k, const float* wt, const float* b, int n_out,
                 float* y, void* stream);

/* ------------------------------------------------------------------------- */
/* IPD -> DOA back end (next row after the forward path)      */
/* ------------------------------------------------------------------------- */

/*
 * Replaces SourceDetectLocalize.forward, meth_mode 'IDL'
 * (FN-SSL/Lightning/Module.py:525-577): spatial spectrum against a DP-IPD template
 * bank, then per source: first-argmax, projection ratio, subtraction.
 *   pred  element (utterance b, pair p, segment t, k) at pred[b*sb + p*sp + t*st + k*sk]:
 *         the network output [nb*np, nt, nf2] (sb = np*nt*nf2, sp = nt*nf2, st = nf2, sk = 1,
 *         nf2 = 2*nf = [cos | sin]) or the reference's re-batched [nb, nt, nf2, np] view
 *   bank  [ncand, nf2, np]   template bank (host: fnssl.doa.template_bank)
 *   ss    [nb, nt, ncand]    spatial spectrum before any subtraction
 *   idx   [nb, nt, nsrc]     winning candidate per source (int32)
 *   vad   [nb, nt, nsrc]     1 ('kNum') or the projection ratio ('unkNum', unk_num = 1)
 */
/*
 * DP-IPD TARGETS of the training step: replaces DPIPD.forward(source_doa) (FN-SSL/Lightning/Module.py:464-498) and the
 * ground-truth half of MyModel.data_preprocess (FN-SSL/Lightning/main.py:227-262) — the reference builds them on the host
 * in numpy for every batch.
 *   doa      [nb, nseg, 2, ns]   (elevation, azimuth) of every source per segment, radians
 *   vad      [nb, nseg, nvad, ns] voice activity per frame of the segment (may be NULL with nvad = 0: all active)
 *   mic_loc  DEVICE [nmic, 3]    microphone positions in metres
 *   pairs    FNSSL_CH_MODE_M: (0, j); FNSSL_CH_MODE_MM: (i, j), i < j, i-major (Module.py:500-514)
 *   bins     bin0 .. bin0 + nf_used - 1 of np.linspace(0, fre_max, nbins) (main.py:130: bins 1..256 of 257)
 *   ipd      [nb, nseg, 2 * nf_used, np] (written) = sum over sources of gate * [cos | sin](2 pi f tau),
 *            tau = r(doa) . (mic_i - mic_j) / speed, gate = (mean_v vad > 0) when use_vad else 1 (main.py:249-258);
 *            the phase is formed in double precision like numpy's, the result rounded to fp32
 *   vad_mean [nb, nseg, ns] (written; may be NULL) = vad.mean(axis 2) (main.py:243), what gt_batch['vad_sources'] becomes
 */
int fnssl_dpipd_targets(const float* doa, const float* vad, int nb, int nseg, int nvad, int ns, const float* mic_loc,
                        int nmic, int ch_mode, int bin0, int nf_used, int nbins, float fre_max, float speed, int use_vad,
                        float* ipd, float* vad_mean, void* stream);

int fnssl_ipd2doa(const float* pred, long long sb, long long sp, long long st, long long sk,
                  const float* bank, int nb, int np, int nt, int nf2,
                  int ncand, int nsrc, int unk_num, float* ss, int* idx, float* vad, void* stream);

/*
 * Replaces the peak search of SourceDetectLocalize.forward, meth_mode 'PD' (FN-SSL/Lightning/Module.py:580-611; the
 * reference: eight shifted copies of the spectrum, then a Python double loop with a sort per frame).
 *   ss     [nframes, nele, nazi]  spatial spectrum (fnssl_ipd2doa's `ss`)
 *   a cell (e, a), a < nazi - 1 (the last azimuth column is redundant, :581), is a peak when strictly larger than its 8
 *   neighbours: azimuth circular over nazi - 1 columns, elevation clamped (rows 0 and nele - 1 never hold a peak, :583-598)
 *   idx    [nframes, nsrc]  flat cell index e * nazi + a of the nsrc largest peaks, descending by value, equal values in
 *                           ascending index order (:608-609); -1 where a frame has fewer peaks
 *   val    [nframes, nsrc]  their spectrum values (0 where idx = -1)
 *   count  [nframes]        min(number of peaks, nsrc)
 */
int fnssl_doa_peaks(const float* ss, int nframes, int nele, int nazi, int nsrc, int* idx, float* val, int* count, void* stream);

/*
 * IPDnet's all-channel features (replaces IPDnet/runIPDnetOn.py:240-254: abs, forgetting_norm over
 * all channels with sample_length 280, real/imag normalise, cat, DC-drop).
 *   spec, magsum  outputs of fnssl_stft;  coef_a/b  DEVICE [nt] from fnssl_forgetting_coefs
 *   mu            [nb, nt] (written)
 *   x             channels [Re ch 0..nch-1, Im ch 0..nch-1], bins 1..256:
 *                 layout 0: [nb, nt, 256, 2*nch];  layout 1: [nb, 2*nch, 256, nt] (what the reference builds)
 */
int fnssl_array_features(const float* spec, const float* magsum,
                         const float* coef_a, const float* coef_b,
                         int nb, int nch, int nt, float eps,
                         float* mu, float* x, int layout, void* stream);

/*
 * The whole array front end in one call, waveforms -> features, WITHOUT materialising the spectrum (replaces the
 * same reference lines as fnssl_stft_ex + fnssl_array_features: IPDnet/runIPDnetOn.py:240-254 with hop 256 /
 * center 0 / sample_length 280, IPDnet2/Module.py:47-64 + run_IPDnet2.py:277-288 with hop 320 / center 1 / 249):
 * a magnitude pass (sum_k |X| per utterance, channel, frame), the recursive mean, and a second transform pass that
 * divides by (mu + eps) and writes each frame's feature row as one contiguous piece.
 *   sig      as fnssl_stft_ex           coef_a/b  DEVICE [nt] from fnssl_forgetting_coefs
 *   magsum   [nb, nch, nt] (written)    mu        [nb, nt] (written)
 *   x        [nb, nt, 256, 2*nch]  channels [Re ch 0..nch-1, Im ch 0..nch-1], bins 1..256 (layout 0 of
 *            fnssl_array_features; the reference's [nb, 2*nch, 256, nt] tensor is its permute(0, 3, 2, 1) view)
 */
int fnssl_array_frontend(const float* sig, int nb, int ns, int nch, long long sb, long long sn, long long sc,
                         int hop, int center, const float* coef_a, const float* coef_b, float eps,
                         float* magsum, float* mu, float* x, void* stream);

/* ------------------------------------------------------------------------- */
/* IPDnet head (next row 8f-3): causal 3x3 Conv2d + time pooling              */
/* ------------------------------------------------------------------------- */

/* Floats of the packed weight stream for Conv2d(ca + cb -> cout, 3x3); 0 if unsupported
 * (cout <= 128, ca % 16 == 0, cb % 4 == 0). */
size_t fnssl_conv3x3_packed_floats(int cout, int ca, int cb);

/* Host-only: pack a PyTorch Conv2d weight [cout, ca + cb, 3, 3] for fnssl_conv3x3_causal. */
int fnssl_conv3x3_pack(const float* w, int cout, int ca, int cb, float* packed);

/*
 * Replaces one conv stage of CausCnnBlock.forward (IPDnet/FixedAarryIPDnet.py:61-73):
 * Conv2d(k = 3x3, padding (1, 2), bias = False) -> activation -> crop the last 2 time steps,
 * i.e. out[b, f, t, :] = act( sum_{df, dt, c} W[:, c, df, dt] * x[b, f+df-1, t+dt-2, c] ).
 * The input is the channel concatenation [xa (ca channels) | xb (cb channels)] of two
 * channels-last tensors; element (b, f, t, c) of xa lives at xa[b*a_sb + f*a_sf + t*a_st + c].
 *   act: 0 none, 1 ReLU, 2 tanh;  out [nb, nf, nt, cout_stride] channels-last (cout valid).
 */
int fnssl_conv3x3_causal(const float* xa, long long a_sb, long long a_sf, long long a_st, int ca,
                         const float* xb, long long b_sb, long long b_sf, long long b_st, int cb,
                         const float* wpack, int cout, int nb, int nf, int nt, int act,
                         float* out, int cout_stride, void* stream);

/* bf16-MFMA variant (BASELINE config 3): weights rounded to bf16 in the stream, activation operands rounded
 * to bf16 as they enter the MFMA, fp32 accumulation and tensors; ca and cb multiples of 16. */
size_t fnssl_conv3x3_packed_floats_bf16(int cout, int ca, int cb);
int fnssl_conv3x3_pack_bf16(const float* w, int cout, int ca, int cb, float* packed);
int fnssl_conv3x3_causal_bf16(const float* xa, long long a_sb, long long a_sf, long long a_st, int ca,
                              const float* xb, long long b_sb, long long b_sf, long long b_st, int cb,
                              const float* wpack, int cout, int nb, int nf, int nt, int act,
                              float* out, int cout_stride, void* stream);

/* Same, with segment A held as bf16 (what the wide bf16 LSTM kernels write): xa_bf16 addresses 2-byte elements,
 * a_sb / a_sf / a_st count elements (multiples of 4), base 8-byte aligned; xb stays fp32. */
int fnssl_conv3x3_causal_bf16a(const void* xa_bf16, long long a_sb, long long a_sf, long long a_st, int ca,
                               const float* xb, long long b_sb, long long b_sf, long long b_st, int cb,
                               const float* wpack, int cout, int nb, int nf, int nt, int act,
                               float* out, int cout_stride, void* stream);

/* y[row, t2, c] = mean_{k < K} x[row, K*t2 + k, c]   (AvgPool2d((1, K)); c % 4 == 0). */
int fnssl_avgpool_time(const float* x, int rows, int nt, int c, int k, float* y, void* stream);

/* Second bf16 formulation (conv_bf16x.hip): activations staged through LDS in coalesced half-lines, 32x32x16 bf16
 * MFMA tiles.  Same contract as fnssl_conv3x3_causal_bf16a (segment A bf16, the skip segment B fp32, fp32 output),
 * its own weight stream; supported: 64 < cout <= 128 with cout % 4 == 0, ca % 32 == 0, cb % 16 == 0, bases and
 * strides 16-byte aligned (packed_bytes returns 0 otherwise and the caller keeps the _bf16a entry).
 * pool = 3 or 4 applies the AvgPool2d((1, pool)) that follows the activation in CausCnnBlock.forward
 * (IPDnet/FixedAarryIPDnet.py:63-70) in the epilogue: out is [nb, nf, nt / pool, cout_stride]; out_bf16 = 1 writes
 * it as bf16 (the next conv's segment A).  pool = 1, out_bf16 = 0: the plain fp32 [nb, nf, nt, cout_stride]. */
size_t fnssl_conv3x3_packed_bytes_bf16x(int cout, int ca, int cb);
int fnssl_conv3x3_pack_bf16x(const float* w, int cout, int ca, int cb, void* packed);
int fnssl_conv3x3_causal_bf16x(const void* xa_bf16, long long a_sb, long long a_sf, long long a_st, int ca,
                               const float* xb, long long b_sb, long long b_sf, long long b_st, int cb,
                               const void* wpack, int cout, int nb, int nf, int nt, int act,
                               int pool, int out_bf16, void* out, int cout_stride, void* stream);

/* fnssl_avgpool_time with the result rounded to bf16 (the next conv's segment A; the rounding is the one the
 * bf16 conv kernels apply to fp32 operands, so the chain's results do not change). */
int fnssl_avgpool_time_bf16(const float* x, int rows, int nt, int c, int k, void* y_bf16, void* stream);

/* ------------------------------------------------------------------------- */
/* Whole network (replaces FN_SSL.forward, FN-SSL/Model.py:72-90)             */
/* ------------------------------------------------------------------------- */

typedef struct {
  /* packed LSTM streams, DEVICE: [block 0..2][0 = full-band, 1 = narrow-band][direction] */
  const float* wpack[3][2][2];
  const float* emb_w;     /* [2, 256] */
  const float* emb_b;     /* [2]      */
  const float* doa_wt;    /* [512, 180] = ipd2doa.weight^T, or NULL */
  const float* doa_b;     /* [180] or NULL                          */
  int input_size;         /* 4                                      */
  int is_online;          /* narrow-band: 1 = uni-dir H=256, 0 = bi-dir H=128 */
  unsigned* fallback_count;   /* optional DEVICE counter handed to every layer (fnssl_lstm_desc.fallback_count); NULL: none */
  const fnssl_tuning* tuning; /* optional per-call tuning handed to every layer (NULL: the calling thread's) */
} fnssl_net;

size_t fnssl_forward_workspace_bytes(int nb, int nf, int nt, int is_online, int chunk_pairs);

/*
 *   x0   [nb, nt, nf, input_size]   (layout 0 of fnssl_pair_features)
 *   out  [nb, nt/12, 2*nf]  (or [nb, nt/12, 180] with the DOA layer)
 *   chunk_pairs: pairs processed per pass (0 = all at once)
 */
int fnssl_forward(const fnssl_net* net, const float* x0, int nb, int nf, int nt,
                  float* out, void* workspace, size_t workspace_bytes, int chunk_pairs,
                  void* stream);

/* ------------------------------------------------------------------------- */
/* IPDnet2 / OnlineSpatialNet (SURVEY.md 8 rows a13 + f4; reference IPDnet2/IPDnet2.py) */
/* ------------------------------------------------------------------------- */
/*
 * Built for the network the reference ships (run_IPDnet2.py:103-119): dim_hidden H = 96, dim_squeeze 8,
 * f-conv kernel 5 / 8 groups, encoder kernel 5, attention 'mamba(16,4)' (d_inner E = 192, d_state 16,
 * d_conv 4, dt_rank 6), dim_output 16, frequency compression 2 then 8 (= 16), time compression in layer 0.
 * dim_input (2 x mics) and the numbers of bins / frames / layers are free.
 *
 * Activations are logical [B, T, F, H] tensors given as fnssl_btf_view structs: element (b, t, f, h) at
 * p[b*sb + t*st + f*sf + h], H contiguous, strides in floats, multiples of 4, 16-byte aligned base; the
 * reference's [B, F, T, H] tensors are the same thing with other strides.  Every kernel maps one THREAD to one
 * time-frequency point; the weights of the small dense layers are read as wave-uniform scalars, so they are
 * passed TRANSPOSED ([in][out], noted per field) — fnssl.spatialnet does the transposes once per model.
 */

/* y[row, :] = LayerNorm(x[row, :]) * w + b over h channels (arch/base/norm.py:11-27); one wave per row,
 * mean / variance by wavefront reduction.  h <= 1024. */
int fnssl_sn_layernorm(const float* x, long long rows, int h, const float* w, const float* b, float eps,
                       float* y, void* stream);

/*
 * Replaces OnlineSpatialNet.encoder = CausalConv1d(cin -> 96, k = 5, look_ahead 0) along time
 * (IPDnet2.py:45-82, call :335).
 *   x          network input, element (b, c, f, t) at x[b*x_sb + c*x_sc + f*x_sf + t*x_st]
 *              (the reference's [B, C, F, T] tensor: x_st = 1)
 *   wT         [cin][5][96] = weight[o, c, k] transposed;  bias [96]
 *   state_in   [nb, cin, nf, 4]: the 4 input frames before this chunk (NULL: zero left padding, :69)
 *   state_out  [nb, cin, nf, 4]: written with the last 4 input frames (NULL: not wanted)
 *   out        view [nb, nt, nf, 96]
 *   precision  FNSSL_PRECISION_FP32 (exact fp32 MFMA), or FNSSL_PRECISION_BF16: both operands of the product
 *              are rounded to bf16 (nearest even) as they enter the matrix pipe, fp32 accumulation, fp32
 *              tensors — BASELINE config 5 as written; needs cin * 5 <= 160.  The same argument of
 *              fnssl_sn_fconv (the grouped conv; nf >= 16), fnssl_sn_full (squeeze, Linear over F, unsqueeze;
 *              nf 16, 64 or 128) and fnssl_sn_mamba (in_proj, x_proj, out_proj) means the same — at other
 *              nf these two have no matrix-pipe kernel and compute in exact fp32 whatever the argument;
 *              LayerNorm, biases and activations, the depthwise conv, dt_proj, the scan and the head are fp32
 *              in both modes.
 */
int fnssl_sn_encoder(const float* x, long long x_sb, long long x_sc, long long x_sf, long long x_st,
                     int nb, int cin, int nf, int nt, const float* wT, const float* bias,
                     const float* state_in, float* state_out,
                     float* out, long long o_sb, long long o_st, long long o_sf, int precision, void* stream);

/* One f-conv branch (IPDnet2.py:105-109): LayerNorm(96) -> Conv1d(96, 96, k 5, groups 8, 'same' zero padding)
 * along F -> PReLU(96). */
typedef struct {
  const float *ln_w, *ln_b;   /* [96]                                                              */
  const float* wT;            /* [8 groups][5 taps][12 in][12 out] = weight[g*12+o, ci, tap]         */
  const float* bias;          /* [96]                                                              */
  const float* prelu;         /* [96]                                                              */
} fnssl_sn_fconv_w;

/*
 * Replaces `x + self._fconv(ml, x)` (IPDnet2.py:146,151,222-233) and, with pool > 1, the AvgPool over
 * frequency that follows in the first layer (:147-148, :152-153).
 *   x [nb, nt, nf, 96] (nf a power of two, 8 <= nf <= 256);  residual 0: only the branch
 *   pool 1, 2 or 8;  out view [nb, nt, nf / pool, 96] (may alias x when pool == 1)
 */
int fnssl_sn_fconv(const fnssl_btf_view* x, int nb, int nt, int nf, const fnssl_sn_fconv_w* w,
                   int residual, int pool, float* out, long long o_sb, long long o_st, long long o_sf,
                   int precision, void* stream);

/* The full-band branch (IPDnet2.py:111-118): LayerNorm -> Conv1d(96 -> 8, 1) + SiLU -> Linear(nf, nf) over F ->
 * Conv1d(8 -> 96, 1) + SiLU. */
typedef struct {
  const float *ln_w, *ln_b;   /* [96]                          */
  const float *wsT, *bs;      /* [96][8] = squeeze weight^T, [8] */
  const float *wfT, *bf;      /* [nf][nf] = full.weight^T, [nf]  */
  const float *wuT, *bu;      /* [8][96] = unsqueeze weight^T, [96] */
} fnssl_sn_full_w;

/* Replaces `x + self._full(x)` (IPDnet2.py:150,235-253; dropout_full off).  nf a power of two, 8 <= nf <= 256.
 * out may alias x. */
int fnssl_sn_full(const fnssl_btf_view* x, int nb, int nt, int nf, const fnssl_sn_full_w* w, int residual,
                  float* out, long long o_sb, long long o_st, long long o_sf, int precision, void* stream);

/* One Mamba block with its LayerNorm (IPDnet2.py:126-132; mamba_ssm.Mamba(d_model 96, d_state 16, d_conv 4)). */
typedef struct {
  const float *ln_w, *ln_b;   /* [96]                                                   */
  const float* winT;          /* [96][384]  = in_proj.weight^T  (x half first, then z)    */
  const float *conv_w, *conv_b; /* [192][4], [192]   depthwise causal conv                */
  const float* wxT;           /* [192][40]  = x_proj.weight^T, columns (dt 6 | B 16 | C 16 | 2 zero) */
  const float *wdt, *bdt;     /* [192][6], [192]   dt_proj                              */
  const float* a;             /* [192][16]  = -exp(A_log)                               */
  const float* d;             /* [192]                                                  */
  const float* woT;           /* [192][96]  = out_proj.weight^T                         */
} fnssl_sn_mamba_w;

size_t fnssl_sn_mamba_workspace_bytes(int nb, int nt, int nf);

/*
 * Replaces `x + self._mamba(x, mamba, norm, dropout)` (IPDnet2.py:155-162, 166-181) and, with time_pool > 1, the
 * time pooling after the layer (:345-349; applied to x + branch, using that out_proj is linear):
 * LayerNorm, in_proj, causal depthwise conv + SiLU, x_proj, dt_proj + softplus, selective scan
 *   h_t = exp(dt_t A) h_{t-1} + dt_t B_t u_t,  y_t = C_t . h_t + D u_t,  gate by SiLU(z), out_proj
 * along T for each of the nb*nf sequences.  (PARITY UNPINNED: the published algorithm, see oracle/ipdnet2_oracle.py.)
 *   conv_state [nb*nf, 3, 192], ssm_state [nb*nf, 192, 16]: NULL, or carried state — read when carry != 0,
 *   always written (sequence index = b*nf + f)
 *   out view [nb, nt / time_pool, nf, 96] (may alias x when time_pool == 1)
 */
int fnssl_sn_mamba(const fnssl_btf_view* x, int nb, int nt, int nf, const fnssl_sn_mamba_w* w, int residual,
                   int time_pool, float* conv_state, float* ssm_state, int carry,
                   float* out, long long o_sb, long long o_st, long long o_sf,
                   void* workspace, size_t workspace_bytes, int precision, void* stream);

/*
 * Replaces FreqInverse + tanh + decoder + the output re-ordering (IPDnet2.py:23-43, 355-364).
 *   x     [nb, nt2, nfc, 96] (nfc compressed bins; nf = 16 * nfc)
 *   wfiP  [16 r][16 o][96] = trans2.weight[o*16 + r, h];  bfiP [16 r][16 o] = trans2.bias[o*16 + r]
 *   wdT   [16][16] = decoder.weight^T ([in][out]), bd [16]
 *   out   [nb, nt2, 2*nf, 4, 2] contiguous
 */
int fnssl_sn_head(const fnssl_btf_view* x, int nb, int nt2, int nfc, const float* wfiP, const float* bfiP,
                  const float* wdT, const float* bd, float* out, void* stream);

#define FNSSL_SN_MAX_LAYERS 16
typedef struct {
  fnssl_sn_fconv_w fconv1, fconv2;
  fnssl_sn_full_w full;
  fnssl_sn_mamba_w mamba[2];       /* mhsa, tconvffn */
} fnssl_sn_layer;

typedef struct {
  int dim_input, num_layers, time_ratio;   /* time_ratio: 5 (time_compression_layer = 0)  */
  const float *enc_wT, *enc_b;
  fnssl_sn_layer layers[FNSSL_SN_MAX_LAYERS];
  const float *wfiP, *bfiP, *wdT, *bd;
  int precision;                           /* FNSSL_PRECISION_FP32 / _BF16, see fnssl_sn_encoder */
} fnssl_sn_net;

/* bytes of workspace fnssl_sn_forward needs, sufficient for every time_ratio it accepts (1..16) */
size_t fnssl_sn_forward_workspace_bytes(int nb, int nf, int nt);
/* floats of the carried state of a whole network (encoder frames + per Mamba block conv / ssm state) */
size_t fnssl_sn_state_floats(const fnssl_sn_net* net, int nb, int nf);

/*
 * Replaces OnlineSpatialNet.forward (IPDnet2.py:331-368).
 *   x [nb, dim_input, nf, nt] with the given strides -> out [nb, nt / 5, 2*nf, 4, 2]
 *   state: NULL (whole utterance), or fnssl_sn_state_floats() floats carried between chunks (read when
 *   carry != 0, always written); streaming chunks must be multiples of time_ratio frames.
 */
int fnssl_sn_forward(const fnssl_sn_net* net, const float* x, long long x_sb, long long x_sc, long long x_sf,
                     long long x_st, int nb, int nf, int nt, float* state, int carry, float* out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* Measurement hooks (bench.py: per-kernel HIP-event timing on the launch stream) */
/* ------------------------------------------------------------------------- */

/* The device's own fp32-MFMA ceiling (calibration, not product): launches CUs x waves_per_simd workgroups of four
 * waves that issue nothing but v_mfma_f32_16x16x4_f32 (iters x 64 per wave); *flop = floating-point operations of the
 * launch.  The caller times it with events on `stream`; bench.py reports flop / time as roofline.peak_measured next to
 * the datasheet peak.  out: >= CUs * waves_per_simd * 256 floats of scratch. */
int fnssl_mfma_f32_peak(float* out, size_t out_floats, int iters, int waves_per_simd, double* flop, void* stream);
/* The same launch, additionally recording per workgroup b: clocks[3b] = shader cycles (s_memtime), clocks[3b + 1] = ticks of
 * the constant 100 MHz counter (s_memrealtime), clocks[3b + 2] = the XCD it ran on (HW_REG_XCC_ID) — cycles / ticks x 100 MHz
 * is the clock that XCD held under a saturated matrix pipe.  bench.py launches it back to back for >= 2 s and reports the
 * rate of the whole window (roofline.peak_measured_sustained) with the slowest / fastest XCD clock of the last launch, so
 * that a box timing 576 instead of 540 ms per step is attributable from the bench line alone.  clocks: device memory,
 * >= CUs * waves_per_simd * 3 entries, or NULL (then exactly fnssl_mfma_f32_peak). */
int fnssl_mfma_f32_peak_clocks(float* out, size_t out_floats, int iters, int waves_per_simd, double* flop,
                               unsigned long long* clocks, size_t clocks_len, void* stream);

/* enable = 1: every kernel launch is bracketed by hipEvents on its stream. */
int fnssl_timing_enable(int enable);
/* Restrict the bracketing to launches of one kernel name (as reported by fnssl_timing_collect); NULL or "": all.
 * bench.py brackets only the roofline kernel inside the timed region: two event records per launch on 106 launches
 * cost IPDnet2 10 % of its step. */
int fnssl_timing_select(const char* name);
/* Drain recorded events (synchronises them) and reset.  Returns the number of
 * distinct kernel names; names/ms/count/flops are HOST arrays of capacity cap. */
int fnssl_timing_collect(int cap, char (*names)[64], double* total_ms, long long* count,
                         double* flops);

#ifdef __cplusplus
}
#endif
#endif /* FNSSL_H_ */
