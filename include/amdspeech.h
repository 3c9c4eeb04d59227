/*
 * amdspeech.h -- C ABI of libamdspeech.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the acoustic-model hot path of domerin0/rnn-speech
 * (audio -> MFCC/fbank -> Linear -> stacked LSTM -> Linear -> CTC loss+grad ->
 * clip + Adam).  The reference has no FFI seam of its own: every one of these
 * ops is a TensorFlow-1.x / librosa call issued from
 * /root/reference/models/AcousticModel.py and /root/reference/util/audioprocessor.py.
 * Each entry point below names the reference call site it replaces; the Python
 * class surface above it (models.AcousticModel, util.audioprocessor.AudioProcessor)
 * is kept verbatim by rnn-speech_amd/ and binds these symbols through ctypes
 * (see INTEGRATION.md for the stub).
 *
 * Conventions
 *   - plain C, no torch types; every tensor is a caller-owned DEVICE pointer,
 *     contiguous, float32 unless stated (int32 for lengths / labels);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all
 *     work is enqueued on it, nothing synchronises the device;
 *   - workspaces are caller-allocated, sized by the *_workspace_bytes queries;
 *   - return value: 0 = ok, negative = AMDSPEECH_E*; amdspeech_last_error()
 *     returns the message of the calling thread's last failure;
 *   - time-major activations [T, B, *]; the LSTM kernel of layer l is the TF
 *     BasicLSTMCell matrix K_l [2H, 4H] (rows: x then h; column blocks i|j|f|o),
 *     bias_l [4H]; forget_bias 1.0 is added at run time, never stored.
 */
#ifndef AMDSPEECH_H
#define AMDSPEECH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMDSPEECH_OK 0
#define AMDSPEECH_EINVAL (-1)   /* bad argument (shape, alignment, null pointer) */
#define AMDSPEECH_EHIP (-2)     /* a HIP runtime call failed */
#define AMDSPEECH_EUNSUPPORTED (-3)
#define AMDSPEECH_ETIMEOUT (-4) /* amdspeech_lstm_status: a bounded wait of a whole-sequence kernel gave up -- that mini-batch's results are
                                   invalid and it can be repeated (AMDSPEECH_LSTM_PER_DIAGONAL); every other error there is a device fault */

int amdspeech_version(void);
const char* amdspeech_last_error(void);
/* Number of compute units of the current device (sanity / grid sizing). */
int amdspeech_device_cu_count(void);

/* ---------------------------------------------------------------- Linear ----
 * y[M,N] = x[M,K] . w[K,N] + b[N]           (b may be NULL)
 * Replaces the per-frame tf.matmul(...) + bias lists of the input and output
 * layers, models/AcousticModel.py:247-250 and :308-309.                      */
int amdspeech_linear_fwd(void* stream, const float* x, const float* w, const float* b,
                         float* y, int M, int K, int N);
/* Backward of the above (what tf.gradients emits for :247-250/:308-309):
 *   dx[M,K]  = dy . w^T              (dx may be NULL: skipped)
 *   dw[K,N] += x^T . dy              (ACCUMULATES -- the reference's gradient
 *   db[N]   += column sums of dy      accumulators, :391-401)                */
int amdspeech_linear_bwd(void* stream, const float* x, const float* w, const float* dy,
                         float* dx, float* dw, float* db, int M, int K, int N);

/* General f32 MFMA GEMM used by the two calls above (exposed for tests/bench):
 * C[M,N] (+)= op(A)[M,K] . op(B)[K,N] (+ bias[N]).  transX != 0 means the
 * operand is stored transposed (A as [K,M], B as [N,K]); ld* are row strides
 * in elements.  accumulate != 0 adds into C.                                 */
int amdspeech_gemm_f32(void* stream, int transA, int transB, int M, int N, int K,
                       const float* A, int lda, const float* B, int ldb,
                       float* C, int ldc, const float* bias, int accumulate);

/* The same product in split precision ("bf16x3", the arithmetic of amdspeech_lstm_desc.precision = 1): every f32 operand value
 * is used as a bf16 pair hi = rne(x), lo = rne(x - hi) and every product as hi.hi + hi.lo + lo.hi on the bf16 MFMA with f32
 * accumulation (~16 significant bits per operand); operands and result stay float32 in memory.  Used by lstm_fwd / lstm_bwd
 * for their batched products at H = 1024 in that mode; exposed for tests and benchmarks.  A and B must be 16-byte aligned. */
int amdspeech_gemm_bf16x3(void* stream, int transA, int transB, int M, int N, int K,
                          const float* A, int lda, const float* B, int ldb,
                          float* C, int ldc, const float* bias, int accumulate);

/* ... and with PLAIN bf16 operands (round 4; the arithmetic of amdspeech_lstm_desc.precision = 2, BASELINE configs[4]'s "bf16
 * MFMA"): every f32 operand value is rounded to ONE bf16 (nearest even, 8 significant bits), every product is one bf16 MFMA,
 * accumulation and the result are float32; operands stay float32 in memory (the master copies).  Same contract otherwise.   */
int amdspeech_gemm_bf16(void* stream, int transA, int transB, int M, int N, int K,
                        const float* A, int lda, const float* B, int ldb,
                        float* C, int ldc, const float* bias, int accumulate);
/* ... the same product through bf16 COPIES of the operands (round 5; what lstm_fwd / lstm_bwd run at H = 1024 with precision = 2):
 * each operand is copied once as bf16 with the contraction index contiguous (a 64 x 64 transpose where it is not), a 256 x 256 x 64
 * kernel streams the copies into LDS with global_load_lds, split K goes through f32 partial tiles (no atomics).  Same values
 * per operand as amdspeech_gemm_bf16 (round to nearest even), another summation order.  `scratch`: 256-byte aligned,
 * amdspeech_gemm_bf16_packed_scratch_bytes(...) bytes -- 0 from that query means the shape is not taken (K % 64, transposed
 * operands in multiples of 64, M, N >= 256): call amdspeech_gemm_bf16 instead.                                              */
size_t amdspeech_gemm_bf16_packed_scratch_bytes(int transA, int transB, int M, int N, int K, int lda, int ldb);
int amdspeech_gemm_bf16_packed(void* stream, int transA, int transB, int M, int N, int K,
                               const float* A, int lda, const float* B, int ldb,
                               float* C, int ldc, const float* bias, int accumulate, void* scratch, size_t scratch_bytes);

/* ----------------------------------------------------------- batch norm ----
 * Optional normalisation of the input-layer output, models/AcousticModel.py:253-259
 * (`batch_normalization : True` in config.ini; off by default): moments over the
 * BATCH axis only, per (t, feature); y = (x - mean) / sqrt(var + eps), biased variance,
 * eps = 1e-3 in the reference, no scale/offset, no running statistics.
 *   x, y, xhat: [T,B,H] (y may alias x; xhat, the saved normalised value, may be NULL
 *   when no backward follows); inv_std: [T,H].  Backward: dx from dy, xhat, inv_std. */
int amdspeech_batchnorm_fwd(void* stream, const float* x, float* y, float* xhat,
                            float* inv_std, int T, int B, int H, float eps);
int amdspeech_batchnorm_bwd(void* stream, const float* dy, const float* xhat,
                            const float* inv_std, float* dx, int T, int B, int H);

/* The same normalisation under data parallelism: tf.nn.moments then spans the GLOBAL batch (n_total = B * world), so every
 * sum over the batch axis is "local sum -> amdspeech_allreduce_sum_f32 -> finish".  Forward: batchnorm_sum(x, NULL) ->
 * all-reduce [T,H] -> batchnorm_sum(x, global_sum) (sum of squared deviations from the global mean) -> all-reduce ->
 * batchnorm_apply.  Backward: batchnorm_bwd_sums ([2][T,H]: sum dy, sum dy*xhat) -> all-reduce -> batchnorm_bwd_apply.  */
int amdspeech_batchnorm_sum(void* stream, const float* x, const float* global_sum, int n_total, float* out,
                            int T, int B, int H);
int amdspeech_batchnorm_apply(void* stream, const float* x, const float* global_sum, const float* global_sq,
                              int n_total, float eps, float* y, float* xhat, float* inv_std, int T, int B, int H);
int amdspeech_batchnorm_bwd_sums(void* stream, const float* dy, const float* xhat, float* sums, int T, int B, int H);
int amdspeech_batchnorm_bwd_apply(void* stream, const float* dy, const float* xhat, const float* inv_std,
                                  const float* global_sums, int n_total, float* dx, int T, int B, int H);

/* ------------------------------------------------------------ LSTM stack ----
 * Replaces tf.contrib.rnn.BasicLSTMCell + DropoutWrapper + MultiRNNCell +
 * tf.nn.dynamic_rnn(sequence_length, initial_state, time_major=True),
 * models/AcousticModel.py:223-237 and :266-298, and its BPTT gradient.
 *
 * Parameters live wherever the caller keeps them: layer l's kernel is at
 * kernels + l*kernel_stride, its bias at biases + l*bias_stride (elements).
 *
 * Workspace (one allocation, reused by fwd and bwd of the same step) holds the
 * repacked weights, the inter-layer activations Z_l [L+1][T][B][H], the state
 * history h/c [L][T+1][B][H], the activated gates [L][T][B][4H] and the gate
 * gradients.  Z_0 (the input of layer 0 = output of the input Linear) and
 * dZ_L (gradient arriving at the top) are views INTO that workspace obtained
 * with amdspeech_lstm_ws_ptr so the neighbouring GEMMs write them in place.  */
typedef struct amdspeech_lstm_desc {
    int T, B, H, L;
    float keep_in, keep_out;   /* DropoutWrapper keep probabilities (1 = off) */
    uint64_t seed;             /* dropout stream; the same seed in fwd and bwd */
    int precision;             /* 0: exact f32 MFMA (default, what the reference computes);
                                  1: "bf16x3" split products hi.hi + hi.lo + lo.hi with f32
                                     accumulation (~16 significant bits per operand), needs H % 32 == 0;
                                  2: "bf16" (round 4) every operand of the recurrent AND the batched products rounded to
                                     ONE bf16 (8 significant bits), one MFMA per product, f32 accumulation; gates, cell
                                     state, gradients, master weights stay f32.  Measured against float64 over ~1000 frames
                                     (DESIGN.md 4.2d): logits 1.3-2.3e-3 of max (outside north_star's 1e-3), CTC loss
                                     7e-5, gradients 3-5e-3 -- an opt-in throughput mode, never the default.  Shapes outside
                                     the dataflow / per-layer kernels run their RECURRENT products in bf16x3 (a superset
                                     in accuracy); the batched products (weight gradients, dZ_0) are single bf16 there too */
    int flags;                 /* 0, or AMDSPEECH_LSTM_* bits below (training cycles on ONE workspace and shape) */
} amdspeech_lstm_desc;

/* The whole-sequence kernels poll hand-off panels inside the workspace that have to hold a sentinel when the launch starts
 * (forward: 330 MB at the benchmark shape; backward: the dX panels and two rings).  In a training cycle
 * lstm_fwd -> lstm_bwd -> lstm_fwd ... on ONE workspace and shape the fills come off the critical path:
 *   ARM_NEXT  (lstm_fwd) prepare, beside the forward kernel (it leaves two XCDs idle) and on a side stream of the library: the
 *             backward call's panels, its transposed weight pack (from `kernels`, which must not change before that call),
 *             and the forward panels of the NEXT forward call -- the workspace holds two sets of them and the calls of a
 *             cycle alternate (until round 3 this call's own set was re-filled behind its kernel, i.e. beside the caller's
 *             output layer).  The next lstm_fwd / lstm_bwd call makes its stream wait for that side stream first.
 *   ARMED     (lstm_bwd) the lstm_fwd before it, on this workspace and with the same T/B/H/L/precision, had ARM_NEXT;
 *             (lstm_fwd) the previous lstm_fwd on this workspace had ARM_NEXT and the same T/B/H/L/precision.
 *             The call then skips its fill.  Passing ARMED when that is not true makes the kernels read stale panels (their
 *             bounded waits then end in AMDSPEECH_ETIMEOUT at the next lstm_status).
 *   SAME_WS   (lstm_fwd, round 5) the previous lstm_fwd on this workspace ran with the same B / H / L / precision -- T may differ --
 *             and nothing but lstm calls has written to the workspace since; ARMED implies it.  What it protects (H = 512, exact
 *             f32, at least one XCD without a recurrence group): the forward kernel's x-product workers hand the recurrence
 *             groups pre-multiplied gate tiles through a write-once history that every workspace layout keeps at offset 0,
 *             frame-major (frame t at the same address whatever T), and whose words carry the LAUNCH's parity in their least
 *             significant mantissa bit.  A call with ARMED / SAME_WS flips the parity the previous launch left and, when it runs
 *             more frames than that launch, re-tags only the frames beyond it; any other lstm_fwd zeroes the frames it will use
 *             first (0.8 GB at 3 x 512, B = 32, T = 1001: once per training run).  A launch that ended in a time-out
 *             invalidates the parity (lstm_status forgets it).
 * The bits are ignored by the paths that have no such panels.
 * Surviving a time-out (round 5).  The whole-sequence kernels need every workgroup of a launch resident at once; when another
 * process holds CUs (or a tool serialises kernels) their bounded waits give up, the launch ends within its limit and
 * amdspeech_lstm_status reports it: that mini-batch's results are invalid.  The caller then discards its gradient contribution
 * and repeats lstm_fwd AND lstm_bwd of the mini-batch with
 *   PER_DIAGONAL    this call runs on the launch-per-diagonal kernels (what AMDSPEECH_FLOW=0 and, at 1024 units, AMDSPEECH_BIG=0
 *                   select for a whole process: no kernel with a bounded wait; same workspace, same layout, same results) -- rnn-speech_amd/acoustic_model.py does exactly that, logs once
 *                   and goes on (the reference's loop never loses a step: models/AcousticModel.py:887-939);
 *   INJECT_TIMEOUT  testing only: the persistent kernels of THIS call (whole-sequence, or per layer at 1024 units) give up on
 *                   their first unsatisfied wait (limit 0); passed to lstm_bwd it does the same to THAT call's persistent kernels
 *                   (lstm_bwd_flow2 with its workers and the CTC leader, lstm_bwd_big / _big1).                                 */
enum { AMDSPEECH_LSTM_ARMED = 1, AMDSPEECH_LSTM_ARM_NEXT = 2, AMDSPEECH_LSTM_SAME_WS = 4, AMDSPEECH_LSTM_PER_DIAGONAL = 8,
       AMDSPEECH_LSTM_INJECT_TIMEOUT = 16 };

enum {
    AMDSPEECH_LSTM_WS_Z0 = 0,      /* float [T][B][H]  in : layer-0 input          */
    AMDSPEECH_LSTM_WS_ZTOP = 1,    /* float [T][B][H]  out: top layer output       */
    AMDSPEECH_LSTM_WS_DZTOP = 2,   /* float [T][B][H]  in : dLoss/d(top output)    */
    AMDSPEECH_LSTM_WS_DZ0 = 3,     /* float [T][B][H]  out: dLoss/d(layer-0 input) */
    AMDSPEECH_LSTM_WS_HFINAL = 4,  /* float [L][B][H] slices of the h history at T */
    AMDSPEECH_LSTM_WS_CFINAL = 5
};
size_t amdspeech_lstm_workspace_bytes(const amdspeech_lstm_desc* d);
/* Pointer of a named region inside `ws` (NULL on bad arguments). For
 * HFINAL/CFINAL the region of layer l is at ptr + l*(T+1)*B*H floats.        */
void* amdspeech_lstm_ws_ptr(const amdspeech_lstm_desc* d, void* ws, int which);

/* Ownership.  The library never allocates or frees a workspace and keeps no pointer into one EXCEPT between a call with
 * ARM_NEXT and the next lstm call on the same workspace (its side stream is then still writing the hand-off panels).  Before
 * freeing (or re-purposing) a workspace that has seen ARM_NEXT, call amdspeech_lstm_workspace_release: it makes `stream` wait
 * for that work and forgets the workspace; memory freed in stream order after it is safe.  All other state of the library is
 * per process and device, created on first use and never tied to caller memory: constant tables of the front end (twiddles,
 * mel filters, DCT; keyed by mode / sample rate / n_mfcc / device), one side stream + two events, the CU-masked streams of
 * the launch-per-diagonal overlap, the profiling events, and the calling thread's error text.  lstm_fwd / lstm_bwd are not
 * re-entrant on ONE workspace; calls on different workspaces are independent.                                               */
int amdspeech_lstm_workspace_release(void* stream, void* ws);

/* Work BESIDE the forward recurrence.  The whole-sequence forward kernel (H <= 512, L * ceil(B/16) <= 8 groups) keeps one
 * recurrence group per XCD (group g on XCD g) and leaves the other 8 - L * ceil(B/16) XCDs without work for the length of the
 * sequence (two of eight, ~5 ms, at 3 x 512 / batch 32).  This orders `stream` behind the point just in front of the last
 * amdspeech_lstm_fwd launch on `ws` and returns the number of idle XCDs (> 0); 0 (nothing ordered) when that call was not such
 * a launch, or (round 5: H = 512, exact f32) when the kernel's own x-product workers occupy the spare XCDs -- place the work
 * elsewhere (beside the CTC stage) -- unless the call carried the fused CTC head (amdspeech_lstm_fwd_ctc): there is no CTC stage
 * then, and the workgroups the kernel keeps in reserve on the spare XCDs are reported again.  What may follow on `stream`:
 *   - kernels small enough to share a CU with a recurrence workgroup (<= 32 VGPRs, no LDS to speak of: fills, packs): they run
 *     at once, everywhere;
 *   - WORK-QUEUE kernels (each workgroup pulls items from a counter until it is empty): the dispatcher deals the workgroups of
 *     any kernel round-robin to all eight XCDs, and those dealt to an XCD full of recurrence workgroups wait there until the
 *     recurrence ends -- but the ones on the idle XCDs drain the queue meanwhile, and the late ones find it empty.  The WORK is
 *     done beside the recurrence; the kernel (and whatever follows it in `stream`) completes when the recurrence does.  The
 *     front end's frame kernel is built this way (amdspeech_frontend_*).
 * A kernel with a fixed item per workgroup gains nothing here: 6/8 of it runs after the recurrence.  A stream confined to the
 * idle XCDs cannot be had (CU masks are one pattern for all XCDs).  Independent, short-lived work only: the dataflow kernels
 * spin on their siblings, so work that itself waited for them would deadlock.  Call it after amdspeech_lstm_fwd has returned. */
int amdspeech_lstm_beside_forward(void* stream, const void* ws);

/* Work BESIDE the weight-gradient launches that follow the backward recurrence (round 5).  amdspeech_lstm_bwd's whole-sequence
 * kernel is followed, on the caller's stream, by the products its in-kernel workers left over (three launches of 0.66 ms at the
 * headline shape, one 256-thread workgroup per CU: every CU has room for more waves).  This orders `stream` behind the point
 * BETWEEN that kernel and those launches and returns 1 | 2: 1 = ordered, 2 = DZ0 is complete at that point (the bottom layer's
 * groups formed it inside the kernel); 0 = the last amdspeech_lstm_bwd on `ws` was not such a launch (nothing ordered).  Short
 * products that only need the backward kernel's results -- dW_o / db_o from ZTOP and dlogits, dW_i / db_i from DZ0 -- then run
 * beside the first of the launches instead of behind the last.  The caller joins `stream` before it reads their results. */
int amdspeech_lstm_beside_tail(void* stream, const void* ws);

/* Forward over the whole stack.  h0/c0: [L][B][H] initial state or NULL (zeros)
 * -- the reference's persistent state Variables, :266-275.  lengths: int32 [B]
 * (device).  Frames t >= lengths[b] emit 0 and copy the state through.       */
int amdspeech_lstm_fwd(void* stream, const amdspeech_lstm_desc* d, void* ws,
                       const float* kernels, long kernel_stride,
                       const float* biases, long bias_stride,
                       const int* lengths, const float* h0, const float* c0);
/* TWO stacks of the same shape over the same batch, each in its own workspace -- the two directions of a bidirectional model
 * (BASELINE configs[4]; the caller hands stack B the time-reversed input in its Z0).  Same results as amdspeech_lstm_fwd(a ...) followed
 * by amdspeech_lstm_fwd(b ..., h0 = c0 = NULL).  Where the forward kernel can place a batch tile's recurrence group on ONE XCD (1024 wide
 * in plain bf16, ceil(B/16) <= 4: lstm_fwd_big1) every layer of the two stacks runs in ONE launch, stack A on XCDs 0 - 3 and stack B on
 * XCDs 4 - 7; everywhere else this IS the two calls -- amdspeech_lstm_pair_fusable(d) says which (1 = side by side), so that a caller
 * that keeps state between calls (the ARMED / ARM_NEXT cycle of the whole-sequence kernels) goes on making them itself.  (Two
 * amdspeech_lstm_fwd calls on two streams do not overlap: see lstm_big_fwd.h.)                                                      */
int amdspeech_lstm_pair_fusable(const amdspeech_lstm_desc* d);
int amdspeech_lstm_fwd_pair(void* stream, const amdspeech_lstm_desc* d_a, void* ws_a, const float* kernels_a, const float* biases_a,
                            const amdspeech_lstm_desc* d_b, void* ws_b, const float* kernels_b, const float* biases_b,
                            long kernel_stride, long bias_stride, const int* lengths, const float* h0_a, const float* c0_a);
/* ... and their backward passes (dZTOP of either workspace filled by the caller): the results of two amdspeech_lstm_bwd calls; side by
 * side under the same conditions (lstm_bwd_big1: W_hh^T as bf16 in the registers of one XCD per batch tile).                          */
int amdspeech_lstm_bwd_pair(void* stream, const amdspeech_lstm_desc* d_a, void* ws_a, const float* kernels_a, float* dkernels_a, float* dbiases_a,
                            const amdspeech_lstm_desc* d_b, void* ws_b, const float* kernels_b, float* dkernels_b, float* dbiases_b,
                            long kernel_stride, long bias_stride, const int* lengths);
/* Synchronous health check of the last forward/backward on `ws` (device sync + 4-byte
 * read): AMDSPEECH_ETIMEOUT if a bounded dataflow wait of the persistent kernels timed out (recoverable: repeat the mini-batch),
 * AMDSPEECH_EHIP if the read itself failed (a sticky device fault: not recoverable). */
int amdspeech_lstm_status(const amdspeech_lstm_desc* d, void* ws);
/* BPTT.  Reads DZTOP, the forward history in ws; writes DZ0 and ACCUMULATES
 * dK_l into dkernels + l*kernel_stride and db_l into dbiases + l*bias_stride. */
int amdspeech_lstm_bwd(void* stream, const amdspeech_lstm_desc* d, void* ws,
                       const float* kernels, long kernel_stride,
                       float* dkernels, float* dbiases, long bias_stride,
                       const int* lengths);

/* The CTC head INSIDE the whole-sequence kernels (round 5).  Between the two recurrence launches of a training step the reference
 * runs its output layer and tf.nn.ctc_loss (models/AcousticModel.py:241-247, :356-357); as separate launches (output Linear,
 * log-softmax, alpha / beta, gradient, dlogits . W_o^T) that is a serial 0.5 ms of a 12.7 ms step.  With a head attached
 *   lstm_fwd_ctc  also forms logits [T][B][C] = Z_top . W_o + b_o, their log-softmax and the alpha recursion, 16 frames behind the
 *                 top layer, on workgroups of the XCDs that carry no recurrence group: logits, loss [B] (= -log p(l|x), 0 for an
 *                 utterance whose targets do not fit its frames) and, in `ctc_ws`, log p / alpha / the extended targets are
 *                 complete when the launch is;
 *   lstm_bwd_ctc  runs beta, the posterior, dlogits [T][B][C] and dZ_top = dlogits . W_o^T ahead of the top layer's recurrence
 *                 groups on the workgroups that later form the weight gradients, then BPTT as lstm_bwd.  DZTOP is produced
 *                 inside the launch: the caller writes nothing there; dW_o / db_o (from ZTOP and dlogits) stay the caller's.
 * Same semantics as amdspeech_ctc_loss_fwd_bwd on the same logits: the loss is bit-identical (same device code), dlogits
 * equal to ~1e-7 (the order LDS atomics meet in).  A head on lstm_fwd_ctc obliges the caller to run lstm_bwd_ctc (not lstm_bwd)
 * for that mini-batch, or no backward pass at all.  amdspeech_lstm_ctc_fusable says whether a descriptor takes the head: the
 * whole-sequence kernels (H % 128 == 0, H <= 512, at least one XCD without a recurrence group), C a multiple of 16 up to 80,
 * 2 U + 1 <= 384 extended states, B within two utterances per follower team; AMDSPEECH_FLOW_CTC=0 switches it off.  `ctc_ws`:
 * amdspeech_ctc_workspace_bytes(T, B, C, U) bytes, 256-byte aligned, the same T as the descriptor's.                          */
typedef struct amdspeech_ctc_head {
    const float* w_out;        /* [H][C] output Linear weight (16-byte aligned) */
    const float* b_out;        /* [C] */
    float* logits;             /* out (fwd): [T][B][C] */
    const int* dense_labels;   /* [B][U] 0-padded targets (the reference's labels_ph) */
    float* loss;               /* out (fwd): [B] */
    float* dlogits;            /* out (bwd): [T][B][C]; may be NULL for lstm_fwd_ctc */
    void* ctc_ws;
    int C, U;
} amdspeech_ctc_head;
int amdspeech_lstm_ctc_fusable(const amdspeech_lstm_desc* d, int C, int U);
int amdspeech_lstm_fwd_ctc(void* stream, const amdspeech_lstm_desc* d, void* ws,
                           const float* kernels, long kernel_stride, const float* biases, long bias_stride,
                           const int* lengths, const float* h0, const float* c0, const amdspeech_ctc_head* head);
int amdspeech_lstm_bwd_ctc(void* stream, const amdspeech_lstm_desc* d, void* ws,
                           const float* kernels, long kernel_stride, float* dkernels, float* dbiases, long bias_stride,
                           const int* lengths, const amdspeech_ctc_head* head);

/* The inverted-dropout multipliers (mask / keep_prob, [T][B][H]) that lstm_fwd / lstm_bwd with this descriptor apply:
 * which = 0 the INPUT mask of `layer`, which = 1 its OUTPUT mask -- tf.contrib.rnn.DropoutWrapper(cell, input_keep_prob,
 * output_keep_prob), models/AcousticModel.py:227-233: independent masks per layer and side, scale 1/keep, the state is never
 * masked.  The masks are a pure function of (seed, layer, side, element index t*B*H + b*H + h); the export exists so that a
 * checker can run the reference's graph with the very masks the kernels used.                                            */
int amdspeech_lstm_dropout_multipliers(void* stream, const amdspeech_lstm_desc* d, int which, int layer, float* out);

/* ------------------------------------------------------------------- CTC ----
 * Replaces tf.nn.ctc_loss(sparse_labels, logits, seq_len,
 * ignore_longer_outputs_than_inputs=True) and its gradient,
 * models/AcousticModel.py:356-357, INCLUDING the label sparsification of
 * :155-159 / :174-178: `dense_labels` is the reference's labels_ph [B, U]
 * (0-padded); entries equal to 0 are dropped, an empty row becomes [C-1], the
 * target is every kept label before the first one >= C-1 (the blank / EOS),
 * required_time is the kept count; rows with lengths[b] == 0 or
 * required_time > lengths[b] get loss 0 and gradient 0.
 *   logits  [T,B,C]   loss [B]   dlogits [T,B,C] = d(sum_b loss_b)/dlogits   */
size_t amdspeech_ctc_workspace_bytes(int T, int B, int C, int U);
int amdspeech_ctc_loss_fwd_bwd(void* stream, const float* logits, const int* dense_labels,
                               const int* lengths, int T, int B, int C, int U,
                               float* loss, float* dlogits, void* ws);
/* The same in two calls, for a caller that wants to start other work on another stream in between: stage 1 = extended targets
 * + log-softmax into the workspace (short, fills the chip), stage 2 = the alpha / beta recursions and the gradient (long, 2 B
 * workgroups: most CUs are idle -- the product overlaps the next batch's front end here).  stage 0 = both (= the call above). */
int amdspeech_ctc_loss_fwd_bwd_staged(void* stream, const float* logits, const int* dense_labels,
                                      const int* lengths, int T, int B, int C, int U, float* loss,
                                      float* dlogits, void* ws, int stage);

/* Greedy decode: per-frame argmax (first maximum), collapse repeats, drop the
 * blank C-1.  Stands where tf.nn.ctc_beam_search_decoder sits at
 * models/AcousticModel.py:312.  ids [B,T] is padded with C (the
 * reference pads its dense prediction with num_labels, :718); out_len [B].
 * ws: int32 scratch of T*B elements.                                          */
int amdspeech_ctc_greedy_decode(void* stream, const float* logits, const int* lengths,
                                int T, int B, int C, int* ids, int* out_len, int* ws);

/* In-place collapse of consecutive duplicate labels of each decoded row (ids [B,T], lens [B],
 * both DEVICE): TensorFlow's merge_repeated=True post-processing of the top path (:312).   */
int amdspeech_merge_repeated(void* stream, int* ids, int* lens, int T, int B, int pad);

/* Levenshtein distance of n_pairs sequence pairs on the device (replaces tf.edit_distance at
 * models/AcousticModel.py:370, un-normalised): a [n_pairs, lda], b [n_pairs, ldb], lengths per
 * pair, out int32 [n_pairs].                                                               */
int amdspeech_edit_distance(void* stream, const int* a, const int* a_len, int lda, const int* b,
                            const int* b_len, int ldb, int n_pairs, int* out);

/* HOST-side CTC prefix beam search (evaluation path): stands where
 * tf.nn.ctc_beam_search_decoder(logits, seq_len) (beam_width 100, top_paths 1,
 * merge_repeated True) sits at models/AcousticModel.py:312.  ALL pointers are HOST
 * memory: logits [T,B,C], lengths [B]; outputs ids [B,T] padded with C, out_len [B],
 * log_prob [B] (may be NULL).  merge_repeated != 0 collapses consecutive duplicate labels
 * of the returned path, as TensorFlow's default does.                                */
int amdspeech_ctc_beam_search_host(const float* logits, const int* lengths, int T, int B, int C,
                                   int beam_width, int merge_repeated, int* ids, int* out_len,
                                   float* log_prob);
/* The same with a cap on the decode threads of the call (max_threads <= 0: one per utterance, bounded by the core
 * count, as above): the asynchronous training-time decoder (rnn_speech_amd.acoustic_model._AsyncBeamDecoder) uses it to
 * keep a steady load on a few cores beside the training thread.                                                   */
int amdspeech_ctc_beam_search_host_mt(const float* logits, const int* lengths, int T, int B, int C,
                                      int beam_width, int merge_repeated, int* ids, int* out_len,
                                      float* log_prob, int max_threads);

/* Levenshtein distance on the HOST (the same un-normalised tf.edit_distance as amdspeech_edit_distance, for predictions that
 * were decoded on the host: the asynchronous training-time beam decoder): all pointers HOST memory.                        */
int amdspeech_edit_distance_host(const int* a, const int* a_len, int lda, const int* b, const int* b_len, int ldb,
                                 int n_pairs, int* out);

/* CRC32C (Castagnoli) of a HOST buffer, continuing from `crc` (0 to start): the checksum
 * TensorFlow-bundle checkpoints carry per tensor and per table block (tf_bundle.py). */
uint32_t amdspeech_crc32c(const void* data, size_t n, uint32_t crc);

/* ------------------------------------------------------------- audio files ---
 * Host-side decode of RIFF/WAVE (PCM 8/16/24/32, IEEE float), FLAC (complete format, frame CRCs
 * always checked, STREAMINFO MD5 when verify != 0) and 16-bit PCM NIST SPHERE files to mono float32
 * in [-1, 1): the decode half of librosa.load(file) at util/audioprocessor.py:49 (channels averaged,
 * integer PCM scaled by 2^-(bits-1)).  `probe` reads the header only.  `decode` with out == NULL
 * reports frames / sample_rate; otherwise `capacity` must be >= frames.  HOST pointers.           */
int amdspeech_audio_probe(const char* path, int* sample_rate, int* channels, long* frames);
int amdspeech_audio_decode(const char* path, float* out, long capacity, long* frames, int* sample_rate,
                           int verify);

/* Resampler: the other half of librosa.load(file, sr=22050) at util/audioprocessor.py:49 -- band-limited
 * sinc interpolation with resampy's "kaiser_best" filter, on the GPU so that decoded PCM goes
 * H2D once and never comes back.  pcm [B, n_max] and out [B, out_max] are DEVICE buffers, n_samples a
 * HOST array; row b receives ceil(n_samples[b] * rate_out / rate_in) samples (amdspeech_resample_num_samples),
 * zero padded to out_max.                                                                          */
size_t amdspeech_resample_workspace_bytes(int B);
int amdspeech_resample_num_samples(int n_samples, int rate_in, int rate_out);
int amdspeech_resample(void* stream, const float* pcm, const int* n_samples, int B, int n_max, int rate_in,
                       int rate_out, float* out, int out_max, void* ws);

/* ------------------------------------------------------------- optimiser ----
 * Replaces tf.clip_by_global_norm + tf.train.AdamOptimizer.apply_gradients over
 * the flat parameter vector, models/AcousticModel.py:388 and :404-406.
 *   g' = g * clip / max(||g||_2, clip)
 *   m = b1 m + (1-b1) g' ; v = b2 v + (1-b2) g'^2 ; p -= lr_t m / (sqrt(v) + eps)
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is formed by the caller.  norm_out (device,
 * 1 float) receives ||g||_2.  ws: float scratch of
 * amdspeech_optim_workspace_bytes(n) bytes.                                   */
size_t amdspeech_optim_workspace_bytes(long n);
int amdspeech_clip_adam(void* stream, float* params, const float* grads, float* m, float* v,
                        long n, float clip, float lr_t, float beta1, float beta2, float eps,
                        float* norm_out, void* ws);

/* -------------------------------------------------------------- front end ---
 * Replaces AudioProcessor._extract_mfcc (librosa.feature.mfcc,
 * util/audioprocessor.py:63-75) and AudioProcessor._extract_fbank
 * (util/audioprocessor.py:77-161) for a batch of utterances.
 *   pcm        float [B][n_max]  mono samples, rows zero-padded
 *   n_samples  int32 [B] (HOST) valid samples per row
 *   feat       float [t_max][B][D] time-major, zero past each utterance's frames
 *   n_frames   int32 [B] (HOST, out) UNtruncated frame counts (reference quirk:
 *              the returned length is not clipped to max_input_seq_length)
 * mfcc: D = n_mfcc (reference default 20).  fbank: D = 120.                   */
size_t amdspeech_frontend_workspace_bytes(int mode, int B, int n_max, int sample_rate);
int amdspeech_frontend_num_frames(int mode, int n_samples, int sample_rate);
int amdspeech_frontend_mfcc(void* stream, const float* pcm, const int* n_samples, int B,
                            int n_max, int sample_rate, int n_mfcc, int t_max,
                            float* feat, int* n_frames, void* ws);
int amdspeech_frontend_fbank(void* stream, const float* pcm, const int* n_samples, int B,
                             int n_max, int sample_rate, int t_max,
                             float* feat, int* n_frames, void* ws);

/* ------------------------------------------------------------- profiling ----
 * Optional HIP-event timing of the recurrence kernels (no reference counterpart;
 * feeds bench.py's roofline line).  When enabled, lstm_fwd / lstm_bwd bracket
 * their recurrence kernel launches (and only those: the GEMMs of the per-layer
 * H = 1024 path are left out) with hipEvents on the caller's stream.
 * amdspeech_profile_get synchronises on the last recorded pairs and returns the
 * elapsed milliseconds and the number of TIME STEPS they covered: T + L - 1
 * diagonals for a whole-stack kernel or launch chain, T * L for the per-layer
 * kernels.  which: 0 = forward, 1 = backward.                                  */
int amdspeech_profile_enable(int on);
int amdspeech_profile_get(int which, float* elapsed_ms, int* time_steps);
/* Algorithmic FLOPs (2 per multiply-add) of the last whole-sequence dataflow launch of that direction, as the library itself
 * split the work: the recurrence's own products ([B,4H]x[4H,H]: L recurrent + L-1 "down" per frame, + dZ_0 when the bottom
 * layer's groups form it; forward: L x [B,2H]x[2H,4H]) and the other products computed INSIDE the same launch (the
 * weight-gradient share of the in-kernel GEMM workers).  Zeros when the last call took another kernel family.           */
int amdspeech_profile_get_flops(int which, double* recurrence_flops, double* other_flops);

/* -------------------------------------------------- data-parallel exchange ----
 * The reference trains on one device and reaches larger batches by ACCUMULATING
 * the gradients of `mini_batch_size` mini-batches before one clip + Adam
 * (models/AcousticModel.py:391-406, driver :916-926).  N data-parallel ranks
 * with one mini-batch each are that accumulation with N = mini_batch_size: every
 * rank sums its own utterances' gradients into its flat buffer, ONE fp32 SUM
 * all-reduce over RCCL (xGMI) makes every buffer the global sum, and every rank
 * applies the identical amdspeech_clip_adam.  No TensorFlow call is replaced:
 * the reference has no multi-device path.
 *
 * Bootstrap: rank 0 calls amdspeech_comm_unique_id and hands the
 * AMDSPEECH_COMM_ID_BYTES to every rank by any host-side means (torch.distributed
 * gloo broadcast, a file, MPI ...); every rank then calls amdspeech_comm_init
 * with its device current (hipSetDevice).  RCCL is bound with dlopen at the
 * first call -- AMDSPEECH_EUNSUPPORTED when no librccl.so can be found.
 * The collectives are enqueued on `stream`, in place; `n` floats.             */
#define AMDSPEECH_COMM_ID_BYTES 128
/* AMDSPEECH_OK when this process can bind RCCL (dlopen + the symbols the collectives need), an error code otherwise: what ranks
 * other than 0 probe with before anybody enters amdspeech_comm_init -- no id, no socket, no thread is created.                */
int amdspeech_comm_available(void);
int amdspeech_comm_unique_id(void* id_out);
int amdspeech_comm_init(const void* id, int rank, int world, void** comm_out);
int amdspeech_comm_destroy(void* comm);
/* What the communicator itself reports (ncclCommUserRank / ncclCommCount / ncclGetVersion) and the path of the RCCL shared
 * object that was bound -- diagnostics for a multi-GPU run (bench.py --gpus N prints them); any out pointer may be NULL.    */
int amdspeech_comm_info(void* comm, int* rank, int* world, int* rccl_version, char* lib_path, int lib_path_len);
int amdspeech_allreduce_sum_f32(void* comm, void* stream, float* buf, long n);
int amdspeech_broadcast_f32(void* comm, void* stream, float* buf, long n, int root);

/* ------------------------------------------------------ bidirectional glue ----
 * out[t,b,:] = in[len_b-1-t, b, :] for t < len_b, 0 beyond; time-major [T,B,H],
 * H a multiple of 4; accumulate != 0 adds into out.  The tf.reverse_sequence a
 * tf.nn.bidirectional_dynamic_rnn wraps around its backward-direction cells (the
 * reference builds a unidirectional dynamic_rnn, models/AcousticModel.py:276-278;
 * BASELINE.json configs[4] asks for the bidirectional variant).  Self-adjoint:
 * the same call reverses the gradients.                                        */
int amdspeech_reverse_sequences(void* stream, const float* in, float* out, const int* lengths,
                                int T, int B, int H, int accumulate);

/* ------------------------------------------------------------------ misc ----
 * y[i] += x[i] (gradient accumulation helper), y[i] = 0.                      */
int amdspeech_axpy(void* stream, float a, const float* x, float* y, long n);
int amdspeech_fill(void* stream, float* y, float value, long n);

#ifdef __cplusplus
}
#endif
#endif /* AMDSPEECH_H */
