/*
 * mvn.h -- C ABI of libmvn_hip.so: the MI355X (gfx950) Viterbi / ViterbiNet detection engine.
 *
 * This is the drop-in boundary for the 'val' hot path of tomerraviv95/meta-viterbinet.
 * The reference has no FFI of its own (pure Python); each entry point below
 * replaces the body of one reference function and is what a ctypes stub in the reference's
 * detector modules would bind (INTEGRATION.md shows that stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated otherwise; the caller owns all memory;
 *   - fp32 row-major; `*_ld` = row stride in elements; S = n_states = 2**memory_length,
 *     a power of two in [2,256] (the reference's np.uint8 bound, va_detector.py:43);
 *   - pointers need their type's natural alignment only (4 bytes for fp32); 16-byte aligned
 *     buffers and row strides that are multiples of 4 get the vector load/store paths;
 *   - calls enqueue work on `stream` (a hipStream_t; NULL = default stream) and return
 *     immediately: they never allocate, free or synchronise, so they can be graph-captured;
 *   - return value: 0 = ok, <0 = bad argument (MVN_E_*), >0 = hipError_t of the failed launch.
 *   - decisions are written as fp32 {0.,1.} like the reference's decoded_word
 *     (va_detector.py:90-93); columns >= T of `dec` are not touched (caller zero-fills).
 *   - arithmetic is IEEE fp32 with one rounding per reference operation; results are
 *     bit-identical to oracle/mvn_oracle.c for finite inputs, for +-inf, and for NaN -- in the samples y, in the weights
 *     and in the state priors: mvn_acs_block_f32, mvn_va_decode_f32, mvn_vnet_decode_f32 (every route, every S),
 *     mvn_acs_sweep_f32, mvn_vnet_decode_count_f32 and mvn_vnet_byword_step_f32 follow torch.min / torch.argmin (NaN when
 *     either candidate is NaN; the first NaN, else the first minimum).  The fast kernels' ACS stage is v_min_f32 (minNum),
 *     which agrees with torch.min unless SOME BUT NOT ALL of a symbol's branch costs are NaN (or +inf meets -inf in a path
 *     metric).  From y, weights and priors that takes a non-finite (or > 1e14 in magnitude) weight or prior, which the kernels
 *     look for once per launch and then run their NaN-propagating form (16-state kernels) or are followed by a guard launch of
 *     the generic kernel that re-decodes the affected blocks (sweep_kernel<S, MODE, true>: an early-exit no-op otherwise).
 *     MATERIALISED costs (mvn_acs_sweep_f32; the logits of the two-kernel ViterbiNet route) are tested one by one on their way
 *     into the recurrence: from the first cost that is NaN, infinite or >= 1e30 in magnitude on, a wave runs the
 *     NaN-propagating stage and decision.
 */
#ifndef MVN_H_
#define MVN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *mvn_stream_t; /* hipStream_t */

#define MVN_OK 0
#define MVN_E_DIMS (-1)      /* negative size, T > ld, ... */
#define MVN_E_STATES (-2)    /* S not a power of two in [2,256] */
#define MVN_E_PRIORS (-3)    /* Bp < 1 or B % Bp != 0 */
#define MVN_E_NULL (-4)      /* required pointer is NULL */
#define MVN_E_WORKSPACE (-5) /* workspace too small for one block */
#define MVN_E_DEVICE (-6)    /* current device is not gfx950 */

#define MVN_E_BARRIER (-7)   /* (status words only) a training launch abandoned its device-wide barrier */

#define MVN_ABI_VERSION 6 /* 6: workspace arguments on mvn_vnet_decode_count_f32 (the dealt 16-state kernel's hand-off lines), + the survivor / traceback entry points (incl. mvn_vnet_decode_surv_f32) and mvn_va_montecarlo_f32; 5: + mvn_vnet_train_kernel_name, mvn_va_byword_step_f32; 4: + trial-batched training / by-word step, status words; 3: training with a workspace; 2: kernel-name queries */

/* ABI version of the loaded library (== MVN_ABI_VERSION). */
int mvn_version(void);

/* Static message for a return code of this library (never NULL). */
const char *mvn_strerror(int code);

/* 0 if the current HIP device is a gfx950 part; fills optional outputs. Host pointers. */
int mvn_device_info(int *n_cu, int *lds_bytes_per_cu, char *arch_name, int arch_name_len);

/*
 * One ACS stage, acs_block of python_code/utils/trellis_utils.py:16-30:
 *   out[b,s] = min_j (in_prob+llrs)[b,(2s+j)%S];  argmin_j[b,s] in {0,1} (int64, first minimum wins,
 *   like torch.min(dim=2)); argmin_j may be NULL.  in_prob, llrs, out: [B,S] contiguous.
 */
int mvn_acs_block_f32(const float *in_prob, const float *llrs, float *out, int64_t *argmin_j,
                      int64_t B, int32_t S, mvn_stream_t stream);

/*
 * ACS sweep over materialised branch costs: the T-step loop
 *     dec[:,i] = argmin(in_prob,1) % 2 ; in_prob = acs_block(in_prob, cost[:,i])
 * of python_code/detectors/VA/va_detector.py:89-97 (== vnet_detector.py:53-59), with
 * acs_block = python_code/utils/trellis_utils.py:16-30 and in_prob starting at zero (:84).
 *   cost [B,T,S] contiguous; dec [B, dec_ld>=T]; final_metric [B,S] or NULL.
 */
int mvn_acs_sweep_f32(const float *cost, float *dec, int64_t dec_ld, float *final_metric,
                      int64_t B, int32_t T, int32_t S, mvn_stream_t stream);

/*
 * The ViterbiNet detector with survivors (round 5): mvn_vnet_decode_f32's decisions and final metrics -- the same bits -- plus the
 * survivor planes of its sweep (branch cost = -logit, python_code/detectors/VNET/vnet_detector.py:57; acs_block's second return value,
 * utils/trellis_utils.py:30), so that mvn_traceback_f32 yields the maximum-likelihood path through the learned branch metrics -- the
 * textbook ViterbiNet decision, where the reference decides from a running argmin.  16 states, T % 4 == 0, 8-byte-aligned surv,
 * 128-byte-aligned workspace: the fused detector itself (vnet16_dealt_kernel<false, true>: survivors out of its decision pass, no
 * logits in HBM).  Otherwise the two-kernel route: the logits of as many blocks as fit `workspace` (at least T*S*4 bytes, 16-byte
 * aligned), then the survivor sweep over them.  mvn_vnet_surv_workspace_bytes: what one pass wants (hand-off lines / B*T*S*4).
 *   surv [B, T, max(1, S/8)] as for mvn_acs_sweep_surv_f32.
 */
size_t mvn_vnet_surv_workspace_bytes(int64_t B, int32_t T, int32_t S);
int mvn_vnet_decode_surv_f32(const float *y, int64_t y_ld, const float *W1, const float *b1, const float *W2, const float *b2,
                             const float *W3, const float *b3, float *dec, int64_t dec_ld, float *final_metric, uint8_t *surv,
                             void *workspace, size_t workspace_bytes, int64_t B, int32_t T, int32_t S, mvn_stream_t stream);

/*
 * The same two sweeps WITH survivor (traceback) pointers -- optional: the reference computes them and drops them (acs_block
 * returns (values, argmin_j), python_code/utils/trellis_utils.py:30; its callers keep the values only, va_detector.py:95,
 * vnet_detector.py:57), so nothing on the parity path stores or reads them.  dec and final_metric are bit for bit those of
 * mvn_acs_sweep_f32 / mvn_va_decode_f32; in addition
 *     surv[b][t][s >> 3] bit (s & 7) = argmin_j of state s at stage t  (torch.min's index: the first minimum, the first NaN),
 *     the predecessor of state s on its surviving path being (2 s + j) % S            (trellis_utils.py:7-13)
 * i.e. max(1, S / 8) bytes per symbol, mvn_survivor_bytes(B, T, S) in all, any alignment (line-aligned rows are written as whole
 * 128-byte lines).  mvn_traceback_f32 walks them back from torch.argmin(final_metric[b]) and returns the textbook
 * maximum-likelihood path: bits [B, bits_ld >= T] fp32 {0,1}, bits[b][t] = the least-significant bit of the path's state before
 * stage t (the bit of symbol t, trellis_utils.py:33-46); states int32 [B, T] = those states, or NULL.
 */
size_t mvn_survivor_bytes(int64_t B, int32_t T, int32_t S);
int mvn_acs_sweep_surv_f32(const float *cost, float *dec, int64_t dec_ld, float *final_metric, uint8_t *surv,
                           int64_t B, int32_t T, int32_t S, mvn_stream_t stream);
int mvn_va_decode_surv_f32(const float *y, int64_t y_ld, const float *state_priors, int64_t Bp, float *dec,
                           int64_t dec_ld, float *final_metric, uint8_t *surv, int64_t B, int32_t T, int32_t S,
                           mvn_stream_t stream);
int mvn_traceback_f32(const uint8_t *surv, const float *final_metric, float *bits, int64_t bits_ld, int32_t *states,
                      int64_t B, int32_t T, int32_t S, mvn_stream_t stream);

/*
 * Introspection for profiling tools (no reference counterpart): name of the device kernel mvn_acs_sweep_f32
 * launches for these buffers and this shape on the current device: the dispatcher's own decision (same MVN_*
 * environment switches, same fall-backs for buffers that are not 16-byte aligned; `cost` / `dec` are only inspected
 * for their alignment and may be NULL = aligned).  `name` is a host pointer; returns 0, or MVN_E_STATES / MVN_E_NULL.
 */
int mvn_acs_sweep_kernel_name(const float *cost, const float *dec, int64_t dec_ld, int64_t B, int32_t T, int32_t S,
                              char *name, int32_t name_len);
/* The same for mvn_va_decode_f32 and mvn_vnet_decode_f32 (two launches are reported as "a + b"; the scratch sweep of
 * the two-launch ViterbiNet route is named for 16-byte aligned buffers). */
int mvn_va_decode_kernel_name(int64_t B, int32_t T, int32_t S, char *name, int32_t name_len);
int mvn_vnet_decode_kernel_name(int64_t B, int32_t T, int32_t S, int32_t want_logits, char *name, int32_t name_len);

/*
 * VADetector.forward(y,'val'), python_code/detectors/VA/va_detector.py:73-98, given the
 * state priors of compute_state_priors (:42-50) as a [Bp,S] table (row b uses b % Bp, the
 * `.repeat` of :64-65): branch costs (y-prior)^2/2 - log(sqrt(2*pi)) (:64-68) are computed in
 * registers, never written to HBM.   y [B, y_ld>=T].
 */
int mvn_va_decode_f32(const float *y, int64_t y_ld, const float *state_priors, int64_t Bp,
                      float *dec, int64_t dec_ld, float *final_metric, int64_t B, int32_t T,
                      int32_t S, mvn_stream_t stream);

/*
 * The ViterbiNet likelihood MLP net(y.reshape(-1,1)),
 * python_code/detectors/VNET/vnet_detector.py:27-33,49 (== meta_vnet_detector.py:26-32):
 * Linear(1,100) Sigmoid Linear(100,50) ReLU Linear(50,S).  Weights in torch layout:
 * W1[100,1] b1[100] W2[50,100] b2[50] W3[S,50] b3[S]; read at call time, never cached.
 *   y [N] contiguous; logits [N,S].
 */
int mvn_vnet_logits_f32(const float *y, const float *W1, const float *b1, const float *W2,
                        const float *b2, const float *W3, const float *b3, float *logits,
                        int64_t N, int32_t S, mvn_stream_t stream);

/* Bytes of scratch mvn_vnet_decode_f32 / mvn_vnet_decode_count_f32 want to run (B,T,S) in one pass.
 *  - Two-kernel route (S = 2; S = 256; every other S != 16 below the batch size from which the fused kernel is the faster
 *    route -- about 6 blocks per CU at 4 ... 64 states, 14 at 128; MVN_UNFUSED=1): the logits of the batch; any size that
 *    holds at least one block (T*S*4 bytes) is accepted and processed in slices.
 *  - S = 16, more than 768 blocks: at most 100 KB of hand-off lines for the dealt kernel (vnet16_dealt_kernel: the batch's
 *    32-symbol units shared evenly by 3 workgroups per CU, a block's 16 path metrics handed from wave to wave); 128-byte
 *    aligned, contents irrelevant before and after, not to be shared by calls that may run concurrently.  Without it
 *    (NULL / smaller / unaligned) the one-wave-per-block kernel runs: same bits, 4 % slower at 10 000 x 1000 and up to 2 x at
 *    a thousand blocks.  Word 0 of the workspace is a status word: non-zero after the call only if a hand-off wait was
 *    abandoned (the lower-numbered workgroup it waits for did not publish within seconds; the affected decisions are NaN).
 *  - 0 when the shape is served by a fused kernel that needs none (S = 4 ... 128 from those batch sizes, any batch and
 *    S = 256 with MVN_FUSED_IP=1; S = 16 up to 768 blocks of up to 1024 symbols: the cooperative kernel). */
size_t mvn_vnet_workspace_bytes(int64_t B, int32_t T, int32_t S);

/*
 * VNETDetector.forward(y,'val'), python_code/detectors/VNET/vnet_detector.py:35-61, and
 * META_VNETDetector.forward(y,'val',var), meta_vnet_detector.py:24-45 (var = the six arrays).
 *   y [B, y_ld>=T]; dec [B, dec_ld>=T]; logits_out [B,T,S] or NULL; final_metric [B,S] or NULL;
 *   workspace: device scratch of mvn_vnet_workspace_bytes(B, T, S) bytes (may be NULL when that is 0 or -- two-kernel route --
 *   logits_out is given; S = 16: optional, see there).
 *   With logits_out the logits are materialised there (S != 16: by the two-kernel route).
 */
int mvn_vnet_decode_f32(const float *y, int64_t y_ld, const float *W1, const float *b1,
                        const float *W2, const float *b2, const float *W3, const float *b3,
                        float *dec, int64_t dec_ld, float *logits_out, float *final_metric,
                        void *workspace, size_t workspace_bytes, int64_t B, int32_t T, int32_t S,
                        mvn_stream_t stream);

/*
 * One Monte-Carlo step of Trainer.single_eval_at_point (python_code/trainers/trainer.py:232-239) in ONE
 * launch, 16 states only: VNETDetector.forward(y,'val') with calculate_error_rates folded into the kernel
 * epilogue, so decisions never travel through HBM (4 B/symbol read, nothing written).
 *   tx [B, tx_ld>=K] fp32 {0,1}: transmitted words; the first K <= T columns are compared;
 *   row_mask: uint8[B] or NULL; rows with mask 0 (pilots, trainer.py:100-102) are decoded but not counted;
 *   counters: device int64[4], += {bit_errors, bits, frame_errors, frames};
 *   dec: optional [B, dec_ld>=T] decisions output (NULL = do not store);
 *   workspace / workspace_bytes: as for mvn_vnet_decode_f32 (mvn_vnet_workspace_bytes(B, T, 16); may be NULL / 0).
 * Returns MVN_E_STATES for S != 16 (use mvn_vnet_decode_f32 + mvn_count_errors there).
 */
int mvn_vnet_decode_count_f32(const float *y, int64_t y_ld, const float *W1, const float *b1,
                              const float *W2, const float *b2, const float *W3, const float *b3,
                              const float *tx, int64_t tx_ld, int32_t K, const uint8_t *row_mask,
                              int64_t *counters, float *dec, int64_t dec_ld, void *workspace,
                              size_t workspace_bytes, int64_t B, int32_t T, int32_t S, mvn_stream_t stream);

/*
 * calculate_error_rates, python_code/utils/metrics.py:7-17, as integer counters so that
 * 1/2/4/8-GPU results are identical: counters[0..3] += {bit_errors, bits, frame_errors, frames}
 * over rows `rows[i]` (int64 device array, or NULL = rows 0..n_rows-1) and the first K columns.
 * counters: device int64[4], accumulated into (caller zeroes it).
 */
int mvn_count_errors(const float *dec, int64_t dec_ld, const float *tx, int64_t tx_ld,
                     const int64_t *rows, int64_t n_rows, int32_t K, int64_t *counters,
                     mvn_stream_t stream);

/*
 * Online (self-supervised) training of the ViterbiNet MLP on ONE word, all iterations in one launch
 * (SURVEY 8f next #3): VNETTrainer.online_training, python_code/trainers/VNET/vnet_trainer.py:49-60 ->
 * Trainer.run_train_loop, trainers/trainer.py:492-505 with CrossEntropyLoss(mean) and torch.optim.Adam
 * (amsgrad off, no weight decay), i.e. n_iter x { logits = net(y); loss = CE(logits[idx_i], labels[idx_i]);
 * backward; Adam.step }.
 *   y [T] received word; labels [T] int32 trellis states (calculate_states, trellis_utils.py:33-46);
 *   batch_idx [n_iter, M] int32 sample indices per iteration (select_batch, trainer.py:534-544), or NULL = all T
 *   samples every iteration (metavnet_trainer.py:41-50);
 *   W1..b3: parameters, updated in place; adam_m/adam_v: [P] exp_avg / exp_avg_sq in parameter order
 *   (P = 250 + 50*100 + 51*S), updated in place; step0 = Adam steps already taken;
 *   loss_out [n_iter] or NULL.  S <= 128 (64 / 128 states: one workgroup per trial, the optimizer's moments stay in global
 *   memory).  fp32; agrees with torch autograd + Adam to rounding, not bitwise.
 * The trainer's other optimizers (deep_learning_setup, trainer.py:163-175) ride on the same arguments: beta1 = MVN_BETA1_RMSPROP
 * runs torch.optim.RMSprop's update (alpha = beta2, eps; square average in adam_v, adam_m untouched; torch's defaults: no
 * momentum, not centered), beta1 = MVN_BETA1_SGD torch.optim.SGD's (p -= lr g; adam_m / adam_v untouched).  The same holds for
 * the _ws_ and _trials_ forms of this call; the meta-learning calls below take Adam only.
 */
#define MVN_BETA1_RMSPROP (-1.0f)
#define MVN_BETA1_SGD (-2.0f)
int mvn_vnet_online_train_f32(const float *y, const int32_t *labels, int32_t T, const int32_t *batch_idx, int32_t M,
                              int32_t n_iter, float *W1, float *b1, float *W2, float *b2, float *W3, float *b3,
                              float *adam_m, float *adam_v, int64_t step0, float lr, float beta1, float beta2, float eps,
                              float *loss_out, int32_t S, mvn_stream_t stream);

/*
 * The same call spread over one workgroup per 32-sample chunk when every iteration uses the whole word (batch_idx == NULL,
 * the Meta-ViterbiNet variant, metavnet_trainer.py:41-64): `workspace` = mvn_vnet_train_workspace_bytes(S) bytes of
 * device memory, 16-byte aligned, used for the gradient exchange between the workgroups (contents irrelevant before and
 * after; must not be shared by calls that may run concurrently).  Results are bit-identical to
 * mvn_vnet_online_train_f32, which also serves every case this form does not (minibatch iterations, words of at most
 * 32 symbols or more than 1024, workspace NULL or too small, MVN_TRAIN_GROUPS=0 in the environment).
 */
size_t mvn_vnet_train_workspace_bytes(int32_t n_states);
/* status: device int32[1] or NULL.  The workgroups of this form meet at a device-wide barrier whose wait is bounded (a launch
 * that cannot become resident must not hang the device).  If a wait is abandoned the weights come back as NaN AND *status is
 * set to 1 (the caller zeroes it; it is never written otherwise): check it wherever the host next synchronises -- the
 * counterpart of the reference's NaN-loss guard, trainer.py:496-498.  Concurrent launches of this form (different streams,
 * different workspaces) are safe while their combined workgroups (one per 32-sample chunk) fit the device's CUs -- words of up
 * to 256 symbols are placed on ONE XCD each (their gradient exchange stays in its L2), the XCD taken in rotation from launch to
 * launch, so that bound is then 32 CUs per XCD: at least 4 such launches per XCD, 32 on the device; to run many words at once
 * use the *_trials_* entry points below, which put them into ONE launch. */
int mvn_vnet_online_train_ws_f32(const float *y, const int32_t *labels, int32_t T, const int32_t *batch_idx, int32_t M,
                                 int32_t n_iter, float *W1, float *b1, float *W2, float *b2, float *W3, float *b3,
                                 float *adam_m, float *adam_v, int64_t step0, float lr, float beta1, float beta2, float eps,
                                 float *loss_out, int32_t S, void *workspace, size_t workspace_bytes, int32_t *status,
                                 mvn_stream_t stream);

/*
 * n_steps online meta-learning steps of Meta-ViterbiNet in ONE launch: Trainer.meta_train_loop
 * (python_code/trainers/trainer.py:425-453) with METAVNETTrainer.calc_loss (metavnet_trainer.py:41-50):
 *   inner SGD step on the support words (lr = meta_lr), query loss through the updated weights, meta-gradient w.r.t.
 *   the original weights (second_order != 0: MAML with the exact Hessian-vector product; 0: first order), one Adam step.
 *   rx_words [Nw, T] fp32 received words and labels [Nw, T] int32 trellis states (calculate_states of the buffered
 *   transmitted words); step k uses the W words support_idx[k*W .. k*W+W-1] and the word query_idx[k] (indices into
 *   the Nw words, already non-negative); weights, adam_m/adam_v [P] and step0 as in mvn_vnet_online_train_f32 (the
 *   reference has one optimizer for both loops); loss_out [n_steps] query losses or NULL.  S <= 32.
 * fp32, deterministic; agrees with torch's double backward + Adam to rounding (tolerance in the tests).
 */
int mvn_vnet_maml_train_f32(const float *rx_words, const int32_t *labels, int32_t T, const int32_t *support_idx, int32_t W,
                            const int32_t *query_idx, int32_t n_steps, float *W1, float *b1, float *W2, float *b2,
                            float *W3, float *b3, float *adam_m, float *adam_v, int64_t step0, float meta_lr,
                            int32_t second_order, float lr, float beta1, float beta2, float eps, float *loss_out, int32_t S,
                            mvn_stream_t stream);

/*
 * The same steps with one workgroup per chunk of a pass (support gradient, query gradient, Hessian-vector product):
 * `workspace` as for mvn_vnet_online_train_ws_f32 (gradient exchange + private copies of the Adam moments).  Bit-identical
 * to mvn_vnet_maml_train_f32, which also serves the cases this form does not (a pass of more than 32 chunks or of one,
 * workspace NULL or too small, MVN_TRAIN_GROUPS=0 in the environment).
 */
int mvn_vnet_maml_train_ws_f32(const float *rx_words, const int32_t *labels, int32_t T, const int32_t *support_idx, int32_t W,
                               const int32_t *query_idx, int32_t n_steps, float *W1, float *b1, float *W2, float *b2,
                               float *W3, float *b3, float *adam_m, float *adam_v, int64_t step0, float meta_lr,
                               int32_t second_order, float lr, float beta1, float beta2, float eps, float *loss_out, int32_t S,
                               void *workspace, size_t workspace_bytes, int32_t *status, mvn_stream_t stream);

/*
 * R independent trials of the two training calls above in ONE launch sequence (the evaluations with online
 * training between blocks do not shard within a trial, so the parallel axis is the (SNR x seed x method) grid the reference
 * walks serially, python_code/plotters/plotter_main.py:117-149; a trial alone occupies 1-9 of the 256 CUs).  Every trial
 * has its own weights, Adam state, word(s), draws and step count, described by one mvn_train_trial_t in a DEVICE array;
 * what is common to the launch (T, M / W, the optimizer's constants, S) is passed by value.  gridDim = (chunks, trials): a
 * trial's workgroups only ever synchronise with each other (per-trial barrier counter in the workspace), trials with n = 0
 * exit at once, and a launch never holds more workgroups than the device has CUs (more trials = more launches on `stream`).
 * Per trial the results are bit-identical to the single-trial entry points.
 */
typedef struct mvn_train_trial {
    const float *y;           /* online: the word [T]; maml: the buffered received words [Nw, T] */
    const int32_t *labels;    /* online: [T]; maml: [Nw, T] (calculate_states of the buffered label words) */
    const int32_t *idx;       /* online: batch_idx [n, M] or NULL = whole word; maml: support_idx [n, W] */
    const int32_t *query_idx; /* maml: [n]; online: unused */
    const float *w_in[6];     /* W1, b1, W2, b2, W3, b3 read when the trial starts (e.g. the saved detector, trainer.py:275) */
    float *w_out[6];          /* ... written when it ends (may alias w_in) */
    float *w_out2[6];         /* optional second copy of the result (all six NULL: none) */
    float *adam_m, *adam_v;   /* [P] exp_avg / exp_avg_sq, updated in place */
    float *loss_out;          /* [n] or NULL */
    int32_t *status;          /* int32[1] or NULL: set to 1 if the trial's barrier wait was abandoned (weights = NaN) */
    double b1pow, b2pow;      /* (double)beta1 ** step0, (double)beta2 ** step0: Adam steps this trial has already taken */
    int32_t n;                /* iterations (online) / meta-learning steps (maml) of this trial; 0 = skip the trial */
    int32_t reserved;
} mvn_train_trial_t;

/* Device workspace for R trials of words of T symbols (W support words for the meta-learning step); 0 when the shapes run
 * on one workgroup per trial (minibatch iterations, T <= 32) and need none. */
size_t mvn_vnet_train_trials_workspace_bytes(int32_t n_states, int32_t T, int32_t W, int32_t R);
/* M = 0: every iteration uses the whole word (idx NULL in every trial); M > 0: minibatches of M samples (idx given). */
int mvn_vnet_online_train_trials_f32(const mvn_train_trial_t *trials, int32_t R, int32_t T, int32_t M, float lr, float beta1,
                                     float beta2, float eps, int32_t S, void *workspace, size_t workspace_bytes,
                                     mvn_stream_t stream);
int mvn_vnet_maml_train_trials_f32(const mvn_train_trial_t *trials, int32_t R, int32_t T, int32_t W, float meta_lr,
                                   int32_t second_order, float lr, float beta1, float beta2, float eps, int32_t S,
                                   void *workspace, size_t workspace_bytes, mvn_stream_t stream);

/*
 * Introspection for tests and profiling tools (no reference counterpart): which device kernel, in which form, a training
 * call launches for these shapes on the current device -- the launchers' own decision (MVN_TRAIN_GROUPS switch, CU count,
 * the "one workgroup per chunk and trial" / "one workgroup per trial" rule for many trials).
 *   kind: 0 = mvn_vnet_online_train_*, 1 = mvn_vnet_maml_train_* first order, 2 = second order;
 *   R: 0 = the single-trial entry points, > 0 = the *_trials_* entry points with R trials that run (n > 0);
 *   M_or_W: minibatch size M (0 = whole word) for kind 0, support words W for kinds 1, 2;
 *   workspace_bytes: what the call passes (0 = no workspace).
 * name (HOST pointer) receives e.g. "maml_train_kernel<16, true> 1x176" or "online_train_groups_kernel<16, true> 5x48 in 2
 * launches one XCD per trial" (workgroups per trial x trials per launch; "one XCD per trial": the grid that places a trial's
 * workgroups on one XCD so that their gradient exchange stays in its L2 -- MVN_TRAIN_XCD=0 keeps the (groups, trials) grid).
 * Returns 0, MVN_E_DIMS, MVN_E_STATES or MVN_E_NULL.
 */
int mvn_vnet_train_kernel_name(int32_t kind, int32_t R, int32_t T, int32_t M_or_W, int32_t S, size_t workspace_bytes, char *name,
                               int32_t name_len);

/*
 * One block step of Trainer.eval_by_word (python_code/trainers/trainer.py:292-316 and the buffer entry of :322-324) for R
 * words in ONE launch, 16 states, Reed-Solomon outer code with nsym <= 8 parity bytes:
 *   data step (pilot = 0): dec = VNETDetector.forward(rx,'val'); msg = RS-decode(dec); nerr = bit errors of msg against tx;
 *                          enc = RS-encode(msg);
 *   pilot step (pilot = 1): enc = RS-encode(tx); nerr = 0; dec and msg are not written (the reference detects the pilot too
 *                          but never uses the result);
 *   label_word = dec if nerr > 0 else enc (what the reference pushes into its buffer), labels = its trellis states
 *   (calculate_states, utils/trellis_utils.py:33-46).
 * Word r uses the weights W1 + r * w_stride[0], b1 + r * w_stride[1], ... (w_stride: HOST int64[6], NULL = one set for all):
 * R independent trials advance one block per launch; R = 1 is the reference's sequential pattern (one launch instead of
 * detect, RS decode, count, RS encode).
 *   rx [R, rx_ld >= T]; tx [R, tx_ld >= T - 8 nsym] message bits; dec / enc / label_word [R, ld >= T] or NULL;
 *   msg [R, msg_ld >= T - 8 nsym] or NULL; labels int32 [R, lab_ld >= T] or NULL; nerr int32 [R] or NULL.
 * T a multiple of 8, T <= 1024, T / 8 > nsym.  Decisions, decoded and encoded words are bit-identical to
 * mvn_vnet_decode_f32 + mvn_rs_decode_bits_f32 + mvn_rs_encode_bits_f32.  Returns MVN_E_STATES for S != 16 and MVN_E_DIMS
 * for nsym > 8 (use the separate entry points there).
 */
int mvn_vnet_byword_step_f32(const float *rx, int64_t rx_ld, const float *tx, int64_t tx_ld, const float *W1, const float *b1,
                             const float *W2, const float *b2, const float *W3, const float *b3, const int64_t *w_stride,
                             float *dec, int64_t dec_ld, float *msg, int64_t msg_ld, float *enc, int64_t enc_ld,
                             float *label_word, int64_t lw_ld, int32_t *labels, int64_t lab_ld, int32_t *nerr, int64_t R,
                             int32_t T, int32_t nsym, int32_t pilot, int32_t S, mvn_stream_t stream);

/*
 * The same block step for the classical Viterbi detector (the reference's eval_by_word takes any detector, trainer.py:295):
 * dec = VADetector.forward(rx, 'val', snr, gamma, count) with the state priors given as in mvn_va_decode_f32 -- word r uses
 * row r % Bp of state_priors [Bp, 16] (compute_state_priors, va_detector.py:42-50, of the word's channel) -- then exactly
 * the codec, label word and trellis states of mvn_vnet_byword_step_f32; same outputs, same limits (16 states, T a multiple of
 * 8, T <= 1024, nsym <= 8).  One wavefront per word.  Decisions are bit-identical to mvn_va_decode_f32, decoded / encoded
 * words to mvn_rs_decode_bits_f32 / mvn_rs_encode_bits_f32.
 */
int mvn_va_byword_step_f32(const float *rx, int64_t rx_ld, const float *tx, int64_t tx_ld, const float *state_priors, int64_t Bp,
                           float *dec, int64_t dec_ld, float *msg, int64_t msg_ld, float *enc, int64_t enc_ld, float *label_word,
                           int64_t lw_ld, int32_t *labels, int64_t lab_ld, int32_t *nerr, int64_t R, int32_t T, int32_t nsym,
                           int32_t pilot, int32_t S, mvn_stream_t stream);

/* The MVN_* environment switches (A/B variants of the kernels) are read once per process; a caller that
 * changes them afterwards (the test-suite does) calls this to have them read again. */
void mvn_reload_switches(void);

/*
 * ISI-AWGN channel: ChannelModelDataset.transmit / ISIAWGNChannel.transmit,
 * python_code/channel/channel_dataset.py:71,87-95 + channel.py:12-35 + modulator.py:12:
 *   y[b,t] = sum_i h[b % Bh][L-1-i] * (1 - 2 c[b,t+i]) + sigma * w[b,t],   c = bits zero-padded past K,
 * evaluated in float64 and stored fp32 (channel_dataset.py:103).  sigma = (10**(snr/10))**-0.5 (channel.py:23,31).
 *   bits [B, ld_bits>=K] fp32 {0,1}; noise [B,T] standard-normal draws, float64 (noise_is_f64=1) or fp32, or NULL;
 *   h [Bh, L] float64 taps (estimate_channel rows); y [B, y_ld>=T].  T is normally K (transmission length).
 */
int mvn_isi_awgn_transmit(const float *bits, int64_t ld_bits, int32_t K, const void *noise, int32_t noise_is_f64,
                          const double *h, int64_t Bh, double sigma, float *y, int64_t y_ld, int64_t B, int32_t T,
                          int32_t L, mvn_stream_t stream);

/*
 * On-device word generator: the inner loop of ChannelModelDataset.get_snr_data for uncoded words
 * (python_code/channel/channel_dataset.py:65-83: random bits, zero padding by L, BPSK (modulator.py:12),
 * ISIAWGNChannel.transmit (channel.py:12-35)) fused in one kernel: bits ~ Bernoulli(1/2) and noise ~ N(0,1) from the
 * counter-based Philox4x32-10 generator (a pure function of (seed, word, position): same DISTRIBUTION as the
 * reference's two RandomState streams, not the same stream -- mvn_isi_awgn_transmit replays recorded draws bit for bit).
 *   tx [B, tx_ld>=T] transmitted bits as fp32 {0.,1.} (NULL: not stored); y [B, y_ld>=T] received words;
 *   h [Bh,L] float64 taps (row b % Bh for word b), sigma = 10^(-snr/20) (channel.py:23,31); L <= 16.
 */
int mvn_generate_words_f32(float *tx, int64_t tx_ld, float *y, int64_t y_ld, const double *h, int64_t Bh, double sigma,
                           uint64_t seed, int64_t B, int32_t T, int32_t L, mvn_stream_t stream);

/*
 * One uncoded Monte-Carlo point of the classical Viterbi detector in ONE launch: the words of mvn_generate_words_f32 (same seed ->
 * the same bits and samples, bit for bit) are generated inside the detector of mvn_va_decode_f32 and the decisions compared with
 * the generated bits on the spot -- Trainer.single_eval_at_point, python_code/trainers/trainer.py:222-241 (draw words,
 * detector(rx, 'val'), calculate_error_rates) with no tx, y or decisions in memory ("fused into the decode kernel so y
 * never touches HBM").  counters[0..3] += {bit_errors, bits, frame_errors, frames} over B words of T symbols, exactly the counters
 * of mvn_generate_words_f32 -> mvn_va_decode_f32 -> mvn_count_errors(K = T).
 *   h [Bh,L] float64 taps (row b % Bh), sigma = 10^(-snr/20); state_priors [Bp,S] (row b % Bp: VADetector.compute_state_priors);
 *   S = 16 or 256 (MVN_E_STATES otherwise: run the three launches); counters: device int64[4], accumulated into.
 */
int mvn_va_montecarlo_f32(const double *h, int64_t Bh, double sigma, uint64_t seed, const float *state_priors, int64_t Bp,
                          int64_t *counters, int64_t B, int32_t T, int32_t L, int32_t S, mvn_stream_t stream);

/*
 * Reed-Solomon outer code, batched over words; bits are fp32 {0,1}, 8 per GF(2^8)
 * symbol, MSB first (np.packbits).  Same code and same behaviour past the correction capacity as
 * python_code/ecc/rs_main.py: encode (:9-18) / decode (:21-37) -- prim 0x11d, generator 2, nsym parity bytes,
 * nbits/8 <= 255, nsym <= 64.
 *   decode: rx_bits [B, ld_in >= nbits] -> msg_bits [B, ld_out >= nbits - 8*nsym];
 *           status: int32[B] or NULL: 0 decoded, 1 = more errors than nsym/2 detected (uncorrected systematic
 *           part returned, rs_main.py:31-33), 2 = the reference would raise (never observed);
 *   encode: msg_bits [B, ld_in >= kbits] -> cw_bits [B, ld_out >= kbits + 8*nsym].
 */
int mvn_rs_decode_bits_f32(const float *rx_bits, int64_t ld_in, float *msg_bits, int64_t ld_out,
                           int32_t *status, int64_t B, int32_t nbits, int32_t nsym, mvn_stream_t stream);
int mvn_rs_encode_bits_f32(const float *msg_bits, int64_t ld_in, float *cw_bits, int64_t ld_out, int64_t B,
                           int32_t kbits, int32_t nsym, mvn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MVN_H_ */
