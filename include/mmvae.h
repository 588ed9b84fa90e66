/*
 * mmvae.h -- C ABI of the MI355X-native cpl-mixVAE train-step engine (libmmvae_hip.so).
 *
 * The reference (AllenInstitute/distributed-vae) has no FFI / operator layer: its hot path is
 * plain PyTorch.  This header is therefore the boundary the build
 * *introduces*; each entry point names the reference code it replaces:
 *
 *   mmvae_forward      mixVAE_model.forward          mmidas/nn_model.py:297-368
 *                      (encoder :263-269, double softmax :337, gumbel_softmax :430-493,
 *                       intermed :271-275, reparameterize :413-428, decoder :277-287)
 *   mmvae_loss         mixVAE_model.loss             mmidas/nn_model.py:495-598 (helpers :39-86)
 *   mmvae_backward     _loss.backward()              mmidas/cpl_mixvae.py:462 (autograd of the above)
 *   mmvae_adam_step    optimizer.step()              mmidas/cpl_mixvae.py:274,:463; train.py:144-147
 *   mmvae_train_step   the per-batch driver          mmidas/cpl_mixvae.py:434-463
 *   mmvae_eval_classify / mmvae_confmat_accumulate / mmvae_consensus
 *                      the per-epoch consensus loop  mmidas/cpl_mixvae.py:563-657, _utils.py:79-129
 *   mmvae_augment      netA(x.expand(A,-1,-1), True, 0.1)  mmidas/cpl_mixvae.py:422-423, augmentation/udagan.py:281-329
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer is caller-owned DEVICE memory (fp32 unless
 *     noted); the library never allocates or frees device memory, creates no streams or events and keeps no
 *     global mutable state besides the last error string (thread-local).  Everything a call needs beyond its
 *     arguments -- the optional second stream, the events that fork and join it, split factors, experiment
 *     switches -- travels in a caller-owned mmvae_exec passed to the call; two engines never share any.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), re-entrant, with no
 *     hidden synchronisation.
 *   - return value: 0 = ok, <0 = error (MMVAE_E_*); mmvae_last_error_string() explains.
 *   - layouts are row-major.  Parameters live in ONE flat fp32 buffer, arm-major:
 *     params[a * per_arm + offset[t]], t indexing the 28 tensors listed at mmvae_param_layout;
 *     weights keep PyTorch's [out, in] layout so state_dict tensors are views of the buffer.
 */
#ifndef MMVAE_H
#define MMVAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMVAE_OK 0
#define MMVAE_E_BADARG (-1)      /* null pointer, non-positive size ...                       */
#define MMVAE_E_UNSUPPORTED (-2) /* shape outside the kernels' limits (see mmvae_check_dims)  */
#define MMVAE_E_LAUNCH (-3)      /* hipLaunch / runtime failure                               */
#define MMVAE_E_WORKSPACE (-4)   /* workspace too small                                       */

#define MMVAE_MAX_ARMS 8
#define MMVAE_N_PARAM_TENSORS 28
#define MMVAE_N_BN 6

/* A arms, B cells per batch (this rank), D genes, H fc_dim, L lowD_dim, C n_categories,
 * S state_dim  (mixVAE_model.__init__, nn_model.py:112-134). */
typedef struct mmvae_dims {
    int32_t A, B, D, H, L, C, S;
} mmvae_dims;

typedef struct mmvae_hyper {
    float tau;         /* nn_model.py:337                                   */
    float temp;        /* Gumbel-softmax temperature, :455                  */
    float beta;        /* KL weight, :551                                   */
    float lam;         /* coupling weight, :581                             */
    float eps;         /* self.eps, also the BatchNorm eps, :211            */
    float bn_momentum; /* :211                                              */
    float x_drop;      /* input dropout p, :165/:264                        */
    float s_drop;      /* state dropout p, :166/:278                        */
    int32_t hard;      /* straight-through one-hot sample, :486-493         */
    int32_t training;  /* module.training: batch-stat BN + dropout active   */
    int32_t eval_flag; /* forward(eval=True): no Gumbel noise, hard sample, :340-343 */
    int32_t gemm_bf16; /* engine of the five D x H GEMMs (fc1, fc11, d(d10), dW1, dW11); low byte:
                          0  fp32 operands on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32: an exact fp32 FMA chain);
                          1  bf16 operands (rounded to nearest even on load), fp32 accumulation on the bf16 matrix pipe --
                             BASELINE.json's bf16 configuration;
                          2  "fp32x3": fp32 operands, each split EXACTLY into three bf16 slices (8 + 8 + 8 significand bits),
                             a product formed from six of the nine slice products on the bf16 matrix pipe with fp32
                             accumulation; what is dropped is <= 2^-26 of |a b|, below the fp32 rounding of the accumulation.
                             fp32-grade results at 6/64 of the matrix-pipe time of engine 0 (the Python binding's "fp32").
                          Every other computation and all parameters stay fp32 under all three.  Engines 1 and 2 need the
                          fast path (D % 4 == 0, fc_dim % 4 == 0, fc_dim <= 124; engine 2's fused fc11 kernel fc_dim <= 111),
                          else engine 0 runs.  Bits 8..11 (diagnostics, engine 2 only): products that stay on engine 0
                          (1 fc1, 2 fc11 + d(d10), 4 dW1, 8 dW11). */
    uint32_t cat_mask[4]; /* category subset of forward(mask=...) (nn_model.py:332-335, the pruning-time forward; eval_model passes
                          the categories whose fcc bias is non-zero, cpl_mixvae.py:1476-1478): bit k of the 128-bit mask set =
                          category k is kept; c = softmax(c_prob[:, kept] / tau) on the kept categories and 0 elsewhere.
                          All four words zero = no mask (every category kept). */
} mmvae_hyper;

/* Noise descriptor.  mode 0 = explicit buffers (parity tests; the reference's RNG stream cannot
 * be replayed on a GPU), mode 1 = in-kernel Philox4x32-10 keyed by (seed, offset), the
 * throughput mode.  Consumption order of the reference per arm: bernoulli[B,D] -> rand[B,C] ->
 * rand_like[B,S] -> bernoulli[B,S] iff s_drop>0. */
typedef struct mmvae_noise {
    int32_t mode;
    int32_t _pad;
    const uint8_t *x_mask;  /* [A,B,D] keep-mask (1 = keep), used iff training && x_drop>0 */
    const float *u_gumbel;  /* [A,B,C] U(0,1), used iff !eval_flag                         */
    const float *u_state;   /* [A,B,S] U(0,1) (uniform, as nn_model.py:427 draws it)       */
    const uint8_t *s_mask;  /* [A,B,S] keep-mask, used iff training && s_drop>0            */
    uint64_t seed;
    uint64_t offset;        /* advance by 1 per step                                       */
} mmvae_noise;


/* Caller-owned execution context (one per engine / workspace; NULL = defaults: single stream, automatic split
 * factors).  The library reads it during a call and writes only `early_recorded`.
 *   side_stream  optional second hipStream_t on the same device.  With it mmvae_backward / mmvae_train_step run the
 *                [dW11 | db11] GEMM, the coupling terms and the loss scalars beside the latency-bound chains of the main
 *                stream (fork / join by the events below).  The caller keeps stream and events alive while work that
 *                uses them is in flight.
 *   ev           MMVAE_N_EVENTS hipEvent_t (hipEventDisableTiming suffices), all non-NULL when side_stream is.  The library records
 *                them itself -- with hipEventRecord, or as the stop event of one of its own kernel launches (hipExtLaunchKernel):
 *                either way an event belongs to ONE engine and is not to be recorded or waited on by the caller while a call
 *                that uses it is in flight.
 *   early_grad_event  data-parallel overlap: when non-NULL (and side_stream is set), mmvae_backward and
 *                mmvae_train_step(do_adam == 0) reduce the gradients of fc11.weight / fc11.bias -- the last two
 *                tensors of every arm's segment of `grads`, 47 % of the parameters -- as soon as their GEMM has
 *                finished and record this event on the side stream, so the caller can start the all-reduce of those
 *                ranges while the rest of backward runs (replaces the reference's FSDP gradient traffic,
 *                train.py:140-143).  All other gradients are final when the call's work on `stream` is.
 *   early_recorded  out: 1 if the call recorded early_grad_event (it does not on shapes the fast kernels do not
 *                take, without a side stream, or with do_adam != 0): only then may the caller wait on it.
 *   split        split factors of the large GEMMs, 0 = automatic: 0 fc1 split-K, 1 fc11 column splits, 2 dW1 batch
 *                splits, 3 small-layer dW batch splits, 4 d(d10) gene splits, 5 dW11 batch splits.  They change the
 *                workspace layout: pass the same context to mmvae_workspace_bytes / mmvae_ws_offset.
 *   tune         MMVAE_TUNE_ENGINE below; the other entries are the implementation's experiment
 *                switches (0 = production behaviour).  The library reads no environment variables. */
#define MMVAE_N_EVENTS 8
#define MMVAE_N_TUNE 24
/* mmvae_exec.tune: 0 everywhere = production behaviour.  One entry is part of the interface: */
#define MMVAE_TUNE_ENGINE 17    /* the GEMM engine the caller is going to run (mmvae_hyper.gemm_bf16 & 0xFF; 0 = not stated).
                                   The split factors of the workspace layout are chosen for the workgroup shapes of that
                                   engine; any engine runs correctly on any layout */
/* Every other index is an experiment switch of the implementation (A/B timing, ablations, test hooks), listed in the
 * library's private header distributed-vae_amd/csrc/tune.h; callers leave them 0. */
typedef struct mmvae_exec {
    void *side_stream;
    void *ev[MMVAE_N_EVENTS];
    void *early_grad_event;
    int32_t early_recorded;
    int32_t split[6];
    int32_t tune[MMVAE_N_TUNE];
} mmvae_exec;

/* Where things are, in floats.  Filled by mmvae_param_layout. Tensor order t = 0..27:
 *  0 fc1.w[H,D] 1 fc1.b 2 fc2.w[H,H] 3 fc2.b 4 fc3.w 5 fc3.b 6 fc4.w 7 fc4.b 8 fc5.w[L,H] 9 fc5.b
 * 10 fcc.w[C,L] 11 fcc.b 12 fc_mu.w[S,L+C] 13 fc_sigma.w[S,L+C] 14 fc_mu.b 15 fc_sigma.b
 * 16 fc6.w[L,C+S] 17 fc6.b 18 fc7.w[H,L] 19 fc7.b 20 fc8.w 21 fc8.b 22 fc9.w 23 fc9.b
 * 24 fc10.w 25 fc10.b 26 fc11.w[D,H] 27 fc11.b */
typedef struct mmvae_param_layout_t {
    int64_t per_arm;                          /* floats per arm (padded)            */
    int64_t offset[MMVAE_N_PARAM_TENSORS];    /* within one arm's segment           */
    int64_t rows[MMVAE_N_PARAM_TENSORS];      /* out features (or length for bias)  */
    int64_t cols[MMVAE_N_PARAM_TENSORS];      /* in features (1 for bias)           */
    /* BatchNorm running buffers: one flat fp32 buffer, arm-major; per arm
     * [mean_i, var_i] for batch_l1..batch_l5, batch_s (nn_model.py:208-255) */
    int64_t bn_per_arm;
    int64_t bn_mean_offset[MMVAE_N_BN];
    int64_t bn_var_offset[MMVAE_N_BN];
    int64_t bn_dim[MMVAE_N_BN];
} mmvae_param_layout_t;

/* Scalars written by mmvae_loss / mmvae_train_step into `loss_out` (device, fp32):
 *  [0] total  [1] loss_joint  [2] mean neg-joint-entropy  [3] mean simplex distance
 *  [4] mean l2 distance  then rec[A], kl[A], ll[A]   (the 9-tuple of nn_model.py:588-598). */
#define MMVAE_LOSS_TOTAL 0
#define MMVAE_LOSS_JOINT 1
#define MMVAE_LOSS_CENT 2
#define MMVAE_LOSS_CDIST 3
#define MMVAE_LOSS_CL2 4
#define MMVAE_LOSS_REC0 5
#define MMVAE_LOSS_FLOATS(A) (5 + 3 * (A))

/* Named workspace regions, for callers that return forward outputs as views and for tests that
 * localise a failing kernel.  All per-arm arrays are [A, B, width]. */
typedef enum mmvae_ws_id {
    MMVAE_WS_X_LOW = 0, /* [A,B,L]  BN5 output                (forward out 3) */
    MMVAE_WS_C_PROB,    /* [A,B,C]  softmax(fcc)              (forward out 9) */
    MMVAE_WS_C,         /* [A,B,C]  softmax(c_prob/tau)       (forward out 4) */
    MMVAE_WS_C_SMP,     /* [A,B,C]  Gumbel-softmax sample     (forward out 6) */
    MMVAE_WS_S_MEAN,    /* [A,B,S]                            (forward out 7) */
    MMVAE_WS_S_LOGVAR,  /* [A,B,S]                            (forward out 8) */
    MMVAE_WS_S_SMP,     /* [A,B,S]                            (forward out 5) */
    MMVAE_WS_Y_SOFT,    /* [A,B,C]  soft sample (== C_SMP unless hard) */
    MMVAE_WS_R1, MMVAE_WS_R2, MMVAE_WS_R3, MMVAE_WS_R4, /* [A,B,H] relu(fc_i), pre-BN */
    MMVAE_WS_R5,        /* [A,B,L] */
    MMVAE_WS_D6,        /* [A,B,L] */
    MMVAE_WS_D7, MMVAE_WS_D8, MMVAE_WS_D9, MMVAE_WS_D10, /* [A,B,H] */
    MMVAE_WS_ZIN,       /* [A,B,C+S] decoder input */
    MMVAE_WS_DZ11,      /* [A,B,D] d loss / d fc11 pre-activation */
    MMVAE_WS_DZ1,       /* [A,B,H] d loss / d fc1 pre-activation  */
    MMVAE_WS_GZIN,      /* [A,B,C+S] */
    MMVAE_WS_GZC,       /* [A,B,C]  d loss / d fcc output */
    MMVAE_WS_G5,        /* [A,B,L]  d loss / d x_low */
    MMVAE_WS_BN_MEAN1,  /* [A,H] batch mean of R1 (then BN_MEAN1+i for layer i+1) */
    MMVAE_WS_GD10_SLAB, /* [n_slab][A,B,H] gene-split partial sums of d loss / d d10 = dZ11 W11 (n_slab: mmvae_splits[4]) */
    MMVAE_WS_G1,        /* [A,B,H]  d loss / d BatchNorm1's output (then G1 + i for i < 4; G5 above)               */
    MMVAE_WS_G2, MMVAE_WS_G3, MMVAE_WS_G4,
    MMVAE_WS_DZ2,       /* [A,B,H]  d loss / d fc2's pre-activation (then DZ2 + i; DZ5 is [A,B,L])                 */
    MMVAE_WS_DZ3, MMVAE_WS_DZ4, MMVAE_WS_DZ5,
    MMVAE_WS_COUNT_
} mmvae_ws_id;

/* ---- queries (host only, no GPU needed) ------------------------------------------------- */
int mmvae_abi_version(void);
const char *mmvae_last_error_string(void);
/* 0 if the kernels support these dims (H,C<=128, L<=64, S<=32, L+C,C+S<=255, A<=MMVAE_MAX_ARMS).  Calls with
 * h->training != 0 additionally need 2 <= B <= 32768 (batch statistics; capacity of the exact batch-sum accumulators); eval
 * mode takes any batch from one cell up. */
int mmvae_check_dims(const mmvae_dims *d);
int mmvae_param_layout(const mmvae_dims *d, mmvae_param_layout_t *out);
/* bytes of caller-provided workspace that forward/loss/backward/train_step need */
size_t mmvae_workspace_bytes(const mmvae_dims *d, const mmvae_exec *ex);
/* offset (in floats) of a named region inside the workspace, or -1 */
int64_t mmvae_ws_offset(const mmvae_dims *d, const mmvae_exec *ex, int ws_id);
/* the split factors the layout uses for these dims and context, in the order of mmvae_exec.split (0 fc1 split-K, 1 fc11 column
 * splits, 2 dW1 batch splits, 3 small-layer dW batch splits, 4 d(d10) gene splits = slabs of MMVAE_WS_GD10_SLAB,
 * 5 dW11 batch splits) */
int mmvae_splits(const mmvae_dims *d, const mmvae_exec *ex, int32_t out[6]);
/* offset (in floats) of a 1024-float block inside the workspace that only diagnostic builds write
 * (in-kernel cycle stamps, enabled by environment switches; never read by any kernel) */
int64_t mmvae_ws_debug_offset(const mmvae_dims *d, const mmvae_exec *ex);

/* ---- compute (device pointers, asynchronous on stream) ----------------------------------- */

/* x: [B,D] shared by all arms when x_arm_stride == 0 (cpl_mixvae.py:425 x.expand(A,-1,-1)),
 * else arm a reads x + a*x_arm_stride (floats).
 * bn_running: flat running mean/var (updated in place when training); num_batches_tracked:
 * int64 [A*6] incremented when training (may be NULL).
 * x_rec: optional [A,B,D] output (forward out 0); NULL = do not materialise.
 * need_grad != 0 additionally stores what backward needs (dZ11, fc10-grad slabs).
 * All other forward outputs stay in `ws` (see mmvae_ws_offset). */
int mmvae_forward(const mmvae_dims *d, const mmvae_hyper *h, const mmvae_noise *nz,
                  const float *params, float *bn_running, int64_t *num_batches_tracked,
                  const float *x, int64_t x_arm_stride, float *x_rec, int need_grad,
                  void *ws, size_t ws_bytes, mmvae_exec *ex, void *stream);

/* Finishes the loss scalars from what forward left in ws.  Must follow mmvae_forward on the same
 * ws/stream. */
int mmvae_loss(const mmvae_dims *d, const mmvae_hyper *h, void *ws, size_t ws_bytes,
               float *loss_out, mmvae_exec *ex, void *stream);

/* Gradient of loss_out[0] * grad_scale w.r.t. every parameter into `grads` (flat, same layout
 * as params; fully overwritten).  Needs forward(need_grad=1) + loss on the same ws. */
int mmvae_backward(const mmvae_dims *d, const mmvae_hyper *h, const mmvae_noise *nz,
                   const float *params, const float *x, int64_t x_arm_stride, float grad_scale,
                   void *ws, size_t ws_bytes, float *grads, mmvae_exec *ex, void *stream);

/* torch.optim.Adam / AdamW semantics on a flat buffer of n floats. step >= 1. */
int mmvae_adam_step(int64_t n, float *params, const float *grads, float *exp_avg,
                    float *exp_avg_sq, int64_t step, float lr, float beta1, float beta2,
                    float adam_eps, float weight_decay, int decoupled, void *stream);

/* forward + loss + backward (+ Adam when do_adam) for one batch: cpl_mixvae.py:434-463.
 * With do_adam == 0 the caller all-reduces `grads` (data parallel) and then calls
 * mmvae_adam_step itself. */
int mmvae_train_step(const mmvae_dims *d, const mmvae_hyper *h, const mmvae_noise *nz,
                     float *params, float *bn_running, int64_t *num_batches_tracked,
                     const float *x, int64_t x_arm_stride, void *ws, size_t ws_bytes,
                     float *grads, float *loss_out, int do_adam, float *exp_avg,
                     float *exp_avg_sq, int64_t step, float lr, float beta1, float beta2,
                     float adam_eps, float weight_decay, int decoupled, mmvae_exec *ex, void *stream);

/* mmvae_train_step on a batch that is never materialised: cell b of the batch is row rows[b] of the resident cells x genes
 * matrix `data` ([n_rows, ld] fp32, ld >= D; rows: int64 [B] on the device, indices outside [0, n_rows) are clamped as
 * mmvae_gather_rows does), shared by all arms (x.expand).  Replaces the batch assembly of the reference's DataLoader
 * (mmidas/utils/dataloader.py:114-132: shuffled index batches collated into a fresh tensor, pinned, copied to the device) AND
 * the per-step row gather of mmvae_gather_rows: fc1, the fused fc11 kernel and dW1 read x through a row map (B 32-bit
 * offsets the step's head launch derives from `rows`), so a shuffled batch costs what a resident one does.  Bit-identical to
 * mmvae_gather_rows + mmvae_train_step.  Offered where it is built -- the fused training step of the fp32x3 and bf16 engines
 * (gemm_bf16 & 0xFF == 2 with fc_dim <= 111, or == 1; x_drop > 0), ld % 4 == 0, 16-byte aligned data, n_rows * ld < 2^30 floats --; otherwise
 * MMVAE_E_UNSUPPORTED: gather the batch and call mmvae_train_step.
 *
 * data_bf16 (optional, NULL = none; the bf16 engine only, BASELINE.json configs[2]): a bf16 copy of `data` with the same
 * shape and leading dimension (in elements), made once per data set by mmvae_to_bf16.  The engine rounds x to bf16 on its way
 * to the matrix pipe anyway; with the copy fc1, dW1 and the fused fc11 kernel read 2 bytes per cell and gene instead of 4, and
 * dZ11 travels from the fused kernel to the dW11 GEMM as bf16 (the values that GEMM takes in any case).  Same results as the
 * step on an fp32 matrix that holds the rounded values -- bit for bit --; against the unrounded matrix only the
 * reconstruction loss changes: it compares with the rounded x (within the configuration's 5e-2 gate).
 * Needs D % 8 == 0, ld % 8 == 0, a 16-byte aligned copy; MMVAE_E_UNSUPPORTED otherwise or with another engine. */
int mmvae_train_step_rows(const mmvae_dims *d, const mmvae_hyper *h, const mmvae_noise *nz, float *params,
                          float *bn_running, int64_t *num_batches_tracked, const float *data,
                          const uint16_t *data_bf16, int64_t ld,
                          int64_t n_rows, const int64_t *rows, void *ws, size_t ws_bytes, float *grads,
                          float *loss_out, int do_adam, float *exp_avg, float *exp_avg_sq, int64_t step, float lr,
                          float beta1, float beta2, float adam_eps, float weight_decay, int decoupled,
                          mmvae_exec *ex, void *stream);

/* ---- evaluation labels and between-arm consensus ----------------
 * Replaces the per-epoch host loop of mmidas/cpl_mixvae.py:563-657: eval-mode forward of every batch,
 * `classify` = argmax of c (mmidas/_utils.py:79-80), `compute_confmat` per arm pair (:84-95),
 * `confmat_normalize` (:98-100), `confmat_mean` (:128-129).
 *
 * mmvae_eval_classify: encoder + latent block of forward(eval=True) with the BatchNorm running statistics
 *   (h->training == 0, h->eval_flag == 1; no decoder, no fc11: the labels need c only), then
 *   labels[a*B + b] = argmax_k c[a][b][k] (first maximum on ties, as np.argmax).  labels: int32 [A,B].
 *   counts != NULL: additionally counts[pair][labels[a][b]][labels[a'][b]] += 1 for every arm pair a < a'
 *   in the order (0,1), (0,2) ... (A-2,A-1); counts: int64 [A(A-1)/2, C, C], zeroed by the caller before
 *   the first batch of an epoch.
 * mmvae_classify: the argmax alone on any [n_cells, C] fp32 matrix.
 * mmvae_confmat_accumulate: the counting alone, labels int32 [A, n].
 * mmvae_consensus: cm_norm[pair][i][j] = counts[pair][i][j] / max(rowsum_j, colsum_j) (0 where that is 0),
 *   consensus[pair] = mean_k cm_norm[pair][k][k]; fp64, numpy's summation order, C <= 128.
 *   cm_norm (double [npairs, C, C]) may be NULL; consensus: double [npairs]. */
int mmvae_eval_classify(const mmvae_dims *d, const mmvae_hyper *h, const float *params,
                        const float *bn_running, const float *x, int64_t x_arm_stride, void *ws,
                        size_t ws_bytes, int32_t *labels, int64_t *counts, mmvae_exec *ex, void *stream);
int mmvae_classify(const float *c_probs, int64_t n_cells, int C, int32_t *labels, void *stream);
int mmvae_confmat_accumulate(const int32_t *labels, int A, int64_t n, int C, int64_t *counts,
                             void *stream);
int mmvae_consensus(const int64_t *counts, int npairs, int C, double *cm_norm, double *consensus,
                    void *stream);

/* ---- augmenter forward in the training loop ----------------------
 * Replaces `self.netA(x.expand(A,-1,-1), True, 0.1)[1]` (mmidas/cpl_mixvae.py:422-423; netA.eval(), :184), i.e.
 * Augmenter_smartseq.forward in eval mode (mmidas/augmentation/udagan.py:281-329, reparam_trick
 * mmidas/augmentation/aug_utils.py:51-65): Linear + BatchNorm1d(running statistics) + ReLU stacks around a
 * noise-conditioned Gaussian bottleneck.  D input_dim, N1 = D / 5, N3 = n_dim, N5 = n_dim / 5, Z latent_dim,
 * NZ noise_dim (udagan.py:218-279); A arms x B cells per call. */
typedef struct mmvae_aug_dims {
    int32_t A, B, D, N1, N3, N5, Z, NZ;
} mmvae_aug_dims;

/* Device pointers to the module's tensors, PyTorch layouts ([out, in] weights, contiguous). */
typedef struct mmvae_aug_tensors {
    const float *w[11], *b[11];            /* fc1 .. fc11 (fc5.weight is [N5, N3 + NZ])                   */
    const float *bn_mean[10], *bn_var[10]; /* batch_fc1 .. batch_fc10 running_mean / running_var          */
    const float *w_mu, *b_mu, *w_sigma, *b_sigma, *bn_mu_mean, *bn_mu_var; /* fc_mu, fc_sigma, batch_fc_mu */
    const float *noise_w;                  /* noise.weight [NZ, NZ] (no bias)                             */
    const float *bnz_weight, *bnz_bias, *bnz_mean, *bnz_var; /* bnz: affine BatchNorm1d, eps 1e-5        */
} mmvae_aug_tensors;

/* floats of the packed-weights buffer / bytes of workspace (shared_x: the arms share x) */
size_t mmvae_aug_packed_floats(const mmvae_aug_dims *d);
size_t mmvae_aug_workspace_bytes(const mmvae_aug_dims *d, int shared_x);
/* Once per set of weights: rows padded to 16 bytes, BatchNorm folded into per-column (scale, shift). */
int mmvae_aug_pack(const mmvae_aug_dims *d, const mmvae_aug_tensors *t, float *packed, void *stream);
/* x: [B,D] shared by the arms (x_arm_stride == 0, as x.expand) or [A,B,D] contiguous (x_arm_stride == B*D).
 * z0: [A,B,NZ] and eps: [A,B,Z] standard-normal draws (the reference's torch.randn / randn_like); scale: the
 * noise scale (0.1 in the trainer).  Outputs: s_out [A,B,Z] (forward out 0), x_aug [A,B,D] (forward out 1),
 * ready as the per-arm input of mmvae_train_step (x_arm_stride = B*D).  gemm_bf16 (0 / 1 / 2 as mmvae_hyper.gemm_bf16): != 0: the ten large Linear layers
 * take bf16 operands with fp32 accumulation (BASELINE.json's bf16 configuration); the latent block, the folded
 * BatchNorm / ReLU epilogues and all stored activations stay fp32. */
int mmvae_augment(const mmvae_aug_dims *d, const float *packed, const float *x, int64_t x_arm_stride,
                  const float *z0, const float *eps, float scale, void *ws, size_t ws_bytes, float *s_out,
                  float *x_aug, int gemm_bf16, const mmvae_exec *ex, void *stream);

/* The same forward on a batch that is ROWS OF A RESIDENT MATRIX, never assembled (the augmented counterpart of
 * mmvae_train_step_rows; the reference gathers the batch in its DataLoader, mmidas/utils/dataloader.py:114-132, and hands it to
 * netA, cpl_mixvae.py:418-423): the matrix is kept as the GEMM engine's tiled bf16 slice planes -- made ONCE per data set by
 * mmvae_tp_planes (n_planes 3: the three exact slices of the fp32x3 engine, gemm_bf16 = 2; 1: the matrix rounded to bf16,
 * gemm_bf16 = 1; mmvae_tp_planes_bytes of ZERO-FILLED device memory, 0 for unsupported arguments: K % 4 == 0, a plane --
 * n_rows rounded up to 256 x K rounded up to 16 x 2 bytes -- below 4 GB) -- and the first
 * layer's loads take the batch's rows out of it through `rows` (int64 [B] on the device, clamped to the matrix).  The arms
 * share x (as x.expand).  Same results, bit for bit, as mmvae_augment on the gathered batch.  MMVAE_E_UNSUPPORTED for
 * gemm_bf16 = 0 (the fp32 matrix-instruction engine has no planes). */
size_t mmvae_tp_planes_bytes(int64_t n_rows, int32_t K, int32_t n_planes);
int mmvae_tp_planes(const float *src, int64_t ld, int64_t n_rows, int32_t K, int32_t n_planes, uint16_t *dst,
                    void *stream);
int mmvae_augment_rows(const mmvae_aug_dims *d, const float *packed, const uint16_t *x_planes, int64_t n_rows,
                       int32_t n_planes, const int64_t *rows, const float *z0, const float *eps, float scale,
                       void *ws, size_t ws_bytes, float *s_out, float *x_aug, int gemm_bf16,
                       const mmvae_exec *ex, void *stream);

/* ---- device-resident data path ------------------------------------
 * out[i, :] = data[idx[i], :], i < n: the batch assembly of the reference's DataLoader
 * (mmidas/utils/dataloader.py:114-132: shuffled index batches collated from a host TensorDataset, pinned, copied to
 * the device) as one row gather in HBM.  data: [n_rows, ld] fp32 with ld >= D; idx: int64 [n] on the device
 * (out-of-range indices are clamped; the host loader validates them); out: [n, D] contiguous. */
int mmvae_gather_rows(const float *data, int64_t ld, int64_t n_rows, const int64_t *idx, int64_t n, int32_t D,
                      float *out, void *stream);
/* dst[r, c] = bf16(src[r, c]) (round to nearest even), r < n_rows, whole rows of ld elements: the bf16 copy of a resident
 * matrix for mmvae_train_step_rows(data_bf16).  src: [n_rows, ld] fp32, ld % 4 == 0, 16-byte aligned; dst: [n_rows, ld]
 * 2-byte elements. */
int mmvae_to_bf16(const float *src, int64_t ld, int64_t n_rows, int32_t D, uint16_t *dst, void *stream);
/* ---- data-parallel gradient exchange (SURVEY.md sections 8b / 8e) ------------------------------------------------------
 * ONE RCCL all-reduce (average) of the flat fp32 gradient buffer per step, issued by the library on the stream the step
 * runs on (stream-ordered behind mmvae_train_step(do_adam = 0), in front of mmvae_adam_step; no host synchronisation, no
 * second stream).  Replaces the reference's FSDP gradient traffic (train.py:140-143; the bring-up of
 * mmidas/_dist_utils.py:12-55 for this collective).  RCCL is resolved at run time (librccl.so.1): without it these entry
 * points return MMVAE_E_UNSUPPORTED and everything else works.  One process per GPU:
 *   rank 0:      mmvae_dp_unique_id(id)         and hands the 128 bytes to every rank (any channel: a file, a store, MPI)
 *   every rank:  mmvae_dp_init(id, rank, world_size, &comm)      on its device (collective: returns when all have called)
 *   per step:    mmvae_allreduce_grads(comm, grads, n, stream)   in place; every rank with the same n
 *   at the end:  mmvae_dp_destroy(comm)
 * The communicator is caller-owned; the library keeps nothing but the resolved RCCL entry points. */
#define MMVAE_DP_ID_BYTES 128
int mmvae_dp_unique_id(uint8_t id[MMVAE_DP_ID_BYTES]);
int mmvae_dp_init(const uint8_t id[MMVAE_DP_ID_BYTES], int rank, int world_size, void **comm);
int mmvae_allreduce_grads(void *comm, float *grads, int64_t n, void *stream);
int mmvae_dp_destroy(void *comm);

/* Writes the noise the Philox mode (nz->mode == 1) would use, in explicit-buffer form, so a test
 * can replay a Philox step through mode 0.  Any output pointer may be NULL. */
int mmvae_dump_noise(const mmvae_dims *d, const mmvae_hyper *h, const mmvae_noise *nz,
                     uint8_t *x_mask, float *u_gumbel, float *u_state, uint8_t *s_mask,
                     void *stream);

/* Launches ONE stage of the step on an already prepared workspace (mmvae_forward(need_grad=1) +
 * mmvae_loss [+ mmvae_backward] must have run on it): for per-kernel timing with HIP events and for
 * rocprof runs.  Stages: 0 fc1 forward (split-K GEMM + epilogue), 1 fused fc11 (x_rec GEMM + loss +
 * dZ11 + d(d10) GEMM), 2 dW1 and dW11 GEMMs, 3 batched small-layer dW GEMM, 4 decoder chain forward,
 * 5 decoder chain backward, 6 latent forward, 7 latent backward, 8 gradient slab reduction (grads),
 * 9 dropout keep-mask bit image; single kernels of the fast path: 10 fc11 x_rec/loss/dZ11, 11 d(d10) GEMM,
 * 12 dW1 GEMM, 13 [dW11|db11] GEMM, 14 fc1 GEMM without its epilogue. */
int mmvae_debug_stage(const mmvae_dims *d, const mmvae_hyper *h, const mmvae_noise *nz, int stage,
                      const float *params, const float *x, int64_t x_arm_stride, void *ws,
                      size_t ws_bytes, float *grads, mmvae_exec *ex, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MMVAE_H */
