/*
 * qbm_hip.h -- C ABI of libqbm_hip.so: the MI355X (gfx950) Gibbs-sampling / contrastive-divergence
 * hot path of Mark-Seebode/QBM-Image-Classification behind plain pointers and sizes.
 *
 * The reference has no FFI: its boundaries are Python duck types.  Each entry point
 * below names the reference interface it sits under (file:line relative to the reference root); the
 * Python mirror of those interfaces lives in qbm_image_classification_amd/ and binds these symbols
 * with ctypes (INTEGRATION.md shows the stub a maintainer of the reference would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (e.g. torch tensor .data_ptr());
 *     nothing is allocated or freed inside the library.  Process-wide state: ONE switch, qbm_rbm_fused_chain_mode (which
 *     launch sequence a CD-k chain takes; both give the same bits), and a per-device cache of one device attribute
 *     (the CU count that switch consults).  Everything else lives in the caller's buffers;
 *   - all matrices are dense row-major float32 unless stated otherwise;
 *   - `stream` is a hipStream_t (NULL = the default stream); every call only enqueues work on it,
 *     never synchronises, and is legal inside hipStreamBeginCapture/EndCapture;
 *   - return value: QBM_OK (0) or a negative QBM_ERR_* code; qbm_last_error() returns the message of
 *     the calling thread's last failure;
 *   - RNG: Philox4x32-10, key = seed, counter = (col>>2, row0+row, offset lo, offset hi); the caller
 *     advances `offset` per draw.  `row0` is the global index of the first local row, so results
 *     do not depend on how a minibatch is sharded over GPUs.
 *   - numerics: DESIGN.md "Numerical contract" (K-quartered fp32 FMA chains on the MFMA pipe).
 */
#ifndef QBM_HIP_H
#define QBM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QBM_ABI_VERSION 15
#define QBM_MAX_CLASSES 15 /* class rows + the bias row share one 16-row gradient tile */
#define QBM_MAX_QUBO_VARS 1024

/* element type of a Gibbs-state tensor: fp32 (what the reference holds) or uint8 {0,1} (exact for
 * binary states and 4x less L2 -> LDS staging traffic, the resource that bounds these kernels) */
#define QBM_F32 0
#define QBM_U8 1
#define QBM_BITS 2 /* bit-packed {0,1} rows, MSB first (numpy.packbits), ceil(V/8) bytes per row: dataset storage only */

#define QBM_OK 0
#define QBM_ERR_ARG (-1) /* bad dimension / null pointer */
#define QBM_ERR_HIP (-2) /* a HIP runtime call or kernel launch failed */

#if defined(__GNUC__)
#define QBM_API __attribute__((visibility("default")))
#else
#define QBM_API
#endif

/* ClassificationRBM state, src/ClassificationRBM.py:26-30:
 * weights[V,H], visible_bias[V], hidden_bias[H], class_weights[C,H], class_bias[C]. */
typedef struct {
    float* W;
    float* b;
    float* d;
    float* U;
    float* c;
    int V, H, C;
} qbm_rbm_params;

/* gradient blocks (the reference's *_grad attributes, src/ClassificationRBM.py:130-138); any pointer
 * may be NULL = "not wanted" (update then treats it as zero, like visible_bias_grad = 0 at :138). */
typedef struct {
    float* gW;
    float* gb;
    float* gd;
    float* gU;
    float* gc;
    float* n_rows; /* optional, 1 float: the gradient launch of qbm_rbm_cd_step / qbm_rbm_disc_step / qbm_rbm_two_phase_grad
                    * stores the number of rows it summed over (0 for an empty shard; -inf when the one-launch chain in
                    * front of it gave up, see qbm_rbm_cd_workspace.sync).  The data-parallel path keeps it right behind the
                    * gradient blocks, so ONE all-reduce yields the global batch size on the device
                    * (qbm_rbm_update_dev) -- no host round trip per step. */
} qbm_rbm_grads;

QBM_API int qbm_abi_version(void);
QBM_API const char* qbm_last_error(void);

/* ---- RBM conditionals --------------------------------------------------------------------- */

/* ClassificationRBM.sample_hidden, src/ClassificationRBM.py:43-47: p_h = sigma(v.W + d + y.U).
 * v[B,V], y1hot[B,C] (NULL = no class term); outputs p_h[B,H] (probabilities, what the reference
 * returns) and/or h_sample[B,H] = (p_h > u) in {0,1} (either may be NULL, not both). */
QBM_API int qbm_rbm_sample_hidden(const qbm_rbm_params* P, const float* v, const float* y1hot, int B,
                                  float* p_h, float* h_sample, uint64_t seed, uint64_t offset, uint32_t row0,
                                  void* stream);

/* The same with a typed input and typed sample outputs: v is fp32 or uint8 (v_dtype), the sample can be
 * written as fp32 (h_f32) and/or uint8 (h_u8). */
QBM_API int qbm_rbm_sample_hidden_ex(const qbm_rbm_params* P, const void* v, int v_dtype, const float* y1hot, int B,
                                     float* p_h, float* h_f32, uint8_t* h_u8, uint64_t seed, uint64_t offset,
                                     uint32_t row0, void* stream);

/* x.W + d, the `precomputed_factor` of src/ClassificationRBM.py:65 and :106.  pre[B,H]. */
QBM_API int qbm_rbm_hidden_preact(const qbm_rbm_params* P, const float* x, int B, float* pre, void* stream);

/* ClassificationRBM.sample_visible, src/ClassificationRBM.py:49-52: p_v = sigma(h.W^T + b). */
QBM_API int qbm_rbm_sample_visible(const qbm_rbm_params* P, const float* h, int B, float* p_v, float* v_sample,
                                   uint64_t seed, uint64_t offset, uint32_t row0, void* stream);

/* sample_visible and sample_class of the same hidden state in ONE launch (the class conditional rides
 * along as extra workgroups): what a Gibbs step h -> (v, y) needs.  h is fp32 or uint8;
 * p_y / y1hot both NULL = visible units only. */
QBM_API int qbm_rbm_sample_visible_class_ex(const qbm_rbm_params* P, const void* h, int h_dtype, int B, float* p_v,
                                            float* v_f32, uint8_t* v_u8, float* p_y, float* y1hot, uint64_t seed,
                                            uint64_t offset_v, uint64_t offset_y, uint32_t row0, void* stream);

/* ClassificationRBM.sample_class, src/ClassificationRBM.py:54-60: exp(h.U^T + c), L1-normalised.
 * p_y[B,C] and/or a categorical one-hot draw y1hot[B,C]. */
QBM_API int qbm_rbm_sample_class(const qbm_rbm_params* P, const float* h, int B, float* p_y, float* y1hot,
                                 uint64_t seed, uint64_t offset, uint32_t row0, void* stream);

/* ClassificationRBM.sample_class_given_x, src/ClassificationRBM.py:62-86 (free-energy evaluation).
 * pre_ws[B,H] workspace (receives x.W + d); neg_free_energy[B,C] = -F(x,y) (may be NULL);
 * p_y_given_x[B,C] (may be NULL). */
QBM_API int qbm_rbm_free_energy(const qbm_rbm_params* P, const float* x, int B, float* pre_ws,
                                float* neg_free_energy, float* p_y_given_x, void* stream);

/* ---- gradients and update ----------------------------------------------------------------- */

/* Two-phase batch gradient (CD-k: SURVEY.md 8.A-7; discriminative: src/ClassificationRBM.py:118-138):
 *   gW = A0^T.P0 - A1^T.P1,  gU = Y0^T.P0 - Y1^T.P1,  gd = colsum(P0) - colsum(P1),
 *   gb = colsum(A0) - colsum(A1),  gc = colsum(Y0) - colsum(Y1)
 * A*[B,V] fp32 or uint8 (a*_dtype), Y*[B,C] (NULL = zeros), P*[B,H].  Outputs as selected by the
 * non-NULL fields of G.
 * W_fused non-NULL: additionally W += factor*gW*lr/batch_size in the same pass
 * (src/ClassificationRBM.py:90); gW itself may then be NULL.  aux_fused non-NULL: its b, d, U, c blocks
 * are updated in the same launch as well (:91-99, including the sparse_constant subtraction). */
QBM_API int qbm_rbm_two_phase_grad(int V, int H, int C, int B, const void* A0, int a0_dtype, const float* Y0,
                                   const float* P0, const void* A1, int a1_dtype, const float* Y1, const float* P1,
                                   const qbm_rbm_grads* G, float* W_fused, const qbm_rbm_params* aux_fused, float lr,
                                   float factor, float batch_size, float sparse_constant, void* stream);

/* ClassificationRBM.update_weights, src/ClassificationRBM.py:88-99:
 * theta += factor*grad*lr/batch_size; b, d, c -= sparse_constant.  G->gW == NULL skips W (already
 * updated by the fused gradient pass). */
QBM_API int qbm_rbm_update(const qbm_rbm_params* P, const qbm_rbm_grads* G, float lr, float factor,
                           float batch_size, float sparse_constant, void* stream);

/* The same with the batch size read from DEVICE memory when batch_size_dev is non-NULL (1 float: the all-reduced
 * qbm_rbm_grads.n_rows of the data-parallel step; `batch_size` is then ignored).  A value that is not a
 * positive finite number makes the launch a no-op: some rank's one-launch chain gave up (its n_rows is -inf), so every
 * rank skips this update alike.  chain_status non-NULL: the status word of a fused chain launch
 * (qbm_rbm_cd_workspace.sync + 512); non-zero = no-op. */
QBM_API int qbm_rbm_update_dev(const qbm_rbm_params* P, const qbm_rbm_grads* G, float lr, float factor, float batch_size,
                               const float* batch_size_dev, float sparse_constant, const uint32_t* chain_status,
                               void* stream);

/* ---- composites: launch sequences on `stream` (graph-capturable) ---------------------------- */

/* fp32 -> uint8 conversion of a {0,1} tensor (n elements), for callers that hold binary data as floats */
QBM_API int qbm_f32_to_u8(const float* x, uint8_t* y, long n, void* stream);

/* caller-owned scratch of one CD-k / PCD step; the Gibbs states are uint8 {0,1} */
typedef struct {
    float* ph0;   /* [B,H] positive-phase hidden probabilities */
    uint8_t* h;   /* [B,H] hidden sample h_0 .. h_{k-1} (h_k is never drawn: the gradient uses p(h_k)) */
    uint8_t* vk;  /* [B,V] */
    float* yk;    /* [B,C] one-hot */
    float* phk;   /* [B,H] */
    uint32_t* sync; /* [QBM_CD_SYNC_WORDS] uint32, zeroed once by the caller and then left alone, or NULL: state of the
                     * launches that synchronise workgroups of one XCD among themselves -- the one-launch chain (k_chain) and
                     * the fused tail of a step (k_hgrad: last hidden phase + gradient + update): round flags of their
                     * XCD-local barriers, the XCC ids, per-workgroup completion marks, and the status word at index 512:
                     * 0 = fine; 1 / 2 = the one-launch chain gave up (a barrier timed out / workgroups of a row block on
                     * different XCDs); 3 / 4 = the fused tail gave up (barrier / placement).  Words 513 / 514: `offset`
                     * (low / high half) of the step that gave up; word 515: the round of a fused tail that gave up (516: round counter).
                     * FAIL-SAFE: while the status word is set, EVERY later launch of qbm_rbm_cd_step on this buffer stores
                     * nothing (n_rows becomes -inf) -- also the steps whose SHAPE the fused launches do not take (the short
                     * last minibatch of an epoch): wherever the mode would hand a step with uint8 data to a fused launch,
                     * an ineligible shape runs one launch per phase with every storing launch reading the word -- and so
                     * do qbm_rbm_update_dev and qbm_rbm_disc_step_ex (qbm_rbm_disc_workspace.status) when handed the word.  Status 1 / 2:
                     * the step's gradient launch did nothing either -- parameters and workspace are as before the step,
                     * which the caller repeats with qbm_rbm_fused_chain_mode(0) after re-zeroing the buffer.  Status 3 / 4
                     * with fuse_update: the column groups that got through their barrier HAVE been updated, the others
                     * not -- the caller calls qbm_rbm_cd_step_finish with the step's arguments and word 515 BEFORE re-zeroing
                     * (it completes exactly what is missing: same bits as an undisturbed step), then repeats the LATER
                     * steps; without fuse_update the parameters are untouched (qbm_rbm_update_dev is handed the status
                     * word or the -inf row count) and the step is repeated like a failed chain.
                     * rbm_ops.HipRbmOps.check_chain_status does all of this.  NULL = always one launch per phase.  One
                     * buffer per stream of concurrent work; workspaces of different shapes on one stream may share it. */
    /* The training loop's monitor, qbm_rbm_cd_step only (all five set, or mon_labels NULL = off): the step also evaluates
     * p(y|x0) of its minibatch under the parameters BEFORE its update -- what discriminative_training reports
     * (src/ClassificationRBM.py:108, :142) -- from v0.W + d, which the positive-phase launch stores on the way (no further
     * contraction; the rows' free energies ride as extra workgroups in the sample_visible launch for B <= 512, a launch of
     * their own otherwise and behind the one-launch chain), and books the minibatch means of the two loss terms into
     * mon_loss_acc (qbm_rbm_disc_workspace.rowloss / loss_acc: same definitions) inside the gradient launch. */
    const int32_t* mon_labels; /* [B] */
    float* mon_pre;            /* [B,H] */
    float* mon_p;              /* [B,C] p(y|x0) */
    float* mon_rowloss;        /* [B,2] */
    double* mon_loss_acc;      /* [2] */
    /* Optional device-side draw counters, [2] uint64: every sampling launch of qbm_rbm_cd_chain / qbm_rbm_cd_step adds word 0 to
     * its draw ids (`offset` ...) and word 1 (mod 2^32) to `row0`.  A training loop that captures an epoch's steps in a
     * hipGraph passes offsets RELATIVE to the epoch in the arguments and moves these two words between replays: the captured
     * launches keep their arguments, the draws advance exactly as with host-side counters.  With the words set the step takes
     * one launch per phase (the one-launch kernels record a step's draw offset for their fail-safe from the arguments). */
    const uint64_t* draw_base;
} qbm_rbm_cd_workspace;
#define QBM_CD_SYNC_WORDS 3136

/* k-step block-Gibbs chain (declared, never implemented in the reference:
 * src/ClassificationRBM.py:19, :179-182).  v0[B,V] fp32 or uint8 (v0_dtype), y0[B,C] fp32.
 * Draw ids: h_0 -> offset; t = 1..k: v_t -> offset+3t-2, y_t -> offset+3t-1, h_t -> offset+3t.
 * chain_v (uint8 [B,V]) / chain_y (fp32 [B,C]) non-NULL = persistent chains (PCD): the negative chain
 * starts from them and the k-th Gibbs iteration writes (v_k, y_k) straight into them; ws->vk / ws->yk then hold
 * the states of iteration k-1 (untouched for k = 1).
 * With ws->sync set, uint8 data and B a multiple of 32 up to 256 (H <= 512, V <= 1024, C <= 12, 16-byte aligned
 * operands) the whole chain is ONE launch (row blocks of 32 pinned to one XCD each, XCD-local barriers between the
 * phases); same draws, same bits. */
QBM_API int qbm_rbm_cd_chain(const qbm_rbm_params* P, const void* v0, int v0_dtype, const float* y0, int B, int k,
                             uint64_t seed, uint64_t offset, uint32_t row0, uint8_t* chain_v, float* chain_y,
                             const qbm_rbm_cd_workspace* ws, void* stream);

/* When qbm_rbm_cd_chain / qbm_rbm_cd_step use the launches that synchronise workgroups inside an XCD: 0 = never (one launch
 * per phase, then the gradient), 1 = where they are faster (the default, or the first character of the environment variable
 * QBM_FUSED_CHAIN): the one-launch chain from k = 2 on; 2 = the one-launch chain whenever the shape allows; 3 = as 1, plus
 * the fused tail (last hidden phase + gradient + update in one launch, k_hgrad: CD-1 is then three launches) where the shape
 * allows -- an opt-in: measured slower than the launch-per-phase tail.  All give the same bits.
 * Process-wide; returns the previous mode (an out-of-range argument only queries). */
QBM_API int qbm_rbm_fused_chain_mode(int mode);

/* chain + gradient (+ update).  fuse_update == 0: the gradients go to G (all five blocks required; the
 * data-parallel path all-reduces them, then calls qbm_rbm_update).  fuse_update != 0: single-GPU path, every
 * parameter block is updated inside the gradient launch (the non-NULL blocks of G are still written).
 * B == 0 (a rank whose shard of a short last minibatch is empty): the non-NULL blocks of G are zeroed. */
QBM_API int qbm_rbm_cd_step(const qbm_rbm_params* P, const qbm_rbm_grads* G, const void* v0, int v0_dtype,
                            const float* y0, int B, int k, uint64_t seed, uint64_t offset, uint32_t row0,
                            uint8_t* chain_v, float* chain_y, const qbm_rbm_cd_workspace* ws, int fuse_update,
                            float lr, float factor, float batch_size, float sparse_constant, void* stream);

/* One data-parallel CD-k / PCD step as ONE call: [apply_pending != 0: the update of the step BEFORE, from the
 * all-reduced G with its row count G->n_rows read on the device -- qbm_rbm_update_dev --] then chain + gradient of this
 * minibatch into G (all five blocks and n_rows required; B == 0 zeroes them).  The caller all-reduces the flat buffer between
 * two calls and applies the last step's update with qbm_rbm_update_dev: per step one call and one collective on the host. */
QBM_API int qbm_rbm_dp_step(const qbm_rbm_params* P, const qbm_rbm_grads* G, const void* v0, int v0_dtype, const float* y0,
                            int B, int k, uint64_t seed, uint64_t offset, uint32_t row0, uint8_t* chain_v, float* chain_y,
                            const qbm_rbm_cd_workspace* ws, int apply_pending, float lr, float factor, float sparse_constant,
                            void* stream);

/* Completes a step whose fused tail gave up (qbm_rbm_cd_workspace.sync: status 3 / 4): the arguments of that
 * qbm_rbm_cd_step call and word 515 of the sync buffer as `round`; to be called before the buffer is re-zeroed and before
 * anything else touches the workspace.  Two launches: the hidden phases, then the gradient items, that the workgroups of
 * the failed launch did not complete. */
QBM_API int qbm_rbm_cd_step_finish(const qbm_rbm_params* P, const qbm_rbm_grads* G, const void* v0, int v0_dtype,
                                   const float* y0, int B, int k, uint64_t seed, uint64_t offset, uint32_t row0,
                                   uint8_t* chain_v, float* chain_y, const qbm_rbm_cd_workspace* ws, int fuse_update,
                                   float lr, float factor, float batch_size, float sparse_constant, uint32_t round,
                                   void* stream);

/* The fused tail of a CD step by itself (what qbm_rbm_cd_step launches last where the shape allows -- uint8 data, B a multiple
 * of 16 up to 256, H a multiple of 128 up to 512, at most 256 strips of 32 rows x 16 hidden units, qbm_rbm_fused_chain_mode 3 --
 * else QBM_ERR_ARG; mode 3 only): p(h_k) = sigma(vk.W + d + yk.U) -> ws->phk, then the two-phase gradient of (v0, y0, ws->ph0) against
 * (vk, yk, ws->phk) and, with fuse_update, the update: ONE launch (k_hgrad).  `round` > 0 numbers the launch on ws->sync:
 * larger than every round used on that buffer before (qbm_rbm_cd_step counts in word 516), or words 576 .. 1599 re-zeroed. */
QBM_API int qbm_rbm_cd_tail(const qbm_rbm_params* P, const qbm_rbm_grads* G, const void* v0, int v0_dtype, const float* y0,
                            int B, const uint8_t* vk, const float* yk, const qbm_rbm_cd_workspace* ws, int fuse_update,
                            float lr, float factor, float batch_size, float sparse_constant, uint32_t round, void* stream);

/* caller-owned scratch of one discriminative step */
typedef struct {
    float* pre;       /* [B,H] x.W + d */
    float* diff;      /* [B,H] pos - neg: o_{y_i} - sum_c o_c p(c|x_i), src/ClassificationRBM.py:113-115, :124-126, :134 */
    float* rowloss;   /* [B,2] per row: CrossEntropyLoss ON THE PROBABILITIES, log(sum_c exp(p_c)) - p_label (:142, the reference's
                       * quirk), and -log(p_label + 1e-8) (:188-190) */
    double* loss_acc; /* optional, [2]: += the minibatch means (fp32: quartered chain over the rows, / B) of the two columns of
                       * rowloss, i.e. the reference's `epoch_error += batch_error.item()` and `epoch_nll += nll.item()`
                       * (:187, :191) on the device -- the training loop reads them once per epoch */
    const uint32_t* status; /* optional: the status word of a CD workspace (qbm_rbm_cd_workspace.sync + 512) -- hybrid training
                             * interleaves discriminative and CD steps: while the word is set the step stores nothing (p_y,
                             * gradients, parameters, loss_acc; n_rows becomes -inf), like the CD steps around it */
} qbm_rbm_disc_workspace;

/* ClassificationRBM.discriminative_training, src/ClassificationRBM.py:101-146: p_y[B,C] = p(y|x) before the update (returned
 * third by the reference), neg_free_energy[B,C] optional.  Three launches: x.W + d; per row -F, p(y|x), pos - neg and the
 * loss terms; then ONE contraction over the batch gW = x^T.(pos - neg) (the reference forms x^T.pos and x^T.neg by
 * broadcasting and subtracts: the same sum up to fp32 rounding) with gd, gU, gc and the loss bookkeeping riding along.
 * fuse_update as above; apply_update == 0 leaves the parameters untouched (gradients only).
 * B == 0: the non-NULL blocks of G are zeroed, the parameters are left alone. */
QBM_API int qbm_rbm_disc_step(const qbm_rbm_params* P, const qbm_rbm_grads* G, const float* x,
                              const int32_t* labels, int B, float* p_y, float* neg_free_energy,
                              const qbm_rbm_disc_workspace* ws, int apply_update, int fuse_update, float lr,
                              float factor, float batch_size, float sparse_constant, void* stream);

/* The same with x[B,V] as fp32 or as uint8 {0,1} (x_dtype): binary inputs as bytes take the uint8 kernels (a quarter of the
 * operand bytes).  Same values as the fp32 input 0.0 / 1.0. */
QBM_API int qbm_rbm_disc_step_ex(const qbm_rbm_params* P, const qbm_rbm_grads* G, const void* x, int x_dtype,
                                 const int32_t* labels, int B, float* p_y, float* neg_free_energy,
                                 const qbm_rbm_disc_workspace* ws, int apply_update, int fuse_update, float lr,
                                 float factor, float batch_size, float sparse_constant, void* stream);

/* ---- device-resident dataset pipeline ----------------------------------
 * The outputs of src/data_loader.py (get_mnist :23-73, get_medmnist :115-169, preprocess_images :240-277) are moved
 * to the device once; torch.utils.data.DataLoader(shuffle=True) of rbm_main.py:70-74 is replaced by an on-device
 * permutation + one gather launch per minibatch. */

/* {0,1} uint8 states from greyscale: out = x > threshold */
QBM_API int qbm_data_binarise(const float* x, long n, float threshold, uint8_t* out, void* stream);

/* [N,V] {0,1} uint8 -> [N, ceil(V/8)] bytes, MSB first (numpy.packbits(axis=1)): 98 B per 28x28 image */
QBM_API int qbm_data_pack_bits(const uint8_t* x, int N, int V, uint8_t* bits, void* stream);

/* sort keys of the epoch permutation: keys[i] = Philox4x32-10(key = seed, counter = (i>>2, 0, epoch))[i&3] >> 1;
 * the permutation is their stable ascending argsort */
QBM_API int qbm_data_shuffle_keys(uint64_t seed, uint64_t epoch, int N, int32_t* keys, void* stream);

/* minibatch gather: dst[r,:] = set[idx[r],:], src_kind QBM_F32 / QBM_U8 / QBM_BITS -> dst_kind QBM_F32 / QBM_U8
 * (fp32 -> uint8 truncates: binary data only); labels (int32 [N], optional) -> labels_out[B] and/or one-hot
 * y1hot[B,C].  idx values must lie in [0, N). */
QBM_API int qbm_data_gather(const void* src, int src_kind, const int32_t* idx, int B, int V, void* dst, int dst_kind,
                            const int32_t* labels, int32_t* labels_out, float* y1hot, int C, void* stream);

/* ---- greedy k-center coreset --------------------------------------------
 * src/coreset.py:28-106 k_center_greedy with the Euclidean metric (euclidean_dist :6-13), whose embeddings are the
 * hidden means the path already produces.  embT is the embedding matrix TRANSPOSED, [D,N] fp32 on the device;
 * seeds_host (HOST array, n_seeds >= 1) are the centres already chosen (:64-67 the random first centre, or
 * `already_selected`); n_picks further centres are chosen (the first pick re-selects a seed, exactly as the
 * reference's +inf / argmax step does, :89, :94).  picks[n_picks] int32 and selected[N] uint8 (the final mask,
 * seeds included) are device outputs.  workspace: qbm_kcenter_workspace_bytes(N, n_seeds) bytes of device memory.
 * The one exception to the conventions above: seeds_host is read on the host and copied to the device inside the call
 * (hipMemcpyAsync), so this entry point is not legal inside a stream capture. */
QBM_API long qbm_kcenter_workspace_bytes(int N, int n_seeds);
QBM_API int qbm_kcenter_greedy(const float* embT, int N, int D, const int32_t* seeds_host, int n_seeds, int n_picks,
                               int32_t* picks, uint8_t* selected, void* workspace, void* stream);

/* The same selection in ONE launch (round 4): 256 persistent workgroups keep their rows of the embedding matrix in LDS and
 * exchange one 8-byte granule per pick, inside an XCD through its L2 and across XCDs write-through (csrc/coreset_kernels.hip).
 * emb is the matrix as stored, [N,D] fp32 ROW-major on the device (D a multiple of 4, 16-byte aligned); seeds_dev (n_seeds >= 1,
 * int32) is a DEVICE array here; workspace: qbm_kcenter_fused_workspace_bytes() bytes.  QBM_ERR_ARG where the shape does not
 * fit (N * D beyond what 256 LDS slices hold, N > 65536, fewer than 256 compute units): call qbm_kcenter_greedy then.  Word 9
 * of the workspace (uint32) is a status: non-zero after the launch = its workgroups were not all resident or an XCD held more
 * than 64 of them, the outputs are invalid and the caller repeats the selection with qbm_kcenter_greedy.  Same contract, same
 * picks as qbm_kcenter_greedy. */
QBM_API long qbm_kcenter_fused_workspace_bytes(void);
QBM_API int qbm_kcenter_greedy_fused(const float* emb, int N, int D, const int32_t* seeds_dev, int n_seeds, int n_picks,
                                     int32_t* picks, uint8_t* selected, void* workspace, void* stream);

/* ---- Disc_QBM: QUBO sampler ------------------------------------------------------------------ */

/* HIP replacement of the annealer behind Disc_QBM.get_samples / get_samples_batch
 * (src/model/faster_dqbm.py:571-684, :961-969; sample_sa :299-306 -> dwave-neal) and behind the
 * sample_Q protocol (src/qubo/sampler.py:26-33).  Energy E(s) = sum_i lin_i s_i + sum_{i<j} J_ij s_i s_j
 * (the QUBO s^T Q s with Q upper-triangular: lin = diag(Q), J = Q_offdiag + Q_offdiag^T), minimised.
 *   lin[B,n]  per-image linear fields;  J[n,n] symmetric couplings with zero diagonal, shared by the
 *   whole batch;  R reads (independent chains) per image;  `sweeps` sequential heat-bath sweeps with a
 *   geometric inverse-temperature schedule beta0 -> beta1 (beta0 == beta1: plain Gibbs sampling).
 * Outputs: states[B,R,n] uint8 in {0,1} (column j = variable j, the reference's sample order) and,
 * if non-NULL, mean[B,n] float64 = average over reads (np.average(np_samples, axis=0), faster_dqbm.py:780).
 * Draw ids: start state -> offset, sweep t -> offset + t; Philox row = b*R + r, col = variable. */
QBM_API int qbm_qubo_gibbs(const float* lin, const float* J, int B, int R, int n, int sweeps, float beta0,
                           float beta1, uint64_t seed, uint64_t offset, uint8_t* states, double* mean,
                           void* stream);

/* The same with chain0 = global index of the first chain (image 0, read 0) of this call: Philox rows are chain0 + b*R + r,
 * so a minibatch sharded over ranks (images of config #5 split over the GPUs) draws exactly the numbers
 * one process would. */
QBM_API int qbm_qubo_gibbs_ex(const float* lin, const float* J, int B, int R, int n, int sweeps, float beta0,
                              float beta1, uint64_t seed, uint64_t offset, uint32_t chain0, uint8_t* states,
                              double* mean, void* stream);

/* The same with the annealing range of every image read from DEVICE memory: beta_dev[B,2] = (hot, cold) per image, as
 * written by qbm_qbm_beta_range (no host round trip between building the QUBO and sampling it); beta_dev == NULL: the
 * scalars beta0 / beta1 for every image. */
QBM_API int qbm_qubo_gibbs_dev(const float* lin, const float* J, int B, int R, int n, int sweeps, const float* beta_dev,
                               float beta0, float beta1, uint64_t seed, uint64_t offset, uint32_t chain0,
                               uint8_t* states, double* mean, void* stream);

/* The sampler's operands from a QUBO matrix as the sample_Q protocol hands it over (src/qubo/sampler.py:26-33: one n x n
 * matrix per call, minimise s^T Q s; upper-triangular by protocol, any square matrix accepted): Q [n,n] float64 (q_is_f64 != 0)
 * or float32 on the device ->  lin[n] = fp32(diag Q),  J[n,n] = fp32(Q_offdiag + Q_offdiag^T) (symmetric, zero diagonal; the sum
 * in Q's own precision),  rowabs[n] / rowmin[n] as qbm_qbm_couplings delivers them (inputs of qbm_qbm_beta_range),
 * any_offdiag[1] = 1 if a quadratic term is non-zero (the reference short-circuits a QUBO without any: :13-17, :28-29). */
QBM_API int qbm_qubo_from_Q(const void* Q, int q_is_f64, int n, float* lin, float* J, double* rowabs, double* rowmin,
                            int* any_offdiag, void* stream);

/* ---- sparse couplings: the rows of J as CSR entries in LDS (round 4) --------------------------------
 * For QUBOs whose triangle does not fit a CU's LDS (n > 276) but whose NON-ZERO couplings do: the conv-QBM QUBOs behind
 * sample_Q (src/qubo/sampler.py:26-33; n = 717: 10 092 non-zero couplings of 513 372).  Same chain, same draws, same
 * states bit for bit as qbm_qubo_gibbs_dev on the same (lin, J): only where a flip finds its row differs.
 *   cap = qbm_qubo_csr_max_entries(n, B * R)      entries the LDS holds beside the chains' fields (0: bad arguments)
 *   ws  = qbm_qubo_csr_workspace_bytes(n, cap)    device bytes, 8-byte aligned
 *   qbm_qubo_csr_build(J, n, cap, ws)             J [n,n] fp32 symmetric, diagonal ignored -> CSR (ascending columns)
 *   qbm_qubo_gibbs_csr(lin, ws, cap, ...)         the sampler; arguments as qbm_qubo_gibbs_dev
 * The first int32 of the workspace is a STATUS word the build writes: 0, or 1 if J holds more than cap non-zero
 * couplings.  With status 1 the sampler launch does nothing (states, mean untouched): the caller reads the word at its next
 * synchronisation and calls qbm_qubo_gibbs_dev instead.  The second int32 is the number of non-zero couplings found. */
QBM_API long qbm_qubo_csr_max_entries(int n, int chains);
QBM_API size_t qbm_qubo_csr_workspace_bytes(int n, long cap);
QBM_API int qbm_qubo_csr_build(const float* J, int n, long cap, void* workspace, void* stream);
QBM_API int qbm_qubo_gibbs_csr(const float* lin, const void* workspace, long cap, int B, int R, int n, int sweeps,
                               const float* beta_dev, float beta0, float beta1, uint64_t seed, uint64_t offset, uint32_t chain0,
                               uint8_t* states, double* mean, void* stream);

/* uint8 -> float32, n elements (sample_Q returns float32 samples) */
QBM_API int qbm_u8_to_f32(const uint8_t* x, float* y, long n, void* stream);

/* ---- Disc_QBM with device-resident parameters ---------------------------------------------------
 * theta = [W_vh (1+D)*H | W_vo D | b_h H | b_o 1] float64, the reference's arrays (src/model/faster_dqbm.py:196-216:
 * weights_all_visible_to_hidden with row 0 = label row, weights_clamped_visible_to_output, biases_hidden,
 * biases_output) in one device buffer; statistics and their differences use the same layout. */

/* x-dependent QUBO diagonal (qbm_qbm_linear_fields below) with the parameters read from theta. */
QBM_API int qbm_qbm_linear_fields_dev(const double* X, const int32_t* labels, const double* theta, int B, int D, int H,
                                      double beta_eff, float* lin, void* stream);

/* Image-independent couplings of Disc_QBM.create_qubo_matrix_from (src/model/faster_dqbm.py:242-243 clamped: W_hh;
 * :256-271 unclamped: label row of W_vh between the output node and the hidden nodes, W_hh) as the symmetric fp32
 * J[n,n] (zero diagonal) the sampler takes: J_ab = fp32(Q_ab / beta_eff).  w_label = W_vh row 0 ([H], unclamped only),
 * W_hh [H,H] float64 upper triangle or NULL (restricted model).  Also rowabs[n] = sum_b |J_ab| and rowmin[n] = smallest
 * non-zero |J_ab| of the row (0 if none), float64: inputs of qbm_qbm_beta_range. */
QBM_API int qbm_qbm_couplings(const double* w_label, const double* W_hh, int H, int clamped, double beta_eff, float* J,
                              double* rowabs, double* rowmin, void* stream);

/* Annealing range per image, beta[B,2] fp32 = (hot, cold): hot = ln 2 / max_i (|lin_i| + rowabs_i),
 * cold = ln 100 / (smallest non-zero |coefficient| of lin_b and J), cold clamped to [10 hot, 1e4 hot]; (0.1, 1) for a QUBO
 * without usable coefficients.  Stands where the reference's annealer (dwave-neal, absent here) derives its own default range;
 * the rule above is THIS BUILD'S (the clamp and the QUBO-domain coefficients included): parity unpinned, DESIGN.md section 2. */
QBM_API int qbm_qbm_beta_range(const float* lin, const double* rowabs, const double* rowmin, int B, int n, float* beta,
                               void* stream);

/* Statistics of both phases and their difference in one pass (Disc_QBM.get_average_configuration_batch :754-848 twice,
 * then `clamped - unclamped` of train_for_one_iteration :1023-1040): mean_c[B,H], mean_u[B,1+H] float64 per-image means
 * over reads -> diff[theta layout].  Sequential over the images, like qbm_qbm_batch_stats. */
QBM_API int qbm_qbm_stats_diff(const double* X, const int32_t* labels, const double* mean_c, const double* mean_u, int B,
                               int D, int H, double* diff, void* stream);

/* The same with one more output: diff[theta length] = the sum over the images of compute_nll's per-image term
 * (src/model/faster_dqbm.py:986-991: -log(float32([1 - p1, p1])[y] + 1e-12), p1 = mean_u[i,0]); float32 terms, float64 sum in
 * image order.  diff holds qbm theta length + 1 doubles. */
QBM_API int qbm_qbm_stats_diff_nll(const double* X, const int32_t* labels, const double* mean_c, const double* mean_u, int B,
                                   int D, int H, double* diff, void* stream);

/* theta -= lr * (diff / batch), src/model/faster_dqbm.py:1041-1059 (batch = the GLOBAL minibatch size when diff was
 * all-reduced over data-parallel ranks). */
QBM_API int qbm_qbm_apply_update(double* theta, const double* diff, int D, int H, double lr, double batch, void* stream);

/* Disc_QBM.get_average_configuration_batch, src/model/faster_dqbm.py:754-848, from the per-image means over
 * reads (float64, images accumulated in order).  labels != NULL = clamped phase (mean[B,H]), NULL = unclamped
 * (mean[B,1+H], variable 0 = output node).  Outputs (overwritten): W_vh[(1+D),H] with the reference's
 * statistics layout (rows 0..D-1 pixels, row D label: SURVEY.md Appendix A Q7), W_vo[D], b_h[H], b_o[1].
 * W_oo and W_hh statistics are identically zero in the reference (Q8) and are not produced. */
QBM_API int qbm_qbm_batch_stats(const double* X, const int32_t* labels, const double* mean, int B, int D, int H,
                                double* W_vh, double* W_vo, double* b_h, double* b_o, void* stream);

/* x-dependent part of Disc_QBM.create_qubo_matrix_from (src/model/faster_dqbm.py:225-284), float64:
 *   labels != NULL (clamped, n = H):   lin_j = (b_h[j] + label*W_vh[0,j] + sum_k x_k W_vh[1+k,j]) / beta_eff
 *   labels == NULL (unclamped, n = 1+H): lin_0 = (b_o + sum_k x_k W_vo[k]) / beta_eff,
 *                                        lin_{1+j} = (b_h[j] + sum_k x_k W_vh[1+k,j]) / beta_eff
 * X[B,D] f64, W_vh[(1+D),H] f64 (row 0 = label row), W_vo[D] f64, b_h[H] f64; lin[B,n] f32. */
QBM_API int qbm_qbm_linear_fields(const double* X, const int32_t* labels, const double* W_vh, const double* W_vo,
                                  const double* b_h, double b_o, int B, int D, int H, double beta_eff, float* lin,
                                  void* stream);

/* ---- test hooks (bit-compare the device math with the oracle) -------------------------------- */
QBM_API int qbm_test_math(int which, const float* x, float* y, int n, void* stream);
QBM_API int qbm_test_uniform(uint64_t seed, uint64_t offset, uint32_t row0, int rows, int cols, float* out,
                             void* stream);
/* The next fused-tail launch (k_hgrad) of this process makes workgroup `workgroup` misbehave -- kind 1: it leaves at its
 * start without a trace (as one that was not resident in time: its partners' barrier times out), kind 2: it gives up in
 * front of the barrier (its partners go on) -- so that tests can walk the fail-safe without a real fault.  -1 = off. */
QBM_API int qbm_test_hgrad_fault(int workgroup, int kind);

#ifdef __cplusplus
}
#endif
#endif /* QBM_HIP_H */
