/*
 * ganmf_hip.h — C ABI of libganmf_hip.so: the MI355X (gfx950) GANMF / DisGANMF training hot path.
 *
 * This is the drop-in boundary.  Every entry point replaces one interaction the reference has
 * with its numeric runtime (a TensorFlow-1.12 tf.Session); the reference site is cited per
 * function (paths relative to the reference repository root).  The Python host classes in
 * ganmf_amd/ bind exactly these symbols through ctypes; nothing else crosses the boundary.
 *
 * Conventions
 *   - plain pointers and sizes only; caller owns every host buffer; the library copies in/out
 *     before returning; the handle owns all device memory, its HIP stream and (optionally) its
 *     RCCL communicator.
 *   - return value 0 = OK, negative = error; ganmf_last_error() returns a thread-local message.
 *     No exceptions cross the ABI and the library never calls exit().
 *   - a handle is driven by one host thread at a time; distinct handles are independent.
 *   - all floating point data are IEEE float32 (the reference's dtype, GANMF.py:108), ids int32
 *     (GANMF.py:109), CSR row pointers int64.
 *   - "training orientation": rows = the generator's users, columns = profile width.  In item
 *     mode the host passes URM^T (GANMF.py:32-36).
 */
#ifndef GANMF_HIP_H
#define GANMF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GANMF_ABI_VERSION 1

typedef struct ganmf_handle ganmf_handle;

enum { GANMF_MODEL_GANMF = 0, GANMF_MODEL_DISGANMF = 1 };
enum { GANMF_ACT_LINEAR = 0, GANMF_ACT_TANH = 1, GANMF_ACT_RELU = 2, GANMF_ACT_SIGMOID = 3 };

/* Tensor ids.  Discriminator tensors are numbered in the order of the reference's
 * tf.get_collection(TRAINABLE_VARIABLES, scope) (GANMF.py:121, DisGANMF.py:121):
 *   GANMF:    0 encoding/kernel [N,e]  1 encoding/bias [e]  2 decoding/kernel [e,N]  3 decoding/bias [N]
 *   DisGANMF: 2l layer_l/kernel  2l+1 layer_l/bias (l < d_layers)  2L D_output/kernel [e,1]  2L+1 D_output/bias [1]
 * Generator tensors (GANMF.py:122): */
#define GANMF_T_USER_EMB 100 /* generator/user_embeddings [U,k] */
#define GANMF_T_ITEM_EMB 101 /* generator/item_embeddings [N,k] */

/* which copy of a tensor */
enum { GANMF_SLOT_PARAM = 0, GANMF_SLOT_ADAM_M = 1, GANMF_SLOT_ADAM_V = 2, GANMF_SLOT_BEST = 3 };

/* Hyper-parameters = the kwargs of fit() (GANMF.py:88-90, DisGANMF.py:83-85) that enter the graph. */
typedef struct ganmf_cfg {
  int32_t abi_version;      /* GANMF_ABI_VERSION */
  int32_t model;            /* GANMF_MODEL_* */
  int64_t num_users;        /* rows held by THIS handle (a shard when world_size > 1) */
  int64_t num_items;        /* profile width N */
  int32_t num_factors;      /* k */
  int32_t emb_dim;          /* GANMF emb_dim / DisGANMF d_nodes */
  int32_t d_layers;         /* DisGANMF only */
  int32_t d_act;            /* DisGANMF only, GANMF_ACT_* */
  int32_t batch_size;       /* rows per minibatch on this handle */
  float d_lr, g_lr, d_reg, g_reg;
  float m;                  /* GANMF hinge margin multiplier */
  float recon_coefficient;  /* alpha */
  int32_t device;           /* HIP device ordinal */
  int32_t world_size;       /* data-parallel replicas (1 = single GPU) */
  int32_t rank;
  int64_t row_offset;       /* global id of local row 0 (DisGANMF feeds float(uid), DisGANMF.py:110) */
  uint32_t flags;           /* GANMF_FLAG_* */
} ganmf_cfg;

#define GANMF_FLAG_NONE 0u
/* Arithmetic of the GEMM K loops (results are float32 tensors in every case):
 *   default          fp32-accurate: per GEMM either the fp32 MFMA or the bf16 matrix cores with each operand
 *                    split exactly into three bf16 pieces and six piece products accumulated in fp32 (chosen by
 *                    the planner from the grid size; both pass every parity test)
 *   GANMF_FLAG_MFMA_F32   force v_mfma_f32_32x32x2_f32 on the fp32 operands
 *   GANMF_FLAG_MFMA_BF16  operands rounded to ONE bf16 (RNE), fp32 accumulate, fp32 master weights and Adam
 *                    (8 significant bits, fp32's exponent range: no scaling needed).
 *   GANMF_FLAG_MFMA_F16   BASELINE configs[4] as written: operands rounded to ONE IEEE fp16 for
 *                    v_mfma_f32_32x32x16_f16 (same matrix-core rate as bf16, 11 significant bits), fp32 accumulate,
 *                    fp32 master weights and Adam.  Gradient-carrying operands are scaled by a power of two at
 *                    conversion and the sum scaled back in fp32 (static loss scaling per GEMM).
 *   In both low-precision modes DisGANMF's float(uid) input column (DisGANMF.py:59,110-111) does NOT go through
 *   the low-precision K loop: its forward term is added in fp32 in the layer-0 epilogue and its weight-row
 *   gradient is an fp32 reduction of its own. */
#define GANMF_FLAG_MFMA_F32 1u
#define GANMF_FLAG_MFMA_BF16 2u
#define GANMF_FLAG_MFMA_F16 4u

/* Replaces: tf.reset_default_graph + build() + optimizers + Session + initialize_all_variables
 * (GANMF.py:97-105,146-149).  Parameters start at zero; the host uploads initial values with
 * ganmf_set_tensor (the reference's Glorot init happens inside TF and is not reproducible). */
int ganmf_create(const ganmf_cfg* cfg, ganmf_handle** out);
int ganmf_destroy(ganmf_handle* h);

/* Data-parallel setup (no reference counterpart: the reference is single-device).  Rank 0 calls
 * ganmf_comm_unique_id, the host broadcasts the 128 bytes, every rank calls ganmf_comm_init. */
int ganmf_comm_unique_id(uint8_t out128[128]);
int ganmf_comm_init(ganmf_handle* h, const uint8_t id128[128]);
/* What the communicator itself reports (bench.py's `parallelism` object): ranks in the RCCL communicator (ncclCommCount) and this
 * handle's rank in it (ncclCommUserRank); the loopback communicator reports its group.  0 / -1 without a communicator. */
int ganmf_comm_info(ganmf_handle* h, int32_t* world_size, int32_t* rank);
/* In-process alternative to the RCCL communicator: the world_size handles that call this with the same group_id
 * (same process, same device, one host thread each) all-reduce among themselves by rendezvous; sums run in rank
 * order.  For exercising the data-parallel path with world_size > 1 on one GPU (tests/test_gpu_dist_local.py). */
int ganmf_comm_init_local(ganmf_handle* h, int32_t group_id);
/* Ends the handle's communicator WITHOUT touching anything else of the handle: the loopback group is marked failed and its waiting
 * members wake up with an error; an RCCL communicator is aborted (ncclCommAbort).  The one entry point that may be called from
 * another host thread while the handle's own thread is inside a training call -- how a driver whose peer rank failed gets the
 * survivors out of a collective that can no longer complete, BEFORE it destroys their handles (ganmf_amd/dist.py _ThreadRank.kill).
 * Training calls on the handle fail from then on; ganmf_destroy is the only thing left to do with it. */
int ganmf_comm_abort(ganmf_handle* h);

/* Replaces the per-minibatch host work `URM_train[uids].toarray()` + feed_dict upload
 * (GANMF.py:183-187,198-201): the CSR matrix (training orientation, this handle's rows) is
 * uploaded ONCE and minibatch rows are expanded on the device. */
int ganmf_set_urm_csr(ganmf_handle* h, const int64_t* indptr, const int32_t* indices,
                      const float* data, int64_t n_rows, int64_t n_cols);

/* Replaces sess.run(var) / sess.run(var.assign(x)) on one variable (GANMF.py:294-302,
 * Utils_.py:292-294,305-310).  `n` must equal the tensor's element count (row-major, unpadded).
 * Data-parallel handles (communicator attached, world_size > 1): the Adam moments (GANMF_SLOT_ADAM_M / _V) of the
 * REPLICATED tensors (item_embeddings and every discriminator tensor) live sharded over the ranks -- rank r updates
 * slice r only -- so these two slots are rejected (-1) for them; parameters and the best snapshot are whole and identical
 * on every rank, and user_embeddings (rank-owned rows) is whole in every slot. */
int ganmf_set_tensor(ganmf_handle* h, int tensor_id, int slot, const float* host, int64_t n);
int ganmf_get_tensor(ganmf_handle* h, int tensor_id, int slot, float* host, int64_t n);
int ganmf_tensor_shape(ganmf_handle* h, int tensor_id, int64_t* rows, int64_t* cols);

/* Adam beta-power accumulators of the two optimizers (AdamOptimizer._finish): 4 floats
 * {b1p_D, b2p_D, b1p_G, b2p_G}; exposed for save/restore and tests. */
int ganmf_get_adam_powers(ganmf_handle* h, float out4[4]);
int ganmf_set_adam_powers(ganmf_handle* h, const float in4[4]);

/* Replaces one pass of the `while epoch` body (GANMF.py:175-203): given this epoch's already
 * shuffled row order `perm` (GANMF.py:175; local row ids), runs d_steps passes of discriminator
 * updates over consecutive slices of batch_size rows (ragged tail kept) and then g_steps passes
 * of generator updates over the same slices.  Losses are the pre-update values the reference's
 * sess.run([train, loss]) returns (GANMF.py:186-187,200-201); arrays hold
 * d_steps*ceil(n/batch) resp. g_steps*ceil(n/batch) floats (may be NULL).
 * With world_size > 1 `n_steps_per_pass` (>= ceil(n/batch)) forces every rank to issue the same
 * number of collectives; pass 0 for the default.  `global_batch_rows[i]` = rows in the i-th slice
 * summed over all ranks (NULL when world_size == 1).  Blocking.
 * Because the reference freezes the generator for a whole discriminator pass and the discriminator for a whole
 * generator pass (GANMF.py:176-189, 191-203), the call does per PASS what does not depend on the steps before it --
 * the CSR rows and generated rows of every full minibatch of a discriminator pass; with g_reg == 0 the all-rows
 * Adam over user_embeddings of a generator pass -- with the float operations of the per-step form, bit for bit
 * (GANMF_TUNE=pass_stage=0,lazy_rows=0 runs every step on its own).  Parameters and moments
 * are consistent whenever the call has returned. */
int ganmf_train_epoch(ganmf_handle* h, const int32_t* perm, int64_t n, int32_t d_steps,
                      int32_t g_steps, int64_t n_steps_per_pass, const int32_t* global_batch_rows,
                      float* d_losses, float* g_losses);

/* The same pass with slices of GIVEN sizes: slice i takes the next local_batch_rows[i] rows of `perm`
 * (0 <= local_batch_rows[i] <= batch_size, their sum == n, n_steps_per_pass entries), global_batch_rows[i]
 * (>= local_batch_rows[i], >= 1) as above.  This is how a row-sharded fit() replays the REFERENCE's minibatch
 * schedule (GANMF.py:175-203) on several GPUs: every global minibatch of the single shuffled permutation is
 * split by row owner, so rank r's i-th slice is "the rows of global minibatch i that rank r owns" -- a variable
 * count -- and the union over ranks is exactly the minibatch the reference would have drawn
 * (ganmf_amd/dist.py split_by_owner).  Loss arrays hold d_steps * n_steps_per_pass resp. g_steps * n_steps_per_pass
 * floats.  Blocking. */
int ganmf_train_epoch_ragged(ganmf_handle* h, const int32_t* perm, int64_t n, int32_t d_steps, int32_t g_steps,
                             int64_t n_steps_per_pass, const int32_t* global_batch_rows,
                             const int32_t* local_batch_rows, float* d_losses, float* g_losses);

/* Single updates on an explicit id list (same arithmetic as inside ganmf_train_epoch); used by
 * tests and by callers that schedule batches themselves.  kind: 0 = D-step, 1 = G-step. */
int ganmf_train_step(ganmf_handle* h, int kind, const int32_t* uids, int32_t n, float* loss);

/* Replaces _compute_item_score (GANMF.py:285-292, DisGANMF.py:257-262).
 *   transposed = 0 (user mode): out[i, :] = U[ids[i]] . V^T          -> [n, num_items]
 *   transposed = 1 (item mode): out[i, :] = (U V^T)^T[ids[i]] = V[ids[i]] . U^T -> [n, num_users]
 * (the item-mode product is formed directly; the full matrix is never materialised). */
int ganmf_scores(ganmf_handle* h, const int32_t* ids, int64_t n, int transposed, float* out);

/* Replaces the per-user part of BaseRecommender.recommend (Base/BaseRecommender.py:189-234): the scores
 * of ganmf_scores stay on the device, already-seen items are set to -inf and the `cutoff` best items per
 * row are selected there; only n*cutoff ids (and their scores) cross PCIe.  SURVEY §8(f) row 1.
 * ganmf_set_seen_csr uploads URM_train in EVALUATION orientation (rows = users as evaluators see them,
 * columns = items; n_rows/n_cols must match the id domain / score width of the chosen `transposed`).
 * Ties go to the smaller item id; rows with fewer than `cutoff` finite scores are padded with -1.
 * cutoff <= GANMF_RECOMMEND_MAX_CUTOFF (the selection runs `cutoff` arg-max rounds per row; full rankings are
 * the job of ganmf_scores + a host sort). */
#define GANMF_RECOMMEND_MAX_CUTOFF 1024
int ganmf_set_seen_csr(ganmf_handle* h, const int64_t* indptr, const int32_t* indices, int64_t n_rows, int64_t n_cols);
int ganmf_recommend(ganmf_handle* h, const int32_t* ids, int64_t n, int transposed, int32_t cutoff, int remove_seen,
                    int32_t* out_items, float* out_scores);

/* The MF contract's two masking rules for everything scored afterwards -- ganmf_scores, ganmf_recommend, ganmf_evaluate
 * (Base/BaseMatrixFactorizationRecommender.py:113-119: `items_to_compute` given -> every OTHER item scores -inf; :128-143: a user
 * without a training interaction ("cold") scores -inf for ALL items; the reference's GANMF._compute_item_score, GANMF.py:285-292,
 * accepts items_to_compute and ignores it).  items == NULL / n_items == 0: no item restriction.  mask_cold_rows != 0: rows that are
 * empty in the matrix of ganmf_set_seen_csr (URM_train, evaluation orientation) are cold; it must be set when the scoring call is
 * made.  The filter stays until it is set again.  A handle starts WITHOUT either mask, which is the reference GANMF's own contract
 * (GANMF.py:285-292: finite scores for every user, items_to_compute ignored); the host classes switch the masks on only under
 * score_contract="mf" (ganmf_amd/GANMF.py). */
int ganmf_set_score_filter(ganmf_handle* h, const int32_t* items, int64_t n_items, int mask_cold_rows);

/* Replaces save_current_model / load_model (GANMF.py:249-255, Utils_.py:292-294): device-side
 * copies of all trainable tensors to / from their `best` twins. */
int ganmf_snapshot_best(ganmf_handle* h);
int ganmf_restore_best(ganmf_handle* h);

/* Measurement hooks (bench.py).  ganmf_profile_enable(1) makes every kernel launch of the
 * training step be bracketed by hipEvents on the handle's stream; ganmf_profile_read returns per
 * kernel class the number of launches, total milliseconds and total algorithmic FLOPs / bytes. */
#define GANMF_PROF_MAX 48
typedef struct ganmf_prof_entry {
  char name[48];
  int64_t launches;
  double ms;
  double flops;   /* algorithmic */
  double bytes;   /* algorithmic HBM bytes */
} ganmf_prof_entry;
int ganmf_profile_enable(ganmf_handle* h, int on);
/* Device time of a region of calls (bench.py's timed K steps; SURVEY 8(d): hipEvent timing): stop = 0 records a start event on
 * the handle's stream, stop = 1 records the stop event, waits for it and returns the milliseconds between the two -- the time the
 * stream spent on everything enqueued in between, host gaps between blocking calls included, launch latency of the first call
 * excluded.  Nothing in the reference corresponds to it (GANMF.py:172-203 times nothing). */
int ganmf_stream_timer(ganmf_handle* h, int stop, double* ms);
int ganmf_profile_read(ganmf_handle* h, ganmf_prof_entry* out, int32_t cap, int32_t* n_out);

/* Hold-out evaluation on the device (replaces the per-user metric loop of
 * Base/Evaluation/Evaluator.py:262-335 with the definitions of Base/Evaluation/metrics.py): ganmf_evaluate ranks the rows
 * `ids` exactly as ganmf_recommend does (cutoff = the largest of `cutoffs`), looks every recommended item up in the test
 * matrix and returns, per cut-off, the SUMS over the n users of
 *   ROC_AUC, PRECISION, PRECISION_RECALL_MIN_DEN, RECALL, MAP, MRR, NDCG, HIT_RATE, ARHR      (GANMF_EVAL_METRICS values, this order)
 * in float64; only n_cutoffs * 9 doubles cross PCIe.  ganmf_set_test_csr uploads URM_test in EVALUATION orientation with
 * column indices sorted inside each row and `gains` = 2^rating - 1 per stored entry; `disc[k]` = 1 / ln(k + 2) and
 * `ideal_cum[i, k]` = the prefix sums of user i's ideal DCG terms (both [.., max cutoff], formed by the caller the way
 * the reference forms them, in float32 where it does).  At most GANMF_EVAL_MAX_CUTOFFS cut-offs per call. */
#define GANMF_EVAL_METRICS 9
#define GANMF_EVAL_MAX_CUTOFFS 8
int ganmf_set_test_csr(ganmf_handle* h, const int64_t* indptr, const int32_t* indices, const double* gains, int64_t n_rows,
                       int64_t n_cols);
int ganmf_evaluate(ganmf_handle* h, const int32_t* ids, int64_t n, int transposed, int remove_seen, const int32_t* cutoffs,
                   int32_t n_cutoffs, const double* disc, const double* ideal_cum, double* sums);

/* Device-resident scoring GEMM timing (no D2H): scores for the first n rows, `iters` launches;
 * returns average milliseconds per launch measured with hipEvents on the handle's stream. */
int ganmf_bench_scores(ganmf_handle* h, int64_t n, int transposed, int32_t iters, float* ms_per_launch);

/* Stand-alone fp32 MFMA GEMM on host buffers, C[M,N] = op(A) . op(B):
 *   a_kmajor = 0: A is [M, K] row-major;  1: A is [K, M] row-major
 *   b_kmajor = 0: B is [N, K] row-major;  1: B is [K, N] row-major
 * tile = 0 auto, 64 or 128; nsplit = 0 auto.  `iters` >= 1 repeats the launch on resident data and
 * returns the average kernel milliseconds in *ms (may be NULL).  Test and bench entry. */
int ganmf_gemm_f32(int device, const float* A, const float* B, float* C, int64_t M, int64_t N,
                   int64_t K, int a_kmajor, int b_kmajor, int tile, int nsplit, int iters, float* ms);

/* Host-only helper of the checkpoint writer/reader (ganmf_amd/tf_bundle.py): CRC-32C (Castagnoli), the
 * checksum tf.train.Saver stores per tensor and per table block (GANMF.py:309-314,337-339).  Pass crc = 0
 * to start; feed the previous return value to continue over a second buffer. */
uint32_t ganmf_crc32c(uint32_t crc, const void* data, uint64_t n);

int ganmf_device_count(void);
int ganmf_abi_version(void);
const char* ganmf_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* GANMF_HIP_H */
