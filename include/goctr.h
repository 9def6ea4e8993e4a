/*
 * goctr.h -- C-ABI of libgoctr_hip.so, the MI355X (gfx950) engine behind go-ctr's hot path.
 *
 * This is the drop-in boundary: plain C, opaque handles, plain pointers and sizes, no torch / C++
 * types.  Every entry point names the reference (auxten/go-ctr, Go) interface it replaces
 * (file:line relative to the reference repo); INTEGRATION.md shows the cgo stub a go-ctr maintainer
 * adds on the Go side.  Conventions:
 *   - every function returns int status, 0 = ok; goctr_last_error() returns a thread-local string;
 *   - no callback into the host language, no host pointer retained after a call returns;
 *   - host buffers are row-major, float32 for DIN / YouTube (gorgonia tensor.Float32,
 *     model/model.go:14), float64 for the sklearn-port MLP and item2vec (as in the reference);
 *   - any host thread may call any entry point on any handle.  Calls that queue work on the engine's main stream
 *     (training, uploads, dataset builds) are serialised engine-wide by an internal lock.  The SERVING entry points --
 *     goctr_batch_predict, goctr_rank, goctr_predict_dense: what concurrent gin handler goroutines reach through
 *     Rank -> BatchPredict -> PredictAbstract.Predict, recommend/api.go:106-131 -- run CONCURRENTLY: each call takes a
 *     serving slot (own HIP stream, pinned staging buffers, forward workspace; GOCTR_SERVE_SLOTS of them, default 8,
 *     handed out first-come-first-served) under a shared lock of the model and of the embedding table, so calls on one
 *     model or on different models overlap on the GPU while a training call on that model waits for them (and they for
 *     it).  Small goctr_rank / goctr_batch_predict calls (<= GOCTR_SERVE_COALESCE rows, default 1024) that arrive while
 *     EVERY slot is busy are coalesced into one launch sequence on the same (recsys, model) pair (a micro-batcher); scores
 *     do not depend on whether or with what a call was coalesced (rows are scored independently, same kernel, same bits);
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails loudly.
 */
#ifndef GOCTR_H
#define GOCTR_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct goctr_model goctr_model;     /* DIN / YouTube weights + Adam state on the device   */
typedef struct goctr_emb goctr_emb;         /* item-embedding table [V,D] f32 resident in HBM      */
typedef struct goctr_dataset goctr_dataset; /* training / scoring rows resident in HBM             */
typedef struct goctr_mlp goctr_mlp;         /* sklearn-port MLP (float64)                          */
typedef struct goctr_w2v goctr_w2v;         /* item2vec state (param, HS node vectors, paths)      */

/* ---------------------------------------------------------------- runtime ---------------- */
/* Binds the calling process to one GPU (one process per GPU) and creates the engine's streams. */
int goctr_init(int device_ordinal);
/* ONE process drives n ranks (SURVEY 8(b)'s goctr_init(n_devices, device_ids)): engine k is bound to HIP device
 * device_ids[k], with its own streams, device arena and lock.  Distinct devices get an RCCL communicator over xGMI
 * (ncclCommInitAll); a device that appears more than once gets several LOGICAL ranks joined by the loop-back communicator
 * (csrc/comm.hip: fixed-rank-order sums and device copies ordered by events and a host barrier -- RCCL rejects duplicate
 * GPUs), which runs every W > 1 code path on a one-GPU box (GOCTR_COMM=loopback forces it for distinct devices too).
 * Handles created afterwards live on engine 0 (or the engine goctr_engine_select chose for the calling thread); every
 * entry point that takes a handle runs on the handle's engine, whatever thread calls it.  A training call whose
 * goctr_train_cfg.devices = n then shards each global batch over the n ranks from n internal host threads -- the single Go
 * process of recommend.Train (recommend/rcmd.go:196-246) needs no launcher.  Idempotent for the same list. */
int goctr_init_devices(int n, const int* device_ids);
int goctr_engine_count(int* n);
/* Tools: device time engine k (rank k) spent in its part of the LAST multi-device training call (cfg.devices = n): events on
 * the rank's own stream around its steps; waits for that rank's part to finish.  bench.py --single-process reports it per rank. */
int goctr_engine_call_ms(int k, double* ms);
/* Tools / tests: bind the CALLING THREAD to engine k for the handles it creates from now on (a Go caller would need
 * runtime.LockOSThread; the single-call entries above need neither this nor the next function). */
int goctr_engine_select(int k);
/* Tools / tests: let the calling thread's engine take part in its group's collectives (per-rank calls from n host
 * threads, one per engine); off by default so that a plain call on one engine of a group is a single-device call. */
int goctr_comm_group_enable(int on);
int goctr_device_count(int* n);
/* blocks until all work queued by this library has finished: waits on every engine's own streams -- deliberately not a
 * device-wide wait, which would invalidate the stream capture of another host thread that is building its step graphs on
 * the same device (a training goroutine beside this caller) */
int goctr_sync(void);
const char* goctr_last_error(void);
const char* goctr_version(void);
/* name / CU count / HBM bytes of the bound device */
int goctr_device_info(char* name, size_t name_cap, int* compute_units, int64_t* hbm_bytes);

/* Data-parallel communicator (RCCL over xGMI).  rank 0 calls goctr_comm_unique_id, the 128 bytes are
 * distributed by the host launcher (any side channel), every rank calls goctr_comm_init.
 * No reference counterpart: go-ctr is single-process. */
int goctr_comm_unique_id(uint8_t id[128]);
int goctr_comm_init(int rank, int world, const uint8_t id[128]);
int goctr_comm_world(int* rank, int* world);
/* How the data-parallel step of the calling thread's engine issues its dense all-reduce: 1 = as a node of the multi-step
 * graphs (the captured RCCL collective passed its self-test on this communicator), -1 = between graph launches (self-test
 * failed, GOCTR_DP_CAPTURE_COMM=0, or a loop-back communicator), 0 = not decided yet (no data-parallel step has run). */
int goctr_comm_capture_mode(int* mode);
/* sum-all-reduce of a host double (used for timing / cost aggregation); world==1 => identity */
int goctr_comm_allreduce_f64(double* v, int n);
int goctr_comm_destroy(void);

/* ---------------------------------------------------------------- DIN / YouTube ----------- */
/* model kinds: model/din/din.go:21 (DinNet) and model/youtube/dnn.go:18 (YoutubeDnn) */
enum { GOCTR_DIN = 0, GOCTR_YOUTUBE = 1 };
/* attention activation: cosine = din.go:231-237 (shipping), euclid = din.go:230 (commented-out variant) */
enum { GOCTR_ATT_COSINE = 0, GOCTR_ATT_EUCLID = 1 };
/* learnable tensors, in Learnable() order (din.go:161-169): mlp0, mlp1, mlp2, att0 */
enum { GOCTR_W0 = 0, GOCTR_W1 = 1, GOCTR_W2 = 2, GOCTR_ATT0 = 3 };

typedef struct {
  int kind;       /* GOCTR_DIN | GOCTR_YOUTUBE */
  int att;        /* GOCTR_ATT_* (DIN only) */
  int U, T, D, C; /* uProfileDim, uBehaviorSize, uBehaviorDim (= iFeatureDim), cFeatureDim
                     -- the arguments of din.NewDinNet (din.go:171-175) */
  int H1, H2;     /* hidden widths; 200 / 80 in the reference (din.go:17-18) */
} goctr_ctr_cfg;

/* replaces din.NewDinNet / youtube.NewYoutubeDnn (din.go:171, dnn.go:119).  Weights start at zero
 * with att0 = 1; the host sets the N(0,1) init (din.go:187-191) or JSON weights through
 * goctr_model_set_weights (NewDinNetFromJson, din.go:82).  Fails if kind==DIN and dims mismatch
 * like din.go:176-178. */
int goctr_model_create(const goctr_ctr_cfg* cfg, goctr_model** out);
void goctr_model_destroy(goctr_model* m);
/* flat row-major float32 arrays exactly as in the dinModel JSON (din.go:41-52): W0 [I,H1],
 * W1 [H1,H2], W2 [H2,1], att0 [1,T];  replaces Marshal / NewDinNetFromJson (din.go:62,82) */
int goctr_model_set_weights(goctr_model* m, int tensor_id, const float* host, size_t n);
int goctr_model_get_weights(goctr_model* m, int tensor_id, float* host, size_t n);
/* EXTENSION with no reference counterpart (the reference trains with frozen embeddings, din.go:161-169 /
 * dnn.go:152-154; SURVEY F3, 8(e) "Trainable embeddings"): lr > 0 makes every following training step on an id-mode
 * dataset also update the rows of the goctr_emb table it is given,  E[id] -= lr * dCost/dE[id]  (plain SGD
 * scatter-add, deterministic).  lr = 0 (the default) restores the reference's semantics.  D <= 64.  With a communicator
 * the table is replicated and the row gradients take a bucketed exchange: owner = id % world, all-to-all
 * of the deduplicated (id, fixed-point row) pairs, exact owner-side sums, all-gather of (id, delta) -- traffic
 * proportional to the ids the batches touch; the replicas stay bit-identical. */
int goctr_model_set_embedding_training(goctr_model* m, double lr);
/* bytes this rank SENT in the last step's sparse-gradient exchange (ids + 64-bit fixed-point rows to their owners, then
 * the owners' (id, delta) lists to every rank; self included); 0 without a communicator */
int goctr_model_sparse_exchange_bytes(goctr_model* m, double* bytes);
/* The resident sparse plan the last embedding-training call built for its dataset (csrc/emb_plan.hip; tests / tools): per
 * batch the (sample << 12 | slot) pairs sorted by embedding row -- stable, so two builds are byte-identical -- with their
 * slot and row, the distinct rows in ascending owner-major order and the run starts.  Call with the array pointers NULL to
 * get the sizes (*n_pairs, *n_slots, *n_batches), then with arrays of pair / pslot / pid [n_pairs], slot_id [n_slots],
 * slot_off [n_slots + n_batches], pair_off / slot_base [n_batches + 1].  Fails when no plan is resident. */
int goctr_model_get_emb_plan(goctr_model* m, int64_t* n_batches, int64_t* n_pairs, int64_t* n_slots, int32_t* pair, int32_t* pslot,
                             int32_t* pid, int32_t* slot_id, uint32_t* slot_off, int64_t* pair_off, int64_t* slot_base);
/* wall time (host clock around the build, which ends with the one synchronisation it needs) and batch count of the resident
 * plan's build: what bench.py's --train-emb lines charge to the first epoch */
int goctr_model_emb_plan_build_ms(goctr_model* m, double* ms, int64_t* n_batches);
/* resets the Adam moments and the step counter (a fresh gorgonia AdamSolver, model.go:88) */
int goctr_model_reset_optimizer(goctr_model* m);
/* Optimizer state for checkpoint / resume ("dinModel JSON ... with optimizer state added for resume";
 * the reference's din.go:41-80 / dnn.go:38-61 JSON holds weights only, so a resumed model.Train there restarts Adam
 * from zero moments).  which: 0 = first moment, 1 = second moment; same shapes as goctr_model_get_weights.
 * step = Adam iteration count = dropout stream position. */
int goctr_model_get_moments(goctr_model* m, int tensor_id, int which, float* host, size_t n);
int goctr_model_set_moments(goctr_model* m, int tensor_id, int which, const float* host, size_t n);
int goctr_model_get_step(goctr_model* m, uint32_t* step);
int goctr_model_set_step(goctr_model* m, uint32_t step);

typedef struct {
  int batch;       /* batchSize  (model.go:28) */
  int epochs;      /* epochs     (model.go:28) */
  int early_stop;  /* earlyStop  (model.go:28; 0 = off) */
  double lr, l2;   /* 0.01, 1e-4 (model.go:88) */
  double beta1, beta2, eps;        /* gorgonia Adam defaults .9 .999 1e-8 */
  int adam_div_by_batch;           /* WithBatchSize(B): 1 (model.go:88) */
  int adam_l2_before_batch_div;    /* gorgonia order, 1 */
  int dropout_mode;                /* 0 off, 1 explicit masks (single-step entry only), 2 counter-hash (default: the
                                      reference ALWAYS trains with Dropout, din.go:307-312 / dnn.go:173-175; its masks
                                      come from Go's math/rand, so mask bits are unpinned, the distribution is not) */
  float p0, p1;                    /* 0.005/0.005 DIN (din.go:204-205), 0.003/0.003 YouTube (dnn.go:136-137) */
  uint32_t seed;
  int devices;                     /* 0 / 1: the model's own engine.  n > 1 (= the n of goctr_init_devices): data parallel
                                      inside this ONE call -- `batch` stays the GLOBAL batch of model.Train (model.go:28), rank r
                                      takes rows [r, r+1) * batch/n of every batch (batch % n == 0), the flat gradient buffer is
                                      all-reduced once per step, every rank applies the same Adam update; with embedding training
                                      the sparse row gradients take the bucketed exchange.  The result lands in the handles the
                                      caller passed (rank 0); replicas on the other engines are kept for the next call.
                                      No reference counterpart: go-ctr trains on one CPU process. */
} goctr_train_cfg;
void goctr_train_cfg_default(goctr_train_cfg* c); /* the reference's literals */

/* --- dense-X (drop-in / parity) mode: the TrainSample layout of recommend/rcmd.go:56-63,132-137.
 * ranges = {UserProfileRange, UserBehaviorRange, ItemFeatureRange, CtxFeatureRange} as 8 ints. */

/* replaces model.Train (model/model.go:27-213) as called from dinImpl.Fit
 * (example/movielens/dinimpl.go:62-67): uploads X,Y once, runs the whole epoch loop on the device
 * (zero-padded last batch, Adam per batch, cost of the LAST batch per epoch, early stop).
 * epoch_costs [epochs] and *epochs_run are outputs. */
int goctr_train_dense(goctr_model* m, const float* X, const float* Y, int64_t rows, int xcols,
                      const int ranges[8], const goctr_train_cfg* cfg, float* epoch_costs, int* epochs_run);
/* replaces model.InitForwardOnlyVm + model.Predict (model.go:215-352) as called from
 * dinImpl.Predict (dinimpl.go:32-42): batches of `batch`, zero padding, first end-start outputs
 * kept; no dropout (the JSON round trip of dinimpl.go:73-89 drops d0/d1). */
int goctr_predict_dense(goctr_model* m, const float* X, int64_t rows, int xcols, const int ranges[8],
                        int batch, float* y_out);
/* One instrumented step WITHOUT the parameter update, for parity tests: forward + BCE
 * (model/cost.go:9-17) + backward over one batch of B rows of which the first `valid` come from
 * X (the rest are zero rows, model.go:357-371).  m0/m1: explicit dropout masks for dropout_mode 1
 * ([B,H1], [B,H2]) or NULL.  Any output pointer may be NULL. */
int goctr_loss_grad_dense(goctr_model* m, const float* X, const float* Y, int valid, int B, int xcols,
                          const int ranges[8], const goctr_train_cfg* cfg, uint32_t step,
                          const float* m0, const float* m1,
                          float* cost, float* gW0, float* gW1, float* gW2, float* gatt0, float* y_out);

/* --- id (performance) mode: the embedding table and the sample keys live in HBM; replaces the
 * host-side string-map gather of recommend.GetSampleVector (rcmd.go:462-536). */
int goctr_emb_create(int64_t V, int D, const float* host_rows /* [V,D] or NULL = zeros */, goctr_emb** out);
int goctr_emb_set_rows(goctr_emb* e, int64_t first, int64_t n, const float* host_rows);
int goctr_emb_get_rows(goctr_emb* e, int64_t first, int64_t n, float* host_rows);
void goctr_emb_destroy(goctr_emb* e);
/* standalone gather = the row-assembly half of GetSampleVector (rcmd.go:497-533): out row =
 * [user | emb[ub_ids[0..T)] | emb[item] | ctx]; id < 0 or >= V => zero row.  Device-resident
 * inputs come from a dataset handle; this host-buffer form is for bit-exact parity checks. */
int goctr_gather_rows(goctr_emb* e, const int32_t* ub_ids, const int32_t* item_ids, const float* user_feat,
                      int U, const float* ctx_feat, int C, int T, int64_t rows, float* X_out);

/* device-resident sample sets (uploaded once; reused by train_steps / predict_dataset) */
int goctr_dataset_create_dense(const float* X, const float* Y /* may be NULL */, int64_t rows, int xcols,
                               const int ranges[8], goctr_dataset** out);
int goctr_dataset_create_ids(const int32_t* ub_ids /*[rows,T]*/, const int32_t* item_ids /*[rows]*/,
                             const float* user_feat /*[rows,U]*/, int U, const float* ctx_feat /*[rows,C]*/,
                             int C, int T, const float* Y /* may be NULL */, int64_t rows, goctr_dataset** out);
void goctr_dataset_destroy(goctr_dataset* d);

/* ---- device-side sample assembly: replaces the per-sample host gather of GetSample /
 * GetSampleVector (recommend/rcmd.go:339-536) and the ubcache lookup it calls (feature/ubcache/cache.go:58-94).
 * The behaviour cache is a CSR resident in HBM: user u's sequence = items/ts[off[u] .. off[u+1]) in timestamp-
 * DESCENDING order (cache.go:8).  A key (user, maxTs) selects Filter(maxTs, T): the first T entries with ts <= maxTs
 * (maxTs == 0: from the newest, cache.go:72-74); unused slots are -1 (zero embedding rows, rcmd.go:497-505). */
typedef struct goctr_ubcache goctr_ubcache;
int goctr_ubcache_create(int64_t n_users, const int64_t* off /*[n_users+1]*/, const int32_t* items, const int64_t* ts,
                         goctr_ubcache** out);
void goctr_ubcache_destroy(goctr_ubcache* c);
/* UserBehaviorCache.Get for `rows` keys at once; out_ids [rows, T] */
int goctr_ubcache_get(goctr_ubcache* c, const int32_t* users, const int64_t* max_ts /* may be NULL = 0 */, int64_t rows,
                      int T, int32_t* out_ids);
/* id-mode dataset assembled on the device from sample keys (rcmd.Sample{UserId, ItemId, Timestamp}, rcmd.go:65-71):
 * behaviour ids from the cache, user_table[user] and item_table[item] rows as the dense side features */
int goctr_dataset_create_keys(goctr_ubcache* c, const float* user_table /*[n_users,U]*/, int64_t n_users, int U,
                              const float* item_table /*[n_items,C]*/, int64_t n_items, int C, const int32_t* users,
                              const int32_t* items, const int64_t* ts, const float* Y /* may be NULL */, int64_t rows, int T,
                              goctr_dataset** out);
int goctr_dataset_get_ids(goctr_dataset* d, int32_t* ub_ids, float* user_feat, float* ctx_feat);

/* ---- recommend.BatchPredict / Rank (recommend/rcmd.go:277-337, 248-275) over resident feature tables.
 * A goctr_recsys bundles what GetSampleVector (rcmd.go:462-536) reads per key: the user / item feature tables (the
 * contents of UserFeatureCache / ItemFeatureCache, rcmd.go:474-491; rows indexed by DENSE user / item index), the behaviour
 * cache (may be NULL: the recSys does not implement UserBehavior, rcmd.go:512 => zero behaviours) and the item-embedding
 * table (itemEmbeddingMap; item index == embedding row, a row >= V is "embedding not found" => zeros, rcmd.go:504-507).
 * The cache and the embedding table are borrowed, not owned. */
typedef struct goctr_recsys goctr_recsys;
int goctr_recsys_create(goctr_ubcache* c, goctr_emb* emb, const float* user_table /*[n_users,U]*/, int64_t n_users, int U,
                        const float* item_table /*[n_items,C]*/, int64_t n_items, int C, goctr_recsys** out);
void goctr_recsys_destroy(goctr_recsys* r);
/* BatchPredict (rcmd.go:277-337): n sample keys (user index, item index, timestamp; ts may be NULL = 0) -> scores [n].
 * A key whose user or item has no feature row (index outside the table = GetUserFeature / GetItemFeature error) is
 * scored as the ALL-ZERO row (rcmd.go:299-302) and flagged in failed[i] (may be NULL); *n_failed (may be NULL) counts
 * them.  If the FIRST key fails the call fails like rcmd.go:293-296.  Reference quirk kept for the host mirror: when
 * the LAST key fails, BatchPredict returns y together with a non-nil err (the named result is never cleared,
 * rcmd.go:291,325-336) and Rank then drops the scores (rcmd.go:258-260) -- check failed[n-1]. */
int goctr_batch_predict(goctr_model* m, goctr_recsys* r, const int32_t* users, const int32_t* items, const int64_t* ts,
                        int64_t n, int batch, float* scores, uint8_t* failed, int64_t* n_failed);
/* Rank (rcmd.go:248-275): one user, n candidate items, one timestamp (time.Now().Unix() there) */
int goctr_rank(goctr_model* m, goctr_recsys* r, int32_t user, const int32_t* items, int64_t n, int64_t ts, int batch,
               float* scores, uint8_t* failed, int64_t* n_failed);

/* The replica a multi-device training call (cfg.devices = n) keeps on engine `rank` (rank 0: the handle itself); NULL before
 * the first such call.  Borrowed: owned by the handle it was asked from.  For checks that the replicas are bit-identical
 * (tests, bench.py's replica checksum) -- every entry point works on it, on its own engine. */
int goctr_model_replica(goctr_model* m, int rank, goctr_model** out);
int goctr_emb_replica(goctr_emb* e, int rank, goctr_emb** out);

/* model.Train's epoch loop over a resident dataset (emb == NULL for dense datasets). */
int goctr_train_dataset(goctr_model* m, goctr_emb* emb, goctr_dataset* d, const goctr_train_cfg* cfg,
                        float* epoch_costs, int* epochs_run);
/* exactly n_steps mini-batch steps (forward, backward, all-reduce when a communicator exists,
 * Adam), cycling through the dataset from batch index first_batch; asynchronous -- returns after
 * queueing, call goctr_sync().  costs_dev_to_host may be NULL; otherwise receives n_steps costs
 * (forces a sync).  This is the unit bench.py times. */
int goctr_train_steps(goctr_model* m, goctr_emb* emb, goctr_dataset* d, const goctr_train_cfg* cfg,
                      int64_t first_batch, int n_steps, float* costs);
/* model.Predict over a resident dataset; y_out host [rows] */
int goctr_predict_dataset(goctr_model* m, goctr_emb* emb, goctr_dataset* d, int batch, float* y_out);
/* scores n_batches batches (cycling) and leaves the scores on the device; async. bench QPS unit. */
int goctr_predict_steps(goctr_model* m, goctr_emb* emb, goctr_dataset* d, int batch, int64_t first_batch,
                        int n_batches);

/* --- per-kernel timing (hipEvent, on the engine's stream) for bench.py's roofline object.
 * While enabled the step runs eagerly (no hipGraph) with an event pair around every launch. */
enum { GOCTR_K_ATTN_FWD = 0, GOCTR_K_GEMM_FWD0, GOCTR_K_GEMM_FWD1, GOCTR_K_GEMM_OUT, GOCTR_K_BWD_DZ1,
       GOCTR_K_BWD_DZ0, GOCTR_K_BWD_DP, GOCTR_K_ATTN_BWD, GOCTR_K_DW0, GOCTR_K_DW1, GOCTR_K_DW2,
       GOCTR_K_REDUCE, GOCTR_K_ALLREDUCE, GOCTR_K_ADAM, GOCTR_K_CHAIN, GOCTR_K_EMB_TRAIN, GOCTR_K_EMB_GRAD, GOCTR_K_EMB_PLAN,
       GOCTR_K_COUNT };
int goctr_prof_enable(int on);
int goctr_prof_reset(void);
/* total milliseconds and launch count per kernel family since the last reset */
int goctr_prof_get(int kernel_id, double* total_ms, int64_t* launches);
const char* goctr_prof_name(int kernel_id);
/* symbol (name + template arguments, e.g. "ctr_chain_x3_kernel<9,false>") of the kernel the family's last profiled launch
 * ran; "" when none.  bench.py refuses rocprofv3 counters committed for another kernel than the one it just timed. */
const char* goctr_prof_kernel(int kernel_id);

/* ---------------------------------------------------------------- sklearn-port MLP (f64) --- */
/* replaces nn.NewMLPClassifier + Fit + Predict (nn/neural_network/multilayer_perceptron.go:81-125,
 * basemlp64.go) behind mlp.SimpleMlpFitWrap / SimpleMlpPredWrap (model/mlp/mlp.go:15-65). */
enum { GOCTR_ACT_IDENTITY = 0, GOCTR_ACT_LOGISTIC = 1, GOCTR_ACT_TANH = 2, GOCTR_ACT_RELU = 3 };
enum { GOCTR_SOLVER_SGD = 0, GOCTR_SOLVER_ADAM = 1 };
typedef struct {
  int n_layers;          /* len(layerUnits): input, hidden..., output */
  int units[8];
  int activation;        /* hidden activation (basemlp64.go:79-117) */
  int solver;            /* sgd | adam (basemlp64.go:733-752) */
  double alpha;          /* L2 */
  double lr_init, beta1, beta2, eps, momentum;
  int nesterov;
  int batch_normalize;   /* max-abs scaling (basemlp64.go:277-308) */
  double weight_decay;   /* basemlp64.go:342-346 */
  int batch, max_iter, n_iter_no_change;
  double tol;
} goctr_mlp_cfg;
void goctr_mlp_cfg_default(goctr_mlp_cfg* c); /* NewBaseMultilayerPerceptron64 (basemlp64.go:228-254) */
int goctr_mlp_create(const goctr_mlp_cfg* cfg, goctr_mlp** out);
void goctr_mlp_destroy(goctr_mlp* p);
size_t goctr_mlp_nparams(const goctr_mlp* p);
/* packed parameters [ b_i | W_i ]... (basemlp64.go:432-463) */
int goctr_mlp_set_params(goctr_mlp* p, const double* theta, size_t n);
int goctr_mlp_get_params(goctr_mlp* p, double* theta, size_t n);
/* backprop (basemlp64.go:340-406) on one batch, no update: loss + packed grads (parity entry) */
int goctr_mlp_loss_grad(goctr_mlp* p, const double* X, const double* Y, int n, double* loss, double* grads);
/* fitStochastic (basemlp64.go:729-857) from float32 rows like SimpleMlpFitWrap.Fit widens them
 * (mlp.go:46-59).  perm: [max_iter][rows] row order per epoch (the host owns the shuffle RNG) or
 * NULL = given order.  rows >= batch; when rows is not a multiple of the batch every epoch ends with ONE short batch, as in
 * the reference (basemlp64.go:790-793; main.go:39-50 trains 79 948 rows at 200) and computed its way (quirk Q11, :800-802:
 * the hidden block and the output deltas keep the previous batch's rows beyond the short batch; the intercept means and
 * the loss mean divide by the batch size, the coefficient blocks by the short row count); the epoch loss is
 * sum(batch loss x batch rows) / rows (:806,:812).  loss_curve [max_iter]. */
int goctr_mlp_fit(goctr_mlp* p, const float* X, const float* Y, int64_t rows, const int32_t* perm,
                  double* loss_curve, int* iters_run);
/* the same over the rows goctr_mlp_upload left in HBM (goctr_mlp_fit = goctr_mlp_upload + this): a host that keeps its
 * TrainSample resident across several Fit calls, and the part bench.py times for BASELINE configs[0] */
int goctr_mlp_fit_resident(goctr_mlp* p, const int32_t* perm, double* loss_curve, int* iters_run);
/* exactly n_steps updates cycling over resident rows (async) -- bench unit.
 * goctr_mlp_upload keeps the rows in HBM twice: float32 as given (rows x F x 4 B) and, for the [F, H, 1] shape, widened once to the
 * float64 operand image of the weight-gradient GEMM (rows x round_up(F + 1, 16) x 8 B; skipped above 64 GiB or with GOCTR_MLP_X64=0) */
int goctr_mlp_upload(goctr_mlp* p, const float* X, const float* Y, int64_t rows);
int goctr_mlp_train_steps(goctr_mlp* p, int64_t first_batch, int n_steps);
/* SimpleMlpPredWrap.Predict (mlp.go:15-39): f32 in, probabilities f32 out */
int goctr_mlp_predict(goctr_mlp* p, const float* X, int64_t rows, float* y_out);

/* ---------------------------------------------------------------- item2vec (f64) ----------- */
/* replaces embedding.TrainEmbedding (feature/embedding/wordemb.go:9-32) -> word2vec.Train
 * (model/word2vec/word2vec.go:90-243).  The host keeps the dictionary (string -> id, counts);
 * the device owns param [V,dim], the Huffman inner-node vectors [V-1,dim] / NS ctx matrix and the
 * root-to-leaf paths. */
typedef struct {
  int dim, window;       /* wordemb.go:9 arguments */
  int optimizer;         /* 0 = hierarchical softmax (wordemb.go:13), 1 = negative sampling */
  int model;             /* 0 = skip-gram (wordemb.go:12, model.go:48-78), 1 = cbow (model.go:96-148) */
  int neg_samples;       /* options.go:51 */
  double init_lr, min_lr;     /* options.go:42,49 */
  int64_t update_lr_batch;    /* options.go:55 */
  int max_depth;              /* options.go:46 */
  int deterministic;     /* 1: single stream, bit-exact vs the oracle; 0: Hogwild over `streams` slices */
  int streams;           /* Hogwild workers: lane groups that walk the doc concurrently (the GPU needs ~10^4 of them) */
  int slices;            /* the reference's goroutines (runtime.NumCPU(), options.go:41; default 16): the doc is cut into
                            `slices` by IndexPerThread (modelutil.go:32-41) and windows are clipped at SLICE ends only
                            (quirk Q18); every slice is shared by streams / slices workers.  0 = one slice per worker */
  int devices;           /* 0 / 1: the engine the handle is created on.  n > 1 (= the n of goctr_init_devices): ONE
                            goctr_w2v_upload_doc / goctr_w2v_train(_resident) call runs the pass data-parallel -- the doc is cut into n
                            contiguous shards at the reference's slice boundaries (rank r takes slices [r, r+1) * slices / n), replicas
                            of param / aux on engines 1 .. n-1 are broadcast when they are out of date, every rank trains its shard and
                            the parameter deltas are combined every `exchange_every` words (Hogwild: per row, the average over the ranks
                            that updated the row; deterministic mode: the plain sum p = p0 + sum_r (p_r - p0)) --
                            embedding.TrainEmbedding stays one call from one Go process.  No reference counterpart (SURVEY 2.3, 8(e) item2vec row). */
  int64_t exchange_every; /* data-parallel passes (devices > 1, or one process per GPU after goctr_comm_init): words PER RANK between two
                            all-reduces of the parameter deltas.  0 = update_lr_batch (10^5: SURVEY 8(e), the cadence of the reference's
                            shared observer, word2vec.go:223-233, options.go:55) -- a pass is then ceil(corpus_len / ranks / 10^5)
                            segments, every rank sees the others' updates between segments like the reference's goroutines see
                            each other's through the shared matrices (word2vec.go:198-243); n > 0 = every n words; < 0 = once per
                            pass (rounds 3-4).  Each exchange moves the whole param + aux matrices over xGMI: raise it for
                            large vocabularies. */
} goctr_w2v_cfg;
void goctr_w2v_cfg_default(goctr_w2v_cfg* c);
/* counts [V] = dictionary cfs (dictionary.go:70-81); builds the Huffman tree on the host with the
 * reference's tie-breaking (huffman.go:23-57) and uploads the paths */
int goctr_w2v_create(const goctr_w2v_cfg* cfg, int64_t V, const int64_t* counts, goctr_w2v** out);
void goctr_w2v_destroy(goctr_w2v* w);
/* word2vec.go:103-111 init is host-side RNG: inject it here */
int goctr_w2v_set_param(goctr_w2v* w, const double* param /*[V,dim]*/);
int goctr_w2v_set_aux(goctr_w2v* w, const double* aux /* HS: [V-1,dim]; NS: [V,dim] */);
int goctr_w2v_get_param(goctr_w2v* w, double* param);
int goctr_w2v_get_aux(goctr_w2v* w, double* aux);
int goctr_w2v_get_paths(goctr_w2v* w, int64_t* path_off /*[V+1]*/, int32_t* nodes, uint8_t* codes, int64_t cap,
                        int64_t* total);
/* The Huffman tree alone (dictionary/huffman.go:23-57, node/node.go:39-42 GetPath): root-to-leaf inner-node ids and codes of
 * every word as a CSR, with the reference's tie-breaking (leaves stable-sorted by count, a merged node in front of every
 * node of equal value), built on the HOST in O(V log V) -- no device needed, so also callable without goctr_init.  Call
 * with nodes == codes == NULL to get *total, then again with cap >= *total.  *build_ms (may be NULL): wall time of the build. */
int goctr_huffman_build(const int64_t* counts, int64_t V, int max_depth, int64_t* path_off /*[V+1]*/, int32_t* nodes,
                        uint8_t* codes, int64_t cap, int64_t* total, double* build_ms);
/* one iteration over doc (word2vec.go:151-175): keep_mask = injected sub-sampling trials
 * (subsample.go:45-52) or NULL; corpus_len = unfiltered corpus length (Q17).  lr in/out. */
int goctr_w2v_train(goctr_w2v* w, const int32_t* doc, int64_t n_words, int64_t corpus_len,
                    const uint8_t* keep_mask, double* lr);
/* Host-only helper (no device needed): the word ranges a pass with goctr_w2v_cfg.devices = `devices` gives its ranks --
 * cuts[r] .. cuts[r + 1], r < devices; `slices` as in goctr_w2v_cfg (the reference's IndexPerThread, modelutil.go:32-41). */
int goctr_w2v_shard_cuts(int64_t n_words, int slices, int devices, int64_t* cuts);
/* same over a doc already resident in HBM (bench unit): upload once, then train passes */
int goctr_w2v_upload_doc(goctr_w2v* w, const int32_t* doc, int64_t n_words, const uint8_t* keep_mask);
int goctr_w2v_train_resident(goctr_w2v* w, int64_t corpus_len, double* lr);
/* GenEmbeddingMap32 (word2vec.go:298-324): param rows narrowed to float32 */
int goctr_w2v_export_f32(goctr_w2v* w, float* out /*[V,dim]*/);

/* ---------------------------------------------------------------- corpus / dictionary --------- */
/* replaces memory.New + Corpus.Load (feature/embedding/corpus/memory/memory.go:36-102), dictionary.Add
 * (corpus/dictionary/dictionary.go:70-81) and Corpus.IndexedDoc with the MaxCount / MinCount filters
 * (memory.go:53-62, corpus/cpsutil/cpsutil.go:58-78) for INTEGER tokens: go-ctr's item2vec words are decimal item
 * ids (ItemSeqGenerator, example/movielens/feature.go:78; recommend/rcmd.go:539).  A token's id is its rank by first
 * appearance, cfs[id] its count - exactly the reference's numbering.  INT64_MIN is reserved. */
typedef struct goctr_corpus goctr_corpus;
int goctr_corpus_create(int64_t capacity_words /* < 2^31 */, goctr_corpus** out);
void goctr_corpus_destroy(goctr_corpus* c);
/* one ItemSeqGenerator batch, in stream order (may be called many times before build) */
int goctr_corpus_append(goctr_corpus* c, const int64_t* keys, int64_t n);
/* min_count / max_count: options.go MinCount (default 5) / MaxCount (default -1 = off) */
int goctr_corpus_build(goctr_corpus* c, int64_t min_count, int64_t max_count);
/* n_words = Corpus.Len(), V = Dictionary.Len(), n_indexed = len(IndexedDoc()); any pointer may be NULL */
int goctr_corpus_info(goctr_corpus* c, int64_t* n_words, int64_t* V, int64_t* n_indexed);
int goctr_corpus_get_dictionary(goctr_corpus* c, int64_t* id2key /*[V] or NULL*/, int64_t* cfs /*[V] or NULL*/);
int goctr_corpus_get_doc(goctr_corpus* c, int32_t* idoc /*[n_words] or NULL*/, int32_t* indexed /*[n_indexed] or NULL*/);
/* word2vec.Train's prelude (word2vec.go:90-135) over a built corpus; param / aux still come from set_param / set_aux */
int goctr_w2v_create_from_corpus(const goctr_w2v_cfg* cfg, goctr_corpus* c, goctr_w2v** out);
/* make the corpus' IndexedDoc the resident training doc (device to device) with a fresh subsampling mask
 * (subsample.go:28-52; threshold < 0: no subsampling).  Follow with goctr_w2v_train_resident(w, n_words, &lr). */
int goctr_w2v_use_corpus(goctr_w2v* w, goctr_corpus* c, double subsample_threshold, uint64_t seed);
int goctr_w2v_get_keep_mask(goctr_w2v* w, uint8_t* keep, int64_t n);

/* ---------------------------------------------------------------- embedding k-NN search
 * Replaces search.Searcher (feature/embedding/search/search.go:52-134): brute-force cosine top-k over all items,
 * float64.  Results are bit-identical to the reference loop: the k best by (similarity descending, item index
 * ascending) among similarity > 0, the ignored item skipped (SearchInternal passes the query word, :79). */
typedef struct goctr_searcher goctr_searcher;
/* search.New (:57-63): items [V, D] float64 row-major (emb.Embedding.Vector); the norms (emb.Embedding.Norm =
 * embutil.Norm, embutil.go:21-27) are computed on the device */
int goctr_searcher_create(const double* items, int64_t V, int D, goctr_searcher** out);
void goctr_searcher_destroy(goctr_searcher* s);
/* Searcher.Search (:92-134) for Q queries per call: queries [Q, D]; ignore [Q] = item index to skip or -1 (may be
 * NULL).  out_idx [Q, k] (-1 = the Go zero-value neighbour), out_sim [Q, k]; Rank = position + 1.  out_count [Q] =
 * length of the slice the reference returns, including its guard-loop quirk (:126-131: k - 1 whenever fewer than k
 * items qualify, the surplus entries empty).  k <= 256. */
int goctr_searcher_search(goctr_searcher* s, const double* queries, int Q, int k, const int64_t* ignore,
                          int64_t* out_idx, double* out_sim, int* out_count);

#ifdef __cplusplus
}
#endif
#endif /* GOCTR_H */
