/* gqe.h — C ABI of libgqe.so: the MI355X (gfx950) conjunctive-query embedding hot path.
 *
 * The reference (williamleif/graphqembed, 100 % Python) exposes NO FFI / plugin /
 * operator registry for this path: its boundary is the Python nn.Module
 * netquery.model.QueryEncoderDecoder (model.py:57-127).  This header is the boundary a
 * native replacement sits behind; graphqembed_amd/model.py re-provides the Python class with
 * the reference's signatures on top of it (ctypes), and INTEGRATION.md shows the stub a
 * maintainer of the reference would add.  Each entry point cites what it replaces.
 *
 * Conventions
 *   - plain C types only; every device pointer is BORROWED (never freed by the library);
 *   - all work is enqueued on the caller's hipStream_t (passed as void*); no hidden syncs;
 *   - every function returns 0 on success or a negative gqe_status; gqe_last_error() gives text;
 *   - one gqe_ctx per (process, device); a ctx is not thread-safe, different ctxs are independent.
 *
 * Memory model
 *   All trainable parameters live in ONE flat fp32 arena ("params", P floats):
 *   [tables of every mode | relation vectors or matrices | Pre/Post matrices], each tensor
 *   starting at a 64-float-aligned offset.  grads / exp_avg / exp_avg_sq are arenas of the
 *   same layout.  A "segment" is one parameter tensor (= one state_dict key of the reference).
 *   Node indices are int32 TABLE ROWS (reference: node_maps[mode][node] + 1,
 *   bio/data_utils.py:20-21), vectors are rows of d floats.
 */
#ifndef GQE_H
#define GQE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GQE_ABI_VERSION 1
#define GQE_MAX_BRANCH 3
#define GQE_MAX_HOPS 3
#define GQE_MAX_BATCHES 64
#define GQE_MAX_DIM 256

typedef enum {
  GQE_OK = 0,
  GQE_ERR_ARG = -1,        /* bad argument / unsupported configuration            */
  GQE_ERR_HIP = -2,        /* a HIP runtime call failed                           */
  GQE_ERR_STATE = -3,      /* arena / workspace not bound                         */
  GQE_ERR_WORKSPACE = -5   /* bound workspace too small for this call             */
} gqe_status;

/* --decoder of the reference (utils.py:128-137) */
typedef enum { GQE_DEC_BILINEAR_DIAG = 0, GQE_DEC_TRANSE = 1, GQE_DEC_BILINEAR = 2 } gqe_decoder;
/* --inter_decoder of the reference (utils.py:139-150) */
typedef enum { GQE_INTER_MIN = 0, GQE_INTER_MEAN = 1, GQE_INTER_MIN_SIMPLE = 2, GQE_INTER_MEAN_SIMPLE = 3 } gqe_inter;
/* Formula.query_type (graph.py:13-24) */
typedef enum {
  GQE_Q_1CHAIN = 0, GQE_Q_2CHAIN = 1, GQE_Q_3CHAIN = 2,
  GQE_Q_2INTER = 3, GQE_Q_3INTER = 4, GQE_Q_3INTER_CHAIN = 5, GQE_Q_3CHAIN_INTER = 6
} gqe_qtype;

typedef struct gqe_ctx gqe_ctx;

typedef struct {
  int32_t abi_version;   /* GQE_ABI_VERSION                                        */
  int32_t device;        /* HIP device ordinal                                     */
  int32_t dim;           /* embedding dim d: multiple of 16, <= GQE_MAX_DIM        */
  int32_t decoder;       /* gqe_decoder                                            */
  int32_t inter;         /* gqe_inter                                              */
  int32_t reserved[3];
} gqe_config;

/* One batch = one margin_loss / forward call of the reference: every query shares one
 * Formula (train_helpers.py:100-106), so relation parameters are batch-uniform.
 * All *_param / *_table fields are arena offsets in floats.
 *
 * chain types (1/2/3-chain): n_anchors = 1; hops[0][0..n_hops[0]) are the relations r1..rk
 *   in target->anchor order, applied on the TARGET side (decoders.py:142-147,200-205,228-233).
 * intersection types: branch i is anchor i; hops[i][*] are the already-reversed relations in
 *   the order they are applied to the anchor (model.py:80-91, 102-105); pre/post are the
 *   SetIntersection matrices of the intersection mode (-1 for the *-simple decoders);
 *   final_param is the projection applied after the intersection (3-chain_inter, model.py:107),
 *   -1 otherwise.
 * Index layout (int32, at idx_offset in the idx buffer):
 *   target[B] | negative[B] (margin calls only; absent for gqe_forward) | anchor_0[B] | ... */
typedef struct {
  int32_t qtype;                                  /* gqe_qtype                               */
  int32_t n_queries;                              /* B >= 1                                  */
  int32_t n_anchors;                              /* 1..3                                    */
  int32_t idx_offset;                             /* in int32 elements                       */
  int64_t target_table;
  int64_t anchor_table[GQE_MAX_BRANCH];
  int32_t n_hops[GQE_MAX_BRANCH];
  int32_t n_final;                                /* 0 or 1                                  */
  int64_t hop_param[GQE_MAX_BRANCH][GQE_MAX_HOPS];
  int64_t final_param;
  int64_t pre_param;
  int64_t post_param;
  float margin;                                   /* model.py:112 (default 1)                */
  float loss_weight;                              /* weight of this batch's mean loss in the
                                                     iteration loss (train_helpers.py:51,69-72);
                                                     gradients are scaled by it              */
  int32_t out_offset;                             /* where this batch's scores go            */
  int32_t n_candidates;                           /* gqe_forward only; 0 = score target[B].  > 0 = evaluation
                                                     against candidate lists: the index layout is
                                                     anchor_0[B] | .. | cand_ptr[B+1] | cand_rows[n_candidates]
                                                     and scores[out_offset + c] is the score of candidate c
                                                     (cand_ptr[q] <= c < cand_ptr[q+1]) as the target of query q.
                                                     Full-Bilinear chain queries project the CANDIDATE
                                                     (decoders.py:142-147): their tiles cover 16 candidates each
                                                     and contract [16 x d] . [d x d] per hop on the matrix cores  */
} gqe_batch;

/* One parameter tensor for the optimiser (torch.optim semantics: a tensor with no gradient
 * this iteration is skipped and keeps its own step counter). */
typedef struct {
  int64_t offset;        /* arena offset (floats), multiple of 4                            */
  int64_t numel;
  int32_t step;          /* this tensor's Adam step count AFTER this update (>= 1); <= 0: the
                            library keeps (and increments) the counter for this offset          */
  int32_t reserved;
} gqe_segment;

int gqe_abi_version(void);
/* 1 if gqe_create accepts (decoder, inter, dim): every multiple of 16 in [16, GQE_MAX_DIM] with every decoder / intersection
 * pair (the reference takes any --embed_dim, bio/train.py:13).  No GPU needed.  (Until round 3 the combinations whose guarded
 * kernels spilled registers were refused: full Bilinear outside {16 .. 64, 128, 256}, the MLP intersections at 208 / 224 / 240;
 * DESIGN.md §3 has the history.) */
int gqe_dim_supported(int32_t decoder, int32_t inter, int32_t dim);
const char* gqe_last_error(const gqe_ctx* ctx);   /* ctx may be NULL: last create error */

/* replaces: QueryEncoderDecoder.__init__ + enc_dec.cuda() (model.py:62-68, bio/train.py:56-57) */
int gqe_create(const gqe_config* cfg, gqe_ctx** out);
int gqe_destroy(gqe_ctx* ctx);

/* Bind the parameter / gradient / Adam-moment arenas (device pointers, n floats each).
 * grads, exp_avg, exp_avg_sq may be NULL for inference-only use. */
int gqe_bind_arena(gqe_ctx* ctx, float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n);
/* The library keeps, in its workspace, copies of the d x d matrices the formulas contract with (the intersection's Pre / Post,
 * the relation matrices of the full Bilinear decoder) in the order the matrix cores consume them.  Its own optimiser steps
 * (gqe_adam_step / gqe_sgd_step / gqe_shard_step / the feeder) rewrite the copies together with the parameters.  A caller that
 * writes parameter VALUES into the arena itself — initialisation after the first forward / backward call, a checkpoint load
 * (load_state_dict), an optimiser of its own — calls this before the next gqe_forward / gqe_margin_fwd_bwd: the copies are rebuilt
 * by one small launch in front of it.  (gqe_bind_arena and gqe_bind_workspace imply it.)  No GPU work, no synchronisation.
 * Debugging an integration: with GQE_CHECK_TILES=1 in the environment every forward / backward call first compares the copies
 * with the parameters (synchronising) and fails with an error that names this function if a write was not announced;
 * GQE_ALWAYS_RETILE=1 rebuilds the copies in front of every call. */
int gqe_params_changed(gqe_ctx* ctx);
/* Training loops that call gqe_margin_fwd_bwd and gqe_adam_step back to back, and read losses[] only behind the step, may let
 * the deferred matrix-gradient launch (Pre / Post / Bilinear relation matrices: dM += L^T R over the batch) wait for the
 * optimiser: with enable != 0 gqe_margin_fwd_bwd launches only the fused kernel; the matrix-gradient units and the block that
 * turns the per-tile hinge sums into losses[] run in FRONT of the next gqe_adam_step's chunks, in the same launch (the pass over
 * the tables does not depend on them and is HBM-bound, the units are a short latency chain), and the d x d matrices are stepped
 * by a small second launch.  (Over tables beyond the Infinity Cache — p + m + v above 192 MB, the non-temporal pass — the pair GEMM
 * keeps its own launch; with GQE_RIDE_SPREAD=1 in the environment every 33rd workgroup of that pass is a unit instead: faster by 2 %
 * on most runs, slower by 6-9 % on one in three, hence off by default.)
 * Any other call that needs the gradients first (gqe_materialize_grads, gqe_sgd_step,
 * gqe_zero_grads, another gqe_margin_fwd_bwd / gqe_forward, lazy or order-independent passes) launches the deferred work on its
 * own, as without the switch.  THE CONTRACT: losses[] (and the dense gradient of the matrices) of a gqe_margin_fwd_bwd call are
 * defined once the next such call has been enqueued on the same stream — not right behind gqe_margin_fwd_bwd — and the losses
 * buffer has to stay allocated until then.  Results are the
 * same sums in another atomic order (float atomics: not bit-reproducible either way).  Off by default; one GPU, replicated
 * parameters (the row-sharded step exchanges the gradients between the two launches). */
int gqe_set_deferred_gemm(gqe_ctx* ctx, int32_t enable);
int64_t gqe_deferred_gemm_rides(gqe_ctx* ctx);   /* Adam passes that carried a deferred launch so far (diagnostics, tests) */

/* Declare which arena tensors are embedding tables (offset in floats, number of rows of d floats).
 * Row gradients of tables are kept as per-row contribution lists (one 4-byte atomic per row instead
 * of d float atomics) that the optimiser pass consumes directly; must precede gqe_workspace_bytes. */
int gqe_set_tables(gqe_ctx* ctx, const int64_t* offsets, const int64_t* rows, int32_t n_tables);

/* Declare a registered table as a BAG mode: an index into that mode is a bag i whose vector is the mean of the
 * table rows bag_ids[bag_ptr[i] .. bag_ptr[i+1]) (device pointers, borrowed) — replaces the reference's
 * nn.EmbeddingBag feature function for Reddit posts (reddit/data_utils_new.py:155,162-169).  max_len bounds the
 * bag length (sizes the link nodes of the gradient lists).  Must precede gqe_workspace_bytes. */
int gqe_set_bag(gqe_ctx* ctx, int64_t table_offset, const int32_t* bag_ptr, const int32_t* bag_ids, int64_t n_bags, int32_t max_len);

/* Capacities that size the workspace (optional; must precede gqe_workspace_bytes).  max_tensors: distinct parameter
 * tensors the optimiser entry points may ever be asked to step (default 256; a Bio-scale schema has dozens of relation
 * types: tables + relation tensors + 2 matrices per mode).  Passes over <= 96 known tensors with <= 32 distinct Adam
 * step counts are described in the kernel arguments; larger ones upload a list of the active tensors with the step.
 * max_formulas: size of the device-resident cache of formula descriptors (default 2048, >= GQE_MAX_BATCHES); the
 * reference draws a Formula per batch from train_queries[type] (train_helpers.py:96-100) and a multi-relational graph has
 * thousands of distinct 3-hop formulas, so least-recently-used descriptors are replaced once the cache is full.
 * 0 keeps a value. */
int gqe_set_limits(gqe_ctx* ctx, int32_t max_tensors, int32_t max_formulas);

/* Workspace the kernels need for up to `max_queries` queries / `max_batches` batches between two
 * optimiser steps (bytes); gqe_bind_workspace binds a buffer of at least that size (256-byte aligned)
 * and resets the gradient lists on `stream`.  The capacities given here are remembered by the ctx. */
int64_t gqe_workspace_bytes(gqe_ctx* ctx, int64_t max_queries, int32_t max_batches);
int gqe_bind_workspace(gqe_ctx* ctx, void* workspace, int64_t bytes, void* stream);

/* replaces: QueryEncoderDecoder.forward (model.py:70-109) for n_batches formulas at once.
 * idx: int32 index buffer (device pointer if idx_on_device, else host pointer: copied through
 * the ctx's pinned staging ring with hipMemcpyAsync).  scores: device, sum of B floats (or of
 * n_candidates for evaluation batches, which also replace the inner loops of eval_auc_queries /
 * eval_perc_queries, utils.py:35-91: the query side is computed once per query). */
int gqe_forward(gqe_ctx* ctx, const gqe_batch* batches, int32_t n_batches,
                const int32_t* idx, int64_t n_idx, int32_t idx_on_device,
                float* scores, void* stream);

/* Ranking statistics of an evaluation on the device, so that only query-level numbers are read back.
 * gqe_rank_candidates: replaces _get_perc_scores (utils.py:26-33, scipy.stats.percentileofscore kind 'rank'): for each of
 *   n_queries lists (cand_ptr[n_queries + 1], device) the percentile of the list's FIRST score among the others —
 *   eval_perc_queries puts the true target first (utils.py:86-88).
 * gqe_auc_pair_counts: replaces roc_auc_score (utils.py:63,66) for a set of positive and negative scores:
 *   *count2 += sum_i sum_j (2 [pos_i > neg_j] + [pos_i == neg_j]); AUC = count2 / (2 n_pos n_neg); NaN scores read as 0
 *   (np.nan_to_num).  count2: device, zeroed by the caller. */
int gqe_rank_candidates(gqe_ctx* ctx, const float* scores, const int32_t* cand_ptr, int32_t n_queries, double* percentile, void* stream);
int gqe_auc_pair_counts(gqe_ctx* ctx, const float* pos, int64_t n_pos, const float* neg, int64_t n_neg, uint64_t* count2, void* stream);

/* replaces: margin_loss forward (model.py:112-127) + loss.backward() (train_helpers.py:78)
 * for n_batches (formula, query-slice) pairs in ONE grouped launch.  Gradients of
 * sum_i loss_weight_i * loss_i are ACCUMULATED: relation / Pre / Post gradients into the bound grads arena,
 * embedding-row gradients into the tables' contribution lists (see gqe_set_tables).
 * losses: device, n_batches + 1 floats (mean hinge loss per batch, then the weighted sum).
 * pos_scores / neg_scores: device, optional (NULL to skip). */
int gqe_margin_fwd_bwd(gqe_ctx* ctx, const gqe_batch* batches, int32_t n_batches,
                       const int32_t* idx, int64_t n_idx, int32_t idx_on_device,
                       float* losses, float* pos_scores, float* neg_scores, void* stream);

/* Fold the pending per-row gradient lists of the tables into the dense gradient arena (+=), for callers
 * that need a dense gradient: torch.optim compatibility (param.grad), the data-parallel all-reduce, tests.
 * The next optimiser pass then also reads (and re-zeroes) the dense table gradient. */
int gqe_materialize_grads(gqe_ctx* ctx, void* stream);
/* ... the same for the listed tables only (row-sharded mode: the replicated bag tables, whose dense gradient is then
 * all-reduced while the sharded tables keep their lists). */
int gqe_materialize_tables(gqe_ctx* ctx, const int64_t* table_offsets, int32_t n_tables, void* stream);

/* ---- lazy (deferred, bit-exact) Adam ---------------------------------------------------------------------
 * torch.optim.Adam on a dense embedding gradient moves EVERY row each step — a row without a gradient still
 * follows its decaying momentum — which is why gqe_adam_step streams 24 B per parameter per iteration.  Those
 * zero-gradient steps depend only on the row's own (p, m, v) and the step number, so they can be replayed later
 * with exactly the same arithmetic: in lazy mode gqe_adam_step updates only the rows the pending margin call
 * touched and records, per row, the step count it is current for; gqe_forward / gqe_margin_fwd_bwd first replay
 * the missing steps of the rows they are about to read (and only those), and a full pass runs per table at least
 * every 32 steps (the bound on a row's debt; the per-table ring of bias corrections holds the last 64 steps).  The parameters every kernel reads, and the arena after
 * gqe_optimizer_sync, are bit-identical to the eager schedule (tests/test_gpu_parity.py::test_lazy_adam_*).
 * Works with gqe_set_exchange (the row launch then walks the gathered slabs; replicas stay bit-identical); tables of
 * bag modes are stepped in full every iteration, next to the sparse launch for the other tables.
 *
 * gqe_set_lazy_adam(ctx, 1)     switch on (any time no gradients are pending); 0 switches off (sync first)
 * gqe_optimizer_sync(ctx, st)   bring every row up to date — before the caller reads or writes the parameter /
 *                               moment arenas directly (checkpoints, state_dict, tests)
 * gqe_lazy_prefetch(ctx, ..)      optional, between a margin call and its optimiser step: names the DEVICE-resident
 *                               index feed of the NEXT gqe_margin_fwd_bwd / gqe_forward call (same batch layout rules).
 *                               The step's row launch then also brings that feed's rows up to date and the next call — if
 *                               it passes the same device pointer, size and batch layout (and the same kind of call: with / without
 *                               negatives) — skips its own catch-up launch: one launch
 *                               over rows(t) U rows(t+1) instead of two.  The feed's CONTENTS must not change in between (the
 *                               library cannot see that: a buffer rewritten in place is the caller's responsibility); the
 *                               declaration holds for one optimiser step.  Results are unchanged (bit-identical). */
int gqe_set_lazy_adam(gqe_ctx* ctx, int32_t enable);
int gqe_optimizer_sync(gqe_ctx* ctx, void* stream);
int gqe_lazy_prefetch(gqe_ctx* ctx, const gqe_batch* batches, int32_t n_batches, const int32_t* idx, int64_t n_idx, int32_t with_negatives);

/* ---- data-parallel gradient exchange (the reference is single-process) -------------------
 * Replicas exchange the gradients in the form the fused kernel produces them — contribution entries (dim floats
 * + the list head they belong to) — instead of the dense P-float table gradients.  The workspace's entry space is
 * `world` slabs of S entries (dim floats each); slab k = [ n contributions of rank k | their int32 list heads |
 * rank k's dense relation / Pre / Post gradients ].  Rank r's fused kernel writes its contributions straight into
 * slab r; gqe_export_entries packs the two tails; the host moves the slabs with ONE in-place all-gather (RCCL or
 * any transport); gqe_import_entries links the other ranks' entries into the local per-row lists and replaces
 * the dense gradients by the sum over the slabs in rank order.  The optimiser pass then sums every list
 * order-independently (integer accumulation), so all replicas round identically and stay bit-equal.
 *
 *   gqe_set_exchange(ctx, rank, world)      once, before gqe_workspace_bytes (sizes the entry space x world)
 *   [gqe_exchange_reserve(ctx, n)]          contributions per slab when ranks may produce different counts
 *   gqe_margin_fwd_bwd(...)                 exactly one per optimiser step in this mode
 *   gqe_export_entries(ctx, &S, &off, st)   all-gather float[world][S][dim] at workspace + off,
 *                                           this rank's part being [rank*S, (rank+1)*S)
 *   gqe_import_entries(ctx, S, stream)
 *   gqe_adam_step / gqe_sgd_step
 */
int gqe_set_exchange(gqe_ctx* ctx, int32_t rank, int32_t world);
int gqe_exchange_reserve(gqe_ctx* ctx, int64_t n_contributions);
int gqe_export_entries(gqe_ctx* ctx, int64_t* slab_entries, int64_t* contrib_offset, void* stream);
int gqe_import_entries(gqe_ctx* ctx, int64_t slab_entries, void* stream);

/* ---- row-sharded data parallelism: "owner computes" (sharded optimiser; the reference is single-process)
 * Instead of replicating the tables, each of `world` ranks OWNS the rows r with r % world == rank of every embedding
 * table (local row r / world) together with their Adam moments, and keeps a replica of the small relation / Pre / Post
 * tensors only.  An iteration on a rank:
 *
 *   gqe_shard_plan(...)        host only: sort the step's index feed by owner -> the rows to REQUEST from each owner
 *                              (list-head indices in the owner's shard), and the POSITION feed that replaces the index
 *                              feed (where each fetched row will sit)
 *   [all-to-all: requests]     requests of all ranks for my rows -> workspace + req_recv
 *   gqe_shard_serve(...)       gather the requested rows of my shards                   -> workspace + rows_send
 *   [all-to-all: rows]         -> workspace + fetched, grouped by owner in request order = the position feed's order
 *   gqe_margin_fwd_bwd(idx = position feed, device)   rows are read from `fetched`; the gradient contribution of the
 *                              row at position p is written to workspace + contrib_send at p; nothing is linked here
 *   [all-to-all: contributions]  -> the owners' workspace + contrib_recv, in the order of the requests they received
 *   gqe_shard_link(...)        link the received contributions onto my rows' gradient lists
 *   [sum of the ranks' relation / Pre / Post gradients: the non-table spans of the gradient arena — an all-reduce in the
 *    phase API; gqe_shard_step moves every rank's copy to every peer INSIDE the contributions' exchange (further sends /
 *    receives of the same ncclGroup, or a second all_to_all of a callback transport) and sums them in rank order: two
 *    collectives per step, the same bits on every rank.  GQE_SHARD_DENSE_ALLREDUCE=1 keeps the all-reduce]
 *   gqe_adam_step(local segments)   the ordinary fused pass over MY shards (24 B per OWNED parameter) + the small tensors
 *
 * Per rank and step the optimiser streams 1/world of the tables and the inbound traffic is (rows + contributions of one
 * rank's batch) instead of growing with world as the all-gather of gqe_set_exchange does.  An owner sums a row's list
 * order-independently (integer accumulation: the result does not depend on the order contributions arrive in), the
 * replicated tensors see the same all-reduced gradient on every rank and stay bit-identical, and the step equals the
 * single-rank step on the concatenated batch up to fp32 summation order.  The transport is the caller's (torch.distributed all_to_all_single over
 * RCCL in graphqembed_amd/parallel.py); the library only names the buffers.  Tables must be registered with their LOCAL
 * row counts, ceil(global rows / world), the same on every rank.  Bag (EmbeddingBag) tables are NOT sharded: a bag table
 * (gqe_set_bag; Reddit: the 50 k-word table behind the posts) stays replicated with its full row count, an index into
 * it is a bag id that gqe_shard_plan passes through, its rows are gathered from the local replica and its gradient is
 * linked onto the local lists; before the optimiser step the host folds those lists into the dense gradient
 * (gqe_materialize_tables), all-reduces that table's span of the gradient arena and steps it like the other replicated
 * tensors.  Candidate lists are not available in this mode; lazy Adam only through gqe_shard_step (below).
 * Driving these phases by hand, EVERY RANK MUST RUN THE SAME FORMULAS in a step: gqe_shard_link marks as pending the tables
 * this rank's own margin call named, so a contribution a peer sends for another table would not be consumed. */
typedef struct {
  int64_t req_send, req_recv;         /* byte offsets in the workspace: int32 requests I send / receive            */
  int64_t rows_send, fetched;         /* float rows I serve / rows I fetched (dim floats each)                      */
  int64_t contrib_send, contrib_recv; /* float contributions of my batch / contributions for my rows               */
  int64_t cap_send, cap_recv;         /* capacities in entries (rows): one rank's feed / what one owner can receive */
} gqe_shard_buffers;
int gqe_set_shard(gqe_ctx* ctx, int32_t rank, int32_t world);   /* before gqe_workspace_bytes; world = 1: every row is this rank's */
/* Sum every row's gradient list order-independently (integer accumulation for lists longer than two entries): results no
 * longer depend on the order in which atomics linked the contributions, i.e. runs are bit-reproducible.  Always on in
 * gqe_set_exchange mode (replicas must round identically); optional elsewhere — a row-sharded row has one owner, so ranks
 * agree without it — and off by default: it costs the optimiser pass ~14 % (53.9 vs 47.1 us on bio-synth). */
int gqe_set_ordered_sums(gqe_ctx* ctx, int32_t enable);
/* Hot rows — what replaces the reference's dense `index_add` backward of nn.Embedding / nn.EmbeddingBag (bio/data_utils.py:17,
 * reddit/data_utils_new.py:155) on HEAVY-TAILED data.  A row's gradient contributions normally hang on a per-row list that one
 * lane group walks (fine for the few entries a row of a sparse graph collects per step).  The optimiser pass measures the lists
 * it walks; a row whose list reaches 24 entries in one step (a hub node, a frequent word) is promoted: from the next step on its
 * contributions are added into 32 dense accumulators of dim floats with float atomics, and the pass sums those instead of
 * chasing hundreds or thousands of links.  Automatic, up to 2048 rows per ctx; off in gqe_set_exchange mode and with
 * gqe_set_ordered_sums (atomic sums are order-dependent); GQE_HOT=0 in the environment disables it, GQE_HOT_MIN_LEN=n changes
 * the promotion threshold.  A row promoted on a list of fewer than 512 entries (a hub node) keeps to 8 of its 32 accumulators: the
 * lane group that steps it reads them two at a time, and four dependent round trips instead of sixteen matter on the critical chain of
 * a split step's second launch (GQE_HOT_FEW_LEN=n moves that, 0 = never).  gqe_hot_rows: how many rows have been promoted so far
 * (synchronises the device). */
int gqe_hot_rows(gqe_ctx* ctx, int32_t* n_hot);
/* Hot WORD rows (contexts with bag tables).  A frequent word collects thousands of contributions per step, one per bag that holds
 * it; an atomic row for each is what made the fused launch of reddit-synth with Zipf(1) words 1.3-1.5 x the uniform one.  A
 * promoted row therefore also gets 2^k sub-lists out of a pool of 65536 list heads (k from the list length that promoted it: about
 * a dozen nodes per sub-list and step): the fused kernel links a bag's node onto one of them — the 4-byte exchange every other word
 * costs — and a gather launch behind it sums every sub-list with a wave of its own into the row's accumulators.  The fused launches
 * switch to sub-lists when the host has seen a promotion (a word of pinned memory the promoting kernel sets: no synchronisation);
 * GQE_HOT_SUB=0 in the environment keeps the atomic rows.  gqe_hot_sub_lists: heads handed out so far, and whether the fused
 * launches link onto them yet (synchronises the device). */
int gqe_hot_sub_lists(gqe_ctx* ctx, int32_t* n_heads, int32_t* active);
int gqe_shard_layout(gqe_ctx* ctx, gqe_shard_buffers* out);     /* after gqe_bind_workspace */
/* idx: HOST index feed of GLOBAL table rows laid out as gqe_batch describes (with_negatives: margin layout).  Outputs
 * (host): positions[n_idx] — the feed to hand to gqe_margin_fwd_bwd / gqe_forward (device copy); requests[n_idx] — grouped
 * by owner, send_counts[world] of them per owner. */
int gqe_shard_plan(gqe_ctx* ctx, const gqe_batch* batches, int32_t n_batches, const int32_t* idx, int64_t n_idx, int32_t with_negatives,
                   int32_t* positions, int32_t* requests, int64_t* send_counts);
int gqe_shard_serve(gqe_ctx* ctx, const int32_t* requests, int64_t n, float* rows_out, void* stream);   /* device pointers */
int gqe_shard_link(gqe_ctx* ctx, const int32_t* requests, int64_t n, void* stream);

/* ---- the row-sharded step as ONE call (what a trainer uses; the phase entry points above are its building blocks) ----------
 * gqe_shard_plan / serve / link driven by hand require every rank to run the SAME formulas in a step (an owner marks as
 * pending only the tables its own batches name) and leave planning and transport to the caller.  The session entry points
 * below take both over:
 *   - planning stays on the host cores and uses no collective: each rank sorts its feed by owner and POSTS the counts, the
 *     request lists and the parameter tensors its batches touch on a plan board in POSIX shared memory ("/gqe_<session>",
 *     the ranks of ONE node); owners read their requests from the board; the kernels read the position feed and the
 *     received requests from pinned host memory (no staging copy, no side stream);
 *   - data moves on the caller's stream over RCCL (ncclSend / ncclRecv groups and ncclAllReduce of librccl, bound at run
 *     time; `nccl_comm` is an ncclComm_t created by the caller) or over the gqe_transport callbacks; world = 1 needs neither;
 *   - the optimiser steps the UNION of the tensors the ranks' batches touch (library-kept per-tensor Adam step counts), so
 *     the replicated relation / Pre / Post tensors stay bit-identical whatever formulas the ranks drew;
 *   - lazy Adam (gqe_set_lazy_adam) works: an owner replays the deferred steps of the rows it is asked for before it serves
 *     them and steps only the rows it received contributions for (a full pass over its shard every 32 steps).
 * Two plans may be posted ahead: a trainer posts step t + 1 before it runs step t, so the only host-side wait of a step
 * (until every peer has posted that step) is hidden behind the previous step.
 *
 *   gqe_shard_open(ctx, session, comm, NULL)                once, after gqe_bind_workspace (every rank: same capacities)
 *   gqe_shard_post(ctx, batches, n, idx, n_idx, 1, segs, n_segs)   plan a margin step: idx = HOST feed of GLOBAL rows (it has
 *                                                           to stay valid until the step has run: the owner sort happens on
 *                                                           the session's planning thread, next to the caller's thread),
 *                                                           segs = the tensors its batches touch (gqe_segment.step unused)
 *   gqe_shard_step(ctx, lr, b1, b2, eps, losses, pos, neg, stream) run the oldest posted plan: serve (+ link) the other ranks'
 *                                                           requests -> all-to-all of rows -> fused forward / backward + pair
 *                                                           GEMM on fetched rows and, in place, on the rows of the own shard ->
 *                                                           all-to-all of contributions -> all-reduce of the small gradients ->
 *                                                           Adam on the own shards
 *   gqe_shard_post(.., 0, NULL, 0) + gqe_shard_forward(ctx, scores, stream)   the same for gqe_forward, candidate lists included:
 *                                                           the list offsets pass through the plan, every candidate is fetched from
 *                                                           its owner like any other row (once per naming: the whole index feed of
 *                                                           a call has to fit the fetched-row buffer, i.e. size the workspace with
 *                                                           max_queries >= n_idx / 5)
 *   gqe_shard_close(ctx)                                    (gqe_destroy closes an open session)
 * Every rank must issue the same sequence of post / step / forward calls. */
typedef struct {
  void* user;
  /* blocks of send_counts[p] / recv_counts[p] elements of elem_bytes, contiguous in peer order on both sides (device pointers),
   * stream-ordered on `stream`; return 0 on success */
  int (*all_to_all)(void* user, const void* send, const int64_t* send_counts, void* recv, const int64_t* recv_counts,
                    int64_t elem_bytes, void* stream);
  int (*all_reduce_sum_f32)(void* user, float* buf, int64_t n, void* stream);
  /* non-zero: all_to_all leaves the CALLER'S OWN block (block `rank` of both buffers) untouched.  The library then keeps that
   * block in place, as it does on its RCCL path: in a margin step the fused kernel reads the rows of the rank's own shard
   * where they live and links their contributions itself (nothing is served or copied for them); in a forward call they are
   * served straight into the fetched-row buffer.  0: the transport moves every block, the own one included. */
  int32_t skips_own_block;
} gqe_transport;
int gqe_shard_open(gqe_ctx* ctx, const char* session, void* nccl_comm, const gqe_transport* transport);
int gqe_shard_close(gqe_ctx* ctx);
/* Host time of the open session, accumulated since gqe_shard_open when GQE_SHARD_PROFILE is set in the environment (else zeros):
 * us[0..1] = the PLANNING thread (owner sort, publication), us[2..9] = the caller's thread inside gqe_shard_step / _forward
 * (collect, serve launch, rows exchange, fused + GEMM launches, contributions exchange, link + all-reduces, optimiser, event);
 * sums over *steps steps.  What a host loop spends beyond us[2..9] per step is time it waited for the GPU (the ring of pinned
 * feeds is 8 steps deep), not work. */
int gqe_shard_profile(gqe_ctx* ctx, double us[10], int64_t* steps);
int gqe_shard_post(gqe_ctx* ctx, const gqe_batch* batches, int32_t n_batches, const int32_t* idx, int64_t n_idx, int32_t with_negatives,
                   const gqe_segment* segs, int32_t n_segs);
int gqe_shard_step(gqe_ctx* ctx, float lr, float beta1, float beta2, float eps, float* losses, float* pos_scores, float* neg_scores,
                   void* stream);
int gqe_shard_forward(gqe_ctx* ctx, float* scores, void* stream);

/* The dense exchange named by the north star, as one call: fold the row-gradient lists into the dense gradient arena
 * (gqe_materialize_grads) and sum the arena over the ranks of `nccl_comm` (an ncclComm_t of RCCL, created by the caller:
 * ncclGetUniqueId / ncclCommInitRank) with ncclAllReduce on `stream`.  Sits between gqe_margin_fwd_bwd and the
 * optimiser step (train_helpers.py:78-79).  librccl.so is loaded on first use (dlopen); replicated tables only. */
int gqe_allreduce_grads(gqe_ctx* ctx, void* nccl_comm, void* stream);

/* replaces: optimizer.step() + optimizer.zero_grad() for torch.optim.Adam
 * (bio/train.py:62, train_helpers.py:50,79): one fused pass p,g,m,v -> p,m,v and g := 0
 * over the listed segments only. */
int gqe_adam_step(gqe_ctx* ctx, const gqe_segment* segs, int32_t n_segs,
                  float lr, float beta1, float beta2, float eps, void* stream);
/* The reference's decoder / encoder EXTENSION POINTS on [d, B] tensors — what model.py composes a query out of, for callers that
 * score one hop or one intersection on their own (the fused path above never calls them).  Device pointers; element (i, b) of an
 * embedding batch at i * B + b (a contiguous torch [d, B]); forward only; the decoder kind / aggregation are the ctx's.
 *   gqe_encode_rows        replaces DirectEncoder.forward(nodes, mode) (encoders.py:40-43): out[:, b] = the L2-normalised row
 *                          rows[b] of the table at table_offset (a bag table: the mean of the bag's word rows, normalised)
 *   gqe_decoder_project    replaces path_dec.project(embeds, rel) (decoders.py:149-150, 207-208, 235-236): M_rel . e | e + w | e * w
 *   gqe_decoder_forward    replaces path_dec.forward(embeds1, embeds2, rels) (decoders.py:142-147, 200-205, 228-233): the chain
 *                          rels[0 .. n_rels) applied to embeds1, then cos(., embeds2) — bilinear-diag: the plain dot product
 *   gqe_set_intersection   replaces inter_dec(embeds1, embeds2, mode, embeds3) (decoders.py:288-300, 311-319): Post . agg_i
 *                          relu(Pre . e_i); pre_param = post_param = -1 for the Simple intersection; embeds3 may be NULL */
int gqe_encode_rows(gqe_ctx* ctx, int64_t table_offset, const int32_t* rows, int32_t B, float* out, void* stream);
int gqe_decoder_project(gqe_ctx* ctx, int64_t rel_param, const float* embeds, int32_t B, float* out, void* stream);
int gqe_decoder_forward(gqe_ctx* ctx, const int64_t* rel_params, int32_t n_rels, const float* embeds1, const float* embeds2, int32_t B,
                        float* scores, void* stream);
int gqe_set_intersection(gqe_ctx* ctx, int64_t pre_param, int64_t post_param, const float* embeds1, const float* embeds2,
                         const float* embeds3, int32_t B, float* out, void* stream);

/* replaces: one whole iteration of run_train's loop body — optimizer.zero_grad(); loss = run_batch(...) [model.margin_loss];
 * loss.backward(); optimizer.step() (train_helpers.py:76-79 with torch.optim.Adam, bio/train.py:62) — as ONE call:
 * gqe_margin_fwd_bwd(batches, idx, losses) followed by gqe_adam_step(segs, lr, beta1, beta2, eps), same arguments, same result.
 * Knowing the optimiser step when the forward / backward is enqueued lets the library order the step's work by what depends on
 * what instead of by call ("split step"): a dense Adam step moves every row of every stepped table, the batches
 * name 15-20 % of them, and the update of a row the batches do not name depends on nothing the forward / backward produces.  The
 * step becomes
 *     launch M  Adam on the d x d matrices of the PREVIOUS gqe_train_step  +  a stamp on every table row this step's feed names
 *     launch A  the fused forward / backward tiles  |  Adam (zero gradient) over every row without a stamp, in the same launch
 *     launch B  loss finalize + matrix-gradient units  |  Adam over the stamped rows (gradient lists / hot accumulators)  |  the
 *               relation vectors
 * — every element gets exactly the arithmetic of gqe_adam_step (the oracle tests run through both).
 * losses[] is defined when the call's work has completed on `stream`, as for gqe_margin_fwd_bwd.
 * THE CONTRACT that differs from the two calls: the Adam step of the d x d matrices (Pre / Post, full-Bilinear relation matrices)
 * is enqueued by the NEXT gqe_train_step, or by whatever other entry point of the library comes first (every one of them
 * settles it: gqe_forward, gqe_margin_fwd_bwd, the optimiser calls, gqe_materialize_grads, gqe_optimizer_sync, ...).  A caller
 * that reads the parameter or moment arenas itself — a checkpoint, an evaluation of its own — calls gqe_optimizer_sync first.
 * Where the split does not apply the call runs the two-call sequence (with the matrix-gradient units riding in the Adam pass as
 * under gqe_set_deferred_gemm when that applies): lazy Adam, gqe_set_exchange / gqe_set_shard, ordered sums, tables
 * far beyond the Infinity Cache (p + m + v of the stepped tables above 384 MB), gradients already pending from an earlier gqe_margin_fwd_bwd, dims whose kernels are not the
 * straight-line ones (d % 64 != 0), more than GQE_LAUNCH_BATCHES batches or 8 stepped tables, feeds of more than 65 535 entries (a row's stamp names its owning feed entry in 16 bits).  GQE_SPLIT=0 in the environment
 * forces that sequence.  gqe_split_steps: how many calls ran as split steps (diagnostics, tests). */
int gqe_train_step(gqe_ctx* ctx, const gqe_batch* batches, int32_t n_batches, const int32_t* idx, int64_t n_idx, int32_t idx_on_device,
                   const gqe_segment* segs, int32_t n_segs, float lr, float beta1, float beta2, float eps, float* losses, void* stream);
int64_t gqe_split_steps(gqe_ctx* ctx);
/* The library's Adam step counter of the tensor at `offset` (what a segment with step <= 0 continues from; a caller's explicit step
 * count is committed to it): how a caller that keeps its own counters — torch.optim's state["step"] — catches up behind a native
 * run (gqe_feeder_run). */
int gqe_adam_step_count(gqe_ctx* ctx, int64_t offset, int32_t* count);
/* ... and the other direction: the library's counter of the tensor at `offset` := `count` (0 forgets it).  What restoring an optimiser
 * checkpoint needs (torch.optim.Adam.load_state_dict restores state["step"], bio/train.py:62 + torch.save / torch.load): the runs
 * that use the library's counters — gqe_feeder_run, segments with step <= 0 — continue the bias correction from the restored count
 * instead of restarting at 1 on warmed-up moments.  GQE_ERR_STATE while a split step's matrix update is still pending
 * (gqe_optimizer_sync first). */
int gqe_set_adam_step_count(gqe_ctx* ctx, int64_t offset, int32_t count);
/* replaces: torch.optim.SGD(momentum=0).step() + zero_grad (bio/train.py:60) */
int gqe_sgd_step(gqe_ctx* ctx, const gqe_segment* segs, int32_t n_segs, float lr, void* stream);
/* replaces: optimizer.zero_grad() alone */
int gqe_zero_grads(gqe_ctx* ctx, const gqe_segment* segs, int32_t n_segs, void* stream);

/* ---- native training feed: replaces run_train's per-iteration Python (train_helpers.py:48-79,
 * 95-107 + model.py:113-120) once the queries are tensorised.  Pools are HOST int32 row arrays (copied):
 * target[n], anchors[k][n], one stored negative and (intersections) one hard negative per query, as the
 * reference's training files carry them (data_utils.py:71 neg_sample_max=1); `formula` supplies the static
 * fields of gqe_batch.  gqe_feeder_run performs n_iterations of: draw a formula per batch (probability
 * proportional to pool size), slice by the reference's wrap-around rule, draw 1-chain negatives uniformly from
 * the rows given by gqe_feeder_set_mode_rows, pack, gqe_margin_fwd_bwd, gqe_adam_step on the touched tensors.
 * losses (device, >= GQE_MAX_BATCHES+1 floats) holds the last iteration's losses. */
typedef struct gqe_feeder gqe_feeder;
int gqe_feeder_create(gqe_ctx* ctx, uint64_t seed, int32_t batch_size, float path_weight, float inter_weight, gqe_feeder** out);
int gqe_feeder_destroy(gqe_feeder* f);
int gqe_feeder_add_pool(gqe_feeder* f, const gqe_batch* formula, int64_t n, const int32_t* target, const int32_t* anchors,
                        const int32_t* neg, const int32_t* hard);
int gqe_feeder_set_mode_rows(gqe_feeder* f, int64_t table_offset, const int32_t* rows, int64_t n);
/* ---- reference streams: run_train's own loop (train_helpers.py:40-107) executed natively, batch for batch ----
 * The reference draws the formula of a batch with np.random.multinomial(1, sizes / sum) (train_helpers.py:96-99) and one negative
 * per query with random.choice — of graph.full_lists[mode] for 1-chain queries, of the query's neg_samples / hard_neg_samples
 * otherwise (model.py:113-120).  With gqe_feeder_set_reference_streams the feeder replays exactly those draws on the caller's two
 * generators: np_state625 / py_state625 = 624 key words + position of np.random.get_state() / random.getstate(), updated in place
 * (the caller hands them back with set_state / setstate after a run; nothing else may draw in between).  Pools then carry every
 * query's negative LISTS (gqe_feeder_add_pool_lists: CSR over table rows; hard lists optional, 1-chain pools none),
 * gqe_feeder_set_pvals the probability vector of each query type as the caller computed it (float64, one per pool in the
 * order the pools were added), gqe_feeder_set_type_order the query types behind 1-chain in the order of the caller's training
 * dictionary (train_helpers.py:63-71).  A run seeded like the reference's then trains on the reference's batches
 * (graphqembed_amd/train_helpers.py, tests/test_gpu_api.py).  gqe_feeder_set_loss_stride(s > 0): iteration i of a run writes
 * its losses at losses + (i - first_iteration) * s — the loss history the loop's moving average and log lines are made of.  NULL
 * states switch back to the feeder's own generator. */
int gqe_feeder_add_pool_lists(gqe_feeder* f, const gqe_batch* formula, int64_t n, const int32_t* target, const int32_t* anchors,
                              const int64_t* neg_ptr, const int32_t* neg_rows, const int64_t* hard_ptr, const int32_t* hard_rows);
int gqe_feeder_set_reference_streams(gqe_feeder* f, uint32_t* np_state625, uint32_t* py_state625);
int gqe_feeder_set_pvals(gqe_feeder* f, int32_t qtype, const double* pvals, int32_t n);
int gqe_feeder_set_type_order(gqe_feeder* f, const int32_t* qtypes, int32_t n);
int gqe_feeder_set_loss_stride(gqe_feeder* f, int64_t stride);
/* --opt sgd (bio/train.py:59-60, torch.optim.SGD with momentum 0): the feeder's iterations close with gqe_sgd_step(lr) instead of
 * the Adam step (gqe_feeder_run's betas / eps are then ignored). */
int gqe_feeder_set_sgd(gqe_feeder* f, int32_t enable);
/* Queries of every batch the feeder has packed so far (throughput accounting: the windows at a list's end are shorter). */
int64_t gqe_feeder_queries(gqe_feeder* f);
/* Host wall time this feeder has spent so far sampling + packing feeds (build_s) and inside gqe_feeder_run altogether (run_s: the
 * former + enqueueing uploads and launches + waiting for ring slots): what the host cores' share of a fed iteration is (bench.py). */
int gqe_feeder_host_seconds(gqe_feeder* f, double* build_s, double* run_s);
/* Debug / tests: the batches and the packed index feed (target | negative | anchors per batch) of one of the last prepared
 * iterations, as the kernels were given them; a NULL / too small output only reports the sizes. */
int gqe_feeder_debug_feed(gqe_feeder* f, int64_t iteration, gqe_batch* batches, int32_t max_batches, int32_t* n_batches, int32_t* idx,
                          int64_t max_idx, int64_t* n_idx);
/* How an iteration's index feed reaches the kernels: 0 = pinned staging ring + hipMemcpyAsync on the library's upload
 * stream (as gqe_margin_fwd_bwd does for any host feed); 1 (default) = the kernels read the feed straight from pinned host
 * memory (16 slots, an event every 8 iterations guards their re-use).  In mode 0 the feeds of eight iterations are sampled together
 * and travel with ONE copy and one pair of cross-stream events. */
int gqe_feeder_set_feed(gqe_feeder* f, int32_t mode);
int gqe_feeder_run(gqe_feeder* f, int64_t first_iteration, int32_t n_iterations, int32_t burn_in, float lr, float beta1,
                   float beta2, float eps, float* losses, void* stream);

/* Timing of the most recent launches of each kernel on the stream they ran on, measured with
 * hipEvents recorded by the library when enabled (bench.py's roofline block uses this:
 * torch.cuda.Event cannot see a raw hipStream).  kernel: 0 = fused fwd/bwd, 1 = param-grad
 * GEMM, 2 = optimiser (table pass, or the sparse row launch in lazy mode), 3 = lazy mode: small dense tensors,
 * 4 = lazy mode: catch-up launch before a read.  Returns the average milliseconds over the recorded launches and
 * their count, then clears the record.  Row-sharded step (gqe_shard_step): 5 = serve + exchange of rows, 6 = exchange of
 * contributions + link + all-reduces of the replicated gradients. */
int gqe_timing_enable(gqe_ctx* ctx, int32_t stride);   /* record every stride-th launch; 0 = off */
/* Debug: when `stamps` (device, 64 int64 per workgroup: the tiles of the next fused launch, then the workgroups of
 * its pair-GEMM launch) is non-NULL the kernels record wall_clock64() (100 MHz) at their phase boundaries
 * (tools/kbench.py decodes them); NULL switches it off. */
int gqe_debug_profile(gqe_ctx* ctx, long long* stamps);
/* Debug: which instantiation gqe_fused_kernel<DEC, MLP, NC, FULL, BWD, FW> a launch of `tiles` tiles runs for (decoder, dim):
 * nc_full_fw[3] = {NC, FULL, FW}.  tests/test_build_meta.py checks every selectable instantiation of the built library. */
int gqe_debug_fused_variant(int32_t decoder, int32_t dim, int32_t tiles, int32_t* nc_full_fw);
int gqe_timing_read(gqe_ctx* ctx, int32_t kernel, float* avg_ms, int32_t* count);

#ifdef __cplusplus
}
#endif
#endif /* GQE_H */
