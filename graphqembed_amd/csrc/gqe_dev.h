// gqe_dev.h — structs shared by the host side (gqe_host.cpp) and the kernels (gqe_kernels.hip).
#ifndef GQE_DEV_H
#define GQE_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gqe.h"

#define GQE_TQ 16            // queries per tile (= per workgroup of the fused kernel)
#define GQE_FWAVES 16        // most wave64s per workgroup of the fused kernel (one query row per wave; the 8-wave shape: two)
#define GQE_FW8_MIN_TILES 512  // d = 128: launches with more tiles than this use the 8-wave shape (two workgroups per CU)
#define GQE_WAVES 4          // wave64s per workgroup of the pair-GEMM / optimiser kernels
#define GQE_THREADS 256
#define GQE_OPT_CHUNK 1024   // floats per optimiser chunk (256 threads x float4)
#define GQE_MAX_SEGS 96      // tensors the kernel-argument form of an optimiser pass can describe (more: table form)
#define GQE_GEMM_KCHUNK 128  // queries per pair-GEMM chunk (a unit walks 1, 2, 4 or 8 chunks: GqeDynPlan.pad[0])
#define GQE_GEMM_MIN_UNITS 600  // ... as many as leave the launch this many units (measured: B = 512 -> 1, 2048 -> 2, 4096 -> 4, 8192 -> 8)
#define GQE_GEMM_MT 64       // edge of the gradient block one pair-GEMM unit produces
#define GQE_PROF_SLOTS 64     // wall_clock64 stamps per workgroup (debug profile)

#define GQE_LAUNCH_BATCHES 16  // batches per fused launch: their dynamic descriptors travel as kernel arguments
#define GQE_DEFAULT_FORMULAS 2048  // default capacity of the device formula-descriptor cache (gqe_set_limits); LRU beyond
#define GQE_NT_STREAM_BYTES (192ll << 20)  // optimiser pass: p + m + v of the stepped tables above this -> non-temporal policy
#define GQE_DEFAULT_TENSORS 256    // default number of distinct parameter tensors the optimiser can be asked to step
#define GQE_MAX_JOBS 8         // deferred matrix-gradient jobs one batch can generate
#define GQE_MAX_BAGS 4         // tables whose rows are bags (nn.EmbeddingBag modes)

// Static part of a batch: everything that depends only on the Formula (and the parameter layout).
// Lives in a device-resident table that grows when a new formula is seen; never re-uploaded otherwise.
struct GqeDevFormula {
  int32_t qtype, n_anchors, n_final, n_slots;
  int32_t n_hops[GQE_MAX_BRANCH];
  int32_t n_jobs;
  int64_t target_table;
  int64_t anchor_table[GQE_MAX_BRANCH];
  int64_t hop_param[GQE_MAX_BRANCH][GQE_MAX_HOPS];
  int64_t final_param, pre_param, post_param;
  int64_t target_head;                    // index of the target table's row 0 in the gradient-list heads
  int64_t anchor_head[GQE_MAX_BRANCH];
  int32_t target_bag;                     // >= 0: the target mode is a bag mode (index into GqeBagTable), else -1
  int32_t anchor_bag[GQE_MAX_BRANCH];
  // deferred dM += L^T R jobs: parameter + the two scratch slots.  Every field is at least 32 bits wide: the kernels read
  // the descriptor with scalar loads, and a byte field would be fetched with a VECTOR load followed by s_waitcnt vmcnt(0)
  // — a full memory round trip behind everything the wave has in flight, once per field (round 2: eight of them sat on
  // an intersection tile's critical path)
  int64_t job_param[GQE_MAX_JOBS];
  int32_t job_L[GQE_MAX_JOBS], job_R[GQE_MAX_JOBS];
  // scratch slots (row blocks of Bpad x d floats); -1 = unused
  int32_t slot_x[GQE_MAX_BRANCH][GQE_MAX_HOPS];   // bilinear: input of hop h of branch i
  int32_t slot_gy[GQE_MAX_BRANCH][GQE_MAX_HOPS];  // bilinear: grad wrt output of hop h of branch i
  int32_t slot_e[GQE_MAX_BRANCH];                 // MLP: e_i (input of Pre)
  int32_t slot_gz[GQE_MAX_BRANCH];                // MLP: grad wrt Pre.e_i
  int32_t slot_hh, slot_gq;                       // MLP: h (input of Post), grad wrt q
  int32_t slot_fx, slot_fg;                       // bilinear final projection: input, grad wrt output
  int32_t slot_act[2][GQE_MAX_HOPS];              // bilinear chain: act_h of the +/- side
  int32_t slot_gact[2][GQE_MAX_HOPS];             // bilinear chain: grad wrt act_{h+1}
  // Operand-ordered copies of the d x d matrices this formula contracts with (GQE_TILE_INDEX below): float offsets into the
  // WORKSPACE of the copy of M; the copy of M^T sits tile_t floats behind it.  -1: not a matrix (vectors of bilinear-diag / TransE)
  int64_t hop_tile[GQE_MAX_BRANCH][GQE_MAX_HOPS];
  int64_t final_tile, pre_tile, post_tile, tile_t;
};

// ---- operand-ordered ("tiled") copies of the d x d matrices ---------------------------------------------------------
// The A operand of v_mfma_f32_16x16x4_f32 for output rows i0 .. i0+15 and k-block kb is, per lane l = (i & 15) + 16 * ((k & 15) >> 2),
// the four floats M[i][16 kb + 4 (l >> 4) .. + 3].  Read from the row-major matrix that is 16 B per lane at a stride of one matrix
// row — 16 half-used cache lines per wave load, and for M^T four 4-byte loads per lane; at d = 256 the Post contraction took
// 8.1 us against a 3.4 us MFMA pipe bound with one tile on the whole chip (the L1's line rate, not L2 contention).  The library
// therefore keeps, for every matrix a formula names, a copy of M and one of M^T in exactly that order — tile (i >> 4, k >> 4) is
// 1 KB, a wave load reads it as ONE contiguous kilobyte:
//     float index of M[i][k] = (((i >> 4) * (d >> 4) + (k >> 4)) * 64 + (i & 15) + 16 * ((k & 15) >> 2)) * 4 + (k & 3)
// The copies are rewritten by the optimiser pass together with the matrix (gqe_opt_kernel, dense segments) and rebuilt by
// gqe_retile_kernel when the caller changed parameters itself (gqe_params_changed) or a formula names a new matrix.
#define GQE_TILE_INDEX(i, k, d) ((((size_t)((i) >> 4) * ((d) >> 4) + ((k) >> 4)) * 64 + ((i) & 15) + 16 * (((k) & 15) >> 2)) * 4 + ((k) & 3))
#define GQE_RETILE_MAX 64   // matrices per launch of the rebuild kernel (kernel arguments)
struct GqeRetileArgs {
  int n;
  long long tile_t;                    // copy of M^T = copy of M + tile_t floats
  long long param[GQE_RETILE_MAX];     // float offset of the matrix in the parameter arena
  long long tile[GQE_RETILE_MAX];      // float offset of its copy in the workspace
};
hipError_t gqe_launch_retile(const GqeRetileArgs& a, const float* params, float* ws, int d, hipStream_t stream);
// GQE_CHECK_TILES=1 (debug): counts the elements of the listed matrices whose copy of M differs from the parameter (bit compare)
hipError_t gqe_launch_tilecheck(const GqeRetileArgs& a, const float* params, const float* ws, int d, int32_t* mismatches, hipStream_t stream);

// Dynamic part of a batch (sizes, offsets of this call), passed BY VALUE in the kernel arguments:
// no plan upload, no event between the host and the launch.
struct GqeDynBatch {
  int32_t formula, B, Bpad, idx_offset;
  int32_t out_offset, tile_begin, has_neg, unit_begin;  // unit_begin: first pair-GEMM unit of this batch
  int64_t entry_base;    // first contribution entry of this batch: [role][query]
  int64_t scratch_base;  // float offset of this batch's scratch rows in the workspace
  float margin, grad_scale, inv_B, loss_weight;
  int32_t loss_index;    // where this batch's loss goes in the caller's losses[]
  int32_t n_candidates;  // > 0: evaluation against candidate lists (forward only): scratch_base = this batch's query
                         // records, unit_begin = its first block of the candidate-scoring kernel
  int32_t n_anchors;     // copy of the formula's anchor count: the tile's index load needs it, and as a kernel argument
                         // it does not wait for the descriptor's first scalar load
  int32_t expand;        // != 0: candidate lists of a full-Bilinear CHAIN batch.  The projection t^T M_r1 .. M_rk runs on the
                         // candidate side (decoders.py:142-147), so there is no per-query vector to score candidates against:
                         // the tiles of this batch cover its CANDIDATES, 16 per tile ([16 x d] . [d x d] on the matrix cores per
                         // hop), target row = the candidate, anchor row = the anchor of the candidate's query; the int32 query of
                         // every candidate sits at ws + scratch_base (gqe_expand_ptr_kernel writes it from cand_ptr)
};

// Bag modes (Reddit posts: nn.EmbeddingBag mean over word rows, reddit/data_utils_new.py:155,162-169):
// CSR of row ids per bag, borrowed device pointers, passed by value with the launch.
// entry -> list head map (exchange): >= 0 a list head, -1 not pushed, <= -2 a bag: -(2 + (slot << 27 | bag index))
#define GQE_BAG_CODE(slot, bag) (-(2 + (((slot) << 27) | (bag))))

struct GqeBagTable {
  const int32_t* ptr[GQE_MAX_BAGS];
  const int32_t* ids[GQE_MAX_BAGS];
  int32_t max_len;   // longest bag of any bag table: the link nodes of contribution entry e are e * max_len + (word position)
  int32_t pad;
};

// Hot rows (heavy-tailed graphs: hub nodes, frequent words).  A row's gradient list is walked by ONE lane group, one dependent
// 4-byte load per entry (~0.27 us each): fine for the handful of entries a row of a uniform graph collects, hopeless for a hub
// that collects hundreds per step (a Zipfian word: thousands).  The optimiser pass measures every list it walks; a row whose
// list reaches `min_len` entries is PROMOTED: it gets a slot of GQE_HOT_REPS dense accumulators of d floats, and from the next
// step on its contributions are ADDED there with fire-and-forget float atomics (replica = the pusher's XCD and wave, so that
// same-address atomics — 24 ns each wherever they come from — spread over GQE_HOT_REPS chains) instead of being written as
// entries and linked.  The pass that steps the row sums the replicas and re-zeroes them.  slot[] is indexed like head[].
// Off (slot == NULL) wherever sums have to be order-independent: the replicated exchange mode, gqe_set_ordered_sums.
// Replicas of a slot are ADJACENT in memory (acc[slot][replica][d]): with acc[replica][slot][d] a slot's replicas lay 2 MB
// apart — a power of two, the same channel — and 8 replicas bought nothing; adjacent, 8 / 16 / 32 replicas take the fused kernel
// of reddit-synth Zipf from 184 us to 126 / 121 / 113 us (64: no further gain; profiles/r04_experiment_hot_layout.log).
#ifndef GQE_HOT_REPS
#define GQE_HOT_REPS 32
#endif
#define GQE_HOT_SLOTS 2048
#define GQE_HOT_MIN_LEN 24
#define GQE_HOT_ROW(rep, slot) ((size_t)(slot) * GQE_HOT_REPS + (rep))
// Hot WORD rows (bag tables).  A frequent word of an nn.EmbeddingBag table collects thousands of contributions per step, one per
// bag that holds it: an atomic row (d floats) for each of them is 125 k x 1 KB of atomic traffic per step on reddit-synth with
// Zipf(1) words — 1.3-1.5 x the fused launch of the uniform data set.  Such a row therefore also gets 2^lg SUB-LISTS (a power of two
// sized from the list length that promoted it: ~GQE_HOT_SUB_LEN entries per sub-list and step) out of a pool.  A sub-list is an
// ARRAY of GQE_HOT_SUB_CAP contribution entries behind a counter: the fused kernel takes a position with one 4-byte atomic add
// (at the end of the kernel, where nothing waits for its round trip) and stores the bag's entry there; what does not fit goes onto
// an overflow chain of ordinary link nodes.  gqe_hot_gather_kernel — launched behind the fused kernel — gives every sub-list a wave:
// one load for the counter, one for the entries, the rows behind them all in flight, ONE atomic row into the word's accumulators.
// (Linked sub-lists were the first form: a chain of a dozen dependent loads per list made the gather 54 us — EXPERIMENTS.md 101.)
// The consumers see what they saw before: accumulators.  slot[] value of a hot row = slot | (lg + 1) << 11 | first sub-list << 15;
// (lg + 1) == 0: no sub-lists (the pool was full) — contributions are added directly.  Plain (non-bag) roles always add directly.
#define GQE_HOT_SUB_POOL 32768
#define GQE_HOT_SUB_CAP 128
#define GQE_HOT_SUB_LEN 32
#define GQE_HOT_SUB_MAX_LG 10
#define GQE_HOT_SLOT_OF(v) ((v) & (GQE_HOT_SLOTS - 1))
#define GQE_HOT_SUB_LG1(v) (((v) >> 11) & 15)
#define GQE_HOT_SUB_BASE(v) (((v) >> 15) & (GQE_HOT_SUB_POOL - 1))
// the pool (GqeHot.sub): counters, overflow-chain heads, the accumulator slot of each sub-list, the entry arrays
#define GQE_HOT_SUB_CNT(sub) (sub)
#define GQE_HOT_SUB_OVF(sub) ((sub) + GQE_HOT_SUB_POOL)
#define GQE_HOT_SUB_SLOT(sub) ((sub) + 2 * GQE_HOT_SUB_POOL)
#define GQE_HOT_SUB_BUF(sub) ((sub) + 3 * GQE_HOT_SUB_POOL)
#define GQE_HOT_SUB_INTS ((size_t)(3 + GQE_HOT_SUB_CAP) * GQE_HOT_SUB_POOL)
// what a lane keeps for push_links instead of a previous list head when its word goes to sub-list i: -2 - i
#define GQE_HOT_SUB_TAG(i) (-2 - (i))
// a row with sub-lists is added to by the gather's waves, one atomic row per sub-list and step — GQE_HOT_SUB_REPS of the
// GQE_HOT_REPS accumulators spread that; the pass that steps the row reads and re-zeroes only those (direct adds to such a row — a
// producer that has no pool yet — keep to them too)
#define GQE_HOT_SUB_REPS 8
// ... and so does a row promoted on a list of fewer than GQE_HOT_FEW_LEN entries (bit 30 of its slot[] value): a hub node's few
// hundred contributions per step need no 32-way spread (36 same-address atomic rows of 24 ns per accumulator), and the lane group
// that steps the row reads its accumulators two at a time — 16 dependent round trips for 32 of them, on the critical chain of a
// split step's second launch
#define GQE_HOT_FEW_LEN 512
#define GQE_HOT_FEW_BIT (1 << 30)
#define GQE_HOT_REPS_OF(v) ((GQE_HOT_SUB_LG1(v) || ((v) & GQE_HOT_FEW_BIT)) ? GQE_HOT_SUB_REPS : GQE_HOT_REPS)
// Row-sharded margin steps run by a session (gqe_shard_step): an index of the position feed that is >= GQE_OWN_ROW names row
// (index - GQE_OWN_ROW) of this rank's OWN shard of the role's table — the fused kernel reads it where it lives and links its
// contribution itself, exactly as in the unsharded step; smaller indices are positions in the fetched-row buffer.
#define GQE_OWN_ROW (1 << 30)

struct GqeHot {
  int32_t* slot;     // [total rows]: -1 = not hot, else the row's accumulator slot
  float* acc;        // [cap][GQE_HOT_REPS][d]
  int32_t* count;    // slots handed out so far (may run past cap: rows promoted beyond it stay on lists)
  int32_t cap, min_len;
  int32_t few_len;   // a row promoted on a shorter list uses GQE_HOT_SUB_REPS accumulators (GQE_HOT_FEW_LEN; the environment may change it)
  // sub-lists of hot word rows (above).  sub == NULL in the struct a PRODUCER gets: it adds directly whatever slot[] says (the
  // host has not seen a promotion yet — `seen` is a word of pinned host memory a promoting kernel sets — or the mode has no
  // gather launch); consumers always get the pointers
  int32_t* sub;        // the pool (GQE_HOT_SUB_CNT / OVF / SLOT / BUF), handed out in blocks of 2^lg sub-lists
  int32_t* sub_count;  // sub-lists handed out so far (may run past the pool: later promotions get none)
  int32_t* seen;       // pinned host memory, or NULL
};

struct GqeDynPlan {
  int32_t n_batches, tiles, units, first;  // first != 0: this launch starts the weighted total (=), else +=
  int32_t total_index, pad[3];             // pad[0]: chunks per pair-GEMM unit; pad[1]: write-through stores (vstore_wt)
  int32_t tile_begin[GQE_LAUNCH_BATCHES];  // contiguous copies for the "which batch am I" scan (one wide
  int32_t unit_begin[GQE_LAUNCH_BATCHES];  // scalar load); entries >= n_batches hold INT_MAX
  GqeDynBatch b[GQE_LAUNCH_BATCHES];
};

// One parameter tensor the optimiser has ever been asked to step ("universe" table, device resident,
// re-uploaded only when a new tensor shows up).  Which tensors a given pass touches, and with which Adam
// bias corrections, travels in the kernel arguments (GqeOptActive / GqeStepCoef): no per-step upload.
struct GqeDevSeg {
  int64_t offset, numel, n_chunks;
  int64_t rows, head_base;  // tables only
  int32_t is_table;
  int32_t table_index;      // tables only: index in gqe_set_tables order (slot of the lazy-Adam per-table arguments)
  float *tile, *tile_T;     // d x d tensors: their operand-ordered copies (GQE_TILE_INDEX), rewritten with the parameter; else NULL
};

// Table form of "which tensors does this pass step, and with which Adam bias corrections": one entry per ACTIVE
// tensor in universe order, uploaded with the step.  Used when the kernel-argument form (GqeOptActive /
// GqeStepCoef) cannot describe the pass: more than GQE_MAX_SEGS tensors, or more than GQE_MAX_STEP_GROUPS distinct
// per-tensor step counts (schemas with dozens of relation types whose counters diverge).
struct GqeActSeg {
  long long chunk_begin;  // first chunk of this tensor in the pass
  int32_t seg;            // universe index
  float step_size, bc2_sqrt;
  int32_t pad;            // != 0: a table whose dense gradient is live (read it, and re-zero it) next to its lists
};

#define GQE_GROUP_DENSE 0x40    // flag in GqeOptActive::group / GqeActSeg::dense: the table's dense gradient is live too
struct GqeOptActive {
  uint8_t group[GQE_MAX_SEGS];  // per universe entry: 0xFF = not stepped, else index into GqeStepCoef (| GQE_GROUP_DENSE)
  // chunk prefix of the pass over the universe (inactive tensors own no chunks), computed by the host: a workgroup used
  // to rebuild it — a global load of every tensor's chunk count, then one thread adding them up, ~3 us before its first
  // useful load, and at the start of the launch every resident workgroup did so at once
  int32_t begin[GQE_MAX_SEGS + 1];
};

#define GQE_MAX_STEP_GROUPS 32
struct GqeStepCoef {
  float step_size[GQE_MAX_STEP_GROUPS];  // lr / (1 - b1^t)
  float bc2_sqrt[GQE_MAX_STEP_GROUPS];   // sqrt(1 - b2^t)
};

// ---- lazy rows (gqe_set_lazy_adam; see gqe_kernels.hip) ----
#define GQE_LAZY_RING 64        // per-table ring of (step_size, bc2_sqrt) of the last 64 Adam steps
#define GQE_LAZY_PERIOD 32      // a full pass at least this often per table: bounds how many steps a row can owe
#define GQE_LAZY_TABLES 8
#define GQE_LAZY_SEGS 96        // index-feed segments one rows launch can cover
struct GqeLazyTabs {
  int n;  // tables in gqe_set_tables order (head_base ascending)
  long long offset[GQE_LAZY_TABLES], head_base[GQE_LAZY_TABLES];
  int target[GQE_LAZY_TABLES];     // bring rows to this Adam step count of their table
  int grad_step[GQE_LAZY_TABLES];  // the step that consumes the row's gradient list (== target), or -1: replay only
  float step_size[GQE_LAZY_TABLES], bc2_sqrt[GQE_LAZY_TABLES];  // coefficients of grad_step
  unsigned char eager[GQE_LAZY_TABLES];  // bag-mode tables: always stepped in full, no per-row counts (every row is current)
};
struct GqeLazyArgs {      // rides along with the optimiser launch
  int32_t* last;          // [total rows] step count each row is current for
  float2* ring;           // [GQE_LAZY_TABLES][GQE_LAZY_RING]
  GqeLazyTabs t;          // slot = GqeDevSeg::table_index
};
struct GqeRowSegs {       // the table rows named by an index feed: segment k = idx[idx_begin[k] .. +count) of table tid[k]
                          // (tid -1: skip; -2: the values are list heads of any table — exchanged slabs)
  int n, total;
  int begin[GQE_LAZY_SEGS + 1];  // prefix sums of the counts
  long long idx_begin[GQE_LAZY_SEGS];  // int32 offset from the launch's idx pointer (64-bit: a launch may span two feeds)
  int8_t tid[GQE_LAZY_SEGS];
};
struct GqeRowSegs32 {     // the same with 32-bit feed offsets: the form the row launch takes when it also carries riding pair-GEMM
                          // units (their plan is 1.4 KB of kernel arguments; the launch has to stay inside the 4 KB it may pass)
  int n, total;
  int begin[GQE_LAZY_SEGS + 1];
  int idx_begin[GQE_LAZY_SEGS];
  int8_t tid[GQE_LAZY_SEGS];
};
struct GqeRowsArgs {
  GqeRowSegs segs;
  GqeLazyTabs t;
  const int32_t* idx;
  int32_t* last;
  float2* ring;
  float *p, *g, *m, *v;
  int32_t* head;
  const int32_t* next;
  const float* contrib;
  int32_t max_entries;
  int d;
  float lr, b1, b2, eps;
  bool with_grad, sorted;
  GqeHot hot;
  // the step's small dense tensors ride in extra workgroups of the same launch (dense_chunks == 0: none)
  const GqeDevSeg* dsegs;
  int n_dsegs;
  long long dense_chunks;
  GqeStepCoef dcoef;
  GqeOptActive dactive;
  const GqeActSeg* dact;  // table form of dcoef / dactive (NULL: kernel-argument form)
  int n_dact;
  hipStream_t stream;
};

#define GQE_OPT_ADAM 0
#define GQE_OPT_SGD 1
#define GQE_OPT_ZERO 2
#define GQE_OPT_MATERIALIZE 3

struct GqeOptArgs {
  int mode;
  bool lists;         // some table has pending gradient lists
  bool sorted;        // sum each list in ascending node id (bit-identical across data-parallel replicas)
  bool dense_tables;  // the dense gradient of SOME stepped table has to be read (and re-zeroed) too: the per-table flags say which
  const GqeDevSeg* segs;
  int n_segs;
  long long total_chunks;
  float *p, *g, *m, *v;
  int32_t* head;
  const int32_t* next;
  const float* contrib;
  const int32_t* link_contrib;
  int32_t max_entries;
  int d;
  float lr, b1, b2, eps;
  GqeStepCoef coef;
  GqeOptActive active;
  const GqeActSeg* act;  // table form of coef / active (NULL: kernel-argument form)
  int n_act;
  bool lazy;          // tables carry per-row step counts (GqeLazyArgs)
  bool nt;            // stream p / m / v of the tables with the non-temporal policy (tables larger than the Infinity Cache)
  GqeLazyArgs lz;
  GqeHot hot;
  hipStream_t stream;
};

// ---- the split step (gqe_train_step; gqe_split.h) --------------------------------------------------------------------
// Adam over the rows a step's index feed does NOT name rides in the fused launch ("rider" workgroups behind the tiles), the
// named rows are stepped by the launch that also carries the pair-GEMM units, the d x d matrices by the small launch in front
// of the NEXT fused launch (which also stamps that step's named rows).
#define GQE_SPLIT_TABLES 8
#ifndef GQE_SPLIT_WROWS
#define GQE_SPLIT_WROWS 8       // consecutive rows of one table a rider wave owns at a time ("wave block": at d = 128 one batch of 4 x 2 row slices)
#endif
struct GqeSplitTabs {
  int n;                                   // stepped tables
  int blk_begin[GQE_SPLIT_TABLES + 1];     // prefix of their wave blocks
  long long offset[GQE_SPLIT_TABLES], head_base[GQE_SPLIT_TABLES], rows[GQE_SPLIT_TABLES];
  float step_size[GQE_SPLIT_TABLES], bc2_sqrt[GQE_SPLIT_TABLES];   // lr / (1 - b1^t), sqrt(1 - b2^t) of the table's step count
};
#define GQE_SPLIT_PWAVES 16     // progress slots per rider (its waves)
#define GQE_SPLIT_MAX_RIDERS 1024
#define GQE_SPLIT_SEGS 64       // index-feed segments of one fused launch: <= GQE_LAUNCH_BATCHES x (target | negative, <= 3 anchors)
struct GqeSplitSegs {           // the rows the step's feed names: segment k = idx[idx_begin[k] .. +count) of stepped table tid[k]
  int n, total;
  int begin[GQE_SPLIT_SEGS + 1];     // prefix sums of the counts
  int idx_begin[GQE_SPLIT_SEGS];     // int32 offset from the launch's idx pointer
  int8_t tid[GQE_SPLIT_SEGS];        // slot in GqeSplitTabs (-1: skip)
};
#define GQE_SPLIT_MAX_ENTRIES 65535   // feed entries of a split step: the owner of a row is named in 16 bits of its stamp
#define GQE_SPLIT_MAX_EPOCH 32767     // ... under a 15-bit epoch

struct GqeSplitRide {
  GqeSplitTabs t;
  float *p, *m, *v;
  const int32_t* stamp;   // [total rows], indexed like head[]: epoch << 16 | e: named by this step's feed, owned by its entry e (which
  int epoch;              // steps the row in the second launch); anything else: an older step's — not named.  epoch += 1 per step (15 bits)
  float b1, b2, eps;
  int blocks;             // rider workgroups of the launch (0: a plain fused launch) ...
  int lead;               // ... of which this many come FIRST in the grid (they start with the launch, on CUs of their own); the
                          // tiles follow, then the other riders (they start where tiles have finished)
  int waves;              // waves per rider workgroup of the fused launch (16 or 8)
  int per, share;         // wave blocks per tail rider; a lead rider owns share * per (split_range, gqe_split.h)
  int32_t* progress;      // [blocks][GQE_SPLIT_PWAVES]: the next wave block of (rider, wave) — set to its first block by the launch
                          // in front (gqe_prestep_kernel), left by the rider when the tiles are through, continued by the second launch
  int32_t* done;          // tiles of the fused launch that have finished (zeroed by the launch in front)
  int tiles;
  int stop;               // != 0: the riders stop when the tiles are through (else they finish their ranges)
};

// everything one fused launch needs (built by gqe_host.cpp, consumed by the per-variant launchers)
struct GqeFusedArgs {
  GqeDynPlan plan;
  const GqeDevFormula* formulas;
  const float* params;
  float* grads;
  float* ws;
  const int32_t* idx;
  int d;
  float *tile_loss, *pos, *neg;  // tile_loss: one partial hinge sum per tile (workspace)
  int inter_min;
  bool bwd;
  long long* prof;
  hipStream_t stream;
  int32_t* head;   // gradient lists: head[table row] -> newest node (-1 = none)
  int32_t* next;   // next[node]; node < max_entries: a contribution entry; else a bag link node
  float* contrib;  // contrib[entry][d]
  GqeBagTable bags;
  int32_t* link_contrib;  // link node -> contribution entry it stands for
  int32_t* link_counter;  // bump allocator of link nodes
  int32_t max_entries;
  const float* fetched;   // row-sharded mode: the rows of this call, fetched from their owners (else NULL)
  float* contrib_bag;     // where bag contributions go (= contrib, except in row-sharded mode: the optimiser's entry space)
  long long bag_shift;    // ... and the first entry index they may use there
  GqeHot hot;             // hot rows (slot == NULL: off)
  GqeSplitRide split;     // split.blocks > 0: rider workgroups behind the tiles (gqe_train_step)
  int force_fw;           // 0: the dispatcher picks the workgroup shape from the tile count; 8 / 16: this shape (split steps)
};

// can the fused kernel this launch would select carry rider workgroups (split.blocks > 0)?  (the straight-line d % 64 == 0 kernels)
int gqe_fused_can_ride(int dec, int mlp, int d, int tiles);
hipError_t gqe_launch_fused(int dec, int mlp, const GqeFusedArgs& a);
// the hot word rows' sub-lists summed into their accumulators (GqeHot): behind every backward fused launch that got hot.sub
hipError_t gqe_launch_hot_gather(const GqeHot& hot, const int32_t* next, const float* contrib, const int32_t* link_contrib, int max_entries, int d,
                                 hipStream_t stream);
int gqe_config_supported(int dec, int inter, int d);   // gqe_kernels.hip, next to the dispatcher
void gqe_fused_variant(int dec, int d, int tiles, int* nc, int* full, int* fw);
hipError_t gqe_launch_pair_gemm(const GqeFusedArgs& a, float* losses);
#define GQE_EVAL_BLOCK 512   // candidates one workgroup of the scoring kernel covers (= GQE_EVAL_UB in gqe_kernels.hip)
hipError_t gqe_launch_eval_score(const GqeFusedArgs& a, int dec, float* scores);   // plan.unit_begin / units = candidate blocks
hipError_t gqe_launch_expand_ptr(const int32_t* cand_ptr, int n_queries, int n_candidates, int32_t* query_of, hipStream_t stream);
hipError_t gqe_launch_rank(const float* scores, const int32_t* ptr, int nq, double* percentile, hipStream_t stream);
hipError_t gqe_launch_auc(const float* pos, long long n_pos, const float* neg, long long n_neg, unsigned long long* count2, hipStream_t stream);
hipError_t gqe_launch_opt(const GqeOptArgs& a);

// A deferred pair-GEMM launch riding in FRONT of an Adam pass's chunks in one launch (gqe_set_deferred_gemm, gqe_opt_gemm_kernel):
// workgroup 0 finalizes the losses, workgroups 1 .. plan.units are GEMM units, the pass's chunks follow.  The pass covers the
// tables and the vectors — nothing the units write; the d x d matrices are stepped by a second, small launch behind it.
#define GQE_RIDE_MAX_UNITS 1024   // beyond: the GEMM is MFMA work of its own (B = 8192: 73 us) and keeps its LDS-staged kernel
struct GqeGemmRide {
  GqeDynPlan plan;
  const GqeDevFormula* formulas;
  const float* ws;
  const float* tile_loss;
  float* losses;
  // 0: the units lead the grid (CUs of their own for the units' few microseconds, then the chunks).  K > 0 — next to the long
  // non-temporal pass over tables beyond the Infinity Cache: every K-th workgroup behind the finalize block is a unit, the others
  // stream chunks — the units' MFMA work runs in the shadow of a launch that is bound by HBM for hundreds of microseconds
  int32_t spread;
};
// units that may ride SPREAD through a non-temporal pass (GqeGemmRide.spread): the limit of what one launch carries
#define GQE_RIDE_MAX_UNITS_SPREAD 16384
hipError_t gqe_launch_opt_gemm(const GqeOptArgs& a, const GqeGemmRide& r);
// ... and that second launch: Adam on up to GQE_MATSTEP_MAX d x d matrices named in the kernel arguments (no universe scan, one
// round trip to memory: it sits between the pass and the next fused kernel, on the step's critical path)
#define GQE_MATSTEP_MAX 48
struct GqeMatStep {
  int n;
  long long off[GQE_MATSTEP_MAX];                  // arena offset of the matrix
  float* tile[GQE_MATSTEP_MAX];                    // its operand-ordered copy (the copy of the transpose: + tile_t floats)
  float step_size[GQE_MATSTEP_MAX], bc2_sqrt[GQE_MATSTEP_MAX];
  long long tile_t;
};
hipError_t gqe_launch_matstep(const GqeMatStep& a, float* p, float* g, float* m, float* v, int d, float b1, float b2, float eps, hipStream_t stream);
hipError_t gqe_launch_rows(const GqeRowsArgs& a);
// lazy Adam's row launch (with gradient) carrying a deferred pair GEMM: workgroup 0 finalizes the losses, workgroups 1 .. units are
// the GEMM units, then the row groups, then the chunks of a.dsegs (which must not contain the d x d matrices: they are stepped by
// gqe_launch_matstep behind this launch).  hipErrorInvalidValue if a feed offset does not fit 32 bits.
hipError_t gqe_launch_rows_ride(const GqeRowsArgs& a, const GqeGemmRide& r);
// the split step's launch M: Adam on the pending d x d matrices (ms.n may be 0) + stamp[row] := 1 for the rows `segs` names
hipError_t gqe_launch_prestep(const GqeMatStep& ms, float* p, float* g, float* m, float* v, int d, float b1, float b2, float eps,
                              const GqeSplitSegs& segs, const GqeSplitRide& ride, const int32_t* idx, int32_t* stamp, hipStream_t stream);
// ... and its launch B: loss finalize + pair-GEMM units | the named rows (claim the stamp, list / hot gradient, Adam; coefficients per
// table in `ride.t`) | riders for what the fused launch left of the stream (ride.ticket) | the chunks of `a` (the relation vectors:
// a pass over non-table, non-matrix tensors)
hipError_t gqe_launch_split_rows(const GqeOptArgs& a, const GqeGemmRide& r, const GqeSplitSegs& segs, const GqeSplitRide& ride, const int32_t* idx,
                                 int32_t* stamp);
// the reference's decoder / encoder extension points on [d, B] tensors (gqe_encode_rows / gqe_decoder_project / gqe_decoder_forward /
// gqe_set_intersection, include/gqe.h)
hipError_t gqe_launch_x_encode(const float* table, const int32_t* rows, int B, int d, const int32_t* bag_ptr, const int32_t* bag_ids, float* out,
                               hipStream_t stream);
hipError_t gqe_launch_x_project(int dec, const float* w, const float* e, int B, int d, float* out, hipStream_t stream);
hipError_t gqe_launch_x_forward(int dec, const float* params, const long long* rel_params, int n_rels, const float* e1, const float* e2, int B, int d,
                                float* scores, hipStream_t stream);
hipError_t gqe_launch_x_intersect(const float* pre, const float* post, int agg_min, const float* e1, const float* e2, const float* e3, int B, int d,
                                  float* out, hipStream_t stream);
// non-table floats of the arena (dense gradients that travel with the exchanged slab)
struct GqeSpans {
  int n;  // < 0: more than 8 spans (unsupported)
  long long off[8], len[8], total;
};
// row-sharded step: the relation / Pre / Post gradients of the ranks travel with the contributions (one exchange instead of an
// all-to-all and an all-reduce).  gqe_launch_dense_stage: `world` copies of this rank's dense gradient (the spans, packed) for a
// transport that needs one send block per peer; gqe_launch_dense_sum: grads[span j] = sum over the ranks IN RANK ORDER (this
// rank's own term read from the arena, the others from block p of `recv`, `stride` floats apart): the same bits on every rank.
hipError_t gqe_launch_dense_stage(const GqeSpans& sp, const float* grads, float* send, long long stride, int world, hipStream_t stream);
hipError_t gqe_launch_dense_sum(const GqeSpans& sp, float* grads, const float* recv, long long stride, int rank, int world, hipStream_t stream);
// row-sharded data parallelism: serve rows of the local shards / link received contributions onto the local lists
struct GqeShardTabs {
  int n;
  long long offset[GQE_LAZY_TABLES], head_base[GQE_LAZY_TABLES];  // local tables in gqe_set_tables order
};
hipError_t gqe_launch_shard_serve(const float* params, const int32_t* req, long long n, float* out, int d, const GqeShardTabs& t,
                                  long long own_lo, long long own_n, float* own_out, int32_t* head, int32_t* next, long long own_entry,
                                  int link, hipStream_t stream);
hipError_t gqe_launch_shard_link(int32_t* head, int32_t* next, const int32_t* req, long long n, long long own_lo, long long own_n,
                                 long long own_entry, hipStream_t stream);
hipError_t gqe_launch_export(float* contrib, const int32_t* rows, const float* grads, int d, long long slab_base, int32_t n,
                             const GqeSpans& sp, hipStream_t stream);
struct GqeImportBags {  // bag tables on the importing side: CSR + where the word table's list heads start
  GqeBagTable csr;
  long long head_base[GQE_MAX_BAGS];
  int32_t* link_contrib;
  int32_t* link_counter;
  int32_t max_entries;
};
hipError_t gqe_launch_import(int32_t* head, int32_t* next, const float* contrib, float* grads, int d, long long slab, int32_t n,
                             int rank, int world, const GqeSpans& sp, const GqeImportBags& bags, hipStream_t stream);

#endif
