// gqe_host.cpp — host side of libgqe.so: context, launch planning, pinned staging, C ABI (include/gqe.h).
//
// Per call the host builds a compact "plan" (device batch descriptors, pair-GEMM jobs, tile->batch map,
// optionally the int32 index feed), writes it into a pinned ring slot and ships it with ONE
// hipMemcpyAsync on the caller's stream; the kernels are then enqueued on the same stream.  Nothing
// here synchronises (except re-use of a ring slot whose copy has not finished yet).
//
// Gradient bookkeeping.  Embedding-row gradients are NOT scattered with 128 atomics per row: every
// (query, role) contribution is written once, coalesced, into a contribution buffer and pushed onto
// a per-table-row linked list with ONE 4-byte atomic exchange (head[row], next[entry]).  The optimiser
// pass walks the (mostly empty, length <= a few) lists while it streams p/m/v, so the dense gradient of
// the tables is never written, re-read or re-zeroed.  gqe_materialize_grads() folds the lists into the
// dense gradient arena for callers that need it (torch.optim compatibility, the DP all-reduce, tests).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "gqe_dev.h"
#include "gqe_mt.h"

namespace {

constexpr int kRing = 4;
constexpr int kFeedGroup = 8;      // feeder: iterations whose feeds share one upload / one pair of cross-stream events (copy mode), one guard event (zero-copy)
constexpr int kStageBufs = 2 + 2 * kFeedGroup;   // staged index buffers in the workspace: 2 for host feeds of single calls, 2 groups for the feeder
constexpr int kMaxSlots = 20;      // upper bound of pair-scratch slots any batch can use
constexpr int kRolesPerQuery = 5;  // target, negative, <= 3 anchors

thread_local std::string g_create_error;

struct RingSlot {
  char* host = nullptr;
  size_t cap = 0;
  hipEvent_t done = nullptr;
  bool in_flight = false;
};

struct TimedLaunch {
  hipEvent_t start, stop;
};

struct Table {
  int64_t offset, rows, head_base;
  bool pending;  // has un-consumed gradient lists
  bool dense = false;  // its dense gradient may be non-zero (materialised lists, or a caller that writes gradients itself):
                       // the next optimiser pass reads (and re-zeroes) it next to the lists
  // lazy Adam (gqe_set_lazy_adam): lstep = Adam steps applied to the table so far (rows are current for <= lstep),
  // since_full = steps since every row was brought to lstep (bounded by the coefficient ring), dirty = some rows lag
  int lstep = 0, since_full = 0;
  bool dirty = false;
};

struct SavedFeed {  // the table rows of the pending margin call, for the sparse optimiser launch
  GqeRowSegs segs;
};

struct Bag {
  int table;  // index into ctx->tables
  const int32_t *ptr, *ids;
  int64_t n_bags;
  int32_t max_len;
};

struct Layout {  // byte offsets inside the bound workspace
  size_t idx_cap, tloss_off, scratch_off, scratch_cap;  // [staged indices x kStageBufs | tile losses | pair scratch]
  size_t shard_req_send, shard_req_recv, shard_fetch, shard_csend;  // row-sharded mode (0 otherwise)
  size_t shard_dense;          // ... the ranks' relation / Pre / Post gradients, exchanged with the contributions: world blocks of
  int64_t shard_dense_floats;  //     shard_dense_floats received + as many staged for a callback transport (0: no such region)
  int64_t shard_cap_send, shard_cap_recv;                          // entries
  size_t seg_off, act_off, formula_off, head_off, rows_off, next_off, contrib_off, linkc_off, counter_off, last_off, ring_off, total;
  // operand-ordered copies of the d x d matrices (gqe_dev.h, GQE_TILE_INDEX): a mirror of the arena's non-table spans for M,
  // a second one for M^T (tile_floats floats each)
  size_t tile_off;
  int64_t tile_floats;
  int n_tile_spans;
  int64_t tile_span_lo[8], tile_span_hi[8], tile_span_base[8];   // arena floats [lo, hi) -> mirror floats from base (base = lo mod 64)
  size_t hot_sub_off;                 // hot word rows: the pool of sub-lists (GQE_HOT_SUB_INTS)
  size_t hot_slot_off, hot_acc_off;   // hot rows (GqeHot): slot per table row, GQE_HOT_REPS x GQE_HOT_SLOTS accumulators of dim floats
  size_t stamp_off;   // split step (gqe_train_step): one int32 per table row, 1 = named by the step's index feed
  size_t progress_off;   // ... and the riders' progress slots (GqeSplitRide::progress)
  int64_t max_entries, max_links;  // max_entries = per-rank capacity x world (the exchange gathers every rank's entries)
};

constexpr int kTimingKinds = 7;

}  // namespace

struct ShardSession;  // gqe_shard_step.h: plan board, transport and plan slots of the one-call row-sharded step

struct gqe_ctx {
  gqe_config cfg{};
  float *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr;
  int64_t n_arena = 0;
  char* ws = nullptr;
  int64_t ws_bytes = 0;
  int64_t cap_queries = 0;
  int32_t cap_batches = 0;
  Layout lay{};
  std::vector<Table> tables;
  std::vector<Bag> bags;
  bool links_used = false;  // link nodes were allocated since the last consumption
  // hot word rows' sub-lists (GqeHot): a word of pinned host memory that a promoting kernel sets, its device address, and whether
  // the fused launches link onto sub-lists yet (set when the host first reads the word as non-zero)
  int32_t* hot_seen = nullptr;
  int32_t* hot_seen_dev = nullptr;
  mutable bool hot_sub_on = false;
  // the d x d matrices the registered formulas contract with (arena offsets, sorted); tiles_dirty: their operand-ordered copies
  // have to be rebuilt before the next fused launch (new workspace, new matrix, gqe_params_changed)
  std::vector<int64_t> matrices;
  bool tiles_dirty = true;
  // gqe_set_deferred_gemm: the pair GEMM (and the losses' finalize block) of the last gqe_margin_fwd_bwd has not been launched
  // yet — it rides in front of the next Adam pass's chunks (GqeGemmRide, gqe_dev.h), or is launched on its own by whatever
  // else comes first (flush_ride)
  bool defer_gemm = false, ride_pending = false;
  int64_t rides = 0;   // Adam passes that carried a deferred pair GEMM (gqe_deferred_gemm_rides)
  // gqe_train_step's split step (gqe_split.h): split_active = the call in progress launched (or is about to launch) rider
  // workgroups in its fused launch; split_t / split_segs / split_idx describe the stepped tables and the rows its feed names;
  // mat_pending = d x d matrices whose Adam step waits for the next step's first launch (or flush_split)
  bool split_active = false, split_launched = false;
  int split_epoch = -1;   // epoch of the current split step's stamps (gqe_split.h: stamp = epoch << 16 | owning feed entry; += 1 per step)
  GqeSplitTabs split_t;
  long long split_stream_bytes = 0;   // 12 B per parameter of the tables the riders stream (p, m, v): sizes the lead riders
  GqeSplitRide split_ride;   // the rider description of the fused launch (its second launch continues the same ticket counter)
  GqeSplitSegs split_segs;
  int split_buf = -1;        // staging buffer that holds the step's (host) index feed: free once the second launch has read it
  const int32_t* split_idx = nullptr;
  float split_b1 = 0.f, split_b2 = 0.f, split_eps = 0.f;
  std::vector<GqeMatStep> mat_pending;
  float mat_b1 = 0.f, mat_b2 = 0.f, mat_eps = 0.f;
  int64_t split_steps = 0;   // steps that ran as split steps (gqe_split_steps)
  GqeFusedArgs ride_fa;
  float* ride_losses = nullptr;
  int64_t total_rows = 0;
  int64_t entries_used = 0;
  bool lazy = false;                 // gqe_set_lazy_adam
  std::vector<SavedFeed> feed;       // index segments of the pending margin call ...
  const int32_t* feed_idx = nullptr; // ... and where its (device) index feed lives
  int feed_buf = -1;                 // staging buffer holding it (-1: caller's device buffer)
  bool feed_valid = false;           // the pending gradient lists come from exactly that call
  // gqe_lazy_prefetch: the (device-resident) index feed of the NEXT call, to be caught up by the coming row launch
  std::vector<SavedFeed> next_feed;
  const int32_t* next_idx = nullptr;
  int64_t next_n_idx = 0;
  const int32_t* caught_idx = nullptr;   // feed whose rows the last optimiser step already brought up to date
  int64_t caught_n_idx = 0;
  uint64_t next_sig = 0, caught_sig = 0; // ... and what it was declared to hold (feed_signature): pointer identity alone would
                                         // accept a recycled buffer that now carries another batch layout
  float lz_lr = 0.f, lz_b1 = 0.f, lz_b2 = 0.f, lz_eps = 0.f;  // hyper-parameters the coefficient ring was written with
  bool lz_hyper = false;
  int rank = 0, world = 1;     // gqe_set_exchange: data-parallel replica id / count
  int64_t step_entries = 0;    // world > 1: contribution entries per rank slab of the pending margin call (n)
  int64_t step_slab = 0;       // ... and the slab size in entries: n + row-id tail + dense-gradient tail (S)
  bool step_exported = false;
  bool imported = false;       // the pending lists include every rank's entries (gqe_import_entries ran)
  int64_t imported_n = 0, imported_slab = 0;
  int64_t slab_hint = 0;       // gqe_exchange_reserve: slab size for the next margin call (0 = its own entry count)
  // row-sharded data parallelism (gqe_set_shard): the registered tables are this rank's shards (local row i = global
  // row i * world + rank); rows are fetched from / contributions sent to their owners by the host's transport
  int shard_rank = 0, shard_world = 1;
  bool shard_on = false;  // gqe_set_shard was called (world = 1 is the degenerate case: this rank owns every row)
  bool shard_sent = false;   // a margin call's contributions sit in the send buffer (not yet linked by their owners)
  std::vector<int> shard_tables;  // tables that margin call named (every rank runs the same formulas: these receive lists)
  RingSlot ring[kRing];
  int ring_next = 0;
  // formula descriptor cache: static per-formula data lives on the device, per-call data travels as kernel
  // arguments, so a steady-state iteration uploads nothing
  // (least-recently-used slots are re-used once cap_formulas descriptors are cached: a multi-relational graph has
  // thousands of distinct 3-hop formulas, and formula ids are per-call kernel arguments, so a slot can be re-uploaded)
  std::vector<GqeDevFormula> formulas;
  std::vector<std::string> formula_keys;      // slot -> cache key
  std::vector<long long> formula_used;        // slot -> stamp of the last call that used it
  std::vector<int> formulas_dirty;            // slots whose device copy is stale
  std::map<std::string, int> formula_ids;
  long long call_stamp = 0;
  int cap_formulas = GQE_DEFAULT_FORMULAS, cap_tensors = GQE_DEFAULT_TENSORS;  // gqe_set_limits
  // host index feeds are uploaded on a side stream into one of two device buffers, so the copy for
  // iteration i+1 overlaps the kernels of iteration i instead of sitting between them
  hipStream_t up = nullptr;
  hipEvent_t plan_ready[2] = {nullptr, nullptr}, plan_free[2] = {nullptr, nullptr};
  bool plan_free_set[2] = {false, false};
  int plan_buf = 0;
  std::string err;
  long long* prof = nullptr;  // optional per-workgroup phase stamps (gqe_debug_profile)
  std::map<int64_t, int> adam_steps;          // per-tensor step counters for callers that pass step <= 0
  std::vector<GqeDevSeg> universe;             // every tensor ever stepped (device copy at lay.seg_off)
  size_t universe_uploaded = 0;                // entries of `universe` the device table already holds
  int timing = 0;                              // record every `timing`-th launch of each kernel (0 = off)
  long long timing_calls[kTimingKinds] = {0, 0, 0, 0, 0, 0, 0};
  bool timing_open[kTimingKinds] = {false, false, false, false, false, false, false};
  // 0 fused, 1 pair GEMM, 2 optimiser (tables), 3 lazy: small tensors, 4 lazy: catch-up before a read,
  // 5 row-sharded step: serve + exchange of rows, 6 row-sharded step: exchange of contributions + link + all-reduces
  std::vector<TimedLaunch> timed[kTimingKinds];
  ShardSession* shard_sess = nullptr;   // gqe_shard_open
  bool shard_internal = false;          // gqe_shard_step is driving the phase entry points
  bool shard_link_early = false;        // ... a margin step: its serve kernel links the entries that will answer the requests
  // ... which keeps this rank's OWN block in place (no copy through the send / receive buffers): requests
  // [own_lo, own_lo + own_n) of the received list are its own, their rows are served into the fetched buffer at row own_fetch
  // and their contributions are linked as entries own_entry + k (the send region is part of the entry space)
  int64_t own_lo = 0, own_n = 0, own_fetch = 0, own_entry = 0;
  // ... unless the step's plan named the own rows directly (GQE_OWN_ROW, margin steps of a session that keeps the own block in
  // place): then nothing is served or linked for the own block — the fused kernel reads the shard and links like an unsharded step
  bool own_direct = false;
  bool ordered_sums = false;            // gqe_set_ordered_sums
  std::vector<TimedLaunch> event_pool;  // recycled hipEvent pairs (creation is not free)
};


// ------------------------------------------------------------------------------------------
// native training feed (first half): per-formula query pools live on the host as int32
// row arrays; an iteration = draw a formula per batch (probability ~ pool size), slice it with the
// reference's wrap-around rule (train_helpers.py:100-105), draw negatives (1-chain: any node of the target
// mode, model.py:118; otherwise the stored negative / hard negative), pack, launch, step — no Python per
// iteration.
// ------------------------------------------------------------------------------------------
struct FeederPool {
  gqe_batch proto;  // static fields of the formula
  int64_t n;
  std::vector<int32_t> target, anchors /*[k][n]*/, neg, hard;
  // reference streams (gqe_feeder_add_pool_lists): every query's negative / hard-negative LIST as the reference's files hold
  // them (rows; CSR), one of which random.choice picks per query and batch (model.py:113-120)
  std::vector<int64_t> neg_ptr, hard_ptr;
  std::vector<int32_t> neg_rows, hard_rows;
};

struct gqe_feeder {
  gqe_ctx* ctx;
  uint64_t rng[2];
  uint64_t neg_rng[2];   // row-sharded runs: 1-chain negatives are drawn from a per-rank stream
  int32_t batch_size;
  float path_weight, inter_weight;
  std::vector<FeederPool> pools;
  std::vector<int> by_type[7];                          // pool indices per query type
  std::vector<double> cum[7];                           // cumulative pool sizes per type
  std::map<int64_t, std::vector<int32_t>> mode_rows;    // table offset -> rows to draw 1-chain negatives from
  std::vector<int32_t> idx;
  std::vector<gqe_batch> batches;
  std::vector<gqe_segment> segs;
  // prepared iterations (a ring of 2 * kFeedGroup): the batches / touched tensors of an iteration, where its feed starts in
  // the group buffer, its length
  struct Prepared {
    int64_t it = -1;
    std::vector<gqe_batch> batches;
    std::vector<gqe_segment> segs;
    std::vector<int32_t> host_idx;     // row-sharded mode: the host feed gqe_shard_post takes
    const int32_t* dev_idx = nullptr;  // where the kernels read the feed (device buffer or pinned host slot)
    const int32_t* host_src = nullptr; // the packed feed in pinned host memory (gqe_feeder_debug_feed)
    int64_t n_idx = 0;
  };
  Prepared prep[2 * kFeedGroup];
  // copy mode: two pinned group buffers (kFeedGroup feeds each) -> two groups of staged buffers in the workspace
  int32_t* grp_pin[2] = {nullptr, nullptr};
  size_t grp_cap = 0;                  // int32 entries per feed slot of a group buffer
  hipEvent_t grp_ready[2] = {nullptr, nullptr}, grp_free[2] = {nullptr, nullptr};
  bool grp_free_set[2] = {false, false};
  // Index feed of an iteration, two ways (gqe_feeder_set_feed):
  //   0  pinned staging + hipMemcpyAsync on the library's upload stream (what gqe_margin_fwd_bwd does for host feeds)
  //   1  the kernels read the feed straight from pinned host memory (default): no copy, no cross-stream dependency and
  //      no marker packet between the iteration's kernels — the upload's two event packets cost ~10 us of a 88 us
  //      iteration, a 70 KB feed read over PCIe ~2 us.  16 pinned slots; an event every 8 iterations guards their re-use.
  int feed_mode = 1;
  int32_t* pin[2 * kFeedGroup] = {};
  size_t pin_cap = 0;
  hipEvent_t pin_ev[2] = {nullptr, nullptr};
  bool pin_ev_set[2] = {false, false};
  long long pin_it = 0;  // iterations fed so far
  // prepared iterations, pinned slots and their guard events are keyed on the iteration number: a run that does not continue
  // where the previous one ended (or changes burn_in) starts from a clean ring
  int64_t next_it = -1;
  int32_t last_burn_in = -1;
  // reference streams (gqe_feeder_set_reference_streams): the formula of a batch is np.random.multinomial's draw and the
  // negatives are random.choice's, replayed on the caller's two MT19937 states (gqe_mt.h) — a run seeded like the reference's
  // trains on the reference's batches (train_helpers.run_train).  pvals[type]: the probability vector as the caller computed it;
  // type_order: the query types behind 1-chain in the order of the caller's training dictionary.
  uint32_t* np_state = nullptr;
  uint32_t* py_state = nullptr;
  std::vector<double> pvals[7];
  std::vector<int> type_order;
  int64_t loss_stride = 0;   // > 0: iteration i of a run writes its losses at losses + (i - first_iteration) * loss_stride
  int64_t run_end = 0;       // (copy mode samples a group of iterations together: never past the end of the run)
  int64_t queries_fed = 0;   // queries of every batch packed so far (gqe_feeder_queries)
  double host_build_s = 0, host_run_s = 0;   // wall time spent sampling + packing feeds / inside gqe_feeder_run (gqe_feeder_host_seconds)
  bool sgd = false;          // gqe_feeder_set_sgd: the iteration closes with gqe_sgd_step(lr) instead of the Adam step
};

namespace {

int fail(gqe_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx)
    ctx->err = buf;
  else
    g_create_error = buf;
  return code;
}

#define HIP_TRY(ctx, call)                                                                             \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) return fail(ctx, GQE_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

bool is_mlp(const gqe_ctx* c) { return c->cfg.inter == GQE_INTER_MIN || c->cfg.inter == GQE_INTER_MEAN; }
bool is_min(const gqe_ctx* c) { return c->cfg.inter == GQE_INTER_MIN || c->cfg.inter == GQE_INTER_MIN_SIMPLE; }

int anchors_of(int qtype) {
  switch (qtype) {
    case GQE_Q_1CHAIN: case GQE_Q_2CHAIN: case GQE_Q_3CHAIN: return 1;
    case GQE_Q_2INTER: case GQE_Q_3INTER_CHAIN: case GQE_Q_3CHAIN_INTER: return 2;
    case GQE_Q_3INTER: return 3;
    default: return -1;
  }
}

// the arena floats that are NOT embedding tables (relation vectors / matrices, Pre / Post), as <= 8 spans
GqeSpans dense_spans(const gqe_ctx* ctx) {
  GqeSpans sp;
  memset(&sp, 0, sizeof sp);
  std::vector<std::pair<int64_t, int64_t>> t;
  for (const Table& tb : ctx->tables) t.emplace_back(tb.offset, tb.rows * (int64_t)ctx->cfg.dim);
  std::sort(t.begin(), t.end());
  int64_t at = 0;
  auto add = [&](int64_t b, int64_t e) {
    if (e <= b) return;
    if (sp.n == 8) { sp.n = -1; return; }
    sp.off[sp.n] = b;
    sp.len[sp.n] = e - b;
    sp.total += e - b;
    ++sp.n;
  };
  for (auto& x : t) {
    if (sp.n < 0) break;
    add(at, x.first);
    at = std::max(at, x.first + x.second);
  }
  if (sp.n >= 0) add(at, ctx->n_arena);
  return sp;
}

// slab of one rank in the exchanged entry space, in entries of dim floats: n contributions, the row ids of those
// entries (int32, packed), the dense gradient spans
int64_t slab_entries(const gqe_ctx* ctx, int64_t n, int64_t dense_floats) {
  const int64_t d = ctx->cfg.dim;
  return n + (n + d - 1) / d + (dense_floats + d - 1) / d;
}

Layout make_layout(const gqe_ctx* ctx, int64_t max_queries, int32_t max_batches) {
  Layout L;
  const int64_t rows = max_queries + (int64_t)GQE_TQ * max_batches;  // queries incl. tile padding
  L.idx_cap = align_up((size_t)rows * kRolesPerQuery * sizeof(int32_t), 256);  // exists kStageBufs times (double-buffered uploads + feeder groups)
  L.tloss_off = (size_t)kStageBufs * L.idx_cap;
  L.scratch_off = L.tloss_off + align_up(sizeof(float) * (size_t)(rows / GQE_TQ + GQE_MAX_BATCHES + 1), 256);
  L.scratch_cap = align_up((size_t)rows * kMaxSlots * ctx->cfg.dim * sizeof(float), 256);
  L.seg_off = L.scratch_off + L.scratch_cap;
  L.act_off = L.seg_off + align_up(sizeof(GqeDevSeg) * (size_t)ctx->cap_tensors, 256);
  L.formula_off = L.act_off + align_up(sizeof(GqeActSeg) * (size_t)ctx->cap_tensors, 256);
  L.head_off = L.formula_off + align_up(sizeof(GqeDevFormula) * (size_t)ctx->cap_formulas, 256);
  L.max_entries = (int64_t)align_up((size_t)(rows * kRolesPerQuery), 64);
  if (ctx->world > 1) {
    const GqeSpans sp = dense_spans(ctx);
    L.max_entries = (int64_t)align_up((size_t)slab_entries(ctx, L.max_entries, sp.n < 0 ? 0 : sp.total), 64) * ctx->world;
  }
  L.shard_cap_send = L.shard_cap_recv = 0;
  if (ctx->shard_on) {
    // a rank requests at most one row per index of a call; in the worst case every rank's requests land on one owner.
    // The contribution entries that owner receives are the ctx's entry space (the lists link into it); it doubles as
    // the buffer the served rows are gathered into (rows are served before the fused kernel, contributions arrive after)
    L.shard_cap_send = L.max_entries;
    L.shard_cap_recv = L.max_entries * ctx->shard_world;
    // behind the received entries: the block this rank's fused kernel writes its contributions to (the SEND buffer is part
    // of the entry space, so the contributions to rows this rank owns itself are linked where they are), and one more
    // block for the contributions of bag (replicated) tables, which are linked locally
    // (the same block takes the contributions to rows of this rank's OWN shard that the fused kernel links itself,
    // GQE_OWN_ROW: a role is a bag role or a plain one, so the (role, query) entries of the two never collide)
    L.max_entries = L.shard_cap_recv + 2 * L.shard_cap_send;
  }
  // entry -> list head it was pushed on (-1: not pushed); sits exactly max_entries ints below next[], so the
  // kernels address it as next[entry - max_entries]
  L.rows_off = L.head_off + align_up(sizeof(int32_t) * (size_t)std::max<int64_t>(ctx->total_rows, 1), 256);
  L.next_off = L.rows_off + sizeof(int32_t) * (size_t)L.max_entries;
  int32_t max_len = 0;  // bag modes: every (entry, word) pair needs a link node
  for (const Bag& bg : ctx->bags) max_len = std::max(max_len, bg.max_len);
  L.max_links = L.max_entries * max_len;
  L.contrib_off = L.next_off + align_up(sizeof(int32_t) * (size_t)(L.max_entries + L.max_links), 256);
  L.linkc_off = L.contrib_off + align_up(sizeof(float) * (size_t)L.max_entries * ctx->cfg.dim, 256);
  L.counter_off = L.linkc_off + align_up(sizeof(int32_t) * (size_t)std::max<int64_t>(L.max_links, 1), 256);
  L.last_off = L.counter_off + 256;  // lazy Adam: per-row step counts + per-table coefficient rings
  L.ring_off = L.last_off + align_up(sizeof(int32_t) * (size_t)std::max<int64_t>(ctx->total_rows, 1), 256);
  L.hot_slot_off = L.ring_off + align_up(sizeof(float) * 2 * GQE_LAZY_TABLES * GQE_LAZY_RING, 256);
  L.hot_acc_off = L.hot_slot_off + align_up(sizeof(int32_t) * (size_t)std::max<int64_t>(ctx->total_rows, 1), 256);
  L.hot_sub_off = L.hot_acc_off + align_up(sizeof(float) * (size_t)GQE_HOT_REPS * GQE_HOT_SLOTS * ctx->cfg.dim, 256);
  L.tile_off = L.hot_sub_off + (ctx->bags.empty() ? 0 : align_up(sizeof(int32_t) * GQE_HOT_SUB_INTS, 256));
  {
    GqeSpans sp = dense_spans(ctx);
    if (sp.n < 0) {   // more than 8 non-table spans: mirror the whole arena
      sp.n = 1;
      sp.off[0] = 0;
      sp.len[0] = ctx->n_arena;
    }
    L.n_tile_spans = sp.n;
    int64_t at = 0;
    for (int k = 0; k < sp.n; ++k) {
      L.tile_span_lo[k] = sp.off[k];
      L.tile_span_hi[k] = sp.off[k] + sp.len[k];
      L.tile_span_base[k] = at + (sp.off[k] & 63);   // mirror(off) = off (mod 64 floats): float4 stores / 16-byte lane loads stay aligned
      at = (int64_t)align_up((size_t)(L.tile_span_base[k] + sp.len[k]), 64);
    }
    L.tile_floats = at;
  }
  L.stamp_off = L.tile_off + align_up(sizeof(float) * 2 * (size_t)L.tile_floats, 256);
  L.progress_off = L.stamp_off + align_up(sizeof(int32_t) * (size_t)std::max<int64_t>(ctx->total_rows, 1), 256);
  L.total = L.progress_off + align_up(sizeof(int32_t) * (size_t)GQE_SPLIT_MAX_RIDERS * GQE_SPLIT_PWAVES, 256);
  L.shard_req_send = L.shard_req_recv = L.shard_fetch = L.shard_csend = L.shard_dense = 0;
  L.shard_dense_floats = 0;
  if (ctx->shard_on) {
    L.shard_req_send = L.total;
    L.shard_req_recv = L.shard_req_send + align_up(sizeof(int32_t) * (size_t)L.shard_cap_send, 256);
    L.shard_fetch = L.shard_req_recv + align_up(sizeof(int32_t) * (size_t)L.shard_cap_recv, 256);
    L.shard_csend = L.contrib_off + sizeof(float) * (size_t)L.shard_cap_recv * ctx->cfg.dim;   // entries [cap_recv, cap_recv + cap_send)
    L.total = L.shard_fetch + align_up(sizeof(float) * (size_t)L.shard_cap_send * ctx->cfg.dim, 256);
    const GqeSpans sp = dense_spans(ctx);
    if (sp.n > 0) {   // (also at one rank: GQE_SHARD_SELF_VIA_RCCL sends the own block through the transport — tests, overhead bench)
      L.shard_dense_floats = (int64_t)align_up((size_t)sp.total, 64);
      L.shard_dense = L.total;
      L.total += align_up(sizeof(float) * 2 * (size_t)L.shard_dense_floats * (size_t)ctx->shard_world, 256);
    }
  }
  return L;
}

// get a pinned ring slot of at least `bytes`; waits only if the slot's previous copy is still running
int ring_acquire(gqe_ctx* ctx, size_t bytes, RingSlot** out) {
  RingSlot& s = ctx->ring[ctx->ring_next];
  ctx->ring_next = (ctx->ring_next + 1) % kRing;
  if (s.in_flight) {
    if (hipEventQuery(s.done) != hipSuccess) HIP_TRY(ctx, hipEventSynchronize(s.done));
    s.in_flight = false;
  }
  if (s.cap < bytes) {
    if (s.host) HIP_TRY(ctx, hipHostFree(s.host));
    s.cap = align_up(bytes * 2, 4096);
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&s.host), s.cap, hipHostMallocDefault));
  }
  if (!s.done) HIP_TRY(ctx, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  *out = &s;
  return GQE_OK;
}

int timing_begin(gqe_ctx* ctx, int kind, hipStream_t st) {
  ctx->timing_open[kind] = false;
  if (!ctx->timing) return GQE_OK;
  if ((ctx->timing_calls[kind]++ % ctx->timing) != 0) return GQE_OK;
  ctx->timing_open[kind] = true;
  TimedLaunch t;
  if (!ctx->event_pool.empty()) {
    t = ctx->event_pool.back();
    ctx->event_pool.pop_back();
  } else {
    HIP_TRY(ctx, hipEventCreate(&t.start));
    HIP_TRY(ctx, hipEventCreate(&t.stop));
  }
  HIP_TRY(ctx, hipEventRecord(t.start, st));
  ctx->timed[kind].push_back(t);
  return GQE_OK;
}

int timing_end(gqe_ctx* ctx, int kind, hipStream_t st) {
  if (!ctx->timing_open[kind]) return GQE_OK;
  HIP_TRY(ctx, hipEventRecord(ctx->timed[kind].back().stop, st));
  return GQE_OK;
}

void drop_formulas(gqe_ctx* ctx) {
  ctx->formulas.clear();
  ctx->formula_keys.clear();
  ctx->formula_used.clear();
  ctx->formulas_dirty.clear();
  ctx->formula_ids.clear();
  ctx->matrices.clear();   // (the registry of what the formulas contract with: rebuilt as they are registered again)
  ctx->tiles_dirty = true;
}

bool off_ok(const gqe_ctx* ctx, int64_t off, int64_t numel) {
  return off >= 0 && (off % 4) == 0 && off + numel <= ctx->n_arena;
}

int table_of(const gqe_ctx* ctx, int64_t offset) {
  for (size_t t = 0; t < ctx->tables.size(); ++t)
    if (ctx->tables[t].offset == offset) return (int)t;
  return -1;
}

// Validate one caller batch and return the index of its (cached) static descriptor.
// float offset (in the workspace) of the operand-ordered copy of the matrix at arena offset `off`; -1 without a workspace
int64_t tile_of(const gqe_ctx* ctx, int64_t off) {
  if (!ctx->ws || off < 0) return -1;
  const Layout& L = ctx->lay;
  for (int k = 0; k < L.n_tile_spans; ++k)
    if (off >= L.tile_span_lo[k] && off < L.tile_span_hi[k]) return (int64_t)(L.tile_off / sizeof(float)) + L.tile_span_base[k] + (off - L.tile_span_lo[k]);
  return -1;
}

// the tile fields of a formula descriptor (they depend on the bound workspace), and its matrices into the registry
// returns false if one of them does not lie in the non-table part of the arena (it has no place in the mirror)
bool formula_tiles(gqe_ctx* ctx, GqeDevFormula& f) {
  const bool bil = ctx->cfg.decoder == GQE_DEC_BILINEAR;
  bool ok = true;
  auto reg = [&](int64_t off) {
    if (off < 0) return (int64_t)-1;
    if (ctx->ws && tile_of(ctx, off) < 0) {
      ok = false;
      return (int64_t)-1;
    }
    auto it = std::lower_bound(ctx->matrices.begin(), ctx->matrices.end(), off);
    if (it == ctx->matrices.end() || *it != off) {
      ctx->matrices.insert(it, off);
      ctx->tiles_dirty = true;
    }
    return tile_of(ctx, off);
  };
  for (int i = 0; i < GQE_MAX_BRANCH; ++i)
    for (int h = 0; h < GQE_MAX_HOPS; ++h) f.hop_tile[i][h] = (bil && h < f.n_hops[i]) ? reg(f.hop_param[i][h]) : -1;
  f.final_tile = (bil && f.n_final) ? reg(f.final_param) : -1;
  f.pre_tile = reg(f.pre_param);     // (-1 unless the formula is an MLP intersection)
  f.post_tile = reg(f.post_param);
  f.tile_t = ctx->lay.tile_floats;
  return ok;
}

// rebuild the copies of every registered matrix from the parameter arena
int retile(gqe_ctx* ctx, hipStream_t st) {
  if (!ctx->ws || !ctx->params) return GQE_OK;
  GqeRetileArgs ra;
  ra.tile_t = ctx->lay.tile_floats;
  for (size_t a = 0; a < ctx->matrices.size(); a += GQE_RETILE_MAX) {
    ra.n = (int)std::min<size_t>(GQE_RETILE_MAX, ctx->matrices.size() - a);
    for (int k = 0; k < ra.n; ++k) {
      ra.param[k] = ctx->matrices[a + (size_t)k];
      ra.tile[k] = tile_of(ctx, ra.param[k]);
      if (ra.tile[k] < 0) return fail(ctx, GQE_ERR_STATE, "internal: the matrix at offset %lld has no place for its operand copy", (long long)ra.param[k]);
    }
    HIP_TRY(ctx, gqe_launch_retile(ra, ctx->params, reinterpret_cast<float*>(ctx->ws), ctx->cfg.dim, st));
  }
  ctx->tiles_dirty = false;
  return GQE_OK;
}

// the deferred pair GEMM as a launch of its own (nothing it can ride on comes next)
int flush_ride(gqe_ctx* ctx, hipStream_t st) {
  if (!ctx->ride_pending) return GQE_OK;
  ctx->ride_pending = false;
  ctx->ride_fa.stream = st;
  int rc = timing_begin(ctx, 1, st);
  if (rc != GQE_OK) return rc;
  HIP_TRY(ctx, gqe_launch_pair_gemm(ctx->ride_fa, ctx->ride_losses));
  return timing_end(ctx, 1, st);
}

// the d x d matrices a split step (gqe_train_step) left for the next step's first launch: stepped now, by a launch of their own
// (whatever comes next is not a split step, or wants the parameters complete)
int flush_split(gqe_ctx* ctx, hipStream_t st) {
  if (ctx->mat_pending.empty()) return GQE_OK;
  std::vector<GqeMatStep> pending;
  pending.swap(ctx->mat_pending);
  for (const GqeMatStep& ms : pending)
    HIP_TRY(ctx, gqe_launch_matstep(ms, ctx->params, ctx->grads, ctx->m, ctx->v, ctx->cfg.dim, ctx->mat_b1, ctx->mat_b2, ctx->mat_eps, st));
  return GQE_OK;
}

// GQE_CHECK_TILES=1 (debug; synchronises the stream): are the copies what the parameters say?  A caller that writes parameter
// values without gqe_params_changed gets an error here instead of results computed with the old matrices.
int check_tiles(gqe_ctx* ctx, hipStream_t st) {
  if (!ctx->ws || !ctx->params || ctx->matrices.empty()) return GQE_OK;
  int32_t* cnt = reinterpret_cast<int32_t*>(ctx->ws + ctx->lay.counter_off + 192);
  HIP_TRY(ctx, hipMemsetAsync(cnt, 0, sizeof(int32_t), st));
  GqeRetileArgs ra;
  ra.tile_t = ctx->lay.tile_floats;
  for (size_t a = 0; a < ctx->matrices.size(); a += GQE_RETILE_MAX) {
    ra.n = (int)std::min<size_t>(GQE_RETILE_MAX, ctx->matrices.size() - a);
    for (int k = 0; k < ra.n; ++k) {
      ra.param[k] = ctx->matrices[a + (size_t)k];
      ra.tile[k] = tile_of(ctx, ra.param[k]);
    }
    HIP_TRY(ctx, gqe_launch_tilecheck(ra, ctx->params, reinterpret_cast<const float*>(ctx->ws), ctx->cfg.dim, cnt, st));
  }
  int32_t bad = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&bad, cnt, sizeof bad, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  if (bad)
    return fail(ctx, GQE_ERR_STATE, "GQE_CHECK_TILES: %d float4 groups of the d x d matrices differ from their operand-ordered copies: parameter values "
                "were written without gqe_params_changed()", bad);
  return GQE_OK;
}

int formula_of(gqe_ctx* ctx, const gqe_batch& s, int bi, int* out_id) {
  const int d = ctx->cfg.dim;
  const bool bil = ctx->cfg.decoder == GQE_DEC_BILINEAR;
  const bool mlp = is_mlp(ctx);
  const int64_t vec = bil ? (int64_t)d * d : d;
  const int na = anchors_of(s.qtype);
  if (na < 0) return fail(ctx, GQE_ERR_ARG, "batch %d: unknown query type %d", bi, s.qtype);
  if (s.n_anchors != na) return fail(ctx, GQE_ERR_ARG, "batch %d: query type %d needs %d anchors, got %d", bi, s.qtype, na, s.n_anchors);
  const bool chain = s.qtype <= GQE_Q_3CHAIN;
  const int nbr = chain ? 1 : na;
  // cache key = every static field the caller controls
  struct Key {
    int32_t qtype, n_anchors, n_final, n_hops[GQE_MAX_BRANCH];
    int64_t target_table, anchor_table[GQE_MAX_BRANCH], hop_param[GQE_MAX_BRANCH][GQE_MAX_HOPS], final_param, pre_param, post_param;
  } key;
  memset(&key, 0, sizeof key);
  key.qtype = s.qtype;
  key.n_anchors = na;
  key.n_final = s.n_final;
  key.target_table = s.target_table;
  for (int i = 0; i < na; ++i) key.anchor_table[i] = s.anchor_table[i];
  for (int i = 0; i < nbr; ++i) {
    key.n_hops[i] = s.n_hops[i];
    for (int h = 0; h < s.n_hops[i] && h < GQE_MAX_HOPS; ++h) key.hop_param[i][h] = s.hop_param[i][h];
  }
  key.final_param = s.n_final ? s.final_param : -1;
  key.pre_param = (!chain && mlp) ? s.pre_param : -1;
  key.post_param = (!chain && mlp) ? s.post_param : -1;
  const std::string ks(reinterpret_cast<const char*>(&key), sizeof key);
  auto it = ctx->formula_ids.find(ks);
  if (it != ctx->formula_ids.end()) {
    *out_id = it->second;
    ctx->formula_used[it->second] = ctx->call_stamp;
    return GQE_OK;
  }
  GqeDevFormula f;
  memset(&f, 0xff, sizeof f);  // all slots / params = -1
  f.qtype = s.qtype;
  f.n_anchors = na;
  f.n_final = 0;
  if (!off_ok(ctx, s.target_table, d)) return fail(ctx, GQE_ERR_ARG, "batch %d: target_table offset %lld outside the arena", bi, (long long)s.target_table);
  f.target_table = s.target_table;
  const int tt = table_of(ctx, s.target_table);
  if (tt < 0) return fail(ctx, GQE_ERR_STATE, "batch %d: target_table %lld is not a registered table (gqe_set_tables)", bi, (long long)s.target_table);
  f.target_head = ctx->tables[tt].head_base;
  auto bag_of = [&](int table) {
    for (size_t k = 0; k < ctx->bags.size(); ++k)
      if (ctx->bags[k].table == table) return (int)k;
    return -1;
  };
  f.target_bag = bag_of(tt);
  for (int i = 0; i < GQE_MAX_BRANCH; ++i) f.anchor_bag[i] = -1;
  for (int i = 0; i < GQE_MAX_BRANCH; ++i) f.n_hops[i] = 0;
  for (int i = 0; i < na; ++i) {
    if (!off_ok(ctx, s.anchor_table[i], d)) return fail(ctx, GQE_ERR_ARG, "batch %d: anchor_table[%d] outside the arena", bi, i);
    f.anchor_table[i] = s.anchor_table[i];
    const int ta = table_of(ctx, s.anchor_table[i]);
    if (ta < 0) return fail(ctx, GQE_ERR_STATE, "batch %d: anchor_table[%d] is not a registered table (gqe_set_tables)", bi, i);
    f.anchor_head[i] = ctx->tables[ta].head_base;
    f.anchor_bag[i] = bag_of(ta);
  }
  for (int i = 0; i < nbr; ++i) {
    const int nh = s.n_hops[i];
    const int max_h = chain ? (s.qtype + 1) : ((s.qtype == GQE_Q_3INTER_CHAIN && i == 1) ? 2 : 1);
    if (nh != max_h) return fail(ctx, GQE_ERR_ARG, "batch %d: branch %d has %d hops, query type %d needs %d", bi, i, nh, s.qtype, max_h);
    f.n_hops[i] = nh;
    for (int h = 0; h < nh; ++h) {
      if (!off_ok(ctx, s.hop_param[i][h], vec)) return fail(ctx, GQE_ERR_ARG, "batch %d: hop_param[%d][%d] outside the arena", bi, i, h);
      f.hop_param[i][h] = s.hop_param[i][h];
    }
  }
  if (!chain) {
    if (s.qtype == GQE_Q_3CHAIN_INTER) {
      if (s.n_final != 1 || !off_ok(ctx, s.final_param, vec)) return fail(ctx, GQE_ERR_ARG, "batch %d: 3-chain_inter needs one final projection", bi);
      f.n_final = 1;
      f.final_param = s.final_param;
    } else if (s.n_final != 0) {
      return fail(ctx, GQE_ERR_ARG, "batch %d: only 3-chain_inter has a final projection", bi);
    }
    if (mlp) {
      if (!off_ok(ctx, s.pre_param, (int64_t)d * d) || !off_ok(ctx, s.post_param, (int64_t)d * d))
        return fail(ctx, GQE_ERR_ARG, "batch %d: pre/post matrices outside the arena", bi);
      f.pre_param = s.pre_param;
      f.post_param = s.post_param;
    }
  }
  // pair-scratch slots + deferred dM jobs (used by training launches only)
  int nslot = 0, njob = 0;
  auto add_job = [&](int64_t param, int Lslot, int Rslot) {
    f.job_param[njob] = param;
    f.job_L[njob] = Lslot;
    f.job_R[njob] = Rslot;
    ++njob;
  };
  if (chain && bil) {
    for (int sde = 0; sde < 2; ++sde)
      for (int h = 0; h < f.n_hops[0]; ++h) {
        f.slot_act[sde][h] = nslot++;
        f.slot_gact[sde][h] = nslot++;
        add_job(f.hop_param[0][h], f.slot_act[sde][h], f.slot_gact[sde][h]);  // act_{h+1} = act_h M_h => dM_h += act_h^T g_{h+1}
      }
  }
  if (!chain) {
    if (bil) {
      for (int i = 0; i < na; ++i)
        for (int h = 0; h < f.n_hops[i]; ++h) {
          f.slot_x[i][h] = nslot++;
          f.slot_gy[i][h] = nslot++;
          add_job(f.hop_param[i][h], f.slot_gy[i][h], f.slot_x[i][h]);  // y = M x  =>  dM += g_y x^T
        }
      if (f.n_final) {
        f.slot_fx = nslot++;
        f.slot_fg = nslot++;
        add_job(f.final_param, f.slot_fg, f.slot_fx);
      }
    }
    if (mlp) {
      for (int i = 0; i < na; ++i) {
        f.slot_e[i] = nslot++;
        f.slot_gz[i] = nslot++;
        add_job(f.pre_param, f.slot_gz[i], f.slot_e[i]);  // z = Pre e  => dPre += g_z e^T
      }
      f.slot_hh = nslot++;
      f.slot_gq = nslot++;
      add_job(f.post_param, f.slot_gq, f.slot_hh);        // q = Post h => dPost += g_q h^T
    }
  }
  if (nslot > kMaxSlots || njob > GQE_MAX_JOBS) return fail(ctx, GQE_ERR_ARG, "internal: %d scratch slots / %d jobs", nslot, njob);
  f.n_slots = nslot;
  f.n_jobs = njob;
  if (!formula_tiles(ctx, f))
    return fail(ctx, GQE_ERR_ARG, "batch %d: a d x d matrix of the formula (Pre / Post / a Bilinear relation matrix) lies inside a registered table", bi);
  int slot;
  if ((int)ctx->formulas.size() < ctx->cap_formulas) {
    slot = (int)ctx->formulas.size();
    ctx->formulas.push_back(f);
    ctx->formula_keys.push_back(ks);
    ctx->formula_used.push_back(ctx->call_stamp);
  } else {
    // cache full: re-use the least recently used slot that the current call does not name
    slot = -1;
    for (int k = 0; k < (int)ctx->formulas.size(); ++k)
      if (ctx->formula_used[k] < ctx->call_stamp && (slot < 0 || ctx->formula_used[k] < ctx->formula_used[slot])) slot = k;
    if (slot < 0)
      return fail(ctx, GQE_ERR_ARG, "one call names more than %d distinct formulas (gqe_set_limits raises the cache size)", ctx->cap_formulas);
    ctx->formula_ids.erase(ctx->formula_keys[slot]);
    ctx->formulas[slot] = f;
    ctx->formula_keys[slot] = ks;
    ctx->formula_used[slot] = ctx->call_stamp;
  }
  ctx->formulas_dirty.push_back(slot);
  *out_id = slot;
  ctx->formula_ids[ks] = slot;
  return GQE_OK;
}

// Hot rows (GqeHot, gqe_dev.h).  `produce`: the struct for a kernel that ADDS contributions (the fused kernel) — off where list
// sums have to be order-independent (replicated exchange mode, gqe_set_ordered_sums) or GQE_HOT=0; consumers always get the live
// struct: whatever sits in the accumulators has to be stepped.
GqeHot hot_args(const gqe_ctx* ctx, bool produce) {
  static const int min_len = [] {   // GQE_HOT_MIN_LEN: list length at which a row is promoted (tests lower it)
    const char* e = getenv("GQE_HOT_MIN_LEN");
    return e && atoi(e) > 0 ? atoi(e) : GQE_HOT_MIN_LEN;
  }();
  static const bool off = [] {
    const char* e = getenv("GQE_HOT");
    return e && atoi(e) == 0;
  }();
  const Layout& L = ctx->lay;
  GqeHot h;
  h.slot = reinterpret_cast<int32_t*>(ctx->ws + L.hot_slot_off);
  h.acc = reinterpret_cast<float*>(ctx->ws + L.hot_acc_off);
  h.count = reinterpret_cast<int32_t*>(ctx->ws + L.counter_off + 64);
  h.cap = GQE_HOT_SLOTS;
  h.min_len = off ? 0x7fffffff : min_len;
  static const int few_len = [] {   // GQE_HOT_FEW_LEN (tuning runs): 0 = every hot row spreads over all its accumulators
    const char* e = getenv("GQE_HOT_FEW_LEN");
    return e ? atoi(e) : GQE_HOT_FEW_LEN;
  }();
  h.few_len = few_len;
  if (produce && (off || ctx->world > 1 || ctx->ordered_sums)) h.slot = nullptr;
  // sub-lists of hot word rows: only contexts with bag tables have the pool.  A producer gets it once the host has seen a
  // promotion (the word of pinned memory a promoting kernel sets; until then it adds directly, which is always correct) —
  // run_queries launches the gather behind every producer that got it
  static const bool sub_off = [] {
    const char* e = getenv("GQE_HOT_SUB");
    return e && atoi(e) == 0;
  }();
  h.sub = h.sub_count = h.seen = nullptr;
  if (!ctx->bags.empty() && !sub_off && ctx->hot_seen) {
    if (!ctx->hot_sub_on && *static_cast<volatile int32_t*>(ctx->hot_seen) != 0) ctx->hot_sub_on = true;
    if (!produce || (h.slot && ctx->hot_sub_on)) {
      h.sub = reinterpret_cast<int32_t*>(ctx->ws + L.hot_sub_off);
      h.sub_count = reinterpret_cast<int32_t*>(ctx->ws + L.counter_off + 68);
      h.seen = ctx->hot_seen_dev;
    }
  }
  return h;
}

// p + m + v of the context's tables beyond what the optimiser pass streams with the default cache policy (GQE_NT_STREAM_BYTES)
// GQE_RIDE_SPREAD=1 (or K): the deferred pair-GEMM units ride spread through a non-temporal pass.  OFF by default: on most boxes the
// step gains 12-14 us of ~630 (reddit-synth), but one run in three lands in a mode where the pass with the units takes 40-60 us
// longer (581 / 598 / 601 us against 540-560) — EXPERIMENTS.md 102
bool ride_spread_off() {
  static const bool off = !(getenv("GQE_RIDE_SPREAD") && atoi(getenv("GQE_RIDE_SPREAD")) > 0);
  return off;
}
bool big_tables(const gqe_ctx* ctx) {
  long long bytes = 0;
  for (const Table& t : ctx->tables) bytes += 12ll * t.rows * ctx->cfg.dim;
  return !ride_spread_off() && bytes > GQE_NT_STREAM_BYTES;
}

bool any_dense(const gqe_ctx* ctx) {
  for (const Table& t : ctx->tables)
    if (t.dense) return true;
  return false;
}

bool is_bag_table(const gqe_ctx* ctx, int t) {
  for (const Bag& bg : ctx->bags)
    if (bg.table == t) return true;
  return false;
}

// ---- lazy Adam helpers -------------------------------------------------------------------------------
bool lazy_table_ok(const gqe_ctx* ctx, int t) {
  if (t < 0 || t >= GQE_LAZY_TABLES) return false;
  for (const Bag& bg : ctx->bags)
    if (bg.table == t) return false;
  return true;
}

// which runs of the index feed name rows of which table (the layout of gqe_batch's index block)
void build_feed(const gqe_ctx* ctx, const gqe_batch* batches, int n_batches, bool bwd, const std::vector<int>& fid,
                std::vector<SavedFeed>& out) {
  out.clear();
  SavedFeed cur;
  memset(&cur, 0, sizeof cur);
  auto push = [&](int64_t idx_begin, int64_t count, int64_t table_offset) {
    if (count <= 0) return;
    if (cur.segs.n == GQE_LAZY_SEGS) {
      out.push_back(cur);
      memset(&cur, 0, sizeof cur);
    }
    GqeRowSegs& g = cur.segs;
    const int t = table_of(ctx, table_offset);
    g.idx_begin[g.n] = (long long)idx_begin;
    g.tid[g.n] = (int8_t)(lazy_table_ok(ctx, t) ? t : -1);
    g.begin[g.n] = g.total;
    g.total += (int)count;
    g.begin[++g.n] = g.total;
  };
  for (int bi = 0; bi < n_batches; ++bi) {
    const gqe_batch& s = batches[bi];
    const GqeDevFormula& f = ctx->formulas[fid[bi]];
    const int64_t B = s.n_queries, o = s.idx_offset;
    if (s.n_candidates > 0) {
      for (int i = 0; i < f.n_anchors; ++i) push(o + i * B, B, f.anchor_table[i]);
      push(o + f.n_anchors * B + B + 1, s.n_candidates, f.target_table);
    } else {
      const int lead = bwd ? 2 : 1;
      push(o, B * lead, f.target_table);  // target [| negative]: the same table
      for (int i = 0; i < f.n_anchors; ++i) push(o + (lead + i) * B, B, f.anchor_table[i]);
    }
  }
  if (cur.segs.n) out.push_back(cur);
}

void lazy_rows_args(const gqe_ctx* ctx, GqeRowsArgs& ra, hipStream_t st) {
  const Layout& L = ctx->lay;
  memset(&ra.t, 0, sizeof ra.t);
  ra.t.n = (int)std::min<size_t>(ctx->tables.size(), GQE_LAZY_TABLES);
  for (size_t t = 0; t < ctx->tables.size() && t < GQE_LAZY_TABLES; ++t) {
    ra.t.offset[t] = ctx->tables[t].offset;
    ra.t.head_base[t] = ctx->tables[t].head_base;
    ra.t.target[t] = ctx->tables[t].lstep;
    ra.t.grad_step[t] = -1;
  }
  ra.last = reinterpret_cast<int32_t*>(ctx->ws + L.last_off);
  ra.ring = reinterpret_cast<float2*>(ctx->ws + L.ring_off);
  ra.p = ctx->params;
  ra.g = ctx->grads;
  ra.m = ctx->m;
  ra.v = ctx->v;
  ra.lr = ctx->lz_lr;
  ra.dsegs = nullptr;
  ra.n_dsegs = 0;
  ra.dense_chunks = 0;
  memset(&ra.dcoef, 0, sizeof ra.dcoef);
  memset(ra.dactive.group, 0xFF, sizeof ra.dactive.group);
  ra.head = reinterpret_cast<int32_t*>(ctx->ws + L.head_off);
  ra.next = reinterpret_cast<const int32_t*>(ctx->ws + L.next_off);
  ra.contrib = reinterpret_cast<const float*>(ctx->ws + L.contrib_off);
  ra.max_entries = (int32_t)L.max_entries;
  ra.d = ctx->cfg.dim;
  ra.b1 = ctx->lz_b1;
  ra.b2 = ctx->lz_b2;
  ra.eps = ctx->lz_eps;
  ra.with_grad = false;
  ra.sorted = false;
  ra.hot = hot_args(ctx, false);
  ra.stream = st;
}

// what a device-resident feed was declared to hold: every field of the batch descriptors that decides which index names a
// row of which table (FNV-1a)
uint64_t feed_signature(const gqe_batch* batches, int n_batches, bool with_negatives) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](int64_t v) {
    for (int k = 0; k < 8; ++k) {
      h ^= (uint64_t)(v >> (8 * k)) & 0xffu;
      h *= 1099511628211ull;
    }
  };
  mix(n_batches);
  mix(with_negatives ? 1 : 0);
  for (int bi = 0; bi < n_batches; ++bi) {
    const gqe_batch& s = batches[bi];
    mix(s.qtype);
    mix(s.n_queries);
    mix(s.idx_offset);
    mix(s.n_candidates);
    mix(s.target_table);
    for (int i = 0; i < GQE_MAX_BRANCH; ++i) mix(i < s.n_anchors ? s.anchor_table[i] : -1);
  }
  return h;
}

bool lazy_any_dirty(const gqe_ctx* ctx) {
  for (const Table& t : ctx->tables)
    if (t.dirty) return true;
  return false;
}

// ---- the split step (gqe_train_step; gqe_split.h) ------------------------------------------------------------------------
// Launch M of a split step, enqueued in front of its fused launch: Adam on the d x d matrices the previous step left pending +
// the stamps of the rows this step's feed names; then the rider description of the fused launch (fa.split).
int split_first_launch(gqe_ctx* ctx, const gqe_batch* batches, int n_batches, const std::vector<int>& fid, const int32_t* d_idx,
                       GqeFusedArgs& fa, hipStream_t st) {
  const Layout& L = ctx->lay;
  const int d = ctx->cfg.dim;
  GqeSplitSegs& sg = ctx->split_segs;
  memset(&sg, 0, sizeof sg);
  auto push = [&](int64_t idx_begin, int64_t count, int64_t table_offset) -> bool {
    if (count <= 0) return true;
    if (sg.n == GQE_SPLIT_SEGS) return false;
    int slot = -1;
    for (int k = 0; k < ctx->split_t.n; ++k)
      if (ctx->split_t.offset[k] == table_offset) slot = k;
    sg.idx_begin[sg.n] = (int)idx_begin;
    sg.tid[sg.n] = (int8_t)slot;
    sg.begin[sg.n] = sg.total;
    sg.total += (int)count;
    sg.begin[++sg.n] = sg.total;
    return true;
  };
  bool ok = true;
  for (int bi = 0; bi < n_batches && ok; ++bi) {
    const gqe_batch& s = batches[bi];
    const GqeDevFormula& f = ctx->formulas[fid[bi]];
    const int64_t B = s.n_queries, o = s.idx_offset;
    ok = push(o, 2 * B, f.target_table);   // target | negative: the same table
    for (int i = 0; i < f.n_anchors && ok; ++i) ok = push(o + (2 + i) * B, B, f.anchor_table[i]);
  }
  if (!ok) return fail(ctx, GQE_ERR_STATE, "internal: split step with more than %d index segments", GQE_SPLIT_SEGS);
  ctx->split_idx = d_idx;
  int32_t* stamp = reinterpret_cast<int32_t*>(ctx->ws + L.stamp_off);
  // stamps carry the step's epoch in their high bits (and the owning feed entry in the low 16, gqe_split.h): a step that failed
  // between its launches leaves nothing a later step would read as its own; the array is cleared when the 15-bit epoch wraps
  static const int epoch0 = [] {   // GQE_SPLIT_DEBUG_EPOCH0 (tests): the first split step's epoch — so that a few steps cross the wrap
    const char* e = getenv("GQE_SPLIT_DEBUG_EPOCH0");
    return e && atoi(e) > 0 && atoi(e) <= GQE_SPLIT_MAX_EPOCH ? atoi(e) : 1;
  }();
  ctx->split_epoch = ctx->split_epoch < 1 ? epoch0 : ctx->split_epoch + 1;
  if (ctx->split_epoch > GQE_SPLIT_MAX_EPOCH) {
    HIP_TRY(ctx, hipMemsetAsync(stamp, 0, sizeof(int32_t) * (size_t)std::max<int64_t>(ctx->total_rows, 1), st));
    ctx->split_epoch = 1;
  }
  if (sg.total > GQE_SPLIT_MAX_ENTRIES) return fail(ctx, GQE_ERR_STATE, "internal: split step with %d feed entries", sg.total);
  // The riders: `lead` of them come first in the grid — they own a CU each from the launch's first microsecond (the tiles are
  // 1024-thread workgroups, one per CU; ~100 streaming CUs come close to saturating the memory system and slow the tiles'
  // latency chains: tools/probes/rider_probe.hip, split_timeline.py) — the others follow the tiles and start where a tile has
  // finished.  Rider j owns a fixed range of wave blocks; all of them stop with the launch's last tile, the second launch
  // continues every range where its rider stopped.
  static const int lead_env = [] {   // GQE_SPLIT_LEAD / GQE_SPLIT_TAIL / GQE_SPLIT_SHAPE: tuning runs only
    const char* e = getenv("GQE_SPLIT_LEAD");
    return e ? atoi(e) : -1;
  }();
  static const int tail_env = [] {
    const char* e = getenv("GQE_SPLIT_TAIL");
    return e ? atoi(e) : -1;
  }();
  static const int shape = [] {
    const char* e = getenv("GQE_SPLIT_SHAPE");
    return e ? atoi(e) : 0;
  }();
  fa.force_fw = (shape == 8 && d == 128) ? 8 : 16;
  fa.split.t = ctx->split_t;
  fa.split.p = ctx->params;
  fa.split.m = ctx->m;
  fa.split.v = ctx->v;
  fa.split.stamp = stamp;
  fa.split.epoch = ctx->split_epoch;
  fa.split.done = reinterpret_cast<int32_t*>(ctx->ws + L.counter_off + 128);
  fa.split.progress = reinterpret_cast<int32_t*>(ctx->ws + L.progress_off);
  fa.split.b1 = ctx->split_b1;
  fa.split.b2 = ctx->split_b2;
  fa.split.eps = ctx->split_eps;
  static const int waves_env = [] {
    const char* e = getenv("GQE_SPLIT_WAVES");
    return e ? atoi(e) : 0;
  }();
  fa.split.waves = (waves_env > 0 && waves_env <= fa.force_fw) ? waves_env : fa.force_fw;
  const int total_blocks = ctx->split_t.blk_begin[ctx->split_t.n];
  // (a stream of more than ~220 MB of p + m + v outlasts the tiles by far — bio-synth d = 256, 298 MB: lead 96 / 128 / 160 / 192:
  // 161.5 / 158.5 / 160.5 / 153.4 us per step — so more CUs stream from the first microsecond; at 150 MB 64-128 are equal)
  const int lead_default = ctx->split_stream_bytes > (220ll << 20) ? 192 : 96;
  fa.split.lead = std::max(0, std::min(lead_env >= 0 ? lead_env : lead_default, GQE_SPLIT_MAX_RIDERS / 2));
  // (one tail rider per CU: full Bilinear 99 -> 92 us, B = 256 69 -> 62 us against 64 of them; the headline is indifferent)
  const int tail = std::max(0, std::min(tail_env >= 0 ? tail_env : 256, GQE_SPLIT_MAX_RIDERS / 2));
  fa.split.blocks = std::max(1, fa.split.lead + tail);
  fa.split.lead = std::min(fa.split.lead, fa.split.blocks);
  // a lead rider streams for the whole launch (~35 us), a tail rider for what is left of it when its CU becomes free: ranges 4 : 1
  static const int share_env = [] {
    const char* e = getenv("GQE_SPLIT_SHARE");
    return e && atoi(e) > 0 ? atoi(e) : 4;
  }();
  fa.split.share = share_env;
  const int units = fa.split.lead * fa.split.share + (fa.split.blocks - fa.split.lead);
  fa.split.per = (total_blocks + units - 1) / units;
  fa.split.tiles = fa.plan.tiles;
  static const bool no_ride = getenv("GQE_SPLIT_DEBUG_NORIDE") != nullptr;   // timing experiments, WRONG results: the riders do nothing
  if (no_ride) fa.split.waves = 0;
  // (stopping the riders with the tiles and finishing the stream next to the units of the second launch: built, measured, off —
  // DESIGN.md §3; GQE_SPLIT_STOP=1 turns it on for experiments)
  static const int stop_env = [] {
    const char* e = getenv("GQE_SPLIT_STOP");
    return e ? atoi(e) : 0;
  }();
  fa.split.stop = stop_env;
  ctx->split_ride = fa.split;
  std::vector<GqeMatStep> pending;
  pending.swap(ctx->mat_pending);
  for (size_t k = 0; k + 1 < pending.size(); ++k)
    HIP_TRY(ctx, gqe_launch_matstep(pending[k], ctx->params, ctx->grads, ctx->m, ctx->v, d, ctx->mat_b1, ctx->mat_b2, ctx->mat_eps, st));
  GqeMatStep none;
  memset(&none, 0, sizeof none);
  int rc = timing_begin(ctx, 1, st);   // (the slot of the pair GEMM's / matrix step's own launch: what is left of them on the stream)
  if (rc != GQE_OK) return rc;
  HIP_TRY(ctx, gqe_launch_prestep(pending.empty() ? none : pending.back(), ctx->params, ctx->grads, ctx->m, ctx->v, d, ctx->mat_b1, ctx->mat_b2,
                                  ctx->mat_eps, sg, fa.split, d_idx, stamp, st));
  rc = timing_end(ctx, 1, st);
  if (rc != GQE_OK) return rc;
  ctx->split_launched = true;
  return GQE_OK;
}

int run_queries(gqe_ctx* ctx, const gqe_batch* batches, int32_t n_batches, const int32_t* idx, int64_t n_idx,
                int32_t idx_on_device, bool bwd, float* losses, float* pos, float* neg, void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  if (!batches || n_batches < 1 || n_batches > GQE_MAX_BATCHES) return fail(ctx, GQE_ERR_ARG, "n_batches must be in [1,%d]", GQE_MAX_BATCHES);
  if (!idx || n_idx < 1) return fail(ctx, GQE_ERR_ARG, "no indices given");
  if (!ctx->params) return fail(ctx, GQE_ERR_STATE, "gqe_bind_arena has not been called");
  if (bwd && !ctx->grads) return fail(ctx, GQE_ERR_STATE, "no gradient arena bound");
  if (bwd && !losses) return fail(ctx, GQE_ERR_ARG, "losses buffer is NULL");
  if (!ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_bind_workspace has not been called");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  {   // a pair GEMM still waiting for its optimiser pass reads the scratch rows this call is about to overwrite
    int rcf = flush_ride(ctx, st);
    if (rcf != GQE_OK) return rcf;
    // matrices a split step left pending: this call contracts with them (a split step's own first launch carries them instead)
    if (!ctx->split_active) rcf = flush_split(ctx, st);
    if (rcf != GQE_OK) return rcf;
  }
  const int d = ctx->cfg.dim;
  const Layout& L = ctx->lay;
  const int macros_sq = ((d + GQE_GEMM_MT - 1) / GQE_GEMM_MT) * ((d + GQE_GEMM_MT - 1) / GQE_GEMM_MT);

  // ---- validate everything and resolve the formula descriptors before anything is enqueued ----
  ++ctx->call_stamp;  // descriptor slots used by this call are not evicted by it
  std::vector<int> fid(n_batches);
  int64_t entries = 0;
  std::vector<int> touched_tables;
  for (int bi = 0; bi < n_batches; ++bi) {
    const gqe_batch& s = batches[bi];
    if (s.n_queries < 1) return fail(ctx, GQE_ERR_ARG, "batch %d: empty batch (n_queries=%d)", bi, s.n_queries);
    int rc = formula_of(ctx, s, bi, &fid[bi]);
    if (rc != GQE_OK) return rc;
    const int na = ctx->formulas[fid[bi]].n_anchors;
    if (s.n_candidates < 0 || (bwd && s.n_candidates != 0)) return fail(ctx, GQE_ERR_ARG, "batch %d: candidate lists are for gqe_forward only", bi);
    const int64_t need_idx = s.n_candidates > 0
        ? (int64_t)s.idx_offset + (int64_t)na * s.n_queries + s.n_queries + 1 + s.n_candidates
        : (int64_t)s.idx_offset + (int64_t)(na + (bwd ? 2 : 1)) * s.n_queries;
    if (s.idx_offset < 0 || need_idx > n_idx) return fail(ctx, GQE_ERR_ARG, "batch %d: index range [%d,%lld) exceeds the %lld indices given", bi, s.idx_offset, (long long)need_idx, (long long)n_idx);
    entries += (int64_t)(2 + na) * s.n_queries;
    if (bwd) {
      touched_tables.push_back(table_of(ctx, s.target_table));
      for (int i = 0; i < na; ++i) touched_tables.push_back(table_of(ctx, s.anchor_table[i]));
    }
  }
  const size_t idx_bytes = idx_on_device ? 0 : (size_t)n_idx * sizeof(int32_t);
  if (idx_bytes > L.idx_cap) return fail(ctx, GQE_ERR_WORKSPACE, "index feed of %lld entries exceeds the bound workspace (%lld queries)", (long long)n_idx, (long long)ctx->cap_queries);
  if (bwd && ctx->world > 1) {
    if (ctx->entries_used) return fail(ctx, GQE_ERR_STATE, "exchange mode: one gqe_margin_fwd_bwd per optimiser step (gradients of the previous call are still pending)");
    const int64_t slab = ctx->slab_hint ? ctx->slab_hint : entries;
    if (entries > slab) return fail(ctx, GQE_ERR_ARG, "this call produces %lld gradient entries, more than the reserved slab of %lld", (long long)entries, (long long)slab);
    const GqeSpans sp = dense_spans(ctx);
    if (sp.n < 0) return fail(ctx, GQE_ERR_STATE, "exchange mode: the non-table parameters form more than 8 spans of the arena");
    if (slab_entries(ctx, slab, sp.total) * ctx->world > L.max_entries)
      return fail(ctx, GQE_ERR_WORKSPACE, "gradient contribution buffer too small for %d ranks x %lld entries", ctx->world, (long long)slab);
  }
  const bool shard = ctx->shard_on;
  if (shard) {
    if (!idx_on_device) return fail(ctx, GQE_ERR_ARG, "row-sharded mode: the index feed is the device-resident position feed of gqe_shard_plan");
    if (n_idx > L.shard_cap_send) return fail(ctx, GQE_ERR_WORKSPACE, "row-sharded mode: %lld indices exceed the fetched-row buffer (%lld rows)", (long long)n_idx, (long long)L.shard_cap_send);
  }
  if (bwd && !shard && ctx->entries_used + entries > L.max_entries)
    return fail(ctx, GQE_ERR_WORKSPACE, "gradient contribution buffer full (%lld + %lld > %lld entries): step or "
                "gqe_materialize_grads first", (long long)ctx->entries_used, (long long)entries, (long long)L.max_entries);

  int rc;
  // ---- the operand-ordered copies of the matrices, when something other than the library's optimiser changed them ----
  static const bool always_retile = getenv("GQE_ALWAYS_RETILE") != nullptr;   // (tests: the copies never trusted)
  if (ctx->tiles_dirty || always_retile) {
    rc = retile(ctx, st);
    if (rc != GQE_OK) return rc;
  } else {
    static const bool check = getenv("GQE_CHECK_TILES") != nullptr;
    if (check) {
      rc = check_tiles(ctx, st);
      if (rc != GQE_OK) return rc;
    }
  }
  // ---- new / replaced formula descriptors -> device table (rare): contiguous runs of stale slots, one copy each ----
  if (!ctx->formulas_dirty.empty()) {
    std::vector<int>& dirty = ctx->formulas_dirty;
    std::sort(dirty.begin(), dirty.end());
    dirty.erase(std::unique(dirty.begin(), dirty.end()), dirty.end());
    for (size_t a = 0; a < dirty.size();) {
      size_t b = a + 1;
      while (b < dirty.size() && dirty[b] == dirty[b - 1] + 1) ++b;
      const size_t off = sizeof(GqeDevFormula) * (size_t)dirty[a];
      const size_t bytes = sizeof(GqeDevFormula) * (b - a);
      RingSlot* slot;
      rc = ring_acquire(ctx, bytes, &slot);
      if (rc != GQE_OK) return rc;
      memcpy(slot->host, ctx->formulas.data() + dirty[a], bytes);
      HIP_TRY(ctx, hipMemcpyAsync(ctx->ws + L.formula_off + off, slot->host, bytes, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipEventRecord(slot->done, st));
      slot->in_flight = true;
      a = b;
    }
    dirty.clear();
  }
  // ---- host index feed -> device, on the side stream ----
  const int32_t* d_idx = idx;
  int buf = -1;
  if (!idx_on_device) {
    if (!ctx->up) {
      HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->up, hipStreamNonBlocking));
      for (int k = 0; k < 2; ++k) {
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->plan_ready[k], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->plan_free[k], hipEventDisableTiming));
      }
    }
    RingSlot* slot;
    rc = ring_acquire(ctx, idx_bytes, &slot);
    if (rc != GQE_OK) return rc;
    memcpy(slot->host, idx, idx_bytes);
    buf = ctx->plan_buf;
    ctx->plan_buf ^= 1;
    // lazy Adam: the pending margin call's index feed may still live in this staging buffer (the sparse optimiser
    // launch walks it); once it is overwritten the step has to fall back to the full pass
    if (ctx->feed_valid && ctx->feed_buf == buf) ctx->feed_valid = false;
    char* dev = ctx->ws + (size_t)buf * L.idx_cap;
    if (ctx->plan_free_set[buf]) HIP_TRY(ctx, hipStreamWaitEvent(ctx->up, ctx->plan_free[buf], 0));
    HIP_TRY(ctx, hipMemcpyAsync(dev, slot->host, idx_bytes, hipMemcpyHostToDevice, ctx->up));
    HIP_TRY(ctx, hipEventRecord(slot->done, ctx->up));
    slot->in_flight = true;
    HIP_TRY(ctx, hipEventRecord(ctx->plan_ready[buf], ctx->up));
    HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->plan_ready[buf], 0));
    d_idx = reinterpret_cast<const int32_t*>(dev);
  }

  if (ctx->lazy && !shard) {   // (row-sharded: the owners bring the rows up to date before they serve them, gqe_shard_step)
    // rows this call reads must be current: replay their deferred zero-gradient Adam steps first
    std::vector<SavedFeed> feed;
    const bool dirty = lazy_any_dirty(ctx);
    if (dirty || bwd) build_feed(ctx, batches, n_batches, bwd, fid, feed);
    // (gqe_lazy_prefetch: the previous optimiser step already brought this very feed's rows up to date)
    const bool caught = idx_on_device && ctx->caught_idx == idx && ctx->caught_n_idx == n_idx &&
                        ctx->caught_sig == feed_signature(batches, n_batches, bwd);
    ctx->caught_idx = nullptr;
    if (dirty && !caught) {
      GqeRowsArgs ra;
      lazy_rows_args(ctx, ra, st);
      ra.idx = d_idx;
      rc = timing_begin(ctx, 4, st);
      if (rc != GQE_OK) return rc;
      for (const SavedFeed& sf : feed) {
        ra.segs = sf.segs;
        HIP_TRY(ctx, gqe_launch_rows(ra));
      }
      rc = timing_end(ctx, 4, st);
      if (rc != GQE_OK) return rc;
    }
    if (bwd) {
      ctx->feed_valid = ctx->entries_used == 0;  // lists pending from an earlier call: the feed no longer names them all
      ctx->feed.swap(feed);
      ctx->feed_idx = d_idx;
      ctx->feed_buf = buf;
    }
  }

  GqeFusedArgs fa;
  fa.formulas = reinterpret_cast<const GqeDevFormula*>(ctx->ws + L.formula_off);
  fa.params = ctx->params;
  fa.grads = ctx->grads;
  fa.ws = reinterpret_cast<float*>(ctx->ws);
  fa.idx = d_idx;
  fa.d = d;
  fa.tile_loss = reinterpret_cast<float*>(ctx->ws + L.tloss_off);
  fa.pos = pos;
  fa.neg = neg;
  fa.inter_min = is_min(ctx) ? 1 : 0;
  fa.bwd = bwd;
  fa.prof = ctx->prof;
  fa.stream = st;
  fa.head = reinterpret_cast<int32_t*>(ctx->ws + L.head_off);
  fa.next = reinterpret_cast<int32_t*>(ctx->ws + L.next_off);
  fa.contrib = reinterpret_cast<float*>(ctx->ws + (shard ? L.shard_csend : L.contrib_off));
  fa.fetched = shard ? reinterpret_cast<const float*>(ctx->ws + L.shard_fetch) : nullptr;
  fa.contrib_bag = reinterpret_cast<float*>(ctx->ws + L.contrib_off);
  fa.bag_shift = shard ? L.shard_cap_recv + L.shard_cap_send : 0;
  memset(&fa.bags, 0, sizeof fa.bags);
  for (size_t k = 0; k < ctx->bags.size(); ++k) {
    fa.bags.ptr[k] = ctx->bags[k].ptr;
    fa.bags.ids[k] = ctx->bags[k].ids;
    fa.bags.max_len = std::max(fa.bags.max_len, ctx->bags[k].max_len);
  }
  fa.link_contrib = reinterpret_cast<int32_t*>(ctx->ws + L.linkc_off);
  fa.link_counter = reinterpret_cast<int32_t*>(ctx->ws + L.counter_off);
  fa.max_entries = (int32_t)L.max_entries;
  fa.hot = hot_args(ctx, true);
  memset(&fa.split, 0, sizeof fa.split);
  fa.force_fw = 0;
  if (bwd && !ctx->bags.empty()) ctx->links_used = true;

  // ---- launches of <= GQE_LAUNCH_BATCHES batches; per-call data travels as kernel arguments ----
  int64_t entry = shard ? 0 : ctx->entries_used;
  if (bwd && shard) {
    if (ctx->shard_sent) return fail(ctx, GQE_ERR_STATE, "row-sharded mode: one gqe_margin_fwd_bwd per optimiser step (send the contributions and gqe_shard_link first)");
    ctx->shard_sent = true;   // (queries whose hinge is inactive write zero contributions themselves: no memset)
  }
  if (bwd && ctx->world > 1) {
    // this rank's slab of the gathered entry space; entries that are not pushed (inactive hinge) must read -1
    const int64_t slab = ctx->slab_hint ? ctx->slab_hint : entries;
    ctx->step_entries = slab;
    ctx->step_slab = slab_entries(ctx, slab, dense_spans(ctx).total);
    ctx->step_exported = false;
    ctx->imported = false;
    entry = (int64_t)ctx->rank * ctx->step_slab;
    HIP_TRY(ctx, hipMemsetAsync(ctx->ws + L.rows_off + sizeof(int32_t) * (size_t)entry, 0xff, sizeof(int32_t) * (size_t)slab, st));
  }
  for (int b0 = 0; b0 < n_batches; b0 += GQE_LAUNCH_BATCHES) {
    const int nb = std::min(GQE_LAUNCH_BATCHES, n_batches - b0);
    GqeDynPlan& P = fa.plan;
    memset(&P, 0, sizeof P);
    P.n_batches = nb;
    P.first = b0 == 0;
    P.total_index = n_batches;
    for (int k = 0; k < GQE_LAUNCH_BATCHES; ++k) P.tile_begin[k] = P.unit_begin[k] = 0x7fffffff;
    int64_t scratch = (int64_t)(L.scratch_off / sizeof(float));
    // Entries / scratch rows are assigned in the caller's batch order; the TILES are laid out longest batch first: the
    // hardware starts workgroups in blockIdx order, and with more tiles than workgroup slots (one 1024-thread
    // workgroup per CU at d >= 128) the tiles that start late should be the short chain tiles (~8 us), not the
    // ~30 us three-branch intersection tiles.
    bool any_candidates = false;
    // gqe_set_deferred_gemm: may this launch's pair GEMM wait for the Adam pass (decided below, once its units are counted)?
    const bool ride_candidate = bwd && (ctx->defer_gemm || ctx->split_active) && n_batches <= GQE_LAUNCH_BATCHES && !shard && ctx->world == 1 && d % 64 == 0 &&
                                !ctx->ordered_sums && (!ctx->prof || ctx->split_active);   // (lazy Adam: the units ride in the step's row launch)
    // pair-GEMM units cover kmul x GQE_GEMM_KCHUNK queries: with thousands of units (large batches) a unit walks several
    // chunks before its one atomic pass over the 64 x 64 block — the units of a block all add into the same lines
    int kmul = 1;
    if (bwd) {
      long long units1 = 0;
      for (int k = 0; k < nb; ++k)
        units1 += (long long)ctx->formulas[fid[b0 + k]].n_jobs * ((align_up(batches[b0 + k].n_queries, GQE_TQ) + GQE_GEMM_KCHUNK - 1) / GQE_GEMM_KCHUNK) * macros_sq;
      while (kmul < 8 && units1 / (kmul * 2) >= GQE_GEMM_MIN_UNITS) kmul *= 2;   // ... but the launch keeps >= GQE_GEMM_MIN_UNITS units
      // units that will ride in the Adam pass's launch (gqe_set_deferred_gemm) have tens of microseconds of slack, and what
      // they cost the pass is their atomic rows: at least 256 queries per unit (one box, 128 / 256 / 512 per unit: headline step
      // 80.0 / 78.2 / 78.3 us, full Bilinear 104.3 / 99.6 / 99.2, B = 1024 102.2 / 98.3 / 98.5, d = 256 182.9 / 177.2 / 183.0,
      // d = 64 — a 22 us pass — 47.2 / 47.5 / 55.3: 512 outlasts a short pass)
      // (a split step's second launch has no 45-us stream to hide behind: its units are the launch's critical chain and keep
      // the 128-query chunks)
      if (ride_candidate && !ctx->split_active && kmul < 2) kmul = 2;
      // ... until there are so many of them that their atomic rows outweigh the chain: 256 queries per unit from 640 units on
      // (round 6, one box: full Bilinear B = 512, 768 units: 85.5 -> 84.5 us per step; the headline's 320 units: 68.1 -> 69.8 the
      // other way, full Bilinear B = 256, 384 units: 69.5 -> 71.5)
      if (ride_candidate && ctx->split_active && kmul < 2 && units1 >= 640) kmul = 2;
      static const int forced = [] {   // GQE_DEBUG_GEMM_KMUL: tuning runs only
        const char* e = getenv("GQE_DEBUG_GEMM_KMUL");
        return e ? atoi(e) : 0;
      }();
      if (forced == 1 || forced == 2 || forced == 4 || forced == 8) kmul = forced;
    }
    P.pad[0] = kmul;
    // a split step's units (gqe_split_rows_kernel) are their launch's critical chain: 64 queries per unit when that keeps the
    // launch under GQE_RIDE_MAX_UNITS (GQE_SPLIT_UNIT_Q: tuning runs)
    int unit_q = 0;
    if (bwd && ride_candidate && ctx->split_active) {
      static const int uq = [] {
        const char* e = getenv("GQE_SPLIT_UNIT_Q");
        return e ? atoi(e) : 0;
      }();
      if (uq >= 16 && uq % 16 == 0) {
        long long u = 0;
        for (int k = 0; k < nb; ++k)
          u += (long long)ctx->formulas[fid[b0 + k]].n_jobs * ((align_up(batches[b0 + k].n_queries, GQE_TQ) + uq - 1) / uq) * macros_sq;
        if (u <= GQE_RIDE_MAX_UNITS) unit_q = uq;
      }
    }
    P.pad[2] = unit_q;
    GqeDynBatch tmp[GQE_LAUNCH_BATCHES];
    int tiles_of[GQE_LAUNCH_BATCHES], units_of[GQE_LAUNCH_BATCHES], cost[GQE_LAUNCH_BATCHES], order[GQE_LAUNCH_BATCHES];
    for (int k = 0; k < nb; ++k) {
      const gqe_batch& s = batches[b0 + k];
      const GqeDevFormula& f = ctx->formulas[fid[b0 + k]];
      GqeDynBatch& b = tmp[k];
      memset(&b, 0, sizeof b);
      b.formula = fid[b0 + k];
      b.B = s.n_queries;
      b.Bpad = (int)align_up(s.n_queries, GQE_TQ);
      b.idx_offset = s.idx_offset;
      b.out_offset = s.out_offset;
      b.has_neg = bwd ? 1 : 0;
      b.entry_base = entry;
      b.scratch_base = scratch;
      b.margin = s.margin;
      b.loss_weight = s.loss_weight;
      b.inv_B = 1.0f / (float)s.n_queries;
      b.grad_scale = s.loss_weight / (float)s.n_queries;
      b.loss_index = b0 + k;
      b.n_candidates = s.n_candidates;
      b.n_anchors = f.n_anchors;
      tiles_of[k] = b.Bpad / GQE_TQ;
      units_of[k] = 0;
      // relative length of one tile's dependent chain: contraction phases dominate (intersections: Pre / Post and
      // their transposes; full Bilinear: one per hop and score side), then the rows gathered / scattered
      const bool chain = f.qtype <= GQE_Q_3CHAIN;
      const bool bil = ctx->cfg.decoder == GQE_DEC_BILINEAR;
      int hops = 0;
      for (int i = 0; i < GQE_MAX_BRANCH; ++i) hops += f.n_hops[i];
      cost[k] = 2 + f.n_anchors + (chain ? (bil ? 4 * hops : 0) : (is_mlp(ctx) ? 6 + 2 * f.n_anchors : 2) + (bil ? 2 * (hops + f.n_final) : 0));
      if (bwd) {
        entry += (int64_t)(2 + f.n_anchors) * s.n_queries;
        scratch += (int64_t)f.n_slots * b.Bpad * d;
        const int cq = unit_q ? unit_q : GQE_GEMM_KCHUNK * kmul;
        units_of[k] = f.n_jobs * ((b.Bpad + cq - 1) / cq) * macros_sq;
      } else if (s.n_candidates > 0 && bil && chain) {
        // candidate lists of a full-Bilinear chain: the projection runs on the candidate side, so the batch's tiles cover its
        // candidates (16 per tile, one [16 x d] . [d x d] contraction per hop on the matrix cores); the query of every
        // candidate goes to the scratch region first
        b.expand = 1;
        tiles_of[k] = (s.n_candidates + GQE_TQ - 1) / GQE_TQ;
        HIP_TRY(ctx, gqe_launch_expand_ptr(d_idx + s.idx_offset + (int64_t)f.n_anchors * s.n_queries, s.n_queries, s.n_candidates,
                                           reinterpret_cast<int32_t*>(ctx->ws) + scratch, st));
        scratch += (int64_t)align_up((size_t)s.n_candidates, 64);
      } else if (s.n_candidates > 0) {
        // evaluation against candidate lists: the fused kernel leaves one record (d + 4 floats) per query in the
        // scratch region, the scoring kernel covers the batch's candidates in blocks
        scratch += (int64_t)b.Bpad * (d + 4);
        units_of[k] = (s.n_candidates + GQE_EVAL_BLOCK - 1) / GQE_EVAL_BLOCK;
        any_candidates = true;
      }
      order[k] = k;
    }
    std::stable_sort(order, order + nb, [&](int a, int c) { return cost[a] > cost[c]; });
    for (int pos = 0; pos < nb; ++pos) {
      GqeDynBatch& b = P.b[pos];
      b = tmp[order[pos]];
      b.tile_begin = P.tile_begin[pos] = P.tiles;
      b.unit_begin = P.unit_begin[pos] = P.units;
      P.tiles += tiles_of[order[pos]];
      P.units += units_of[order[pos]];
    }
    // few tiles: what the next kernels read is written through (gqe_fused.h, vstore_wt); thousands: plain stores
    P.pad[1] = (bwd && P.tiles <= GQE_FW8_MIN_TILES) ? 1 : 0;
    if ((size_t)(scratch * (int64_t)sizeof(float)) > L.scratch_off + L.scratch_cap)
      return fail(ctx, GQE_ERR_WORKSPACE, "workspace too small for this call (bound for %lld queries, %d batches)",
                  (long long)ctx->cap_queries, ctx->cap_batches);
    if (bwd && ctx->split_active) {
      rc = split_first_launch(ctx, batches, n_batches, fid, d_idx, fa, st);
      if (rc != GQE_OK) return rc;
    }
    rc = timing_begin(ctx, 0, st);
    if (rc != GQE_OK) return rc;
    HIP_TRY(ctx, gqe_launch_fused(ctx->cfg.decoder, is_mlp(ctx) ? 1 : 0, fa));
    if (bwd && fa.hot.slot && fa.hot.sub)   // hot word rows: their sub-lists into their accumulators (inside the fused launch's timing slot)
      HIP_TRY(ctx, gqe_launch_hot_gather(fa.hot, fa.next, fa.contrib_bag, fa.link_contrib, fa.max_entries, d, st));
    rc = timing_end(ctx, 0, st);
    if (rc != GQE_OK) return rc;
    if (any_candidates) HIP_TRY(ctx, gqe_launch_eval_score(fa, ctx->cfg.decoder, pos));
    if (bwd) {
      // deferred matrix gradients + the finalize block that turns per-tile hinge sums into losses[]
      // gqe_set_deferred_gemm: one launch of at most a few thousand units, and nobody reads the dense gradient before the
      // optimiser does — the units wait for the Adam pass and run in front of its chunks (GqeGemmRide)
      // (a launch without matrix jobs — chains of the element-wise decoders: the reference's whole edge-only burn-in phase —
      // still defers: its finalize block rides, and the step loses a launch that did nothing else)
      // (more units than that are MFMA work of their own — unless the context's tables are beyond the Infinity Cache: next to
      // their long non-temporal pass the units ride SPREAD through the grid, GqeGemmRide.spread; run_opt launches them alone
      // in front of any other pass)
      const bool ride = ride_candidate && (P.units <= GQE_RIDE_MAX_UNITS || (P.units <= GQE_RIDE_MAX_UNITS_SPREAD && big_tables(ctx) && !ctx->lazy && !ctx->split_active));
      if (ride) {
        ctx->ride_fa = fa;
        ctx->ride_losses = losses;
        ctx->ride_pending = true;
      } else {
        rc = timing_begin(ctx, 1, st);
        if (rc != GQE_OK) return rc;
        HIP_TRY(ctx, gqe_launch_pair_gemm(fa, losses));
        rc = timing_end(ctx, 1, st);
        if (rc != GQE_OK) return rc;
      }
    }
  }
  if (bwd && shard) {
    ctx->shard_tables = touched_tables;
    for (int t : touched_tables)
      if (is_bag_table(ctx, t)) ctx->tables[(size_t)t].pending = true;   // bag tables: linked locally by this launch
  }
  if (bwd && !shard) {
    ctx->entries_used = ctx->world > 1 ? ctx->step_slab * ctx->world : entry;
    for (int t : touched_tables) ctx->tables[t].pending = true;
  }
  if (buf >= 0) {
    HIP_TRY(ctx, hipEventRecord(ctx->plan_free[buf], st));
    ctx->plan_free_set[buf] = true;
  }
  ctx->split_buf = ctx->split_launched ? buf : -1;   // (a split step's second launch reads the staged feed again: it re-records the event)
  return GQE_OK;
}

// every d x d tensor outside the tables is kept in operand order next to its parameter (whether a formula names it yet or not)
void universe_tiles(const gqe_ctx* ctx, GqeDevSeg& g) {
  g.tile = g.tile_T = nullptr;
  const int64_t d = ctx->cfg.dim;
  if (g.is_table || g.numel != d * d) return;
  const int64_t t = tile_of(ctx, g.offset);
  if (t < 0) return;
  g.tile = reinterpret_cast<float*>(ctx->ws) + t;
  g.tile_T = g.tile + ctx->lay.tile_floats;
}

int universe_index(gqe_ctx* ctx, int64_t offset, int64_t numel, int table) {
  for (size_t i = 0; i < ctx->universe.size(); ++i)
    if (ctx->universe[i].offset == offset && ctx->universe[i].numel == numel) return (int)i;
  if ((int)ctx->universe.size() >= ctx->cap_tensors) return -1;
  const int d = ctx->cfg.dim;
  GqeDevSeg g;
  memset(&g, 0, sizeof g);
  g.offset = offset;
  g.numel = numel;
  g.is_table = table >= 0 ? 1 : 0;
  g.table_index = table;
  if (table >= 0) {
    const int tpr = d / 4, rpc = (64 / tpr) * GQE_WAVES;   // a row's threads stay inside one wave (gqe_kernels.hip, opt_body)
    g.rows = ctx->tables[table].rows;
    g.head_base = ctx->tables[table].head_base;
    g.n_chunks = (g.rows + rpc - 1) / rpc;
  } else {
    g.n_chunks = (numel + GQE_OPT_CHUNK - 1) / GQE_OPT_CHUNK;
    universe_tiles(ctx, g);
  }
  ctx->universe.push_back(g);
  return (int)ctx->universe.size() - 1;
}

// mode: GQE_OPT_ADAM / SGD / ZERO / MATERIALIZE (gqe_dev.h)
int run_opt(gqe_ctx* ctx, int mode_in, const gqe_segment* segs, int32_t n_segs, float lr, float b1, float b2, float eps,
            void* stream);

#define GQE_OPT_FLUSH 4  // internal: lazy Adam, bring every lagging row of the dirty tables to its table's step

// the lazy step's row launch with the pending pair GEMM in front of its row groups (gqe_rows_ride_kernel)
int lazy_rows_ride(gqe_ctx* ctx, const GqeRowsArgs& ra) {
  GqeGemmRide r;
  r.plan = ctx->ride_fa.plan;
  r.formulas = ctx->ride_fa.formulas;
  r.ws = ctx->ride_fa.ws;
  r.tile_loss = ctx->ride_fa.tile_loss;
  r.losses = ctx->ride_losses;
  r.spread = 0;
  const hipError_t e = gqe_launch_rows_ride(ra, r);
  if (e == hipErrorInvalidValue) {   // (a feed offset beyond 32 bits: the two launches instead — the caller's ride flag stays harmless)
    const int rc = flush_ride(ctx, ra.stream);
    if (rc != GQE_OK) return rc;
    HIP_TRY(ctx, gqe_launch_rows(ra));
    return GQE_OK;
  }
  HIP_TRY(ctx, e);
  ctx->ride_pending = false;
  ++ctx->rides;
  return GQE_OK;
}

int run_opt(gqe_ctx* ctx, int mode_in, const gqe_segment* segs, int32_t n_segs, float lr, float b1, float b2, float eps,
            void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  const bool flush = mode_in == GQE_OPT_FLUSH;
  const int mode = flush ? GQE_OPT_ADAM : mode_in;
  if (!ctx->params || !ctx->grads) return fail(ctx, GQE_ERR_STATE, "parameter / gradient arenas not bound");
  if (mode == GQE_OPT_ADAM && (!ctx->m || !ctx->v)) return fail(ctx, GQE_ERR_STATE, "Adam moment arenas not bound");
  if (!ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_bind_workspace has not been called");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int d = ctx->cfg.dim;
  // a deferred pair GEMM rides in this pass if it is a plain eager Adam pass (decided at the launch below); everything else
  // needs the matrix gradients in place first
  const bool may_ride = ctx->ride_pending && mode_in == GQE_OPT_ADAM && !ctx->ordered_sums && ctx->world == 1;   // (lazy: decided where the sparse launch is)
  if (ctx->ride_pending && !may_ride) {
    const int rcf = flush_ride(ctx, st);
    if (rcf != GQE_OK) return rcf;
  }
  if (!ctx->split_launched) {   // matrices a split step left pending: stepped before anything else touches their gradient
    const int rcf = flush_split(ctx, st);
    if (rcf != GQE_OK) return rcf;
  }
  if (flush && !lazy_any_dirty(ctx)) return GQE_OK;
  if (ctx->lazy && !flush && lazy_any_dirty(ctx) &&
      (mode == GQE_OPT_SGD || (mode == GQE_OPT_ADAM && ctx->lz_hyper &&
                               (lr != ctx->lz_lr || b1 != ctx->lz_b1 || b2 != ctx->lz_b2 || eps != ctx->lz_eps)))) {
    // deferred steps were recorded with the previous hyper-parameters (or SGD would move rows that still owe
    // Adam steps): settle them first
    int rcf = run_opt(ctx, GQE_OPT_FLUSH, nullptr, 0, ctx->lz_lr, ctx->lz_b1, ctx->lz_b2, ctx->lz_eps, stream);
    if (rcf != GQE_OK) return rcf;
  }
  if (flush) {
    lr = ctx->lz_lr;
    b1 = ctx->lz_b1;
    b2 = ctx->lz_b2;
    eps = ctx->lz_eps;
  }
  if (!flush) ctx->caught_idx = nullptr;   // rows move on: what an earlier step caught up is no longer current
  GqeOptArgs oa;
  oa.lazy = false;
  memset(&oa.lz, 0, sizeof oa.lz);
  memset(&oa.coef, 0, sizeof oa.coef);
  memset(oa.active.group, 0xFF, sizeof oa.active.group);
  oa.act = nullptr;
  oa.n_act = 0;
  std::vector<char> seen(ctx->tables.size(), 0);
  bool lists = false;
  std::vector<int> ustep;  // per universe entry: the step count this pass applies (0 = the tensor is not stepped)
  auto activate = [&](int64_t offset, int64_t numel, int step, int table) -> int {
    const int ui = universe_index(ctx, offset, numel, table);
    if (ui < 0)
      return fail(ctx, GQE_ERR_ARG, "more than %d distinct parameter tensors: raise the limit with gqe_set_limits (before gqe_workspace_bytes)",
                  ctx->cap_tensors);
    if ((size_t)ui >= ustep.size()) ustep.resize((size_t)ui + 1, 0);
    if (ustep[ui]) return fail(ctx, GQE_ERR_ARG, "tensor at offset %lld listed twice", (long long)offset);
    if (table >= 0) {
      seen[table] = 1;
      lists = lists || ctx->tables[table].pending;
    }
    ustep[ui] = mode == GQE_OPT_ADAM ? std::max(step, 1) : 1;
    return GQE_OK;
  };
  int rc;
  if (flush) {
    for (size_t t = 0; t < ctx->tables.size(); ++t)
      if (ctx->tables[t].dirty) {
        rc = activate(ctx->tables[t].offset, ctx->tables[t].rows * d, 1, (int)t);
        if (rc != GQE_OK) return rc;
      }
    lists = false;
  } else if (mode == GQE_OPT_MATERIALIZE) {
    bool any = false;
    for (size_t t = 0; t < ctx->tables.size(); ++t) {
      bool wanted = n_segs == 0;   // segs given (gqe_materialize_tables): only those tables
      for (int i = 0; i < n_segs; ++i) wanted = wanted || segs[i].offset == ctx->tables[t].offset;
      if (ctx->tables[t].pending && wanted) {
        rc = activate(ctx->tables[t].offset, ctx->tables[t].rows * d, 1, (int)t);
        if (rc != GQE_OK) return rc;
        any = true;
      }
    }
    for (size_t t = 0; t < ctx->tables.size(); ++t) {   // the dense gradient of these tables becomes authoritative
      bool wanted = n_segs == 0;
      for (int i = 0; i < n_segs; ++i) wanted = wanted || segs[i].offset == ctx->tables[t].offset;
      if (wanted) ctx->tables[t].dense = true;
    }
    if (!any) {  // nothing (of what was asked for) pending
      if (n_segs == 0) ctx->entries_used = 0;
      return GQE_OK;
    }
  } else {
    if (!segs || n_segs < 1) return fail(ctx, GQE_ERR_ARG, "no segments given");
    if (n_segs > ctx->cap_tensors)
      return fail(ctx, GQE_ERR_ARG, "%d segments, but the ctx was sized for %d parameter tensors: raise the limit with gqe_set_limits (before gqe_workspace_bytes)",
                  n_segs, ctx->cap_tensors);
    for (int i = 0; i < n_segs; ++i) {
      const gqe_segment& s = segs[i];
      if (s.offset < 0 || (s.offset % 4) != 0 || s.numel < 1 || s.offset + s.numel > ctx->n_arena)
        return fail(ctx, GQE_ERR_ARG, "segment %d [%lld,+%lld) outside the arena or misaligned", i, (long long)s.offset, (long long)s.numel);
      int step = s.step;
      if (mode == GQE_OPT_ADAM && step < 1) step = ++ctx->adam_steps[s.offset];   // library-kept counter
      else if (mode == GQE_OPT_ADAM) ctx->adam_steps[s.offset] = step;
      const int t = table_of(ctx, s.offset);
      if (t >= 0 && s.numel != ctx->tables[t].rows * d) return fail(ctx, GQE_ERR_ARG, "segment %d covers a table only partly", i);
      rc = activate(s.offset, s.numel, step, t);
      if (rc != GQE_OK) return rc;
    }
    if (mode != GQE_OPT_ZERO)
      for (size_t t = 0; t < ctx->tables.size(); ++t)
        if (ctx->tables[t].pending && !seen[t])
          return fail(ctx, GQE_ERR_STATE, "table at offset %lld has pending gradients but is not among the stepped segments",
                      (long long)ctx->tables[t].offset);
  }
  // A stepped dense segment that COVERS a registered d x d matrix without being that matrix's own universe entry (one flat
  // segment over all dense parameters, a stacked [R, d, d] relation tensor): the pass moves the parameter but not its
  // operand-ordered copies (universe_tiles: numel == d * d only).  The copies are rebuilt in front of the next fused launch,
  // and such a pass never carries riding GEMM units (its chunks would read and zero a matrix gradient the units of the same
  // launch are still adding to: there is no order inside a launch).
  bool merged_matrix = false;
  if (mode == GQE_OPT_ADAM || mode == GQE_OPT_SGD)
    for (size_t ui = 0; ui < ustep.size() && !merged_matrix; ++ui) {
      if (!ustep[ui]) continue;
      const GqeDevSeg& u = ctx->universe[ui];
      if (u.is_table || u.tile) continue;
      auto it = std::lower_bound(ctx->matrices.begin(), ctx->matrices.end(), u.offset);
      merged_matrix = it != ctx->matrices.end() && *it < u.offset + u.numel;
      if (!merged_matrix && it != ctx->matrices.begin()) merged_matrix = *(it - 1) + (int64_t)d * d > u.offset;   // starts inside one
    }
  if (merged_matrix) ctx->tiles_dirty = true;
  if (ctx->universe_uploaded != ctx->universe.size()) {
    const size_t seg_bytes = sizeof(GqeDevSeg) * ctx->universe.size();
    RingSlot* slot;  // happens only when a tensor is stepped for the first time
    rc = ring_acquire(ctx, seg_bytes, &slot);
    if (rc != GQE_OK) return rc;
    memcpy(slot->host, ctx->universe.data(), seg_bytes);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ws + ctx->lay.seg_off, slot->host, seg_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipEventRecord(slot->done, st));
    slot->in_flight = true;
    ctx->universe_uploaded = ctx->universe.size();
  }
  // ---- Adam coefficients per stepped tensor.  torch.optim.Adam: step_size = lr / (1 - b1^t);
  // denom = sqrt(v) / sqrt(1 - b2^t) + eps  (python doubles).  Tensors with the same step count share a pair.
  const size_t nu = ctx->universe.size();
  ustep.resize(nu, 0);
  std::vector<float> uss(nu, lr), ubc(nu, 1.f);
  std::vector<int> distinct;
  for (size_t ui = 0; ui < nu; ++ui) {
    if (!ustep[ui]) continue;
    if (std::find(distinct.begin(), distinct.end(), ustep[ui]) == distinct.end()) distinct.push_back(ustep[ui]);
    if (mode == GQE_OPT_ADAM) {
      uss[ui] = (float)((double)lr / (1.0 - std::pow((double)b1, (double)ustep[ui])));
      ubc[ui] = (float)std::sqrt(1.0 - std::pow((double)b2, (double)ustep[ui]));
    }
  }
  // The pass is described to the kernels either in their arguments (<= GQE_MAX_SEGS tensors known to the ctx and
  // <= GQE_MAX_STEP_GROUPS distinct step counts: nothing is uploaded) or as a list of the active tensors that is
  // uploaded with the step (large schemas: dozens of relation types whose step counters diverge).
  // the FLUSH pass only replays deferred steps: it must neither read nor re-zero a materialised dense gradient
  // that is still waiting for its optimiser step
  oa.dense_tables = !flush && (any_dense(ctx) || mode == GQE_OPT_ZERO);
  // which tables: per universe entry (gqe_materialize_tables folds the replicated bag tables only — the owned shards of a
  // row-sharded run keep streaming 24 B per parameter)
  auto seg_dense = [&](size_t ui) {
    const GqeDevSeg& u = ctx->universe[ui];
    return u.is_table && oa.dense_tables && mode != GQE_OPT_MATERIALIZE && (mode == GQE_OPT_ZERO || ctx->tables[(size_t)u.table_index].dense);
  };
  const bool table_form = nu > GQE_MAX_SEGS || distinct.size() > GQE_MAX_STEP_GROUPS;
  std::vector<GqeActSeg> staging;
  const GqeActSeg* act_dev = reinterpret_cast<const GqeActSeg*>(ctx->ws + ctx->lay.act_off);
  bool prefix_overflow = false;  // the kernel-argument prefix counts chunks in 32 bits (2^31 chunks = 2 T parameters)
  auto emit = [&](auto keep, GqeOptActive& active, GqeStepCoef& coef, const GqeActSeg** act, int* n_act) -> long long {
    long long chunks = 0;
    memset(active.group, 0xFF, sizeof active.group);
    memset(&coef, 0, sizeof coef);
    *act = nullptr;
    *n_act = 0;
    if (!table_form) {
      int group_step[GQE_MAX_STEP_GROUPS], n_groups = 0;
      for (size_t ui = 0; ui <= nu; ++ui) active.begin[ui] = 0;
      for (size_t ui = 0; ui < nu; ++ui) {
        active.begin[ui] = (int32_t)chunks;
        active.begin[ui + 1] = (int32_t)chunks;
        if (!ustep[ui] || !keep(ui)) continue;
        int gi = 0;
        while (gi < n_groups && group_step[gi] != ustep[ui]) ++gi;
        if (gi == n_groups) {
          group_step[n_groups++] = ustep[ui];
          coef.step_size[gi] = uss[ui];
          coef.bc2_sqrt[gi] = ubc[ui];
        }
        active.group[ui] = (uint8_t)(gi | (seg_dense(ui) ? GQE_GROUP_DENSE : 0));
        chunks += ctx->universe[ui].n_chunks;
        active.begin[ui + 1] = (int32_t)chunks;
      }
      prefix_overflow = prefix_overflow || chunks > 0x7fffffffll;
    } else {
      const size_t first = staging.size();
      for (size_t ui = 0; ui < nu; ++ui) {
        if (!ustep[ui] || !keep(ui)) continue;
        GqeActSeg a;
        a.chunk_begin = chunks;
        a.seg = (int32_t)ui;
        a.step_size = uss[ui];
        a.bc2_sqrt = ubc[ui];
        a.pad = seg_dense(ui) ? 1 : 0;
        staging.push_back(a);
        chunks += ctx->universe[ui].n_chunks;
      }
      *act = act_dev + first;
      *n_act = (int)(staging.size() - first);
    }
    return chunks;
  };
  auto upload_staging = [&]() -> int {
    if (staging.empty()) return GQE_OK;
    const size_t bytes = sizeof(GqeActSeg) * staging.size();
    RingSlot* slot;
    int r = ring_acquire(ctx, bytes, &slot);
    if (r != GQE_OK) return r;
    memcpy(slot->host, staging.data(), bytes);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ws + ctx->lay.act_off, slot->host, bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipEventRecord(slot->done, st));
    slot->in_flight = true;
    return GQE_OK;
  };
  oa.mode = mode;
  {
    // the bytes of p, m, v the pass streams through the tables it steps: beyond the Infinity Cache -> non-temporal policy
    long long stream = 0;
    for (size_t ui = 0; ui < nu; ++ui)
      if (ustep[ui] && ctx->universe[ui].is_table) stream += 12ll * ctx->universe[ui].numel;
    oa.nt = mode == GQE_OPT_ADAM && stream > GQE_NT_STREAM_BYTES;
  }
  oa.lists = lists;
  // replicas must sum a row's contributions in the same order (exchange mode); elsewhere — a row-sharded row has ONE owner —
  // order-independent sums only buy run-to-run reproducibility and are the caller's choice (gqe_set_ordered_sums)
  oa.sorted = ctx->world > 1 || ctx->ordered_sums;
  oa.segs = reinterpret_cast<const GqeDevSeg*>(ctx->ws + ctx->lay.seg_off);
  oa.n_segs = (int)nu;
  oa.p = ctx->params;
  oa.g = ctx->grads;
  oa.m = ctx->m;
  oa.v = ctx->v;
  oa.head = reinterpret_cast<int32_t*>(ctx->ws + ctx->lay.head_off);
  oa.next = reinterpret_cast<const int32_t*>(ctx->ws + ctx->lay.next_off);
  oa.contrib = reinterpret_cast<const float*>(ctx->ws + ctx->lay.contrib_off);
  oa.link_contrib = reinterpret_cast<const int32_t*>(ctx->ws + ctx->lay.linkc_off);
  oa.max_entries = (int32_t)ctx->lay.max_entries;
  oa.hot = hot_args(ctx, false);
  oa.d = d;
  oa.lr = lr;
  oa.b1 = b1;
  oa.b2 = b2;
  oa.eps = eps;
  oa.stream = st;
  auto everything = [](size_t) { return true; };
  auto dense_only = [&](size_t ui) { return ctx->universe[ui].is_table == 0; };
  const bool timed = !flush && mode != GQE_OPT_MATERIALIZE && mode != GQE_OPT_ZERO;  // kernel 2 = the optimiser step proper
  bool sparse = false;
  if (ctx->lazy && mode == GQE_OPT_ADAM) {
    const Layout& L = ctx->lay;
    // ---- lazy Adam: per-row step counts.  Either the sparse launch over the rows of the pending margin call, or
    // the full pass (which also replays whatever any row is behind).
    if (!flush) {
      for (size_t t = 0; t < ctx->tables.size(); ++t)
        if (seen[t] && ctx->adam_steps[ctx->tables[t].offset] != ctx->tables[t].lstep + 1) {
          Table& tb = ctx->tables[t];
          const int want = ctx->adam_steps[tb.offset];
          if (tb.dirty)
            return fail(ctx, GQE_ERR_ARG, "lazy Adam: table at offset %lld is at step %d with rows still lagging, cannot jump to step %d",
                        (long long)tb.offset, tb.lstep, want);
          // a caller-supplied step count (resumed checkpoint): every row is current, so re-base the row counters
          tb.lstep = want - 1;
          tb.since_full = 0;
          HIP_TRY(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->ws + L.last_off + sizeof(int32_t) * (size_t)tb.head_base),
                                         tb.lstep, (size_t)tb.rows, st));
        }
      ctx->lz_lr = lr;
      ctx->lz_b1 = b1;
      ctx->lz_b2 = b2;
      ctx->lz_eps = eps;
      ctx->lz_hyper = true;
      sparse = lists && ctx->feed_valid && !any_dense(ctx) && (ctx->world == 1 || ctx->imported) &&
               (int)ctx->tables.size() <= GQE_LAZY_TABLES && (64 % (d / 4)) == 0;
      for (size_t t = 0; t < ctx->tables.size() && sparse; ++t)
        if (seen[t] && lazy_table_ok(ctx, (int)t) && ctx->tables[t].since_full >= GQE_LAZY_PERIOD)
          sparse = false;  // bound the replay depth of any row
    }
    if (sparse) {
      GqeRowsArgs ra;
      lazy_rows_args(ctx, ra, st);
      ra.with_grad = true;
      ra.sorted = oa.sorted;
      ra.idx = ctx->feed_idx;
      std::vector<SavedFeed> gathered;
      if (ctx->world > 1) {
        // the lists now hold every rank's entries: walk the gathered slabs' list-head tails instead of the local feed
        const int64_t n = ctx->imported_n, S = ctx->imported_slab;
        SavedFeed cur;
        memset(&cur, 0, sizeof cur);
        for (int k = 0; k < ctx->world; ++k) {
          if (cur.segs.n == GQE_LAZY_SEGS) {
            gathered.push_back(cur);
            memset(&cur, 0, sizeof cur);
          }
          GqeRowSegs& g = cur.segs;
          g.idx_begin[g.n] = (long long)((k * S + n) * d);   // int32 view of the contribution array: the slab's head tail
          g.tid[g.n] = -2;
          g.begin[g.n] = g.total;
          g.total += (int)n;
          g.begin[++g.n] = g.total;
        }
        gathered.push_back(cur);
        ra.idx = reinterpret_cast<const int32_t*>(ctx->ws + L.contrib_off);
      }
      const std::vector<SavedFeed>& feeds = ctx->world > 1 ? gathered : ctx->feed;
      for (size_t ui = 0; ui < nu; ++ui) {
        const GqeDevSeg& u = ctx->universe[ui];
        if (!ustep[ui] || !u.is_table || !lazy_table_ok(ctx, u.table_index)) continue;
        const int t = u.table_index;
        ra.t.target[t] = ra.t.grad_step[t] = ctx->tables[t].lstep + 1;
        ra.t.step_size[t] = uss[ui];
        ra.t.bc2_sqrt[t] = ubc[ui];
      }
      // bag-mode tables (their gradient lists hang on word rows no index feed names) are stepped in full by the
      // ordinary pass: every row of such a table is always current
      GqeOptArgs ob = oa;
      ob.total_chunks = emit([&](size_t ui) { return ctx->universe[ui].is_table && !lazy_table_ok(ctx, ctx->universe[ui].table_index); },
                             ob.active, ob.coef, &ob.act, &ob.n_act);
      // A deferred pair GEMM (gqe_set_deferred_gemm) rides in the row launch — the row groups do not read the matrix gradients —
      // when ONE launch covers the step (the usual case: one feed, or this feed + the prefetched next one); the d x d matrices
      // then leave the launch's dense chunks and are stepped by gqe_matstep_kernel behind it, as in the eager deferred step.
      auto is_matrix = [&](size_t ui) { return ctx->universe[ui].tile != nullptr; };
      const bool one_launch = feeds.size() == 1 && (!ctx->next_idx || (ctx->next_feed.size() == 1 && feeds[0].segs.n + ctx->next_feed[0].segs.n <= GQE_LAZY_SEGS));
      const bool ride = ctx->ride_pending && may_ride && one_launch && !ra.sorted && !merged_matrix && (64 % (d / 4)) == 0;
      if (ctx->ride_pending && !ride) {
        rc = flush_ride(ctx, st);
        if (rc != GQE_OK) return rc;
      }
      // the small dense tensors ride in extra workgroups of the (first) row launch: the ordinary pass, tables masked out
      ra.dsegs = oa.segs;
      ra.n_dsegs = oa.n_segs;
      ra.dense_chunks = ride ? emit([&](size_t ui) { return dense_only(ui) && !is_matrix(ui); }, ra.dactive, ra.dcoef, &ra.dact, &ra.n_dact)
                             : emit(dense_only, ra.dactive, ra.dcoef, &ra.dact, &ra.n_dact);
      if (prefix_overflow) return fail(ctx, GQE_ERR_ARG, "optimiser pass over more than 2^31 chunks");
      rc = upload_staging();
      if (rc != GQE_OK) return rc;
      rc = timing_begin(ctx, 2, st);
      if (rc != GQE_OK) return rc;
      bool merged = false;
      if (ctx->next_idx && feeds.size() == 1 && ctx->next_feed.size() == 1 &&
          feeds[0].segs.n + ctx->next_feed[0].segs.n <= GQE_LAZY_SEGS) {
        // gqe_lazy_prefetch, the usual case: rows(t) and rows(t+1) in ONE launch — the next feed's segments are
        // addressed relative to this feed's pointer
        GqeRowSegs both = feeds[0].segs;
        const GqeRowSegs& nx = ctx->next_feed[0].segs;
        const long long shift = ctx->next_idx - ra.idx;
        for (int k = 0; k < nx.n; ++k) {
          both.idx_begin[both.n] = nx.idx_begin[k] + shift;
          both.tid[both.n] = nx.tid[k];
          both.begin[both.n] = both.total;
          both.total += nx.begin[k + 1] - nx.begin[k];
          both.begin[++both.n] = both.total;
        }
        ra.segs = both;
        if (ride) {
          rc = lazy_rows_ride(ctx, ra);
          if (rc != GQE_OK) return rc;
        } else {
          HIP_TRY(ctx, gqe_launch_rows(ra));
        }
        ra.dense_chunks = 0;
        merged = true;
        ctx->caught_idx = ctx->next_idx;
        ctx->caught_n_idx = ctx->next_n_idx;
        ctx->caught_sig = ctx->next_sig;
      }
      for (const SavedFeed& sf : feeds) {
        if (merged) break;
        ra.segs = sf.segs;
        if (ride) {   // (one_launch: this loop runs once and nothing follows it)
          rc = lazy_rows_ride(ctx, ra);
          if (rc != GQE_OK) return rc;
        } else {
          HIP_TRY(ctx, gqe_launch_rows(ra));
        }
        ra.dense_chunks = 0;
      }
      if (ctx->next_idx && !merged) {
        // gqe_lazy_prefetch: the rows of the NEXT call ride in the same launches — with-gradient semantics, so that a
        // row named by both feeds is stepped by whichever entry claims it first; rows of the next feed alone have empty
        // lists and are simply replayed up to the step count this pass establishes
        ra.idx = ctx->next_idx;
        for (const SavedFeed& sf : ctx->next_feed) {
          ra.segs = sf.segs;
          HIP_TRY(ctx, gqe_launch_rows(ra));
        }
        ctx->caught_idx = ctx->next_idx;
        ctx->caught_n_idx = ctx->next_n_idx;
        ctx->caught_sig = ctx->next_sig;
      }
      rc = timing_end(ctx, 2, st);
      if (rc != GQE_OK) return rc;
      if (ride) {   // the d x d matrices, behind the kernel boundary that completes the units' sums
        GqeMatStep ms;
        memset(&ms, 0, sizeof ms);
        ms.tile_t = ctx->lay.tile_floats;
        bool any = false;
        for (size_t ui = 0; ui < nu; ++ui) any = any || (ustep[ui] && is_matrix(ui));
        if (any) {
          rc = timing_begin(ctx, 1, st);
          if (rc != GQE_OK) return rc;
        }
        for (size_t ui = 0; ui < nu; ++ui) {
          if (!ustep[ui] || !is_matrix(ui)) continue;
          ms.off[ms.n] = ctx->universe[ui].offset;
          ms.tile[ms.n] = ctx->universe[ui].tile;
          ms.step_size[ms.n] = uss[ui];
          ms.bc2_sqrt[ms.n] = ubc[ui];
          if (++ms.n == GQE_MATSTEP_MAX) {
            HIP_TRY(ctx, gqe_launch_matstep(ms, oa.p, oa.g, oa.m, oa.v, d, b1, b2, eps, st));
            ms.n = 0;
          }
        }
        HIP_TRY(ctx, gqe_launch_matstep(ms, oa.p, oa.g, oa.m, oa.v, d, b1, b2, eps, st));
        if (any) {
          rc = timing_end(ctx, 1, st);
          if (rc != GQE_OK) return rc;
        }
      }
      if (ctx->feed_buf >= 0) HIP_TRY(ctx, hipEventRecord(ctx->plan_free[ctx->feed_buf], st));  // the staged feed may go now
      if (ob.total_chunks > 0) {
        ob.lazy = false;
        rc = timing_begin(ctx, 3, st);
        if (rc != GQE_OK) return rc;
        HIP_TRY(ctx, gqe_launch_opt(ob));
        rc = timing_end(ctx, 3, st);
        if (rc != GQE_OK) return rc;
      }
      for (size_t t = 0; t < ctx->tables.size(); ++t)
        if (seen[t]) {
          ++ctx->tables[t].lstep;
          if (lazy_table_ok(ctx, (int)t)) {
            ++ctx->tables[t].since_full;
            ctx->tables[t].dirty = true;
          }
        }
    } else {
      if (ctx->ride_pending) {   // the full pass reads the matrix gradients in its dense chunks: the deferred GEMM first
        rc = flush_ride(ctx, st);
        if (rc != GQE_OK) return rc;
      }
      oa.lazy = true;
      oa.lz.last = reinterpret_cast<int32_t*>(ctx->ws + L.last_off);
      oa.lz.ring = reinterpret_cast<float2*>(ctx->ws + L.ring_off);
      if ((int)ctx->tables.size() > GQE_LAZY_TABLES) return fail(ctx, GQE_ERR_STATE, "lazy Adam supports at most %d tables", GQE_LAZY_TABLES);
      for (size_t t = 0; t < ctx->tables.size(); ++t) {
        oa.lz.t.target[t] = ctx->tables[t].lstep + ((seen[t] && !flush) ? 1 : 0);
        oa.lz.t.grad_step[t] = (seen[t] && !flush) ? ctx->tables[t].lstep + 1 : -1;
        oa.lz.t.eager[t] = lazy_table_ok(ctx, (int)t) ? 0 : 1;
      }
      oa.total_chunks = emit(everything, oa.active, oa.coef, &oa.act, &oa.n_act);
      if (prefix_overflow) return fail(ctx, GQE_ERR_ARG, "optimiser pass over more than 2^31 chunks");
      rc = upload_staging();
      if (rc != GQE_OK) return rc;
      if (timed) {
        rc = timing_begin(ctx, 2, st);
        if (rc != GQE_OK) return rc;
      }
      if (oa.total_chunks > 0) HIP_TRY(ctx, gqe_launch_opt(oa));
      // every row of the tables this pass covered is now current for its table's step count: the per-row counts are set
      // behind the launch, not by it (a row's threads may sit in two waves: see the kernel)
      for (size_t t = 0; t < ctx->tables.size(); ++t)
        if (seen[t] && lazy_table_ok(ctx, (int)t))
          HIP_TRY(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->ws + L.last_off + sizeof(int32_t) * (size_t)ctx->tables[t].head_base),
                                         oa.lz.t.target[t], (size_t)ctx->tables[t].rows, st));
      if (timed) {
        rc = timing_end(ctx, 2, st);
        if (rc != GQE_OK) return rc;
      }
      for (size_t t = 0; t < ctx->tables.size(); ++t)
        if (seen[t]) {
          if (!flush) ++ctx->tables[t].lstep;
          ctx->tables[t].since_full = 0;
          ctx->tables[t].dirty = false;
        }
    }
    if (flush) return GQE_OK;
  } else {
    if (ctx->split_launched) {
      // ---- launch B of a split step (gqe_train_step): the fused launch already stepped every row its feed does not name ----
      if (mode_in != GQE_OPT_ADAM || oa.sorted || oa.dense_tables || merged_matrix)
        return fail(ctx, GQE_ERR_STATE, "internal: split step reached a pass it cannot finish");
      auto is_matrix = [&](size_t ui) { return ctx->universe[ui].tile != nullptr; };
      for (size_t ui = 0; ui < nu; ++ui) {   // the tables: exactly the ones the riders stepped, with the same coefficients
        if (!ustep[ui] || !ctx->universe[ui].is_table || is_bag_table(ctx, ctx->universe[ui].table_index)) continue;
        int slot = -1;
        for (int k = 0; k < ctx->split_t.n; ++k)
          if (ctx->split_t.offset[k] == ctx->universe[ui].offset) slot = k;
        if (slot < 0 || ctx->split_t.step_size[slot] != uss[ui] || ctx->split_t.bc2_sqrt[slot] != ubc[ui])
          return fail(ctx, GQE_ERR_STATE, "internal: split step: the second launch disagrees with the riders about table %lld", (long long)ctx->universe[ui].offset);
      }
      // the launch's ordinary chunk loop: the vectors, and bag tables in full (their lists hang on rows no feed names)
      oa.total_chunks = emit([&](size_t ui) {
        const GqeDevSeg& u = ctx->universe[ui];
        return u.is_table ? is_bag_table(ctx, u.table_index) : !is_matrix(ui);
      }, oa.active, oa.coef, &oa.act, &oa.n_act);
      if (prefix_overflow) return fail(ctx, GQE_ERR_ARG, "optimiser pass over more than 2^31 chunks");
      rc = upload_staging();
      if (rc != GQE_OK) return rc;
      GqeGemmRide r;
      r.spread = 0;
      if (ctx->ride_pending) {
        r.plan = ctx->ride_fa.plan;
        r.formulas = ctx->ride_fa.formulas;
        r.ws = ctx->ride_fa.ws;
        r.tile_loss = ctx->ride_fa.tile_loss;
        r.losses = ctx->ride_losses;
        ctx->ride_pending = false;
        ++ctx->rides;
      } else {   // (more units than ride along: the pair GEMM and the finalize block already ran as a launch of their own)
        memset(&r, 0, sizeof r);
        r.plan.units = -1;
      }
      // (the chunk loop walks gradient lists only for a bag table: the launch without one runs the kernel compiled without them)
      oa.lists = false;
      for (size_t ui = 0; ui < nu; ++ui)
        if (ustep[ui] && ctx->universe[ui].is_table && is_bag_table(ctx, ctx->universe[ui].table_index)) oa.lists = true;
      rc = timing_begin(ctx, 2, st);
      if (rc != GQE_OK) return rc;
      HIP_TRY(ctx, gqe_launch_split_rows(oa, r, ctx->split_segs, ctx->split_ride, ctx->split_idx,
                                         reinterpret_cast<int32_t*>(ctx->ws + ctx->lay.stamp_off)));
      rc = timing_end(ctx, 2, st);
      if (rc != GQE_OK) return rc;
      if (ctx->split_buf >= 0) HIP_TRY(ctx, hipEventRecord(ctx->plan_free[ctx->split_buf], st));   // the staged feed may go NOW (not behind the fused launch)
      ctx->split_buf = -1;
      ctx->split_launched = false;
      ++ctx->split_steps;
      // the d x d matrices wait for the next step's first launch (split_first_launch) or for flush_split
      ctx->mat_pending.clear();
      ctx->mat_b1 = b1;
      ctx->mat_b2 = b2;
      ctx->mat_eps = eps;
      GqeMatStep ms;
      memset(&ms, 0, sizeof ms);
      ms.tile_t = ctx->lay.tile_floats;
      for (size_t ui = 0; ui < nu; ++ui) {
        if (!ustep[ui] || !is_matrix(ui)) continue;
        ms.off[ms.n] = ctx->universe[ui].offset;
        ms.tile[ms.n] = ctx->universe[ui].tile;
        ms.step_size[ms.n] = uss[ui];
        ms.bc2_sqrt[ms.n] = ubc[ui];
        if (++ms.n == GQE_MATSTEP_MAX) {
          ctx->mat_pending.push_back(ms);
          ms.n = 0;
        }
      }
      if (ms.n) ctx->mat_pending.push_back(ms);
      if (mode == GQE_OPT_ADAM)
        for (size_t t = 0; t < ctx->tables.size(); ++t)
          if (seen[t]) ctx->tables[t].lstep = ctx->adam_steps[ctx->tables[t].offset];
      goto consumed;
    }
    {
    // (next to the non-temporal pass over tables beyond the Infinity Cache the units do not LEAD the grid — reddit-synth 663 ->
    // 687 us per step with that — they are SPREAD through it: every K-th workgroup, K = 1 mod 8)
    const int ride_units = ctx->ride_pending ? ctx->ride_fa.plan.units : 0;
    int spread = 0;
    if (oa.nt && ride_units > 0 && !ride_spread_off()) {
      const long long chunks = emit([&](size_t ui) { return ctx->universe[ui].tile == nullptr; }, oa.active, oa.coef, &oa.act, &oa.n_act);
      const long long room = std::min<long long>(chunks, 262144);
      static const int forced = getenv("GQE_RIDE_SPREAD") ? atoi(getenv("GQE_RIDE_SPREAD")) : 0;   // (tuning runs: K, 1 mod 8)
      for (int k : {forced > 1 && forced % 8 == 1 ? forced : 33, 33, 17, 9})
        if (!spread && (long long)k * ride_units <= room) spread = k;
    }
    const bool ride = ctx->ride_pending && may_ride && oa.lists && !oa.sorted && !oa.dense_tables && !oa.lazy && !merged_matrix &&
                      (oa.nt ? !ride_spread_off() && (spread > 0 || ride_units == 0) : ride_units <= GQE_RIDE_MAX_UNITS);
    if (ctx->ride_pending && !ride) {
      rc = flush_ride(ctx, st);
      if (rc != GQE_OK) return rc;
    }
    if (!ride) {
      oa.total_chunks = emit(everything, oa.active, oa.coef, &oa.act, &oa.n_act);
      if (prefix_overflow) return fail(ctx, GQE_ERR_ARG, "optimiser pass over more than 2^31 chunks");
      rc = upload_staging();
      if (rc != GQE_OK) return rc;
    }
    if (ride) {
      // two launches: the pass over everything the GEMM units do not write (tables, vectors) with the units in front of its
      // chunks, then the d x d matrices — behind a kernel boundary, which is the only fence this part offers that does not
      // write back the L2 the streaming chunks are filling (profiles/r04_exp_gemm_rides_in_optimiser_launch.log)
      auto is_matrix = [&](size_t ui) { return ctx->universe[ui].tile != nullptr; };
      GqeOptArgs ob = oa;
      oa.total_chunks = emit([&](size_t ui) { return !is_matrix(ui); }, oa.active, oa.coef, &oa.act, &oa.n_act);
      ob.total_chunks = emit([&](size_t ui) { return is_matrix(ui); }, ob.active, ob.coef, &ob.act, &ob.n_act);
      if (prefix_overflow) return fail(ctx, GQE_ERR_ARG, "optimiser pass over more than 2^31 chunks");
      rc = upload_staging();
      if (rc != GQE_OK) return rc;
      GqeGemmRide r;
      r.plan = ctx->ride_fa.plan;
      r.formulas = ctx->ride_fa.formulas;
      r.ws = ctx->ride_fa.ws;
      r.tile_loss = ctx->ride_fa.tile_loss;
      r.losses = ctx->ride_losses;
      r.spread = spread;
      ctx->ride_pending = false;
      ++ctx->rides;
      if (timed) {
        rc = timing_begin(ctx, 2, st);
        if (rc != GQE_OK) return rc;
      }
      HIP_TRY(ctx, gqe_launch_opt_gemm(oa, r));
      if (timed) {
        rc = timing_end(ctx, 2, st);
        if (rc != GQE_OK) return rc;
      }
      const bool timed_mat = timed && ob.total_chunks > 0;
      if (timed_mat) {
        rc = timing_begin(ctx, 1, st);   // (the slot of the pair GEMM's own launch: what is left of it on the stream)
        if (rc != GQE_OK) return rc;
      }
      if (ob.total_chunks > 0) {
        // the matrices by name in the kernel arguments, GQE_MATSTEP_MAX per launch (one launch at every shape measured)
        GqeMatStep ms;
        ms.n = 0;
        ms.tile_t = ctx->lay.tile_floats;
        for (size_t ui = 0; ui < nu; ++ui) {
          if (!ustep[ui] || !is_matrix(ui)) continue;
          ms.off[ms.n] = ctx->universe[ui].offset;
          ms.tile[ms.n] = ctx->universe[ui].tile;
          ms.step_size[ms.n] = uss[ui];
          ms.bc2_sqrt[ms.n] = ubc[ui];
          if (++ms.n == GQE_MATSTEP_MAX) {
            HIP_TRY(ctx, gqe_launch_matstep(ms, oa.p, oa.g, oa.m, oa.v, d, b1, b2, eps, st));
            ms.n = 0;
          }
        }
        HIP_TRY(ctx, gqe_launch_matstep(ms, oa.p, oa.g, oa.m, oa.v, d, b1, b2, eps, st));
      }
      if (timed_mat) {
        rc = timing_end(ctx, 1, st);
        if (rc != GQE_OK) return rc;
      }
    } else {
      if (timed) {
        rc = timing_begin(ctx, 2, st);
        if (rc != GQE_OK) return rc;
      }
      if (oa.total_chunks > 0) HIP_TRY(ctx, gqe_launch_opt(oa));
      if (timed) {
        rc = timing_end(ctx, 2, st);
        if (rc != GQE_OK) return rc;
      }
    }
    if (mode == GQE_OPT_ADAM)
      for (size_t t = 0; t < ctx->tables.size(); ++t)
        if (seen[t]) ctx->tables[t].lstep = ctx->adam_steps[ctx->tables[t].offset];  // keeps gqe_set_lazy_adam(1) possible later
    }
  }
consumed:
  ctx->next_idx = nullptr;                            // a prefetch declaration holds for one optimiser step
  if (mode != GQE_OPT_ADAM) ctx->feed_valid = false;  // lists were dropped / folded: the saved feed no longer describes them
  // bookkeeping: which lists are consumed now
  bool any_pending = false;
  for (size_t t = 0; t < ctx->tables.size(); ++t) {
    if (seen[t]) ctx->tables[t].pending = false;
    any_pending = any_pending || ctx->tables[t].pending;
  }
  if (!any_pending) {
    ctx->entries_used = 0;
    ctx->links_used = false;   // (link nodes are numbered by their entry: nothing to recycle)
  }
  if (mode != GQE_OPT_MATERIALIZE)
    for (size_t t = 0; t < ctx->tables.size(); ++t)
      if (seen[t]) ctx->tables[t].dense = false;   // read and re-zeroed by this pass
  return GQE_OK;
}

}  // namespace

extern "C" {

int gqe_abi_version(void) { return GQE_ABI_VERSION; }

int gqe_dim_supported(int32_t decoder, int32_t inter, int32_t dim) { return gqe_config_supported(decoder, inter, dim); }

int gqe_debug_fused_variant(int32_t decoder, int32_t dim, int32_t tiles, int32_t* nc_full_fw) {
  if (!nc_full_fw) return GQE_ERR_ARG;
  gqe_fused_variant(decoder, dim, tiles, &nc_full_fw[0], &nc_full_fw[1], &nc_full_fw[2]);
  return GQE_OK;
}

const char* gqe_last_error(const gqe_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int gqe_create(const gqe_config* cfg, gqe_ctx** out) {
  if (!cfg || !out) return fail(nullptr, GQE_ERR_ARG, "null argument");
  if (cfg->abi_version != GQE_ABI_VERSION) return fail(nullptr, GQE_ERR_ARG, "ABI version mismatch: caller %d, library %d", cfg->abi_version, GQE_ABI_VERSION);
  if (cfg->dim < 16 || cfg->dim > GQE_MAX_DIM || cfg->dim % 16) return fail(nullptr, GQE_ERR_ARG, "dim must be a multiple of 16 in [16,%d], got %d", GQE_MAX_DIM, cfg->dim);
  if (cfg->decoder < 0 || cfg->decoder > 2) return fail(nullptr, GQE_ERR_ARG, "Metapath decoder not recognized.");
  if (cfg->inter < 0 || cfg->inter > 3) return fail(nullptr, GQE_ERR_ARG, "Intersection decoder not recognized.");
  if (!gqe_config_supported(cfg->decoder, cfg->inter, cfg->dim))
    return fail(nullptr, GQE_ERR_ARG, "dim %d is not supported with this decoder / intersection decoder: its fused kernel spills registers on gfx950 "
                "and spilling variants are not vouched for (gqe_dim_supported; full Bilinear: 16, 32, 48, 64, 128, 256; SetIntersection "
                "min / mean: not in (192, 256))", cfg->dim);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1) return fail(nullptr, GQE_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(e));
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, GQE_ERR_ARG, "device %d out of range (%d devices)", cfg->device, ndev);
  e = hipSetDevice(cfg->device);
  if (e != hipSuccess) return fail(nullptr, GQE_ERR_HIP, "hipSetDevice failed: %s", hipGetErrorString(e));
  gqe_ctx* ctx = new gqe_ctx();
  ctx->cfg = *cfg;
  *out = ctx;
  return GQE_OK;
}

void shard_session_free_fwd(gqe_ctx* ctx);

int gqe_destroy(gqe_ctx* ctx) {
  if (!ctx) return GQE_OK;
  shard_session_free_fwd(ctx);
  if (ctx->hot_seen) (void)hipHostFree(ctx->hot_seen);
  for (auto& s : ctx->ring) {
    if (s.in_flight) (void)hipEventSynchronize(s.done);
    if (s.done) (void)hipEventDestroy(s.done);
    if (s.host) (void)hipHostFree(s.host);
  }
  if (ctx->up) {
    (void)hipStreamSynchronize(ctx->up);
    for (int k = 0; k < 2; ++k) {
      (void)hipEventDestroy(ctx->plan_ready[k]);
      (void)hipEventDestroy(ctx->plan_free[k]);
    }
    (void)hipStreamDestroy(ctx->up);
  }
  for (auto& tv : ctx->timed)
    for (auto& t : tv) ctx->event_pool.push_back(t);
  for (auto& t : ctx->event_pool) {
    (void)hipEventDestroy(t.start);
    (void)hipEventDestroy(t.stop);
  }
  delete ctx;
  return GQE_OK;
}

int gqe_bind_arena(gqe_ctx* ctx, float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n) {
  if (!ctx) return GQE_ERR_ARG;
  if (!params || n < 1) return fail(ctx, GQE_ERR_ARG, "params arena is NULL or empty");
  if ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
       reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15)
    return fail(ctx, GQE_ERR_ARG, "arenas must be 16-byte aligned");
  if (!ctx->mat_pending.empty())
    return fail(ctx, GQE_ERR_STATE, "the last gqe_train_step's matrix step is pending: call gqe_optimizer_sync before re-binding the arenas");
  ctx->params = params;
  ctx->grads = grads;
  ctx->m = exp_avg;
  ctx->v = exp_avg_sq;
  ctx->n_arena = n;
  ctx->tiles_dirty = true;
  return GQE_OK;
}

int gqe_set_deferred_gemm(gqe_ctx* ctx, int32_t enable) {
  if (!ctx) return GQE_ERR_ARG;
  if (ctx->ride_pending) return fail(ctx, GQE_ERR_STATE, "a deferred pair GEMM is pending: step first");
  ctx->defer_gemm = enable != 0;
  return GQE_OK;
}

int64_t gqe_deferred_gemm_rides(gqe_ctx* ctx) { return ctx ? ctx->rides : -1; }

int gqe_params_changed(gqe_ctx* ctx) {
  if (!ctx) return GQE_ERR_ARG;
  ctx->tiles_dirty = true;
  return GQE_OK;
}

int gqe_set_tables(gqe_ctx* ctx, const int64_t* offsets, const int64_t* rows, int32_t n_tables) {
  if (!ctx) return GQE_ERR_ARG;
  if (!offsets || !rows || n_tables < 1 || n_tables > ctx->cap_tensors) return fail(ctx, GQE_ERR_ARG, "bad table list");
  if (ctx->entries_used) return fail(ctx, GQE_ERR_STATE, "gradients pending; step or materialize before changing the tables");
  if (ctx->shard_on && n_tables > GQE_LAZY_TABLES) return fail(ctx, GQE_ERR_ARG, "row-sharded mode supports at most %d tables", GQE_LAZY_TABLES);
  ctx->tables.clear();
  ctx->bags.clear();
  ctx->total_rows = 0;
  for (int t = 0; t < n_tables; ++t) {
    if (offsets[t] < 0 || (offsets[t] % 4) || rows[t] < 1) return fail(ctx, GQE_ERR_ARG, "table %d: bad offset / rows", t);
    {
      Table tb;
      tb.offset = offsets[t];
      tb.rows = rows[t];
      tb.head_base = ctx->total_rows;
      tb.pending = false;
      ctx->tables.push_back(tb);
    }
    ctx->total_rows += rows[t];
  }
  ctx->universe.clear();
  ctx->universe_uploaded = 0;
  drop_formulas(ctx);
  ctx->ws = nullptr;  // the workspace layout depends on the tables: it must be bound again
  return GQE_OK;
}

int gqe_set_bag(gqe_ctx* ctx, int64_t table_offset, const int32_t* bag_ptr, const int32_t* bag_ids, int64_t n_bags, int32_t max_len) {
  if (!ctx) return GQE_ERR_ARG;
  if (!bag_ptr || !bag_ids || n_bags < 1 || max_len < 1) return fail(ctx, GQE_ERR_ARG, "bad bag description");
  if (ctx->entries_used) return fail(ctx, GQE_ERR_STATE, "gradients pending; step or materialize before changing the bags");
  const int t = table_of(ctx, table_offset);
  if (t < 0) return fail(ctx, GQE_ERR_STATE, "gqe_set_bag: offset %lld is not a registered table", (long long)table_offset);
  for (Bag& bg : ctx->bags)
    if (bg.table == t) {
      bg = Bag{t, bag_ptr, bag_ids, n_bags, max_len};
      ctx->ws = nullptr;
      return GQE_OK;
    }
  if (ctx->bags.size() >= GQE_MAX_BAGS) return fail(ctx, GQE_ERR_ARG, "more than %d bag tables", GQE_MAX_BAGS);
  ctx->bags.push_back(Bag{t, bag_ptr, bag_ids, n_bags, max_len});
  drop_formulas(ctx);
  ctx->ws = nullptr;  // the workspace layout depends on the bags: it must be bound again
  return GQE_OK;
}

int64_t gqe_workspace_bytes(gqe_ctx* ctx, int64_t max_queries, int32_t max_batches) {
  if (!ctx || max_queries < 1 || max_batches < 1 || max_batches > GQE_MAX_BATCHES) return GQE_ERR_ARG;
  if (ctx->shard_on && (int)ctx->tables.size() > GQE_LAZY_TABLES) return fail(ctx, GQE_ERR_ARG, "row-sharded mode supports at most %d tables", GQE_LAZY_TABLES);
  ctx->cap_queries = max_queries;
  ctx->cap_batches = max_batches;
  return (int64_t)make_layout(ctx, max_queries, max_batches).total;
}

int gqe_bind_workspace(gqe_ctx* ctx, void* workspace, int64_t bytes, void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  if (!workspace) return fail(ctx, GQE_ERR_ARG, "workspace is NULL");
  if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(ctx, GQE_ERR_ARG, "workspace must be 256-byte aligned");
  if (ctx->cap_queries < 1) return fail(ctx, GQE_ERR_STATE, "call gqe_workspace_bytes first");
  if (ctx->entries_used) return fail(ctx, GQE_ERR_STATE, "gradients pending; step or materialize before re-binding the workspace");
  if (ctx->ws) {   // a split step's pending matrix step writes the operand-ordered copies of the OLD workspace: settle it there
    const int rcf = flush_split(ctx, reinterpret_cast<hipStream_t>(stream));
    if (rcf != GQE_OK) return rcf;
  }
  ctx->mat_pending.clear();
  if (lazy_any_dirty(ctx)) return fail(ctx, GQE_ERR_STATE, "lazy Adam: call gqe_optimizer_sync before re-binding the workspace");
  // an open row-sharded session sized its plan board and pinned feeds for the bound capacities, its planning thread may be
  // sorting a posted feed against them right now, and the peers did not grow with this rank
  if (ctx->shard_sess) return fail(ctx, GQE_ERR_STATE, "a row-sharded session is open: gqe_shard_close first, bind the same (larger) capacities on every rank, re-open");
  if (ctx->shard_on && (int)ctx->tables.size() > GQE_LAZY_TABLES) return fail(ctx, GQE_ERR_ARG, "row-sharded mode supports at most %d tables", GQE_LAZY_TABLES);
  const Layout L = make_layout(ctx, ctx->cap_queries, ctx->cap_batches);
  if ((int64_t)L.total > bytes) return fail(ctx, GQE_ERR_WORKSPACE, "workspace has %lld bytes, %zu needed", (long long)bytes, L.total);
  ctx->ws = static_cast<char*>(workspace);
  ctx->ws_bytes = bytes;
  ctx->lay = L;
  ctx->universe_uploaded = 0;
  for (GqeDevSeg& g : ctx->universe) universe_tiles(ctx, g);
  ctx->formulas_dirty.clear();  // the new workspace holds no descriptors yet: all cached slots are stale
  for (int k = 0; k < (int)ctx->formulas.size(); ++k) {
    (void)formula_tiles(ctx, ctx->formulas[(size_t)k]);   // (checked when the formula was registered: the tables have not changed since)
    ctx->formulas_dirty.push_back(k);
  }
  ctx->tiles_dirty = true;
  // empty gradient lists: head[row] = -1; link-node allocator at 0
  HIP_TRY(ctx, hipMemsetAsync(ctx->ws + L.head_off, 0xff, L.rows_off - L.head_off, reinterpret_cast<hipStream_t>(stream)));
  HIP_TRY(ctx, hipMemsetAsync(ctx->ws + L.counter_off, 0, 256, reinterpret_cast<hipStream_t>(stream)));   // + the hot-slot counter
  ctx->links_used = false;
  ctx->ride_pending = false;
  // split steps: no row is stamped
  HIP_TRY(ctx, hipMemsetAsync(ctx->ws + L.stamp_off, 0, L.total - L.stamp_off > 0 ? align_up(sizeof(int32_t) * (size_t)std::max<int64_t>(ctx->total_rows, 1), 256) : 0,
                              reinterpret_cast<hipStream_t>(stream)));
  ctx->split_epoch = -1;
  ctx->split_launched = false;
  // lazy Adam: every row is current for its table's step count, empty rings
  HIP_TRY(ctx, hipMemsetAsync(ctx->ws + L.ring_off, 0, L.hot_slot_off - L.ring_off, reinterpret_cast<hipStream_t>(stream)));
  // hot rows: none yet, empty accumulators
  HIP_TRY(ctx, hipMemsetAsync(ctx->ws + L.hot_slot_off, 0xff, L.hot_acc_off - L.hot_slot_off, reinterpret_cast<hipStream_t>(stream)));
  HIP_TRY(ctx, hipMemsetAsync(ctx->ws + L.hot_acc_off, 0, sizeof(float) * (size_t)GQE_HOT_REPS * GQE_HOT_SLOTS * ctx->cfg.dim, reinterpret_cast<hipStream_t>(stream)));
  if (L.tile_off > L.hot_sub_off) {   // hot word rows' sub-lists: empty counters, empty overflow chains
    HIP_TRY(ctx, hipMemsetAsync(ctx->ws + L.hot_sub_off, 0, sizeof(int32_t) * (size_t)GQE_HOT_SUB_POOL, reinterpret_cast<hipStream_t>(stream)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->ws + L.hot_sub_off + sizeof(int32_t) * (size_t)GQE_HOT_SUB_POOL, 0xff, sizeof(int32_t) * (size_t)GQE_HOT_SUB_POOL,
                                reinterpret_cast<hipStream_t>(stream)));
  }
  if (!ctx->bags.empty() && !ctx->hot_seen) {
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->hot_seen), 64, hipHostMallocMapped));
    HIP_TRY(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->hot_seen_dev), ctx->hot_seen, 0));
  }
  // (a promotion still in flight on another stream could set the word again after this: the gather then runs over empty sub-lists)
  if (ctx->hot_seen) *static_cast<volatile int32_t*>(ctx->hot_seen) = 0;
  ctx->hot_sub_on = false;
  for (auto& t : ctx->tables) {
    t.pending = false;
    t.since_full = 0;
    HIP_TRY(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->ws + L.last_off + sizeof(int32_t) * (size_t)t.head_base),
                                   t.lstep, (size_t)t.rows, reinterpret_cast<hipStream_t>(stream)));
  }
  ctx->feed_valid = false;
  return GQE_OK;
}

int gqe_set_lazy_adam(gqe_ctx* ctx, int32_t enable) {
  if (!ctx) return GQE_ERR_ARG;
  if (enable) {
    if ((int)ctx->tables.size() > GQE_LAZY_TABLES) return fail(ctx, GQE_ERR_ARG, "lazy Adam supports at most %d tables", GQE_LAZY_TABLES);
    if (any_dense(ctx) || ctx->entries_used) return fail(ctx, GQE_ERR_STATE, "gradients pending: step first");
  } else if (lazy_any_dirty(ctx)) {
    return fail(ctx, GQE_ERR_STATE, "rows still owe Adam steps: call gqe_optimizer_sync before leaving lazy mode");
  }
  ctx->lazy = enable != 0;
  return GQE_OK;
}

int gqe_lazy_prefetch(gqe_ctx* ctx, const gqe_batch* batches, int32_t n_batches, const int32_t* idx, int64_t n_idx, int32_t with_negatives) {
  if (!ctx) return GQE_ERR_ARG;
  ctx->next_idx = nullptr;
  if (!ctx->lazy) return GQE_OK;   // nothing is deferred in eager mode
  if (!batches || n_batches < 1 || n_batches > GQE_MAX_BATCHES || !idx || n_idx < 1) return fail(ctx, GQE_ERR_ARG, "gqe_lazy_prefetch: bad arguments");
  if (!ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_bind_workspace has not been called");
  std::vector<int> fid((size_t)n_batches);
  ++ctx->call_stamp;
  for (int bi = 0; bi < n_batches; ++bi) {
    const gqe_batch& s = batches[bi];
    if (s.n_queries < 1) return fail(ctx, GQE_ERR_ARG, "batch %d: empty batch", bi);
    int rc = formula_of(ctx, s, bi, &fid[(size_t)bi]);
    if (rc != GQE_OK) return rc;
    const int na = ctx->formulas[(size_t)fid[(size_t)bi]].n_anchors;
    const int64_t need = s.n_candidates > 0 ? (int64_t)s.idx_offset + (int64_t)na * s.n_queries + s.n_queries + 1 + s.n_candidates
                                            : (int64_t)s.idx_offset + (int64_t)(na + (with_negatives ? 2 : 1)) * s.n_queries;
    if (s.idx_offset < 0 || need > n_idx) return fail(ctx, GQE_ERR_ARG, "batch %d: index range exceeds the %lld indices given", bi, (long long)n_idx);
  }
  build_feed(ctx, batches, n_batches, with_negatives != 0, fid, ctx->next_feed);
  ctx->next_idx = idx;
  ctx->next_n_idx = n_idx;
  ctx->next_sig = feed_signature(batches, n_batches, with_negatives != 0);
  return GQE_OK;
}

int gqe_optimizer_sync(gqe_ctx* ctx, void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  if (ctx->ws) {   // gqe_train_step: the d x d matrices of the last split step
    const int rc = flush_split(ctx, reinterpret_cast<hipStream_t>(stream));
    if (rc != GQE_OK) return rc;
  }
  if (!ctx->lazy || !ctx->ws) return GQE_OK;
  return run_opt(ctx, GQE_OPT_FLUSH, nullptr, 0, 0.f, 0.f, 0.f, 0.f, stream);
}

int gqe_set_exchange(gqe_ctx* ctx, int32_t rank, int32_t world) {
  if (!ctx) return GQE_ERR_ARG;
  if (world < 1 || world > 1024 || rank < 0 || rank >= world) return fail(ctx, GQE_ERR_ARG, "need 0 <= rank < world <= 1024, got rank %d world %d", rank, world);
  if (ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_set_exchange must precede gqe_workspace_bytes / gqe_bind_workspace");
  if (ctx->shard_on && world > 1) return fail(ctx, GQE_ERR_STATE, "gqe_set_shard and gqe_set_exchange are mutually exclusive");
  ctx->rank = rank;
  ctx->world = world;
  return GQE_OK;
}

int gqe_set_limits(gqe_ctx* ctx, int32_t max_tensors, int32_t max_formulas) {
  if (!ctx) return GQE_ERR_ARG;
  if (ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_set_limits must precede gqe_workspace_bytes / gqe_bind_workspace");
  if (max_tensors > 0) {
    if (max_tensors > (1 << 20)) return fail(ctx, GQE_ERR_ARG, "max_tensors out of range");
    ctx->cap_tensors = std::max<int32_t>(max_tensors, (int32_t)ctx->tables.size());
  }
  if (max_formulas > 0) {
    if (max_formulas < GQE_MAX_BATCHES || max_formulas > (1 << 22))
      return fail(ctx, GQE_ERR_ARG, "max_formulas must be in [%d, %d]", GQE_MAX_BATCHES, 1 << 22);
    ctx->cap_formulas = max_formulas;
    drop_formulas(ctx);
  }
  return GQE_OK;
}

int gqe_set_ordered_sums(gqe_ctx* ctx, int32_t enable) {
  if (!ctx) return GQE_ERR_ARG;
  if (enable && !ctx->ordered_sums && ctx->ws) {
    // order-independent sums and hot rows exclude each other (float atomics arrive in any order): the kernels of this mode are
    // compiled without the accumulator path, so no row may stay hot and nothing may sit in an accumulator
    bool pending = ctx->entries_used != 0;
    for (const Table& t : ctx->tables) pending = pending || t.pending;
    if (pending) return fail(ctx, GQE_ERR_STATE, "gradients pending: step or gqe_zero_grads before gqe_set_ordered_sums");
    HIP_TRY(ctx, hipMemset(ctx->ws + ctx->lay.hot_slot_off, 0xff, ctx->lay.hot_acc_off - ctx->lay.hot_slot_off));
  }
  ctx->ordered_sums = enable != 0;
  return GQE_OK;
}

int gqe_hot_rows(gqe_ctx* ctx, int32_t* n_hot) {
  if (!ctx || !n_hot) return GQE_ERR_ARG;
  if (!ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_bind_workspace has not been called");
  int32_t n = 0;
  HIP_TRY(ctx, hipMemcpy(&n, ctx->ws + ctx->lay.counter_off + 64, sizeof n, hipMemcpyDeviceToHost));
  *n_hot = std::min<int32_t>(n, GQE_HOT_SLOTS);
  return GQE_OK;
}

int gqe_hot_sub_lists(gqe_ctx* ctx, int32_t* n_heads, int32_t* active) {
  if (!ctx || !n_heads || !active) return GQE_ERR_ARG;
  if (!ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_bind_workspace has not been called");
  int32_t n = 0;
  HIP_TRY(ctx, hipMemcpy(&n, ctx->ws + ctx->lay.counter_off + 68, sizeof n, hipMemcpyDeviceToHost));
  *n_heads = ctx->bags.empty() ? 0 : std::min<int32_t>(n, GQE_HOT_SUB_POOL);
  *active = hot_args(ctx, true).sub != nullptr;
  return GQE_OK;
}

int gqe_set_shard(gqe_ctx* ctx, int32_t rank, int32_t world) {
  if (!ctx) return GQE_ERR_ARG;
  if (world < 1 || world > 1024 || rank < 0 || rank >= world) return fail(ctx, GQE_ERR_ARG, "need 0 <= rank < world <= 1024, got rank %d world %d", rank, world);
  if (ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_set_shard must precede gqe_workspace_bytes / gqe_bind_workspace");
  if (ctx->world > 1) return fail(ctx, GQE_ERR_STATE, "gqe_set_shard and gqe_set_exchange are mutually exclusive");
  if ((int)ctx->tables.size() > GQE_LAZY_TABLES) return fail(ctx, GQE_ERR_ARG, "row-sharded mode supports at most %d tables", GQE_LAZY_TABLES);
  ctx->shard_rank = rank;
  ctx->shard_world = world;
  ctx->shard_on = true;
  return GQE_OK;
}

int gqe_shard_layout(gqe_ctx* ctx, gqe_shard_buffers* out) {
  if (!ctx || !out) return GQE_ERR_ARG;
  if (!ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_bind_workspace has not been called");
  if (!ctx->shard_on) return fail(ctx, GQE_ERR_STATE, "gqe_set_shard has not been called");
  const Layout& L = ctx->lay;
  out->req_send = (int64_t)L.shard_req_send;
  out->req_recv = (int64_t)L.shard_req_recv;
  out->rows_send = (int64_t)L.contrib_off;   // served rows are gathered into the (then idle) contribution entry space
  out->fetched = (int64_t)L.shard_fetch;
  out->contrib_send = (int64_t)L.shard_csend;
  out->contrib_recv = (int64_t)L.contrib_off;
  out->cap_send = L.shard_cap_send;
  out->cap_recv = L.shard_cap_recv;
  return GQE_OK;
}

// The owner sort of gqe_shard_plan.  Reads the ctx only (tables, bags, shard geometry) and reports errors through `err`,
// so the session's planning thread can run it next to the caller's thread (gqe_shard_step.h).
// own_direct: rows of this rank's own shard are named as GQE_OWN_ROW + local row in the position feed (margin steps of a
// session; the request list still holds them — the lazy replay and the sparse optimiser feed walk it).
int shard_plan_impl(const gqe_ctx* ctx, const gqe_batch* batches, int32_t n_batches, const int32_t* idx, int64_t n_idx, int32_t with_negatives,
                    int32_t* positions, int32_t* requests, int64_t* send_counts, char* err, size_t err_len, bool own_direct = false) {
  auto bad = [&](int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, err_len, fmt, ap);
    va_end(ap);
    return code;
  };
  if (!ctx->shard_on) return bad(GQE_ERR_STATE, "gqe_set_shard has not been called");
  if ((int)ctx->tables.size() > GQE_LAZY_TABLES) return bad(GQE_ERR_STATE, "row-sharded mode supports at most %d tables", GQE_LAZY_TABLES);
  if (!batches || n_batches < 1 || n_batches > GQE_MAX_BATCHES || !idx || n_idx < 1 || !positions || !requests || !send_counts)
    return bad(GQE_ERR_ARG, "gqe_shard_plan: bad arguments");
  const int W = ctx->shard_world, me = ctx->shard_rank;
  if (!with_negatives) own_direct = false;
  // the runs of the feed that name rows of one table (the layout of gqe_batch's index block), in feed order
  struct Run { int64_t off, n; int table; };
  Run runs[GQE_MAX_BATCHES * (3 + GQE_MAX_BRANCH)];   // table < 0: a run of list offsets (not rows)
  int n_runs = 0;
  for (int bi = 0; bi < n_batches; ++bi) {
    const gqe_batch& s = batches[bi];
    const int na = anchors_of(s.qtype);
    if (na < 0 || s.n_anchors != na || s.n_queries < 1) return bad(GQE_ERR_ARG, "batch %d: bad query type / anchors / size", bi);
    if (s.n_candidates < 0 || (s.n_candidates > 0 && with_negatives)) return bad(GQE_ERR_ARG, "batch %d: candidate lists are for gqe_forward only", bi);
    const int lead = s.n_candidates > 0 ? 0 : (with_negatives ? 2 : 1);
    const int64_t B = s.n_queries, o = s.idx_offset;
    const int64_t tail = s.n_candidates > 0 ? B + 1 + (int64_t)s.n_candidates : 0;   // cand_ptr[B + 1] | cand_rows[n_candidates]
    if (o < 0 || o + (lead + na) * B + tail > n_idx) return bad(GQE_ERR_ARG, "batch %d: index range exceeds the %lld indices given", bi, (long long)n_idx);
    const int tt = table_of(ctx, s.target_table);
    if (tt < 0) return bad(GQE_ERR_STATE, "batch %d: target_table is not a registered table", bi);
    if (lead) runs[n_runs++] = Run{o, lead * B, tt};
    for (int i = 0; i < na; ++i) {
      const int ta = table_of(ctx, s.anchor_table[i]);
      if (ta < 0) return bad(GQE_ERR_STATE, "batch %d: anchor_table[%d] is not a registered table", bi, i);
      runs[n_runs++] = Run{o + (lead + i) * B, B, ta};
    }
    if (s.n_candidates > 0) {
      // evaluation against candidate lists: the list offsets pass through, the candidates are rows of the target table
      // (fetched like any other row — a candidate named by several queries is fetched once per naming)
      runs[n_runs++] = Run{o + na * B, B + 1, -1};
      runs[n_runs++] = Run{o + na * B + B + 1, (int64_t)s.n_candidates, tt};
    }
  }
  std::sort(runs, runs + n_runs, [](const Run& a, const Run& b) { return a.off < b.off; });
  int64_t covered = 0;
  for (int k = 0; k < n_runs; ++k) {   // the batches must tile the feed: every index belongs to exactly one of them
    if (runs[k].off != covered) return bad(GQE_ERR_ARG, "index %lld of the feed belongs to %s", (long long)std::min(covered, runs[k].off), runs[k].off > covered ? "no batch" : "two batches");
    covered += runs[k].n;
  }
  if (covered != n_idx) return bad(GQE_ERR_ARG, "index %lld of the feed belongs to no batch", (long long)covered);
  bool bag_table[GQE_LAZY_TABLES] = {false, false, false, false, false, false, false, false};   // replicated: their indices (bag ids) pass through
  for (const Bag& bg : ctx->bags) bag_table[bg.table] = true;
  // counting sort of the feed by owner (row % W): request = list-head index of the row in the owner's shard, position =
  // where the fetched row (and later its gradient contribution) sits in the owner-grouped buffers
  int64_t at[1024];
  for (int o = 0; o < W; ++o) at[o] = 0;
  const bool pow2 = (W & (W - 1)) == 0;
  const int mask = W - 1;
  int shift = 0;
  while ((1 << shift) < W) ++shift;
  for (int k = 0; k < n_runs; ++k) {
    if (runs[k].table < 0 || bag_table[runs[k].table]) continue;
    const int32_t* p = idx + runs[k].off;
    int32_t neg = 0;
    if (W == 1) {
      for (int64_t j = 0; j < runs[k].n; ++j) neg |= p[j];
      at[0] += runs[k].n;
    } else if (pow2) {
      // four counter sets: consecutive indices with the same owner would otherwise chain through one memory cell
      int64_t c4[4][8];
      const bool small = W <= 8;
      if (small) memset(c4, 0, sizeof c4);
      int64_t j = 0;
      for (; small && j + 4 <= runs[k].n; j += 4) {
        neg |= p[j] | p[j + 1] | p[j + 2] | p[j + 3];
        ++c4[0][p[j] & mask];
        ++c4[1][p[j + 1] & mask];
        ++c4[2][p[j + 2] & mask];
        ++c4[3][p[j + 3] & mask];
      }
      for (; j < runs[k].n; ++j) {
        neg |= p[j];
        ++at[p[j] & mask];
      }
      if (small)
        for (int o = 0; o < W; ++o) at[o] += c4[0][o] + c4[1][o] + c4[2][o] + c4[3][o];
    } else
      for (int64_t j = 0; j < runs[k].n; ++j) {
        neg |= p[j];
        ++at[p[j] < 0 ? 0 : p[j] % W];
      }
    if (neg < 0) return bad(GQE_ERR_ARG, "the feed holds a negative index (batch run at %lld)", (long long)runs[k].off);
  }
  int64_t run = 0;
  for (int o = 0; o < W; ++o) {
    const int64_t c = at[o];
    send_counts[o] = c;
    at[o] = run;
    run += c;
  }
  for (int k = 0; k < n_runs; ++k) {
    const int32_t* p = idx + runs[k].off;
    int32_t* out = positions + runs[k].off;
    if (runs[k].table < 0) {
      memcpy(out, p, sizeof(int32_t) * (size_t)runs[k].n);
      continue;
    }
    if (bag_table[runs[k].table]) {
      for (int64_t j = 0; j < runs[k].n; ++j) {
        if (p[j] < 0) return bad(GQE_ERR_ARG, "index %lld of the feed is negative", (long long)(runs[k].off + j));
        out[j] = p[j];
      }
      continue;
    }
    const Table& tb = ctx->tables[(size_t)runs[k].table];
    const int64_t rows = tb.rows, hb = tb.head_base;
    int64_t worst = 0;
    if (W == 1) {   // every row is this rank's: positions are the feed order (no counter chain through memory)
      const int64_t base = at[0];
      for (int64_t j = 0; j < runs[k].n; ++j) {
        const int64_t r = p[j];
        worst = std::max(worst, r);
        out[j] = own_direct ? (int32_t)(GQE_OWN_ROW + r) : (int32_t)(base + j);
        requests[base + j] = (int32_t)(hb + r);
      }
      at[0] += runs[k].n;
      if (worst >= rows)
        return bad(GQE_ERR_ARG, "a global row of the batch run at %lld is outside its table (%lld local rows x %d ranks)", (long long)runs[k].off, (long long)rows, W);
      continue;
    }
    for (int64_t j = 0; j < runs[k].n; ++j) {
      const int64_t r = p[j];
      const int o = pow2 ? (int)(r & mask) : (int)(r % W);
      const int64_t local = pow2 ? (r >> shift) : (r / W);
      worst = std::max(worst, local);
      const int64_t pos = at[o]++;
      out[j] = own_direct && o == me ? (int32_t)(GQE_OWN_ROW + local) : (int32_t)pos;
      requests[pos] = (int32_t)(hb + local);
    }
    if (own_direct && rows >= GQE_OWN_ROW) return bad(GQE_ERR_ARG, "row-sharded step: a shard of more than 2^30 rows");
    if (worst >= rows)
      return bad(GQE_ERR_ARG, "a global row of the batch run at %lld is outside its table (%lld local rows x %d ranks)", (long long)runs[k].off, (long long)rows, W);
  }
  return GQE_OK;
}

int gqe_shard_plan(gqe_ctx* ctx, const gqe_batch* batches, int32_t n_batches, const int32_t* idx, int64_t n_idx, int32_t with_negatives,
                   int32_t* positions, int32_t* requests, int64_t* send_counts) {
  if (!ctx) return GQE_ERR_ARG;
  char err[256] = "";
  const int rc = shard_plan_impl(ctx, batches, n_batches, idx, n_idx, with_negatives, positions, requests, send_counts, err, sizeof err);
  return rc == GQE_OK ? GQE_OK : fail(ctx, rc, "%s", err);
}

int gqe_shard_serve(gqe_ctx* ctx, const int32_t* requests, int64_t n, float* rows_out, void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  if (!ctx->ws || !ctx->params) return fail(ctx, GQE_ERR_STATE, "arena / workspace not bound");
  if (!ctx->shard_on) return fail(ctx, GQE_ERR_STATE, "gqe_set_shard has not been called");
  if (n < 0 || n > ctx->lay.shard_cap_recv || (n > 0 && (!requests || !rows_out))) return fail(ctx, GQE_ERR_ARG, "gqe_shard_serve: bad arguments");
  if ((int)ctx->tables.size() > GQE_LAZY_TABLES) return fail(ctx, GQE_ERR_STATE, "row-sharded mode supports at most %d tables", GQE_LAZY_TABLES);
  if (ctx->lazy && !ctx->shard_internal)
    return fail(ctx, GQE_ERR_STATE, "row-sharded mode with lazy Adam runs through gqe_shard_step (the owner has to bring the rows it serves up to date)");
  if (ctx->entries_used) return fail(ctx, GQE_ERR_STATE, "row-sharded mode: received contributions are still pending (step first)");
  GqeShardTabs t;
  memset(&t, 0, sizeof t);
  t.n = (int)ctx->tables.size();
  for (size_t k = 0; k < ctx->tables.size(); ++k) {
    t.offset[k] = ctx->tables[k].offset;
    t.head_base[k] = ctx->tables[k].head_base;
  }
  const bool own = ctx->shard_internal && ctx->own_n > 0;
  const bool direct = own && ctx->own_direct;   // the own block is neither served nor linked here
  float* fetched = reinterpret_cast<float*>(ctx->ws + ctx->lay.shard_fetch);
  // gqe_shard_step (margin steps): the serve kernel also links the entries that will answer these requests (shard_link_early)
  const int link = ctx->shard_internal && ctx->shard_link_early ? 1 : 0;
  HIP_TRY(ctx, gqe_launch_shard_serve(ctx->params, requests, n, rows_out, ctx->cfg.dim, t, own ? ctx->own_lo : 0, own ? ctx->own_n : 0,
                                      own && !direct ? fetched + ctx->own_fetch * ctx->cfg.dim : nullptr,
                                      reinterpret_cast<int32_t*>(ctx->ws + ctx->lay.head_off), reinterpret_cast<int32_t*>(ctx->ws + ctx->lay.next_off),
                                      own ? ctx->own_entry : 0, link, reinterpret_cast<hipStream_t>(stream)));
  return GQE_OK;
}

int gqe_shard_link(gqe_ctx* ctx, const int32_t* requests, int64_t n, void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  if (!ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_bind_workspace has not been called");
  if (!ctx->shard_on) return fail(ctx, GQE_ERR_STATE, "gqe_set_shard has not been called");
  if (n < 0 || n > ctx->lay.shard_cap_recv || (n > 0 && !requests)) return fail(ctx, GQE_ERR_ARG, "gqe_shard_link: bad arguments");
  if (ctx->entries_used) return fail(ctx, GQE_ERR_STATE, "row-sharded mode: the previous step's contributions are still linked (step first)");
  const Layout& L = ctx->lay;
  const bool own = ctx->shard_internal && ctx->own_n > 0;
  if (!(ctx->shard_internal && ctx->shard_link_early))   // (gqe_shard_step: the serve kernel of this step linked them already; bookkeeping only)
    HIP_TRY(ctx, gqe_launch_shard_link(reinterpret_cast<int32_t*>(ctx->ws + L.head_off), reinterpret_cast<int32_t*>(ctx->ws + L.next_off), requests, n,
                                       own ? ctx->own_lo : 0, own ? ctx->own_n : 0, own ? (ctx->own_direct ? -1 : ctx->own_entry) : 0,
                                       reinterpret_cast<hipStream_t>(stream)));
  ctx->shard_sent = false;
  if (n > 0) {
    ctx->entries_used = n;
    for (int t : ctx->shard_tables) ctx->tables[(size_t)t].pending = true;  // every rank ran the same formulas
  }
  for (const Table& tb : ctx->tables)
    if (tb.pending && ctx->entries_used == 0) ctx->entries_used = 1;   // (local bag contributions only)
  return GQE_OK;
}

int gqe_exchange_reserve(gqe_ctx* ctx, int64_t slab_entries) {
  if (!ctx) return GQE_ERR_ARG;
  if (slab_entries < 0) return fail(ctx, GQE_ERR_ARG, "slab_entries must be >= 0");
  ctx->slab_hint = slab_entries;
  return GQE_OK;
}

int gqe_export_entries(gqe_ctx* ctx, int64_t* slab_entries_out, int64_t* contrib_offset, void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  if (!ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_bind_workspace has not been called");
  if (ctx->world < 2) return fail(ctx, GQE_ERR_STATE, "gqe_set_exchange(world > 1) has not been called");
  if (ctx->entries_used == 0 || ctx->step_slab == 0) return fail(ctx, GQE_ERR_STATE, "no margin call is pending");
  const Layout& L = ctx->lay;
  if (!ctx->step_exported) {
    // flush the deferred work that still adds into the dense gradients, then pack the slab's tails
    HIP_TRY(ctx, gqe_launch_export(reinterpret_cast<float*>(ctx->ws + L.contrib_off), reinterpret_cast<const int32_t*>(ctx->ws + L.rows_off),
                                   ctx->grads, ctx->cfg.dim, (long long)ctx->rank * ctx->step_slab, (int32_t)ctx->step_entries,
                                   dense_spans(ctx), reinterpret_cast<hipStream_t>(stream)));
    ctx->step_exported = true;
  }
  if (slab_entries_out) *slab_entries_out = ctx->step_slab;
  if (contrib_offset) *contrib_offset = (int64_t)L.contrib_off;
  return GQE_OK;
}

int gqe_import_entries(gqe_ctx* ctx, int64_t slab, void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  if (!ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_bind_workspace has not been called");
  if (ctx->world < 2) return fail(ctx, GQE_ERR_STATE, "gqe_set_exchange(world > 1) has not been called");
  if (!ctx->step_exported || slab != ctx->step_slab || ctx->entries_used == 0)
    return fail(ctx, GQE_ERR_ARG, "import of %lld-entry slabs, but the exported slab of the pending margin call has %lld (0: none exported)",
                (long long)slab, (long long)(ctx->step_exported ? ctx->step_slab : 0));
  const Layout& L = ctx->lay;
  GqeImportBags ib;
  memset(&ib, 0, sizeof ib);
  for (size_t k = 0; k < ctx->bags.size(); ++k) {
    ib.csr.ptr[k] = ctx->bags[k].ptr;
    ib.csr.ids[k] = ctx->bags[k].ids;
    ib.csr.max_len = std::max(ib.csr.max_len, ctx->bags[k].max_len);
    ib.head_base[k] = ctx->tables[ctx->bags[k].table].head_base;
  }
  ib.link_contrib = reinterpret_cast<int32_t*>(ctx->ws + L.linkc_off);
  ib.link_counter = reinterpret_cast<int32_t*>(ctx->ws + L.counter_off);
  ib.max_entries = (int32_t)L.max_entries;
  if (!ctx->bags.empty()) ctx->links_used = true;
  HIP_TRY(ctx, gqe_launch_import(reinterpret_cast<int32_t*>(ctx->ws + L.head_off), reinterpret_cast<int32_t*>(ctx->ws + L.next_off),
                                 reinterpret_cast<const float*>(ctx->ws + L.contrib_off), ctx->grads, ctx->cfg.dim, (long long)slab,
                                 (int32_t)ctx->step_entries, ctx->rank, ctx->world, dense_spans(ctx), ib, reinterpret_cast<hipStream_t>(stream)));
  ctx->imported = true;
  ctx->imported_n = ctx->step_entries;
  ctx->imported_slab = slab;
  ctx->step_exported = false;  // imported: a second import of the same step is refused
  ctx->step_slab = 0;
  return GQE_OK;
}

int gqe_forward(gqe_ctx* ctx, const gqe_batch* batches, int32_t n_batches, const int32_t* idx, int64_t n_idx,
                int32_t idx_on_device, float* scores, void* stream) {
  if (ctx && !scores) return fail(ctx, GQE_ERR_ARG, "scores buffer is NULL");
  return run_queries(ctx, batches, n_batches, idx, n_idx, idx_on_device, false, nullptr, scores, nullptr, stream);
}

int gqe_margin_fwd_bwd(gqe_ctx* ctx, const gqe_batch* batches, int32_t n_batches, const int32_t* idx, int64_t n_idx,
                       int32_t idx_on_device, float* losses, float* pos_scores, float* neg_scores, void* stream) {
  return run_queries(ctx, batches, n_batches, idx, n_idx, idx_on_device, true, losses, pos_scores, neg_scores, stream);
}

int gqe_rank_candidates(gqe_ctx* ctx, const float* scores, const int32_t* cand_ptr, int32_t n_queries, double* percentile, void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  if (n_queries < 0 || (n_queries > 0 && (!scores || !cand_ptr || !percentile))) return fail(ctx, GQE_ERR_ARG, "gqe_rank_candidates: bad arguments");
  HIP_TRY(ctx, gqe_launch_rank(scores, cand_ptr, n_queries, percentile, reinterpret_cast<hipStream_t>(stream)));
  return GQE_OK;
}

int gqe_auc_pair_counts(gqe_ctx* ctx, const float* pos, int64_t n_pos, const float* neg, int64_t n_neg, uint64_t* count2, void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  if (n_pos < 0 || n_neg < 0 || !count2 || (n_pos > 0 && !pos) || (n_neg > 0 && !neg)) return fail(ctx, GQE_ERR_ARG, "gqe_auc_pair_counts: bad arguments");
  HIP_TRY(ctx, gqe_launch_auc(pos, n_pos, neg, n_neg, reinterpret_cast<unsigned long long*>(count2), reinterpret_cast<hipStream_t>(stream)));
  return GQE_OK;
}

int gqe_materialize_grads(gqe_ctx* ctx, void* stream) {
  return run_opt(ctx, GQE_OPT_MATERIALIZE, nullptr, 0, 0.f, 0.f, 0.f, 0.f, stream);
}

int gqe_materialize_tables(gqe_ctx* ctx, const int64_t* table_offsets, int32_t n_tables, void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  if (!table_offsets || n_tables < 1 || n_tables > GQE_MAX_BATCHES) return fail(ctx, GQE_ERR_ARG, "gqe_materialize_tables: bad table list");
  std::vector<gqe_segment> segs((size_t)n_tables);
  for (int i = 0; i < n_tables; ++i) {
    if (table_of(ctx, table_offsets[i]) < 0) return fail(ctx, GQE_ERR_ARG, "offset %lld is not a registered table", (long long)table_offsets[i]);
    segs[(size_t)i] = gqe_segment{table_offsets[i], 0, 0, 0};
  }
  return run_opt(ctx, GQE_OPT_MATERIALIZE, segs.data(), n_tables, 0.f, 0.f, 0.f, 0.f, stream);
}

// RCCL is bound at run time (dlopen): the library links against the HIP runtime only, and a single-GPU user never
// needs librccl.  Signature of ncclAllReduce (rccl.h): (sendbuff, recvbuff, count, datatype, op, comm, stream).
typedef int (*gqe_nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);

int gqe_allreduce_grads(gqe_ctx* ctx, void* nccl_comm, void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  if (!nccl_comm) return fail(ctx, GQE_ERR_ARG, "gqe_allreduce_grads: communicator is NULL");
  if (!ctx->grads) return fail(ctx, GQE_ERR_STATE, "no gradient arena bound");
  if (ctx->world > 1 || ctx->shard_on)
    return fail(ctx, GQE_ERR_STATE, "gqe_allreduce_grads is the dense exchange of replicated tables: not in gqe_set_exchange / gqe_set_shard mode");
  static gqe_nccl_allreduce_fn allreduce = nullptr;
  if (!allreduce) {
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(ctx, GQE_ERR_STATE, "cannot load librccl.so: %s", dlerror());
    allreduce = reinterpret_cast<gqe_nccl_allreduce_fn>(dlsym(h, "ncclAllReduce"));
    if (!allreduce) return fail(ctx, GQE_ERR_STATE, "librccl.so has no ncclAllReduce");
  }
  int rc = run_opt(ctx, GQE_OPT_MATERIALIZE, nullptr, 0, 0.f, 0.f, 0.f, 0.f, stream);   // row lists -> dense gradient
  if (rc != GQE_OK) return rc;
  const int nccl_float32 = 7, nccl_sum = 0;   // ncclDataType_t / ncclRedOp_t (rccl.h)
  const int nr = allreduce(ctx->grads, ctx->grads, (size_t)ctx->n_arena, nccl_float32, nccl_sum, nccl_comm, reinterpret_cast<hipStream_t>(stream));
  if (nr != 0) return fail(ctx, GQE_ERR_HIP, "ncclAllReduce failed with ncclResult_t %d", nr);
  return GQE_OK;
}

int gqe_adam_step(gqe_ctx* ctx, const gqe_segment* segs, int32_t n_segs, float lr, float beta1, float beta2, float eps, void* stream) {
  return run_opt(ctx, GQE_OPT_ADAM, segs, n_segs, lr, beta1, beta2, eps, stream);
}

// ---- the decoder / encoder extension points on [d, B] tensors (include/gqe.h) ----
namespace {
int x_settle(gqe_ctx* ctx, void* stream) {   // whatever a split step / deferred launch / lazy rows still owe the parameters
  if (!ctx->params) return fail(ctx, GQE_ERR_STATE, "gqe_bind_arena has not been called");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (ctx->ws) {
    int rc = flush_ride(ctx, st);
    if (rc == GQE_OK) rc = flush_split(ctx, st);
    if (rc != GQE_OK) return rc;
  }
  if (ctx->lazy && ctx->ws) return run_opt(ctx, GQE_OPT_FLUSH, nullptr, 0, 0.f, 0.f, 0.f, 0.f, stream);
  return GQE_OK;
}
bool x_param_ok(const gqe_ctx* ctx, int64_t off, int64_t numel) { return off >= 0 && off + numel <= ctx->n_arena; }
}  // namespace

int gqe_encode_rows(gqe_ctx* ctx, int64_t table_offset, const int32_t* rows, int32_t B, float* out, void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  if (!rows || !out || B < 0) return fail(ctx, GQE_ERR_ARG, "gqe_encode_rows: bad arguments");
  const int t = table_of(ctx, table_offset);
  if (t < 0) return fail(ctx, GQE_ERR_ARG, "gqe_encode_rows: no table at offset %lld", (long long)table_offset);
  int rc = x_settle(ctx, stream);
  if (rc != GQE_OK) return rc;
  const int32_t *bp = nullptr, *bi = nullptr;
  for (const Bag& bg : ctx->bags)
    if (bg.table == t) {
      bp = bg.ptr;
      bi = bg.ids;
    }
  HIP_TRY(ctx, gqe_launch_x_encode(ctx->params + table_offset, rows, B, ctx->cfg.dim, bp, bi, out, reinterpret_cast<hipStream_t>(stream)));
  return GQE_OK;
}

int gqe_decoder_project(gqe_ctx* ctx, int64_t rel_param, const float* embeds, int32_t B, float* out, void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  const int64_t d = ctx->cfg.dim, numel = ctx->cfg.decoder == GQE_DEC_BILINEAR ? d * d : d;
  if (!embeds || !out || B < 0 || !x_param_ok(ctx, rel_param, numel)) return fail(ctx, GQE_ERR_ARG, "gqe_decoder_project: bad arguments");
  int rc = x_settle(ctx, stream);
  if (rc != GQE_OK) return rc;
  HIP_TRY(ctx, gqe_launch_x_project(ctx->cfg.decoder, ctx->params + rel_param, embeds, B, (int)d, out, reinterpret_cast<hipStream_t>(stream)));
  return GQE_OK;
}

int gqe_decoder_forward(gqe_ctx* ctx, const int64_t* rel_params, int32_t n_rels, const float* embeds1, const float* embeds2, int32_t B,
                        float* scores, void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  const int64_t d = ctx->cfg.dim, numel = ctx->cfg.decoder == GQE_DEC_BILINEAR ? d * d : d;
  if (!embeds1 || !embeds2 || !scores || B < 0 || n_rels < 0 || n_rels > GQE_MAX_HOPS || (n_rels && !rel_params))
    return fail(ctx, GQE_ERR_ARG, "gqe_decoder_forward: bad arguments (at most %d relations)", GQE_MAX_HOPS);
  long long rp[GQE_MAX_HOPS] = {0, 0, 0};
  for (int k = 0; k < n_rels; ++k) {
    if (!x_param_ok(ctx, rel_params[k], numel)) return fail(ctx, GQE_ERR_ARG, "gqe_decoder_forward: relation %d outside the arena", k);
    rp[k] = rel_params[k];
  }
  int rc = x_settle(ctx, stream);
  if (rc != GQE_OK) return rc;
  HIP_TRY(ctx, gqe_launch_x_forward(ctx->cfg.decoder, ctx->params, rp, n_rels, embeds1, embeds2, B, (int)d, scores, reinterpret_cast<hipStream_t>(stream)));
  return GQE_OK;
}

int gqe_set_intersection(gqe_ctx* ctx, int64_t pre_param, int64_t post_param, const float* embeds1, const float* embeds2,
                         const float* embeds3, int32_t B, float* out, void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  const int64_t d = ctx->cfg.dim;
  if (!embeds1 || !embeds2 || !out || B < 0) return fail(ctx, GQE_ERR_ARG, "gqe_set_intersection: bad arguments");
  if (is_mlp(ctx) != (pre_param >= 0) || (pre_param >= 0) != (post_param >= 0))
    return fail(ctx, GQE_ERR_ARG, "gqe_set_intersection: Pre / Post are given exactly for the MLP intersections");
  if (pre_param >= 0 && (!x_param_ok(ctx, pre_param, d * d) || !x_param_ok(ctx, post_param, d * d)))
    return fail(ctx, GQE_ERR_ARG, "gqe_set_intersection: Pre / Post outside the arena");
  int rc = x_settle(ctx, stream);
  if (rc != GQE_OK) return rc;
  HIP_TRY(ctx, gqe_launch_x_intersect(pre_param >= 0 ? ctx->params + pre_param : nullptr, post_param >= 0 ? ctx->params + post_param : nullptr,
                                      is_min(ctx) ? 1 : 0, embeds1, embeds2, embeds3, B, (int)d, out, reinterpret_cast<hipStream_t>(stream)));
  return GQE_OK;
}

int64_t gqe_split_steps(gqe_ctx* ctx) { return ctx ? ctx->split_steps : -1; }

int gqe_train_step(gqe_ctx* ctx, const gqe_batch* batches, int32_t n_batches, const int32_t* idx, int64_t n_idx, int32_t idx_on_device,
                   const gqe_segment* segs, int32_t n_segs, float lr, float beta1, float beta2, float eps, float* losses, void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  if (!batches || n_batches < 1 || !segs || n_segs < 1) return fail(ctx, GQE_ERR_ARG, "gqe_train_step: no batches / no segments given");
  if (!ctx->params || !ctx->grads || !ctx->m || !ctx->v) return fail(ctx, GQE_ERR_STATE, "gqe_train_step: parameter, gradient and Adam moment arenas must be bound");
  if (!ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_bind_workspace has not been called");
  static const bool env_off = [] {
    const char* e = getenv("GQE_SPLIT");
    return e && atoi(e) == 0;
  }();
  const int d = ctx->cfg.dim;
  // ---- may this step run split?  Everything the second launch will check is checked here, before anything is enqueued ----
  static const bool prof_ok = getenv("GQE_SPLIT_PROF") != nullptr;   // (debug profile of a split step: tools/probes/split_timeline.py)
  // (bag tables — nn.EmbeddingBag word tables: their gradient lists hang on word rows no feed names — are stepped in full by the
  // second launch; the riders cover the plain tables)
  bool split = !env_off && !ctx->lazy && !ctx->ordered_sums && ctx->world == 1 && !ctx->shard_on && (!ctx->prof || prof_ok) &&
               n_batches <= GQE_LAUNCH_BATCHES && n_segs <= ctx->cap_tensors && ctx->entries_used == 0 && !any_dense(ctx) &&
               gqe_fused_can_ride(ctx->cfg.decoder, is_mlp(ctx) ? 1 : 0, d, 0);
  for (const Table& t : ctx->tables) split = split && !t.pending;
  {
    // launches of more tiles than the 16-wave shape is chosen for (thousands of tiles, two or three 8-wave workgroups per CU)
    // are long MFMA-bound launches next to which the Adam stream is small change — and riders that share a CU with tiles slow
    // the tiles' latency chains far more than they gain (tools/probes/split_timeline.py): the two-call sequence
    long long tiles = 0;
    for (int bi = 0; bi < n_batches; ++bi) tiles += (batches[bi].n_queries + GQE_TQ - 1) / GQE_TQ;
    static const bool many_ok = getenv("GQE_SPLIT_MANY_TILES") != nullptr;   // (experiment 85: the split step on launches of thousands of tiles)
    split = split && (tiles <= GQE_FW8_MIN_TILES || many_ok);
    // (a row's stamp carries the feed entry that owns it in its low 16 bits: gqe_split.h)
    long long feed_entries = 0;
    for (int bi = 0; bi < n_batches; ++bi) feed_entries += (long long)batches[bi].n_queries * (2 + GQE_MAX_BRANCH);
    split = split && feed_entries <= GQE_SPLIT_MAX_ENTRIES;
  }
  std::vector<gqe_segment> resolved(segs, segs + n_segs);
  GqeSplitTabs st;
  memset(&st, 0, sizeof st);
  std::vector<int64_t> bag_offsets;
  long long stream_bytes = 0;
  for (int i = 0; i < n_segs && split; ++i) {
    gqe_segment& s = resolved[(size_t)i];
    if (s.offset < 0 || (s.offset % 4) != 0 || s.numel < 1 || s.offset + s.numel > ctx->n_arena) split = false;   // (run_opt reports it)
    if (!split) break;
    for (int j = 0; j < i; ++j) split = split && resolved[(size_t)j].offset != s.offset;
    // the step count this call applies: the caller's, or the library's counter + 1 (run_opt commits it)
    if (s.step < 1) {
      auto it = ctx->adam_steps.find(s.offset);
      s.step = (it == ctx->adam_steps.end() ? 0 : it->second) + 1;
    }
    const int t = table_of(ctx, s.offset);
    if (t < 0) {
      // a dense segment must be a vector or exactly one registered matrix (what the riding units and the matrix step assume)
      auto it = std::lower_bound(ctx->matrices.begin(), ctx->matrices.end(), s.offset);
      const bool covers = (it != ctx->matrices.end() && *it < s.offset + s.numel) ||
                          (it != ctx->matrices.begin() && *(it - 1) + (int64_t)d * d > s.offset);
      if (covers && !(s.numel == (int64_t)d * d && it != ctx->matrices.end() && *it == s.offset)) split = false;
      if (covers && tile_of(ctx, s.offset) < 0) split = false;
      continue;
    }
    if (s.numel != ctx->tables[(size_t)t].rows * d) {
      split = false;
      break;
    }
    if (is_bag_table(ctx, t)) {   // stepped by the second launch's ordinary chunk loop (lists + link nodes), not by riders
      bag_offsets.push_back(s.offset);
      continue;
    }
    if (st.n == GQE_SPLIT_TABLES) {
      split = false;
      break;
    }
    const int k = st.n++;
    st.offset[k] = s.offset;
    st.head_base[k] = ctx->tables[(size_t)t].head_base;
    st.rows[k] = ctx->tables[(size_t)t].rows;
    st.step_size[k] = (float)((double)lr / (1.0 - std::pow((double)beta1, (double)s.step)));
    st.bc2_sqrt[k] = (float)std::sqrt(1.0 - std::pow((double)beta2, (double)s.step));
    st.blk_begin[k + 1] = st.blk_begin[k] + (int)((ctx->tables[(size_t)t].rows + GQE_SPLIT_WROWS - 1) / GQE_SPLIT_WROWS);
    stream_bytes += 12ll * s.numel;
  }
  static const long long max_stream = [] {   // GQE_SPLIT_MAX_STREAM_MB: tuning runs only
    const char* e = getenv("GQE_SPLIT_MAX_STREAM_MB");
    return e ? (long long)atoll(e) << 20 : 2 * (long long)GQE_NT_STREAM_BYTES;   // (d = 256 bio-synth, 298 MB of p + m + v: 179 -> 160 us per step)
  }();
  split = split && st.n > 0 && stream_bytes <= max_stream;
  // every table the batches name has to be among the stepped ones (the second launch owns their stamped rows)
  for (int bi = 0; bi < n_batches && split; ++bi) {
    const gqe_batch& b = batches[bi];
    auto stepped = [&](int64_t off) {
      for (int k = 0; k < st.n; ++k)
        if (st.offset[k] == off) return true;
      return std::find(bag_offsets.begin(), bag_offsets.end(), off) != bag_offsets.end();
    };
    split = stepped(b.target_table) && b.n_candidates == 0 && b.n_anchors >= 1 && b.n_anchors <= GQE_MAX_BRANCH;
    for (int i = 0; i < b.n_anchors && i < GQE_MAX_BRANCH && split; ++i) split = stepped(b.anchor_table[i]);
  }
  if (!split) {
    // the two-call sequence; the matrix-gradient units may still ride in the Adam pass (the losses are defined behind it)
    const bool defer = ctx->defer_gemm;
    ctx->defer_gemm = true;
    int rc = run_queries(ctx, batches, n_batches, idx, n_idx, idx_on_device, true, losses, nullptr, nullptr, stream);
    ctx->defer_gemm = defer;
    if (rc != GQE_OK) return rc;
    rc = run_opt(ctx, GQE_OPT_ADAM, segs, n_segs, lr, beta1, beta2, eps, stream);
    if (rc != GQE_OK) return rc;
    return flush_ride(ctx, reinterpret_cast<hipStream_t>(stream));   // (a pass that could not carry them: losses[] as documented)
  }
  ctx->split_t = st;
  ctx->split_stream_bytes = stream_bytes;
  ctx->split_b1 = beta1;
  ctx->split_b2 = beta2;
  ctx->split_eps = eps;
  ctx->split_active = true;
  ctx->split_launched = false;
  int rc = run_queries(ctx, batches, n_batches, idx, n_idx, idx_on_device, true, losses, nullptr, nullptr, stream);
  ctx->split_active = false;
  if (rc == GQE_OK) rc = run_opt(ctx, GQE_OPT_ADAM, resolved.data(), n_segs, lr, beta1, beta2, eps, stream);
  if (rc != GQE_OK) {
    // (riders may have run without their second launch: the parameters are in an undefined state, as after any failed step)
    ctx->split_launched = false;
    return rc;
  }
  return GQE_OK;
}

int gqe_adam_step_count(gqe_ctx* ctx, int64_t offset, int32_t* count) {
  if (!ctx || !count) return GQE_ERR_ARG;
  auto it = ctx->adam_steps.find(offset);
  *count = it == ctx->adam_steps.end() ? 0 : it->second;
  return GQE_OK;
}

int gqe_set_adam_step_count(gqe_ctx* ctx, int64_t offset, int32_t count) {
  if (!ctx || count < 0) return GQE_ERR_ARG;
  if (offset < 0 || offset >= ctx->n_arena) return fail(ctx, GQE_ERR_ARG, "gqe_set_adam_step_count: offset %lld is outside the arena", (long long)offset);
  if (ctx->split_launched || !ctx->mat_pending.empty())
    return fail(ctx, GQE_ERR_STATE, "gqe_set_adam_step_count: a split step is still pending (gqe_optimizer_sync comes first)");
  if (count == 0) ctx->adam_steps.erase(offset);
  else ctx->adam_steps[offset] = count;
  return GQE_OK;
}

int gqe_sgd_step(gqe_ctx* ctx, const gqe_segment* segs, int32_t n_segs, float lr, void* stream) {
  return run_opt(ctx, GQE_OPT_SGD, segs, n_segs, lr, 0.f, 0.f, 0.f, stream);
}

int gqe_zero_grads(gqe_ctx* ctx, const gqe_segment* segs, int32_t n_segs, void* stream) {
  return run_opt(ctx, GQE_OPT_ZERO, segs, n_segs, 0.f, 0.f, 0.f, 0.f, stream);
}


// ---- native training feed -------------------------------------------------------------------
static inline uint64_t feeder_next(gqe_feeder* f) {  // xoroshiro128+
  const uint64_t s0 = f->rng[0];
  uint64_t s1 = f->rng[1];
  const uint64_t r = s0 + s1;
  s1 ^= s0;
  f->rng[0] = ((s0 << 24) | (s0 >> 40)) ^ s1 ^ (s1 << 16);
  f->rng[1] = (s1 << 37) | (s1 >> 27);
  return r;
}

static inline uint64_t feeder_next_neg(gqe_feeder* f) {  // the same generator on the per-rank state
  const uint64_t s0 = f->neg_rng[0];
  uint64_t s1 = f->neg_rng[1];
  const uint64_t r = s0 + s1;
  s1 ^= s0;
  f->neg_rng[0] = ((s0 << 24) | (s0 >> 40)) ^ s1 ^ (s1 << 16);
  f->neg_rng[1] = (s1 << 37) | (s1 >> 27);
  return r;
}

int gqe_feeder_create(gqe_ctx* ctx, uint64_t seed, int32_t batch_size, float path_weight, float inter_weight, gqe_feeder** out) {
  if (!ctx || !out) return GQE_ERR_ARG;
  if (batch_size < 1) return fail(ctx, GQE_ERR_ARG, "batch_size must be >= 1");
  gqe_feeder* f = new gqe_feeder();
  f->ctx = ctx;
  uint64_t z = seed + 0x9E3779B97F4A7C15ull;  // splitmix64 seeding
  for (int k = 0; k < 2; ++k) {
    z += 0x9E3779B97F4A7C15ull;
    uint64_t x = z;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    f->rng[k] = x ^ (x >> 31);
  }
  z += 0x9E3779B97F4A7C15ull * (uint64_t)(1 + (ctx->shard_on ? ctx->shard_rank : 0));
  for (int k = 0; k < 2; ++k) {
    z += 0x9E3779B97F4A7C15ull;
    uint64_t x = z;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    f->neg_rng[k] = x ^ (x >> 31);
  }
  f->batch_size = batch_size;
  f->path_weight = path_weight;
  f->inter_weight = inter_weight;
  *out = f;
  return GQE_OK;
}

int gqe_feeder_destroy(gqe_feeder* f) {
  if (!f) return GQE_OK;
  for (int k = 0; k < 2; ++k)
    if (f->pin_ev[k]) {
      if (f->pin_ev_set[k]) (void)hipEventSynchronize(f->pin_ev[k]);
      (void)hipEventDestroy(f->pin_ev[k]);
    }
  for (int k = 0; k < 2 * kFeedGroup; ++k)
    if (f->pin[k]) (void)hipHostFree(f->pin[k]);
  for (int k = 0; k < 2; ++k) {
    if (f->grp_ready[k]) {
      (void)hipEventSynchronize(f->grp_ready[k]);
      (void)hipEventDestroy(f->grp_ready[k]);
    }
    if (f->grp_free[k]) {
      if (f->grp_free_set[k]) (void)hipEventSynchronize(f->grp_free[k]);
      (void)hipEventDestroy(f->grp_free[k]);
    }
    if (f->grp_pin[k]) (void)hipHostFree(f->grp_pin[k]);
  }
  delete f;
  return GQE_OK;
}

int gqe_feeder_set_feed(gqe_feeder* f, int32_t mode) {
  if (!f) return GQE_ERR_ARG;
  if (mode != 0 && mode != 1) return fail(f->ctx, GQE_ERR_ARG, "feed mode must be 0 (pinned staging + hipMemcpyAsync) or 1 (kernels read pinned host memory)");
  f->feed_mode = mode;
  return GQE_OK;
}

int gqe_feeder_add_pool(gqe_feeder* f, const gqe_batch* formula, int64_t n, const int32_t* target, const int32_t* anchors,
                        const int32_t* neg, const int32_t* hard) {
  if (!f) return GQE_ERR_ARG;
  gqe_ctx* ctx = f->ctx;
  if (!formula || n < 1 || !target || !anchors) return fail(ctx, GQE_ERR_ARG, "bad pool");
  const int na = anchors_of(formula->qtype);
  if (na < 0 || na != formula->n_anchors) return fail(ctx, GQE_ERR_ARG, "pool: bad query type / anchor count");
  if (formula->qtype != GQE_Q_1CHAIN && !neg) return fail(ctx, GQE_ERR_ARG, "pool: stored negatives are required except for 1-chain");
  int fid;
  gqe_batch probe = *formula;
  probe.n_queries = 1;
  int rc = formula_of(ctx, probe, (int)f->pools.size(), &fid);  // validates the static fields
  if (rc != GQE_OK) return rc;
  FeederPool p;
  p.proto = *formula;
  p.n = n;
  p.target.assign(target, target + n);
  p.anchors.assign(anchors, anchors + (size_t)na * n);
  if (neg) p.neg.assign(neg, neg + n);
  if (hard) p.hard.assign(hard, hard + n);
  f->pools.push_back(std::move(p));
  const int t = formula->qtype;
  f->by_type[t].push_back((int)f->pools.size() - 1);
  f->cum[t].push_back((f->cum[t].empty() ? 0.0 : f->cum[t].back()) + (double)n);
  return GQE_OK;
}

int gqe_feeder_set_mode_rows(gqe_feeder* f, int64_t table_offset, const int32_t* rows, int64_t n) {
  if (!f) return GQE_ERR_ARG;
  if (!rows || n < 1) return fail(f->ctx, GQE_ERR_ARG, "bad row list");
  f->mode_rows[table_offset].assign(rows, rows + n);
  return GQE_OK;
}

int gqe_feeder_add_pool_lists(gqe_feeder* f, const gqe_batch* formula, int64_t n, const int32_t* target, const int32_t* anchors,
                              const int64_t* neg_ptr, const int32_t* neg_rows, const int64_t* hard_ptr, const int32_t* hard_rows) {
  if (!f) return GQE_ERR_ARG;
  gqe_ctx* ctx = f->ctx;
  if (!formula || n < 1) return fail(ctx, GQE_ERR_ARG, "bad pool");
  if (formula->qtype != GQE_Q_1CHAIN && (!neg_ptr || !neg_rows)) return fail(ctx, GQE_ERR_ARG, "pool: negative lists are required except for 1-chain");
  auto check = [&](const int64_t* ptr) {
    if (!ptr) return true;
    if (ptr[0] != 0) return false;
    for (int64_t q = 0; q < n; ++q)
      if (ptr[q + 1] <= ptr[q] || ptr[q + 1] - ptr[q] > 0xffffffffll) return false;   // random.choice of an empty list raises
    return true;
  };
  if (!check(neg_ptr) || !check(hard_ptr)) return fail(ctx, GQE_ERR_ARG, "pool: every query needs a non-empty negative list");
  // (the stored-negative arrays of the pool stay empty: with lists, a batch's negatives are drawn)
  std::vector<int32_t> one((size_t)n, 0);
  int rc = gqe_feeder_add_pool(f, formula, n, target, anchors, formula->qtype == GQE_Q_1CHAIN ? nullptr : one.data(), nullptr);
  if (rc != GQE_OK) return rc;
  FeederPool& p = f->pools.back();
  p.neg.clear();
  if (neg_ptr && neg_rows) {
    p.neg_ptr.assign(neg_ptr, neg_ptr + n + 1);
    p.neg_rows.assign(neg_rows, neg_rows + neg_ptr[n]);
  }
  if (hard_ptr && hard_rows) {
    p.hard_ptr.assign(hard_ptr, hard_ptr + n + 1);
    p.hard_rows.assign(hard_rows, hard_rows + hard_ptr[n]);
  }
  return GQE_OK;
}

int gqe_feeder_set_reference_streams(gqe_feeder* f, uint32_t* np_state625, uint32_t* py_state625) {
  if (!f) return GQE_ERR_ARG;
  if ((np_state625 == nullptr) != (py_state625 == nullptr)) return fail(f->ctx, GQE_ERR_ARG, "reference streams: both states or neither");
  if (np_state625 && (np_state625[624] > 624 || py_state625[624] > 624)) return fail(f->ctx, GQE_ERR_ARG, "reference streams: bad generator position");
  f->np_state = np_state625;
  f->py_state = py_state625;
  return GQE_OK;
}

int gqe_feeder_set_type_order(gqe_feeder* f, const int32_t* qtypes, int32_t n) {
  if (!f) return GQE_ERR_ARG;
  if (n < 0 || (n > 0 && !qtypes)) return fail(f->ctx, GQE_ERR_ARG, "bad type order");
  std::vector<int> order;
  for (int i = 0; i < n; ++i) {
    if (qtypes[i] <= GQE_Q_1CHAIN || qtypes[i] > GQE_Q_3CHAIN_INTER) return fail(f->ctx, GQE_ERR_ARG, "type order: query types behind 1-chain only");
    if (std::find(order.begin(), order.end(), qtypes[i]) != order.end()) return fail(f->ctx, GQE_ERR_ARG, "type order: a type twice");
    order.push_back(qtypes[i]);
  }
  f->type_order = order;
  return GQE_OK;
}

int gqe_feeder_set_pvals(gqe_feeder* f, int32_t qtype, const double* pvals, int32_t n) {
  if (!f) return GQE_ERR_ARG;
  if (qtype < 0 || qtype > GQE_Q_3CHAIN_INTER || !pvals || n != (int32_t)f->by_type[qtype].size())
    return fail(f->ctx, GQE_ERR_ARG, "pvals: one probability per pool of the type");
  f->pvals[qtype].assign(pvals, pvals + n);
  return GQE_OK;
}

int gqe_feeder_set_sgd(gqe_feeder* f, int32_t enable) {
  if (!f) return GQE_ERR_ARG;
  f->sgd = enable != 0;
  return GQE_OK;
}

int gqe_feeder_set_loss_stride(gqe_feeder* f, int64_t stride) {
  if (!f) return GQE_ERR_ARG;
  if (stride != 0 && stride < GQE_LAUNCH_BATCHES + 1) return fail(f->ctx, GQE_ERR_ARG, "loss stride: 0 or at least %d floats", GQE_LAUNCH_BATCHES + 1);
  f->loss_stride = stride;
  return GQE_OK;
}

// one (formula, slice) batch appended to f->batches / f->idx
static int feeder_batch(gqe_feeder* f, int qtype, int64_t it, float weight, bool hard) {
  gqe_ctx* ctx = f->ctx;
  const std::vector<int>& cand = f->by_type[qtype];
  if (cand.empty()) return GQE_OK;
  size_t pick = 0;
  const bool ref = f->np_state != nullptr;
  if (ref) {
    // np.random.multinomial(1, sizes / sum).argmax() on the caller's generator (train_helpers.py:96-99; one pool: no draw, as in numpy)
    if (f->pvals[qtype].size() != cand.size()) return fail(ctx, GQE_ERR_STATE, "reference streams: gqe_feeder_set_pvals missing for query type %d", qtype);
    uint32_t pos = f->np_state[624];
    pick = (size_t)gqe_mt::np_multinomial_one(f->np_state, pos, f->pvals[qtype].data(), (int64_t)cand.size());
    f->np_state[624] = pos;
  } else if (cand.size() > 1) {
    const double u = (double)(feeder_next(f) >> 11) * (1.0 / 9007199254740992.0) * f->cum[qtype].back();
    pick = std::lower_bound(f->cum[qtype].begin(), f->cum[qtype].end(), u) - f->cum[qtype].begin();
    if (pick >= cand.size()) pick = cand.size() - 1;
  }
  const FeederPool& p = f->pools[cand[pick]];
  if (ref && qtype != GQE_Q_1CHAIN && (hard ? p.hard_ptr.empty() : p.neg_ptr.empty()))
    return fail(ctx, GQE_ERR_ARG, "reference streams: pool of query type %d has no %snegative lists", qtype, hard ? "hard-" : "");
  if (!ref && qtype != GQE_Q_1CHAIN && p.neg.empty()) return fail(ctx, GQE_ERR_ARG, "pool of query type %d holds negative lists: reference streams only", qtype);
  if (!ref && hard && p.hard.empty()) return fail(ctx, GQE_ERR_ARG, "pool of query type %d has no hard negatives", qtype);
  const int64_t n = p.n, B = f->batch_size;
  // row-sharded data parallelism: the W ranks of iteration `it` train the W consecutive slices it * W + rank of the same
  // formula draw (the reference's wrap-around rule, train_helpers.py:102-105) and weight their mean losses by n_rank / n_all
  const int W = ctx->shard_on ? ctx->shard_world : 1, rk = ctx->shard_on ? ctx->shard_rank : 0;
  auto slice = [&](int64_t i, int64_t* a, int64_t* e) {
    *a = (i * B) % n;
    *e = std::min(((i + 1) * B) % n, n);
    if (*e <= *a) *e = n;
  };
  int64_t start, end;
  slice(it * W + rk, &start, &end);
  const int64_t m = end - start;
  if (W > 1) {
    int64_t n_all = 0;
    for (int r = 0; r < W; ++r) {
      int64_t a, e;
      slice(it * W + r, &a, &e);
      n_all += e - a;
    }
    weight *= (float)((double)m / (double)n_all);
  }
  gqe_batch b = p.proto;
  b.n_queries = (int32_t)m;
  b.idx_offset = (int32_t)f->idx.size();
  b.out_offset = 0;
  b.margin = 1.f;
  b.loss_weight = weight;
  b.n_candidates = 0;
  f->idx.insert(f->idx.end(), p.target.begin() + start, p.target.begin() + end);
  if (qtype == GQE_Q_1CHAIN) {
    auto it_rows = f->mode_rows.find(p.proto.target_table);
    if (it_rows == f->mode_rows.end()) return fail(ctx, GQE_ERR_STATE, "gqe_feeder_set_mode_rows missing for the 1-chain target table");
    const std::vector<int32_t>& rows = it_rows->second;
    if (ref) {   // random.choice(graph.full_lists[mode]) per query (model.py:113-114)
      uint32_t pos = f->py_state[624];
      for (int64_t k = 0; k < m; ++k) f->idx.push_back(rows[(size_t)gqe_mt::py_randbelow(f->py_state, pos, (int64_t)rows.size())]);
      f->py_state[624] = pos;
    } else {
      // (several ranks: the negatives come from a per-rank stream, the formula draws stay on the shared one)
      for (int64_t k = 0; k < m; ++k) f->idx.push_back(rows[(W > 1 ? feeder_next_neg(f) : feeder_next(f)) % rows.size()]);
    }
  } else if (ref) {   // random.choice(query.neg_samples / hard_neg_samples) per query (model.py:115-120)
    const std::vector<int64_t>& ptr = hard ? p.hard_ptr : p.neg_ptr;
    const std::vector<int32_t>& rows = hard ? p.hard_rows : p.neg_rows;
    uint32_t pos = f->py_state[624];
    for (int64_t q = start; q < end; ++q) f->idx.push_back(rows[(size_t)(ptr[q] + gqe_mt::py_randbelow(f->py_state, pos, ptr[q + 1] - ptr[q]))]);
    f->py_state[624] = pos;
  } else {
    const std::vector<int32_t>& src = hard ? p.hard : p.neg;
    f->idx.insert(f->idx.end(), src.begin() + start, src.begin() + end);
  }
  for (int a = 0; a < p.proto.n_anchors; ++a)
    f->idx.insert(f->idx.end(), p.anchors.begin() + (size_t)a * n + start, p.anchors.begin() + (size_t)a * n + end);
  f->batches.push_back(b);
  f->queries_fed += m;
  return GQE_OK;
}

static void feeder_touch(gqe_feeder* f, int64_t offset, int64_t numel) {
  if (offset < 0) return;
  for (const gqe_segment& s : f->segs)
    if (s.offset == offset) return;
  gqe_segment s;
  s.offset = offset;
  s.numel = numel;
  s.step = 0;  // library-kept Adam step counters
  s.reserved = 0;
  f->segs.push_back(s);
}

// sample + pack iteration `it` into f->batches / f->idx / f->segs: the reference's schedule (train_helpers.py:50-72): 1-chain
// always; after burn-in every other type, chains once (path_weight), intersections with regular and with hard negatives
// (inter_weight each)
static int feeder_build_inner(gqe_feeder* f, int64_t it, int32_t burn_in);
static int feeder_build(gqe_feeder* f, int64_t it, int32_t burn_in) {
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = feeder_build_inner(f, it, burn_in);
  f->host_build_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}

static int feeder_build_inner(gqe_feeder* f, int64_t it, int32_t burn_in) {
  gqe_ctx* ctx = f->ctx;
  const int d = ctx->cfg.dim;
  const bool bil = ctx->cfg.decoder == GQE_DEC_BILINEAR;
  const int64_t vec = bil ? (int64_t)d * d : d;
  f->batches.clear();
  f->idx.clear();
  f->segs.clear();
  int rc = feeder_batch(f, GQE_Q_1CHAIN, it, 1.f, false);
  if (rc != GQE_OK) return rc;
  if (it >= burn_in) {
    std::vector<int> order = f->type_order;   // (gqe_feeder_set_type_order: the caller's dictionary order; else the enum's)
    if (order.empty())
      for (int t = GQE_Q_2CHAIN; t <= GQE_Q_3CHAIN_INTER; ++t) order.push_back(t);
    for (int t : order) {
      if (f->by_type[t].empty()) continue;
      const bool inter = t >= GQE_Q_2INTER;
      rc = feeder_batch(f, t, it, inter ? f->inter_weight : f->path_weight, false);
      if (rc != GQE_OK) return rc;
      if (inter) {
        rc = feeder_batch(f, t, it, f->inter_weight, true);
        if (rc != GQE_OK) return rc;
      }
    }
  }
  if (f->batches.empty()) return fail(ctx, GQE_ERR_STATE, "feeder has no 1-chain pool");
  for (const gqe_batch& b : f->batches) {
    const int tt = table_of(ctx, b.target_table);
    feeder_touch(f, b.target_table, ctx->tables[tt].rows * d);
    const bool chain = b.qtype <= GQE_Q_3CHAIN;
    for (int a = 0; a < b.n_anchors; ++a) feeder_touch(f, b.anchor_table[a], ctx->tables[table_of(ctx, b.anchor_table[a])].rows * d);
    for (int i = 0; i < (chain ? 1 : b.n_anchors); ++i)
      for (int h = 0; h < b.n_hops[i]; ++h) feeder_touch(f, b.hop_param[i][h], vec);
    if (!chain) {
      if (b.n_final) feeder_touch(f, b.final_param, vec);
      if (is_mlp(ctx)) {
        feeder_touch(f, b.pre_param, (int64_t)d * d);
        feeder_touch(f, b.post_param, (int64_t)d * d);
      }
    }
  }
  return GQE_OK;
}

int shard_plans_ahead(gqe_ctx* ctx);   // gqe_shard_step.h: plans posted but not yet run

// make iteration `it` ready: sampled, packed, and its index feed where the kernels will read it.
//   zero-copy (feed mode 1): one of 16 pinned host slots (two groups of eight; an event per group guards their re-use);
//   copy (feed mode 0): the iterations of a GROUP of eight are sampled together and travel with ONE hipMemcpyAsync on the
//     library's upload stream into one of two groups of staged buffers in the workspace — two cross-stream event packets
//     per eight iterations instead of per iteration (they sit between kernels that otherwise overlap);
//   row-sharded: a host feed of global rows for gqe_shard_post.
static int feeder_ensure(gqe_feeder* f, int64_t it, int32_t burn_in, hipStream_t st) {
  gqe_ctx* ctx = f->ctx;
  gqe_feeder::Prepared& P = f->prep[it % (2 * kFeedGroup)];
  if (P.it == it) return GQE_OK;
  const Layout& L = ctx->lay;
  const size_t slot_ints = L.idx_cap / sizeof(int32_t);
  int rc;
  auto stash = [&](gqe_feeder::Prepared& Q, int64_t i) {
    Q.it = i;
    Q.batches = f->batches;
    Q.segs = f->segs;
    Q.n_idx = (int64_t)f->idx.size();
  };
  if (ctx->shard_on) {
    rc = feeder_build(f, it, burn_in);
    if (rc != GQE_OK) return rc;
    stash(P, it);
    P.host_idx = f->idx;
    P.dev_idx = nullptr;
    return GQE_OK;
  }
  if (f->grp_cap != slot_ints) {   // first use, or the workspace was re-bound with another capacity: nothing may still read the old buffers
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (ctx->up) HIP_TRY(ctx, hipStreamSynchronize(ctx->up));
    for (int k = 0; k < 2 * kFeedGroup; ++k) {
      if (f->pin[k]) HIP_TRY(ctx, hipHostFree(f->pin[k]));
      f->pin[k] = nullptr;
    }
    for (int k = 0; k < 2; ++k) {
      if (f->grp_pin[k]) HIP_TRY(ctx, hipHostFree(f->grp_pin[k]));
      f->grp_pin[k] = nullptr;
      if (!f->pin_ev[k]) HIP_TRY(ctx, hipEventCreateWithFlags(&f->pin_ev[k], hipEventDisableTiming));
      if (!f->grp_ready[k]) HIP_TRY(ctx, hipEventCreateWithFlags(&f->grp_ready[k], hipEventDisableTiming));
      if (!f->grp_free[k]) HIP_TRY(ctx, hipEventCreateWithFlags(&f->grp_free[k], hipEventDisableTiming));
      f->pin_ev_set[k] = f->grp_free_set[k] = false;
    }
    f->grp_cap = slot_ints;
    for (auto& q : f->prep) q.it = -1;
  }
  if (f->feed_mode == 1) {
    if (!f->pin[0])
      for (int k = 0; k < 2 * kFeedGroup; ++k) HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&f->pin[k]), L.idx_cap, hipHostMallocDefault));
    const long long group = it / kFeedGroup;
    if (f->pin_ev_set[group & 1]) {   // the slots of group - 2 come up for re-use
      if (hipEventQuery(f->pin_ev[group & 1]) != hipSuccess) HIP_TRY(ctx, hipEventSynchronize(f->pin_ev[group & 1]));
      f->pin_ev_set[group & 1] = false;
    }
    rc = feeder_build(f, it, burn_in);
    if (rc != GQE_OK) return rc;
    if (f->idx.size() > slot_ints) return fail(ctx, GQE_ERR_WORKSPACE, "index feed of %zu entries exceeds the bound workspace", f->idx.size());
    int32_t* slot = f->pin[it % (2 * kFeedGroup)];
    memcpy(slot, f->idx.data(), f->idx.size() * sizeof(int32_t));
    stash(P, it);
    P.dev_idx = slot;
    P.host_src = slot;
    return GQE_OK;
  }
  // ---- copy mode: the rest of this iteration's group in one upload ----
  if (!ctx->up) {
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->up, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
      HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->plan_ready[k], hipEventDisableTiming));
      HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->plan_free[k], hipEventDisableTiming));
    }
  }
  const long long group = it / kFeedGroup;
  const int g = (int)(group & 1);
  if (!f->grp_pin[g]) HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&f->grp_pin[g]), (size_t)kFeedGroup * L.idx_cap, hipHostMallocDefault));
  HIP_TRY(ctx, hipEventSynchronize(f->grp_ready[g]));   // the previous upload out of this pinned buffer (two groups ago) is long done
  const int k0 = (int)(it % kFeedGroup);
  int k1 = kFeedGroup;   // (never past the end of the run: the samples of a later run are drawn by it — reference streams stop where the run stops)
  if (group * kFeedGroup + k1 > f->run_end) k1 = (int)(f->run_end - group * kFeedGroup);
  for (int k = k0; k < k1; ++k) {
    const int64_t i = group * kFeedGroup + k;
    rc = feeder_build(f, i, burn_in);
    if (rc != GQE_OK) return rc;
    if (f->idx.size() > slot_ints) return fail(ctx, GQE_ERR_WORKSPACE, "index feed of %zu entries exceeds the bound workspace", f->idx.size());
    memcpy(f->grp_pin[g] + (size_t)k * slot_ints, f->idx.data(), f->idx.size() * sizeof(int32_t));
    gqe_feeder::Prepared& Q = f->prep[i % (2 * kFeedGroup)];
    stash(Q, i);
    Q.dev_idx = reinterpret_cast<const int32_t*>(ctx->ws + (size_t)(2 + g * kFeedGroup + k) * L.idx_cap);
    Q.host_src = f->grp_pin[g] + (size_t)k * slot_ints;
  }
  char* dev = ctx->ws + (size_t)(2 + g * kFeedGroup + k0) * L.idx_cap;
  if (f->grp_free_set[g]) HIP_TRY(ctx, hipStreamWaitEvent(ctx->up, f->grp_free[g], 0));   // the kernels of group - 2 have read the buffers
  HIP_TRY(ctx, hipMemcpyAsync(dev, f->grp_pin[g] + (size_t)k0 * slot_ints, (size_t)(k1 - k0) * L.idx_cap, hipMemcpyHostToDevice, ctx->up));
  HIP_TRY(ctx, hipEventRecord(f->grp_ready[g], ctx->up));
  HIP_TRY(ctx, hipStreamWaitEvent(st, f->grp_ready[g], 0));
  return GQE_OK;
}

static int feeder_run_inner(gqe_feeder* f, int64_t first_iteration, int32_t n_iterations, int32_t burn_in, float lr, float beta1, float beta2, float eps,
                            float* losses, void* stream);

int gqe_feeder_run(gqe_feeder* f, int64_t first_iteration, int32_t n_iterations, int32_t burn_in, float lr, float beta1,
                   float beta2, float eps, float* losses, void* stream) {
  if (!f) return GQE_ERR_ARG;
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = feeder_run_inner(f, first_iteration, n_iterations, burn_in, lr, beta1, beta2, eps, losses, stream);
  f->host_run_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}

static int feeder_run_inner(gqe_feeder* f, int64_t first_iteration, int32_t n_iterations, int32_t burn_in, float lr, float beta1, float beta2, float eps,
                            float* losses, void* stream) {
  gqe_ctx* ctx = f->ctx;
  if (n_iterations < 1 || !losses) return fail(ctx, GQE_ERR_ARG, "bad arguments");
  if (!ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_bind_workspace has not been called");
  const bool shard = ctx->shard_on;
  if (shard && !ctx->shard_sess) return fail(ctx, GQE_ERR_STATE, "row-sharded ctx: gqe_shard_open comes before gqe_feeder_run");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t end = first_iteration + n_iterations;
  if (f->next_it >= 0 && (first_iteration != f->next_it || burn_in != f->last_burn_in)) {
    // re-running a range would silently re-use the cached samples (and ignore another burn_in), and a slot of the ring could
    // be overwritten while kernels of the old range still read it: its guard event belongs to another group
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (ctx->up) HIP_TRY(ctx, hipStreamSynchronize(ctx->up));
    for (auto& q : f->prep) q.it = -1;
    for (int k = 0; k < 2; ++k) f->pin_ev_set[k] = f->grp_free_set[k] = false;
  }
  f->next_it = end;
  f->last_burn_in = burn_in;
  f->run_end = end;
  float* const losses0 = losses;
  for (int64_t it = first_iteration; it < end; ++it) {
    losses = losses0 + (it - first_iteration) * f->loss_stride;   // (gqe_feeder_set_loss_stride: a history of the run's losses)
    int rc = feeder_ensure(f, it, burn_in, st);
    if (rc != GQE_OK) return rc;
    if (f->loss_stride > 0 && (int64_t)f->prep[it % (2 * kFeedGroup)].batches.size() + 1 > f->loss_stride)
      return fail(ctx, GQE_ERR_ARG, "iteration of %zu batches: its losses do not fit the loss stride", f->prep[it % (2 * kFeedGroup)].batches.size());
    // one iteration of look-ahead where it pays: lazy Adam (the step's row launch also brings the next feed's rows up to
    // date: one launch instead of two) and row-sharded runs (the next plan is posted before this step runs)
    const bool ahead = (ctx->lazy || shard) && it + 1 < end;
    if (ahead) {
      rc = feeder_ensure(f, it + 1, burn_in, st);
      if (rc != GQE_OK) return rc;
    }
    gqe_feeder::Prepared& P = f->prep[it % (2 * kFeedGroup)];
    gqe_feeder::Prepared& N = f->prep[(it + 1) % (2 * kFeedGroup)];
    if (shard) {
      if (shard_plans_ahead(ctx) == 0) {
        rc = gqe_shard_post(ctx, P.batches.data(), (int32_t)P.batches.size(), P.host_idx.data(), P.n_idx, 1, P.segs.data(), (int32_t)P.segs.size());
        if (rc != GQE_OK) return rc;
      }
      if (ahead) {
        rc = gqe_shard_post(ctx, N.batches.data(), (int32_t)N.batches.size(), N.host_idx.data(), N.n_idx, 1, N.segs.data(), (int32_t)N.segs.size());
        if (rc != GQE_OK) return rc;
      }
      rc = gqe_shard_step(ctx, lr, beta1, beta2, eps, losses, nullptr, nullptr, stream);
      if (rc != GQE_OK) return rc;
      continue;
    }
    // the iteration as one call: the split step where it applies (gqe_train_step), the two-call sequence elsewhere — lazy Adam:
    // with the next iteration's feed declared in front of it, so that the step's row launch (which also carries the pair-GEMM
    // units) brings that feed's rows up to date
    if (ctx->lazy && ahead) {
      rc = gqe_lazy_prefetch(ctx, N.batches.data(), (int32_t)N.batches.size(), N.dev_idx, N.n_idx, 1);
      if (rc != GQE_OK) return rc;
    }
    if (f->sgd) {   // --opt sgd (bio/train.py:59-60): forward / backward, then p -= lr g on what the batches touched
      rc = run_queries(ctx, P.batches.data(), (int32_t)P.batches.size(), P.dev_idx, P.n_idx, 1, true, losses, nullptr, nullptr, stream);
      if (rc == GQE_OK) rc = gqe_sgd_step(ctx, P.segs.data(), (int32_t)P.segs.size(), lr, stream);
    } else {
      rc = gqe_train_step(ctx, P.batches.data(), (int32_t)P.batches.size(), P.dev_idx, P.n_idx, 1, P.segs.data(), (int32_t)P.segs.size(), lr, beta1,
                          beta2, eps, losses, stream);
    }
    if (rc != GQE_OK) return rc;
    // everything that reads the feeds of this group (fused kernel, lazy row launches) is enqueued
    const bool group_done = (it % kFeedGroup) == kFeedGroup - 1 || it == end - 1;
    const int g = (int)((it / kFeedGroup) & 1);
    if (group_done && f->feed_mode == 1 && (it % kFeedGroup) == kFeedGroup - 1) {
      HIP_TRY(ctx, hipEventRecord(f->pin_ev[g], st));
      f->pin_ev_set[g] = true;
    }
    if (group_done && f->feed_mode == 0) {
      HIP_TRY(ctx, hipEventRecord(f->grp_free[g], st));
      f->grp_free_set[g] = true;
    }
  }
  return GQE_OK;
}

int64_t gqe_feeder_queries(gqe_feeder* f) { return f ? f->queries_fed : 0; }

int gqe_feeder_host_seconds(gqe_feeder* f, double* build_s, double* run_s) {
  if (!f || !build_s || !run_s) return GQE_ERR_ARG;
  *build_s = f->host_build_s;
  *run_s = f->host_run_s;
  return GQE_OK;
}

int gqe_feeder_debug_feed(gqe_feeder* f, int64_t iteration, gqe_batch* batches, int32_t max_batches, int32_t* n_batches, int32_t* idx,
                          int64_t max_idx, int64_t* n_idx) {
  if (!f || !n_batches || !n_idx) return GQE_ERR_ARG;
  const gqe_feeder::Prepared& P = f->prep[((iteration % (2 * kFeedGroup)) + 2 * kFeedGroup) % (2 * kFeedGroup)];
  if (P.it != iteration || !P.host_src) return fail(f->ctx, GQE_ERR_STATE, "iteration %lld is not among the prepared ones", (long long)iteration);
  *n_batches = (int32_t)P.batches.size();
  *n_idx = P.n_idx;
  if (batches && max_batches >= *n_batches) std::copy(P.batches.begin(), P.batches.end(), batches);
  if (idx && max_idx >= P.n_idx) memcpy(idx, P.host_src, sizeof(int32_t) * (size_t)P.n_idx);
  return GQE_OK;
}

int gqe_debug_profile(gqe_ctx* ctx, long long* stamps) {
  if (!ctx) return GQE_ERR_ARG;
  ctx->prof = stamps;
  return GQE_OK;
}

int gqe_timing_enable(gqe_ctx* ctx, int32_t stride) {
  if (!ctx) return GQE_ERR_ARG;
  ctx->timing = stride > 0 ? stride : 0;
  for (int k = 0; k < kTimingKinds; ++k) ctx->timing_calls[k] = 0;
  return GQE_OK;
}

int gqe_timing_read(gqe_ctx* ctx, int32_t kernel, float* avg_ms, int32_t* count) {
  if (!ctx || kernel < 0 || kernel >= kTimingKinds || !avg_ms || !count) return GQE_ERR_ARG;
  double total = 0;
  int n = 0;
  for (auto& t : ctx->timed[kernel]) {
    float ms = 0;
    HIP_TRY(ctx, hipEventSynchronize(t.stop));
    HIP_TRY(ctx, hipEventElapsedTime(&ms, t.start, t.stop));
    total += ms;
    ++n;
    ctx->event_pool.push_back(t);
  }
  ctx->timed[kernel].clear();
  *avg_ms = n ? (float)(total / n) : 0.f;
  *count = n;
  return GQE_OK;
}

}  // extern "C"

#include "gqe_shard_step.h"   // gqe_shard_open / post / step / forward / close: the row-sharded step as one call
