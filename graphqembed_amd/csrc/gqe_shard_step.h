// gqe_shard_step.h — the row-sharded training step as ONE library call (include/gqe.h: gqe_shard_open / gqe_shard_post /
// gqe_shard_step / gqe_shard_forward / gqe_shard_close).  Included by gqe_host.cpp (same translation unit: it drives the
// internal run_queries / run_opt and the phase entry points).
//
// Who does what
//   * PLANNING stays on the host cores and never touches a GPU or a collective: every rank sorts its step's index feed by
//     owner (gqe_shard_plan's counting sort) and POSTS the result — per-owner counts, the request lists, the parameter
//     tensors its batches touch — on a plan board in POSIX shared memory that the ranks of the node share.  An owner
//     reads the requests addressed to it straight from the board.  Posting never waits for a peer's GPU; a step waits
//     (host-side) only until every peer has posted the same step, which a trainer hides by posting step t + 1 before it
//     runs step t (two board slots).
//   * the position feed and the received requests are read by the kernels from pinned host memory (70-90 KB per step over
//     PCIe: ~2 us; no staging copy, no side stream, no event between kernels);
//   * DATA moves over the transport, on the caller's stream: rows to the requesters, gradient contributions back to the
//     owners (two all-to-alls), the small relation / Pre / Post gradients (one all-reduce).  Built in: RCCL (ncclSend /
//     ncclRecv groups + ncclAllReduce, librccl bound at run time like gqe_allreduce_grads); any other transport through
//     the gqe_transport callbacks (the 2-rank tests drive the SAME code over gloo); world = 1 needs none.
//   * the touched-tensor sets travel with the posts: the optimiser steps the UNION over the ranks, so ranks need not run the
//     same formulas (the phase API's contract), and replicated tensors stay bit-identical.
#ifndef GQE_SHARD_STEP_H
#define GQE_SHARD_STEP_H

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#define GQE_SHARD_MAX_WORLD 64
#define GQE_SHARD_MAX_SEGS 512
#define GQE_SHARD_SLOTS 2
#define GQE_SHARD_PINS 8   // ring of pinned feed buffers: a buffer is re-used 8 steps later, so the host never waits for the GPU to
                           // finish a step it has just enqueued (with one buffer per board slot the post of step t + 1 waited for
                           // step t - 1 to COMPLETE: the host could not run ahead and every step paid its enqueue time serially)
#define GQE_SHARD_MAGIC 0x4751455f53484431ull   // "GQE_SHD1"
#define GQE_SHARD_WAIT_SECONDS 120.0

namespace {

struct ShardBoardHeader {
  std::atomic<uint64_t> magic;  // written last by the creating rank
  uint32_t world, slots;
  uint64_t cap_req, post_bytes;
};

struct ShardPost {  // one per (rank, slot); followed by int32 requests[cap_req], grouped by owner
  std::atomic<uint64_t> seq;  // step + 1 once the post is complete
  int32_t kind;               // 0 forward, 1 margin
  int32_t n_segs;
  int64_t counts[GQE_SHARD_MAX_WORLD];
  int64_t seg_off[GQE_SHARD_MAX_SEGS], seg_numel[GQE_SHARD_MAX_SEGS];
};

typedef int (*nccl_group_fn)(void);
typedef int (*nccl_sendrecv_fn)(void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);

struct ShardPlanSlot {
  bool posted = false;
  uint64_t step = 0;
  int kind = 0;
  std::vector<gqe_batch> batches;
  int64_t n_idx = 0;
  std::vector<gqe_segment> segs;
  int64_t send_counts[GQE_SHARD_MAX_WORLD];
  int64_t n_send = 0;
  int pin = 0;                  // which pinned buffer set holds this plan's feeds
  // the owner sort runs on the session's planning thread: gqe_shard_post hands it over and returns
  const int32_t* idx = nullptr; // the caller's host feed (has to stay valid until the plan is run)
  bool with_neg = false;
  std::atomic<int> planned{0};  // 0: queued / being planned, 1: posted on the board (or failed: rc below)
  int plan_rc = 0;
  char plan_err[256] = "";
};

struct ShardPins {
  int32_t* pos = nullptr;       // position feed (pinned host, read by the fused kernel)
  int32_t* req = nullptr;       // requests this rank received, in source-rank order (pinned host, read by serve / link / rows)
  hipEvent_t done = nullptr;    // everything that reads the two buffers has run
  bool done_set = false;
};

}  // namespace

struct ShardSession {
  // host time per phase, accumulated when GQE_SHARD_PROFILE is set (printed by gqe_shard_close)
  bool profile = false;
  bool own_direct = false;  // margin steps name the rows of the own shard directly (GQE_OWN_ROW): whenever the own block stays in place
  bool self_rccl = false;   // GQE_SHARD_SELF_VIA_RCCL: send this rank's own block through RCCL too (measuring / testing the transport with one rank)
  double host_us[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long host_n = 0;
  int world = 1, rank = 0;
  std::string name;
  bool owner = false;
  char* base = nullptr;
  size_t bytes = 0;
  ShardBoardHeader* hdr = nullptr;
  size_t post_bytes = 0;
  int64_t cap_req = 0;
  gqe_transport tr{};
  bool custom = false;
  void* comm = nullptr;
  nccl_group_fn group_start = nullptr, group_end = nullptr;
  nccl_sendrecv_fn send = nullptr, recv = nullptr;
  nccl_allreduce_fn allreduce = nullptr;
  ShardPlanSlot slot[GQE_SHARD_SLOTS];
  ShardPins pins[GQE_SHARD_PINS];
  uint64_t next_post = 0, next_run = 0;
  // a step failed AFTER its plan was consumed: the plans posted ahead no longer line up with what the caller believes is next
  // (and the peers may be mid-exchange), so every later post / step is refused until the session is closed and re-opened
  bool poisoned = false;
  std::string poison_why;
  // "everything up to step e has run" events, recorded every 4th step (two kept): what a pinned buffer set waits for before it
  // is re-used GQE_SHARD_PINS steps later — an event per step cost the host 1.6 us of each
  hipEvent_t ring_ev[2] = {nullptr, nullptr};
  uint64_t ring_ev_step[2] = {0, 0};
  bool ring_ev_set[2] = {false, false};
  std::vector<std::pair<int64_t, int64_t>> uni;   // scratch of shard_collect: union of the touched tensors (no per-step allocation)
  // planning thread (one per session): takes the step numbers gqe_shard_post queues
  gqe_ctx* ctx = nullptr;
  std::thread planner;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<uint64_t> queue;
  bool stop = false;
  ShardPost* post(int r, int s) const { return reinterpret_cast<ShardPost*>(base + sizeof(ShardBoardHeader) + ((size_t)r * GQE_SHARD_SLOTS + s) * post_bytes); }
  int32_t* requests(int r, int s) const { return reinterpret_cast<int32_t*>(reinterpret_cast<char*>(post(r, s)) + sizeof(ShardPost)); }
  std::atomic<uint64_t>* ack(int reader, int writer, int s) const {
    return reinterpret_cast<std::atomic<uint64_t>*>(base + sizeof(ShardBoardHeader) + (size_t)world * GQE_SHARD_SLOTS * post_bytes) +
           ((size_t)reader * world + writer) * GQE_SHARD_SLOTS + s;
  }
};

namespace {

struct ShardClock {   // adds the time since the last mark to slot k
  ShardSession* S;
  std::chrono::steady_clock::time_point t;
  explicit ShardClock(ShardSession* s) : S(s), t(std::chrono::steady_clock::now()) {}
  void mark(int k) {
    if (!S->profile) return;
    const auto n = std::chrono::steady_clock::now();
    S->host_us[k] += std::chrono::duration<double, std::micro>(n - t).count();
    t = n;
  }
};
const char* const kShardPhase[10] = {"post: owner sort", "post: publish", "step: collect", "step: serve launch", "step: rows exchange",
                                     "step: fused + GEMM launches", "step: contributions exchange", "step: link + all-reduces",
                                     "step: optimiser", "step: event"};

size_t shard_board_bytes(int world, int64_t cap_req, size_t* post_bytes) {
  *post_bytes = align_up(sizeof(ShardPost) + sizeof(int32_t) * (size_t)cap_req, 64);
  return sizeof(ShardBoardHeader) + (size_t)world * GQE_SHARD_SLOTS * *post_bytes +
         sizeof(std::atomic<uint64_t>) * (size_t)world * world * GQE_SHARD_SLOTS;
}

template <class Pred>
bool shard_wait(Pred ready) {
  if (ready()) return true;
  const auto t0 = std::chrono::steady_clock::now();
  for (long spins = 0;; ++spins) {
    if (ready()) return true;
    if (spins > 2000) sched_yield();
    if ((spins & 1023) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > GQE_SHARD_WAIT_SECONDS) return false;
  }
}

void shard_session_free(ShardSession* S) {
  if (!S) return;
  if (S->planner.joinable()) {
    {
      std::lock_guard<std::mutex> lk(S->mu);
      S->stop = true;
    }
    S->cv.notify_all();
    S->planner.join();
  }
  if (S->profile && S->host_n > 0) {
    double tot = 0;
    for (double v : S->host_us) tot += v;
    fprintf(stderr, "[gqe shard profile] rank %d, %lld steps: host time per step %.1f us\n", S->rank, S->host_n, tot / (double)S->host_n);
    for (int k = 0; k < 10; ++k) fprintf(stderr, "[gqe shard profile]   %-34s %7.1f us\n", kShardPhase[k], S->host_us[k] / (double)S->host_n);
  }
  if (S->next_run > 0) (void)hipDeviceSynchronize();   // steps behind the last recorded event may still read the pinned feeds
  for (int k = 0; k < 2; ++k)
    if (S->ring_ev[k]) {
      if (S->ring_ev_set[k]) (void)hipEventSynchronize(S->ring_ev[k]);
      (void)hipEventDestroy(S->ring_ev[k]);
    }
  for (auto& pn : S->pins) {
    if (pn.done) {
      if (pn.done_set) (void)hipEventSynchronize(pn.done);
      (void)hipEventDestroy(pn.done);
    }
    if (pn.pos) (void)hipHostFree(pn.pos);
    if (pn.req) (void)hipHostFree(pn.req);
  }
  if (S->base) {
    if (S->name.empty())
      free(S->base);
    else {
      munmap(S->base, S->bytes);
      if (S->owner) shm_unlink(S->name.c_str());
    }
  }
  delete S;
}

// ---- planning thread ----------------------------------------------------------------------------------------
// One step's plan: wait until every peer has read the board slot's previous post, sort the feed by owner straight into the
// pinned position buffer and the board's request area, publish.  Runs next to the caller's thread, which meanwhile
// enqueues the previous step; touches only the slot, the board and read-only parts of the ctx.
void shard_plan_job(ShardSession* S, uint64_t t) {
  const int s = (int)(t % GQE_SHARD_SLOTS), W = S->world, me = S->rank;
  ShardPlanSlot& sl = S->slot[s];
  ShardPins& pn = S->pins[sl.pin];
  ShardClock clk(S);
  int rc = GQE_OK;
  if (t >= GQE_SHARD_SLOTS) {
    for (int j = 0; j < W && rc == GQE_OK; ++j)
      if (!shard_wait([&] { return S->ack(j, me, s)->load(std::memory_order_acquire) >= t + 1 - GQE_SHARD_SLOTS; })) {
        snprintf(sl.plan_err, sizeof sl.plan_err, "row-sharded post %llu: rank %d has not consumed step %llu within %.0f s", (unsigned long long)t, j,
                 (unsigned long long)(t - GQE_SHARD_SLOTS), GQE_SHARD_WAIT_SECONDS);
        rc = GQE_ERR_STATE;
      }
  }
  ShardPost* P = S->post(me, s);
  if (rc == GQE_OK)
    rc = shard_plan_impl(S->ctx, sl.batches.data(), (int32_t)sl.batches.size(), sl.idx, sl.n_idx, sl.with_neg ? 1 : 0, pn.pos, S->requests(me, s),
                         sl.send_counts, sl.plan_err, sizeof sl.plan_err, S->own_direct);
  clk.mark(0);
  sl.n_send = 0;
  for (int o = 0; o < W; ++o) {
    P->counts[o] = rc == GQE_OK ? sl.send_counts[o] : 0;
    sl.n_send += P->counts[o];
  }
  P->kind = rc == GQE_OK ? (sl.with_neg ? 1 : 0) : -1;   // a failed plan is published too: the peers must not wait for it forever
  P->n_segs = sl.with_neg && rc == GQE_OK ? (int32_t)sl.segs.size() : 0;
  for (int k = 0; k < P->n_segs; ++k) {
    P->seg_off[k] = sl.segs[(size_t)k].offset;
    P->seg_numel[k] = sl.segs[(size_t)k].numel;
  }
  P->seq.store(t + 1, std::memory_order_release);
  sl.plan_rc = rc;
  clk.mark(1);
  sl.planned.store(1, std::memory_order_release);
}

void shard_planner_main(ShardSession* S) {
  for (;;) {
    uint64_t t;
    {
      std::unique_lock<std::mutex> lk(S->mu);
      S->cv.wait(lk, [&] { return S->stop || !S->queue.empty(); });
      if (S->queue.empty()) return;   // stop requested and nothing left
      t = S->queue.front();
      S->queue.erase(S->queue.begin());
    }
    shard_plan_job(S, t);
  }
}

// ---- transports -------------------------------------------------------------------------------------------
// all-to-all of variable blocks (elem_bytes each), blocks contiguous in peer order on both sides
// own_in_place: this rank's own block is not moved (the caller arranged for producer and consumer to meet in place)
// `dense` (the contributions' exchange of a margin step): every rank's relation / Pre / Post gradient goes to every peer in the
// SAME exchange — with RCCL as further sends / receives of the one group (the spans straight out of the gradient arena, block p
// of the receive region for peer p), with a callback transport as a second all-to-all of staged copies.  The caller sums the
// blocks in rank order (gqe_launch_dense_sum): what the all-reduce did, without a third collective.
struct ShardDense {
  const GqeSpans* sp = nullptr;
  float* grads = nullptr;
  float* recv = nullptr;    // [world][stride]
  float* stage = nullptr;   // [world][stride] (callback transports)
  int64_t stride = 0;
};

int shard_all_to_all(gqe_ctx* ctx, const void* send, const int64_t* send_counts, void* recv, const int64_t* recv_counts, int64_t elem_bytes,
                     bool own_in_place, hipStream_t st, const ShardDense* dense = nullptr) {
  ShardSession* S = ctx->shard_sess;
  if (S->custom) {
    int rc = S->tr.all_to_all(S->tr.user, send, send_counts, recv, recv_counts, elem_bytes, st);
    if (rc != 0) return fail(ctx, GQE_ERR_HIP, "transport all_to_all failed with %d", rc);
    if (dense && S->world > 1) {
      HIP_TRY(ctx, gqe_launch_dense_stage(*dense->sp, dense->grads, dense->stage, dense->stride, S->world, st));
      int64_t ones[GQE_SHARD_MAX_WORLD];
      for (int p = 0; p < S->world; ++p) ones[p] = 1;
      rc = S->tr.all_to_all(S->tr.user, dense->stage, ones, dense->recv, ones, dense->stride * 4, st);
      if (rc != 0) return fail(ctx, GQE_ERR_HIP, "transport all_to_all (dense gradients) failed with %d", rc);
    }
    return GQE_OK;
  }
  if (!S->comm) {  // world = 1 without a communicator: the block goes from the send to the receive buffer
    if (send_counts[0] > 0 && !own_in_place)
      HIP_TRY(ctx, hipMemcpyAsync(recv, send, (size_t)(send_counts[0] * elem_bytes), hipMemcpyDeviceToDevice, st));
    return GQE_OK;
  }
  const int nccl_float32 = 7;   // ncclDataType_t (rccl.h); every block is rows of floats
  // this rank's own block never leaves the device: a plain copy on the stream (1 / world of the rows, and everything
  // when world = 1); the other blocks travel as one ncclSend / ncclRecv group
  int64_t so = 0, ro = 0, others = 0;
  for (int p = 0; p < S->world; ++p)
    if (p != S->rank || S->self_rccl) others += send_counts[p] + recv_counts[p] + (dense ? 1 : 0);
  int nr = 0, ne = 0;
  if (others > 0) nr = S->group_start();
  for (int p = 0; p < S->world && nr == 0; ++p) {
    if (p == S->rank && !S->self_rccl) {
      if (send_counts[p] != recv_counts[p]) return fail(ctx, GQE_ERR_STATE, "row-sharded exchange: this rank's own block has two sizes");
      if (send_counts[p] > 0 && !own_in_place)
        HIP_TRY(ctx, hipMemcpyAsync(static_cast<char*>(recv) + ro * elem_bytes, static_cast<const char*>(send) + so * elem_bytes,
                                    (size_t)(send_counts[p] * elem_bytes), hipMemcpyDeviceToDevice, st));
    } else {
      if (send_counts[p] > 0)
        nr = S->send(const_cast<char*>(static_cast<const char*>(send)) + so * elem_bytes, (size_t)(send_counts[p] * elem_bytes / 4), nccl_float32, p, S->comm, st);
      if (nr == 0 && recv_counts[p] > 0)
        nr = S->recv(static_cast<char*>(recv) + ro * elem_bytes, (size_t)(recv_counts[p] * elem_bytes / 4), nccl_float32, p, S->comm, st);
    }
    so += send_counts[p];
    ro += recv_counts[p];
    if (dense && nr == 0 && (p != S->rank || S->self_rccl)) {   // (this rank's own term is read from the arena by the sum)
      int64_t at = 0;
      for (int k = 0; k < dense->sp->n && nr == 0; ++k) {
        nr = S->send(dense->grads + dense->sp->off[k], (size_t)dense->sp->len[k], nccl_float32, p, S->comm, st);
        if (nr == 0) nr = S->recv(dense->recv + (int64_t)p * dense->stride + at, (size_t)dense->sp->len[k], nccl_float32, p, S->comm, st);
        at += dense->sp->len[k];
      }
    }
  }
  if (others > 0) ne = S->group_end();
  if (nr != 0 || ne != 0) return fail(ctx, GQE_ERR_HIP, "ncclSend / ncclRecv group failed with ncclResult_t %d / %d", nr, ne);
  return GQE_OK;
}

int shard_all_reduce(gqe_ctx* ctx, float* buf, int64_t n, hipStream_t st) {
  ShardSession* S = ctx->shard_sess;
  if (n < 1) return GQE_OK;
  if (S->custom) {
    const int rc = S->tr.all_reduce_sum_f32(S->tr.user, buf, n, st);
    if (rc != 0) return fail(ctx, GQE_ERR_HIP, "transport all_reduce failed with %d", rc);
    return GQE_OK;
  }
  if (!S->comm) return GQE_OK;   // a single rank without a communicator
  const int nr = S->allreduce(buf, buf, (size_t)n, 7 /* ncclFloat32 */, 0 /* ncclSum */, S->comm, st);
  if (nr != 0) return fail(ctx, GQE_ERR_HIP, "ncclAllReduce failed with ncclResult_t %d", nr);
  return GQE_OK;
}

// ---- one posted plan -> the launches of its step ----------------------------------------------------------
struct ShardCollected {
  int64_t recv_counts[GQE_SHARD_MAX_WORLD];
  int64_t n_recv = 0;
  std::vector<gqe_segment> segs;   // union of the ranks' touched tensors (arena order)
};

// wait until every rank has posted step t, gather the requests addressed to this rank (source-rank order) and the union
// of the touched tensors, acknowledge
int shard_collect(gqe_ctx* ctx, ShardSession* S, uint64_t t, int s, int kind, ShardCollected& out) {
  const int W = S->world, me = S->rank;
  std::vector<std::pair<int64_t, int64_t>>& uni = S->uni;
  uni.clear();
  int32_t* dst = S->pins[S->slot[s].pin].req;
  const int64_t cap = ctx->lay.shard_cap_recv;
  for (int j = 0; j < W; ++j) {
    ShardPost* P = S->post(j, s);
    if (!shard_wait([&] { return P->seq.load(std::memory_order_acquire) >= t + 1; }))
      return fail(ctx, GQE_ERR_STATE, "row-sharded step %llu: rank %d did not post its plan within %.0f s", (unsigned long long)t, j, GQE_SHARD_WAIT_SECONDS);
    if (P->kind == -1) return fail(ctx, GQE_ERR_STATE, "row-sharded step %llu: rank %d could not plan its step", (unsigned long long)t, j);
    if (P->seq.load(std::memory_order_acquire) != t + 1 || P->kind != kind)
      return fail(ctx, GQE_ERR_STATE, "row-sharded step %llu: rank %d posted step %llu of kind %d (ranks must run the same sequence of forward / margin steps)",
                  (unsigned long long)t, j, (unsigned long long)P->seq.load() - 1, P->kind);
    int64_t off = 0;
    for (int o = 0; o < me; ++o) off += P->counts[o];
    const int64_t n = P->counts[me];
    if (n < 0 || out.n_recv + n > cap) return fail(ctx, GQE_ERR_WORKSPACE, "row-sharded step: more than %lld rows requested from this rank", (long long)cap);
    memcpy(dst + out.n_recv, S->requests(j, s) + off, sizeof(int32_t) * (size_t)n);
    out.recv_counts[j] = n;
    out.n_recv += n;
    for (int k = 0; k < P->n_segs; ++k) {
      size_t at = 0;
      while (at < uni.size() && uni[at].first != P->seg_off[k]) ++at;   // a few dozen tensors: a scan beats a tree
      if (at < uni.size() && uni[at].second != P->seg_numel[k])
        return fail(ctx, GQE_ERR_ARG, "row-sharded step: ranks disagree about the tensor at offset %lld", (long long)P->seg_off[k]);
      if (at == uni.size()) uni.emplace_back(P->seg_off[k], P->seg_numel[k]);
    }
    S->ack(me, j, s)->store(t + 1, std::memory_order_release);
  }
  std::sort(uni.begin(), uni.end());   // arena order
  out.segs.reserve(uni.size());
  for (auto& kv : uni) out.segs.push_back(gqe_segment{kv.first, kv.second, 0, 0});
  return GQE_OK;
}

int shard_run_consumed(gqe_ctx* ctx, ShardSession* S, uint64_t t, int s, int kind, ShardCollected& col, ShardClock& clk, float lr, float b1,
                       float b2, float eps, float* losses, float* pos, float* neg, void* stream);

int shard_poisoned(gqe_ctx* ctx, ShardSession* S) {
  return fail(ctx, GQE_ERR_STATE, "row-sharded session: an earlier step failed after its plan was consumed (%s): gqe_shard_close and re-open it",
              S->poison_why.c_str());
}

int shard_run(gqe_ctx* ctx, int kind, float lr, float b1, float b2, float eps, float* losses, float* pos, float* neg, void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  ShardSession* S = ctx->shard_sess;
  if (!S) return fail(ctx, GQE_ERR_STATE, "gqe_shard_open has not been called");
  if (!ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_bind_workspace has not been called");
  if (S->poisoned) return shard_poisoned(ctx, S);
  if (S->next_run == S->next_post) return fail(ctx, GQE_ERR_STATE, "no posted plan to run (gqe_shard_post first)");
  const uint64_t t = S->next_run;
  const int s = (int)(t % GQE_SHARD_SLOTS);
  ShardPlanSlot& sl = S->slot[s];
  if (sl.kind != kind) return fail(ctx, GQE_ERR_STATE, "the oldest posted plan is a %s step", sl.kind ? "margin" : "forward");
  if (kind == 1 && !losses) return fail(ctx, GQE_ERR_ARG, "losses buffer is NULL");
  if (kind == 0 && !pos) return fail(ctx, GQE_ERR_ARG, "scores buffer is NULL");
  ShardCollected col;
  memset(col.recv_counts, 0, sizeof col.recv_counts);
  ShardClock clk(S);
  // the planning thread has (normally long) finished this step's owner sort
  if (!shard_wait([&] { return sl.planned.load(std::memory_order_acquire) == 1; })) return fail(ctx, GQE_ERR_STATE, "the planning thread did not finish");
  if (sl.plan_rc != GQE_OK) {
    ++S->next_run;
    sl.posted = false;
    return fail(ctx, sl.plan_rc, "%s", sl.plan_err);
  }
  int rc = shard_collect(ctx, S, t, s, kind, col);
  if (rc != GQE_OK) return rc;
  clk.mark(2);
  ++S->next_run;   // the plan is consumed whatever happens below
  sl.posted = false;
  rc = shard_run_consumed(ctx, S, t, s, kind, col, clk, lr, b1, b2, eps, losses, pos, neg, stream);
  if (rc != GQE_OK) {
    // the caller may catch the error and go on: without this the plan posted ahead would be taken for the next step's, the
    // open hipEvent brackets would never close and the buffer-reuse event of this step would be missing
    S->poisoned = true;
    S->poison_why = ctx->err;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int k = 5; k <= 6; ++k)
      if (ctx->timing_open[k]) {
        (void)hipEventRecord(ctx->timed[k].back().stop, st);
        ctx->timing_open[k] = false;
      }
    if ((t & 3) == 3) {
      const int k = (int)((t >> 2) & 1);
      if (!S->ring_ev[k]) (void)hipEventCreateWithFlags(&S->ring_ev[k], hipEventDisableTiming);
      if (S->ring_ev[k] && hipEventRecord(S->ring_ev[k], st) == hipSuccess) {
        S->ring_ev_step[k] = t;
        S->ring_ev_set[k] = true;
      }
    }
  }
  return rc;
}

int shard_run_consumed(gqe_ctx* ctx, ShardSession* S, uint64_t t, int s, int kind, ShardCollected& col, ShardClock& clk, float lr, float b1,
                       float b2, float eps, float* losses, float* pos, float* neg, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const Layout& L = ctx->lay;
  const int d = ctx->cfg.dim;
  ShardPlanSlot& sl = S->slot[s];
  ShardPins& pn = S->pins[sl.pin];
  int rc;
  float* rows_send = reinterpret_cast<float*>(ctx->ws + L.contrib_off);   // the (idle) entry space doubles as the serve buffer
  float* fetched = reinterpret_cast<float*>(ctx->ws + L.shard_fetch);
  float* csend = reinterpret_cast<float*>(ctx->ws + L.shard_csend);
  float* crecv = reinterpret_cast<float*>(ctx->ws + L.contrib_off);
  // The serve kernel of a margin step pushes this step's entries onto the gradient lists (below): that must not happen on top of
  // contributions an earlier call left linked — checked HERE, before anything is launched (gqe_shard_link's own check comes
  // after the serve kernel and would leave head / next inconsistent behind its error).
  if (kind == 1 && (ctx->entries_used != 0 || ctx->shard_sent))
    return fail(ctx, GQE_ERR_STATE, "row-sharded step: contributions of an earlier call are still pending (step or materialize first)");
  ctx->shard_internal = true;
  // a margin step links while it serves: which entry answers which request is known now, so no link launch is needed between
  // the contributions' all-to-all and the optimiser pass (GQE_SHARD_LINK_LATE=1 keeps the separate launch: A / B runs)
  static const bool link_late = getenv("GQE_SHARD_LINK_LATE") != nullptr;
  ctx->shard_link_early = kind == 1 && !link_late;
  struct Guard { gqe_ctx* c; ~Guard() { c->shard_internal = false; c->shard_link_early = false; c->own_direct = false; } } guard{ctx};
  // ---- rows: the owners bring what they serve up to date (lazy Adam), gather it, and the rows travel to the requesters ----
  rc = timing_begin(ctx, 5, st);
  if (rc != GQE_OK) return rc;
  GqeRowSegs rsegs;   // the rows this rank serves / receives contributions for, as ONE segment of list heads
  memset(&rsegs, 0, sizeof rsegs);
  rsegs.n = 1;
  rsegs.total = (int)col.n_recv;
  rsegs.begin[0] = 0;
  rsegs.begin[1] = (int)col.n_recv;
  rsegs.idx_begin[0] = 0;
  rsegs.tid[0] = -2;
  if (ctx->lazy && lazy_any_dirty(ctx) && col.n_recv > 0) {
    GqeRowsArgs ra;
    lazy_rows_args(ctx, ra, st);
    ra.idx = pn.req;
    ra.segs = rsegs;
    HIP_TRY(ctx, gqe_launch_rows(ra));
  }
  // this rank's own block stays in place (unless a caller-supplied transport moves every block — gqe_transport.skips_own_block
  // = 0 — or the debug switch sends it through RCCL): its rows are served straight into the fetched buffer, its contributions are linked where the fused
  // kernel writes them
  const bool in_place = S->custom ? S->tr.skips_own_block != 0 : !S->self_rccl;
  ctx->own_lo = ctx->own_n = ctx->own_fetch = ctx->own_entry = 0;
  ctx->own_direct = kind == 1 && S->own_direct;
  if (in_place) {
    for (int j = 0; j < S->rank; ++j) {
      ctx->own_lo += col.recv_counts[j];
      ctx->own_fetch += sl.send_counts[j];
    }
    ctx->own_n = col.recv_counts[S->rank];
    ctx->own_entry = L.shard_cap_recv + ctx->own_fetch;
    if (ctx->own_n != sl.send_counts[S->rank]) return fail(ctx, GQE_ERR_STATE, "row-sharded exchange: this rank's own block has two sizes");
  }
  rc = gqe_shard_serve(ctx, pn.req, col.n_recv, rows_send, stream);
  if (rc != GQE_OK) return rc;
  clk.mark(3);
  rc = shard_all_to_all(ctx, rows_send, col.recv_counts, fetched, sl.send_counts, (int64_t)d * 4, in_place, st);
  if (rc != GQE_OK) return rc;
  clk.mark(4);
  rc = timing_end(ctx, 5, st);
  if (rc != GQE_OK) return rc;
  // ---- the fused kernels on the fetched rows ----
  rc = run_queries(ctx, sl.batches.data(), (int32_t)sl.batches.size(), pn.pos, sl.n_idx, 1, kind == 1, losses, pos, neg, stream);
  if (rc != GQE_OK) return rc;
  clk.mark(5);
  if (kind == 1) {
    // ---- contributions to the owners, the small gradients summed over the ranks, Adam on the own shards ----
    rc = timing_begin(ctx, 6, st);
    if (rc != GQE_OK) return rc;
    // the relation / Pre / Post gradients of the ranks ride in the same exchange (GQE_SHARD_DENSE_ALLREDUCE=1: the all-reduce
    // behind it instead, as before round 5)
    static const bool dense_allreduce = getenv("GQE_SHARD_DENSE_ALLREDUCE") != nullptr;
    const GqeSpans dsp = dense_spans(ctx);
    if (dsp.n < 0) return fail(ctx, GQE_ERR_STATE, "row-sharded step: the non-table parameters form more than 8 spans of the arena");
    ShardDense dn;
    const bool dense_rides = !dense_allreduce && (S->world > 1 || (S->self_rccl && S->comm)) && dsp.total > 0 && L.shard_dense_floats >= dsp.total;
    if (dense_rides) {
      dn.sp = &dsp;
      dn.grads = ctx->grads;
      dn.recv = reinterpret_cast<float*>(ctx->ws + L.shard_dense);
      dn.stride = L.shard_dense_floats;
      dn.stage = dn.recv + (size_t)L.shard_dense_floats * (size_t)S->world;
    }
    rc = shard_all_to_all(ctx, csend, sl.send_counts, crecv, col.recv_counts, (int64_t)d * 4, in_place, st, dense_rides ? &dn : nullptr);
    if (rc != GQE_OK) return rc;
    if (dense_rides) HIP_TRY(ctx, gqe_launch_dense_sum(dsp, ctx->grads, dn.recv, dn.stride, S->rank, S->world, st));
    clk.mark(6);
    // the tables the UNION of the ranks' batches names may receive lists (whatever this rank's own batches named)
    ctx->shard_tables.clear();
    for (const gqe_segment& g : col.segs) {
      const int tb = table_of(ctx, g.offset);
      if (tb >= 0 && !is_bag_table(ctx, tb)) ctx->shard_tables.push_back(tb);
    }
    rc = gqe_shard_link(ctx, pn.req, col.n_recv, stream);
    if (rc != GQE_OK) return rc;
    if (S->world > 1 || S->comm) {   // (a single rank over RCCL still issues the calls: tools/shard_overhead_bench.py)
      std::vector<int64_t> bag_offs;
      for (const Bag& bg : ctx->bags)
        if (ctx->tables[(size_t)bg.table].pending) bag_offs.push_back(ctx->tables[(size_t)bg.table].offset);
      if (!bag_offs.empty()) {   // replicated bag tables: lists -> dense gradient, summed over the ranks
        rc = gqe_materialize_tables(ctx, bag_offs.data(), (int32_t)bag_offs.size(), stream);
        if (rc != GQE_OK) return rc;
        for (int64_t off : bag_offs) {
          rc = shard_all_reduce(ctx, ctx->grads + off, ctx->tables[(size_t)table_of(ctx, off)].rows * d, st);
          if (rc != GQE_OK) return rc;
        }
      }
      for (int k = 0; k < dsp.n && !dense_rides; ++k) {
        rc = shard_all_reduce(ctx, ctx->grads + dsp.off[k], dsp.len[k], st);
        if (rc != GQE_OK) return rc;
      }
    }
    rc = timing_end(ctx, 6, st);
    if (rc != GQE_OK) return rc;
    clk.mark(7);
    if (ctx->lazy) {   // the sparse optimiser launch walks the rows this rank received contributions for
      ctx->feed.assign(1, SavedFeed{rsegs});
      ctx->feed_idx = pn.req;
      ctx->feed_buf = -1;
      ctx->feed_valid = col.n_recv > 0;
    }
    if (!col.segs.empty()) {
      rc = run_opt(ctx, GQE_OPT_ADAM, col.segs.data(), (int32_t)col.segs.size(), lr, b1, b2, eps, stream);
      if (rc != GQE_OK) return rc;
    }
    clk.mark(8);
  }
  if ((t & 3) == 3) {   // (steps 3, 7, 11, ...: the buffers of step t are re-used at step t + 8, by when step t | 3 has been enqueued)
    const int k = (int)((t >> 2) & 1);
    if (!S->ring_ev[k]) HIP_TRY(ctx, hipEventCreateWithFlags(&S->ring_ev[k], hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(S->ring_ev[k], st));
    S->ring_ev_step[k] = t;
    S->ring_ev_set[k] = true;
  }
  clk.mark(9);
  ++S->host_n;
  return GQE_OK;
}

}  // namespace

int shard_plans_ahead(gqe_ctx* ctx) { return ctx->shard_sess ? (int)(ctx->shard_sess->next_post - ctx->shard_sess->next_run) : 0; }

void shard_session_free_fwd(gqe_ctx* ctx) {
  shard_session_free(ctx->shard_sess);
  ctx->shard_sess = nullptr;
}

extern "C" {

int gqe_shard_open(gqe_ctx* ctx, const char* session, void* nccl_comm, const gqe_transport* transport) {
  if (!ctx) return GQE_ERR_ARG;
  if (!ctx->shard_on) return fail(ctx, GQE_ERR_STATE, "gqe_set_shard has not been called");
  if (!ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_shard_open comes after gqe_bind_workspace");
  if (ctx->shard_sess) return fail(ctx, GQE_ERR_STATE, "a row-sharded session is already open");
  const int W = ctx->shard_world;
  if (W > GQE_SHARD_MAX_WORLD) return fail(ctx, GQE_ERR_ARG, "gqe_shard_open supports at most %d ranks", GQE_SHARD_MAX_WORLD);
  if (W > 1 && !nccl_comm && !transport) return fail(ctx, GQE_ERR_ARG, "world > 1 needs an RCCL communicator or a transport");
  if (W > 1 && (!session || !*session)) return fail(ctx, GQE_ERR_ARG, "world > 1 needs a session name (the same on every rank)");
  if (transport && (!transport->all_to_all || !transport->all_reduce_sum_f32)) return fail(ctx, GQE_ERR_ARG, "transport callbacks missing");
  ShardSession* S = new ShardSession();
  S->world = W;
  S->rank = ctx->shard_rank;
  S->profile = getenv("GQE_SHARD_PROFILE") != nullptr;
  S->self_rccl = getenv("GQE_SHARD_SELF_VIA_RCCL") != nullptr;
  S->cap_req = ctx->lay.shard_cap_send;
  if (transport) {
    S->tr = *transport;
    S->custom = true;
  } else if (nccl_comm) {
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      delete S;
      return fail(ctx, GQE_ERR_STATE, "cannot load librccl.so: %s", dlerror());
    }
    S->group_start = reinterpret_cast<nccl_group_fn>(dlsym(h, "ncclGroupStart"));
    S->group_end = reinterpret_cast<nccl_group_fn>(dlsym(h, "ncclGroupEnd"));
    S->send = reinterpret_cast<nccl_sendrecv_fn>(dlsym(h, "ncclSend"));
    S->recv = reinterpret_cast<nccl_sendrecv_fn>(dlsym(h, "ncclRecv"));
    S->allreduce = reinterpret_cast<nccl_allreduce_fn>(dlsym(h, "ncclAllReduce"));
    if (!S->group_start || !S->group_end || !S->send || !S->recv || !S->allreduce) {
      delete S;
      return fail(ctx, GQE_ERR_STATE, "librccl.so lacks ncclGroupStart / ncclGroupEnd / ncclSend / ncclRecv / ncclAllReduce");
    }
    S->comm = nccl_comm;
  }
  S->own_direct = (S->custom ? S->tr.skips_own_block != 0 : !S->self_rccl) && getenv("GQE_SHARD_OWN_VIA_BUFFER") == nullptr;   // (the switch: A / B runs)
  // ---- the plan board ----
  S->bytes = shard_board_bytes(W, S->cap_req, &S->post_bytes);
  if (W == 1) {
    S->base = static_cast<char*>(calloc(1, S->bytes));
    if (!S->base) {
      delete S;
      return fail(ctx, GQE_ERR_STATE, "out of memory");
    }
  } else {
    S->name = std::string("/gqe_") + session;
    S->owner = S->rank == 0;
    int fd = -1;
    if (S->owner) {
      shm_unlink(S->name.c_str());   // a stale board of a crashed run
      fd = shm_open(S->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, (off_t)S->bytes) != 0) {
        if (fd >= 0) close(fd);
        const std::string nm = S->name;
        S->name.clear();
        delete S;
        return fail(ctx, GQE_ERR_STATE, "cannot create the plan board %s in shared memory", nm.c_str());
      }
    } else {
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {   // until rank 0 has created and sized it
        fd = shm_open(S->name.c_str(), O_RDWR, 0600);
        struct stat sb;
        if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= S->bytes) break;
        if (fd >= 0) close(fd);
        fd = -1;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > GQE_SHARD_WAIT_SECONDS) break;
        usleep(1000);
      }
      if (fd < 0) {
        const std::string nm = S->name;
        S->name.clear();
        delete S;
        return fail(ctx, GQE_ERR_STATE, "rank 0 did not create the plan board %s", nm.c_str());
      }
    }
    void* m = mmap(nullptr, S->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
      if (S->owner) shm_unlink(S->name.c_str());
      S->name.clear();
      delete S;
      return fail(ctx, GQE_ERR_STATE, "cannot map the plan board");
    }
    S->base = static_cast<char*>(m);
  }
  S->hdr = reinterpret_cast<ShardBoardHeader*>(S->base);
  if (W == 1 || S->owner) {   // (a fresh shared-memory object is zero-filled: seq = ack = 0)
    S->hdr->world = (uint32_t)W;
    S->hdr->slots = GQE_SHARD_SLOTS;
    S->hdr->cap_req = (uint64_t)S->cap_req;
    S->hdr->post_bytes = S->post_bytes;
    S->hdr->magic.store(GQE_SHARD_MAGIC, std::memory_order_release);
  } else {
    if (!shard_wait([&] { return S->hdr->magic.load(std::memory_order_acquire) == GQE_SHARD_MAGIC; }) || (int)S->hdr->world != W ||
        (int64_t)S->hdr->cap_req != S->cap_req) {
      shard_session_free(S);
      return fail(ctx, GQE_ERR_STATE, "the plan board was created for another world size / workspace capacity (every rank must bind the same capacities)");
    }
  }
  S->ctx = ctx;
  S->planner = std::thread(shard_planner_main, S);
  ctx->shard_sess = S;
  return GQE_OK;
}

int gqe_shard_close(gqe_ctx* ctx) {
  if (!ctx) return GQE_ERR_ARG;
  shard_session_free(ctx->shard_sess);
  ctx->shard_sess = nullptr;
  return GQE_OK;
}

int gqe_shard_profile(gqe_ctx* ctx, double us[10], int64_t* steps) {
  if (!ctx || !us || !steps) return GQE_ERR_ARG;
  ShardSession* S = ctx->shard_sess;
  if (!S) return fail(ctx, GQE_ERR_STATE, "gqe_shard_open has not been called");
  for (int k = 0; k < 10; ++k) us[k] = S->host_us[k];
  *steps = S->host_n;
  return GQE_OK;
}

int gqe_shard_post(gqe_ctx* ctx, const gqe_batch* batches, int32_t n_batches, const int32_t* idx, int64_t n_idx, int32_t with_negatives,
                   const gqe_segment* segs, int32_t n_segs) {
  if (!ctx) return GQE_ERR_ARG;
  ShardSession* S = ctx->shard_sess;
  if (!S) return fail(ctx, GQE_ERR_STATE, "gqe_shard_open has not been called");
  if (!ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_bind_workspace has not been called");
  if (S->poisoned) return shard_poisoned(ctx, S);
  if (S->cap_req != ctx->lay.shard_cap_send) return fail(ctx, GQE_ERR_STATE, "the workspace was re-bound with other capacities: close and re-open the session");
  if (S->next_post - S->next_run >= GQE_SHARD_SLOTS) return fail(ctx, GQE_ERR_STATE, "%d plans are already posted: run one first", GQE_SHARD_SLOTS);
  if (!batches || n_batches < 1 || n_batches > GQE_MAX_BATCHES || !idx || n_idx < 1) return fail(ctx, GQE_ERR_ARG, "gqe_shard_post: bad arguments");
  if (n_idx > S->cap_req) return fail(ctx, GQE_ERR_WORKSPACE, "row-sharded mode: %lld indices exceed the bound capacity (%lld rows)", (long long)n_idx, (long long)S->cap_req);
  if (with_negatives && (n_segs < 1 || !segs)) return fail(ctx, GQE_ERR_ARG, "gqe_shard_post: a margin step needs the parameter tensors its batches touch");
  if (n_segs < 0 || n_segs > GQE_SHARD_MAX_SEGS) return fail(ctx, GQE_ERR_ARG, "gqe_shard_post: at most %d tensors per step", GQE_SHARD_MAX_SEGS);
  const uint64_t t = S->next_post;
  const int s = (int)(t % GQE_SHARD_SLOTS), W = S->world, me = S->rank;
  ShardPlanSlot& sl = S->slot[s];
  // the pinned buffers were last used GQE_SHARD_PINS steps ago (their kernels have long run); every peer has read the
  // board slot's previous post (step t - 2)
  sl.pin = (int)(t % GQE_SHARD_PINS);
  ShardPins& pn = S->pins[sl.pin];
  if (t >= GQE_SHARD_PINS) {
    // the step that used these buffers last (t - 8) has run: the OLDER recorded event that covers it, else the newer one
    const uint64_t need = t - GQE_SHARD_PINS;
    int pick = -1;
    for (int k = 0; k < 2; ++k)
      if (S->ring_ev_set[k] && S->ring_ev_step[k] >= need && (pick < 0 || S->ring_ev_step[k] < S->ring_ev_step[pick])) pick = k;
    if (pick < 0) return fail(ctx, GQE_ERR_STATE, "row-sharded post %llu: step %llu has not been run", (unsigned long long)t, (unsigned long long)need);
    if (hipEventQuery(S->ring_ev[pick]) != hipSuccess) HIP_TRY(ctx, hipEventSynchronize(S->ring_ev[pick]));
  }
  if (!pn.pos) {
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&pn.pos), sizeof(int32_t) * (size_t)std::max<int64_t>(S->cap_req, 1), hipHostMallocDefault));
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&pn.req), sizeof(int32_t) * (size_t)std::max<int64_t>(ctx->lay.shard_cap_recv, 1), hipHostMallocDefault));
  }
  for (int k = 0; k < n_segs && with_negatives; ++k)
    if (segs[k].offset < 0 || segs[k].numel < 1 || segs[k].offset + segs[k].numel > ctx->n_arena)
      return fail(ctx, GQE_ERR_ARG, "gqe_shard_post: segment %d outside the arena", k);
  sl.posted = true;
  sl.step = t;
  sl.kind = with_negatives ? 1 : 0;
  sl.with_neg = with_negatives != 0;
  sl.batches.assign(batches, batches + n_batches);
  sl.segs.assign(segs, segs + (with_negatives ? n_segs : 0));
  sl.idx = idx;
  sl.n_idx = n_idx;
  sl.plan_rc = GQE_OK;
  sl.plan_err[0] = 0;
  sl.planned.store(0, std::memory_order_release);
  ++S->next_post;
  {
    std::lock_guard<std::mutex> lk(S->mu);
    S->queue.push_back(t);
  }
  S->cv.notify_one();
  (void)me;
  (void)W;
  return GQE_OK;
}

int gqe_shard_step(gqe_ctx* ctx, float lr, float beta1, float beta2, float eps, float* losses, float* pos_scores, float* neg_scores, void* stream) {
  return shard_run(ctx, 1, lr, beta1, beta2, eps, losses, pos_scores, neg_scores, stream);
}

int gqe_shard_forward(gqe_ctx* ctx, float* scores, void* stream) {
  return shard_run(ctx, 0, 0.f, 0.f, 0.f, 0.f, nullptr, scores, nullptr, stream);
}

}  // extern "C"

#endif
