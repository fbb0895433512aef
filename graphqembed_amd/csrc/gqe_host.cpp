// gqe_host.cpp — host side of libgqe.so: context, launch planning, pinned staging, C ABI (include/gqe.h).
//
// Per call the host builds a compact "plan" (device batch descriptors, pair-GEMM jobs, optionally the
// int32 index feed), writes it into a pinned ring slot and ships it with ONE hipMemcpyAsync on the
// caller's stream; the kernels are then enqueued on the same stream.  Nothing here synchronises.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "gqe_dev.h"

namespace {

constexpr int kRing = 4;
constexpr int kMaxSlots = 20;  // upper bound of scratch slots any batch can use

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

}  // namespace

struct gqe_ctx {
  gqe_config cfg{};
  float *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr;
  int64_t n_arena = 0;
  char* ws = nullptr;
  int64_t ws_bytes = 0;
  RingSlot ring[kRing];
  int ring_next = 0;
  std::string err;
  bool timing = false;
  std::vector<TimedLaunch> timed[3];
  std::vector<TimedLaunch> event_pool;  // recycled hipEvent pairs (creation is not free)
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

#define HIP_TRY(ctx, call)                                                                         \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
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

// get a pinned ring slot of at least `bytes`; waits only if the slot's previous copy is still running
int ring_acquire(gqe_ctx* ctx, size_t bytes, RingSlot** out) {
  RingSlot& s = ctx->ring[ctx->ring_next];
  ctx->ring_next = (ctx->ring_next + 1) % kRing;
  if (s.in_flight) {
    HIP_TRY(ctx, hipEventSynchronize(s.done));
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
  if (!ctx->timing) return GQE_OK;
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
  if (!ctx->timing) return GQE_OK;
  HIP_TRY(ctx, hipEventRecord(ctx->timed[kind].back().stop, st));
  return GQE_OK;
}

struct Plan {
  std::vector<GqeDevBatch> batches;
  std::vector<GqeGemmJob> jobs;
  int tiles = 0;
  int units = 0;
  int64_t scratch_floats = 0;
};

bool off_ok(const gqe_ctx* ctx, int64_t off, int64_t numel) {
  return off >= 0 && (off % 4) == 0 && off + numel <= ctx->n_arena;
}

// Translate the caller's batches into device descriptors, scratch slots and deferred matrix-gradient jobs.
int build_plan(gqe_ctx* ctx, const gqe_batch* in, int n, int64_t n_idx, bool bwd, int64_t scratch_origin, Plan* plan) {
  const int d = ctx->cfg.dim;
  const bool bil = ctx->cfg.decoder == GQE_DEC_BILINEAR;
  const bool mlp = is_mlp(ctx);
  const int64_t vec = bil ? (int64_t)d * d : d;
  int64_t scratch = scratch_origin;
  for (int bi = 0; bi < n; ++bi) {
    const gqe_batch& s = in[bi];
    const int na = anchors_of(s.qtype);
    if (na < 0) return fail(ctx, GQE_ERR_ARG, "batch %d: unknown query type %d", bi, s.qtype);
    if (s.n_queries < 1) return fail(ctx, GQE_ERR_ARG, "batch %d: empty batch (n_queries=%d)", bi, s.n_queries);
    if (s.n_anchors != na) return fail(ctx, GQE_ERR_ARG, "batch %d: query type %d needs %d anchors, got %d", bi, s.qtype, na, s.n_anchors);
    const bool chain = s.qtype <= GQE_Q_3CHAIN;
    const int64_t need_idx = (int64_t)s.idx_offset + (int64_t)(na + (bwd ? 2 : 1)) * s.n_queries;
    if (s.idx_offset < 0 || need_idx > n_idx) return fail(ctx, GQE_ERR_ARG, "batch %d: index range [%d,%lld) exceeds the %lld indices given", bi, s.idx_offset, (long long)need_idx, (long long)n_idx);
    GqeDevBatch b;
    memset(&b, 0xff, sizeof b);  // all slots / params = -1
    b.qtype = s.qtype;
    b.B = s.n_queries;
    b.n_anchors = na;
    b.idx_offset = s.idx_offset;
    b.tile_begin = plan->tiles;
    b.out_offset = s.out_offset;
    b.has_neg = bwd ? 1 : 0;
    b.n_final = 0;
    b.Bpad = (int)align_up(s.n_queries, GQE_TQ);
    b.margin = s.margin;
    b.loss_weight = s.loss_weight;
    b.inv_B = 1.0f / (float)s.n_queries;
    b.grad_scale = s.loss_weight / (float)s.n_queries;
    if (!off_ok(ctx, s.target_table, d)) return fail(ctx, GQE_ERR_ARG, "batch %d: target_table offset %lld outside the arena", bi, (long long)s.target_table);
    b.target_table = s.target_table;
    const int nbr = chain ? 1 : na;
    for (int i = 0; i < GQE_MAX_BRANCH; ++i) b.n_hops[i] = 0;
    for (int i = 0; i < na; ++i) {
      if (!off_ok(ctx, s.anchor_table[i], d)) return fail(ctx, GQE_ERR_ARG, "batch %d: anchor_table[%d] outside the arena", bi, i);
      b.anchor_table[i] = s.anchor_table[i];
    }
    for (int i = 0; i < nbr; ++i) {
      const int nh = s.n_hops[i];
      const int max_h = chain ? (s.qtype + 1) : ((s.qtype == GQE_Q_3INTER_CHAIN && i == 1) ? 2 : 1);
      if (nh != max_h) return fail(ctx, GQE_ERR_ARG, "batch %d: branch %d has %d hops, query type %d needs %d", bi, i, nh, s.qtype, max_h);
      b.n_hops[i] = nh;
      for (int h = 0; h < nh; ++h) {
        if (!off_ok(ctx, s.hop_param[i][h], vec)) return fail(ctx, GQE_ERR_ARG, "batch %d: hop_param[%d][%d] outside the arena", bi, i, h);
        b.hop_param[i][h] = s.hop_param[i][h];
      }
    }
    if (!chain) {
      if (s.qtype == GQE_Q_3CHAIN_INTER) {
        if (s.n_final != 1 || !off_ok(ctx, s.final_param, vec)) return fail(ctx, GQE_ERR_ARG, "batch %d: 3-chain_inter needs one final projection", bi);
        b.n_final = 1;
        b.final_param = s.final_param;
      } else if (s.n_final != 0) {
        return fail(ctx, GQE_ERR_ARG, "batch %d: only 3-chain_inter has a final projection", bi);
      }
      if (mlp) {
        if (!off_ok(ctx, s.pre_param, (int64_t)d * d) || !off_ok(ctx, s.post_param, (int64_t)d * d))
          return fail(ctx, GQE_ERR_ARG, "batch %d: pre/post matrices outside the arena", bi);
        b.pre_param = s.pre_param;
        b.post_param = s.post_param;
      }
    }
    // ---- scratch slots + deferred dM jobs (training only) ----
    int nslot = 0;
    b.scratch_base = scratch;
    const int64_t slot_floats = (int64_t)b.Bpad * d;
    auto slot_off = [&](int slot) { return b.scratch_base + (int64_t)slot * slot_floats; };
    auto add_job = [&](int64_t param, int Lslot, int Rslot) {
      GqeGemmJob j;
      j.param_off = param;
      j.L_off = slot_off(Lslot);
      j.R_off = slot_off(Rslot);
      j.K = b.Bpad;
      const int chunks = (b.Bpad + GQE_GEMM_KCHUNK - 1) / GQE_GEMM_KCHUNK;
      j.unit_begin = plan->units;
      plan->units += chunks * (d / 16) * (d / 16);
      j.unit_end = plan->units;
      j.pad = 0;
      plan->jobs.push_back(j);
    };
    if (bwd) {
      if (chain && bil) {
        for (int sde = 0; sde < 2; ++sde)
          for (int h = 0; h < b.n_hops[0]; ++h) {
            b.slot_act[sde][h] = nslot++;
            b.slot_gact[sde][h] = nslot++;
            // act_{h+1} = act_h M_h  =>  dM_h += act_h^T g_{h+1}
            add_job(b.hop_param[0][h], b.slot_act[sde][h], b.slot_gact[sde][h]);
          }
      }
      if (!chain) {
        if (bil) {
          for (int i = 0; i < na; ++i)
            for (int h = 0; h < b.n_hops[i]; ++h) {
              b.slot_x[i][h] = nslot++;
              b.slot_gy[i][h] = nslot++;
              // y = M x  =>  dM += g_y x^T
              add_job(b.hop_param[i][h], b.slot_gy[i][h], b.slot_x[i][h]);
            }
          if (b.n_final) {
            b.slot_fx = nslot++;
            b.slot_fg = nslot++;
            add_job(b.final_param, b.slot_fg, b.slot_fx);
          }
        }
        if (mlp) {
          for (int i = 0; i < na; ++i) {
            b.slot_e[i] = nslot++;
            b.slot_gz[i] = nslot++;
            add_job(b.pre_param, b.slot_gz[i], b.slot_e[i]);  // z = Pre e  => dPre += g_z e^T
          }
          b.slot_hh = nslot++;
          b.slot_gq = nslot++;
          add_job(b.post_param, b.slot_gq, b.slot_hh);        // q = Post h => dPost += g_q h^T
        }
      }
    }
    if (nslot > kMaxSlots) return fail(ctx, GQE_ERR_ARG, "internal: %d scratch slots", nslot);
    scratch += (int64_t)nslot * slot_floats;
    plan->tiles += b.Bpad / GQE_TQ;
    plan->batches.push_back(b);
  }
  plan->scratch_floats = scratch - scratch_origin;
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
  const int d = ctx->cfg.dim;

  // workspace: [plan bytes | idx (if staged) | pad to 256 B | scratch floats]
  Plan plan;
  const size_t plan_cap = align_up(sizeof(GqeDevBatch) * GQE_MAX_BATCHES, 256) + align_up(sizeof(GqeGemmJob) * GQE_MAX_BATCHES * 16, 256);
  const size_t idx_bytes = idx_on_device ? 0 : (size_t)n_idx * sizeof(int32_t);
  const size_t scratch_origin_bytes = align_up(plan_cap + idx_bytes, 256);
  int rc = build_plan(ctx, batches, n_batches, n_idx, bwd, (int64_t)(scratch_origin_bytes / sizeof(float)), &plan);
  if (rc != GQE_OK) return rc;
  const size_t need = scratch_origin_bytes + (size_t)plan.scratch_floats * sizeof(float);
  const int64_t usable = ctx->ws_bytes - (int64_t)align_up(sizeof(GqeDevSeg) * GQE_MAX_SEGS, 256);
  if ((int64_t)need > usable)
    return fail(ctx, GQE_ERR_WORKSPACE, "workspace too small: need %zu bytes, usable %lld", need, (long long)usable);

  const size_t batch_bytes = sizeof(GqeDevBatch) * plan.batches.size();
  const size_t jobs_off = align_up(sizeof(GqeDevBatch) * GQE_MAX_BATCHES, 256);
  const size_t jobs_bytes = sizeof(GqeGemmJob) * plan.jobs.size();
  if (jobs_off + jobs_bytes > plan_cap) return fail(ctx, GQE_ERR_ARG, "too many deferred matrix-gradient jobs (%zu)", plan.jobs.size());
  const size_t idx_off = plan_cap;
  const size_t copy_bytes = idx_on_device ? (jobs_bytes ? jobs_off + jobs_bytes : batch_bytes) : idx_off + idx_bytes;
  RingSlot* slot;
  rc = ring_acquire(ctx, copy_bytes, &slot);
  if (rc != GQE_OK) return rc;
  memcpy(slot->host, plan.batches.data(), batch_bytes);
  if (jobs_bytes) memcpy(slot->host + jobs_off, plan.jobs.data(), jobs_bytes);
  if (!idx_on_device) memcpy(slot->host + idx_off, idx, idx_bytes);
  HIP_TRY(ctx, hipMemcpyAsync(ctx->ws, slot->host, copy_bytes, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipEventRecord(slot->done, st));
  slot->in_flight = true;

  const GqeDevBatch* d_batches = reinterpret_cast<const GqeDevBatch*>(ctx->ws);
  const GqeGemmJob* d_jobs = reinterpret_cast<const GqeGemmJob*>(ctx->ws + jobs_off);
  const int32_t* d_idx = idx_on_device ? idx : reinterpret_cast<const int32_t*>(ctx->ws + idx_off);
  float* d_ws = reinterpret_cast<float*>(ctx->ws);
  if (bwd) HIP_TRY(ctx, hipMemsetAsync(losses, 0, sizeof(float) * (n_batches + 1), st));

  rc = timing_begin(ctx, 0, st);
  if (rc != GQE_OK) return rc;
  HIP_TRY(ctx, gqe_launch_fused(ctx->cfg.decoder, is_mlp(ctx) ? 1 : 0, is_min(ctx) ? 1 : 0, bwd, plan.tiles, st, d_batches,
                                n_batches, ctx->params, ctx->grads, d_ws, d_idx, d, losses, pos, neg));
  rc = timing_end(ctx, 0, st);
  if (rc != GQE_OK) return rc;
  if (bwd && plan.units > 0) {
    rc = timing_begin(ctx, 1, st);
    if (rc != GQE_OK) return rc;
    HIP_TRY(ctx, gqe_launch_pair_gemm(plan.units, st, d_jobs, d_ws, ctx->grads, d));
    rc = timing_end(ctx, 1, st);
    if (rc != GQE_OK) return rc;
  }
  return GQE_OK;
}

int run_opt(gqe_ctx* ctx, int mode, const gqe_segment* segs, int32_t n_segs, float lr, float b1, float b2, float eps,
            void* stream) {
  if (!ctx) return GQE_ERR_ARG;
  if (!segs || n_segs < 1 || n_segs > GQE_MAX_SEGS) return fail(ctx, GQE_ERR_ARG, "n_segs must be in [1,%d]", GQE_MAX_SEGS);
  if (!ctx->params || !ctx->grads) return fail(ctx, GQE_ERR_STATE, "parameter / gradient arenas not bound");
  if (mode == 0 && (!ctx->m || !ctx->v)) return fail(ctx, GQE_ERR_STATE, "Adam moment arenas not bound");
  if (!ctx->ws) return fail(ctx, GQE_ERR_STATE, "gqe_bind_workspace has not been called");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  std::vector<GqeDevSeg> ds(n_segs);
  long long chunks = 0;
  for (int i = 0; i < n_segs; ++i) {
    const gqe_segment& s = segs[i];
    if (s.offset < 0 || (s.offset % 4) != 0 || s.numel < 1 || s.offset + s.numel > ctx->n_arena)
      return fail(ctx, GQE_ERR_ARG, "segment %d [%lld,+%lld) outside the arena or misaligned", i, (long long)s.offset, (long long)s.numel);
    if (mode == 0 && s.step < 1) return fail(ctx, GQE_ERR_ARG, "segment %d: Adam step must be >= 1", i);
    ds[i].offset = s.offset;
    ds[i].numel = s.numel;
    ds[i].chunk_begin = chunks;
    chunks += (s.numel + GQE_OPT_CHUNK - 1) / GQE_OPT_CHUNK;
    if (mode == 0) {
      // torch.optim.Adam: step_size = lr / (1 - b1^t); denom = sqrt(v) / sqrt(1 - b2^t) + eps  (python doubles)
      const double bc1 = 1.0 - std::pow((double)b1, (double)s.step);
      const double bc2 = 1.0 - std::pow((double)b2, (double)s.step);
      ds[i].step_size = (float)((double)lr / bc1);
      ds[i].bc2_sqrt = (float)std::sqrt(bc2);
    } else {
      ds[i].step_size = lr;
      ds[i].bc2_sqrt = 1.f;
    }
  }
  // the segment table has its own region at the tail of the workspace
  const size_t seg_bytes = sizeof(GqeDevSeg) * n_segs;
  const size_t seg_off = (size_t)ctx->ws_bytes - align_up(sizeof(GqeDevSeg) * GQE_MAX_SEGS, 256);
  RingSlot* slot;
  int rc = ring_acquire(ctx, seg_bytes, &slot);
  if (rc != GQE_OK) return rc;
  memcpy(slot->host, ds.data(), seg_bytes);
  HIP_TRY(ctx, hipMemcpyAsync(ctx->ws + seg_off, slot->host, seg_bytes, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipEventRecord(slot->done, st));
  slot->in_flight = true;
  rc = timing_begin(ctx, 2, st);
  if (rc != GQE_OK) return rc;
  HIP_TRY(ctx, gqe_launch_opt(mode, st, reinterpret_cast<const GqeDevSeg*>(ctx->ws + seg_off), n_segs, chunks, ctx->params,
                              ctx->grads, ctx->m, ctx->v, lr, b1, b2, eps));
  return timing_end(ctx, 2, st);
}

}  // namespace

extern "C" {

int gqe_abi_version(void) { return GQE_ABI_VERSION; }

const char* gqe_last_error(const gqe_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int gqe_create(const gqe_config* cfg, gqe_ctx** out) {
  if (!cfg || !out) return fail(nullptr, GQE_ERR_ARG, "null argument");
  if (cfg->abi_version != GQE_ABI_VERSION) return fail(nullptr, GQE_ERR_ARG, "ABI version mismatch: caller %d, library %d", cfg->abi_version, GQE_ABI_VERSION);
  if (cfg->dim < 16 || cfg->dim > GQE_MAX_DIM || cfg->dim % 16) return fail(nullptr, GQE_ERR_ARG, "dim must be a multiple of 16 in [16,%d], got %d", GQE_MAX_DIM, cfg->dim);
  if (cfg->decoder < 0 || cfg->decoder > 2) return fail(nullptr, GQE_ERR_ARG, "Metapath decoder not recognized.");
  if (cfg->inter < 0 || cfg->inter > 3) return fail(nullptr, GQE_ERR_ARG, "Intersection decoder not recognized.");
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

int gqe_destroy(gqe_ctx* ctx) {
  if (!ctx) return GQE_OK;
  for (auto& s : ctx->ring) {
    if (s.in_flight) (void)hipEventSynchronize(s.done);
    if (s.done) (void)hipEventDestroy(s.done);
    if (s.host) (void)hipHostFree(s.host);
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
  ctx->params = params;
  ctx->grads = grads;
  ctx->m = exp_avg;
  ctx->v = exp_avg_sq;
  ctx->n_arena = n;
  return GQE_OK;
}

int64_t gqe_workspace_bytes(const gqe_ctx* ctx, int64_t max_queries, int32_t max_batches) {
  if (!ctx || max_queries < 1 || max_batches < 1) return GQE_ERR_ARG;
  const size_t plan_cap = align_up(sizeof(GqeDevBatch) * GQE_MAX_BATCHES, 256) + align_up(sizeof(GqeGemmJob) * GQE_MAX_BATCHES * 16, 256);
  const int64_t rows = max_queries + (int64_t)GQE_TQ * max_batches;
  const size_t idx_bytes = (size_t)rows * (2 + GQE_MAX_BRANCH) * sizeof(int32_t);
  const size_t scratch = (size_t)rows * kMaxSlots * ctx->cfg.dim * sizeof(float);
  const size_t seg_tail = align_up(sizeof(GqeDevSeg) * GQE_MAX_SEGS, 256);
  return (int64_t)(align_up(plan_cap + idx_bytes, 256) + scratch + seg_tail + 256);
}

int gqe_bind_workspace(gqe_ctx* ctx, void* workspace, int64_t bytes) {
  if (!ctx) return GQE_ERR_ARG;
  if (!workspace || bytes < (int64_t)(1 << 16)) return fail(ctx, GQE_ERR_ARG, "workspace is NULL or smaller than 64 KiB");
  if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(ctx, GQE_ERR_ARG, "workspace must be 256-byte aligned");
  ctx->ws = static_cast<char*>(workspace);
  // the last 256-aligned block is reserved for the optimiser's segment table
  ctx->ws_bytes = bytes / 256 * 256;
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

int gqe_adam_step(gqe_ctx* ctx, const gqe_segment* segs, int32_t n_segs, float lr, float beta1, float beta2, float eps, void* stream) {
  return run_opt(ctx, 0, segs, n_segs, lr, beta1, beta2, eps, stream);
}

int gqe_sgd_step(gqe_ctx* ctx, const gqe_segment* segs, int32_t n_segs, float lr, void* stream) {
  return run_opt(ctx, 1, segs, n_segs, lr, 0.f, 0.f, 0.f, stream);
}

int gqe_zero_grads(gqe_ctx* ctx, const gqe_segment* segs, int32_t n_segs, void* stream) {
  return run_opt(ctx, 2, segs, n_segs, 0.f, 0.f, 0.f, 0.f, stream);
}

int gqe_timing_enable(gqe_ctx* ctx, int32_t on) {
  if (!ctx) return GQE_ERR_ARG;
  ctx->timing = on != 0;
  return GQE_OK;
}

int gqe_timing_read(gqe_ctx* ctx, int32_t kernel, float* avg_ms, int32_t* count) {
  if (!ctx || kernel < 0 || kernel > 2 || !avg_ms || !count) return GQE_ERR_ARG;
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
