// gqe_kernels.hip — pair-GEMM (deferred matrix gradients), fused optimiser pass and the dispatcher of
// the fused query kernel (gqe_fused.h, instantiated per decoder variant in gqe_fused_inst.hip).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "gqe_common.h"
#include "gqe_adam.h"
#define GQE_SPLIT_U 1   // the riders of the second launch live in the optimiser kernels' budget (<= 64 VGPRs, eight waves per SIMD: two slices in flight spilled 9 registers there)
#include "gqe_split.h"

// ------------------------------------------------------------------------------------------
// deferred matrix gradients:  dM[i][j] += sum_b L[b][i] * R[b][j]   (rank-B update on the matrix cores)
// one wave per unit = (batch, job, 128-query K chunk, 16x16 output tile); A[i][k=b] = L[b][i0+i],
// B[k=b][j] = R[b][j0+j]: both operands are read straight from the scratch rows (64 B contiguous per 16
// lanes).  Jobs are derived on the device from the cached formula descriptor + the kernel-argument plan.
// The extra last block turns the per-tile hinge sums of the fused kernel into losses[].
// ------------------------------------------------------------------------------------------
// finalize block: per-batch mean hinge loss (model.py:124-126) and the weighted iteration loss from the per-tile partials of
// the fused kernel — plain stores, nothing to zero, no atomics.
__device__ __forceinline__ void finalize_losses(const GqeDynPlan& plan, const float* __restrict__ tile_loss, float* __restrict__ losses) {
  __shared__ float s_w[GQE_LAUNCH_BATCHES];
  const int t = threadIdx.x;
  // one wave per batch: lanes stride the tile partials (independent loads), DPP-reduce
  for (int k = t >> 6; k < plan.n_batches; k += GQE_WAVES) {
    const GqeDynBatch b = plan.b[k];
    float l = 0.f;
    for (int i = t & 63; i < b.Bpad / GQE_TQ; i += 64) l += tile_loss[b.tile_begin + i];
    l = wave_sum(l) * b.inv_B;
    if ((t & 63) == 0) {
      losses[b.loss_index] = l;
      s_w[k] = l * b.loss_weight;
    }
  }
  __syncthreads();
  if (t == 0) {
    float tot = plan.first ? 0.f : losses[plan.total_index];
    for (int k = 0; k < plan.n_batches; ++k) tot += s_w[k];
    losses[plan.total_index] = tot;
  }
}

__device__ __forceinline__ int unit_of_workgroup(int x, int units);

__global__ __launch_bounds__(GQE_THREADS) void gqe_pair_gemm_kernel(const GqeDynPlan plan,
                                                                   const GqeDevFormula* __restrict__ formulas,
                                                                   const float* __restrict__ ws,
                                                                   float* __restrict__ grads, int d,
                                                                   const float* __restrict__ tile_loss,
                                                                   float* __restrict__ losses, long long* __restrict__ prof) {
  // debug profile (gqe_debug_profile): this launch's workgroups stamp behind the fused kernel's tiles
#define GQE_GSTAMP(k)                                                                                                          \
  do {                                                                                                                        \
    if (prof && threadIdx.x == 0) prof[((size_t)plan.tiles + blockIdx.x) * GQE_PROF_SLOTS + (k)] = (long long)wall_clock64(); \
  } while (0)
  GQE_GSTAMP(0);
  if (blockIdx.x == 0) {   // (first, so that it is not queued behind the GEMM units)
    finalize_losses(plan, tile_loss, losses);
    return;
  }
  // one workgroup = one unit (batch, job, K chunk of GQE_GEMM_KCHUNK queries, 64x64 block of the d x d gradient).
  // Both operand panels go through LDS in 64-query halves (float4 global loads, 80-float rows: the four k-rows an
  // MFMA step reads land 16 banks apart); each wave owns a 2x2 group of 16x16 MFMA tiles.
  if ((int)blockIdx.x - 1 >= plan.units) return;
  const int unit = unit_of_workgroup((int)blockIdx.x - 1, plan.units);   // (the blocks of one (job, chunk) on one XCD: defined below)
  constexpr int MT = GQE_GEMM_MT, KS = 64, STR = MT + 16;
  __shared__ float sL[KS * STR], sR[KS * STR];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lq = lane & 15, lk = lane >> 4;
  int bi = 0;
#pragma unroll
  for (int k = 1; k < GQE_LAUNCH_BATCHES; ++k) bi += (unit >= plan.unit_begin[k]) ? 1 : 0;
  const GqeDynBatch b = plan.b[bi];
  const GqeDevFormula* __restrict__ f = formulas + b.formula;
  const int mper = (d + MT - 1) / MT, macros = mper * mper;
  const int kmul = plan.pad[0];   // chunks of GQE_GEMM_KCHUNK queries this unit walks before its atomic pass
  const int chunks = (b.Bpad + GQE_GEMM_KCHUNK * kmul - 1) / (GQE_GEMM_KCHUNK * kmul);
  int u = unit - b.unit_begin;
  const int job = u / (chunks * macros);
  u -= job * chunks * macros;
  const int chunk = u / macros;
  const int mt = u - chunk * macros;
  const int i0 = (mt / mper) * MT, j0 = (mt % mper) * MT;
  const int nib = (min(d - i0, MT)) >> 4, njb = (min(d - j0, MT)) >> 4;  // 16-wide tile columns present in this block
  const int k_first = chunk * GQE_GEMM_KCHUNK * kmul;
  const size_t slot_floats = (size_t)b.Bpad * d;
  const float* L = ws + b.scratch_base + (size_t)f->job_L[job] * slot_floats;
  const float* R = ws + b.scratch_base + (size_t)f->job_R[job] * slot_floats;
  GQE_GSTAMP(1);
  const int ib0 = (wave >> 1) * 2, jb0 = (wave & 1) * 2;
  f32x4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int kc = 0; kc < kmul; ++kc) {
  const int k_begin = k_first + kc * GQE_GEMM_KCHUNK;
  if (k_begin >= b.Bpad) break;
  if (kc) __syncthreads();  // the previous chunk's second half is consumed
  // the whole K chunk is requested up front (16 float4 per thread) and fed through the LDS panels half by half.
  // Out-of-range lanes read the panel's first floats (always mapped) and are zeroed afterwards: a predicated load
  // would fence each load behind its own wait.
  constexpr int NQ = GQE_GEMM_KCHUNK * 16 / GQE_THREADS;  // float4 per thread per operand
  float4 vl[NQ], vr[NQ];
  unsigned okm = 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int idx = tid + q * GQE_THREADS, row = idx >> 4, c4 = (idx & 15) * 4;
    const int k = k_begin + row;
    const bool okk = k < b.Bpad;
    const bool okl = okk && i0 + c4 < d, okr = okk && j0 + c4 < d;
    vl[q] = *reinterpret_cast<const float4*>(L + (okl ? (size_t)k * d + i0 + c4 : 0));
    vr[q] = *reinterpret_cast<const float4*>(R + (okr ? (size_t)k * d + j0 + c4 : 0));
    okm |= (okl ? 1u : 0u) << (2 * q) | (okr ? 2u : 0u) << (2 * q);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (!(okm >> (2 * q) & 1u)) vl[q] = zero4;
    if (!(okm >> (2 * q) & 2u)) vr[q] = zero4;
  }
  GQE_GSTAMP(2);
#pragma unroll
  for (int h = 0; h < GQE_GEMM_KCHUNK / KS; ++h) {
    if (h) __syncthreads();  // the previous half is consumed
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + q * GQE_THREADS, row = idx >> 4, c4 = (idx & 15) * 4;
      *reinterpret_cast<float4*>(sL + row * STR + c4) = vl[h * 4 + q];
      *reinterpret_cast<float4*>(sR + row * STR + c4) = vr[h * 4 + q];
    }
    __syncthreads();
    GQE_GSTAMP(3 + 2 * h);
    if (ib0 < nib && jb0 < njb) {
      // the operands of step s + 1 are requested before the four MFMAs of step s are issued (sched_barrier pins the
      // order): left alone, every step waited a full LDS round trip in front of its MFMAs — 1.5 us per 64-query half
      // instead of the 0.85 us the 64 MFMAs of a wave take
      const float* pl = sL + lk * STR + lq + ib0 * 16;
      const float* pr = sR + lk * STR + lq + jb0 * 16;
      float a0[2], a1[2], b0[2], b1[2];
      a0[0] = pl[0];
      a1[0] = pl[16];
      b0[0] = pr[0];
      b1[0] = pr[16];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < KS / 4; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s + 1 < KS / 4) {
          a0[nxt] = pl[(s + 1) * 4 * STR];
          a1[nxt] = pl[(s + 1) * 4 * STR + 16];
          b0[nxt] = pr[(s + 1) * 4 * STR];
          b1[nxt] = pr[(s + 1) * 4 * STR + 16];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[cur], b0[cur], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[cur], b1[cur], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[cur], b0[cur], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[cur], b1[cur], acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    GQE_GSTAMP(4 + 2 * h);
  }
  }  // chunks of this unit
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      if (ib0 + x >= nib || jb0 + y >= njb) continue;
      float* out = grads + f->job_param[job] + (size_t)(i0 + (ib0 + x) * 16 + 4 * lk) * d + j0 + (jb0 + y) * 16 + lq;
#pragma unroll
      for (int r = 0; r < 4; ++r) unsafeAtomicAdd(out + (size_t)r * d, acc[x][y][r]);
    }
  GQE_GSTAMP(7);
#undef GQE_GSTAMP
}

// ------------------------------------------------------------------------------------------
// fused optimiser pass over a list of parameter tensors (segments).
//   dense segments (relation vectors / matrices): 1024-float chunks, float4 per thread:
//       p,g,m,v -> p,m,v ; g := 0
//   table segments: d/4 threads per embedding row, 256/(d/4) rows per chunk.  The row's gradient is the
//       sum of its contribution list (head[row] -> next[...], usually empty, rarely longer than a few)
//       [+ the dense gradient when a caller materialised it]; the list is reset on the way.  p/m/v stream
//       at 24 B/param, the dense table gradient is neither read nor re-zeroed.
//   MODE = ADAM | SGD | ZERO (drop gradients) | MATERIALIZE (fold the lists into the dense gradient).
// ------------------------------------------------------------------------------------------
// p / m / v of a table are read once and written once per pass.  When the tables outgrow the 256 MB Infinity Cache
// (reddit-synth: 1.7 GB of p / m / v) the pass streams them with the non-temporal policy, which neither allocates in
// nor evicts from L2 / MALL: 610 -> 538 us per pass (5.8 -> 6.6 TB/s) and the next fused kernel's gathers find more of
// their rows cached (129 -> 119 us).  Tables that FIT the cache (bio-synth: 150 MB) keep the default policy — there the
// next pass re-reads what this one wrote from the cache, and nt measured 49.5 -> 54.2 us.
template <bool NT>
__device__ __forceinline__ float4 ld_stream(const float* p) {
  if (NT) {
    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(t[0], t[1], t[2], t[3]);
  }
  return *reinterpret_cast<const float4*>(p);
}
template <bool NT>
__device__ __forceinline__ void st_stream(float* p, const float4& v) {
  if (NT) {
    f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
  } else {
    *reinterpret_cast<float4*>(p) = v;
  }
}

template <int MODE>
__device__ __forceinline__ void opt_update(float4& pp, float4& mm, float4& vv, const float4& gg, float step_size,
                                           float bc2_sqrt, float lr, float b1, float b2, float eps) {
  if (MODE == GQE_OPT_ADAM) {
    const float b1c = 1.f - b1, b2c = 1.f - b2, ibc = __builtin_amdgcn_rcpf(bc2_sqrt);
    gqe_adam1(pp.x, mm.x, vv.x, gg.x, step_size, ibc, b1c, b2, b2c, eps);
    gqe_adam1(pp.y, mm.y, vv.y, gg.y, step_size, ibc, b1c, b2, b2c, eps);
    gqe_adam1(pp.z, mm.z, vv.z, gg.z, step_size, ibc, b1c, b2, b2c, eps);
    gqe_adam1(pp.w, mm.w, vv.w, gg.w, step_size, ibc, b1c, b2, b2c, eps);
  } else {
    pp.x -= lr * gg.x;
    pp.y -= lr * gg.y;
    pp.z -= lr * gg.z;
    pp.w -= lr * gg.w;
  }
}

// Sum of a row's gradient list (float4 slice c4 of every contribution).  len = number of nodes walked.
template <bool SORTED>
__device__ __forceinline__ float4 list_gradient(int h0, const int32_t* __restrict__ next, const float* __restrict__ contrib,
                                                const int32_t* __restrict__ link_contrib, int max_entries, int d, int c4, int& len) {
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 acc = zero4;
  len = 0;
  for (int h = h0; h >= 0; h = next[h]) {
    const int ce = (h < max_entries) ? h : link_contrib[h - max_entries];  // bag link node -> its contribution
    const float4 c = *reinterpret_cast<const float4*>(contrib + (size_t)ce * d + c4);
    acc.x += c.x;
    acc.y += c.y;
    acc.z += c.z;
    acc.w += c.w;
    ++len;
  }
  if (SORTED && len > 2) {
    // list order = arrival order of the atomic exchanges, which differs between replicas.  a + b commutes;
    // longer lists are re-summed order-independently: every term is scaled by a power of two chosen from
    // the list's largest magnitude (a max is order-independent, the scaling exact), rounded to a 64-bit
    // integer with 40 fraction bits below that magnitude, and the integers are added (associative).
    float4 mx = zero4;
    for (int x = h0; x >= 0; x = next[x]) {
      const int ce = (x < max_entries) ? x : link_contrib[x - max_entries];
      const float4 c = *reinterpret_cast<const float4*>(contrib + (size_t)ce * d + c4);
      mx.x = fmaxf(mx.x, fabsf(c.x));
      mx.y = fmaxf(mx.y, fabsf(c.y));
      mx.z = fmaxf(mx.z, fabsf(c.z));
      mx.w = fmaxf(mx.w, fabsf(c.w));
    }
    int ex, ey, ez, ew;
    frexpf(mx.x, &ex);
    frexpf(mx.y, &ey);
    frexpf(mx.z, &ez);
    frexpf(mx.w, &ew);
    ex = max(ex, -80);  // keeps 2^(40 - e) finite for vanishing gradients
    ey = max(ey, -80);
    ez = max(ez, -80);
    ew = max(ew, -80);
    const float sx = ldexpf(1.f, 40 - ex), sy = ldexpf(1.f, 40 - ey), sz = ldexpf(1.f, 40 - ez), sw = ldexpf(1.f, 40 - ew);
    long long ax = 0, ay = 0, az = 0, aw = 0;
    for (int x = h0; x >= 0; x = next[x]) {
      const int ce = (x < max_entries) ? x : link_contrib[x - max_entries];
      const float4 c = *reinterpret_cast<const float4*>(contrib + (size_t)ce * d + c4);
      ax += __float2ll_rn(c.x * sx);
      ay += __float2ll_rn(c.y * sy);
      az += __float2ll_rn(c.z * sz);
      aw += __float2ll_rn(c.w * sw);
    }
    acc.x = (float)ldexp((double)ax, ex - 40);
    acc.y = (float)ldexp((double)ay, ey - 40);
    acc.z = (float)ldexp((double)az, ez - 40);
    acc.w = (float)ldexp((double)aw, ew - 40);
  }
  return acc;
}

// ---- hot rows (GqeHot, gqe_dev.h) -------------------------------------------------------------------------
// the gradient a hot row collected this step: the sum of its replicas, which are re-zeroed on the way
__device__ __forceinline__ float4 hot_take(const GqeHot& hot, int hs, int d, int c4) {
  // (hot rows are a few hundred of ~10^5: two loads in flight, not GQE_HOT_REPS — the streaming pass around this keeps its
  // <= 64 VGPRs, i.e. its occupancy)
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 s = zero4;
  const int reps = GQE_HOT_REPS_OF(hs);   // (the value of slot[] also names the row's sub-lists: such a row uses fewer accumulators)
  hs = GQE_HOT_SLOT_OF(hs);
#pragma unroll 1
  for (int x = 0; x < reps; x += 2) {
    float* p0 = hot.acc + GQE_HOT_ROW(x, hs) * d + c4;
    float* p1 = p0 + d;
    const float4 a = *reinterpret_cast<const float4*>(p0), b = *reinterpret_cast<const float4*>(p1);
    s.x += a.x + b.x;
    s.y += a.y + b.y;
    s.z += a.z + b.z;
    s.w += a.w + b.w;
    *reinterpret_cast<float4*>(p0) = zero4;
    *reinterpret_cast<float4*>(p1) = zero4;
  }
  return s;
}

// a list of `len` entries was just walked for row `hrow`: long enough -> the row's later contributions go to a slot
__device__ __forceinline__ void hot_promote(const GqeHot& hot, long long hrow, int len) {
  if (len < hot.min_len) return;
  // every slot taken: no further increments (on heavy-tailed data thousands of long rows per step would otherwise bump the
  // counter for ever — one contended atomic each, and after ~1e5 steps an int32 wrap that hands slots out twice)
  if (__hip_atomic_load(hot.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= hot.cap) return;
  const int s = __hip_atomic_fetch_add(hot.count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (s < 0 || s >= hot.cap) return;
  // sub-lists for the row's bag entries (gqe_dev.h): ~GQE_HOT_SUB_LEN entries each at the length that promoted the row
  int v = s;
  if (hot.sub) {
    int lg = 0;
    while ((GQE_HOT_SUB_LEN << lg) < len && lg < GQE_HOT_SUB_MAX_LG) ++lg;
    const int n = 1 << lg;
    if (__hip_atomic_load(hot.sub_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + n <= GQE_HOT_SUB_POOL) {
      const int base = __hip_atomic_fetch_add(hot.sub_count, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (base >= 0 && base + n <= GQE_HOT_SUB_POOL) {
        for (int i = 0; i < n; ++i) GQE_HOT_SUB_SLOT(hot.sub)[base + i] = s;
        v = s | ((lg + 1) << 11) | (base << 15);
      }
    }
  }
  if (len < hot.few_len) v |= GQE_HOT_FEW_BIT;
  hot.slot[hrow] = v;
  if (hot.seen) __hip_atomic_store(hot.seen, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // the host starts the gather launches
}

// The sub-lists of the hot word rows (gqe_dev.h), behind a fused launch with bag roles: one wave per sub-list — its counter, its
// entries (lanes i and i + 64 = entries i, i + 64), the contribution rows behind them (eight in flight), then the SUM
// into the row's accumulators with one atomic row; the counter is reset.  An overflow chain (more than GQE_HOT_SUB_CAP entries
// in one step) is walked node by node.
__global__ void __launch_bounds__(256) gqe_hot_gather_kernel(const GqeHot hot, const int32_t* __restrict__ next, const float* __restrict__ contrib,
                                                             const int32_t* __restrict__ link_contrib, int max_entries, int d) {
  const int lane = threadIdx.x & 63;
  const int waves = (int)gridDim.x * 4;
  int32_t* cnt = GQE_HOT_SUB_CNT(hot.sub);
  int32_t* ovf = GQE_HOT_SUB_OVF(hot.sub);
  // (everything a sub-list's wave needs before its rows is requested at once — the pool's size bounds h, the number of sub-lists
  // handed out is only needed for the answer: one round trip in front of the rows instead of three)
  const int n_sub = min(__hip_atomic_load(hot.sub_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), GQE_HOT_SUB_POOL);
  for (int h = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6)); h < GQE_HOT_SUB_POOL; h += waves) {
    const int32_t* buf = GQE_HOT_SUB_BUF(hot.sub) + (size_t)h * GQE_HOT_SUB_CAP;
    const int b0 = buf[lane], b1 = buf[lane + 64];   // (GQE_HOT_SUB_CAP == 128; stale beyond the counter)
    const int taken = __builtin_amdgcn_readfirstlane(cnt[h]);
    const int x0 = __builtin_amdgcn_readfirstlane(ovf[h]);
    const int slot = __builtin_amdgcn_readfirstlane(GQE_HOT_SUB_SLOT(hot.sub)[h]);
    if (h >= n_sub) break;
    if (taken == 0 && x0 < 0) continue;
    const int n = min(taken, GQE_HOT_SUB_CAP);
    if (lane == 0) {
      cnt[h] = 0;
      if (x0 >= 0) ovf[h] = -1;
    }
    const int mine0 = lane < n ? b0 : 0, mine1 = lane + 64 < n ? b1 : 0;
    float* acc = hot.acc + GQE_HOT_ROW(h & (GQE_HOT_SUB_REPS - 1), slot) * d;
    // columns lane, lane + 64, ...: every load and every atomic instruction of the wave covers 256 contiguous bytes (16-byte
    // lane slices made the atomic rows four times as many 64-byte operations: 42 us for this kernel instead of 18)
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};   // (d <= 256)
#define GQE_SUB_ENTRY(j) ((j) < 64 ? __builtin_amdgcn_readlane(mine0, (j)) : __builtin_amdgcn_readlane(mine1, (j) - 64))
    int j = 0;
    for (; j + 8 <= n; j += 8) {   // eight rows in flight (a group never straddles entry 64)
      const float* r[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) r[k] = contrib + (size_t)GQE_SUB_ENTRY(j + k) * d + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (lane + 64 * c < d) {
          float a[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) a[k] = r[k][64 * c];
          s0[c] += (a[0] + a[2]) + (a[4] + a[6]);
          s1[c] += (a[1] + a[3]) + (a[5] + a[7]);
        }
    }
    for (; j < n; ++j) {
      const float* r0 = contrib + (size_t)GQE_SUB_ENTRY(j) * d + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (lane + 64 * c < d) s0[c] += r0[64 * c];
    }
#undef GQE_SUB_ENTRY
    for (int y = x0; y >= 0; y = __builtin_amdgcn_readfirstlane(next[y])) {   // the overflow chain (rare)
      const float* r0 = contrib + (size_t)__builtin_amdgcn_readfirstlane(link_contrib[y - max_entries]) * d + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (lane + 64 * c < d) s1[c] += r0[64 * c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (lane + 64 * c < d) unsafeAtomicAdd(acc + lane + 64 * c, s0[c] + s1[c]);
  }
}

hipError_t gqe_launch_hot_gather(const GqeHot& hot, const int32_t* next, const float* contrib, const int32_t* link_contrib, int max_entries, int d,
                                 hipStream_t stream) {
  hipLaunchKernelGGL(gqe_hot_gather_kernel, dim3(2048), dim3(256), 0, stream, hot, next, contrib, link_contrib, max_entries, d);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Lazy rows (gqe_set_lazy_adam).  A row without a gradient still moves under Adam (its momentum decays), which
// is why the eager pass streams every row of every stepped table each iteration.  But those zero-gradient steps
// depend on nothing except the row's own (p, m, v) and the step number: they can be REPLAYED later, in registers,
// with exactly the arithmetic the eager pass would have executed (adam1), the first time the row is needed
// again — by a forward that reads it, or by the step that gives it a gradient.  last[row] = the table's Adam
// step count the row is current for; the per-step bias-correction pair of the last 64 steps of each table sits
// in a device ring (older steps cannot be pending: the host runs a full pass before a ring slot is reused).
// Result: bit-identical parameters, HBM traffic proportional to the rows a step touches.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void lazy_advance(float4& pp, float4& mm, float4& vv, const float4& gg, int from, int target,
                                             int grad_step, const float2* __restrict__ ring, float step_size_now,
                                             float bc2_now, int now_step, float b1, float b2, float eps) {
  const float b1c = 1.f - b1, b2c = 1.f - b2;
  for (int j = from + 1; j <= target; ++j) {
    float ss = step_size_now, bc = bc2_now;
    if (j != now_step) {
      const float2 c = ring[j & (GQE_LAZY_RING - 1)];
      ss = c.x;
      bc = c.y;
    }
    const bool gs = j == grad_step;
    const float ibc = __builtin_amdgcn_rcpf(bc);
    gqe_adam1(pp.x, mm.x, vv.x, gs ? gg.x : 0.f, ss, ibc, b1c, b2, b2c, eps);
    gqe_adam1(pp.y, mm.y, vv.y, gs ? gg.y : 0.f, ss, ibc, b1c, b2, b2c, eps);
    gqe_adam1(pp.z, mm.z, vv.z, gs ? gg.z : 0.f, ss, ibc, b1c, b2, b2c, eps);
    gqe_adam1(pp.w, mm.w, vv.w, gs ? gg.w : 0.f, ss, ibc, b1c, b2, b2c, eps);
  }
}

// The operand-ordered copies of a d x d parameter (gqe_dev.h, GQE_TILE_INDEX): M[i][k .. k+3] goes to one float4 of the copy of M
// and to column i of rows k .. k+3 of the copy of M^T.
__device__ __forceinline__ void tile_store(float* __restrict__ A, float* __restrict__ T, int d, long long e0, const float4& pp) {
  const int i = (int)(e0 / d), k = (int)(e0 - (long long)i * d);
  *reinterpret_cast<float4*>(A + GQE_TILE_INDEX(i, k, d)) = pp;
  float* t = T + GQE_TILE_INDEX(k, i, d);   // (k % 4 == 0: rows k .. k+3 of M^T are lanes l .. l+3 of one tile, 4 floats apart)
  t[0] = pp.x;
  t[4] = pp.y;
  t[8] = pp.z;
  t[12] = pp.w;
}

// rebuilds the copies of the listed matrices from the parameter arena (gqe_params_changed, a formula naming a new matrix)
__global__ __launch_bounds__(GQE_THREADS) void gqe_retile_kernel(const GqeRetileArgs a, const float* __restrict__ params, float* __restrict__ ws, int d) {
  const int per = (d * d + GQE_OPT_CHUNK - 1) / GQE_OPT_CHUNK;   // chunks of 1024 floats per matrix
  const int mi = blockIdx.x / per;
  const long long e0 = (long long)(blockIdx.x - mi * per) * GQE_OPT_CHUNK + (long long)threadIdx.x * 4;
  if (mi >= a.n || e0 >= (long long)d * d) return;
  const float4 pp = *reinterpret_cast<const float4*>(params + a.param[mi] + e0);
  tile_store(ws + a.tile[mi], ws + a.tile[mi] + a.tile_t, d, e0, pp);
}

__global__ __launch_bounds__(GQE_THREADS) void gqe_tilecheck_kernel(const GqeRetileArgs a, const float* __restrict__ params, const float* __restrict__ ws,
                                                                   int d, int32_t* __restrict__ mismatches) {
  const int per = (d * d + GQE_OPT_CHUNK - 1) / GQE_OPT_CHUNK;
  const int mi = blockIdx.x / per;
  const long long e0 = (long long)(blockIdx.x - mi * per) * GQE_OPT_CHUNK + (long long)threadIdx.x * 4;
  if (mi >= a.n || e0 >= (long long)d * d) return;
  const int i = (int)(e0 / d), k = (int)(e0 - (long long)i * d);
  const float4 pp = *reinterpret_cast<const float4*>(params + a.param[mi] + e0);
  const float4 tt = *reinterpret_cast<const float4*>(ws + a.tile[mi] + GQE_TILE_INDEX(i, k, d));
  const float* t = ws + a.tile[mi] + a.tile_t + GQE_TILE_INDEX(k, i, d);
  const bool same = __float_as_int(pp.x) == __float_as_int(tt.x) && __float_as_int(pp.y) == __float_as_int(tt.y) &&
                    __float_as_int(pp.z) == __float_as_int(tt.z) && __float_as_int(pp.w) == __float_as_int(tt.w) &&
                    __float_as_int(pp.x) == __float_as_int(t[0]) && __float_as_int(pp.y) == __float_as_int(t[4]) &&
                    __float_as_int(pp.z) == __float_as_int(t[8]) && __float_as_int(pp.w) == __float_as_int(t[12]);
  if (!same) atomicAdd(mismatches, 1);
}

hipError_t gqe_launch_tilecheck(const GqeRetileArgs& a, const float* params, const float* ws, int d, int32_t* mismatches, hipStream_t stream) {
  if (a.n < 1) return hipSuccess;
  const int per = (d * d + GQE_OPT_CHUNK - 1) / GQE_OPT_CHUNK;
  hipLaunchKernelGGL(gqe_tilecheck_kernel, dim3((unsigned)(a.n * per)), dim3(GQE_THREADS), 0, stream, a, params, ws, d, mismatches);
  return hipGetLastError();
}

hipError_t gqe_launch_retile(const GqeRetileArgs& a, const float* params, float* ws, int d, hipStream_t stream) {
  if (a.n < 1) return hipSuccess;
  const int per = (d * d + GQE_OPT_CHUNK - 1) / GQE_OPT_CHUNK;
  hipLaunchKernelGGL(gqe_retile_kernel, dim3((unsigned)(a.n * per)), dim3(GQE_THREADS), 0, stream, a, params, ws, d);
  return hipGetLastError();
}

template <int MODE, bool LISTS, bool DENSE_T, bool SORTED, bool LAZY, bool NT = false>
__device__ __forceinline__ void opt_body(const long long first_chunk, const long long chunk_stride,
                                         const GqeDevSeg* __restrict__ segs, int n_segs,
                                                             long long total_chunks, float* __restrict__ p,
                                                             float* __restrict__ g, float* __restrict__ m,
                                                             float* __restrict__ v, int32_t* __restrict__ head,
                                                             const int32_t* __restrict__ next,
                                                             const float* __restrict__ contrib,
                                                             const int32_t* __restrict__ link_contrib, int max_entries,
                                                             int d, float lr, float b1, float b2, float eps,
                                                             const GqeStepCoef& coef, const GqeOptActive& active,
                                                             const GqeActSeg* __restrict__ act, int n_act, const GqeLazyArgs& lazy,
                                                             const GqeHot& hot) {
  // Which tensor owns chunk ch, and its Adam coefficients.  Kernel-argument form (act == NULL): a prefix over the
  // <= GQE_MAX_SEGS universe entries in LDS.  Table form: a binary search in the uploaded list of active tensors.
  __shared__ int32_t s_begin[GQE_MAX_SEGS + 1];  // chunk prefix over the universe (from the host); inactive tensors own 0 chunks
  if (!act) {
    if ((int)threadIdx.x <= n_segs) s_begin[threadIdx.x] = active.begin[threadIdx.x];
    __syncthreads();
  }
  // d / 4 threads per table row, and a row NEVER straddles two waves: a wave holds floor(64 / tpr) rows (its last lanes idle
  // when d / 4 does not divide 64: d = 48, 80, 96, ...).  The row's first thread resets the list head (and, in lazy mode, used
  // to set the row's step count) after every thread of the row has read it — guaranteed only inside one wave.
  const int tpr = d >> 2;
  const int rpw = 64 / tpr;                       // rows per wave
  const int rpc = rpw * GQE_WAVES;                // table rows per chunk (gqe_host.cpp, universe_index: the same formula)
  const int wl = threadIdx.x & 63, wrow = wl / tpr;
  const int lr_row = wrow < rpw ? (int)(threadIdx.x >> 6) * rpw + wrow : rpc;   // rpc = an idle lane
  const int c4 = (wl - wrow * tpr) * 4;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  int si = 0;
  for (long long ch = first_chunk; ch < total_chunks; ch += chunk_stride) {
    long long chunk_begin;
    float step_size, bc2_sqrt;
    bool seg_dense;  // tables: this table's dense gradient is live (materialised / written by the caller) and has to be read too
    if (act) {
      int lo = 0, hi = n_act - 1;  // last entry whose first chunk is <= ch
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (act[mid].chunk_begin <= ch) lo = mid; else hi = mid - 1;
      }
      const GqeActSeg a = act[lo];
      si = a.seg;
      chunk_begin = a.chunk_begin;
      step_size = a.step_size;
      bc2_sqrt = a.bc2_sqrt;
      seg_dense = a.pad != 0;
    } else {
      while (s_begin[si + 1] <= ch) ++si;  // chunks are visited in increasing order
      chunk_begin = s_begin[si];
      const int grp = active.group[si];
      seg_dense = (grp & GQE_GROUP_DENSE) != 0;
      step_size = coef.step_size[grp & (GQE_GROUP_DENSE - 1)];
      bc2_sqrt = coef.bc2_sqrt[grp & (GQE_GROUP_DENSE - 1)];
    }
    const GqeDevSeg sg = segs[si];
    if (sg.is_table) {
      const long long row = (ch - chunk_begin) * rpc + lr_row;
      if (lr_row >= rpc || row >= sg.rows) continue;
      const long long off = sg.offset + row * d + c4;
      float4 gg = zero4;
      const bool dense_here = DENSE_T && seg_dense;
      if (dense_here) {
        gg = *reinterpret_cast<const float4*>(g + off);
        if (MODE != GQE_OPT_MATERIALIZE) *reinterpret_cast<float4*>(g + off) = zero4;
      }
      bool had = false;
      // (requested next to the list head: no extra round trip; never in the order-independent instantiations, which no hot row
      // can reach: gqe_set_ordered_sums clears the slots, the exchange mode never promotes)
      const int hs = (!SORTED && hot.slot) ? hot.slot[sg.head_base + row] : -1;
      if (LISTS) {
        const int h0 = head[sg.head_base + row];
        had = h0 >= 0;
        int len;
        const float4 acc = list_gradient<SORTED>(h0, next, contrib, link_contrib, max_entries, d, c4, len);
        if (had && c4 == 0) {
          head[sg.head_base + row] = -1;
          if (!SORTED && hot.slot && hs < 0) hot_promote(hot, sg.head_base + row, len);
        }
        gg.x += acc.x;
        gg.y += acc.y;
        gg.z += acc.z;
        gg.w += acc.w;
      }
      if (hs >= 0) {   // a hot row: what it collected sits in its accumulators (usually next to an empty list)
        const float4 acc = hot_take(hot, hs, d, c4);
        had = true;
        gg.x += acc.x;
        gg.y += acc.y;
        gg.z += acc.z;
        gg.w += acc.w;
      }
      if (MODE == GQE_OPT_ZERO) continue;
      if (MODE == GQE_OPT_MATERIALIZE) {
        if (had) {
          if (!dense_here) {
            const float4 old = *reinterpret_cast<const float4*>(g + off);
            gg.x += old.x;
            gg.y += old.y;
            gg.z += old.z;
            gg.w += old.w;
          }
          *reinterpret_cast<float4*>(g + off) = gg;
        }
        continue;
      }
      // SGD without momentum (bio/train.py:60): a row without a gradient does not move — p - lr * 0 is p, bit for bit — so it is
      // neither read nor written (the pass streamed 100 MB of untouched rows: 26.6 us of a 61 us step)
      if (MODE == GQE_OPT_SGD && !had && !dense_here) continue;
      float4 pp = ld_stream<NT>(p + off);
      float4 mm = zero4, vv = zero4;
      if (MODE == GQE_OPT_ADAM) {
        mm = ld_stream<NT>(m + off);
        vv = ld_stream<NT>(v + off);
      }
      if (LAZY && MODE == GQE_OPT_ADAM) {
        // full pass in lazy mode: replay what the row is behind, then (grad_step == target) the step with gradient
        const int lt = sg.table_index;
        const int target = lazy.t.target[lt];
        const int from = lazy.t.eager[lt] ? target - 1 : lazy.last[sg.head_base + row];
        if (from < target) {
          lazy_advance(pp, mm, vv, gg, from, target, lazy.t.grad_step[lt], lazy.ring + lt * GQE_LAZY_RING, step_size, bc2_sqrt,
                       lazy.t.grad_step[lt], b1, b2, eps);
          // (last[row] := target is the HOST's memset behind this launch: the d / 4 threads of a row straddle two waves
          // whenever d / 4 does not divide 64 — d = 48, 80, 96, ... — and a store from the row's first thread here was read
          // as `from` by the threads of the other wave, which then skipped their elements)
          *reinterpret_cast<float4*>(m + off) = mm;
          *reinterpret_cast<float4*>(v + off) = vv;
          *reinterpret_cast<float4*>(p + off) = pp;
        }
        if (ch == chunk_begin && threadIdx.x == 0 && lazy.t.grad_step[lt] > 0)
          lazy.ring[lt * GQE_LAZY_RING + (lazy.t.grad_step[lt] & (GQE_LAZY_RING - 1))] = make_float2(step_size, bc2_sqrt);
        continue;
      }
      opt_update<MODE>(pp, mm, vv, gg, step_size, bc2_sqrt, lr, b1, b2, eps);
      if (MODE == GQE_OPT_ADAM) {
        st_stream<NT>(m + off, mm);
        st_stream<NT>(v + off, vv);
      }
      st_stream<NT>(p + off, pp);
      continue;
    }
    if (MODE == GQE_OPT_MATERIALIZE) continue;
    // ---- dense segment ----
    const long long e0 = (ch - chunk_begin) * GQE_OPT_CHUNK + (long long)threadIdx.x * 4;
    if (e0 >= sg.numel) continue;
    const long long off = sg.offset + e0;
    if (e0 + 4 <= sg.numel) {
      const float4 gg = *reinterpret_cast<const float4*>(g + off);
      *reinterpret_cast<float4*>(g + off) = zero4;
      if (MODE == GQE_OPT_ZERO) continue;
      float4 pp = *reinterpret_cast<const float4*>(p + off);
      float4 mm = zero4, vv = zero4;
      if (MODE == GQE_OPT_ADAM) {
        mm = *reinterpret_cast<const float4*>(m + off);
        vv = *reinterpret_cast<const float4*>(v + off);
      }
      opt_update<MODE>(pp, mm, vv, gg, step_size, bc2_sqrt, lr, b1, b2, eps);
      if (MODE == GQE_OPT_ADAM) {
        *reinterpret_cast<float4*>(m + off) = mm;
        *reinterpret_cast<float4*>(v + off) = vv;
      }
      *reinterpret_cast<float4*>(p + off) = pp;
      if (sg.tile) tile_store(sg.tile, sg.tile_T, d, e0, pp);   // a d x d matrix: its operand-ordered copies follow it
    } else {
      for (long long k = e0; k < sg.numel; ++k) {
        const long long o = sg.offset + k;
        const float gg = g[o];
        g[o] = 0.f;
        if (MODE == GQE_OPT_ZERO) continue;
        if (MODE == GQE_OPT_ADAM) {
          float pp = p[o], mm = m[o], vv = v[o];
          gqe_adam1(pp, mm, vv, gg, step_size, __builtin_amdgcn_rcpf(bc2_sqrt), 1.f - b1, b2, 1.f - b2, eps);
          m[o] = mm;
          v[o] = vv;
          p[o] = pp;
        } else {
          p[o] -= lr * gg;
        }
      }
    }
  }
}

template <int MODE, bool LISTS, bool DENSE_T, bool SORTED, bool LAZY, bool NT>
__global__ __launch_bounds__(GQE_THREADS) void gqe_opt_kernel(const GqeDevSeg* __restrict__ segs, int n_segs,
                                                             long long total_chunks, float* __restrict__ p,
                                                             float* __restrict__ g, float* __restrict__ m,
                                                             float* __restrict__ v, int32_t* __restrict__ head,
                                                             const int32_t* __restrict__ next,
                                                             const float* __restrict__ contrib,
                                                             const int32_t* __restrict__ link_contrib, int max_entries,
                                                             int d, float lr, float b1, float b2, float eps,
                                                             GqeStepCoef coef, GqeOptActive active,
                                                             const GqeActSeg* __restrict__ act, int n_act, GqeLazyArgs lazy, GqeHot hot) {
  opt_body<MODE, LISTS, DENSE_T, SORTED, LAZY, NT>(blockIdx.x, gridDim.x, segs, n_segs, total_chunks, p, g, m, v, head, next,
                                                    contrib, link_contrib, max_entries, d, lr, b1, b2, eps, coef, active, act, n_act, lazy, hot);
}

// ------------------------------------------------------------------------------------------
// launchers (called from gqe_host.cpp)
// ------------------------------------------------------------------------------------------
#define GQE_DECL(DEC, MLP) \
  hipError_t gqe_launch_fused_##DEC##_##MLP##_w16(const GqeFusedArgs& a); \
  hipError_t gqe_launch_fused_##DEC##_##MLP##_w8(const GqeFusedArgs& a);
GQE_DECL(0, 0) GQE_DECL(0, 1) GQE_DECL(1, 0) GQE_DECL(1, 1) GQE_DECL(2, 0) GQE_DECL(2, 1)
#undef GQE_DECL

// Which workgroup shape runs a launch (gqe_fused.h): 16 waves (one query row per wave) for every accepted d — straight-line FULL
// code at d = 64 / 128 / 256, the guarded form (buffer-addressed rows and matrices, padded LDS tiles) elsewhere; 8 waves (two rows
// per wave) only at d = 128 when the launch has so many tiles that two or three co-resident workgroups per CU beat the shorter
// per-tile chain of the 16-wave shape.
int gqe_fused_waves(int dec, int d, int tiles) {
  static const int min_tiles = [] {   // GQE_DEBUG_FW8_MIN_TILES: tuning runs only
    const char* e = getenv("GQE_DEBUG_FW8_MIN_TILES");
    return e ? atoi(e) : GQE_FW8_MIN_TILES;
  }();
  (void)dec;
  if (d == 128 && tiles > min_tiles) return 8;
  return 16;
}

// Which (decoder, intersection, dim) the library accepts: every multiple of 16 up to GQE_MAX_DIM, with every decoder pair —
// what the reference's `--embed_dim` allows in practice (bio/train.py:13).  Every kernel the dispatcher can select is free of
// register spills and of the allocator's "spill / copy ahead of the EXEC restore" placement;
// tests/test_build_meta.py checks that against the code objects and the assembly of the built library.  (Round 3 refused full
// Bilinear at d not in {16 .. 64, 128, 256} and the MLP intersections at d in (192, 256): their guarded kernels spilled; the
// guarded kernels now address rows and matrices through buffer descriptors — no lane-divergent bounds guard — and run their
// contractions over the padded extent with compile-time trip counts.)
int gqe_config_supported(int dec, int inter, int d) {
  (void)inter;
  if (dec < 0 || dec > DEC_BILINEAR) return 0;
  if (d < 16 || d > GQE_MAX_DIM || d % 16) return 0;
  return 1;
}

// the template arguments (NC, FULL, FW) of the kernel gqe_launch_fused runs for (decoder, d) in a launch of `tiles` tiles
// (mirrors launch_fused_dm in gqe_fused.h; tests/test_build_meta.py looks the variant up in the built library)
void gqe_fused_variant(int dec, int d, int tiles, int* nc, int* full, int* fw) {
  *nc = (d + 63) / 64;
  *full = ((d % 64) == 0 && d != 192) ? 1 : 0;   // (d = 192 runs the guarded NC = 3 kernel)
  *fw = gqe_fused_waves(dec, d, tiles);
}

// rider workgroups (gqe_train_step) are compiled into the backward kernels of the straight-line dims
int gqe_fused_can_ride(int dec, int mlp, int d, int tiles) {
  (void)dec;
  (void)mlp;
  (void)tiles;
  return (d % 64) == 0 && d != 192 && (64 % (d >> 2)) == 0;
}

hipError_t gqe_launch_fused(int dec, int mlp, const GqeFusedArgs& a) {
  const int key = dec * 2 + (mlp ? 1 : 0);
  const int fw = (a.force_fw == 8 && a.d == 128) ? 8 : (a.force_fw == 16 ? 16 : gqe_fused_waves(dec, a.d, a.plan.tiles));
  if (fw == 8) {
    switch (key) {
      case 0: return gqe_launch_fused_0_0_w8(a);
      case 1: return gqe_launch_fused_0_1_w8(a);
      case 2: return gqe_launch_fused_1_0_w8(a);
      case 3: return gqe_launch_fused_1_1_w8(a);
      case 4: return gqe_launch_fused_2_0_w8(a);
      default: return gqe_launch_fused_2_1_w8(a);
    }
  }
  switch (key) {
    case 0: return gqe_launch_fused_0_0_w16(a);
    case 1: return gqe_launch_fused_0_1_w16(a);
    case 2: return gqe_launch_fused_1_0_w16(a);
    case 3: return gqe_launch_fused_1_1_w16(a);
    case 4: return gqe_launch_fused_2_0_w16(a);
    default: return gqe_launch_fused_2_1_w16(a);
  }
}

hipError_t gqe_launch_pair_gemm(const GqeFusedArgs& a, float* losses) {
  const int blocks = a.plan.units + 1;  // + the finalize block
  hipLaunchKernelGGL(gqe_pair_gemm_kernel, dim3(blocks), dim3(GQE_THREADS), 0, a.stream, a.plan, a.formulas, a.ws, a.grads, a.d,
                     a.tile_loss, losses, a.prof);
  return hipGetLastError();
}

// ---- the pair GEMM riding in an Adam pass's launch (GqeGemmRide, gqe_dev.h; gqe_set_deferred_gemm) ----------------------
// The table chunks of the pass do not depend on the matrix gradients, and the pass is HBM-bound while a GEMM unit is a short
// latency chain: side by side in one launch the units cost the pass nothing measurable and the step loses the GEMM's own launch
// (12.6 us + a kernel boundary at the headline shape).  A unit here lives inside the optimiser kernel's register budget
// (<= 64 VGPRs: eight streaming waves per SIMD) and takes no LDS (a static allocation would be charged to every chunk's
// workgroup; as gqe_pair_gemm_kernel's unit — 146 VGPRs, 40 KB — the whole launch ran at three workgroups per CU and the pass
// alone lost 4.5 us).  It may be slow: the table chunks stream for ~45 us.  Wave w owns the 64 x 16 block (rows i0 .. i0+63,
// columns j0 + 16 w ..) of the unit's 64 x 64 block; per step of four queries a lane reads ONE float4 of L — four consecutive
// gradient rows: the A operands of four MFMAs whose output rows interleave (row = i0 + 4 * (lane & 15) + t) — and one float of R.
#define GQE_RIDE_DEPTH 4   // steps of operands in flight per lane (8 measured the same)
// Which unit a workgroup of a units' launch takes.  The 64 x 64 blocks of one (job, query chunk) are consecutive units and read the
// same operand rows (at d = 128: four blocks, each half of L and of R read twice); consecutive workgroups go to different XCDs, each
// with an L2 of its own (PMC, round 6: the units of the headline step fetched 20.7 MB for 10.5 MB of operand rows).  Workgroups x,
// x + 8, x + 16, x + 24 share an XCD: those get four consecutive units.  (x = workgroup index among the units; a bijection on the
// first units / 32 * 32 of them, the identity on the rest.)
__device__ __forceinline__ int unit_of_workgroup(int x, int units) {
  const int full = (units >> 5) << 5;
  return x < full ? (((((x >> 3) >> 2) << 3) | (x & 7)) << 2) | ((x >> 3) & 3) : x;
}

__device__ __forceinline__ void gemm_ride_unit(const GqeDynPlan& plan, const GqeDevFormula* __restrict__ formulas, const float* __restrict__ ws,
                                               float* __restrict__ grads, int d, int unit) {
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int il = lane & 15, lk = lane >> 4;
  int bi = 0;
#pragma unroll
  for (int k = 1; k < GQE_LAUNCH_BATCHES; ++k) bi += (unit >= plan.unit_begin[k]) ? 1 : 0;
  const GqeDynBatch b = plan.b[bi];
  const GqeDevFormula* __restrict__ f = formulas + b.formula;
  constexpr int MT = GQE_GEMM_MT;
  const int mper = d / MT, macros = mper * mper;   // (d % 64 == 0: the host only lets such launches ride)
  // queries per unit: pad[0] chunks of GQE_GEMM_KCHUNK — or pad[2] queries (a split step's units are the critical chain of their
  // launch: shorter units, more of them)
  const int cq = plan.pad[2] ? plan.pad[2] : GQE_GEMM_KCHUNK * plan.pad[0];
  const int chunks = (b.Bpad + cq - 1) / cq;
  int u = unit - b.unit_begin;
  const int job = u / (chunks * macros);
  u -= job * chunks * macros;
  const int chunk = u / macros;
  const int mt = u - chunk * macros;
  const int i0 = (mt / mper) * MT, j0 = (mt % mper) * MT;
  const int k_first = chunk * cq;
  const int k_end = min(k_first + cq, b.Bpad);   // Bpad % 16 == 0: whole steps of four queries
  const size_t slot_floats = (size_t)b.Bpad * d;
  const float* L = ws + b.scratch_base + (size_t)f->job_L[job] * slot_floats + (size_t)lk * d + i0 + 4 * il;
  const float* R = ws + b.scratch_base + (size_t)f->job_R[job] * slot_floats + (size_t)lk * d + j0 + 16 * wave + il;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int steps = (k_end - k_first) >> 2;
  float4 a[GQE_RIDE_DEPTH];
  float bv[GQE_RIDE_DEPTH];
#pragma unroll
  for (int s = 0; s < GQE_RIDE_DEPTH; ++s) {
    const int k = k_first + 4 * min(s, steps - 1);
    a[s] = *reinterpret_cast<const float4*>(L + (size_t)k * d);
    bv[s] = R[(size_t)k * d];
  }
  for (int s0 = 0; s0 < steps; s0 += GQE_RIDE_DEPTH) {
#pragma unroll
    for (int j = 0; j < GQE_RIDE_DEPTH; ++j) {
      const int s = s0 + j;
      if (s < steps) {   // wave-uniform
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, bv[j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, bv[j], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, bv[j], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, bv[j], acc[3], 0, 0, 0);
        const int k = k_first + 4 * min(s + GQE_RIDE_DEPTH, steps - 1);   // (the last steps re-read the final rows: no branch)
        a[j] = *reinterpret_cast<const float4*>(L + (size_t)k * d);
        bv[j] = R[(size_t)k * d];
      }
    }
  }
  // D[4 * lk + r][il] of MFMA t is gradient row i0 + 4 * (4 * lk + r) + t, column j0 + 16 * wave + il
  float* out = grads + f->job_param[job] + (size_t)(i0 + 16 * lk) * d + j0 + 16 * wave + il;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) unsafeAtomicAdd(out + (size_t)(4 * r + t) * d, acc[t][r]);
}

template <bool NT>
__global__ __launch_bounds__(GQE_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void gqe_opt_gemm_kernel(
    const GqeDevSeg* __restrict__ segs, int n_segs, long long total_chunks, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
    float* __restrict__ v, int32_t* __restrict__ head, const int32_t* __restrict__ next, const float* __restrict__ contrib,
    const int32_t* __restrict__ link_contrib, int max_entries, int d, float lr, float b1, float b2, float eps, GqeStepCoef coef,
    GqeOptActive active, const GqeActSeg* __restrict__ act, int n_act, GqeHot hot, GqeGemmRide ride) {
  if (blockIdx.x == 0) {
    finalize_losses(ride.plan, ride.tile_loss, ride.losses);
    return;
  }
  const int front = ride.plan.units + 1;
  long long first = (long long)blockIdx.x - front;
  if (ride.spread > 0) {
    // workgroups 1, 1 + K, 1 + 2 K, ... are units (K = 1 mod 8: unit j on the XCD unit_of_workgroup expects); the others stream
    const int x = (int)blockIdx.x - 1, j = x / ride.spread;
    if (j < ride.plan.units && j * ride.spread == x) {
      gemm_ride_unit(ride.plan, ride.formulas, ride.ws, g, d, unit_of_workgroup(j, ride.plan.units));
      return;
    }
    first = x - min(ride.plan.units, j + 1);
  } else if ((int)blockIdx.x < front) {
    gemm_ride_unit(ride.plan, ride.formulas, ride.ws, g, d, unit_of_workgroup((int)blockIdx.x - 1, ride.plan.units));
    return;
  }
  GqeLazyArgs lazy;   // (never read: LAZY = false)
  opt_body<GQE_OPT_ADAM, true, false, false, false, NT>(first, (long long)gridDim.x - front, segs, n_segs, total_chunks, p, g, m,
                                                       v, head, next, contrib, link_contrib, max_entries, d, lr, b1, b2, eps, coef, active, act,
                                                       n_act, lazy, hot);
}

__device__ __forceinline__ void matstep_body(const GqeMatStep& a, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                             float* __restrict__ v, int d, float b1, float b2, float eps, int block) {
  const int per = (d * d) / GQE_OPT_CHUNK;   // (d % 64 == 0: whole chunks)
  const int mi = block / per;
  const long long e0 = (long long)(block - mi * per) * GQE_OPT_CHUNK + (long long)threadIdx.x * 4;
  const long long off = a.off[mi] + e0;
  const float4 gg = *reinterpret_cast<const float4*>(g + off);
  float4 pp = *reinterpret_cast<const float4*>(p + off);
  float4 mm = *reinterpret_cast<const float4*>(m + off);
  float4 vv = *reinterpret_cast<const float4*>(v + off);
  *reinterpret_cast<float4*>(g + off) = make_float4(0.f, 0.f, 0.f, 0.f);
  opt_update<GQE_OPT_ADAM>(pp, mm, vv, gg, a.step_size[mi], a.bc2_sqrt[mi], 0.f, b1, b2, eps);
  *reinterpret_cast<float4*>(m + off) = mm;
  *reinterpret_cast<float4*>(v + off) = vv;
  *reinterpret_cast<float4*>(p + off) = pp;
  tile_store(a.tile[mi], a.tile[mi] + a.tile_t, d, e0, pp);
}

__global__ __launch_bounds__(GQE_THREADS) void gqe_matstep_kernel(const GqeMatStep a, float* __restrict__ p, float* __restrict__ g,
                                                                 float* __restrict__ m, float* __restrict__ v, int d, float b1, float b2, float eps) {
  matstep_body(a, p, g, m, v, d, b1, b2, eps, (int)blockIdx.x);
}

hipError_t gqe_launch_matstep(const GqeMatStep& a, float* p, float* g, float* m, float* v, int d, float b1, float b2, float eps, hipStream_t stream) {
  if (a.n < 1) return hipSuccess;
  hipLaunchKernelGGL(gqe_matstep_kernel, dim3((unsigned)(a.n * ((d * d) / GQE_OPT_CHUNK))), dim3(GQE_THREADS), 0, stream, a, p, g, m, v, d, b1, b2, eps);
  return hipGetLastError();
}

hipError_t gqe_launch_opt_gemm(const GqeOptArgs& a, const GqeGemmRide& r) {
  long long blocks = a.total_chunks < 262144 ? a.total_chunks : 262144;
  if (blocks < 1) blocks = 1;
  blocks += r.plan.units + 1;
#define GQE_RIDE(N)                                                                                                                     \
  hipLaunchKernelGGL((gqe_opt_gemm_kernel<N>), dim3((unsigned)blocks), dim3(GQE_THREADS), 0, a.stream, a.segs, a.n_segs, a.total_chunks, a.p, \
                     a.g, a.m, a.v, a.head, a.next, a.contrib, a.link_contrib, a.max_entries, a.d, a.lr, a.b1, a.b2, a.eps, a.coef, a.active,  \
                     a.act, a.n_act, a.hot, r)
  if (a.nt) GQE_RIDE(true); else GQE_RIDE(false);
#undef GQE_RIDE
  return hipGetLastError();
}

template <int MODE>
static void launch_opt_mode(const GqeOptArgs& a, unsigned blocks) {
#define GON(L, D, S, Z, N)                                                                                                \
  hipLaunchKernelGGL((gqe_opt_kernel<MODE, L, D, S, Z, N>), dim3(blocks), dim3(GQE_THREADS), 0, a.stream, a.segs, a.n_segs, \
                     a.total_chunks, a.p, a.g, a.m, a.v, a.head, a.next, a.contrib, a.link_contrib, a.max_entries, a.d, a.lr, a.b1,  \
                     a.b2, a.eps, a.coef, a.active, a.act, a.n_act, a.lz, a.hot)
#define GO(L, D, S, Z) GON(L, D, S, Z, false)
  if (a.nt && MODE == GQE_OPT_ADAM && !a.lazy) {  // eager Adam over tables larger than the Infinity Cache: non-temporal p / m / v
    if (a.lists) {
      if (a.sorted) {
        if (a.dense_tables) GON(true, true, true, false, true); else GON(true, false, true, false, true);
      } else {
        if (a.dense_tables) GON(true, true, false, false, true); else GON(true, false, false, false, true);
      }
    } else {
      if (a.dense_tables) GON(false, true, false, false, true); else GON(false, false, false, false, true);
    }
  } else if (a.lazy && MODE == GQE_OPT_ADAM) {  // lazy full pass
    if (a.lists && a.sorted) {
      if (a.dense_tables) GO(true, true, true, true); else GO(true, false, true, true);
    } else if (a.lists) {
      if (a.dense_tables) GO(true, true, false, true); else GO(true, false, false, true);
    } else {
      if (a.dense_tables) GO(false, true, false, true); else GO(false, false, false, true);
    }
  } else if (a.lists) {
    if (a.sorted) {
      if (a.dense_tables) GO(true, true, true, false); else GO(true, false, true, false);
    } else {
      if (a.dense_tables) GO(true, true, false, false); else GO(true, false, false, false);
    }
  } else {
    if (a.dense_tables) GO(false, true, false, false); else GO(false, false, false, false);
  }
#undef GO
#undef GON
}

// ------------------------------------------------------------------------------------------
// lazy rows: advance exactly the table rows an index feed names (duplicates are resolved by an atomic max on the
// row's step count: the first arrival owns the work).  WITH_GRAD: the final step consumes the row's gradient
// list — the launch that replaces the full-table pass after a fused forward/backward; without: replay only —
// the launch that makes the rows current before a forward reads them.
// ------------------------------------------------------------------------------------------
template <bool WITH_GRAD, bool SORTED, typename SEGS>
__device__ __forceinline__ void rows_body(const int row_block, const int n_row_blocks, const int dense_block, const int n_dense_blocks,
                                          const SEGS& segs, const GqeLazyTabs& t,
                                                              const int32_t* __restrict__ idx, int32_t* __restrict__ last,
                                                              float2* __restrict__ ring, float* __restrict__ p,
                                                              float* __restrict__ g, float* __restrict__ m,
                                                              float* __restrict__ v, int32_t* __restrict__ head,
                                                              const int32_t* __restrict__ next,
                                                              const float* __restrict__ contrib, int max_entries, int d,
                                                              float lr, float b1, float b2, float eps,
                                                              const GqeDevSeg* __restrict__ dsegs, int n_dsegs,
                                                              long long dense_chunks, const GqeStepCoef& dcoef, const GqeOptActive& dactive,
                                                              const GqeActSeg* __restrict__ dact, int n_dact, const GqeHot& hot) {
  if (row_block >= n_row_blocks) {
    // the step's small dense tensors (relation vectors / matrices, Pre / Post): the ordinary chunk loop
    GqeLazyArgs none;
    none.last = nullptr;
    none.ring = nullptr;
    GqeHot no_hot;
    no_hot.slot = nullptr;
    opt_body<GQE_OPT_ADAM, false, false, false, false>((long long)dense_block, (long long)n_dense_blocks,
                                                       dsegs, n_dsegs, dense_chunks, p, g, m, v, head, next, contrib, nullptr,
                                                       max_entries, d, lr, b1, b2, eps, dcoef, dactive, dact, n_dact, none, no_hot);
    return;
  }
  // the coefficient rings go through LDS: a replay of k steps would otherwise chain k dependent global loads
  __shared__ float2 s_ring[GQE_LAZY_TABLES * GQE_LAZY_RING];
  for (int i = threadIdx.x; i < GQE_LAZY_TABLES * GQE_LAZY_RING; i += GQE_THREADS) s_ring[i] = ring[i];
  __syncthreads();
  if (WITH_GRAD && row_block == 0 && threadIdx.x < GQE_LAZY_TABLES && t.grad_step[threadIdx.x] > 0)
    ring[threadIdx.x * GQE_LAZY_RING + (t.grad_step[threadIdx.x] & (GQE_LAZY_RING - 1))] =
        make_float2(t.step_size[threadIdx.x], t.bc2_sqrt[threadIdx.x]);
  const int tpr = d >> 2;  // threads per row (a divisor of 64: the group never straddles a wave)
  const int e = (int)(((long long)row_block * GQE_THREADS + threadIdx.x) / tpr);
  if (e >= segs.total) return;
  const int c4 = (threadIdx.x % tpr) * 4;
  int k = 0;  // segment of entry e: begin[k] <= e < begin[k+1]; a scan with uniform (scalar) loads of the kernel
  for (int i = 1; i < segs.n; ++i) k += (e >= segs.begin[i]) ? 1 : 0;  // arguments, not a per-lane search through memory
  int lt = segs.tid[k];
  if (lt == -1) return;  // rows of a table that is not tracked per row (bag mode)
  int row = idx[segs.idx_begin[k] + (e - segs.begin[k])];
  if (row < 0) return;  // padding query / contribution that was never pushed
  if (lt == -2) {       // exchanged slabs name list heads (head_base + row): recover the table
    lt = 0;
#pragma unroll
    for (int i = 1; i < GQE_LAZY_TABLES; ++i) lt += (i < t.n && row >= t.head_base[i]) ? 1 : 0;
    row -= (int)t.head_base[lt];
  }
  const int target = t.target[lt];
  const long long hrow = t.head_base[lt] + row;
  const long long off = t.offset[lt] + (long long)row * d + c4;
  // everything the update needs is requested before the claim's round trip is awaited
  int from = 0;
  if (c4 == 0) from = __hip_atomic_fetch_max(last + hrow, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float4 pp = *reinterpret_cast<const float4*>(p + off);
  float4 mm = *reinterpret_cast<const float4*>(m + off);
  float4 vv = *reinterpret_cast<const float4*>(v + off);
  const int h0 = WITH_GRAD ? head[hrow] : -1;
  const int hs = (WITH_GRAD && !SORTED && hot.slot) ? hot.slot[hrow] : -1;
  from = __shfl(from, (threadIdx.x & 63) / tpr * tpr);
  if (from >= target) return;  // someone else brought (or is bringing) the row there
  float4 gg = make_float4(0.f, 0.f, 0.f, 0.f);
  if (WITH_GRAD && h0 >= 0) {
    int len;
    gg = list_gradient<SORTED>(h0, next, contrib, nullptr, max_entries, d, c4, len);
    if (c4 == 0) {
      head[hrow] = -1;
      if (!SORTED && hot.slot && hs < 0) hot_promote(hot, hrow, len);
    }
  }
  if (hs >= 0) {   // a hot row: its contributions of this step sit in its accumulators
    const float4 acc = hot_take(hot, hs, d, c4);
    gg.x += acc.x;
    gg.y += acc.y;
    gg.z += acc.z;
    gg.w += acc.w;
  }
  lazy_advance(pp, mm, vv, gg, from, target, t.grad_step[lt], s_ring + lt * GQE_LAZY_RING, t.step_size[lt], t.bc2_sqrt[lt],
               t.grad_step[lt], b1, b2, eps);
  *reinterpret_cast<float4*>(m + off) = mm;
  *reinterpret_cast<float4*>(v + off) = vv;
  *reinterpret_cast<float4*>(p + off) = pp;
}

template <bool WITH_GRAD, bool SORTED>
__global__ __launch_bounds__(GQE_THREADS) void gqe_rows_kernel(const GqeRowSegs segs, const GqeLazyTabs t,
                                                              const int32_t* __restrict__ idx, int32_t* __restrict__ last,
                                                              float2* __restrict__ ring, float* __restrict__ p,
                                                              float* __restrict__ g, float* __restrict__ m,
                                                              float* __restrict__ v, int32_t* __restrict__ head,
                                                              const int32_t* __restrict__ next,
                                                              const float* __restrict__ contrib, int max_entries, int d,
                                                              float lr, float b1, float b2, float eps, int n_row_blocks,
                                                              const GqeDevSeg* __restrict__ dsegs, int n_dsegs,
                                                              long long dense_chunks, GqeStepCoef dcoef, GqeOptActive dactive,
                                                              const GqeActSeg* __restrict__ dact, int n_dact, GqeHot hot) {
  rows_body<WITH_GRAD, SORTED>((int)blockIdx.x, n_row_blocks, (int)blockIdx.x - n_row_blocks, (int)gridDim.x - n_row_blocks, segs, t, idx, last, ring, p, g,
                               m, v, head, next, contrib, max_entries, d, lr, b1, b2, eps, dsegs, n_dsegs, dense_chunks, dcoef, dactive, dact, n_dact,
                               hot);
}

// The row launch that closes a lazy-Adam step carrying the step's deferred pair GEMM (gqe_set_deferred_gemm): the row groups do
// not read the matrix gradients, so the units and the loss finalize run in front of them in the same launch — as they do in
// front of the eager pass's chunks (gqe_opt_gemm_kernel) — and the d x d matrices are stepped by gqe_matstep_kernel behind it.
__global__ __launch_bounds__(GQE_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void gqe_rows_ride_kernel(
    const GqeRowSegs32 segs, const GqeLazyTabs t, const int32_t* __restrict__ idx, int32_t* __restrict__ last, float2* __restrict__ ring,
    float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int32_t* __restrict__ head,
    const int32_t* __restrict__ next, const float* __restrict__ contrib, int max_entries, int d, float lr, float b1, float b2, float eps,
    int n_row_blocks, const GqeDevSeg* __restrict__ dsegs, int n_dsegs, long long dense_chunks, GqeStepCoef dcoef, GqeOptActive dactive,
    const GqeActSeg* __restrict__ dact, int n_dact, GqeHot hot, GqeGemmRide ride) {
  const int front = ride.plan.units + 1;
  if ((int)blockIdx.x < front) {
    if (blockIdx.x == 0) finalize_losses(ride.plan, ride.tile_loss, ride.losses);
    else gemm_ride_unit(ride.plan, ride.formulas, ride.ws, g, d, unit_of_workgroup((int)blockIdx.x - 1, ride.plan.units));
    return;
  }
  const int rb = (int)blockIdx.x - front;
  rows_body<true, false>(rb, n_row_blocks, rb - n_row_blocks, (int)gridDim.x - front - n_row_blocks, segs, t, idx, last, ring, p, g, m, v, head, next,
                         contrib, max_entries, d, lr, b1, b2, eps, dsegs, n_dsegs, dense_chunks, dcoef, dactive, dact, n_dact, hot);
}

hipError_t gqe_launch_rows_ride(const GqeRowsArgs& a, const GqeGemmRide& r) {
  if (a.d < 4 || (64 % (a.d >> 2)) != 0 || !a.with_grad || a.sorted) return hipErrorInvalidValue;
  GqeRowSegs32 s32;
  memset(&s32, 0, sizeof s32);
  s32.n = a.segs.n;
  s32.total = a.segs.total;
  for (int k = 0; k <= a.segs.n; ++k) s32.begin[k] = a.segs.begin[k];
  for (int k = 0; k < a.segs.n; ++k) {
    if (a.segs.idx_begin[k] < -0x7fffffffll || a.segs.idx_begin[k] > 0x7fffffffll) return hipErrorInvalidValue;
    s32.idx_begin[k] = (int)a.segs.idx_begin[k];
    s32.tid[k] = a.segs.tid[k];
  }
  const long long threads = (long long)a.segs.total * (a.d >> 2);
  const unsigned row_blocks = (unsigned)((threads + GQE_THREADS - 1) / GQE_THREADS);
  const unsigned dense_blocks = (unsigned)(a.dense_chunks < 512 ? a.dense_chunks : 512);
  hipLaunchKernelGGL(gqe_rows_ride_kernel, dim3((unsigned)(r.plan.units + 1) + row_blocks + dense_blocks), dim3(GQE_THREADS), 0, a.stream, s32, a.t, a.idx,
                     a.last, a.ring, a.p, a.g, a.m, a.v, a.head, a.next, a.contrib, a.max_entries, a.d, a.lr, a.b1, a.b2, a.eps, (int)row_blocks, a.dsegs,
                     a.n_dsegs, a.dense_chunks, a.dcoef, a.dactive, a.dact, a.n_dact, a.hot, r);
  return hipGetLastError();
}

hipError_t gqe_launch_rows(const GqeRowsArgs& a) {
  if (a.segs.total < 1 && a.dense_chunks < 1) return hipSuccess;
  // the kernel's lane groups (d / 4 lanes per row, claim broadcast with one __shfl) must not straddle a wave
  if (a.segs.total > 0 && (a.d < 4 || (64 % (a.d >> 2)) != 0)) return hipErrorInvalidValue;
  const long long threads = (long long)a.segs.total * (a.d >> 2);
  const unsigned row_blocks = (unsigned)((threads + GQE_THREADS - 1) / GQE_THREADS);
  const unsigned dense_blocks = (unsigned)(a.dense_chunks < 512 ? a.dense_chunks : 512);
#define GO(G, S)                                                                                                          \
  hipLaunchKernelGGL((gqe_rows_kernel<G, S>), dim3(row_blocks + dense_blocks), dim3(GQE_THREADS), 0, a.stream, a.segs, a.t,  \
                     a.idx, a.last, a.ring, a.p, a.g, a.m, a.v, a.head, a.next, a.contrib, a.max_entries, a.d, a.lr, a.b1,   \
                     a.b2, a.eps, (int)row_blocks, a.dsegs, a.n_dsegs, a.dense_chunks, a.dcoef, a.dactive, a.dact, a.n_dact, a.hot)
  if (a.with_grad) {
    if (a.sorted) GO(true, true); else GO(true, false);
  } else {
    GO(false, false);
  }
#undef GO
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// the split step (gqe_train_step; gqe_split.h, GqeSplitRide in gqe_dev.h)
// ------------------------------------------------------------------------------------------
// which segment of the feed owns entry e (a scan with uniform loads of the kernel arguments), its table slot and its row
__device__ __forceinline__ bool split_entry(const GqeSplitSegs& segs, const int32_t* __restrict__ idx, int e, int& lt, int& row) {
  int k = 0;
  for (int i = 1; i < segs.n; ++i) k += (e >= segs.begin[i]) ? 1 : 0;
  lt = segs.tid[k];
  if (lt < 0) return false;
  row = idx[segs.idx_begin[k] + (e - segs.begin[k])];
  return row >= 0;   // (negative: a padding query)
}

// launch M: Adam on the d x d matrices the previous step left pending (their gradients were completed by that step's riding GEMM
// units, a kernel boundary ago; their operand-ordered copies are rewritten for the fused launch behind this one), and the stamps
// of the rows THIS step's feed names: epoch << 16 | feed entry.  Entries that name the same row race with plain stores; the
// one that stays is the row's owner in the second launch (no atomic claim there).
__global__ __launch_bounds__(GQE_THREADS) void gqe_prestep_kernel(const GqeMatStep a, float* __restrict__ p, float* __restrict__ g,
                                                                 float* __restrict__ m, float* __restrict__ v, int d, float b1, float b2,
                                                                 float eps, int mat_blocks, int mark_blocks, const GqeSplitSegs segs,
                                                                 const GqeSplitRide ride, const int32_t* __restrict__ idx,
                                                                 int32_t* __restrict__ stamp) {
  if ((int)blockIdx.x < mat_blocks) {
    matstep_body(a, p, g, m, v, d, b1, b2, eps, (int)blockIdx.x);
    return;
  }
  if ((int)blockIdx.x >= mat_blocks + mark_blocks) {
    // the riders' bookkeeping (gqe_split.h): nobody has started, no tile has finished
    const int i = ((int)blockIdx.x - mat_blocks - mark_blocks) * GQE_THREADS + (int)threadIdx.x;
    if (i == 0) *ride.done = 0;
    if (i < ride.blocks * GQE_SPLIT_PWAVES) {
      const int j = i / GQE_SPLIT_PWAVES, w = i - j * GQE_SPLIT_PWAVES;
      int lo, hi;
      split_range(ride, j, lo, hi);
      ride.progress[i] = w < ride.waves ? lo + w : 0x7fffffff;
    }
    return;
  }
  const int e = ((int)blockIdx.x - mat_blocks) * GQE_THREADS + (int)threadIdx.x;
  if (e >= segs.total) return;
  int lt, row;
  if (!split_entry(segs, idx, e, lt, row)) return;
  stamp[ride.t.head_base[lt] + row] = (ride.epoch << 16) | e;   // (several entries, one row: one of the stores stays — that entry owns the row)
}

hipError_t gqe_launch_prestep(const GqeMatStep& ms, float* p, float* g, float* m, float* v, int d, float b1, float b2, float eps,
                              const GqeSplitSegs& segs, const GqeSplitRide& ride, const int32_t* idx, int32_t* stamp, hipStream_t stream) {
  const int mat_blocks = ms.n * ((d * d) / GQE_OPT_CHUNK);
  const int mark_blocks = (segs.total + GQE_THREADS - 1) / GQE_THREADS;
  const int book_blocks = (ride.blocks * GQE_SPLIT_PWAVES + GQE_THREADS - 1) / GQE_THREADS + 1;
  hipLaunchKernelGGL(gqe_prestep_kernel, dim3((unsigned)(mat_blocks + mark_blocks + book_blocks)), dim3(GQE_THREADS), 0, stream, ms, p, g, m, v, d, b1, b2,
                     eps, mat_blocks, mark_blocks, segs, ride, idx, stamp);
  return hipGetLastError();
}

// launch B: workgroup 0 finalizes the losses, workgroups 1 .. units are the pair-GEMM units (gemm_ride_unit), then `row_blocks`
// workgroups step the named rows — d / 4 lanes per feed entry; the entry the row's stamp names (launch M) owns the row
// (a row named twice is stepped once), sums its gradient list and hot accumulators and applies Adam with exactly the eager pass's
// arithmetic — and the rest are the ordinary chunk loop over the step's vectors (relation vectors: dense gradients the fused
// tiles accumulated with atomics, complete since the kernel boundary).
// (EXTRAS: the launch has leftover riders or a bag table in its chunk loop — the common launch has neither, and its instantiation
// is compiled without them: fewer scalar registers to spill in front of every row's dependent chain)
template <bool EXTRAS>
__global__ __launch_bounds__(GQE_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void gqe_split_rows_kernel(
    const GqeDevSeg* __restrict__ segs, int n_segs, long long total_chunks, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
    float* __restrict__ v, int32_t* __restrict__ head, const int32_t* __restrict__ next, const float* __restrict__ contrib, int max_entries, int d,
    float lr, float b1, float b2, float eps, GqeStepCoef coef, GqeOptActive active, const GqeActSeg* __restrict__ act, int n_act, GqeHot hot,
    GqeGemmRide ride, const GqeSplitSegs rsegs, const GqeSplitRide sr, const int32_t* __restrict__ idx, int32_t* __restrict__ stamp, int row_blocks,
    int rider_blocks, const int32_t* __restrict__ link_contrib) {
  const GqeSplitTabs& t = sr.t;
  const int front = ride.plan.units + 1;   // (units == -1: the pair GEMM and the finalize block ran as a launch of their own)
  if ((int)blockIdx.x < front) {
    if (blockIdx.x == 0) {
      finalize_losses(ride.plan, ride.tile_loss, ride.losses);
    } else {
      gemm_ride_unit(ride.plan, ride.formulas, ride.ws, g, d, unit_of_workgroup((int)blockIdx.x - 1, ride.plan.units));
    }
    return;
  }
  if ((int)blockIdx.x >= front + row_blocks + rider_blocks) {
    // the ordinary chunk loop over the step's vectors — and over bag tables (link_contrib != NULL: word tables, stepped in full:
    // their gradient lists and hot accumulators hang on rows no feed names)
    GqeLazyArgs lazy;   // (never read: LAZY = false)
    if (EXTRAS && link_contrib) {
      opt_body<GQE_OPT_ADAM, true, false, false, false>((long long)blockIdx.x - front - row_blocks - rider_blocks,
                                                        (long long)gridDim.x - front - row_blocks - rider_blocks, segs, n_segs, total_chunks, p, g, m,
                                                        v, head, next, contrib, link_contrib, max_entries, d, lr, b1, b2, eps, coef, active, act, n_act,
                                                        lazy, hot);
    } else {
      GqeHot no_hot;
      no_hot.slot = nullptr;
      opt_body<GQE_OPT_ADAM, false, false, false, false>((long long)blockIdx.x - front - row_blocks - rider_blocks,
                                                         (long long)gridDim.x - front - row_blocks - rider_blocks, segs, n_segs, total_chunks, p, g,
                                                         m, v, head, next, contrib, nullptr, max_entries, d, lr, b1, b2, eps, coef, active, act, n_act,
                                                         lazy, no_hot);
    }
    return;
  }
  if (EXTRAS && (int)blockIdx.x >= front + row_blocks) {   // what the fused launch's riders left of the untouched rows
    split_leftover(sr, d, ((int)blockIdx.x - front - row_blocks) * GQE_WAVES + (int)(threadIdx.x >> 6));
    return;
  }
  // (Finding the named rows by their stamps instead of through the feed — a wave per wave block of 8 rows, no index indirection,
  // no claim, no duplicate entries — measured 20.1 us for this launch against 18.6: four sequential passes of dependent loads
  // per wave outlast one lane group per entry; experiment 68)
  const int tpr = d >> 2;  // lanes per row: a divisor of 64 (checked by the launcher), the group never straddles a wave
  const int e = (int)((((long long)blockIdx.x - front) * GQE_THREADS + threadIdx.x) / tpr);
  if (e >= rsegs.total) return;
  const int c4 = (threadIdx.x % tpr) * 4;
  int lt, row;
  if (!split_entry(rsegs, idx, e, lt, row)) return;
  const long long hrow = t.head_base[lt] + row;
  const long long off = t.offset[lt] + (long long)row * d + c4;
  // the owner of the row is the entry its stamp names (written by launch M: of the entries that name a row, one store stayed):
  // a load next to the others, no atomic claim
  const int owner = stamp[hrow];
  float4 pp = *reinterpret_cast<const float4*>(p + off);
  float4 mm = *reinterpret_cast<const float4*>(m + off);
  float4 vv = *reinterpret_cast<const float4*>(v + off);
  const int h0 = head[hrow];
  const int hs = hot.slot ? hot.slot[hrow] : -1;
  if (owner != ((sr.epoch << 16) | e)) return;  // another entry of the feed names the same row and owns it
  float4 gg = make_float4(0.f, 0.f, 0.f, 0.f);
  if (h0 >= 0) {
    int len;
    gg = list_gradient<false>(h0, next, contrib, nullptr, max_entries, d, c4, len);
    if (c4 == 0) {
      head[hrow] = -1;
      if (hot.slot && hs < 0) hot_promote(hot, hrow, len);
    }
  }
  if (hs >= 0) {   // a hot row: its contributions of this step sit in its accumulators
    const float4 acc = hot_take(hot, hs, d, c4);
    gg.x += acc.x;
    gg.y += acc.y;
    gg.z += acc.z;
    gg.w += acc.w;
  }
  opt_update<GQE_OPT_ADAM>(pp, mm, vv, gg, t.step_size[lt], t.bc2_sqrt[lt], lr, b1, b2, eps);
  *reinterpret_cast<float4*>(m + off) = mm;
  *reinterpret_cast<float4*>(v + off) = vv;
  *reinterpret_cast<float4*>(p + off) = pp;
}

hipError_t gqe_launch_split_rows(const GqeOptArgs& a, const GqeGemmRide& r, const GqeSplitSegs& segs, const GqeSplitRide& ride, const int32_t* idx,
                                 int32_t* stamp) {
  if (a.d < 4 || (64 % (a.d >> 2)) != 0) return hipErrorInvalidValue;
  const long long threads = (long long)segs.total * (a.d >> 2);
  const int row_blocks = (int)((threads + GQE_THREADS - 1) / GQE_THREADS);
  // (bag tables in the chunk loop: tens of thousands of chunks — as many workgroups as the plain optimiser launch would take)
  const unsigned dense_blocks = (unsigned)(a.total_chunks < 262144 ? a.total_chunks : 262144);
  static const int dbg = [] {   // GQE_SPLIT_DEBUG_B (timing experiments, WRONG results): 1 = without the GEMM units, 2 = without the named rows
    const char* e = getenv("GQE_SPLIT_DEBUG_B");
    return e ? atoi(e) : 0;
  }();
  GqeGemmRide r2 = r;
  int rb = row_blocks;
  if (dbg & 1) r2.plan.units = 0;
  if (dbg & 2) rb = 0;
  // one wave per (rider, wave) pair of the fused launch: it continues where that wave stopped
  const int riders = (ride.blocks > 0 && ride.stop) ? (ride.blocks * ride.waves + GQE_WAVES - 1) / GQE_WAVES : 0;   // (riders that do not stop leave nothing)
  const unsigned blocks = (unsigned)(r2.plan.units + 1) + (unsigned)rb + (unsigned)riders + dense_blocks;
  if (riders > 0 || a.lists)
    hipLaunchKernelGGL(gqe_split_rows_kernel<true>, dim3(blocks), dim3(GQE_THREADS), 0, a.stream, a.segs, a.n_segs, a.total_chunks, a.p, a.g, a.m, a.v, a.head,
                       a.next, a.contrib, a.max_entries, a.d, a.lr, a.b1, a.b2, a.eps, a.coef, a.active, a.act, a.n_act, a.hot, r2, segs, ride, idx, stamp,
                       rb, riders, a.lists ? a.link_contrib : nullptr);
  else
    hipLaunchKernelGGL(gqe_split_rows_kernel<false>, dim3(blocks), dim3(GQE_THREADS), 0, a.stream, a.segs, a.n_segs, a.total_chunks, a.p, a.g, a.m, a.v, a.head,
                       a.next, a.contrib, a.max_entries, a.d, a.lr, a.b1, a.b2, a.eps, a.coef, a.active, a.act, a.n_act, a.hot, r2, segs, ride, idx, stamp,
                       rb, 0, nullptr);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// The reference's decoder / encoder extension points on [d, B] tensors (decoders.py:142-150, 200-208, 228-236, 288-300,
// 311-319; encoders.py:40-43): the phases of the fused kernel as small forward-only launches for callers that score one hop
// or one intersection on their own.  Element (i, b) of an embedding batch sits at i * B + b (a contiguous torch [d, B]).
// One wave per query column; a lane owns elements lane, lane + 64, ... (d <= GQE_MAX_DIM = 256: four per lane); the d x d
// contractions run through an LDS copy of the wave's vector.  Not a hot path: no tiling over queries, no MFMA.
// ------------------------------------------------------------------------------------------
#define GQE_XNC 4   // elements per lane (GQE_MAX_DIM / 64)
struct XVec {
  float v[GQE_XNC];
};
__device__ __forceinline__ XVec x_load(const float* __restrict__ e, int d, int B, int b, int lane) {
  XVec r;
#pragma unroll
  for (int c = 0; c < GQE_XNC; ++c) {
    const int j = lane + 64 * c;
    r.v[c] = j < d ? e[(size_t)j * B + b] : 0.f;
  }
  return r;
}
__device__ __forceinline__ void x_store(float* __restrict__ e, const XVec& x, int d, int B, int b, int lane) {
#pragma unroll
  for (int c = 0; c < GQE_XNC; ++c) {
    const int j = lane + 64 * c;
    if (j < d) e[(size_t)j * B + b] = x.v[c];
  }
}
__device__ __forceinline__ float x_dot(const XVec& a, const XVec& b) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < GQE_XNC; ++c) s += a.v[c] * b.v[c];
  return wave_sum(s);
}
// y = M x (TRANS: y = M^T x, i.e. the row vector x^T M) for the wave's vector; `buf`: d floats of LDS owned by the wave
template <bool TRANS>
__device__ __forceinline__ XVec x_matvec(const float* __restrict__ M, const XVec& x, int d, float* buf, int lane) {
#pragma unroll
  for (int c = 0; c < GQE_XNC; ++c) {
    const int j = lane + 64 * c;
    if (j < d) buf[j] = x.v[c];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the wave's own LDS writes before its reads (no barrier: the buffer is the wave's)
  XVec y;
#pragma unroll
  for (int c = 0; c < GQE_XNC; ++c) {
    const int j = lane + 64 * c;
    float acc = 0.f;
    if (j < d)
      for (int k = 0; k < d; ++k) acc += (TRANS ? M[(size_t)k * d + j] : M[(size_t)j * d + k]) * buf[k];
    y.v[c] = acc;
  }
  return y;
}

// DirectEncoder.forward(nodes, mode) (encoders.py:40-43): rows of the table (a bag mode: the mean of the bag's word rows),
// L2-normalised without eps, as columns of out[d, B]
__global__ __launch_bounds__(GQE_THREADS) void gqe_x_encode_kernel(const float* __restrict__ table, const int32_t* __restrict__ rows, int B, int d,
                                                                  const int32_t* __restrict__ bag_ptr, const int32_t* __restrict__ bag_ids,
                                                                  float* __restrict__ out) {
  const int b = blockIdx.x * GQE_WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  const int r = rows[b];
  XVec x;
#pragma unroll
  for (int c = 0; c < GQE_XNC; ++c) x.v[c] = 0.f;
  if (bag_ptr) {
    const int lo = bag_ptr[r], hi = bag_ptr[r + 1];
    for (int k = lo; k < hi; ++k) {
      const float* w = table + (size_t)bag_ids[k] * d;
#pragma unroll
      for (int c = 0; c < GQE_XNC; ++c) {
        const int j = lane + 64 * c;
        if (j < d) x.v[c] += w[j];
      }
    }
    const float inv = 1.f / (float)(hi - lo);
#pragma unroll
    for (int c = 0; c < GQE_XNC; ++c) x.v[c] *= inv;
  } else {
    const float* w = table + (size_t)r * d;
#pragma unroll
    for (int c = 0; c < GQE_XNC; ++c) {
      const int j = lane + 64 * c;
      if (j < d) x.v[c] = w[j];
    }
  }
  const float inv = 1.f / sqrtf(x_dot(x, x));
#pragma unroll
  for (int c = 0; c < GQE_XNC; ++c) x.v[c] *= inv;
  x_store(out, x, d, B, b, lane);
}

// path_dec.project(embeds, rel): M . e (bilinear), e + w (TransE), e * w (bilinear-diag)
__global__ __launch_bounds__(GQE_THREADS) void gqe_x_project_kernel(int dec, const float* __restrict__ w, const float* __restrict__ e, int B, int d,
                                                                   float* __restrict__ out) {
  __shared__ float s_buf[GQE_WAVES][GQE_MAX_DIM];
  const int b = blockIdx.x * GQE_WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  XVec x = x_load(e, d, B, b, lane);
  if (dec == DEC_BILINEAR) {
    x = x_matvec<false>(w, x, d, s_buf[threadIdx.x >> 6], lane);
  } else {
#pragma unroll
    for (int c = 0; c < GQE_XNC; ++c) {
      const int j = lane + 64 * c;
      if (j < d) x.v[c] = dec == DEC_TRANSE ? x.v[c] + w[j] : x.v[c] * w[j];
    }
  }
  x_store(out, x, d, B, b, lane);
}

struct GqeXRels {
  int n;
  long long param[GQE_MAX_HOPS];
};
// path_dec.forward(embeds1, embeds2, rels): the relation chain applied to embeds1 (row vector times M_r for bilinear), then the
// cosine with embeds2 (per-norm clamp 1e-8) — bilinear-diag: the plain dot product (decoders.py:232)
__global__ __launch_bounds__(GQE_THREADS) void gqe_x_forward_kernel(int dec, const float* __restrict__ params, const GqeXRels rels,
                                                                   const float* __restrict__ e1, const float* __restrict__ e2, int B, int d,
                                                                   float* __restrict__ scores) {
  __shared__ float s_buf[GQE_WAVES][GQE_MAX_DIM];
  const int b = blockIdx.x * GQE_WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  XVec x = x_load(e1, d, B, b, lane);
  const XVec y = x_load(e2, d, B, b, lane);
  for (int r = 0; r < rels.n; ++r) {
    const float* w = params + rels.param[r];
    if (dec == DEC_BILINEAR) {
      x = x_matvec<true>(w, x, d, s_buf[threadIdx.x >> 6], lane);
    } else {
#pragma unroll
      for (int c = 0; c < GQE_XNC; ++c) {
        const int j = lane + 64 * c;
        if (j < d) x.v[c] = dec == DEC_TRANSE ? x.v[c] + w[j] : x.v[c] * w[j];
      }
    }
  }
  float s = x_dot(x, y);
  if (dec != DEC_DIAG) s /= fmaxf(sqrtf(x_dot(x, x)), COS_EPS) * fmaxf(sqrtf(x_dot(y, y)), COS_EPS);
  if (lane == 0) scores[b] = s;
}

// inter_dec(embeds1, embeds2, mode[, embeds3]): Post . agg_i relu(Pre . e_i) (pre == NULL: agg_i e_i); agg = first-arg-min / mean
__global__ __launch_bounds__(GQE_THREADS) void gqe_x_intersect_kernel(const float* __restrict__ pre, const float* __restrict__ post, int agg_min,
                                                                     const float* __restrict__ e1, const float* __restrict__ e2,
                                                                     const float* __restrict__ e3, int B, int d, float* __restrict__ out) {
  __shared__ float s_buf[GQE_WAVES][GQE_MAX_DIM];
  const int b = blockIdx.x * GQE_WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  const float* es[3] = {e1, e2, e3};
  const int n = e3 ? 3 : 2;
  XVec h;
  for (int i = 0; i < n; ++i) {
    XVec z = x_load(es[i], d, B, b, lane);
    if (pre) {
      z = x_matvec<false>(pre, z, d, s_buf[threadIdx.x >> 6], lane);
#pragma unroll
      for (int c = 0; c < GQE_XNC; ++c) z.v[c] = fmaxf(z.v[c], 0.f);
    }
#pragma unroll
    for (int c = 0; c < GQE_XNC; ++c) h.v[c] = i == 0 ? z.v[c] : (agg_min ? fminf(h.v[c], z.v[c]) : h.v[c] + z.v[c]);
  }
  if (!agg_min) {
    const float inv = 1.f / (float)n;
#pragma unroll
    for (int c = 0; c < GQE_XNC; ++c) h.v[c] *= inv;
  }
  if (post) h = x_matvec<false>(post, h, d, s_buf[threadIdx.x >> 6], lane);
  x_store(out, h, d, B, b, lane);
}

hipError_t gqe_launch_x_encode(const float* table, const int32_t* rows, int B, int d, const int32_t* bag_ptr, const int32_t* bag_ids, float* out,
                               hipStream_t stream) {
  if (B < 1) return hipSuccess;
  hipLaunchKernelGGL(gqe_x_encode_kernel, dim3((unsigned)((B + GQE_WAVES - 1) / GQE_WAVES)), dim3(GQE_THREADS), 0, stream, table, rows, B, d, bag_ptr,
                     bag_ids, out);
  return hipGetLastError();
}
hipError_t gqe_launch_x_project(int dec, const float* w, const float* e, int B, int d, float* out, hipStream_t stream) {
  if (B < 1) return hipSuccess;
  hipLaunchKernelGGL(gqe_x_project_kernel, dim3((unsigned)((B + GQE_WAVES - 1) / GQE_WAVES)), dim3(GQE_THREADS), 0, stream, dec, w, e, B, d, out);
  return hipGetLastError();
}
hipError_t gqe_launch_x_forward(int dec, const float* params, const long long* rel_params, int n_rels, const float* e1, const float* e2, int B, int d,
                                float* scores, hipStream_t stream) {
  if (B < 1) return hipSuccess;
  GqeXRels r;
  r.n = n_rels;
  for (int k = 0; k < GQE_MAX_HOPS; ++k) r.param[k] = k < n_rels ? rel_params[k] : 0;
  hipLaunchKernelGGL(gqe_x_forward_kernel, dim3((unsigned)((B + GQE_WAVES - 1) / GQE_WAVES)), dim3(GQE_THREADS), 0, stream, dec, params, r, e1, e2, B, d,
                     scores);
  return hipGetLastError();
}
hipError_t gqe_launch_x_intersect(const float* pre, const float* post, int agg_min, const float* e1, const float* e2, const float* e3, int B, int d,
                                  float* out, hipStream_t stream) {
  if (B < 1) return hipSuccess;
  hipLaunchKernelGGL(gqe_x_intersect_kernel, dim3((unsigned)((B + GQE_WAVES - 1) / GQE_WAVES)), dim3(GQE_THREADS), 0, stream, pre, post, agg_min, e1, e2,
                     e3, B, d, out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// data-parallel exchange.  Entry space = world slabs of S entries (d floats each); slab k:
//   [0, n)            contribution vectors of rank k (written by its fused kernel)
//   [n, n + R)        int32 list head of each contribution (-1: not pushed), R = ceil(n / d)
//   [n + R, S)        rank k's dense gradient spans (relation vectors / matrices, Pre / Post)
// export packs the two tails of the own slab; after ONE all-gather of the slabs, import links the other ranks'
// contributions into the local lists and replaces the dense gradients by the sum over the slabs in rank order
// (the same order on every replica).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ long long span_address(const GqeSpans& sp, long long j) {
  int k = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i)
    if (k == i && i + 1 < sp.n && j >= sp.len[i]) {
      j -= sp.len[i];
      ++k;
    }
  return sp.off[k] + j;
}

// the row-sharded step's dense gradients (gqe_dev.h): stage `world` packed copies / sum the ranks' blocks in rank order
__global__ __launch_bounds__(GQE_THREADS) void gqe_dense_stage_kernel(const GqeSpans sp, const float* __restrict__ grads, float* __restrict__ send,
                                                                     long long stride, int world) {
  const long long j = (long long)blockIdx.x * GQE_THREADS + threadIdx.x;
  if (j >= sp.total) return;
  const float x = grads[span_address(sp, j)];
  for (int p = 0; p < world; ++p) send[(long long)p * stride + j] = x;
}

__global__ __launch_bounds__(GQE_THREADS) void gqe_dense_sum_kernel(const GqeSpans sp, float* __restrict__ grads, const float* __restrict__ recv,
                                                                   long long stride, int rank, int world) {
  const long long j = (long long)blockIdx.x * GQE_THREADS + threadIdx.x;
  if (j >= sp.total) return;
  const long long a = span_address(sp, j);
  const float own = grads[a];
  float sum = 0.f;
  for (int p = 0; p < world; ++p) sum += (p == rank) ? own : recv[(long long)p * stride + j];
  grads[a] = sum;
}

hipError_t gqe_launch_dense_stage(const GqeSpans& sp, const float* grads, float* send, long long stride, int world, hipStream_t stream) {
  if (sp.total < 1) return hipSuccess;
  hipLaunchKernelGGL(gqe_dense_stage_kernel, dim3((unsigned)((sp.total + GQE_THREADS - 1) / GQE_THREADS)), dim3(GQE_THREADS), 0, stream, sp, grads, send,
                     stride, world);
  return hipGetLastError();
}

hipError_t gqe_launch_dense_sum(const GqeSpans& sp, float* grads, const float* recv, long long stride, int rank, int world, hipStream_t stream) {
  if (sp.total < 1) return hipSuccess;
  hipLaunchKernelGGL(gqe_dense_sum_kernel, dim3((unsigned)((sp.total + GQE_THREADS - 1) / GQE_THREADS)), dim3(GQE_THREADS), 0, stream, sp, grads, recv, stride,
                     rank, world);
  return hipGetLastError();
}

__global__ __launch_bounds__(GQE_THREADS) void gqe_export_kernel(float* __restrict__ contrib, const int32_t* __restrict__ rows,
                                                                const float* __restrict__ grads, int d, long long slab_base,
                                                                int32_t n, const GqeSpans sp) {
  const long long t = (long long)blockIdx.x * GQE_THREADS + threadIdx.x;
  const long long R = (n + d - 1) / d;
  if (t < n) {
    reinterpret_cast<int32_t*>(contrib + (slab_base + n) * d)[t] = rows[slab_base + t];
  } else if (t - n < sp.total) {
    const long long j = t - n;
    contrib[(slab_base + n + R) * d + j] = grads[span_address(sp, j)];
  }
}

__global__ __launch_bounds__(GQE_THREADS) void gqe_import_kernel(int32_t* __restrict__ head, int32_t* __restrict__ next,
                                                                const float* __restrict__ contrib, float* __restrict__ grads, int d,
                                                                long long slab, int32_t n, int rank, int world, const GqeSpans sp,
                                                                const GqeImportBags bags) {
  const long long t = (long long)blockIdx.x * GQE_THREADS + threadIdx.x;
  const long long links = (long long)n * (world - 1);
  const long long R = (n + d - 1) / d;
  if (t < links) {
    int k = (int)(t / n);
    const int i = (int)(t - (long long)k * n);
    if (k >= rank) ++k;
    const int h = reinterpret_cast<const int32_t*>(contrib + (k * slab + n) * d)[i];
    if (h == -1) return;  // never pushed (inactive hinge / padding)
    const int e = (int)(k * slab + i);
    if (h >= 0) {
      next[e] = __hip_atomic_exchange(head + h, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      // a bag contribution: one link node per word row of the bag, exactly what the producer's fused kernel did
      const int code = -h - 2, slot = code >> 27, bag = code & ((1 << 27) - 1);
      const int p0 = bags.csr.ptr[slot][bag], len = bags.csr.ptr[slot][bag + 1] - p0;
      const int base = e * bags.csr.max_len;   // the entry's own block of link nodes (as the producer's fused kernel numbers them)
      for (int j = 0; j < len; ++j) {
        const int node = base + j;
        const int w = bags.csr.ids[slot][p0 + j];
        next[bags.max_entries + node] =
            __hip_atomic_exchange(head + bags.head_base[slot] + w, bags.max_entries + node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bags.link_contrib[node] = e;
      }
    }
  } else if (t - links < sp.total) {
    const long long j = t - links;
    float sum = 0.f;
    for (int k = 0; k < world; ++k) sum += contrib[(k * slab + n + R) * d + j];
    grads[span_address(sp, j)] = sum;
  }
}

// ------------------------------------------------------------------------------------------
// Evaluation against candidate lists (replaces the inner loops of eval_auc_queries / eval_perc_queries, utils.py:35-91).
// The fused kernel left one record per query (its query-side vector + scalars, gqe_fused.h); this kernel streams the
// candidate rows: a group of LPR = pow2 >= d/4 lanes owns a row (one float4 per lane), a wave keeps U loads per lane in
// flight (64/LPR * U rows), the dot products are reduced inside the group with DPP steps, and the group's last lane
// writes the score — no cross-lane broadcast, no LDS, no barrier.  Work = flat blocks of GQE_EVAL_UB candidates of a
// batch; a wave finds the query of its first candidate with a 64-ary search in cand_ptr and then walks the segments.
//   score (decoders.py:142-147,200-205,228-233; model.py:97,108), with t = x / |x| (encoders.py:41-43):
//   KIND 0  cos(t, v) = (x.v / |x|) / (max(|t|, eps) * s0)                          intersections
//   KIND 1  t . v     =  x.v / |x|                                                  bilinear-diag chains
//   KIND 2  cos(v, t + w) = (x.v / |x| + s1) / (s0 * max(|t + w|, eps))             TransE chains, w = sum of the hops
// ------------------------------------------------------------------------------------------
template <int LPR>
__device__ __forceinline__ float group_sum(float x) {
  // sum over the LPR consecutive lanes of a group; valid (at least) in the group's LAST lane
  x += dpp_get<0xB1, 0xF>(x);                   // quad_perm [1,0,3,2]
  x += dpp_get<0x4E, 0xF>(x);                   // quad_perm [2,3,0,1]   -> 4 lanes
  if (LPR >= 8) x += dpp_get<0x141, 0xF>(x);    // row_half_mirror        -> 8
  if (LPR >= 16) x += dpp_get<0x140, 0xF>(x);   // row_mirror             -> 16
  if (LPR >= 32) x += dpp_get<0x142, 0xA>(x);   // row_bcast15 -> rows 1,3: lanes 16-31 / 48-63 hold 32-lane sums
  if (LPR >= 64) x += dpp_get<0x143, 0xC>(x);   // row_bcast31 -> row 3 holds the wave's sum
  return x;
}

// `cond ? *p : zero` on float4 lvalues selects between two ADDRESSES — the zero then lives in scratch and every use is a
// scratch load; this keeps the choice in registers
__device__ __forceinline__ float4 load4_if(bool cond, const float* p) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cond) v = *reinterpret_cast<const float4*>(p);
  return v;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

#define GQE_EVAL_UB 512   // candidates per workgroup (4 waves x 128)
#define GQE_EVAL_U 8      // row loads a lane keeps in flight

template <int LPR>
__global__ __launch_bounds__(GQE_THREADS) void gqe_eval_score_kernel(const GqeDynPlan plan, const GqeDevFormula* __restrict__ formulas,
                                                                    const float* __restrict__ params, const float* __restrict__ rows_base,
                                                                    const float* __restrict__ ws, const int32_t* __restrict__ idx,
                                                                    float* __restrict__ out, int d, int dec, const GqeBagTable bags) {
  constexpr int RW = 64 / LPR;  // rows per wave-wide load
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int bi = 0;
#pragma unroll
  for (int k = 1; k < GQE_LAUNCH_BATCHES; ++k) bi += ((int)blockIdx.x >= plan.unit_begin[k]) ? 1 : 0;
  const GqeDynBatch b = plan.b[bi];
  if (b.n_candidates <= 0) return;
  const GqeDevFormula* __restrict__ f = formulas + b.formula;
  const int B = b.B, n = f->n_anchors;
  const int32_t* __restrict__ cand_ptr = idx + b.idx_offset + (size_t)n * B;
  const int32_t* __restrict__ cand_rows = cand_ptr + B + 1;
  const bool chain = f->qtype <= 2;
  const int kind = !chain ? 0 : (dec == DEC_DIAG ? 1 : 2);
  const int g = lane / LPR, c4 = (lane % LPR) * 4;
  const bool act = c4 < d;
  const bool writer = (lane % LPR) == LPR - 1;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // this wave's flat range of the batch's candidates
  const long long blk = (long long)((int)blockIdx.x - b.unit_begin) * GQE_EVAL_UB + wave * (GQE_EVAL_UB / 4);
  int c = (int)min((long long)b.n_candidates, blk);
  const int c_end = (int)min((long long)b.n_candidates, blk + GQE_EVAL_UB / 4);
  if (c >= c_end) return;
  // TransE chains: w = sum of the hop vectors (batch-uniform)
  float4 w4 = zero4;
  if (kind == 2 && act)
    for (int h = 0; h < f->n_hops[0]; ++h) {
      const float4 t = *reinterpret_cast<const float4*>(params + f->hop_param[0][h] + c4);
      w4.x += t.x; w4.y += t.y; w4.z += t.z; w4.w += t.w;
    }
  // query of the first candidate: largest q with cand_ptr[q] <= c (64-ary search, the lanes probe in parallel)
  int lo = 0, hi = B;  // invariant: cand_ptr[lo] <= c < cand_ptr[hi] (cand_ptr[B] = n_candidates > c)
  while (hi - lo > 1) {
    const int step = (hi - lo + 63) / 64;
    const int probe = min(lo + (lane + 1) * step, hi);
    const bool le = probe < hi && cand_ptr[probe] <= c;
    const int k = __popcll(__ballot(le));
    const int nlo = lo + k * step;
    hi = min(nlo + step, hi);
    lo = nlo;
  }
  int q = lo;
  const int tbag = f->target_bag;
  // row-sharded mode: a candidate index is a position in the fetched-row buffer, whatever its table; bag tables stay
  // replicated (their word rows come from the local arena)
  const float* __restrict__ table = (rows_base != params && tbag < 0) ? rows_base : params + f->target_table;
  while (c < c_end) {
    const int seg_end = min(cand_ptr[q + 1], c_end);
    if (seg_end <= c) {  // (empty candidate list)
      ++q;
      continue;
    }
    const float* __restrict__ rec = ws + b.scratch_base + (size_t)q * (d + 4);
    const float4 v4 = load4_if(act, rec + c4);
    const float s0 = rec[d], s1 = rec[d + 1], s2 = rec[d + 2];
    for (; c < seg_end; c += RW * GQE_EVAL_U) {
      float4 x[GQE_EVAL_U];
      if (tbag < 0) {
        int row[GQE_EVAL_U];
#pragma unroll
        for (int u = 0; u < GQE_EVAL_U; ++u) {
          const int ci = c + u * RW + g;
          row[u] = cand_rows[ci < seg_end ? ci : c];
        }
#pragma unroll
        for (int u = 0; u < GQE_EVAL_U; ++u) x[u] = load4_if(act, table + (size_t)row[u] * d + c4);
      } else {
        // bag mode (Reddit posts): a candidate's raw vector is the mean of its word rows
        // (a select chain: indexing the by-value argument with a runtime index would put the table in scratch)
        const int32_t* __restrict__ bptr = bags.ptr[0];
        const int32_t* __restrict__ bids = bags.ids[0];
#pragma unroll
        for (int t = 1; t < GQE_MAX_BAGS; ++t)
          if (t == tbag) {
            bptr = bags.ptr[t];
            bids = bags.ids[t];
          }
#pragma unroll
        for (int u = 0; u < GQE_EVAL_U; ++u) {
          const int ci = c + u * RW + g;
          const int bag = cand_rows[ci < seg_end ? ci : c];
          const int p0 = bptr[bag], len = bptr[bag + 1] - p0;
          float4 acc = zero4;
          for (int k = 0; k < len; ++k) {
            const float4 t = load4_if(act, table + (size_t)bids[p0 + k] * d + c4);
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
          }
          const float il = 1.f / (float)len;
          x[u] = make_float4(acc.x * il, acc.y * il, acc.z * il, acc.w * il);
        }
      }
#pragma unroll
      for (int u = 0; u < GQE_EVAL_U; ++u) {
        const int ci = c + u * RW + g;
        const float xx = group_sum<LPR>(dot4(x[u], x[u]));
        const float xv = group_sum<LPR>(dot4(x[u], v4));
        const float inv = __builtin_amdgcn_rsqf(xx);             // encoders.py:41-43: t = x / |x|  (1-ulp instructions, see gqe_sqrt)
        float sc;
        if (kind == 1) {
          sc = xv * inv;
        } else if (kind == 0) {
          const float nt = fmaxf(gqe_sqrt(xx * inv * inv), COS_EPS);  // |t|, 1 up to rounding
          sc = xv * inv * gqe_rcp(nt * s0);
        } else {
          const float xw = group_sum<LPR>(dot4(x[u], w4));
          const float nu = fmaxf(gqe_sqrt(fmaf(2.f * xw, inv, xx * inv * inv) + s2), COS_EPS);   // |t + w|
          sc = fmaf(xv, inv, s1) * gqe_rcp(s0 * nu);
        }
        if (writer && ci < seg_end) out[b.out_offset + ci] = sc;
      }
    }
    c = seg_end;
    ++q;
  }
}

// query of every candidate of a batch (largest q with cand_ptr[q] <= c): the fused kernel's candidate tiles of a
// full-Bilinear chain batch look their anchor up through it
__global__ __launch_bounds__(GQE_THREADS) void gqe_expand_ptr_kernel(const int32_t* __restrict__ ptr, int nq, int nc, int32_t* __restrict__ query_of) {
  const int c = (int)blockIdx.x * GQE_THREADS + threadIdx.x;
  if (c >= nc) return;
  int lo = 0, hi = nq;   // invariant: ptr[lo] <= c < ptr[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ptr[mid] <= c) lo = mid; else hi = mid;
  }
  query_of[c] = lo;
}

hipError_t gqe_launch_expand_ptr(const int32_t* cand_ptr, int n_queries, int n_candidates, int32_t* query_of, hipStream_t stream) {
  if (n_candidates < 1) return hipSuccess;
  hipLaunchKernelGGL(gqe_expand_ptr_kernel, dim3((unsigned)((n_candidates + GQE_THREADS - 1) / GQE_THREADS)), dim3(GQE_THREADS), 0, stream, cand_ptr,
                     n_queries, n_candidates, query_of);
  return hipGetLastError();
}

hipError_t gqe_launch_eval_score(const GqeFusedArgs& a, int dec, float* scores) {
  if (a.plan.units < 1) return hipSuccess;
  const float* rows_base = a.fetched ? a.fetched : a.params;
  int lpr = 4;
  while (lpr * 4 < a.d) lpr *= 2;
#define GO(L)                                                                                                                 \
  hipLaunchKernelGGL((gqe_eval_score_kernel<L>), dim3(a.plan.units), dim3(GQE_THREADS), 0, a.stream, a.plan, a.formulas, a.params, \
                     rows_base, a.ws, a.idx, scores, a.d, dec, a.bags)
  switch (lpr) {
    case 4: GO(4); break;
    case 8: GO(8); break;
    case 16: GO(16); break;
    case 32: GO(32); break;
    default: GO(64); break;
  }
#undef GO
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Ranking statistics on the device: only query-level numbers leave the GPU.
//   percentile (utils.py:26-33: scipy.stats.percentileofscore, kind 'rank') of a list's FIRST score among the others;
//   pair counts for the ROC AUC (utils.py:63,66: sklearn roc_auc_score = Mann-Whitney with ties at 1/2):
//   count2 += sum_i sum_j 2 [p_i > n_j] + [p_i == n_j], NaN scores read as 0 (np.nan_to_num).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GQE_THREADS) void gqe_rank_kernel(const float* __restrict__ scores, const int32_t* __restrict__ ptr, int nq,
                                                              double* __restrict__ percentile) {
  const int q = (int)(((long long)blockIdx.x * GQE_THREADS + threadIdx.x) >> 6);
  if (q >= nq) return;
  const int lane = threadIdx.x & 63;
  const int p0 = ptr[q], p1 = ptr[q + 1];
  const float s = scores[p0];
  float left = 0.f, right = 0.f, bad = (s != s) ? 1.f : 0.f;   // counts < 2^24: exact in fp32
  for (int i = p0 + 1 + lane; i < p1; i += 64) {
    const float x = scores[i];
    left += (x < s) ? 1.f : 0.f;
    right += (x <= s) ? 1.f : 0.f;
    bad += (x != x) ? 1.f : 0.f;
  }
  left = wave_sum(left);
  right = wave_sum(right);
  bad = wave_sum(bad);
  const int n = p1 - p0 - 1;
  // a NaN anywhere in the list (or as the target's score) makes the percentile nan, as scipy's percentileofscore
  // (nan_policy 'propagate') does: the mean over queries must then be nan, not pulled towards 0
  if (lane == 0) percentile[q] = (n > 0 && bad == 0.f) ? ((double)left + (double)right + (right > left ? 1.0 : 0.0)) * 50.0 / (double)n : nan("");
}

__global__ __launch_bounds__(GQE_THREADS) void gqe_auc_kernel(const float* __restrict__ pos, long long n_pos, const float* __restrict__ neg,
                                                             long long n_neg, unsigned long long* __restrict__ count2) {
  __shared__ float s_neg[1024];
  const long long i = (long long)blockIdx.x * GQE_THREADS + threadIdx.x;
  float p = i < n_pos ? pos[i] : 0.f;
  if (p != p) p = 0.f;
  unsigned long long acc = 0;
  const long long j0 = (long long)blockIdx.y * 16384, j1 = min(n_neg, j0 + 16384);
  for (long long t0 = j0; t0 < j1; t0 += 1024) {
    __syncthreads();
    for (int k = threadIdx.x; k < 1024; k += GQE_THREADS) {
      float x = t0 + k < j1 ? neg[t0 + k] : 0.f;
      s_neg[k] = (x != x) ? 0.f : x;
    }
    __syncthreads();
    const int m = (int)min((long long)1024, j1 - t0);
    unsigned int a = 0;
    for (int k = 0; k < m; ++k) a += (p > s_neg[k]) ? 2u : ((p == s_neg[k]) ? 1u : 0u);
    acc += a;
  }
  if (i >= n_pos) acc = 0;
  // wave reduction of the 64-bit counts, then one atomic per wave
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(count2, acc);
}

hipError_t gqe_launch_rank(const float* scores, const int32_t* ptr, int nq, double* percentile, hipStream_t stream) {
  if (nq < 1) return hipSuccess;
  const long long threads = (long long)nq * 64;
  hipLaunchKernelGGL(gqe_rank_kernel, dim3((unsigned)((threads + GQE_THREADS - 1) / GQE_THREADS)), dim3(GQE_THREADS), 0, stream, scores, ptr,
                     nq, percentile);
  return hipGetLastError();
}

hipError_t gqe_launch_auc(const float* pos, long long n_pos, const float* neg, long long n_neg, unsigned long long* count2, hipStream_t stream) {
  if (n_pos < 1 || n_neg < 1) return hipSuccess;
  dim3 grid((unsigned)((n_pos + GQE_THREADS - 1) / GQE_THREADS), (unsigned)((n_neg + 16383) / 16384));
  hipLaunchKernelGGL(gqe_auc_kernel, grid, dim3(GQE_THREADS), 0, stream, pos, n_pos, neg, n_neg, count2);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// row-sharded data parallelism ("owner computes", gqe_set_shard).  A request names a row of a local shard by its list
// head index (head_base of the local table + local row), which is also where its gradient contribution is linked.
// ------------------------------------------------------------------------------------------
// d/4 lanes per row, four rows per lane group in flight (the requests and the four row loads are all issued before the first
// store: a 512-byte row per 32 lanes with one load in flight measured 1.8 TB/s on 9 MB)
#define GQE_SERVE_U 4
// Requests [own_lo, own_lo + own_n) are this rank's own: their rows go straight to where the fused kernel reads them
// (own_out, the own block of the fetched-row buffer) instead of through the send buffer and a copy.
// link != 0 (a margin step driven by gqe_shard_step): the contribution that will answer request j is linked onto its row's list
// HERE — entry j of the receive buffer, or own_entry + (j - own_lo) for this rank's own block — while the row is being served:
// which entry belongs to which row is known from the request list alone, long before the contributions arrive, so the step
// needs no separate link launch between the second all-to-all and the optimiser pass.
// own_out == NULL with own_n > 0: the own block is not served at all (the fused kernel reads those rows from the shard and
// links their contributions itself, GQE_OWN_ROW): the launch covers the n - own_n requests of the other ranks.
__global__ __launch_bounds__(GQE_THREADS) void gqe_shard_serve_kernel(const float* __restrict__ p, const int32_t* __restrict__ req,
                                                                     long long n, float* __restrict__ out, int d, const GqeShardTabs t,
                                                                     long long own_lo, long long own_n, float* __restrict__ own_out,
                                                                     int32_t* __restrict__ head, int32_t* __restrict__ next, long long own_entry,
                                                                     int link) {
  const int tpr = d >> 2;
  const int gpb = GQE_THREADS / tpr;                     // lane groups per workgroup
  const int g = threadIdx.x / tpr, c4 = (threadIdx.x - g * tpr) * 4;
  if (g >= gpb) return;
  const bool skip_own = own_out == nullptr && own_n > 0;
  const long long j0 = ((long long)blockIdx.x * gpb + g) * GQE_SERVE_U;
  long long jj[GQE_SERVE_U];   // the request a slot of this lane group handles (n: none)
  int h[GQE_SERVE_U];
#pragma unroll
  for (int u = 0; u < GQE_SERVE_U; ++u) {
    jj[u] = j0 + u;
    if (skip_own && jj[u] >= own_lo) jj[u] += own_n;
    if (jj[u] > n) jj[u] = n;
    h[u] = jj[u] < n ? req[jj[u]] : -1;
  }
  float4 v[GQE_SERVE_U];
#pragma unroll
  for (int u = 0; u < GQE_SERVE_U; ++u) {
    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (h[u] >= 0) {
      int k = 0;
#pragma unroll
      for (int i = 1; i < GQE_LAZY_TABLES; ++i) k += (i < t.n && h[u] >= t.head_base[i]) ? 1 : 0;
      v[u] = *reinterpret_cast<const float4*>(p + t.offset[k] + (long long)(h[u] - t.head_base[k]) * d + c4);
    }
  }
#pragma unroll
  for (int u = 0; u < GQE_SERVE_U; ++u) {
    const long long j = jj[u], k = skip_own ? -1 : j - own_lo;
    if (j < n) *reinterpret_cast<float4*>((k >= 0 && k < own_n ? own_out + k * d : out + j * d) + c4) = v[u];
    if (link && c4 == 0 && j < n && h[u] >= 0) {
      const int e = (int)(k >= 0 && k < own_n ? own_entry + k : j);
      next[e] = __hip_atomic_exchange(head + h[u], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Contribution j answers request j: entry j of the receive buffer — except this rank's own block [own_lo, own_lo + own_n),
// whose contributions stayed where the fused kernel wrote them: entries own_entry + (j - own_lo).
__global__ __launch_bounds__(GQE_THREADS) void gqe_shard_link_kernel(int32_t* __restrict__ head, int32_t* __restrict__ next,
                                                                    const int32_t* __restrict__ req, long long n, long long own_lo,
                                                                    long long own_n, long long own_entry) {
  const long long j = (long long)blockIdx.x * GQE_THREADS + threadIdx.x;
  if (j >= n) return;
  const int h = req[j];
  const long long k = j - own_lo;
  const bool own = k >= 0 && k < own_n;
  if (own && own_entry < 0) return;   // (the fused kernel linked the own block's contributions itself: GQE_OWN_ROW)
  const int e = (int)(own ? own_entry + k : j);
  if (h >= 0) next[e] = __hip_atomic_exchange(head + h, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

hipError_t gqe_launch_shard_serve(const float* params, const int32_t* req, long long n, float* out, int d, const GqeShardTabs& t,
                                  long long own_lo, long long own_n, float* own_out, int32_t* head, int32_t* next, long long own_entry,
                                  int link, hipStream_t stream) {
  const long long work = own_out == nullptr && own_n > 0 ? n - own_n : n;
  if (work < 1) return hipSuccess;
  const long long rows_per_block = (long long)(GQE_THREADS / (d >> 2)) * GQE_SERVE_U;
  hipLaunchKernelGGL(gqe_shard_serve_kernel, dim3((unsigned)((work + rows_per_block - 1) / rows_per_block)), dim3(GQE_THREADS), 0, stream,
                     params, req, n, out, d, t, own_lo, own_n, own_out, head, next, own_entry, link);
  return hipGetLastError();
}

hipError_t gqe_launch_shard_link(int32_t* head, int32_t* next, const int32_t* req, long long n, long long own_lo, long long own_n,
                                 long long own_entry, hipStream_t stream) {
  if (n < 1) return hipSuccess;
  hipLaunchKernelGGL(gqe_shard_link_kernel, dim3((unsigned)((n + GQE_THREADS - 1) / GQE_THREADS)), dim3(GQE_THREADS), 0, stream, head,
                     next, req, n, own_lo, own_n, own_entry);
  return hipGetLastError();
}

hipError_t gqe_launch_export(float* contrib, const int32_t* rows, const float* grads, int d, long long slab_base, int32_t n,
                             const GqeSpans& sp, hipStream_t stream) {
  const long long total = (long long)n + sp.total;
  if (total < 1) return hipSuccess;
  hipLaunchKernelGGL(gqe_export_kernel, dim3((unsigned)((total + GQE_THREADS - 1) / GQE_THREADS)), dim3(GQE_THREADS), 0, stream,
                     contrib, rows, grads, d, slab_base, n, sp);
  return hipGetLastError();
}

hipError_t gqe_launch_import(int32_t* head, int32_t* next, const float* contrib, float* grads, int d, long long slab, int32_t n,
                             int rank, int world, const GqeSpans& sp, const GqeImportBags& bags, hipStream_t stream) {
  const long long total = (long long)n * (world - 1) + sp.total;
  if (total < 1) return hipSuccess;
  hipLaunchKernelGGL(gqe_import_kernel, dim3((unsigned)((total + GQE_THREADS - 1) / GQE_THREADS)), dim3(GQE_THREADS), 0, stream,
                     head, next, contrib, grads, d, slab, n, rank, world, sp, bags);
  return hipGetLastError();
}

hipError_t gqe_launch_opt(const GqeOptArgs& a) {
  long long blocks = a.total_chunks < 262144 ? a.total_chunks : 262144;  // one chunk per workgroup measures best (12288 chunks at Bio d=128: 49.7 vs 50.8 us with 4096 grid-striding workgroups)
  if (blocks < 1) blocks = 1;
  switch (a.mode) {
    case GQE_OPT_ADAM: launch_opt_mode<GQE_OPT_ADAM>(a, (unsigned)blocks); break;
    case GQE_OPT_SGD: launch_opt_mode<GQE_OPT_SGD>(a, (unsigned)blocks); break;
    case GQE_OPT_ZERO: launch_opt_mode<GQE_OPT_ZERO>(a, (unsigned)blocks); break;
    default: launch_opt_mode<GQE_OPT_MATERIALIZE>(a, (unsigned)blocks); break;
  }
  return hipGetLastError();
}
