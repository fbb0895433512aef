// gqe_split.h — the "split" training step (gqe_train_step, include/gqe.h): device side of the Adam work that shares a launch
// with the fused forward / backward tiles.
//
// A dense Adam step moves every row of every stepped table, but a step's batches name ~15-20 % of them.  The other rows have
// no gradient: their update reads nothing the fused kernel writes, and the fused kernel reads none of them.  So the step is
//     launch M   Adam on the d x d matrices of the PREVIOUS step (gqe_prestep_kernel) + stamp[row] := epoch << 16 | e for every
//                entry e of this step's index feed (plain stores: of the entries that name a row, one — whichever store lands
//                last — is left in the stamp and OWNS the row in launch B; the epoch grows by 1 per step, 15 bits),
//     launch A   the fused tiles  |  "rider" workgroups: Adam (zero gradient) over rows without this step's stamp, until the
//                last tile has finished,
//     launch B   loss finalize + pair-GEMM units  |  the named rows: the entry the stamp names owns the row (a load: duplicates in
//                the feed resolved without an atomic), list / hot-accumulator gradient, Adam  |  what the riders left of the other
//                rows  |  the relation vectors,
// and the matrices of this step wait for the next launch M (or gqe_optimizer_sync / any other entry point).
// There is no dependency inside a launch (on this part a dependency is a kernel boundary).
//
// The rider lives in the fused kernel's register / LDS allocation: a wave owns blocks of GQE_SPLIT_WROWS consecutive rows of
// one table, addressed through one buffer descriptor per arena (p, m, v) whose extent is the block — rows past the table's end
// and rows whose stamp is set get an out-of-range offset: they read nothing and store nothing, decided by the hardware's range
// check with all of EXEC enabled, so the loop has no lane-divergent branch and GQE_SPLIT_U row slices are in flight per lane.
#ifndef GQE_SPLIT_H
#define GQE_SPLIT_H

#include "gqe_adam.h"
#include "gqe_common.h"

// row slices (float4 of p, m, v each) a lane keeps in flight: 4 in the 16-wave kernels (<= 128 VGPRs), 2 next to the 8-wave tiles
// (held to 80 VGPRs: three workgroups per CU — with 4 the rider spilled a register there)
#ifndef GQE_SPLIT_U
#if defined(GQE_FW) && GQE_FW == 8
#define GQE_SPLIT_U 2
#else
#define GQE_SPLIT_U 4
#endif
#endif

#ifndef GQE_SPLIT_AUX
#define GQE_SPLIT_AUX 0   // cache policy of the riders' loads and stores (2 = nt)
#endif
__device__ __forceinline__ float4 split_ld(__amdgpu_buffer_rsrc_t rs, int voff) {
  const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, GQE_SPLIT_AUX));
  return make_float4(t[0], t[1], t[2], t[3]);
}
__device__ __forceinline__ void split_st(__amdgpu_buffer_rsrc_t rs, int voff, const float4& x) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const f32x4 t = {x.x, x.y, x.z, x.w};
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), rs, voff, 0, GQE_SPLIT_AUX);
}

// One wave block (GQE_SPLIT_WROWS consecutive rows of one table): `st` = the stamps of its rows, lane l holding row l's (loaded
// by the caller one block ahead).  d % 4 == 0 and (d / 4) divides 64.
__device__ __forceinline__ int split_stamps(const GqeSplitRide& r, int wb, int lane) {
  int ti = 0;
#pragma unroll
  for (int k = 1; k < GQE_SPLIT_TABLES; ++k) ti += (k < r.t.n && wb >= r.t.blk_begin[k]) ? 1 : 0;
  const long long r0 = (long long)(wb - r.t.blk_begin[ti]) * GQE_SPLIT_WROWS;
  // 1 = named by this step's feed (stamp = epoch << 16 | owning feed entry), or past the table's end: nothing to do
  if (lane >= GQE_SPLIT_WROWS || r0 + lane >= r.t.rows[ti]) return 1;
  const int s = r.stamp[r.t.head_base[ti] + r0 + lane];
  return (s >> 16) == r.epoch ? 1 : 0;
}

// rider j's wave blocks [lo, hi): the lead riders (j < r.lead: on a CU of their own for the whole launch) own r.share times the
// tail riders' r.per
__device__ __forceinline__ void split_range(const GqeSplitRide& r, int j, int& lo, int& hi) {
  const int total = r.t.blk_begin[r.t.n];
  const int big = r.per * r.share;
  lo = j < r.lead ? j * big : r.lead * big + (j - r.lead) * r.per;
  hi = min(lo + (j < r.lead ? big : r.per), total);
  lo = min(lo, total);
}

__device__ __forceinline__ void split_block(const GqeSplitRide& r, const int d, const int wb, const int st) {
  const int tpr = d >> 2, rpw = 64 / tpr;
  const int lane = threadIdx.x & 63, wrow = lane / tpr;
  const int c4 = (lane - wrow * tpr) * 4;
  const float b1c = 1.f - r.b1, b2c = 1.f - r.b2;
  const int oob = 1 << 30;
  int ti = 0;
#pragma unroll
  for (int k = 1; k < GQE_SPLIT_TABLES; ++k) ti += (k < r.t.n && wb >= r.t.blk_begin[k]) ? 1 : 0;
  const long long r0 = (long long)(wb - r.t.blk_begin[ti]) * GQE_SPLIT_WROWS;
  const long long left = r.t.rows[ti] - r0;
  const int nrows = left < GQE_SPLIT_WROWS ? (int)left : GQE_SPLIT_WROWS;
  const long long base = r.t.offset[ti] + r0 * d;
  const int bytes = nrows * d * 4;
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(r.p + base, 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(r.m + base, 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(r.v + base, 0, bytes, 0x00020000);
  const float ss = r.t.step_size[ti], ibc = __builtin_amdgcn_rcpf(r.t.bc2_sqrt[ti]);
#pragma unroll 1
  for (int it = 0; it < GQE_SPLIT_WROWS; it += rpw * GQE_SPLIT_U) {
    float4 pp[GQE_SPLIT_U], mm[GQE_SPLIT_U], vv[GQE_SPLIT_U];
    int vo[GQE_SPLIT_U];
#pragma unroll
    for (int u = 0; u < GQE_SPLIT_U; ++u) {
      const int lr = it + u * rpw + wrow;
      const int named = __shfl(st, lr & 63);
      vo[u] = (wrow < rpw && lr < GQE_SPLIT_WROWS && named == 0) ? (lr * d + c4) * 4 : oob;
      pp[u] = split_ld(rp, vo[u]);
      mm[u] = split_ld(rm, vo[u]);
      vv[u] = split_ld(rv, vo[u]);
    }
#pragma unroll
    for (int u = 0; u < GQE_SPLIT_U; ++u) {
      // exactly the eager pass's arithmetic on a row without a gradient (opt_update -> gqe_adam1 with g = 0)
      gqe_adam1(pp[u].x, mm[u].x, vv[u].x, 0.f, ss, ibc, b1c, r.b2, b2c, r.eps);
      gqe_adam1(pp[u].y, mm[u].y, vv[u].y, 0.f, ss, ibc, b1c, r.b2, b2c, r.eps);
      gqe_adam1(pp[u].z, mm[u].z, vv[u].z, 0.f, ss, ibc, b1c, r.b2, b2c, r.eps);
      gqe_adam1(pp[u].w, mm[u].w, vv[u].w, 0.f, ss, ibc, b1c, r.b2, b2c, r.eps);
      split_st(rm, vo[u], mm[u]);
      split_st(rv, vo[u], vv[u]);
      split_st(rp, vo[u], pp[u]);
    }
  }
}

// A rider workgroup of the FUSED launch (WAVES waves): rider j owns the wave blocks [j * per, (j + 1) * per) of the step, wave w of
// it the blocks lo + w, lo + w + WAVES, ... — no barrier, no atomic.  Riders start at different times (a few lead the launch on
// CUs of their own, the others follow the tiles as CUs become free) and all of them STOP when the launch's last tile has
// finished (r.done: a relaxed counter the tiles bump on their way out — a hint, nothing is ordered by it): every wave leaves the
// index of its next block in progress[j][w], and the step's second launch — which streams at full bandwidth next to the
// latency-bound matrix-gradient units — takes over from there (split_leftover).  The launch ends with its tiles, whatever the
// riders got done by then.
template <int WAVES>
__device__ __forceinline__ void split_rider(const GqeSplitRide& r, const int d, const int j, int* flag) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wave >= r.waves) return;   // (r.waves < WAVES: a throttled rider — fewer streaming waves per CU)
  int lo, hi;
  split_range(r, j, lo, hi);
  int b = lo + wave;
  // Stopping with the tiles (r.stop): ONE wave of the workgroup polls the finished-tile count, every other block, and raises a
  // flag in LDS that the others read — every wave polling the counter itself (1 500 waves on one address, each every ~2.5 us)
  // saturated that address's L2 channel and slowed everything that shares it, the tiles included (96 -> 71 us per step).
  if (r.stop && lane == 0) *flag = 0;
  if (b < hi) {
    int st = split_stamps(r, b, lane);
    int done = 0, round = 0, poll = 0;
    if (r.stop && wave == 0) poll = __hip_atomic_load(r.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (b < hi && !done) {
      const int nb = b + r.waves;
      const int st_next = nb < hi ? split_stamps(r, nb, lane) : 1;   // requested a block ahead
      split_block(r, d, b, st);
      st = st_next;
      b = nb;
      if (r.stop) {
        if (wave == 0 && (++round & 1) == 0) {
          if (__builtin_amdgcn_readfirstlane(poll) >= r.tiles && lane == 0) *flag = 1;
          poll = __hip_atomic_load(r.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (read at the next poll: a round trip nobody waits for)
        }
        done = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(flag));
      }
    }
  }
  if (lane == 0) r.progress[j * GQE_SPLIT_PWAVES + wave] = b;
}

// The second launch's riders: wave q of the launch continues pair (rider j, wave w) of the fused launch from progress[j][w].
__device__ __forceinline__ void split_leftover(const GqeSplitRide& r, const int d, const int q) {
  const int lane = threadIdx.x & 63;
  const int j = q / r.waves, w = q - j * r.waves;
  if (j >= r.blocks) return;
  int lo, hi;
  split_range(r, j, lo, hi);
  int b = __builtin_amdgcn_readfirstlane(r.progress[j * GQE_SPLIT_PWAVES + w]);
  if (b >= hi) return;
  int st = split_stamps(r, b, lane);
  while (b < hi) {
    const int nb = b + r.waves;
    const int st_next = nb < hi ? split_stamps(r, nb, lane) : 1;
    split_block(r, d, b, st);
    st = st_next;
    b = nb;
  }
}

#endif
