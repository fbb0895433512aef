// gqe_fused.h — the fused forward(+backward) kernel of the conjunctive-query path (gfx950).
//
// What is computed follows netquery/model.py:70-127, encoders.py:40-43, decoders.py:142-150,
// 200-208, 228-236, 288-319 (maths: SURVEY.md Appendix B / oracle/netquery_numpy.py).  How:
//   * one workgroup = 16 wave64 (GQE_FW) owns a TILE of 16 queries of ONE batch (= one Formula), so every
//     relation parameter is workgroup-uniform; one grouped launch covers all batches of an iteration;
//   * the tile's table rows are fetched with one coalesced index read, then EVERY embedding row the
//     tile needs (target, negative, up to 3 anchors per query) is requested at once — one wave per row,
//     lanes strided over the row — and stays in registers for the whole kernel (the backward re-uses it);
//   * reductions over a row are DPP wave reductions whose result lands in an SGPR;
//   * element-wise decoders (bilinear-diag, TransE) are wave ops on those registers;
//   * d x d contractions (Bilinear hops, SetIntersection Pre/Post, and their transposes in the backward)
//     run on the matrix cores with the exact-fp32 v_mfma_f32_16x16x4_f32: 16-query LDS tiles are the B
//     operand, the matrix streams from L2 as the A operand with all of a wave's loads in flight at once;
//     the Pre matrix is loaded ONCE for all 2-3 branches, and relu + first-arg-min / mean run on the MFMA
//     accumulators before anything is written back;
//   * the positive and the negative score share the query side (the reference recomputes it);
//   * a row gradient is written once, coalesced, as a contribution entry and linked onto its table row
//     with ONE 4-byte atomic exchange (the optimiser pass sums the lists); relation-vector gradients and
//     the loss are reduced per workgroup before they touch an atomic; d x d matrix gradients are deferred
//     to the pair-GEMM kernel through (left,right) scratch rows.
#ifndef GQE_FUSED_H
#define GQE_FUSED_H

#include "gqe_common.h"

// Waves per workgroup of this translation unit's kernels (gqe_fused_inst.hip is compiled once per (decoder, MLP, GQE_FW)):
//   16 = one query row per wave: the shortest dependent chain per tile, 1024-thread workgroups (<= 128 VGPRs per lane);
//    8 = two rows per wave, 512-thread workgroups (<= 256 VGPRs): d > 128 without spills, and two workgroups per CU at
//        d = 128 (2 x 74 KB LDS) when a launch has more tiles than the chip has CUs.
#ifndef GQE_FW
#define GQE_FW 16
#endif
#define GQE_FWT (64 * GQE_FW)
#define RPW (GQE_TQ / GQE_FW)  // query rows owned by one wave
#include "gqe_split.h"   // (after GQE_FW: the rider's depth depends on the workgroup shape)

struct TileEnv {
  GqeDynBatch b;  // by value: the plan is a kernel argument, never take its address
  const GqeDevFormula* f;
  const float* params;
  // Row-sharded data parallelism (gqe_set_shard): the embedding rows of this call were fetched from their owner ranks
  // into one buffer, in the order of the index feed; an index is then a position in that buffer (whatever its table),
  // the row's gradient contribution is written to the same position of the send buffer, and nothing is linked here —
  // the owner links what it receives.  Indices >= GQE_OWN_ROW (gqe_dev.h) are rows of this rank's own shard: gathered from the
  // arena and linked here, like every row of an unsharded step.
  const float* rows;  // where rows are gathered from: the parameter arena, or the fetched-row buffer
  bool sharded;
  // Bag (EmbeddingBag) tables stay replicated in row-sharded mode: their rows are gathered from the local replica, and a
  // bag's contribution is linked onto the LOCAL word-row lists — it lives in the entry space the optimiser walks, behind
  // the entries received from the other ranks (bag_shift), not in the send buffer.
  float* contrib_bag;
  long long bag_shift;
  float* grads;
  float* ws;  // scratch (floats)
  int32_t* head;
  int32_t* next;
  float* contrib;
  int32_t* link_contrib;
  int32_t* link_counter;
  int32_t max_entries;
  int d, DP, wave, lane, q0;  // q0 = first query of this tile
  bool wt;                    // write-through stores for what the next kernels read (vstore_wt)
  // hot rows (GqeHot, gqe_dev.h; only the two pointers travel — the capacity is GQE_HOT_SLOTS): hot_slot == NULL -> every
  // contribution is an entry on a list
  const int32_t* hot_slot;
  float* hot_acc;
  int32_t* hot_sub;           // sub-lists of the hot word rows (GqeHot.sub; NULL: hot words are added directly)
  int rep;                    // the replica of a hot row's accumulators this wave adds to: (its XCD, its wave index mod 4)
  int hotv;                   // lane role * RPW + rr: the hot slot of the plain row this wave owns for that role (-1: not hot).
                              // ONE vector load issued in front of the row gathers (same in-order counter: it has landed when
                              // the rows have) — five scalar loads would share their counter with the LDS reads of the indices
                              // and chain five L2 round trips in front of the gathers (+1.5 us per tile, measured)
};

// A hot row's contribution is ADDED to one of its GQE_HOT_REPS dense accumulators (fire-and-forget float atomics) instead of
// being written as an entry and linked: the optimiser pass then reads GQE_HOT_REPS vectors instead of chasing a list of
// hundreds (hub nodes) or thousands (frequent words) of entries, one dependent load each.
template <int NC, bool FULL>
__device__ __forceinline__ void hot_add(const TileEnv& e, int slot, const Vec<NC>& gx) {
  float* acc = e.hot_acc + GQE_HOT_ROW(e.rep & (GQE_HOT_REPS_OF(slot) - 1), GQE_HOT_SLOT_OF(slot)) * e.d;
  gatomic_add<NC, FULL>(acc, gx, e.d, e.lane);
}

// Rows another kernel (on whatever XCD) reads next — scratch rows for the pair GEMM, contribution entries for the optimiser
// pass — are WRITTEN THROUGH (sc1) in launches of few tiles: the L2 of an XCD is not coherent with the others', so plain
// stores sit dirty in it until the write-back at the kernel boundary, and at B = 512 that burst is on the step's critical
// path (fused 28.7 -> 27.4 us, step 87.0 -> 85.6 us in an A / B / C on one box).  Launches with thousands of tiles keep plain
// stores: there the boundary is amortised and write-through cost 1.5 % (B = 8192: 227 -> 230.5 us).
template <int NC, bool FULL>
__device__ __forceinline__ void vstore_wt(bool wt, float* p, const Vec<NC>& x, int d, int lane) {
  if (wt)   // workgroup-uniform
    gstore_sc1<NC, FULL>(p, x, d, lane);
  else
    gstore<NC, FULL>(p, x, d, lane);
}

__device__ __forceinline__ float* scratch_row(const TileEnv& e, int slot, int r) {
  return e.ws + e.b.scratch_base + ((size_t)slot * e.b.Bpad + e.q0 + r) * e.d;
}

// ------------------------------------------------------------------------------------------
// MFMA tile contractions.  v_mfma_f32_16x16x4_f32: lane l feeds A[i=l&15][k=l>>4] and
// B[k=l>>4][j=l&15], receives D[i=4*(l>>4)+r][j=l&15] (r = 0..3).  A lane fetches 4 consecutive k
// per k-block, so MFMA step s of k-block kb contracts k = 16*kb + 4*(l>>4) + s for A and B alike.
// Wave w owns the 16-row output slabs i0 = 16*(w, w+8, ...).
//   TRANS = false: A[i][k] = M[i][k]  (M . x : "project", decoders.py:150; Pre/Post forward)
//   TRANS = true : A[i][k] = M[k][i]  (M^T . x : x^T M of decoders.py:145; every backward)
// (Since the matrices are read from operand-ordered copies the two differ only in WHICH copy the caller passes — GQE_TILED(TRANS,
// field) — and the template argument of the contraction helpers documents that choice at the call site.)
// ------------------------------------------------------------------------------------------
// The matrix is read from its OPERAND-ORDERED copy (gqe_dev.h, GQE_TILE_INDEX; M selects the copy of M or of M^T — the loader
// is the same for both): the A operand of row block i0 and k-block kb is one contiguous kilobyte, 16 B per lane.  The copy is
// addressed as ONE buffer of d * d floats: the lane's offset inside the row block's tiles is computed once, the k-block step is a
// wave-uniform immediate of the instruction (no VALU address arithmetic per load).
// Guarded kernels (FULL = false, d < 64 NC): the contractions run over the PADDED extent with compile-time trip counts — a
// run-time k-block count made every slab element conditionally zero and the row-block loop a real loop, and the 16-wave
// kernels spilled hundreds of registers at their 128-VGPR limit.  k-blocks past d / 16 of a row block read the NEXT row
// block's tiles (finite values) or, behind the last one, nothing (out of range: 0) — and meet the zero columns of the source tile.
template <int KB, int KBT, bool FULL>
__device__ __forceinline__ void load_a_slab(float4 (&a)[KB], const float* __restrict__ M, int d, int i0, int lq, int lk, int kb0) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(M), 0, d * d * 4, 0x00020000);
  const int voff = (i0 >> 4) * (d >> 4) * 1024 + (lq + 16 * lk) * 16;
#pragma unroll
  for (int j = 0; j < KB; ++j) {
    const int kb = kb0 + j;
    if (kb >= KBT) {   // (compile time: the last group of a slab whose k-block count is not a multiple of the group)
      a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, kb * 1024, 0);
      a[j] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    }
  }
}

__device__ __forceinline__ f32x4 mfma4(const float4& a, const float4& b, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  return acc;
}

// The A slab of an output row block is fetched in groups of <= KG k-blocks (KG float4 per lane in flight): all of it
// at d <= 128, two / three / four rounds beyond — a 16-float4 slab (d = 256) next to the rows a wave keeps in registers
// would spill.
// the operand-ordered copy of a matrix of the formula (TR: of its transpose), gqe_dev.h
#define GQE_TILED(TR, field) (ws + f->field + ((TR) ? f->tile_t : 0))
// KB k-blocks in groups of KG: double-buffered half-groups whenever the slab takes more than one group and splits evenly
#define GQE_DBUF(KB, KG) (!GQE_NO_DBUF && GQE_FW == 16 && (KB) > (KG) && (KG) % 2 == 0 && (KB) % ((KG) / 2) == 0)
#ifndef GQE_NO_DBUF
#define GQE_NO_DBUF 0
#endif
#define GQE_KG ((GQE_FW == 8 && NC >= 2) ? (NC >= 4 ? 2 : 4) : ((GQE_DEC == DEC_BILINEAR && NC >= 4) ? 4 : 8))  // 8-wave d > 128 kernels (two rows per role) and the full-Bilinear d = 256 kernel: smaller groups keep them off the spill cliff

// ---- matrices staged in LDS (the intersection's Pre / Post at d <= 128) -----------------------------------------
// A d x d matrix every tile of a batch contracts with sits in L2, ~1 us away, and a contraction phase cannot start
// before its slab arrives.  The staged path takes that round trip off the tile's dependent chain: all threads of the
// workgroup request the NEXT matrix into registers (MR float4 each) while the current phase runs, and drop it into one
// LDS buffer between two phases.  What is staged is the matrix's OPERAND-ORDERED copy (gqe_dev.h, GQE_TILE_INDEX; the copy of M
// for M . x, the copy of M^T for M^T . x): global -> registers -> LDS is a straight copy of d * d floats, and the MFMA A operand
// of (row block, k-block) is one `ds_read_b128` per lane out of a contiguous kilobyte — no bank conflict, for either orientation
// (until round 4 the row-major matrix sat in a [d][d + 4] buffer: M^T . x read it with four `ds_read_b32` down a column).
template <int MR>  // MR = 1 (d = 64) or 4 (d = 128) float4 per thread; named members: the values have to stay in VGPRs
struct MatRegs {
  float4 v0, v1, v2, v3;
};

__device__ __forceinline__ float4 mat_ld(const float* __restrict__ M, int j) {
  return *reinterpret_cast<const float4*>(M + 4 * ((int)threadIdx.x + GQE_FWT * j));
}

template <int MR>
__device__ __forceinline__ void mat_issue(MatRegs<MR>& m, const float* __restrict__ M) {
  m.v0 = mat_ld(M, 0);
  if (MR > 1) {
    m.v1 = mat_ld(M, 1);
    m.v2 = mat_ld(M, 2);
    m.v3 = mat_ld(M, 3);
  }
}

__device__ __forceinline__ void mat_st(float* __restrict__ mb, int d, int DP, int j, const float4& v) {
  *reinterpret_cast<float4*>(mb + 4 * ((int)threadIdx.x + GQE_FWT * j)) = v;
}

template <int MR>
__device__ __forceinline__ void mat_commit(const MatRegs<MR>& m, float* __restrict__ mb, int d, int DP) {
  mat_st(mb, d, DP, 0, m.v0);
  if (MR > 1) {
    mat_st(mb, d, DP, 1, m.v1);
    mat_st(mb, d, DP, 2, m.v2);
    mat_st(mb, d, DP, 3, m.v3);
  }
}

template <int NC>
__device__ __forceinline__ float4 lds_a(const float* __restrict__ mb, int i0, int lq, int lk, int kb) {
  return *reinterpret_cast<const float4*>(mb + (((i0 >> 4) * (4 * NC) + kb) * 64 + lq + 16 * lk) * 4);
}

// Staged contractions (A from the LDS copy `mb`, row stride DP; FULL dims, 16-wave tiles: waves 0 .. d/16-1 own one
// output slab each).  The operands of k-block group g+1 are requested before the MFMAs of group g are issued
// (sched_barrier fences pin the source order: left alone, the scheduler puts each LDS read right in front of its use
// and every MFMA group then waits a full LDS round trip).

template <bool TRANS, int NC>
__device__ __forceinline__ void tile_matmul_staged(float* __restrict__ dst, const float* __restrict__ mb,
                                                   const float* __restrict__ src, int DP, int wave, int lane) {
  constexpr int KB = 4 * NC, G = 2, NG = KB / G;  // groups of two k-blocks: 8 MFMAs cover the next group's reads
  const int lq = lane & 15, lk = lane >> 4, i0 = wave * 16;
  if (i0 >= 64 * NC) return;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float4 a[2][G], b[2][G];
#pragma unroll
  for (int j = 0; j < G; ++j) {
    a[0][j] = lds_a<NC>(mb, i0, lq, lk, j);
    b[0][j] = *reinterpret_cast<const float4*>(src + lq * DP + j * 16 + 4 * lk);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) {
#pragma unroll
      for (int j = 0; j < G; ++j) {
        a[(g + 1) & 1][j] = lds_a<NC>(mb, i0, lq, lk, (g + 1) * G + j);
        b[(g + 1) & 1][j] = *reinterpret_cast<const float4*>(src + lq * DP + ((g + 1) * G + j) * 16 + 4 * lk);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < G; ++j) acc = mfma4(a[g & 1][j], b[g & 1][j], acc);
    __builtin_amdgcn_sched_barrier(0);
  }
  *reinterpret_cast<float4*>(dst + lq * DP + i0 + 4 * lk) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

template <int NC, int NB>
__device__ __forceinline__ void pre_intersect_staged(float* __restrict__ th, int* __restrict__ tmeta,
                                                     const float* __restrict__ mb, float* const (&te)[GQE_MAX_BRANCH],
                                                     int DP, int wave, int lane, int inter_min) {
  constexpr int KB = 4 * NC;
  const int lq = lane & 15, lk = lane >> 4, i0 = wave * 16;
  if (i0 >= 64 * NC) return;
  f32x4 acc[NB];
#pragma unroll
  for (int bi = 0; bi < NB; ++bi) acc[bi] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 a[2], b[2][NB];
  a[0] = lds_a<NC>(mb, i0, lq, lk, 0);
#pragma unroll
  for (int bi = 0; bi < NB; ++bi) b[0][bi] = *reinterpret_cast<const float4*>(te[bi] + lq * DP + 4 * lk);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (kb + 1 < KB) {
      a[(kb + 1) & 1] = lds_a<NC>(mb, i0, lq, lk, kb + 1);
#pragma unroll
      for (int bi = 0; bi < NB; ++bi)
        b[(kb + 1) & 1][bi] = *reinterpret_cast<const float4*>(te[bi] + lq * DP + (kb + 1) * 16 + 4 * lk);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) acc[bi] = mfma4(a[kb & 1], b[kb & 1][bi], acc[bi]);
    __builtin_amdgcn_sched_barrier(0);
  }
  float hv[4];
  int mv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float best = fmaxf(acc[0][r], 0.f);
    int meta = (acc[0][r] > 0.f) << 4;
#pragma unroll
    for (int bi = 1; bi < NB; ++bi) {
      const float z = acc[bi][r];
      const float v = fmaxf(z, 0.f);
      meta |= (z > 0.f) << (4 + bi);
      if (inter_min) {
        if (v < best) {  // strict: torch.min keeps the FIRST minimum
          best = v;
          meta = (meta & ~3) | bi;
        }
      } else {
        best += v;
      }
    }
    hv[r] = inter_min ? best : best / (float)NB;
    // bits 8 + b: branch b receives this element's gradient (relu'(z_b) and, for min, b is the arg-min) — the backward
    // contraction then builds its B operand with two VALU ops per element instead of mask_gz's five
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
      const int on = ((meta >> (4 + bi)) & 1) & (inter_min ? (int)((meta & 3) == bi) : 1);
      meta |= on << (8 + bi);
    }
    mv[r] = meta;
  }
  *reinterpret_cast<float4*>(th + lq * DP + i0 + 4 * lk) = make_float4(hv[0], hv[1], hv[2], hv[3]);
  *reinterpret_cast<int4*>(tmeta + lq * DP + i0 + 4 * lk) = make_int4(mv[0], mv[1], mv[2], mv[3]);
}

__device__ __forceinline__ float keep_if_bit(float g, int meta, int bit) {  // g if bit `bit` of meta is set, else +0
  return __int_as_float(__float_as_int(g) & __builtin_amdgcn_sbfe(meta, bit, 1));   // v_bfe_i32 (0 / -1) + v_and
}

// dst[q][i] = sum_k A[i][k] src[q][k]   (one source tile), A streamed from L2
// (one accumulator: the 8-wave d <= 128 kernels can afford the whole slab in flight here even at their 80-VGPR budget)
#define GQE_KG1 ((GQE_FW == 8 && NC == 2) ? 8 : GQE_KG)
template <bool TRANS, int NC, bool FULL>
__device__ __forceinline__ void tile_matmul(float* __restrict__ dst, const float* __restrict__ M,
                                            const float* __restrict__ src, int d, int DP, int wave, int lane) {
  constexpr int KB = 4 * NC, KG = KB < GQE_KG1 ? KB : GQE_KG1;
  const int lq = lane & 15, lk = lane >> 4;
  auto block = [&](const int i0) __attribute__((always_inline)) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (GQE_DBUF(KB, KG)) {
      // the slab does not fit the registers at once: two half-groups in flight, the next one requested as soon as the MFMAs that
      // read its registers are issued (load-all / contract-all per group left every wave of the tile waiting for L2 at the same
      // time, twice per contraction: Post at d = 256 took 8.1 us against a 3.4 us pipe bound)
      constexpr int H = KG / 2, NG = KB / H;
      float4 a[2][H];
      load_a_slab<H, KB, FULL>(a[0], M, d, i0, lq, lk, 0);
      load_a_slab<H, KB, FULL>(a[1], M, d, i0, lq, lk, H);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int kb = 0; kb < H; ++kb) {
          const float4 b = *reinterpret_cast<const float4*>(src + lq * DP + (g * H + kb) * 16 + 4 * lk);
          acc = mfma4(a[g & 1][kb], b, acc);
        }
        if (g + 2 < NG) {
          __builtin_amdgcn_sched_barrier(0);   // (not hoisted above the MFMAs: a third half-group in registers)
          load_a_slab<H, KB, FULL>(a[g & 1], M, d, i0, lq, lk, (g + 2) * H);
        }
      }
    } else {
#pragma unroll
      for (int g0 = 0; g0 < KB; g0 += KG) {
        float4 a[KG];
        load_a_slab<KG, KB, FULL>(a, M, d, i0, lq, lk, g0);
#pragma unroll
        for (int kb = 0; kb < KG; ++kb) {
          if (g0 + kb < KB) {
            const float4 b = *reinterpret_cast<const float4*>(src + lq * DP + (g0 + kb) * 16 + 4 * lk);
            acc = mfma4(a[kb], b, acc);
          }
        }
      }
    }
    *reinterpret_cast<float4*>(dst + lq * DP + i0 + 4 * lk) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  };
  if (FULL) {   // (the form the straight-line kernels were tuned with: their register allocation is sensitive to it)
    for (int i0 = wave * 16; i0 < d; i0 += GQE_FW * 16) block(i0);
  } else {      // compile-time trip count, row blocks past d skipped by a wave-uniform test: a loop over a run-time d made the
                // guarded 16-wave kernels spill hundreds of registers
#pragma unroll
    for (int i0b = 0; i0b < 64 * NC; i0b += GQE_FW * 16)
      if (i0b + wave * 16 < d) block(i0b + wave * 16);
  }
}

// SetIntersection forward for NB branches at once (decoders.py:288-299):
//   z_b = Pre . e_b ; h = agg_b relu(z_b) ; meta = (relu signs << 4) | first arg-min
// The Pre slab is loaded once and shared by the branches; relu / min / mean run on the accumulators.
template <int NC, int NB, bool FULL>
__device__ __forceinline__ void pre_intersect(float* __restrict__ th, int* __restrict__ tmeta,
                                              const float* __restrict__ P, float* const (&te)[GQE_MAX_BRANCH], int d,
                                              int DP, int wave, int lane, int inter_min) {
  constexpr int KB = 4 * NC, KG = KB < GQE_KG ? KB : GQE_KG;
  const int lq = lane & 15, lk = lane >> 4;
  auto block = [&](const int i0) __attribute__((always_inline)) {
    f32x4 acc[NB];
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) acc[bi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (GQE_DBUF(KB, KG)) {   // two half-groups in flight (tile_matmul)
      constexpr int H = KG / 2, NG = KB / H;
      float4 a[2][H];
      load_a_slab<H, KB, FULL>(a[0], P, d, i0, lq, lk, 0);
      load_a_slab<H, KB, FULL>(a[1], P, d, i0, lq, lk, H);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int kb = 0; kb < H; ++kb) {
#pragma unroll
          for (int bi = 0; bi < NB; ++bi) {
            const float4 b = *reinterpret_cast<const float4*>(te[bi] + lq * DP + (g * H + kb) * 16 + 4 * lk);
            acc[bi] = mfma4(a[g & 1][kb], b, acc[bi]);
          }
        }
        if (g + 2 < NG) {
          __builtin_amdgcn_sched_barrier(0);
          load_a_slab<H, KB, FULL>(a[g & 1], P, d, i0, lq, lk, (g + 2) * H);
        }
      }
    } else {
#pragma unroll
      for (int g0 = 0; g0 < KB; g0 += KG) {
        float4 a[KG];
        load_a_slab<KG, KB, FULL>(a, P, d, i0, lq, lk, g0);
#pragma unroll
        for (int kb = 0; kb < KG; ++kb) {
          if (g0 + kb < KB) {
#pragma unroll
            for (int bi = 0; bi < NB; ++bi) {
              const float4 b = *reinterpret_cast<const float4*>(te[bi] + lq * DP + (g0 + kb) * 16 + 4 * lk);
              acc[bi] = mfma4(a[kb], b, acc[bi]);
            }
          }
        }
      }
    }
    float hv[4];
    int mv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float best = fmaxf(acc[0][r], 0.f);
      int meta = (acc[0][r] > 0.f) << 4;
#pragma unroll
      for (int bi = 1; bi < NB; ++bi) {
        const float z = acc[bi][r];
        const float v = fmaxf(z, 0.f);
        meta |= (z > 0.f) << (4 + bi);
        if (inter_min) {
          if (v < best) {  // strict: torch.min keeps the FIRST minimum
            best = v;
            meta = (meta & ~3) | bi;
          }
        } else {
          best += v;
        }
      }
      hv[r] = inter_min ? best : best / (float)NB;
      // bits 8 + b: branch b receives this element's gradient (see pre_intersect_staged)
#pragma unroll
      for (int bi = 0; bi < NB; ++bi) {
        const int on = ((meta >> (4 + bi)) & 1) & (inter_min ? (int)((meta & 3) == bi) : 1);
        meta |= on << (8 + bi);
      }
      mv[r] = meta;
    }
    *reinterpret_cast<float4*>(th + lq * DP + i0 + 4 * lk) = make_float4(hv[0], hv[1], hv[2], hv[3]);
    *reinterpret_cast<int4*>(tmeta + lq * DP + i0 + 4 * lk) = make_int4(mv[0], mv[1], mv[2], mv[3]);
  };
  if (FULL) {   // (the form the straight-line kernels were tuned with: their register allocation is sensitive to it)
    for (int i0 = wave * 16; i0 < d; i0 += GQE_FW * 16) block(i0);
  } else {      // compile-time trip count, row blocks past d skipped by a wave-uniform test: a loop over a run-time d made the
                // guarded 16-wave kernels spill hundreds of registers
#pragma unroll
    for (int i0b = 0; i0b < 64 * NC; i0b += GQE_FW * 16)
      if (i0b + wave * 16 < d) block(i0b + wave * 16);
  }
}

__device__ __forceinline__ float mask_gz(float gh, int meta, int bi, int inter_min, float inv_n, bool mlp) {
  float g = inter_min ? (((meta & 3) == bi) ? gh : 0.f) : gh * inv_n;
  if (mlp && !((meta >> (4 + bi)) & 1)) g = 0.f;  // relu'(z) = [z > 0]
  return g;
}

template <int NC, int NB>
__device__ __forceinline__ void pre_intersect_bwd_staged(float* const (&te)[GQE_MAX_BRANCH], const float* __restrict__ mb,
                                                         const float* __restrict__ tgh, const int* __restrict__ tmeta,
                                                         int DP, int wave, int lane, int inter_min) {
  constexpr int KB = 4 * NC;
  const int lq = lane & 15, lk = lane >> 4, i0 = wave * 16;
  if (i0 >= 64 * NC) return;
  const float gsc = inter_min ? 1.f : 1.f / (float)NB;  // mean: every live branch gets g_h / n (mask_gz)
  f32x4 acc[NB];
#pragma unroll
  for (int bi = 0; bi < NB; ++bi) acc[bi] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 a[2], gh[2];
  int4 mt[2];
  a[0] = lds_a<NC>(mb, i0, lq, lk, 0);
  gh[0] = *reinterpret_cast<const float4*>(tgh + lq * DP + 4 * lk);
  mt[0] = *reinterpret_cast<const int4*>(tmeta + lq * DP + 4 * lk);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (kb + 1 < KB) {
      a[(kb + 1) & 1] = lds_a<NC>(mb, i0, lq, lk, kb + 1);
      gh[(kb + 1) & 1] = *reinterpret_cast<const float4*>(tgh + lq * DP + (kb + 1) * 16 + 4 * lk);
      mt[(kb + 1) & 1] = *reinterpret_cast<const int4*>(tmeta + lq * DP + (kb + 1) * 16 + 4 * lk);
    }
    __builtin_amdgcn_sched_barrier(0);
    const float4 g = make_float4(gh[kb & 1].x * gsc, gh[kb & 1].y * gsc, gh[kb & 1].z * gsc, gh[kb & 1].w * gsc);
    const int4 m = mt[kb & 1];
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
      const float4 b = make_float4(keep_if_bit(g.x, m.x, 8 + bi), keep_if_bit(g.y, m.y, 8 + bi), keep_if_bit(g.z, m.z, 8 + bi),
                                   keep_if_bit(g.w, m.w, 8 + bi));
      acc[bi] = mfma4(a[kb & 1], b, acc[bi]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int bi = 0; bi < NB; ++bi)
    *reinterpret_cast<float4*>(te[bi] + lq * DP + i0 + 4 * lk) = make_float4(acc[bi][0], acc[bi][1], acc[bi][2], acc[bi][3]);
}

// SetIntersection backward for NB branches at once: g_e_b = Pre^T . g_z_b with
// g_z_b = mask_b(meta) (.) g_h built on the fly as the B operand (never materialised in LDS).
template <int NC, int NB, bool FULL>
__device__ __forceinline__ void pre_intersect_bwd(float* const (&te)[GQE_MAX_BRANCH], const float* __restrict__ P,
                                                  const float* __restrict__ tgh, const int* __restrict__ tmeta, int d,
                                                  int DP, int wave, int lane, int inter_min) {
  constexpr int KB = 4 * NC, KG = KB < GQE_KG ? KB : GQE_KG;
  const int lq = lane & 15, lk = lane >> 4;
  const float gsc = inter_min ? 1.f : 1.f / (float)NB;  // mean: every live branch gets g_h / n (mask_gz)
  auto block = [&](const int i0) __attribute__((always_inline)) {
    f32x4 acc[NB];
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) acc[bi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // one k-block of the batched contraction: the B operand of branch b is g_h masked by bit 8 + b of the meta word
    // (pre_intersect): two VALU ops per element instead of mask_gz's five
    auto kblock = [&](const float4& av, const int kbi) __attribute__((always_inline)) {
      const float4 g4 = *reinterpret_cast<const float4*>(tgh + lq * DP + kbi * 16 + 4 * lk);
      const float4 gh = make_float4(g4.x * gsc, g4.y * gsc, g4.z * gsc, g4.w * gsc);
      const int4 mt = *reinterpret_cast<const int4*>(tmeta + lq * DP + kbi * 16 + 4 * lk);
#pragma unroll
      for (int bi = 0; bi < NB; ++bi) {
        const float4 b = make_float4(keep_if_bit(gh.x, mt.x, 8 + bi), keep_if_bit(gh.y, mt.y, 8 + bi), keep_if_bit(gh.z, mt.z, 8 + bi),
                                     keep_if_bit(gh.w, mt.w, 8 + bi));
        acc[bi] = mfma4(av, b, acc[bi]);
      }
    };
    if constexpr (GQE_DBUF(KB, KG)) {   // two half-groups in flight (tile_matmul)
      constexpr int H = KG / 2, NG = KB / H;
      float4 a[2][H];
      load_a_slab<H, KB, FULL>(a[0], P, d, i0, lq, lk, 0);
      load_a_slab<H, KB, FULL>(a[1], P, d, i0, lq, lk, H);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int kb = 0; kb < H; ++kb) kblock(a[g & 1][kb], g * H + kb);
        if (g + 2 < NG) {
          __builtin_amdgcn_sched_barrier(0);
          load_a_slab<H, KB, FULL>(a[g & 1], P, d, i0, lq, lk, (g + 2) * H);
        }
      }
    } else {
#pragma unroll
      for (int g0 = 0; g0 < KB; g0 += KG) {
        float4 a[KG];
        load_a_slab<KG, KB, FULL>(a, P, d, i0, lq, lk, g0);
#pragma unroll
        for (int kb = 0; kb < KG; ++kb)
          if (g0 + kb < KB) kblock(a[kb], g0 + kb);
      }
    }
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
      *reinterpret_cast<float4*>(te[bi] + lq * DP + i0 + 4 * lk) = make_float4(acc[bi][0], acc[bi][1], acc[bi][2], acc[bi][3]);
  };
  if (FULL) {   // (the form the straight-line kernels were tuned with: their register allocation is sensitive to it)
    for (int i0 = wave * 16; i0 < d; i0 += GQE_FW * 16) block(i0);
  } else {      // compile-time trip count, row blocks past d skipped by a wave-uniform test: a loop over a run-time d made the
                // guarded 16-wave kernels spill hundreds of registers
#pragma unroll
    for (int i0b = 0; i0b < 64 * NC; i0b += GQE_FW * 16)
      if (i0b + wave * 16 < d) block(i0b + wave * 16);
  }
}

// ------------------------------------------------------------------------------------------
// The RPW embedding rows one wave owns for one role (target / negative / anchor i), in registers for
// the whole kernel: gathered once, L2-normalised (encoders.py:41-43: x / ||x||, no eps), reused by the
// backward.  row < 0 marks a padding query (vector 0).
// ------------------------------------------------------------------------------------------
template <int NC>
struct RowSet {
  Vec<NC> x[RPW];
  float nrm[RPW];
  int row[RPW];  // bag modes: the row is bag `row` = ids[ptr[row] .. ptr[row + 1])
};

template <int NC, bool FULL>
__device__ __forceinline__ void rows_issue(RowSet<NC>& rs, const TileEnv& e, int64_t table, const int* s_rows) {
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = __builtin_amdgcn_readfirstlane(s_rows[e.wave * RPW + rr]);   // wave-uniform: scalar address arithmetic
    rs.row[rr] = row;
    // row-sharded: a position in the fetched-row buffer, or GQE_OWN_ROW + a row of this rank's own shard (read where it lives)
    const bool fetched = e.sharded && row < GQE_OWN_ROW;
    const int at = row < 0 ? 0 : (e.sharded && !fetched ? row - GQE_OWN_ROW : row);
    rs.x[rr] = gload<NC, FULL>((fetched ? e.rows : e.params + table) + (size_t)at * e.d, e.d, e.lane);
  }
}

// bag mode: raw vector = mean of the bag's word rows (nn.EmbeddingBag, mode 'mean'), then normalised like any other row
// by rows_finish.  A tile may gather up to five bag roles (target, negative, three anchors); done role after role each
// would put its own chain  s_idx -> ptr[row] -> ids[..] -> word rows  (three dependent trips to memory before the first row
// arrives, then one per group of rows) on the tile's critical path.  So the gather runs in three PHASES over all bag roles:
// (1) the spans (ptr[row], ptr[row + 1]) of every role — scalar loads: the row is wave-uniform; (2) the word ids of every
// role (one coalesced load each); (3) the word rows, BU at a time (all loads of a group in flight before the first add).
// The rows of the non-bag roles are requested between (1) and (2).  At the start of the kernel nothing else is live yet,
// so sixteen rows in flight (64 VGPRs at d = 256) do not raise the kernel's register peak.
struct BagSpan {
  int p0, len;   // wave-uniform
};

template <int NC>
__device__ __forceinline__ void bag_spans(BagSpan (&sp)[RPW], RowSet<NC>& rs, const TileEnv& e, const int* s_rows,
                                          const int32_t* __restrict__ ptr) {
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = __builtin_amdgcn_readfirstlane(s_rows[e.wave * RPW + rr]);
    rs.row[rr] = row;
    sp[rr].p0 = 0;
    sp[rr].len = 0;
    if (row >= 0) {
      sp[rr].p0 = ptr[row];
      sp[rr].len = ptr[row + 1] - sp[rr].p0;
    }
  }
}

__device__ __forceinline__ void bag_ids(int (&wid)[RPW], const BagSpan (&sp)[RPW], const TileEnv& e, const int32_t* __restrict__ ids) {
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) wid[rr] = (e.lane < min(sp[rr].len, 64)) ? ids[sp[rr].p0 + e.lane] : 0;
}

template <int NC, bool FULL>
__device__ __forceinline__ void bag_rows(RowSet<NC>& rs, const BagSpan (&sp)[RPW], const int (&wid0)[RPW], const TileEnv& e, int64_t table,
                                         const int32_t* __restrict__ ids) {
  // word rows in flight per wave (the adds keep the word order: the sum equals a one-by-one loop's); the full-Bilinear d = 256
  // kernel sits at its 128-VGPR limit: eight there
  constexpr int BU = (GQE_DEC == DEC_BILINEAR && NC >= 4) ? 8 : 16;
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    Vec<NC> acc = vzero<NC>();
    const int p0 = sp[rr].p0, len = sp[rr].len;
    for (int c0 = 0; c0 < len; c0 += 64) {
      const int m = min(64, len - c0);
      const int wid = (c0 == 0) ? wid0[rr] : ((e.lane < m) ? ids[p0 + c0 + e.lane] : 0);
      for (int k0 = 0; k0 < m; k0 += BU) {
        Vec<NC> v[BU];
#pragma unroll
        for (int u = 0; u < BU; ++u) {
          const int w = __builtin_amdgcn_readlane(wid, min(k0 + u, m - 1));
          v[u] = gload<NC, FULL>(e.params + table + (size_t)w * e.d, e.d, e.lane);
        }
#pragma unroll
        for (int u = 0; u < BU; ++u)
          if (k0 + u < m) VEC_OP(acc, acc.v[c] + v[u].v[c]);
      }
    }
    if (len > 0) {
      const float inv = gqe_rcp((float)len);
      VEC_OP(acc, acc.v[c] * inv);
    }
    rs.x[rr] = acc;
  }
}

// Evaluation against candidate lists (gqe_forward with n_candidates > 0): this kernel only computes the QUERY side,
// once per query, and leaves a record per query in the workspace; gqe_eval_score_kernel (gqe_kernels.hip) then streams
// the candidate rows.  Record of query q of a batch: ws[scratch_base + q * (d + 4) ...] = d floats v, then 3 scalars:
//   intersections      v = the intersected (and projected) query vector, s0 = max(|v|, eps)       score = cos(t, v)
//   bilinear-diag chain v = a (.) prod w                                                          score = t . v
//   TransE chain       v = a, s0 = max(|a|, eps), s1 = a . sum w, s2 = |sum w|^2                 score = cos(a, t + sum w)
template <int NC, bool FULL>
__device__ __forceinline__ void store_query_record(const TileEnv& e, int r, const Vec<NC>& v, float s0, float s1, float s2) {
  float* rec = e.ws + e.b.scratch_base + (size_t)(e.q0 + r) * (e.d + 4);
  gstore<NC, FULL>(rec, v, e.d, e.lane);
  if (e.lane == 0) {
    rec[e.d] = s0;
    rec[e.d + 1] = s1;
    rec[e.d + 2] = s2;
  }
}

template <int NC>
__device__ __forceinline__ void rows_finish(RowSet<NC>& rs) {
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    if (rs.row[rr] < 0) {
      rs.x[rr] = vzero<NC>();
      rs.nrm[rr] = 1.f;
    } else {
      const float n = gqe_sqrt(vdot<NC>(rs.x[rr], rs.x[rr]));
      // wave-uniform, but computed by the VALU: parked in an SGPR until the backward divides by it (up to ten of these norms
      // otherwise sit in VGPRs through every contraction phase — the 80-VGPR kernels spilled exactly them)
      rs.nrm[rr] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(n)));
      const float inv = gqe_rcp(n);
      VEC_OP(rs.x[rr], rs.x[rr].v[c] * inv);
    }
  }
}

// backward of x/||x||:  (g - xhat (xhat.g)) / ||x||.  The row gradient is written ONCE, coalesced, as
// contribution entry (role, query) and pushed onto the table row's list with one 4-byte atomic exchange;
// the optimiser pass (or gqe_materialize_grads) sums the lists.  role: 0 target, 1 negative, 2+i anchor i.
#define GQE_NO_PUSH (-2147483647 - 1)

template <int NC, bool FULL>
__device__ __forceinline__ void scatter_norm_bwd(const TileEnv& e, int64_t head_base, int role, int r, int row,
                                                 const Vec<NC>& xhat, float nrm, const Vec<NC>& g, int& old_head, int hot) {
  const float pg = vdot<NC>(xhat, g);
  const float inv = gqe_rcp(nrm);
  Vec<NC> gx;
  VEC_OP(gx, (g.v[c] - xhat.v[c] * pg) * inv);
  if (hot >= 0) {   // wave-uniform: a hot row (hub node) — no entry, no link
    hot_add<NC, FULL>(e, hot, gx);
    return;
  }
  // row-sharded: the contribution to a FETCHED row goes to the row's position in the send buffer (its owner links it); one to a
  // row of this rank's own shard is an entry like any other, in the block behind the send buffer (bag_shift; 0 otherwise)
  const bool fetched = e.sharded && row < GQE_OWN_ROW;   // wave-uniform
  const int64_t entry = fetched ? (int64_t)row : e.bag_shift + e.b.entry_base + (int64_t)role * e.b.B + (e.q0 + r);
  vstore_wt<NC, FULL>(e.wt, (fetched ? e.contrib : e.contrib_bag) + entry * e.d, gx, e.d, e.lane);
  // The returned previous head is only needed for next[entry]; that store is deferred to the end of the
  // kernel (push_links) so that the wave never stalls on the atomic's round trip.
  if (e.lane == 0 && !fetched) {
    if (e.sharded) row -= GQE_OWN_ROW;
    const int old = __hip_atomic_exchange(e.head + head_base + row, (int)entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    e.next[entry - e.max_entries] = (int)(head_base + row);  // entry -> list head, for the data-parallel exchange
    // 8-wave kernels (the many-tile throughput shape, up to three workgroups per CU, some held to 80 VGPRs): the link is stored
    // at once — the wave waits for the exchange's round trip, which the co-resident workgroups cover, and ten registers per
    // lane (one previous head per row and role) are not carried to the end of the kernel
    if (GQE_FW == 8)
      e.next[entry] = old;
    else
      old_head = old;
  }
}

// A bag's entry onto sub-list `sub` of a hot word row (gqe_dev.h): a position from the counter, the entry stored there; a full
// array (GQE_HOT_SUB_CAP: four times what the row's promotion sized it for) overflows onto a chain of ordinary link nodes.
__device__ __forceinline__ void hot_sub_store(const TileEnv& e, int sub, int pos, int entry, int node) {
  if (pos < GQE_HOT_SUB_CAP) {
    GQE_HOT_SUB_BUF(e.hot_sub)[(size_t)sub * GQE_HOT_SUB_CAP + pos] = entry;
  } else {
    e.link_contrib[node] = entry;
    e.next[e.max_entries + node] = __hip_atomic_exchange(GQE_HOT_SUB_OVF(e.hot_sub) + sub, e.max_entries + node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void hot_sub_push(const TileEnv& e, int sub, int entry, int node) {
  hot_sub_store(e, sub, __hip_atomic_fetch_add(GQE_HOT_SUB_CNT(e.hot_sub) + sub, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), entry, node);
}

// bag mode: one contribution (already divided by the bag length: EmbeddingBag mean backward) shared by every
// word row of the bag through link nodes: node -> (contribution entry, next).  The nodes of entry e are e * max_len + k
// (k = position of the word in the bag): no allocator, no counter to reset.  As for plain rows, the previous list heads
// the exchanges return are only needed for next[node]; those stores are deferred to the end of the kernel (push_bag_links)
// for the first 64 words of a bag, so that the wave does not stall on the atomics' round trip once per bag role.
template <int NC, bool FULL>
__device__ __forceinline__ void scatter_norm_bwd_bag(const TileEnv& e, int64_t head_base, int role, int r,
                                                     const int32_t* __restrict__ ptr, const int32_t* __restrict__ ids,
                                                     const Vec<NC>& xhat, float nrm, const Vec<NC>& g, int bag_slot,
                                                     int bag_index, int max_len, int& old_head, int& bag_len) {
  const int bi = __builtin_amdgcn_readfirstlane(bag_index);   // wave-uniform: the span comes from scalar loads and stays in SGPRs
  const int p0 = ptr[bi], len = ptr[bi + 1] - p0;             // re-read (cache hit) rather than carried in registers
  // (not in the full-Bilinear d = 256 kernel, nor in the 8-wave d <= 128 kernels that are held to 80 VGPRs for three workgroups
  // per CU: no register to spare)
  constexpr bool DEFER = !(GQE_DEC == DEC_BILINEAR && NC >= 4) && !(GQE_FW == 8 && NC == 2);
  bag_len = DEFER ? min(len, 64) : 0;   // lanes whose link push_links still owes
  const float pg = vdot<NC>(xhat, g);
  const float inv = gqe_rcp(nrm * (float)len);
  Vec<NC> gx;
  VEC_OP(gx, (g.v[c] - xhat.v[c] * pg) * inv);
  const int64_t entry = e.bag_shift + e.b.entry_base + (int64_t)role * e.b.B + (e.q0 + r);
  vstore_wt<NC, FULL>(e.wt, e.contrib_bag + entry * e.d, gx, e.d, e.lane);
  // entry -> "bag b of bag table s", for the data-parallel exchange: the importer re-expands the bag itself
  if (e.lane == 0) e.next[entry - e.max_entries] = GQE_BAG_CODE(bag_slot, bi);
  const int base = (int)entry * max_len;
  for (int c0 = 0; c0 < len; c0 += 64) {
    const int k = c0 + e.lane;
    int hs = -1;   // this lane's word row is hot and has no sub-lists: its slot
    if (k < len) {
      const int node = base + k;
      const int w = ids[p0 + k];
      if (e.hot_slot) hs = e.hot_slot[head_base + w];
      int sub = -1;   // a frequent word with sub-lists (gqe_dev.h): the one this bag's entry goes to
      if (hs >= 0 && e.hot_sub && GQE_HOT_SUB_LG1(hs)) {
        sub = GQE_HOT_SUB_BASE(hs) + ((int)entry & ((1 << (GQE_HOT_SUB_LG1(hs) - 1)) - 1));
        hs = -1;
      }
      if (sub >= 0) {
        if (DEFER && c0 == 0)
          old_head = GQE_HOT_SUB_TAG(sub);   // push_links takes the position: nothing here waits for an atomic's round trip
        else
          hot_sub_push(e, sub, (int)entry, node);
      } else if (hs < 0) {
        e.link_contrib[node] = (int)entry;
        const int old = __hip_atomic_exchange(e.head + head_base + w, e.max_entries + node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (DEFER && c0 == 0)
          old_head = old;   // lane k keeps word k's previous head until push_links
        else
          e.next[e.max_entries + node] = old;
      } else if (DEFER && c0 == 0) {
        old_head = GQE_NO_PUSH;   // nothing was linked for this word
      }
    }
    // frequent words without sub-lists: the bag's contribution is added to the word's accumulators, one wave-wide atomic row each
    if (e.hot_slot) {
      unsigned long long todo = __ballot(hs >= 0);
      while (todo) {
        const int l = __builtin_ctzll(todo);
        todo &= todo - 1;
#ifndef GQE_DEBUG_SKIP_BAG_HOT   // (timing experiments, WRONG results: what the hot words' atomic rows cost a bag launch)
        hot_add<NC, FULL>(e, __builtin_amdgcn_readlane(hs, l), gx);
#endif
      }
    }
  }
}

// Is a query's negative its target?  (model.py:118 draws a 1-chain negative from the whole mode.)  Then s+ and s- are the same number
// and every gradient of the query — both rows, the anchor, the relation parameters — cancels EXACTLY in the reference (+g and -g of
// identical arithmetic), while two separately rounded halves would leave noise that Adam's sign-like step turns into lr-sized moves
// of rows the reference does not move: such a query sends nothing.  Unsharded, the two index entries name the same row.  Row-sharded,
// an entry is a position in the fetched-row buffer (the same row fetched twice has two positions): the rows themselves are compared
// — bit-identical contents make s+ == s- and cancel the gradients just the same.
template <int NC>
__device__ __forceinline__ bool same_row(const TileEnv& e, int rt, int rn, const Vec<NC>& tp, const Vec<NC>& tn) {
  if (rt == rn) return true;
  if (!e.sharded) return false;
  bool eq = true;
#pragma unroll
  for (int c = 0; c < NC; ++c) eq = eq && (tp.v[c] == tn.v[c]);
  return __all(eq) != 0;
}

// Row-sharded mode: every fetched row owns a slot of the send buffer, and the owner links whatever arrives — a query
// whose hinge is inactive has to send zeros (the buffer still holds the previous step's contribution there).
template <int NC, bool FULL>
__device__ __forceinline__ void sharded_zero(const TileEnv& e, int bag, int row) {
  if (e.sharded && bag < 0 && row >= 0 && row < GQE_OWN_ROW) gstore<NC, FULL>(e.contrib + (size_t)row * e.d, vzero<NC>(), e.d, e.lane);
}

template <int NC, bool FULL>
__device__ __forceinline__ void scatter_row(const TileEnv& e, const GqeBagTable& bags, int bag, int64_t head_base, int role, int r,
                                            const RowSet<NC>& rs, int rr, const Vec<NC>& g, int& old_head, int& bag_len) {
  if (bag < 0)
    scatter_norm_bwd<NC, FULL>(e, head_base, role, r, rs.row[rr], rs.x[rr], rs.nrm[rr], g, old_head,
                         e.hot_slot ? __builtin_amdgcn_readlane(e.hotv, role * RPW + rr) : -1);
  else
    scatter_norm_bwd_bag<NC, FULL>(e, head_base, role, r, bags.ptr[bag], bags.ids[bag], rs.x[rr], rs.nrm[rr], g, bag, rs.row[rr], bags.max_len, old_head, bag_len);
}

// next[entry] = previous head, for every contribution this wave pushed; bag roles: lane k holds the previous head of the
// bag's k-th word row (node = entry * max_len + k)
__device__ __forceinline__ void push_links(const TileEnv& e, const int (&olds)[RPW][2 + GQE_MAX_BRANCH], const int (&blens)[RPW][2 + GQE_MAX_BRANCH],
                                           int max_len) {
  int any_bag = 0;   // wave-uniform
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
    for (int role = 0; role < 2 + GQE_MAX_BRANCH; ++role) any_bag |= blens[rr][role];
  if (any_bag == 0) {   // no bag role (every workload without EmbeddingBag modes): lane 0 alone, one EXEC region for all roles
    if (e.lane != 0) return;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
      for (int role = 0; role < 2 + GQE_MAX_BRANCH; ++role)
        if (olds[rr][role] != GQE_NO_PUSH)
          e.next[e.bag_shift + e.b.entry_base + (int64_t)role * e.b.B + (e.q0 + e.wave * RPW + rr)] = olds[rr][role];
    return;
  }
  // hot words with sub-lists (lanes that kept GQE_HOT_SUB_TAG(sub)): the positions of all roles first — their atomics travel
  // together, one round trip per wave at the end of the kernel — then the stores
  int pos[RPW][2 + GQE_MAX_BRANCH];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
    for (int role = 0; role < 2 + GQE_MAX_BRANCH; ++role) {
      const int o = olds[rr][role];
      pos[rr][role] = -1;
      if (e.hot_sub && blens[rr][role] > 0 && e.lane < blens[rr][role] && o <= GQE_HOT_SUB_TAG(0) && o != GQE_NO_PUSH)
        pos[rr][role] = __hip_atomic_fetch_add(GQE_HOT_SUB_CNT(e.hot_sub) + (GQE_HOT_SUB_TAG(0) - o), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
    for (int role = 0; role < 2 + GQE_MAX_BRANCH; ++role) {
      const int64_t q = e.b.entry_base + (int64_t)role * e.b.B + (e.q0 + e.wave * RPW + rr);
      const int o = olds[rr][role];
      if (blens[rr][role] > 0) {
        if (pos[rr][role] >= 0)
          hot_sub_store(e, GQE_HOT_SUB_TAG(0) - o, pos[rr][role], (int)(e.bag_shift + q), (int)(e.bag_shift + q) * max_len + e.lane);
        else if (e.lane < blens[rr][role] && o != GQE_NO_PUSH && (!e.hot_sub || o > GQE_HOT_SUB_TAG(0)))
          e.next[e.max_entries + (int)(e.bag_shift + q) * max_len + e.lane] = o;
      } else if (e.lane == 0 && o != GQE_NO_PUSH) {
        e.next[e.bag_shift + q] = o;
      }
    }
}

// Relation-vector gradients (bilinear-diag / TransE): every wave keeps its partial sums in registers
// (slot k = 2*branch + hop, 6 = final projection; chains use slot = hop) until the end of the kernel, when
// one LDS round (the tiles are dead by then) reduces the waves and wave k flushes vector k with one
// atomic row: two barriers per tile instead of two per vector.
#define GQE_VG_SLOTS 7

template <int NC>
struct VecGrads {
  Vec<NC> g[GQE_VG_SLOTS];
  int64_t param[GQE_VG_SLOTS];  // -1 = unused (workgroup-uniform)
};

// g[k] is only defined once param[k] >= 0: nothing is zero-initialised, so the slots cost no registers before the
// backward assigns them (at d = 256 the seven slots would otherwise hold 28 VGPRs through every MFMA phase).
template <int NC>
__device__ __forceinline__ void vecgrads_init(VecGrads<NC>& vg) {
#pragma unroll
  for (int k = 0; k < GQE_VG_SLOTS; ++k) vg.param[k] = -1;
}

template <int NC, bool FULL>
__device__ __forceinline__ void vecgrads_commit(const TileEnv& e, float* lds /* >= SLOTS * GQE_FW * 64 NC floats */, long long* s_param,
                                                const VecGrads<NC>& vg, float* red, float loss_part,
                                                const int (&olds)[RPW][2 + GQE_MAX_BRANCH], const int (&blens)[RPW][2 + GQE_MAX_BRANCH], int max_len) {
  __syncthreads();  // every wave is done with the tiles this staging area overlays
#pragma unroll
  for (int k = 0; k < GQE_VG_SLOTS; ++k) {
    if (vg.param[k] >= 0) lstore<NC>(lds + (size_t)(k * GQE_FW + e.wave) * (64 * NC), vg.g[k], e.lane);
    if (threadIdx.x == 0) s_param[k] = vg.param[k];
  }
  if (e.lane == 0) red[e.wave] = loss_part;  // the hinge partials ride on the same two barriers (red lies behind the staging area)
  __syncthreads();
  const long long param = e.wave < GQE_VG_SLOTS ? s_param[e.wave] : -1;
  Vec<NC> s = vzero<NC>();
  if (param >= 0) {
    s = lload<NC>(lds + (size_t)(e.wave * GQE_FW) * (64 * NC), e.lane);
#pragma unroll
    for (int w = 1; w < GQE_FW; ++w) {
      Vec<NC> t = lload<NC>(lds + (size_t)(e.wave * GQE_FW + w) * (64 * NC), e.lane);
      VEC_OP(s, s.v[c] + t.v[c]);
    }
  }
  // the list links (they wait for the heads the scatter's exchanges returned) go out BEFORE this wave's atomic row: behind
  // it their wait would also cover the atomics' own round trip (0.8 us on the waves that flush a vector)
  push_links(e, olds, blens, max_len);
  if (param >= 0) gatomic_add<NC, FULL>(e.grads + param, s, e.d, e.lane);
}

template <int NC, bool FULL>
__device__ __forceinline__ void tile_to_scratch(const TileEnv& e, int slot, const float* tile) {
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int r = e.wave * RPW + rr;
    vstore_wt<NC, FULL>(e.wt, scratch_row(e, slot, r), lload<NC>(tile + r * e.DP, e.lane), e.d, e.lane);
  }
}

// COMPACT: the d = 128 many-tile kernels (8 waves, MLP intersection on element-wise decoders) keep five LDS tiles instead of
// eight (q and g_q live in branch tiles that are dead by then) and are held to 80 VGPRs: THREE workgroups per CU instead of
// two — the launch is a throughput problem there (thousands of tiles), and a tile's phases are latency / barrier chains.
template <int DEC, bool MLP, int NC, bool FULL, int FW>
struct FusedShape {
  static constexpr bool COMPACT = FW == 8 && NC == 2 && FULL && MLP && DEC != DEC_BILINEAR;
  static constexpr int MIN_WAVES_PER_EU = COMPACT ? 6 : FW / 4;   // (FW / 4 is what the workgroup size implies anyway)
};

// LEAN (backward kernels of the straight-line dims): the launch has no EmbeddingBag role, no fetched rows (row-sharded mode) and
// no debug profile — the common training launch.  The uniform branches around those features, and the scalar registers their
// operands hold, go at compile time: the COMPACT kernel (thousands of tiles, issue-bound: every instruction of a tile is paid
// 4 600 times at B = 8192) drops from 13.9 k to 5.6 k instructions and from 119 to 30 spilled SGPRs, 226 -> 207 us.
// LEAN == 2 (16-wave kernels): the row-sharded step's launch — every row comes from the fetched buffer or this rank's own shard,
// no bag role, no profile, no riders.
template <int DEC, bool MLP, int NC, bool FULL, bool BWD, int FW, int LEAN = 0>
__global__ __launch_bounds__(GQE_FWT, (FusedShape<DEC, MLP, NC, FULL, FW>::MIN_WAVES_PER_EU)) void gqe_fused_kernel(const GqeDynPlan plan,
                                                                const GqeDevFormula* __restrict__ formulas,
                                                                const float* __restrict__ params,
                                                                float* __restrict__ grads, float* __restrict__ ws,
                                                                const int32_t* __restrict__ idx, int d_arg,
                                                                float* __restrict__ tile_loss, float* __restrict__ pos_out,
                                                                float* __restrict__ neg_out, int inter_min,
                                                                int32_t* __restrict__ head, int32_t* __restrict__ next,
                                                                float* __restrict__ contrib, const GqeBagTable bags,
                                                                int32_t* __restrict__ link_contrib,
                                                                int32_t* __restrict__ link_counter, int max_entries,
                                                                const float* __restrict__ fetched_arg,
                                                                float* __restrict__ contrib_bag, long long bag_shift,
                                                                const GqeHot hot, long long* __restrict__ prof_arg, const GqeSplitRide ride) {
  const float* __restrict__ fetched = LEAN == 1 ? nullptr : fetched_arg;   // (LEAN == 2: the row-sharded launch — fetched rows, and that is a compile-time fact)
#ifdef GQE_LEAN_PROF   // (debug builds: the profile stamps stay in the lean kernels — tools/probes/split_timeline.py on the production code path)
  long long* __restrict__ prof = prof_arg;
#else
  long long* __restrict__ prof = LEAN ? nullptr : prof_arg;
#endif
  static_assert(FW == GQE_FW, "FW only distinguishes the kernels of the per-GQE_FW translation units");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // gqe_train_step (gqe_split.h): the workgroups behind the tiles are riders — Adam over the table rows this step's batches do
  // not name.  They read and write nothing a tile touches, so they need no order against the tiles: they fill the wave slots
  // the tiles leave free (8-wave tiles) or the CUs whose tiles have finished (16-wave tiles) with HBM-bound work.
  // Grid: [ride.lead riders][plan.tiles tiles][the other riders]
  constexpr bool RIDE = BWD && FULL && !(LEAN && FusedShape<DEC, MLP, NC, FULL, FW>::COMPACT) && LEAN != 2;   // (lean COMPACT and row-sharded launches carry no riders: registers)
  const int tile_id = RIDE ? (int)blockIdx.x - ride.lead : (int)blockIdx.x;
  if (RIDE && (tile_id < 0 || tile_id >= plan.tiles)) {
    // (debug profile: start / end of the rider workgroup and the CU it ran on, in the rows behind the tiles')
    const size_t prow = (size_t)plan.tiles + (tile_id < 0 ? blockIdx.x : blockIdx.x - plan.tiles);
    if (prof && threadIdx.x == 0) {
      int hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      int xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      prof[prow * GQE_PROF_SLOTS + 0] = (long long)wall_clock64();
      prof[prow * GQE_PROF_SLOTS + 1] = ((long long)(xcc & 15) << 32) | (unsigned)hw;
    }
    split_rider<GQE_FW>(ride, 64 * NC, tile_id < 0 ? (int)blockIdx.x : (int)blockIdx.x - plan.tiles, reinterpret_cast<int*>(smem));
    if (prof && threadIdx.x == 0) prof[prow * GQE_PROF_SLOTS + 8] = (long long)wall_clock64();
    return;
  }
  // Debug profile (gqe_debug_profile; tools/kbench.py reads it): GQE_PROF_SLOTS wall_clock64 stamps per workgroup.
  // Slots 0 .. 15: thread 0 at the phase boundaries (0 start, 1 indices, 2 rows, 3 Pre, 4 Post / final, 5 scores, 6 Post^T,
  // 7 hops + scatter, 8 end, 9 .. 14 inside the intersection phases); slots 16 + 4 p + k: lane 0 of waves 0 / 4 / 8 / 12 at
  // point p of the backward (how far apart the waves of a tile finish a vector phase).  prof == NULL: a uniform branch.
#define GQE_STAMP(k)                                                                                              \
  do {                                                                                                            \
    if (prof && threadIdx.x == 0) prof[(size_t)tile_id * GQE_PROF_SLOTS + (k)] = (long long)wall_clock64(); \
  } while (0)
#define GQE_WSTAMP(p)                                                                                               \
  do {                                                                                                              \
    if (prof && (threadIdx.x & 255) == 0)                                                                           \
      prof[(size_t)tile_id * GQE_PROF_SLOTS + 16 + (p) * 4 + (threadIdx.x >> 8)] = (long long)wall_clock64();    \
  } while (0)
  GQE_STAMP(0);
#ifdef GQE_LEAN_PROF   // (where the tile ran: slot 63)
  if (prof && threadIdx.x == 0) {
    int hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    prof[(size_t)tile_id * GQE_PROF_SLOTS + 63] = ((long long)(xcc & 15) << 32) | (unsigned)hw;
  }
#endif
  const int d = FULL ? 64 * NC : d_arg;
  int bi = 0;  // which batch owns this tile: the plan is a kernel argument (SGPRs), 16 scalar compares
#pragma unroll
  for (int k = 1; k < GQE_LAUNCH_BATCHES; ++k) bi += (tile_id >= plan.tile_begin[k]) ? 1 : 0;
  const GqeDynBatch b = plan.b[bi];
  const GqeDevFormula* __restrict__ f = formulas + b.formula;
  TileEnv e;
  e.b = b;
  e.f = f;
  e.params = params;
  e.rows = fetched ? fetched : params;
  e.sharded = LEAN == 2 ? true : fetched != nullptr;
  e.contrib_bag = contrib_bag;
  e.bag_shift = bag_shift;
  e.grads = grads;
  e.ws = ws;
  e.head = head;
  e.next = next;
  e.contrib = contrib;
  e.link_contrib = link_contrib;
  e.link_counter = link_counter;
  e.max_entries = max_entries;
  e.d = d;
  e.wt = plan.pad[1] != 0;
  e.DP = 64 * NC + 4;   // tiles are padded to whole 64-float chunks: columns past d hold 0 (guarded kernels, see gqe_common.h)
  // the wave index as an SGPR: everything derived from it (the rows a wave owns, their bounds checks, row base addresses) is then
  // scalar arithmetic and scalar branches instead of 64-bit VALU address math and EXEC masks issued for all 64 lanes
  // (not in the full-Bilinear d = 256 kernel: the extra SGPRs spill into VGPR lanes it does not have)
  e.wave = (DEC == DEC_BILINEAR && NC >= 4) ? (int)(threadIdx.x >> 6) : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  e.lane = threadIdx.x & 63;
  e.q0 = (tile_id - b.tile_begin) * GQE_TQ;
  e.hot_slot = BWD ? hot.slot : nullptr;
  e.hot_acc = hot.acc;
  e.hot_sub = BWD ? hot.sub : nullptr;
  e.rep = 0;
  e.hotv = -1;
  if (BWD && hot.slot) {
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    e.rep = ((xcc & 7) * (GQE_HOT_REPS / 8) + (e.wave % (GQE_HOT_REPS / 8))) & (GQE_HOT_REPS - 1);
  }
  const int DP = e.DP, lane = e.lane, wave = e.wave;
  const int B = b.B;
  const bool has_neg = BWD;   // (== b.has_neg: the host sets it for exactly the backward launches — a compile-time fact here)
  const int n = b.n_anchors;  // == f->n_anchors, without the descriptor round trip in front of the index load
  // evaluation against candidate lists (forward only): index layout anchors[n][B] | cand_ptr[B+1] | cand_rows[..];
  // every query is scored against its own list, the query side being computed once (the reference re-encodes
  // and re-projects the anchors for every candidate, utils.py:50-60,78-88)
  const bool eval_mode = !BWD && b.n_candidates > 0;
  // ... except for full-Bilinear chains: t^T M_r1 .. M_rk is a projection of the CANDIDATE (decoders.py:142-147), so this
  // batch's tiles cover its candidates — row r of the tile is candidate q0 + r, whose anchor is its query's
  const bool expand = !BWD && DEC == DEC_BILINEAR && b.expand != 0;

  // ---- LDS carve: 7 float tiles [16][DP] + meta tile + red[8][d] + index block ----
  float* te[GQE_MAX_BRANCH];
  te[0] = smem;
  te[1] = te[0] + GQE_TQ * DP;
  te[2] = te[1] + GQE_TQ * DP;
  constexpr bool COMPACT = FusedShape<DEC, MLP, NC, FULL, FW>::COMPACT;
  float* tt = COMPACT ? nullptr : te[2] + GQE_TQ * DP;          // temp / ping-pong (full Bilinear only)
  float* tacc = (COMPACT ? te[2] : tt) + GQE_TQ * DP;           // h (MLP) or q (simple); later g_h
  float* tq = COMPACT ? te[1] : tacc + GQE_TQ * DP;             // q     (COMPACT: the branch tiles are dead between Pre and Pre^T;
  float* tg = COMPACT ? te[2] : tq + GQE_TQ * DP;               // g_q    te[0] stays free for the final projection)
  int* tmeta = reinterpret_cast<int*>((COMPACT ? tacc : tg) + GQE_TQ * DP);
  float* red = reinterpret_cast<float*>(tmeta + GQE_TQ * DP);
  int* s_idx = reinterpret_cast<int*>(red + (COMPACT ? 64 : GQE_FW * d));  // [5][16]: target, negative, anchor 0..2
  // staged matrices (see mat_issue): the MLP intersection's Pre / Post, d <= 128, 16-wave tiles
  constexpr bool STAGE = MLP && DEC != DEC_BILINEAR && FULL && NC <= 2 && FW == 16;
  constexpr int MR = STAGE ? NC * NC : 1;
  float* mbuf = reinterpret_cast<float*>(s_idx + 5 * GQE_TQ);  // [d][DP], only carved for STAGE kernels
  if (!FULL) {
    // guarded kernels: the columns past d of every tile have to read 0 (element-wise phases and row dot products run over whole
    // 64-float chunks without a guard); contraction epilogues only ever write columns < d, row stores write zeros there.  The
    // last chunk of every row is cleared once — its columns below d are rewritten by whoever fills the tile.
    float* const tiles[8] = {te[0], te[1], te[2], tt, tacc, tq, tg, reinterpret_cast<float*>(tmeta)};
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) tiles[k][((int)(threadIdx.x >> 6) * RPW + rr) * e.DP + 64 * (NC - 1) + e.lane] = 0.f;
  }
  MatRegs<MR> mr;
  // The descriptor fields the gathers need — requested here, pinned (GQE_PIN) behind the index load below, so that
  // their scalar round trip runs next to that load instead of after the barrier (left alone, the compiler sinks every
  // descriptor read to its first use: one more dependent trip to L2 per phase).
  // 16-wave tiles only: the 8-wave kernels of the guarded d in (128, 256) variants sit at 256 VGPRs with spills, and
  // there the early copies changed the allocation into one that loses lanes >= 16 of a relation gradient
  // (tests/test_gpu_parity.py::test_eight_wave_workgroups_vs_oracle d = 144) — they keep reading the fields where used.
#define GQE_PIN(x) asm volatile("" : "+s"(x))
  constexpr bool EARLY = FW == 16 && NC <= 2 && FULL;  // (the d = 256 and the guarded d % 64 != 0 kernels sit at the 128-VGPR limit:
                                                       // a few more live registers spill there — and a spilling d = 96 kernel faulted)
#define GQE_DSC(early, field) (EARLY ? (early) : (field))
#define GQE_TBAG (LEAN ? -1 : f->target_bag)      // (LEAN: no bag role in the launch)
#define GQE_ABAG(i) (LEAN ? -1 : f->anchor_bag[i])
  int qtype = f->qtype;
  int64_t t_table = 0;
  int tbag = GQE_TBAG;
  int64_t a_table[GQE_MAX_BRANCH] = {0, 0, 0};
  int a_bag[GQE_MAX_BRANCH] = {-1, -1, -1};
  int64_t t_head = 0, a_head[GQE_MAX_BRANCH] = {0, 0, 0};   // list-head bases: where a row's hot slot is looked up (backward only)
  if (EARLY) {
    t_table = f->target_table;
    if (BWD) t_head = f->target_head;
#pragma unroll
    for (int i = 0; i < GQE_MAX_BRANCH; ++i) {
      a_table[i] = f->anchor_table[i];
      a_bag[i] = GQE_ABAG(i);
      if (BWD) a_head[i] = f->anchor_head[i];
    }
  }

  // ---- the tile's table rows: one coalesced read, then every gather is issued at once ----
  if (threadIdx.x < 5 * GQE_TQ) {
    const int role = threadIdx.x / GQE_TQ, r = threadIdx.x % GQE_TQ;
    const int q = e.q0 + r;
    int v = -1;
    const bool used = (role == 0 && (!eval_mode || expand)) || (role == 1 && has_neg) || (role >= 2 && role - 2 < n);
    if (expand) {
      if (used && q < b.n_candidates) {
        const int32_t* __restrict__ lists = idx + b.idx_offset + (size_t)n * B;   // cand_ptr[B + 1] | cand_rows[n_candidates]
        v = role == 0 ? lists[B + 1 + q] : idx[b.idx_offset + reinterpret_cast<const int32_t*>(ws)[b.scratch_base + q]];
      }
    } else if (used && q < B) {
      const int src = (role == 0) ? 0 : (role == 1) ? 1 : (has_neg ? role : (eval_mode ? role - 2 : role - 1));
      v = idx[b.idx_offset + (size_t)src * B + q];
    }
    s_idx[threadIdx.x] = v;
  }
  if (EARLY) {
    GQE_PIN(qtype);
    GQE_PIN(t_table);
    GQE_PIN(tbag);
    if (BWD) GQE_PIN(t_head);
#pragma unroll
    for (int i = 0; i < GQE_MAX_BRANCH; ++i) {
      GQE_PIN(a_table[i]);
      GQE_PIN(a_bag[i]);
      if (BWD) GQE_PIN(a_head[i]);
    }
  }
#undef GQE_PIN
  const bool stage = STAGE && qtype > 2;
  __syncthreads();
  GQE_STAMP(1);
  if (BWD && e.hot_slot && !e.sharded && lane < 5 * RPW) {
    // hot slots of the plain rows this wave owns: lane role * RPW + rr (bag roles: their WORD rows are looked up where they are linked)
    const int role = lane / RPW;
    const int row = s_idx[role * GQE_TQ + wave * RPW + (lane - role * RPW)];
    int64_t hb = GQE_DSC(t_head, f->target_head);
    int bg = tbag;
#pragma unroll
    for (int i = 0; i < GQE_MAX_BRANCH; ++i)
      if (role == 2 + i) {
        hb = GQE_DSC(a_head[i], f->anchor_head[i]);
        bg = GQE_DSC(a_bag[i], GQE_ABAG(i));
      }
    if (row >= 0 && bg < 0) e.hotv = e.hot_slot[hb + row];
  }
  // the relation vectors of an intersection tile (<= 2 per branch + the final projection): requested here, in front of
  // the rows, instead of one dependent L2 round trip per branch in the forward and again in the backward
  // (16-wave tiles; the 8-wave tiles keep two rows per role in registers and load the vectors where they use them)
  constexpr bool PREW = FW == 16 && NC <= 2 && FULL;  // d = 256 / guarded d: the vectors would cost registers a kernel at the 128-VGPR limit does not have
  Vec<NC> W0[GQE_MAX_BRANCH], W1[GQE_MAX_BRANCH], WF;
  if (PREW && DEC != DEC_BILINEAR && qtype > 2) {
#pragma unroll
    for (int i = 0; i < GQE_MAX_BRANCH; ++i) {
      if (i < n) {
        W0[i] = gload<NC, FULL>(params + f->hop_param[i][0], d, lane);
        W1[i] = (f->n_hops[i] > 1) ? gload<NC, FULL>(params + f->hop_param[i][1], d, lane) : W0[i];
      }
    }
    if (f->n_final) WF = gload<NC, FULL>(params + f->final_param, d, lane);
  }
  RowSet<NC> RA[GQE_MAX_BRANCH], RT, RN;
  // (the full-Bilinear d = 256 kernel has two VGPRs to spare: there the bag roles are gathered one after the other)
  constexpr bool PHASED = !(DEC == DEC_BILINEAR && NC >= 4);
  if (PHASED) {
    // bag roles: spans of all of them, then (after the plain rows are requested) their word ids, then their word rows
    BagSpan st[RPW], sn[RPW], sa[GQE_MAX_BRANCH][RPW];
    int wt[RPW], wn[RPW], wa[GQE_MAX_BRANCH][RPW];
    int abag[GQE_MAX_BRANCH] = {-1, -1, -1};
    const int64_t tt = GQE_DSC(t_table, f->target_table);
    if (tbag >= 0) {
      bag_spans<NC>(st, RT, e, s_idx, bags.ptr[tbag]);
      if (has_neg) bag_spans<NC>(sn, RN, e, s_idx + GQE_TQ, bags.ptr[tbag]);
    }
#pragma unroll
    for (int i = 0; i < GQE_MAX_BRANCH; ++i) {
      if (i < n) {
        abag[i] = GQE_DSC(a_bag[i], GQE_ABAG(i));
        if (abag[i] >= 0) bag_spans<NC>(sa[i], RA[i], e, s_idx + (2 + i) * GQE_TQ, bags.ptr[abag[i]]);
      }
    }
    if (tbag < 0) {
      rows_issue<NC, FULL>(RT, e, tt, s_idx);
      if (has_neg) rows_issue<NC, FULL>(RN, e, tt, s_idx + GQE_TQ);
    }
#pragma unroll
    for (int i = 0; i < GQE_MAX_BRANCH; ++i)
      if (i < n && abag[i] < 0) rows_issue<NC, FULL>(RA[i], e, GQE_DSC(a_table[i], f->anchor_table[i]), s_idx + (2 + i) * GQE_TQ);
    if (tbag >= 0) {
      bag_ids(wt, st, e, bags.ids[tbag]);
      if (has_neg) bag_ids(wn, sn, e, bags.ids[tbag]);
    }
#pragma unroll
    for (int i = 0; i < GQE_MAX_BRANCH; ++i)
      if (i < n && abag[i] >= 0) bag_ids(wa[i], sa[i], e, bags.ids[abag[i]]);
    if (tbag >= 0) {
      bag_rows<NC, FULL>(RT, st, wt, e, tt, bags.ids[tbag]);
      if (has_neg) bag_rows<NC, FULL>(RN, sn, wn, e, tt, bags.ids[tbag]);
    }
#pragma unroll
    for (int i = 0; i < GQE_MAX_BRANCH; ++i)
      if (i < n && abag[i] >= 0) bag_rows<NC, FULL>(RA[i], sa[i], wa[i], e, GQE_DSC(a_table[i], f->anchor_table[i]), bags.ids[abag[i]]);
  } else {
    auto gather = [&](RowSet<NC>& rs, int64_t table, const int* s_rows, int bag) {
      if (bag < 0) {
        rows_issue<NC, FULL>(rs, e, table, s_rows);
      } else {
        BagSpan sp[RPW];
        int wid[RPW];
        bag_spans<NC>(sp, rs, e, s_rows, bags.ptr[bag]);
        bag_ids(wid, sp, e, bags.ids[bag]);
        bag_rows<NC, FULL>(rs, sp, wid, e, table, bags.ids[bag]);
      }
    };
    gather(RT, GQE_DSC(t_table, f->target_table), s_idx, tbag);
    if (has_neg) gather(RN, GQE_DSC(t_table, f->target_table), s_idx + GQE_TQ, tbag);
#pragma unroll
    for (int i = 0; i < GQE_MAX_BRANCH; ++i)
      if (i < n) gather(RA[i], GQE_DSC(a_table[i], f->anchor_table[i]), s_idx + (2 + i) * GQE_TQ, GQE_DSC(a_bag[i], GQE_ABAG(i)));
  }
  rows_finish<NC>(RT);
  if (has_neg) {
    rows_finish<NC>(RN);
  } else {
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      RN.x[rr] = vzero<NC>();
      RN.nrm[rr] = 1.f;
      RN.row[rr] = -1;
    }
  }
#pragma unroll
  for (int i = 0; i < GQE_MAX_BRANCH; ++i)
    if (i < n) rows_finish<NC>(RA[i]);

  // Pre is requested once the rows are in (192 tiles pulling the same 64 KB out of the same L2 lines next to the
  // gathers delayed those by ~1.5 us); it lands while the branch vectors are built
  if (stage) mat_issue<MR>(mr, GQE_TILED(false, pre_tile));
  GQE_STAMP(9);
  const bool is_chain = qtype <= 2;
  const float gscale = b.grad_scale;  // loss_weight / B
  float loss_part = 0.f;
  VecGrads<NC> vg;  // relation-vector gradient partials of this wave (DEC != bilinear)
  vecgrads_init<NC>(vg);
  int olds[RPW][2 + GQE_MAX_BRANCH];  // previous list heads returned by this wave's pushes
  int blens[RPW][2 + GQE_MAX_BRANCH]; // bag roles: lanes that hold one (the bag's first <= 64 words); 0: a plain row
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
    for (int role = 0; role < 2 + GQE_MAX_BRANCH; ++role) {
      olds[rr][role] = GQE_NO_PUSH;
      blens[rr][role] = 0;
    }
  GQE_STAMP(2);

  if (is_chain) {
    // =====================================================================================
    // chains: score(target, anchor) with the relations applied on the TARGET side
    // =====================================================================================
    const int K = f->n_hops[0];
    if (DEC != DEC_BILINEAR) {
      // ---- bilinear-diag / TransE: everything stays in registers, one wave per query ----
      Vec<NC> w[GQE_MAX_HOPS];
      Vec<NC> wcomb;  // diag: prod_h w_h ; transe: sum_h w_h
      VEC_OP(wcomb, (DEC == DEC_DIAG) ? 1.f : 0.f);
#pragma unroll
      for (int h = 0; h < GQE_MAX_HOPS; ++h) {
        if (h < K) {
          w[h] = gload<NC, FULL>(params + f->hop_param[0][h], d, lane);
          VEC_OP(wcomb, (DEC == DEC_DIAG) ? wcomb.v[c] * w[h].v[c] : wcomb.v[c] + w[h].v[c]);
        } else {
          VEC_OP(w[h], (DEC == DEC_DIAG) ? 1.f : 0.f);
        }
      }
      Vec<NC> gw_acc = vzero<NC>();  // diag: sum_rows (cp t+ + cn t-) (.) a ; transe: sum_rows (gu+ + gu-)
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int q = e.q0 + wave * RPW + rr;
        if (q >= B) continue;  // wave-uniform
        const Vec<NC>& a = RA[0].x[rr];
        const Vec<NC>& tp = RT.x[rr];
        const Vec<NC>& tn = RN.x[rr];
        float sp, sn = 0.f;
        float nap = 1.f, nup = 1.f, nun = 1.f;  // transe cosine norms
        Vec<NC> up, un;
        if (DEC == DEC_DIAG) {
          // decoders.py:228-233: acts = t * w1 * .. * wk ; score = sum(acts * a)   (raw dot)
          up = tp;
          un = tn;
#pragma unroll
          for (int h = 0; h < GQE_MAX_HOPS; ++h) {
            if (h < K) {
              VEC_OP(up, up.v[c] * w[h].v[c]);
              VEC_OP(un, un.v[c] * w[h].v[c]);
            }
          }
          sp = vdot<NC>(up, a);
          if (has_neg) sn = vdot<NC>(un, a);
        } else {
          // decoders.py:200-205: u = t + sum w ; score = cos(a, u)
          VEC_OP(up, tp.v[c] + wcomb.v[c]);
          VEC_OP(un, tn.v[c] + wcomb.v[c]);
          nap = fmaxf(gqe_sqrt(vdot<NC>(a, a)), COS_EPS);
          nup = fmaxf(gqe_sqrt(vdot<NC>(up, up)), COS_EPS);
          sp = vdot<NC>(a, up) * gqe_rcp(nap * nup);
          if (has_neg) {
            nun = fmaxf(gqe_sqrt(vdot<NC>(un, un)), COS_EPS);
            sn = vdot<NC>(a, un) * gqe_rcp(nap * nun);
          }
        }
        if (eval_mode) {  // the query side of this row, for the candidate-scoring kernel
          Vec<NC> v = a;
          if (DEC == DEC_DIAG) VEC_OP(v, a.v[c] * wcomb.v[c]);
          const float aw = (DEC == DEC_DIAG) ? 0.f : vdot<NC>(a, wcomb), ww = (DEC == DEC_DIAG) ? 0.f : vdot<NC>(wcomb, wcomb);
          store_query_record<NC, FULL>(e, wave * RPW + rr, v, nap, aw, ww);
          continue;
        }
        if (lane == 0) {
          if (pos_out) pos_out[b.out_offset + q] = sp;
          if (neg_out && has_neg) neg_out[b.out_offset + q] = sn;
        }
        if (!BWD) continue;
        const float hinge = b.margin - (sp - sn);
        if (hinge > 0.f) loss_part += hinge;
        if (hinge > 0.f && !same_row<NC>(e, RT.row[rr], RN.row[rr], tp, tn)) {   // (target == negative: nothing to send, see same_row)
          const float cp = -gscale, cn = gscale;
          Vec<NC> ga, gtp, gtn;
          if (DEC == DEC_DIAG) {
            Vec<NC> tmix;
            VEC_OP(tmix, cp * tp.v[c] + cn * tn.v[c]);
            VEC_OP(ga, tmix.v[c] * wcomb.v[c]);
            VEC_OP(gtp, cp * wcomb.v[c] * a.v[c]);
            VEC_OP(gtn, cn * wcomb.v[c] * a.v[c]);
            VEC_OP(gw_acc, gw_acc.v[c] + tmix.v[c] * a.v[c]);
          } else {
            // d cos(a,u)/da = u/(na nu) - s a/na^2 ; d/du = a/(na nu) - s u/nu^2
            const float ipp = gqe_rcp(nap * nup), ipn = gqe_rcp(nap * nun);
            const float iup = sp * gqe_rcp(nup * nup), iun = sn * gqe_rcp(nun * nun), iaa = gqe_rcp(nap * nap);
            VEC_OP(gtp, cp * (a.v[c] * ipp - up.v[c] * iup));
            VEC_OP(gtn, cn * (a.v[c] * ipn - un.v[c] * iun));
            VEC_OP(ga, cp * (up.v[c] * ipp - sp * a.v[c] * iaa) + cn * (un.v[c] * ipn - sn * a.v[c] * iaa));
            VEC_OP(gw_acc, gw_acc.v[c] + gtp.v[c] + gtn.v[c]);
          }
          scatter_row<NC, FULL>(e, bags, GQE_TBAG, f->target_head, 0, wave * RPW + rr, RT, rr, gtp, olds[rr][0], blens[rr][0]);
          scatter_row<NC, FULL>(e, bags, GQE_TBAG, f->target_head, 1, wave * RPW + rr, RN, rr, gtn, olds[rr][1], blens[rr][1]);
          scatter_row<NC, FULL>(e, bags, GQE_ABAG(0), f->anchor_head[0], 2, wave * RPW + rr, RA[0], rr, ga, olds[rr][2], blens[rr][2]);
        } else {
          sharded_zero<NC, FULL>(e, GQE_TBAG, RT.row[rr]);
          sharded_zero<NC, FULL>(e, GQE_TBAG, RN.row[rr]);
          sharded_zero<NC, FULL>(e, GQE_ABAG(0), RA[0].row[rr]);
        }
      }
      if (BWD) {
#pragma unroll
        for (int h = 0; h < GQE_MAX_HOPS; ++h) {
          if (h < K) {
            Vec<NC> part = gw_acc;
            if (DEC == DEC_DIAG) {
              // d/dw_h = sum_rows (..) (.) prod_{j != h} w_j   (w[j >= K] = 1)
#pragma unroll
              for (int j = 0; j < GQE_MAX_HOPS; ++j)
                if (j != h) VEC_OP(part, part.v[c] * w[j].v[c]);
            }
            vg.g[h] = part;
            vg.param[h] = f->hop_param[0][h];
          }
        }
      }
    } else {
      // ---- full Bilinear chain (decoders.py:142-147): act = t^T M1..Mk ; s = cos(act, a) ----
      // te[0]/te[1]: u+ ping-pong, te[2]/tt: u- ping-pong; the anchor stays in registers
      float* cur[2] = {te[0], te[2]};
      float* alt[2] = {te[1], tt};
      const int nside = has_neg ? 2 : 1;
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
        lstore<NC>(cur[0] + r * DP, RT.x[rr], lane);
        if (has_neg) lstore<NC>(cur[1] + r * DP, RN.x[rr], lane);
        if (BWD) {
          vstore_wt<NC, FULL>(e.wt, scratch_row(e, f->slot_act[0][0], r), RT.x[rr], d, lane);
          vstore_wt<NC, FULL>(e.wt, scratch_row(e, f->slot_act[1][0], r), RN.x[rr], d, lane);
        }
      }
      for (int h = 0; h < K; ++h) {
        __syncthreads();
        for (int s = 0; s < nside; ++s)
          tile_matmul<true, NC, FULL>(alt[s], GQE_TILED(true, hop_tile[0][h]), cur[s], d, DP, wave, lane);
        __syncthreads();
        for (int s = 0; s < nside; ++s) {
          float* tmp = cur[s];
          cur[s] = alt[s];
          alt[s] = tmp;
        }
        if (BWD && h + 1 < K)
          for (int s = 0; s < nside; ++s) tile_to_scratch<NC, FULL>(e, f->slot_act[s][h + 1], cur[s]);
      }
      // scores + gradient seeds; g_u overwrites u in place
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
        const int q = e.q0 + r;
        const Vec<NC>& a = RA[0].x[rr];
        const float nac = fmaxf(gqe_sqrt(vdot<NC>(a, a)), COS_EPS);
        Vec<NC> u[2];
        float su[2] = {0.f, 0.f}, nu[2] = {1.f, 1.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s < nside) {
            u[s] = lload<NC>(cur[s] + r * DP, lane);
            nu[s] = fmaxf(gqe_sqrt(vdot<NC>(u[s], u[s])), COS_EPS);
            su[s] = vdot<NC>(u[s], a) * gqe_rcp(nu[s] * nac);
          } else {
            u[s] = vzero<NC>();
          }
        }
        if (q < (expand ? b.n_candidates : B) && lane == 0) {
          if (pos_out) pos_out[b.out_offset + q] = su[0];
          if (neg_out && has_neg) neg_out[b.out_offset + q] = su[1];
        }
        if (!BWD) continue;
        const float hinge = b.margin - (su[0] - su[1]);
        if ((q < B) && hinge > 0.f) loss_part += hinge;
        const bool act = (q < B) && hinge > 0.f && !same_row<NC>(e, RT.row[rr], RN.row[rr], RT.x[rr], RN.x[rr]);   // (target == negative: see same_row)
        const float cf[2] = {act ? -gscale : 0.f, act ? gscale : 0.f};
        Vec<NC> ga = vzero<NC>();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const float iun = gqe_rcp(nu[s] * nac), iuu = su[s] * gqe_rcp(nu[s] * nu[s]), iaa = su[s] * gqe_rcp(nac * nac);
          Vec<NC> gu;
          VEC_OP(gu, cf[s] * (a.v[c] * iun - u[s].v[c] * iuu));
          VEC_OP(ga, ga.v[c] + cf[s] * (u[s].v[c] * iun - a.v[c] * iaa));
          lstore<NC>(cur[s] + r * DP, gu, lane);
        }
        if (act) scatter_row<NC, FULL>(e, bags, GQE_ABAG(0), f->anchor_head[0], 2, wave * RPW + rr, RA[0], rr, ga, olds[rr][2], blens[rr][2]);
        else if (q < B) sharded_zero<NC, FULL>(e, GQE_ABAG(0), RA[0].row[rr]);
      }
      if (BWD) {
        // back through the hops: act_{h+1} = act_h M_h  =>  g_act_h = g_act_{h+1} M_h^T (= M . g per row),
        // dM_h += act_h^T g_act_{h+1}  (deferred: pair (slot_act[s][h], slot_gact[s][h]))
        for (int h = K - 1; h >= 0; --h) {
          for (int s = 0; s < 2; ++s) tile_to_scratch<NC, FULL>(e, f->slot_gact[s][h], cur[s]);
          __syncthreads();
          for (int s = 0; s < 2; ++s)
            tile_matmul<false, NC, FULL>(alt[s], GQE_TILED(false, hop_tile[0][h]), cur[s], d, DP, wave, lane);
          __syncthreads();
          for (int s = 0; s < 2; ++s) {
            float* tmp = cur[s];
            cur[s] = alt[s];
            alt[s] = tmp;
          }
        }
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
          const int r = wave * RPW + rr;
          if (e.q0 + r >= B) continue;
          scatter_row<NC, FULL>(e, bags, GQE_TBAG, f->target_head, 0, r, RT, rr, lload<NC>(cur[0] + r * DP, lane), olds[rr][0], blens[rr][0]);
          scatter_row<NC, FULL>(e, bags, GQE_TBAG, f->target_head, 1, r, RN, rr, lload<NC>(cur[1] + r * DP, lane), olds[rr][1], blens[rr][1]);
        }
      }
    }
  } else {
    // =====================================================================================
    // intersections: q = I( Proj(a_1), Proj(a_2)[, Proj(a_3)] ) [-> Proj]; s = cos(t, q)
    // =====================================================================================
    const float inv_n = 1.f / (float)n;
    // ---- branch vectors e_i -> te[i] ----
#pragma unroll
    for (int i = 0; i < GQE_MAX_BRANCH; ++i) {
      if (i >= n) continue;
      const int nh = f->n_hops[i];
      if (DEC == DEC_BILINEAR) {
        // stage so that the ping-pong te[i] <-> tt ends in te[i]
        float* src = (nh & 1) ? tt : te[i];
        float* dst = (nh & 1) ? te[i] : tt;
        if (i > 0) __syncthreads();  // tt is shared by the branches
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
          const int r = wave * RPW + rr;
          lstore<NC>(src + r * DP, RA[i].x[rr], lane);
          if (BWD) vstore_wt<NC, FULL>(e.wt, scratch_row(e, f->slot_x[i][0], r), RA[i].x[rr], d, lane);
        }
        for (int h = 0; h < nh; ++h) {
          __syncthreads();
          tile_matmul<false, NC, FULL>(dst, GQE_TILED(false, hop_tile[i][h]), src, d, DP, wave, lane);
          __syncthreads();
          float* tmp = src;
          src = dst;
          dst = tmp;
          if (BWD && h + 1 < nh) tile_to_scratch<NC, FULL>(e, f->slot_x[i][h + 1], src);
        }
        if (MLP && BWD) tile_to_scratch<NC, FULL>(e, f->slot_e[i], te[i]);
      } else {
        if (!PREW) {
          W0[i] = gload<NC, FULL>(params + f->hop_param[i][0], d, lane);
          W1[i] = (nh > 1) ? gload<NC, FULL>(params + f->hop_param[i][1], d, lane) : W0[i];
        }
        const Vec<NC>& w0 = W0[i];
        const Vec<NC>& w1 = W1[i];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
          const int r = wave * RPW + rr;
          Vec<NC> x = RA[i].x[rr];
          VEC_OP(x, (DEC == DEC_DIAG) ? x.v[c] * w0.v[c] : x.v[c] + w0.v[c]);
          if (nh > 1) VEC_OP(x, (DEC == DEC_DIAG) ? x.v[c] * w1.v[c] : x.v[c] + w1.v[c]);
          lstore<NC>(te[i] + r * DP, x, lane);
          if (MLP && BWD) vstore_wt<NC, FULL>(e.wt, scratch_row(e, f->slot_e[i], r), x, d, lane);
        }
      }
    }
    // ---- intersection -> tacc (+ tmeta) ----
    if (MLP) {
      if (STAGE) {  // Pre arrived while the branch vectors were built; Post is requested now and lands behind the phase
        mat_commit<MR>(mr, mbuf, d, DP);
        mat_issue<MR>(mr, GQE_TILED(false, post_tile));
      }
      __syncthreads();
      GQE_STAMP(10);
      if (STAGE) {
        if (n == 3)
          pre_intersect_staged<NC, 3>(tacc, tmeta, mbuf, te, DP, wave, lane, inter_min);
        else
          pre_intersect_staged<NC, 2>(tacc, tmeta, mbuf, te, DP, wave, lane, inter_min);
        GQE_STAMP(11);
        __syncthreads();
        mat_commit<MR>(mr, mbuf, d, DP);                      // Post replaces Pre
        if (BWD) mat_issue<MR>(mr, GQE_TILED(true, post_tile));   // ... and the copy of Post^T is requested for the backward
      } else {
        if (n == 3)
          pre_intersect<NC, 3, FULL>(tacc, tmeta, GQE_TILED(false, pre_tile), te, d, DP, wave, lane, inter_min);
        else
          pre_intersect<NC, 2, FULL>(tacc, tmeta, GQE_TILED(false, pre_tile), te, d, DP, wave, lane, inter_min);
      }
      __syncthreads();
    } else {
      // element-wise first-arg-min / mean over the branches, own rows
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int j = lane + 64 * c;
          float best = te[0][r * DP + j];
          int meta = 0x70;
#pragma unroll
          for (int i = 1; i < GQE_MAX_BRANCH; ++i) {
            if (i < n) {
              const float v = te[i][r * DP + j];
              if (inter_min) {
                if (v < best) {
                  best = v;
                  meta = 0x70 | i;
                }
              } else {
                best += v;
              }
            }
          }
          tacc[r * DP + j] = inter_min ? best : best / (float)n;
          tmeta[r * DP + j] = meta;
        }
      }
    }
    GQE_STAMP(3);
    float* tqq = tacc;  // where q lives
    if (MLP) {
      if (BWD) tile_to_scratch<NC, FULL>(e, f->slot_hh, tacc);
      if (STAGE)
        tile_matmul_staged<false, NC>(tq, mbuf, tacc, DP, wave, lane);
      else
        tile_matmul<false, NC, FULL>(tq, GQE_TILED(false, post_tile), tacc, d, DP, wave, lane);  // q = Post . h
      __syncthreads();
      tqq = tq;
    }
    // optional projection after the intersection (3-chain_inter, model.py:107); te[] are free now
    float* tqpre = tqq;  // q before the final projection (needed by its backward)
    if (f->n_final) {
      if (DEC == DEC_BILINEAR) {
        if (BWD) tile_to_scratch<NC, FULL>(e, f->slot_fx, tqq);
        if (!MLP) __syncthreads();
        tile_matmul<false, NC, FULL>(te[0], GQE_TILED(false, final_tile), tqq, d, DP, wave, lane);
        __syncthreads();
      } else {
        if (!PREW) WF = gload<NC, FULL>(params + f->final_param, d, lane);
        const Vec<NC>& w = WF;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
          const int r = wave * RPW + rr;
          Vec<NC> x = lload<NC>(tqq + r * DP, lane);
          VEC_OP(x, (DEC == DEC_DIAG) ? x.v[c] * w.v[c] : x.v[c] + w.v[c]);
          lstore<NC>(te[0] + r * DP, x, lane);
        }
      }
      tqq = te[0];
    }
    // ---- scores, hinge, gradient seeds (own rows; no cross-wave traffic) ----
    GQE_STAMP(4);
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int r = wave * RPW + rr;
      const int q = e.q0 + r;
      Vec<NC> qv = lload<NC>(tqq + r * DP, lane);
      const float nq = fmaxf(gqe_sqrt(vdot<NC>(qv, qv)), COS_EPS);
      if (eval_mode) {
        if (q < B) store_query_record<NC, FULL>(e, r, qv, nq, 0.f, 0.f);
        continue;
      }
      const Vec<NC>& tp = RT.x[rr];
      const Vec<NC>& tn = RN.x[rr];
      const float ncp = fmaxf(gqe_sqrt(vdot<NC>(tp, tp)), COS_EPS);
      const float ncn = fmaxf(gqe_sqrt(vdot<NC>(tn, tn)), COS_EPS);
      const float sp = vdot<NC>(tp, qv) * gqe_rcp(ncp * nq);
      const float sn = has_neg ? vdot<NC>(tn, qv) * gqe_rcp(ncn * nq) : 0.f;
      if (q < B && lane == 0) {
        if (pos_out) pos_out[b.out_offset + q] = sp;
        if (neg_out && has_neg) neg_out[b.out_offset + q] = sn;
      }
      if (!BWD) continue;
      const float hinge = b.margin - (sp - sn);
      if ((q < B) && hinge > 0.f) loss_part += hinge;
      const bool act = (q < B) && hinge > 0.f && !same_row<NC>(e, RT.row[rr], RN.row[rr], tp, tn);   // (target == negative: see same_row)
      const float cp = act ? -gscale : 0.f, cn = act ? gscale : 0.f;
      const float ipq = gqe_rcp(ncp * nq), inq = gqe_rcp(ncn * nq), iqq = gqe_rcp(nq * nq);
      Vec<NC> gq, gtp, gtn;
      VEC_OP(gq, cp * (tp.v[c] * ipq - sp * qv.v[c] * iqq) + cn * (tn.v[c] * inq - sn * qv.v[c] * iqq));
      lstore<NC>(tg + r * DP, gq, lane);
      if (act) {
        const float ipp = sp * gqe_rcp(ncp * ncp), inn = sn * gqe_rcp(ncn * ncn);
        VEC_OP(gtp, cp * (qv.v[c] * ipq - tp.v[c] * ipp));
        VEC_OP(gtn, cn * (qv.v[c] * inq - tn.v[c] * inn));
        scatter_row<NC, FULL>(e, bags, GQE_TBAG, f->target_head, 0, wave * RPW + rr, RT, rr, gtp, olds[rr][0], blens[rr][0]);
        scatter_row<NC, FULL>(e, bags, GQE_TBAG, f->target_head, 1, wave * RPW + rr, RN, rr, gtn, olds[rr][1], blens[rr][1]);
      } else if (q < B) {
        sharded_zero<NC, FULL>(e, GQE_TBAG, RT.row[rr]);
        sharded_zero<NC, FULL>(e, GQE_TBAG, RN.row[rr]);
      }
    }
    GQE_STAMP(5);
    GQE_WSTAMP(0);
    if (BWD) {
      // ---- backward of the final projection: g (tile tgc) -> grad wrt q_pre ----
      float* tgc = tg;
      if (f->n_final) {
        if (DEC == DEC_BILINEAR) {
          tile_to_scratch<NC, FULL>(e, f->slot_fg, tg);
          __syncthreads();
          tile_matmul<true, NC, FULL>(te[1], GQE_TILED(true, final_tile), tg, d, DP, wave, lane);
          __syncthreads();
          tgc = te[1];
        } else {
          if (!PREW) WF = gload<NC, FULL>(params + f->final_param, d, lane);
          const Vec<NC>& w = WF;
          Vec<NC> gw = vzero<NC>();
#pragma unroll
          for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave * RPW + rr;
            Vec<NC> g = lload<NC>(tg + r * DP, lane);
            if (DEC == DEC_DIAG) {
              Vec<NC> qp = lload<NC>(tqpre + r * DP, lane);
              VEC_OP(gw, gw.v[c] + g.v[c] * qp.v[c]);
              VEC_OP(g, g.v[c] * w.v[c]);
              lstore<NC>(tg + r * DP, g, lane);
            } else {
              VEC_OP(gw, gw.v[c] + g.v[c]);
            }
          }
          vg.g[6] = gw;
          vg.param[6] = f->final_param;
        }
      }
      // ---- backward of Post: g_h -> tacc (h itself is already parked in scratch) ----
      float* tgh = tgc;  // grad wrt h (MLP) or wrt the intersection output (simple)
      if (MLP) {
        tile_to_scratch<NC, FULL>(e, f->slot_gq, tgc);
        GQE_WSTAMP(1);
        if (STAGE) {   // Post^T (requested behind the forward's Post contraction) replaces Post; Pre^T is requested for the next phase
          mat_commit<MR>(mr, mbuf, d, DP);
          mat_issue<MR>(mr, GQE_TILED(true, pre_tile));
        }
        __syncthreads();
        if (STAGE) {
          GQE_WSTAMP(2);
          tile_matmul_staged<true, NC>(tacc, mbuf, tgc, DP, wave, lane);  // the copy of Post^T (committed in front of the barrier above)
          GQE_WSTAMP(3);
          __syncthreads();
          GQE_WSTAMP(4);
          mat_commit<MR>(mr, mbuf, d, DP);  // the copy of Pre^T; the barrier in front of its contraction is below
          GQE_WSTAMP(5);
        } else {
          tile_matmul<true, NC, FULL>(tacc, GQE_TILED(true, post_tile), tgc, d, DP, wave, lane);  // g_h = Post^T g_q
          __syncthreads();
        }
        tgh = tacc;
      }
      GQE_STAMP(6);
      if (COMPACT) {
        // the heads the target / negative pushes returned (score phase, two contractions ago) are stored now: four registers
        // less across the Pre^T contraction, where this kernel's 80-VGPR budget is tightest
        if (lane == 0) {
#pragma unroll
          for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
            for (int role = 0; role < 2; ++role)
              if (olds[rr][role] != GQE_NO_PUSH && blens[rr][role] == 0)
                e.next[e.b.entry_base + (int64_t)role * e.b.B + (e.q0 + wave * RPW + rr)] = olds[rr][role];
        }
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
          for (int role = 0; role < 2; ++role)
            if (blens[rr][role] == 0) olds[rr][role] = GQE_NO_PUSH;
      }
      // ---- backward of Pre for all branches: g_e_i -> te[i] ----
      if (MLP) {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {  // g_z_i rows for the deferred dPre (own rows): g_h and the meta word are read once
          const int r = wave * RPW + rr;    // for all branches; bit 8 + i of the meta word says whether branch i receives the element
          const float gsc = inter_min ? 1.f : inv_n;
          float gh[NC];
          int mt[NC];
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const int j = lane + 64 * c;
            gh[c] = tgh[r * DP + j] * gsc;   // (columns past d: 0)
            mt[c] = tmeta[r * DP + j];
          }
#pragma unroll
          for (int i = 0; i < GQE_MAX_BRANCH; ++i) {
            if (i >= n) continue;
            Vec<NC> gz;
#pragma unroll
            for (int c = 0; c < NC; ++c) gz.v[c] = keep_if_bit(gh[c], mt[c], 8 + i);
            vstore_wt<NC, FULL>(e.wt, scratch_row(e, f->slot_gz[i], r), gz, d, lane);
          }
        }
        if (STAGE) {
          __syncthreads();
          GQE_STAMP(12);
          GQE_WSTAMP(6);
          if (n == 3)
            pre_intersect_bwd_staged<NC, 3>(te, mbuf, tgh, tmeta, DP, wave, lane, inter_min);
          else
            pre_intersect_bwd_staged<NC, 2>(te, mbuf, tgh, tmeta, DP, wave, lane, inter_min);
        } else {
          if (n == 3)
            pre_intersect_bwd<NC, 3, FULL>(te, GQE_TILED(true, pre_tile), tgh, tmeta, d, DP, wave, lane, inter_min);
          else
            pre_intersect_bwd<NC, 2, FULL>(te, GQE_TILED(true, pre_tile), tgh, tmeta, d, DP, wave, lane, inter_min);
        }
        GQE_STAMP(13);
        GQE_WSTAMP(7);
        __syncthreads();
        GQE_STAMP(14);
      } else if (DEC == DEC_BILINEAR) {
        // simple intersection + Bilinear hops: the masked gradient has to be a tile for the MFMA
        if (tgc == te[1]) __syncthreads();
#pragma unroll
        for (int i = 0; i < GQE_MAX_BRANCH; ++i) {
          if (i >= n) continue;
          float* dstt = (tgh == te[i]) ? tt : te[i];
#pragma unroll
          for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave * RPW + rr;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              const int j = lane + 64 * c;
              dstt[r * DP + j] = mask_gz(tgh[r * DP + j], tmeta[r * DP + j], i, inter_min, inv_n, false);
            }
          }
        }
      }
      // ---- back through the hops of every branch, scatter the anchors ----
#pragma unroll
      for (int i = 0; i < GQE_MAX_BRANCH; ++i) {
        if (i >= n) continue;
        const int nh = f->n_hops[i];
        if (DEC == DEC_BILINEAR) {
          float* tcur = (!MLP && tgh == te[i]) ? tt : te[i];
          float* tnext = tq;  // tq is dead in the backward; tt may hold another branch's masked gradient
          for (int h = nh - 1; h >= 0; --h) {
            tile_to_scratch<NC, FULL>(e, f->slot_gy[i][h], tcur);
            __syncthreads();
            tile_matmul<true, NC, FULL>(tnext, GQE_TILED(true, hop_tile[i][h]), tcur, d, DP, wave, lane);
            __syncthreads();
            float* tmp = tcur;
            tcur = tnext;
            tnext = tmp;
          }
#pragma unroll
          for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave * RPW + rr;
            if (e.q0 + r >= B) continue;
            scatter_row<NC, FULL>(e, bags, GQE_ABAG(i), f->anchor_head[i], 2 + i, r, RA[i], rr, lload<NC>(tcur + r * DP, lane),
                            olds[rr][2 + i], blens[rr][2 + i]);
          }
          __syncthreads();  // tt / tq are rewritten by the next branch
        } else {
          if (!PREW) {
            W0[i] = gload<NC, FULL>(params + f->hop_param[i][0], d, lane);
            W1[i] = (nh > 1) ? gload<NC, FULL>(params + f->hop_param[i][1], d, lane) : W0[i];
          }
          const Vec<NC>& w0 = W0[i];
          const Vec<NC>& w1 = W1[i];
          Vec<NC> gw0 = vzero<NC>(), gw1 = vzero<NC>();
#pragma unroll
          for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave * RPW + rr;
            if (e.q0 + r >= B) continue;
            const Vec<NC>& x = RA[i].x[rr];
            Vec<NC> g;
            if (MLP) {
              g = lload<NC>(te[i] + r * DP, lane);
            } else {
#pragma unroll
              for (int c = 0; c < NC; ++c) {
                const int j = lane + 64 * c;
                g.v[c] = mask_gz(tgh[r * DP + j], tmeta[r * DP + j], i, inter_min, inv_n, false);
              }
            }
            if (DEC == DEC_DIAG) {
              // x_1 = x_0 (.) w0, e = x_1 (.) w1 : walk back from the last hop
              if (nh > 1) {
                VEC_OP(gw1, gw1.v[c] + g.v[c] * x.v[c] * w0.v[c]);
                VEC_OP(g, g.v[c] * w1.v[c]);
              }
              VEC_OP(gw0, gw0.v[c] + g.v[c] * x.v[c]);
              VEC_OP(g, g.v[c] * w0.v[c]);
            } else {
              VEC_OP(gw0, gw0.v[c] + g.v[c]);
            }
            scatter_row<NC, FULL>(e, bags, GQE_ABAG(i), f->anchor_head[i], 2 + i, r, RA[i], rr, g, olds[rr][2 + i], blens[rr][2 + i]);
          }
          // (slot indices spelled out per branch: indexed with the loop variable, the TransE 8-wave d = 128 kernels kept the
          // slots in a dynamically indexed private array — 40 B of scratch without a single spilled register)
#define GQE_VG_SET(I)                                          \
  if (i == I) {                                                \
    vg.g[2 * I] = gw0;                                         \
    vg.param[2 * I] = f->hop_param[I][0];                      \
    if (nh > 1) {                                              \
      vg.g[2 * I + 1] = (DEC == DEC_DIAG) ? gw1 : gw0;         \
      vg.param[2 * I + 1] = f->hop_param[I][1];                \
    }                                                          \
  }
          GQE_VG_SET(0)
          GQE_VG_SET(1)
          GQE_VG_SET(2)
#undef GQE_VG_SET
        }
      }
    }
  }
  GQE_STAMP(7);
  GQE_WSTAMP(8);
  if (BWD) {
    // mean hinge loss of the batch (model.py:124-126) and the weighted iteration loss: reduce the waves in
    // LDS (thousands of same-address device atomics serialise at ~12 ns each) and park one partial per tile.
    if (DEC != DEC_BILINEAR) {
      vecgrads_commit<NC, FULL>(e, smem, reinterpret_cast<long long*>(s_idx), vg, red, loss_part, olds, blens, bags.max_len);
    } else {
      __syncthreads();
      if (lane == 0) red[wave] = loss_part;
      __syncthreads();
      push_links(e, olds, blens, bags.max_len);
    }
    GQE_WSTAMP(10);
    if (threadIdx.x == 0) {
      float l = 0.f;
#pragma unroll
      for (int w = 0; w < GQE_FW; ++w) l += red[w];
      tile_loss[tile_id] = l;  // summed per batch by the finalize block of the pair-GEMM launch
    }
  }
  GQE_STAMP(8);
  // (split steps: the launch's riders stop taking rounds once every tile has left)
  if (RIDE && ride.blocks > 0 && ride.stop && threadIdx.x == 0) __hip_atomic_fetch_add(ride.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#undef GQE_STAMP
#undef GQE_WSTAMP
#undef GQE_DSC
}

// ------------------------------------------------------------------------------------------
// per-(DEC, MLP) launcher, instantiated once per translation unit (gqe_fused_inst.hip)
// ------------------------------------------------------------------------------------------
inline size_t gqe_fused_lds_bytes_impl(int d, bool stage, bool compact) {
  const int DP = 64 * ((d + 63) / 64) + 4;   // tiles padded to whole 64-float chunks
  if (compact) return (size_t)(5 * GQE_TQ * DP + 64 + 5 * GQE_TQ) * sizeof(float);
  return (size_t)(8 * GQE_TQ * DP + GQE_FW * d + 5 * GQE_TQ + (stage ? d * DP : 0)) * sizeof(float);
}

template <int DEC, bool MLP, int NC, bool FULL>
static hipError_t launch_fused_v(const GqeFusedArgs& a) {
  static const size_t lds_pad = [] {   // GQE_DEBUG_LDS_PAD: occupancy experiments only
    const char* e = getenv("GQE_DEBUG_LDS_PAD");
    return e ? (size_t)atol(e) : (size_t)0;
  }();
  const size_t lds = gqe_fused_lds_bytes_impl(a.d, MLP && DEC != DEC_BILINEAR && FULL && NC <= 2 && GQE_FW == 16,
                                              FusedShape<DEC, MLP, NC, FULL, GQE_FW>::COMPACT) + lds_pad;
  const int riders = (a.bwd && FULL) ? a.split.blocks : 0;
  if (a.split.blocks > 0 && !riders) return hipErrorInvalidValue;   // (gqe_fused_can_ride said no: the host does not ask)
  static const bool lean_off = getenv("GQE_NO_LEAN") != nullptr;   // (A / B runs)
  constexpr bool COMPACT = FusedShape<DEC, MLP, NC, FULL, GQE_FW>::COMPACT;
#ifdef GQE_LEAN_PROF
  const bool prof_ok = true;
#else
  const bool prof_ok = !a.prof;
#endif
  if (FULL && a.bwd && !lean_off && !a.fetched && prof_ok && a.bags.max_len == 0 && (!COMPACT || riders == 0)) {
    if constexpr (FULL)
      hipLaunchKernelGGL((gqe_fused_kernel<DEC, MLP, NC, FULL, true, GQE_FW, 1>), dim3(a.plan.tiles + riders), dim3(GQE_FWT), lds, a.stream, a.plan,
                         a.formulas, a.params, a.grads, a.ws, a.idx, a.d, a.tile_loss, a.pos, a.neg, a.inter_min, a.head, a.next, a.contrib, a.bags,
                         a.link_contrib, a.link_counter, a.max_entries, a.fetched, a.contrib_bag, a.bag_shift, a.hot, a.prof, a.split);
    return hipGetLastError();
  }
  if (FULL && GQE_FW == 16 && a.bwd && !lean_off && a.fetched && prof_ok && a.bags.max_len == 0 && riders == 0) {
    if constexpr (FULL && GQE_FW == 16)
      hipLaunchKernelGGL((gqe_fused_kernel<DEC, MLP, NC, FULL, true, GQE_FW, 2>), dim3(a.plan.tiles), dim3(GQE_FWT), lds, a.stream, a.plan,
                         a.formulas, a.params, a.grads, a.ws, a.idx, a.d, a.tile_loss, a.pos, a.neg, a.inter_min, a.head, a.next, a.contrib, a.bags,
                         a.link_contrib, a.link_counter, a.max_entries, a.fetched, a.contrib_bag, a.bag_shift, a.hot, a.prof, a.split);
    return hipGetLastError();
  }
  if (a.bwd)
    hipLaunchKernelGGL((gqe_fused_kernel<DEC, MLP, NC, FULL, true, GQE_FW>), dim3(a.plan.tiles + riders), dim3(GQE_FWT), lds, a.stream, a.plan,
                       a.formulas, a.params, a.grads, a.ws, a.idx, a.d, a.tile_loss, a.pos, a.neg, a.inter_min, a.head, a.next, a.contrib, a.bags, a.link_contrib, a.link_counter,
                       a.max_entries, a.fetched, a.contrib_bag, a.bag_shift, a.hot, a.prof, a.split);
  else
    hipLaunchKernelGGL((gqe_fused_kernel<DEC, MLP, NC, FULL, false, GQE_FW>), dim3(a.plan.tiles), dim3(GQE_FWT), lds, a.stream, a.plan,
                       a.formulas, a.params, a.grads, a.ws, a.idx, a.d, a.tile_loss, a.pos, a.neg, a.inter_min, a.head, a.next, a.contrib, a.bags, a.link_contrib, a.link_counter,
                       a.max_entries, a.fetched, a.contrib_bag, a.bag_shift, a.hot, a.prof, a.split);
  return hipGetLastError();
}

template <int DEC, bool MLP>
static hipError_t launch_fused_dm(const GqeFusedArgs& a) {
  const int nc = (a.d + 63) / 64;
  const bool full = (a.d % 64) == 0;
#if GQE_FW == 16
  // every accepted d (a multiple of 16 up to 256) runs a 16-wave kernel of NC = ceil(d / 64) chunks: straight-line FULL code at
  // d = 64 / 128 / 256, the guarded form (buffer-addressed rows, padded tiles: gqe_common.h) everywhere else
  switch (nc) {
    case 1: return full ? launch_fused_v<DEC, MLP, 1, true>(a) : launch_fused_v<DEC, MLP, 1, false>(a);
    case 2: return full ? launch_fused_v<DEC, MLP, 2, true>(a) : launch_fused_v<DEC, MLP, 2, false>(a);
    case 3: return launch_fused_v<DEC, MLP, 3, false>(a);   // (d = 192 too)
    case 4: return full ? launch_fused_v<DEC, MLP, 4, true>(a) : launch_fused_v<DEC, MLP, 4, false>(a);
    default: return hipErrorInvalidValue;
  }
#else
  // 8 waves, two rows per wave: d = 128 launches with many tiles (two / three workgroups per CU)
  if (nc == 2 && full) return launch_fused_v<DEC, MLP, 2, true>(a);
  return hipErrorInvalidValue;
#endif
}

#endif
