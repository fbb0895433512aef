// gqe_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the conjunctive-query hot path.
//
// What is computed follows netquery/model.py:70-127, encoders.py:40-43, decoders.py:142-150,
// 200-208, 228-236, 288-319 (maths restated in SURVEY.md Appendix B / oracle/netquery_numpy.py);
// how it is computed is specific to this chip:
//   * one workgroup (4 wave64) owns a TILE of 16 queries of ONE batch (= one Formula), so every
//     relation parameter is workgroup-uniform; a grouped launch covers all batches of an iteration;
//   * embedding rows (d fp32) are read as whole rows, one wave per row, lanes strided over the row
//     (64 lanes x 4 B = 256 B contiguous per instruction), normalised with wave reductions;
//   * element-wise decoders (bilinear-diag, TransE) and the min/mean set reduction are wave ops;
//   * d x d contractions (full-Bilinear hops, SetIntersection Pre/Post) run on the matrix cores with
//     the exact-f32 v_mfma_f32_16x16x4_f32: the 16-query tile in LDS is the B operand, the
//     parameter matrix streams from L2 as the A operand (float4 per lane, 4 MFMAs per load);
//   * the positive and the negative score share the query side (the reference recomputes it);
//   * row gradients are scattered with hardware fp32 atomics into the dense grad arena;
//     rank-B matrix gradients are deferred to a second MFMA kernel over (left,right) row pairs
//     parked in an L2/MALL-resident scratch;
//   * the optimiser is one fused float4 pass (p,g,m,v -> p,m,v, g:=0).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gqe_dev.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DEC_DIAG 0
#define DEC_TRANSE 1
#define DEC_BILINEAR 2
#define COS_EPS 1e-8f

// ------------------------------------------------------------------------------------------
// small vector-of-a-row helpers: a wave owns a row of d floats, lane l holds j = l + 64*c
// ------------------------------------------------------------------------------------------
template <int NC>
struct Vec {
  float v[NC];
};

template <int NC>
__device__ __forceinline__ Vec<NC> vzero() {
  Vec<NC> r;
#pragma unroll
  for (int c = 0; c < NC; ++c) r.v[c] = 0.f;
  return r;
}

template <int NC>
__device__ __forceinline__ Vec<NC> vload(const float* p, int d, int lane) {
  Vec<NC> r;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int j = lane + 64 * c;
    r.v[c] = (j < d) ? p[j] : 0.f;
  }
  return r;
}

template <int NC>
__device__ __forceinline__ void vstore(float* p, const Vec<NC>& x, int d, int lane) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int j = lane + 64 * c;
    if (j < d) p[j] = x.v[c];
  }
}

template <int NC>
__device__ __forceinline__ void vatomic_add(float* p, const Vec<NC>& x, int d, int lane) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int j = lane + 64 * c;
    if (j < d) unsafeAtomicAdd(p + j, x.v[c]);
  }
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}

template <int NC>
__device__ __forceinline__ float vdot(const Vec<NC>& a, const Vec<NC>& b) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) s += a.v[c] * b.v[c];
  return wave_sum(s);
}

#define VEC_OP(out, expr)                          \
  _Pragma("unroll") for (int c = 0; c < NC; ++c) { \
    (out).v[c] = (expr);                           \
  }

// ------------------------------------------------------------------------------------------
// tile matmul on the matrix cores.
//   TRANS = false: dst[q][i] = sum_k M[i][k] * src[q][k]      (M . x, "project", decoders.py:150)
//   TRANS = true : dst[q][i] = sum_k M[k][i] * src[q][k]      (M^T . x; x^T M, decoders.py:145)
// src/dst: LDS tiles [16][DP]; M: global, row-major d x d.  Wave w computes the 16-row slabs
// i0 = 16*(w, w+4, ...).  v_mfma_f32_16x16x4_f32: lane l feeds A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]
// and receives D[i=4*(l>>4)+r][j=l&15], r = 0..3.  Each lane loads 4 consecutive k at once, so the
// k index of MFMA step s is k0 + 4*(l>>4) + s for A and B alike (any consistent k order is valid).
// ------------------------------------------------------------------------------------------
template <bool TRANS>
__device__ __forceinline__ void tile_matmul(float* __restrict__ dst, const float* __restrict__ M,
                                            const float* __restrict__ src, int d, int DP, int wave, int lane) {
  const int lq = lane & 15;
  const int lk = lane >> 4;
  for (int it = wave; it * 16 < d; it += GQE_WAVES) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int i0 = it * 16;
    for (int k0 = 0; k0 < d; k0 += 16) {
      const float4 b = *reinterpret_cast<const float4*>(src + lq * DP + k0 + 4 * lk);
      float4 a;
      if (!TRANS) {
        a = *reinterpret_cast<const float4*>(M + (size_t)(i0 + lq) * d + k0 + 4 * lk);
      } else {
        const float* mp = M + (size_t)(k0 + 4 * lk) * d + i0 + lq;
        a.x = mp[0];
        a.y = mp[d];
        a.z = mp[2 * d];
        a.w = mp[3 * d];
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    }
    *reinterpret_cast<float4*>(dst + lq * DP + i0 + 4 * lk) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

// ------------------------------------------------------------------------------------------
// the fused per-tile kernel
// ------------------------------------------------------------------------------------------
struct TileEnv {
  const GqeDevBatch* b;
  const float* params;
  float* grads;
  float* ws;  // scratch (floats)
  const int32_t* idx;
  int d, DP, wave, lane, q0;  // q0 = first query of this tile
};

__device__ __forceinline__ float* scratch_row(const TileEnv& e, int slot, int r) {
  return e.ws + e.b->scratch_base + ((size_t)slot * e.b->Bpad + e.q0 + r) * e.d;
}

// gather one table row and L2-normalise it (encoders.py:41-43: x / ||x||, no eps)
template <int NC>
__device__ __forceinline__ Vec<NC> gather_norm(const TileEnv& e, int64_t table, const int32_t* rows, int r,
                                               float& nrm, int& row) {
  const int q = e.q0 + r;
  if (q >= e.b->B) {
    nrm = 1.f;
    row = -1;
    return vzero<NC>();
  }
  row = rows[q];
  Vec<NC> x = vload<NC>(e.params + table + (size_t)row * e.d, e.d, e.lane);
  nrm = sqrtf(vdot<NC>(x, x));
  VEC_OP(x, x.v[c] / nrm);
  return x;
}

// backward of x/||x||:  (g - xhat (xhat.g)) / ||x||, scattered into the table's dense gradient
template <int NC>
__device__ __forceinline__ void scatter_norm_bwd(const TileEnv& e, int64_t table, int row, const Vec<NC>& xhat,
                                                 float nrm, const Vec<NC>& g) {
  const float pg = vdot<NC>(xhat, g);
  Vec<NC> gx;
  VEC_OP(gx, (g.v[c] - xhat.v[c] * pg) / nrm);
  vatomic_add<NC>(e.grads + table + (size_t)row * e.d, gx, e.d, e.lane);
}

// cross-wave reduction of a per-wave partial relation-vector gradient, then one atomic row per block
template <int NC>
__device__ __forceinline__ void flush_vec_grad(const TileEnv& e, float* red /*[GQE_WAVES][d]*/, int64_t param,
                                               const Vec<NC>& part) {
  __syncthreads();
  vstore<NC>(red + e.wave * e.d, part, e.d, e.lane);
  __syncthreads();
  if (e.wave == 0) {
    Vec<NC> s = vload<NC>(red, e.d, e.lane);
    for (int w = 1; w < GQE_WAVES; ++w) {
      Vec<NC> t = vload<NC>(red + w * e.d, e.d, e.lane);
      VEC_OP(s, s.v[c] + t.v[c]);
    }
    vatomic_add<NC>(e.grads + param, s, e.d, e.lane);
  }
}

template <int DEC, bool MLP, int NC, bool BWD>
__global__ __launch_bounds__(GQE_THREADS) void gqe_fused_kernel(const GqeDevBatch* __restrict__ batches, int n_batches,
                                                               const float* __restrict__ params,
                                                               float* __restrict__ grads, float* __restrict__ ws,
                                                               const int32_t* __restrict__ idx, int d,
                                                               float* __restrict__ losses, float* __restrict__ pos_out,
                                                               float* __restrict__ neg_out, int inter_min) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // ---- which batch / tile am I ----
  int bi = 0;
  while (bi + 1 < n_batches && (int)blockIdx.x >= batches[bi + 1].tile_begin) ++bi;
  const GqeDevBatch* __restrict__ b = batches + bi;
  TileEnv e;
  e.b = b;
  e.params = params;
  e.grads = grads;
  e.ws = ws;
  e.idx = idx;
  e.d = d;
  e.DP = d + 4;
  e.wave = threadIdx.x >> 6;
  e.lane = threadIdx.x & 63;
  e.q0 = ((int)blockIdx.x - b->tile_begin) * GQE_TQ;
  const int DP = e.DP, lane = e.lane, wave = e.wave;
  const int B = b->B;
  const int32_t* itarget = idx + b->idx_offset;
  const int32_t* ineg = itarget + B;
  const int32_t* ianchor = itarget + (b->has_neg ? 2 : 1) * (size_t)B;

  // ---- LDS carve: 5 tiles [16][DP] + red[4][d] + scalars ----
  float* t0 = smem;
  float* t1 = t0 + GQE_TQ * DP;
  float* t2 = t1 + GQE_TQ * DP;
  float* t3 = t2 + GQE_TQ * DP;
  int* tmeta = reinterpret_cast<int*>(t3 + GQE_TQ * DP);
  float* red = reinterpret_cast<float*>(tmeta + GQE_TQ * DP);
  float* sc_nrm = red + GQE_WAVES * d;  // [3][16] anchor norms
  float* sc_cp = sc_nrm + 3 * GQE_TQ;   // [16] d loss / d s+
  float* sc_cn = sc_cp + GQE_TQ;        // [16] d loss / d s-
  float* sc_misc = sc_cn + GQE_TQ;      // [4][16] chain/bilinear: s+, s-, |u+|, |u-|

  const bool is_chain = b->qtype <= 2;
  const float gscale = b->grad_scale;  // loss_weight / B
  float loss_part = 0.f;

  if (is_chain) {
    // =====================================================================================
    // chains: score(target, anchor) with the relations applied on the TARGET side
    // =====================================================================================
    const int K = b->n_hops[0];
    if (DEC != DEC_BILINEAR) {
      // ---- bilinear-diag / TransE: everything stays in registers, one wave per query ----
      Vec<NC> w[GQE_MAX_HOPS];
      Vec<NC> wcomb;  // diag: prod_h w_h ; transe: sum_h w_h
      VEC_OP(wcomb, (DEC == DEC_DIAG) ? 1.f : 0.f);
      for (int h = 0; h < K; ++h) {
        w[h] = vload<NC>(params + b->hop_param[0][h], d, lane);
        VEC_OP(wcomb, (DEC == DEC_DIAG) ? wcomb.v[c] * w[h].v[c] : wcomb.v[c] + w[h].v[c]);
      }
      Vec<NC> gw_acc = vzero<NC>();  // diag: sum_rows (cp t+ + cn t-) (.) a ; transe: sum_rows (gu+ + gu-)
      for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
        const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
        const int q = e.q0 + r;
        if (q >= B) continue;  // wave-uniform
        float na, ntp, ntn = 1.f;
        int ra, rtp, rtn = -1;
        Vec<NC> a = gather_norm<NC>(e, b->anchor_table[0], ianchor, r, na, ra);
        Vec<NC> tp = gather_norm<NC>(e, b->target_table, itarget, r, ntp, rtp);
        Vec<NC> tn = vzero<NC>();
        if (b->has_neg) tn = gather_norm<NC>(e, b->target_table, ineg, r, ntn, rtn);
        float sp, sn = 0.f;
        float nap = 1.f, nup = 1.f, nun = 1.f;  // transe cosine norms
        Vec<NC> up, un;
        if (DEC == DEC_DIAG) {
          // decoders.py:228-233: acts = t * w1 * .. * wk ; score = sum(acts * a)   (raw dot)
          up = tp;
          un = tn;
          for (int h = 0; h < K; ++h) {
            VEC_OP(up, up.v[c] * w[h].v[c]);
            VEC_OP(un, un.v[c] * w[h].v[c]);
          }
          sp = vdot<NC>(up, a);
          if (b->has_neg) sn = vdot<NC>(un, a);
        } else {
          // decoders.py:200-205: u = t + sum w ; score = cos(a, u)
          VEC_OP(up, tp.v[c] + wcomb.v[c]);
          VEC_OP(un, tn.v[c] + wcomb.v[c]);
          nap = fmaxf(sqrtf(vdot<NC>(a, a)), COS_EPS);
          nup = fmaxf(sqrtf(vdot<NC>(up, up)), COS_EPS);
          sp = vdot<NC>(a, up) / (nap * nup);
          if (b->has_neg) {
            nun = fmaxf(sqrtf(vdot<NC>(un, un)), COS_EPS);
            sn = vdot<NC>(a, un) / (nap * nun);
          }
        }
        if (lane == 0) {
          if (pos_out) pos_out[b->out_offset + q] = sp;
          if (neg_out && b->has_neg) neg_out[b->out_offset + q] = sn;
        }
        if (!BWD) continue;
        const float hinge = b->margin - (sp - sn);
        if (hinge > 0.f) {
          loss_part += hinge;
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
            VEC_OP(gtp, cp * (a.v[c] / (nap * nup) - sp * up.v[c] / (nup * nup)));
            VEC_OP(gtn, cn * (a.v[c] / (nap * nun) - sn * un.v[c] / (nun * nun)));
            VEC_OP(ga, cp * (up.v[c] / (nap * nup) - sp * a.v[c] / (nap * nap)) +
                           cn * (un.v[c] / (nap * nun) - sn * a.v[c] / (nap * nap)));
            VEC_OP(gw_acc, gw_acc.v[c] + gtp.v[c] + gtn.v[c]);
          }
          scatter_norm_bwd<NC>(e, b->target_table, rtp, tp, ntp, gtp);
          scatter_norm_bwd<NC>(e, b->target_table, rtn, tn, ntn, gtn);
          scatter_norm_bwd<NC>(e, b->anchor_table[0], ra, a, na, ga);
        }
      }
      if (BWD) {
        for (int h = 0; h < K; ++h) {
          Vec<NC> part = gw_acc;
          if (DEC == DEC_DIAG) {
            // d/dw_h = sum_rows (..) (.) prod_{j != h} w_j
            for (int j = 0; j < K; ++j)
              if (j != h) VEC_OP(part, part.v[c] * w[j].v[c]);
          }
          flush_vec_grad<NC>(e, red, b->hop_param[0][h], part);
        }
      }
    } else {
      // ---- full Bilinear chain (decoders.py:142-147): act = t^T M1..Mk ; s = cos(act, a) ----
      // t0/t1: u+ ping-pong, t2/t3: u- ping-pong; anchor stays in registers (re-gathered in bwd)
      float* cur[2] = {t0, t2};
      float* alt[2] = {t1, t3};
      const int nside = b->has_neg ? 2 : 1;
      for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
        const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
        for (int s = 0; s < nside; ++s) {
          float nt;
          int rt;
          Vec<NC> t = gather_norm<NC>(e, b->target_table, s ? ineg : itarget, r, nt, rt);
          vstore<NC>(cur[s] + r * DP, t, d, lane);
          if (BWD) vstore<NC>(scratch_row(e, b->slot_act[s][0], r), t, d, lane);
        }
      }
      for (int h = 0; h < K; ++h) {
        __syncthreads();
        for (int s = 0; s < nside; ++s) tile_matmul<true>(alt[s], params + b->hop_param[0][h], cur[s], d, DP, wave, lane);
        __syncthreads();
        for (int s = 0; s < nside; ++s) {
          float* tmp = cur[s];
          cur[s] = alt[s];
          alt[s] = tmp;
        }
        if (BWD && h + 1 < K) {
          for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
            const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
            for (int s = 0; s < nside; ++s)
              vstore<NC>(scratch_row(e, b->slot_act[s][h + 1], r), vload<NC>(cur[s] + r * DP, d, lane), d, lane);
          }
        }
      }
      // scores + gradient seeds; g_u overwrites u in place
      for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
        const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
        const int q = e.q0 + r;
        float na;
        int ra;
        Vec<NC> a = gather_norm<NC>(e, b->anchor_table[0], ianchor, r, na, ra);
        const float nac = fmaxf(sqrtf(vdot<NC>(a, a)), COS_EPS);
        Vec<NC> u[2];
        float su[2] = {0.f, 0.f}, nu[2] = {1.f, 1.f};
        for (int s = 0; s < nside; ++s) {
          u[s] = vload<NC>(cur[s] + r * DP, d, lane);
          nu[s] = fmaxf(sqrtf(vdot<NC>(u[s], u[s])), COS_EPS);
          su[s] = vdot<NC>(u[s], a) / (nu[s] * nac);
        }
        if (q < B && lane == 0) {
          if (pos_out) pos_out[b->out_offset + q] = su[0];
          if (neg_out && b->has_neg) neg_out[b->out_offset + q] = su[1];
        }
        if (!BWD) continue;
        const float hinge = b->margin - (su[0] - su[1]);
        const bool act = (q < B) && hinge > 0.f;
        if (act) loss_part += hinge;
        const float cf[2] = {act ? -gscale : 0.f, act ? gscale : 0.f};
        Vec<NC> ga = vzero<NC>();
        for (int s = 0; s < 2; ++s) {
          Vec<NC> gu;
          VEC_OP(gu, cf[s] * (a.v[c] / (nu[s] * nac) - su[s] * u[s].v[c] / (nu[s] * nu[s])));
          VEC_OP(ga, ga.v[c] + cf[s] * (u[s].v[c] / (nu[s] * nac) - su[s] * a.v[c] / (nac * nac)));
          vstore<NC>(cur[s] + r * DP, gu, d, lane);
        }
        if (act) scatter_norm_bwd<NC>(e, b->anchor_table[0], ra, a, na, ga);
      }
      if (BWD) {
        // back through the hops: act_{h+1} = act_h M_h  =>  g_act_h = g_act_{h+1} M_h^T (= M . g per row),
        // dM_h += act_h^T g_act_{h+1}  (deferred: pair (slot_act[s][h], slot_gact[s][h]))
        for (int h = K - 1; h >= 0; --h) {
          for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
            const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
            for (int s = 0; s < 2; ++s)
              vstore<NC>(scratch_row(e, b->slot_gact[s][h], r), vload<NC>(cur[s] + r * DP, d, lane), d, lane);
          }
          __syncthreads();
          for (int s = 0; s < 2; ++s) tile_matmul<false>(alt[s], params + b->hop_param[0][h], cur[s], d, DP, wave, lane);
          __syncthreads();
          for (int s = 0; s < 2; ++s) {
            float* tmp = cur[s];
            cur[s] = alt[s];
            alt[s] = tmp;
          }
        }
        for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
          const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
          if (e.q0 + r >= B) continue;
          for (int s = 0; s < 2; ++s) {
            float nt;
            int rt;
            Vec<NC> t = gather_norm<NC>(e, b->target_table, s ? ineg : itarget, r, nt, rt);
            Vec<NC> g = vload<NC>(cur[s] + r * DP, d, lane);
            scatter_norm_bwd<NC>(e, b->target_table, rt, t, nt, g);
          }
        }
      }
    }
  } else {
    // =====================================================================================
    // intersections: q = I( Proj(a_1), Proj(a_2)[, Proj(a_3)] ) [-> Proj]; s = cos(t, q)
    // =====================================================================================
    const int n = b->n_anchors;
    float* tx = t0;    // working tile (branch vector)
    float* ty = t1;    // matmul output / second working tile
    float* tacc = t2;  // running min / sum  -> h (MLP) or q (simple)
    float* tq = t3;    // query vector q
    for (int i = 0; i < n; ++i) {
      for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
        const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
        float nrm;
        int row;
        Vec<NC> x = gather_norm<NC>(e, b->anchor_table[i], ianchor + (size_t)i * B, r, nrm, row);
        if (lane == 0) sc_nrm[i * GQE_TQ + r] = nrm;
        if (DEC == DEC_BILINEAR) {
          if (BWD) vstore<NC>(scratch_row(e, b->slot_x[i][0], r), x, d, lane);
        } else {
          for (int h = 0; h < b->n_hops[i]; ++h) {
            Vec<NC> w = vload<NC>(params + b->hop_param[i][h], d, lane);
            VEC_OP(x, (DEC == DEC_DIAG) ? x.v[c] * w.v[c] : x.v[c] + w.v[c]);
          }
        }
        vstore<NC>(tx + r * DP, x, d, lane);
      }
      if (DEC == DEC_BILINEAR) {
        for (int h = 0; h < b->n_hops[i]; ++h) {
          __syncthreads();
          tile_matmul<false>(ty, params + b->hop_param[i][h], tx, d, DP, wave, lane);
          __syncthreads();
          float* tmp = tx;
          tx = ty;
          ty = tmp;
          if (BWD && h + 1 < b->n_hops[i]) {
            for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
              const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
              vstore<NC>(scratch_row(e, b->slot_x[i][h + 1], r), vload<NC>(tx + r * DP, d, lane), d, lane);
            }
          }
        }
      }
      // tx = e_i
      const float* tv = tx;
      if (MLP) {
        if (BWD) {
          for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
            const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
            vstore<NC>(scratch_row(e, b->slot_e[i], r), vload<NC>(tx + r * DP, d, lane), d, lane);
          }
        }
        __syncthreads();
        tile_matmul<false>(ty, params + b->pre_param, tx, d, DP, wave, lane);  // z_i = Pre . e_i
        __syncthreads();
        tv = ty;
      }
      // accumulate: first-index arg-min (torch.min) or sum (torch.mean); relu for the MLP form
      for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
        const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int j = lane + 64 * c;
          if (j >= d) continue;
          float v = tv[r * DP + j];
          int pos = 1;
          if (MLP) {
            pos = v > 0.f;
            v = fmaxf(v, 0.f);
          }
          float acc = v;
          int meta = pos << 4;
          if (i > 0) {
            acc = tacc[r * DP + j];
            meta = tmeta[r * DP + j] | (pos << (4 + i));
            if (inter_min) {
              if (v < acc) {
                acc = v;
                meta = (meta & ~3) | i;
              }
            } else {
              acc += v;
            }
          }
          tacc[r * DP + j] = acc;
          tmeta[r * DP + j] = meta;
        }
      }
      if (MLP) __syncthreads();  // ty is overwritten by the next branch's matmul
    }
    if (!inter_min) {
      for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
        const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
        Vec<NC> h = vload<NC>(tacc + r * DP, d, lane);
        VEC_OP(h, h.v[c] / (float)n);
        vstore<NC>(tacc + r * DP, h, d, lane);
      }
    }
    float* tqq = tacc;  // where q lives
    if (MLP) {
      if (BWD) {
        for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
          const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
          vstore<NC>(scratch_row(e, b->slot_hh, r), vload<NC>(tacc + r * DP, d, lane), d, lane);
        }
      }
      __syncthreads();
      tile_matmul<false>(tq, params + b->post_param, tacc, d, DP, wave, lane);  // q = Post . h
      __syncthreads();
      tqq = tq;
    }
    // optional projection after the intersection (3-chain_inter, model.py:107)
    float* tqpre = tqq;  // q before the final projection (needed by its backward)
    if (b->n_final) {
      if (DEC == DEC_BILINEAR) {
        if (BWD) {
          for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
            const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
            vstore<NC>(scratch_row(e, b->slot_fx, r), vload<NC>(tqq + r * DP, d, lane), d, lane);
          }
        }
        __syncthreads();
        tile_matmul<false>(tx, params + b->final_param, tqq, d, DP, wave, lane);
        __syncthreads();
        tqq = tx;
      } else {
        Vec<NC> w = vload<NC>(params + b->final_param, d, lane);
        for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
          const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
          Vec<NC> x = vload<NC>(tqq + r * DP, d, lane);
          VEC_OP(x, (DEC == DEC_DIAG) ? x.v[c] * w.v[c] : x.v[c] + w.v[c]);
          vstore<NC>(tx + r * DP, x, d, lane);
        }
        tqq = tx;
      }
    }
    // ---- scores, hinge, gradient seeds (own rows; no cross-wave traffic) ----
    float* tg = (tqq == tx) ? ty : tx;  // free tile for g_q
    for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
      const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
      const int q = e.q0 + r;
      Vec<NC> qv = vload<NC>(tqq + r * DP, d, lane);
      const float nq = fmaxf(sqrtf(vdot<NC>(qv, qv)), COS_EPS);
      float ntp, ntn = 1.f;
      int rtp, rtn = -1;
      Vec<NC> tp = gather_norm<NC>(e, b->target_table, itarget, r, ntp, rtp);
      Vec<NC> tn = vzero<NC>();
      if (b->has_neg) tn = gather_norm<NC>(e, b->target_table, ineg, r, ntn, rtn);
      const float ncp = fmaxf(sqrtf(vdot<NC>(tp, tp)), COS_EPS);
      const float ncn = fmaxf(sqrtf(vdot<NC>(tn, tn)), COS_EPS);
      const float sp = vdot<NC>(tp, qv) / (ncp * nq);
      const float sn = b->has_neg ? vdot<NC>(tn, qv) / (ncn * nq) : 0.f;
      if (q < B && lane == 0) {
        if (pos_out) pos_out[b->out_offset + q] = sp;
        if (neg_out && b->has_neg) neg_out[b->out_offset + q] = sn;
      }
      if (!BWD) continue;
      const float hinge = b->margin - (sp - sn);
      const bool act = (q < B) && hinge > 0.f;
      if (act) loss_part += hinge;
      const float cp = act ? -gscale : 0.f, cn = act ? gscale : 0.f;
      Vec<NC> gq, gtp, gtn;
      VEC_OP(gq, cp * (tp.v[c] / (ncp * nq) - sp * qv.v[c] / (nq * nq)) +
                     cn * (tn.v[c] / (ncn * nq) - sn * qv.v[c] / (nq * nq)));
      vstore<NC>(tg + r * DP, gq, d, lane);
      if (act) {
        VEC_OP(gtp, cp * (qv.v[c] / (ncp * nq) - sp * tp.v[c] / (ncp * ncp)));
        VEC_OP(gtn, cn * (qv.v[c] / (ncn * nq) - sn * tn.v[c] / (ncn * ncn)));
        scatter_norm_bwd<NC>(e, b->target_table, rtp, tp, ntp, gtp);
        scatter_norm_bwd<NC>(e, b->target_table, rtn, tn, ntn, gtn);
      }
    }
    if (BWD) {
      // ---- backward of the final projection ----
      float* tfree = (tg == tx) ? ty : ((tqq == tx) ? tx : ty);  // a tile not holding g (tqq is dead now)
      if (b->n_final) {
        if (DEC == DEC_BILINEAR) {
          for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
            const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
            vstore<NC>(scratch_row(e, b->slot_fg, r), vload<NC>(tg + r * DP, d, lane), d, lane);
          }
          __syncthreads();
          tile_matmul<true>(tfree, params + b->final_param, tg, d, DP, wave, lane);
          __syncthreads();
          float* tmp = tg;
          tg = tfree;
          tfree = tmp;
        } else {
          Vec<NC> w = vload<NC>(params + b->final_param, d, lane);
          Vec<NC> gw = vzero<NC>();
          for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
            const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
            Vec<NC> g = vload<NC>(tg + r * DP, d, lane);
            if (DEC == DEC_DIAG) {
              Vec<NC> qp = vload<NC>(tqpre + r * DP, d, lane);
              VEC_OP(gw, gw.v[c] + g.v[c] * qp.v[c]);
              VEC_OP(g, g.v[c] * w.v[c]);
              vstore<NC>(tg + r * DP, g, d, lane);
            } else {
              VEC_OP(gw, gw.v[c] + g.v[c]);
            }
          }
          flush_vec_grad<NC>(e, red, b->final_param, gw);
        }
      }
      // ---- backward of Post ----
      float* tgh = tg;  // grad wrt h (MLP) or wrt q (simple)
      if (MLP) {
        for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
          const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
          vstore<NC>(scratch_row(e, b->slot_gq, r), vload<NC>(tg + r * DP, d, lane), d, lane);
        }
        __syncthreads();
        tile_matmul<true>(tq, params + b->post_param, tg, d, DP, wave, lane);  // g_h = Post^T g_q
        __syncthreads();
        tgh = tq;
      }
      // tiles still needed: tgh, tmeta.  free: everything else except tgh.
      float* ta = (tgh == t0) ? t1 : t0;
      float* tb = (tgh == t0 || tgh == t1) ? t2 : t1;
      if (ta == tgh || tb == tgh || ta == tb) {  // defensive; cannot happen with 4 tiles
        ta = t2;
        tb = (tgh == t3) ? t1 : t3;
      }
      Vec<NC> gw_hop[GQE_MAX_HOPS];
      for (int i = 0; i < n; ++i) {
        // g wrt z_i (MLP) or e_i (simple), own rows
        for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
          const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const int j = lane + 64 * c;
            if (j >= d) continue;
            const float gh = tgh[r * DP + j];
            const int meta = tmeta[r * DP + j];
            float g;
            if (inter_min)
              g = ((meta & 3) == i) ? gh : 0.f;
            else
              g = gh / (float)n;
            if (MLP && !((meta >> (4 + i)) & 1)) g = 0.f;  // relu'(z) = [z > 0]
            ta[r * DP + j] = g;
          }
          if (MLP) vstore<NC>(scratch_row(e, b->slot_gz[i], r), vload<NC>(ta + r * DP, d, lane), d, lane);
        }
        float* tge = ta;
        if (MLP) {
          __syncthreads();
          tile_matmul<true>(tb, params + b->pre_param, ta, d, DP, wave, lane);  // g_e = Pre^T g_z
          __syncthreads();
          tge = tb;
        }
        const int nh = b->n_hops[i];
        if (DEC == DEC_BILINEAR) {
          float* tcur = tge;
          float* tnext = (tge == ta) ? tb : ta;
          for (int h = nh - 1; h >= 0; --h) {
            for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
              const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
              vstore<NC>(scratch_row(e, b->slot_gy[i][h], r), vload<NC>(tcur + r * DP, d, lane), d, lane);
            }
            __syncthreads();
            tile_matmul<true>(tnext, params + b->hop_param[i][h], tcur, d, DP, wave, lane);
            __syncthreads();
            float* tmp = tcur;
            tcur = tnext;
            tnext = tmp;
          }
          for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
            const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
            if (e.q0 + r >= B) continue;
            float nrm;
            int row;
            Vec<NC> x = gather_norm<NC>(e, b->anchor_table[i], ianchor + (size_t)i * B, r, nrm, row);
            Vec<NC> g = vload<NC>(tcur + r * DP, d, lane);
            scatter_norm_bwd<NC>(e, b->anchor_table[i], row, x, nrm, g);
          }
          __syncthreads();  // ta/tb are rewritten by the next branch
        } else {
          Vec<NC> w[GQE_MAX_HOPS];
          for (int h = 0; h < nh; ++h) {
            w[h] = vload<NC>(params + b->hop_param[i][h], d, lane);
            gw_hop[h] = vzero<NC>();
          }
          for (int rr = 0; rr < GQE_TQ / GQE_WAVES; ++rr) {
            const int r = wave * (GQE_TQ / GQE_WAVES) + rr;
            if (e.q0 + r >= B) continue;
            float nrm;
            int row;
            Vec<NC> x = gather_norm<NC>(e, b->anchor_table[i], ianchor + (size_t)i * B, r, nrm, row);
            Vec<NC> g = vload<NC>(tge + r * DP, d, lane);
            if (DEC == DEC_DIAG) {
              // inputs of the hops: x_0 = xhat, x_{h+1} = x_h (.) w_h ; walk back from the last hop
              Vec<NC> xin[GQE_MAX_HOPS];
              xin[0] = x;
              for (int h = 1; h < nh; ++h) VEC_OP(xin[h], xin[h - 1].v[c] * w[h - 1].v[c]);
              for (int h = nh - 1; h >= 0; --h) {
                VEC_OP(gw_hop[h], gw_hop[h].v[c] + g.v[c] * xin[h].v[c]);
                VEC_OP(g, g.v[c] * w[h].v[c]);
              }
            } else {
              for (int h = 0; h < nh; ++h) VEC_OP(gw_hop[h], gw_hop[h].v[c] + g.v[c]);
            }
            scatter_norm_bwd<NC>(e, b->anchor_table[i], row, x, nrm, g);
          }
          for (int h = 0; h < nh; ++h) flush_vec_grad<NC>(e, red, b->hop_param[i][h], gw_hop[h]);
          __syncthreads();
        }
      }
    }
  }
  if (BWD) {
    // mean hinge loss of the batch (model.py:124-126) and the weighted iteration loss
    if (lane == 0 && loss_part != 0.f) {
      const float l = loss_part * b->inv_B;
      unsafeAtomicAdd(losses + bi, l);
      unsafeAtomicAdd(losses + n_batches, l * b->loss_weight);
    }
  }
}

// ------------------------------------------------------------------------------------------
// deferred matrix gradients:  dM[i][j] += sum_b L[b][i] * R[b][j]   (rank-B update on the matrix cores)
// one wave per (job, 16x16 output tile, K chunk); A[i][k=b] = L[b][i0+i], B[k=b][j] = R[b][j0+j]:
// both operands are read straight from the scratch rows (64 B contiguous per 16 lanes).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GQE_THREADS) void gqe_pair_gemm_kernel(const GqeGemmJob* __restrict__ jobs, int n_units,
                                                                   const float* __restrict__ ws,
                                                                   float* __restrict__ grads, int d) {
  const int wave_global = (int)blockIdx.x * GQE_WAVES + (threadIdx.x >> 6);
  if (wave_global >= n_units) return;
  const int lane = threadIdx.x & 63;
  const int lq = lane & 15, lk = lane >> 4;
  const int tiles_per_dim = d / 16;
  const int tiles = tiles_per_dim * tiles_per_dim;
  // unit -> (job, chunk, tile): jobs carry a prefix sum of their units
  int lo = 0;
  while (jobs[lo].unit_end <= wave_global) ++lo;
  const GqeGemmJob* job = jobs + lo;
  const int u = wave_global - job->unit_begin;
  const int chunk = u / tiles;
  const int t = u % tiles;
  const int i0 = (t / tiles_per_dim) * 16, j0 = (t % tiles_per_dim) * 16;
  const int k_begin = chunk * GQE_GEMM_KCHUNK;
  const int k_end = min(job->K, k_begin + GQE_GEMM_KCHUNK);
  const float* L = ws + job->L_off;
  const float* R = ws + job->R_off;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = k_begin; k0 < k_end; k0 += 16) {
    const float* lp = L + (size_t)(k0 + 4 * lk) * d + i0 + lq;
    const float* rp = R + (size_t)(k0 + 4 * lk) * d + j0 + lq;
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(lp[s * d], rp[s * d], acc, 0, 0, 0);
  }
  float* out = grads + job->param_off + (size_t)(i0 + 4 * lk) * d + j0 + lq;
#pragma unroll
  for (int r = 0; r < 4; ++r) unsafeAtomicAdd(out + (size_t)r * d, acc[r]);
}

// ------------------------------------------------------------------------------------------
// fused optimiser passes over a list of parameter tensors (segments), 1024 floats per chunk
// ------------------------------------------------------------------------------------------
#define OPT_ADAM 0
#define OPT_SGD 1
#define OPT_ZERO 2

template <int MODE>
__global__ __launch_bounds__(GQE_THREADS) void gqe_opt_kernel(const GqeDevSeg* __restrict__ segs, int n_segs,
                                                             long long total_chunks, float* __restrict__ p,
                                                             float* __restrict__ g, float* __restrict__ m,
                                                             float* __restrict__ v, float lr, float b1, float b2,
                                                             float eps) {
  __shared__ long long s_begin[GQE_MAX_SEGS + 1];
  for (int i = threadIdx.x; i <= n_segs; i += blockDim.x) s_begin[i] = (i < n_segs) ? segs[i].chunk_begin : total_chunks;
  __syncthreads();
  int si = 0;
  for (long long ch = blockIdx.x; ch < total_chunks; ch += gridDim.x) {
    while (s_begin[si + 1] <= ch) ++si;  // chunks are visited in increasing order
    const GqeDevSeg sg = segs[si];
    const long long e0 = (ch - sg.chunk_begin) * GQE_OPT_CHUNK + (long long)threadIdx.x * 4;
    if (e0 >= sg.numel) continue;
    const long long off = sg.offset + e0;
    if (e0 + 4 <= sg.numel) {
      float4 gg = *reinterpret_cast<const float4*>(g + off);
      if (MODE == OPT_ZERO) {
        *reinterpret_cast<float4*>(g + off) = make_float4(0.f, 0.f, 0.f, 0.f);
        continue;
      }
      float4 pp = *reinterpret_cast<const float4*>(p + off);
      if (MODE == OPT_ADAM) {
        float4 mm = *reinterpret_cast<const float4*>(m + off);
        float4 vv = *reinterpret_cast<const float4*>(v + off);
#define ADAM1(x)                                              \
  mm.x = mm.x + (1.f - b1) * (gg.x - mm.x);                   \
  vv.x = vv.x * b2 + (1.f - b2) * gg.x * gg.x;                \
  pp.x = pp.x - sg.step_size * (mm.x / (sqrtf(vv.x) / sg.bc2_sqrt + eps));
        ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
        *reinterpret_cast<float4*>(m + off) = mm;
        *reinterpret_cast<float4*>(v + off) = vv;
      } else {
        pp.x -= lr * gg.x;
        pp.y -= lr * gg.y;
        pp.z -= lr * gg.z;
        pp.w -= lr * gg.w;
      }
      *reinterpret_cast<float4*>(p + off) = pp;
      *reinterpret_cast<float4*>(g + off) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      for (long long k = e0; k < sg.numel; ++k) {
        const long long o = sg.offset + k;
        const float gg = g[o];
        g[o] = 0.f;
        if (MODE == OPT_ZERO) continue;
        if (MODE == OPT_ADAM) {
          const float mm = m[o] + (1.f - b1) * (gg - m[o]);
          const float vv = v[o] * b2 + (1.f - b2) * gg * gg;
          m[o] = mm;
          v[o] = vv;
          p[o] = p[o] - sg.step_size * (mm / (sqrtf(vv) / sg.bc2_sqrt + eps));
        } else {
          p[o] -= lr * gg;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// launchers (called from gqe_host.cpp)
// ------------------------------------------------------------------------------------------
template <int DEC, bool MLP, int NC>
static hipError_t launch_fused_nc(bool bwd, int tiles, size_t lds, hipStream_t st, const GqeDevBatch* db, int nb,
                                  const float* params, float* grads, float* ws, const int32_t* idx, int d,
                                  float* losses, float* pos, float* neg, int inter_min) {
  if (bwd) {
    hipLaunchKernelGGL((gqe_fused_kernel<DEC, MLP, NC, true>), dim3(tiles), dim3(GQE_THREADS), lds, st, db, nb, params,
                       grads, ws, idx, d, losses, pos, neg, inter_min);
  } else {
    hipLaunchKernelGGL((gqe_fused_kernel<DEC, MLP, NC, false>), dim3(tiles), dim3(GQE_THREADS), lds, st, db, nb,
                       params, grads, ws, idx, d, losses, pos, neg, inter_min);
  }
  return hipGetLastError();
}

template <int DEC, bool MLP>
static hipError_t launch_fused_dm(int nc, bool bwd, int tiles, size_t lds, hipStream_t st, const GqeDevBatch* db,
                                  int nb, const float* params, float* grads, float* ws, const int32_t* idx, int d,
                                  float* losses, float* pos, float* neg, int inter_min) {
  switch (nc) {
    case 1: return launch_fused_nc<DEC, MLP, 1>(bwd, tiles, lds, st, db, nb, params, grads, ws, idx, d, losses, pos, neg, inter_min);
    case 2: return launch_fused_nc<DEC, MLP, 2>(bwd, tiles, lds, st, db, nb, params, grads, ws, idx, d, losses, pos, neg, inter_min);
    case 3: return launch_fused_nc<DEC, MLP, 3>(bwd, tiles, lds, st, db, nb, params, grads, ws, idx, d, losses, pos, neg, inter_min);
    default: return launch_fused_nc<DEC, MLP, 4>(bwd, tiles, lds, st, db, nb, params, grads, ws, idx, d, losses, pos, neg, inter_min);
  }
}

size_t gqe_fused_lds_bytes(int d) {
  const int DP = d + 4;
  return (size_t)(5 * GQE_TQ * DP + GQE_WAVES * d + 9 * GQE_TQ) * sizeof(float);
}

hipError_t gqe_launch_fused(int dec, int mlp, int inter_min, bool bwd, int tiles, hipStream_t st, const GqeDevBatch* db,
                            int nb, const float* params, float* grads, float* ws, const int32_t* idx, int d,
                            float* losses, float* pos, float* neg) {
  const int nc = (d + 63) / 64;
  const size_t lds = gqe_fused_lds_bytes(d);
#define GO(DEC, MLP) return launch_fused_dm<DEC, MLP>(nc, bwd, tiles, lds, st, db, nb, params, grads, ws, idx, d, losses, pos, neg, inter_min)
  if (dec == DEC_DIAG) {
    if (mlp) GO(DEC_DIAG, true); else GO(DEC_DIAG, false);
  } else if (dec == DEC_TRANSE) {
    if (mlp) GO(DEC_TRANSE, true); else GO(DEC_TRANSE, false);
  } else {
    if (mlp) GO(DEC_BILINEAR, true); else GO(DEC_BILINEAR, false);
  }
#undef GO
}

hipError_t gqe_launch_pair_gemm(int n_units, hipStream_t st, const GqeGemmJob* jobs, const float* ws, float* grads, int d) {
  const int blocks = (n_units + GQE_WAVES - 1) / GQE_WAVES;
  hipLaunchKernelGGL(gqe_pair_gemm_kernel, dim3(blocks), dim3(GQE_THREADS), 0, st, jobs, n_units, ws, grads, d);
  return hipGetLastError();
}

hipError_t gqe_launch_opt(int mode, hipStream_t st, const GqeDevSeg* segs, int n_segs, long long total_chunks, float* p,
                          float* g, float* m, float* v, float lr, float b1, float b2, float eps) {
  long long blocks = total_chunks < 4096 ? total_chunks : 4096;
  if (blocks < 1) blocks = 1;
  if (mode == OPT_ADAM)
    hipLaunchKernelGGL(gqe_opt_kernel<OPT_ADAM>, dim3((unsigned)blocks), dim3(GQE_THREADS), 0, st, segs, n_segs, total_chunks, p, g, m, v, lr, b1, b2, eps);
  else if (mode == OPT_SGD)
    hipLaunchKernelGGL(gqe_opt_kernel<OPT_SGD>, dim3((unsigned)blocks), dim3(GQE_THREADS), 0, st, segs, n_segs, total_chunks, p, g, m, v, lr, b1, b2, eps);
  else
    hipLaunchKernelGGL(gqe_opt_kernel<OPT_ZERO>, dim3((unsigned)blocks), dim3(GQE_THREADS), 0, st, segs, n_segs, total_chunks, p, g, m, v, lr, b1, b2, eps);
  return hipGetLastError();
}
