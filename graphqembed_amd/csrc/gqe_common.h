// gqe_common.h — device helpers shared by the gfx950 kernels.
#ifndef GQE_COMMON_H
#define GQE_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gqe_dev.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DEC_DIAG 0
#define DEC_TRANSE 1
#define DEC_BILINEAR 2
#define COS_EPS 1e-8f

// ------------------------------------------------------------------------------------------
// vector-of-a-row helpers: a wave owns a row of d floats, lane l holds elements j = l + 64*c.
// When d == 64*NC is a compile-time fact (FULL kernels) the j < d guards fold away.
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

// ------------------------------------------------------------------------------------------
// A row of d floats in GLOBAL memory seen by a wave (lane l <-> elements l + 64 c) when d is NOT a multiple of 64 ("guarded"
// kernels, FULL = false): buffer instructions with the ROW as the buffer (num_records = 4 d bytes).  Lanes past the end of the
// row read 0 and store nothing — decided by the hardware's range check with all of EXEC enabled, where `if (j < d)` made
// every access a lane-divergent branch: more live registers, and the join blocks of those branches are where ROCm 7.2's
// register allocator placed spill code ahead of the EXEC restore — the reason dims were refused.
// INVARIANT the guarded kernels rely on: the lanes past d of every Vec hold 0, and so do the columns past d of every LDS tile
// (tiles are [16][64 NC + 4]); element-wise work then needs no guard at all.
// The row pointer has to be wave-uniform (it becomes the SGPR descriptor); FULL = true: plain loads / stores, d == 64 NC.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gqe_row_rsrc(const float* p, int d) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, d * 4, 0x00020000);   // raw buffer, 32-bit data format
}

template <int NC, bool FULL>
__device__ __forceinline__ Vec<NC> gload(const float* p, int d, int lane) {
  Vec<NC> r;
  if (FULL) {
#pragma unroll
    for (int c = 0; c < NC; ++c) r.v[c] = p[lane + 64 * c];
  } else {
    const __amdgpu_buffer_rsrc_t rs = gqe_row_rsrc(p, d);
#pragma unroll
    for (int c = 0; c < NC; ++c) r.v[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (lane + 64 * c) * 4, 0, 0));
  }
  return r;
}

// aux: 0 default policy, 16 = sc1 (written through: what a kernel on another XCD reads next)
template <int NC, bool FULL>
__device__ __forceinline__ void gstore(float* p, const Vec<NC>& x, int d, int lane) {
  if (FULL) {
#pragma unroll
    for (int c = 0; c < NC; ++c) p[lane + 64 * c] = x.v[c];
  } else {
    const __amdgpu_buffer_rsrc_t rs = gqe_row_rsrc(p, d);
#pragma unroll
    for (int c = 0; c < NC; ++c) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x.v[c]), rs, (lane + 64 * c) * 4, 0, 0);
  }
}

template <int NC, bool FULL>
__device__ __forceinline__ void gstore_sc1(float* p, const Vec<NC>& x, int d, int lane) {
  if (FULL) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int j = lane + 64 * c;
      asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p + j), "v"(x.v[c]) : "memory");
    }
  } else {
    const __amdgpu_buffer_rsrc_t rs = gqe_row_rsrc(p, d);
#pragma unroll
    for (int c = 0; c < NC; ++c) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x.v[c]), rs, (lane + 64 * c) * 4, 0, 16);
  }
}

template <int NC, bool FULL>
__device__ __forceinline__ void gatomic_add(float* p, const Vec<NC>& x, int d, int lane) {
  if (FULL) {
#pragma unroll
    for (int c = 0; c < NC; ++c) unsafeAtomicAdd(p + lane + 64 * c, x.v[c]);
  } else {
    const __amdgpu_buffer_rsrc_t rs = gqe_row_rsrc(p, d);
#pragma unroll
    for (int c = 0; c < NC; ++c) (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(x.v[c], rs, (lane + 64 * c) * 4, 0, 0);
  }
}

// a row of an LDS tile [16][64 NC + 4]: never guarded (see the invariant above)
template <int NC>
__device__ __forceinline__ Vec<NC> lload(const float* t, int lane) {
  Vec<NC> r;
#pragma unroll
  for (int c = 0; c < NC; ++c) r.v[c] = t[lane + 64 * c];
  return r;
}

template <int NC>
__device__ __forceinline__ void lstore(float* t, const Vec<NC>& x, int lane) {
#pragma unroll
  for (int c = 0; c < NC; ++c) t[lane + 64 * c] = x.v[c];
}

// Wave-uniform scalar math of the fused kernel (norms, cosines): gfx950 has no scalar float ALU, so every sqrtf / division
// of a per-query scalar is a VALU sequence issued for the whole wave — 16 instructions for an IEEE sqrtf, 11 for a
// division, ~125 of the ~400 VALU instructions of an intersection tile's scoring phase, and with four waves per SIMD the
// vector phases are issue-bound.  v_sqrt_f32 / v_rcp_f32 are accurate to 1 ulp, far inside the parity tolerances
// (scores 1e-5 absolute against the fp64 oracle).
__device__ __forceinline__ float gqe_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float gqe_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

template <int NC>
__device__ __forceinline__ void vatomic_add(float* p, const Vec<NC>& x, int d, int lane) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int j = lane + 64 * c;
    if (j < d) unsafeAtomicAdd(p + j, x.v[c]);
  }
}

// wave64 all-reduce on the DPP path (no LDS crossbar): quad swaps, row mirrors, row broadcasts, then the
// total (lane 63) is broadcast through an SGPR with v_readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xF, false));
}

__device__ __forceinline__ float wave_sum(float x) {
  x += dpp_get<0xB1, 0xF>(x);   // quad_perm [1,0,3,2]
  x += dpp_get<0x4E, 0xF>(x);   // quad_perm [2,3,0,1]
  x += dpp_get<0x141, 0xF>(x);  // row_half_mirror
  x += dpp_get<0x140, 0xF>(x);  // row_mirror          -> every lane holds its row's sum
  x += dpp_get<0x142, 0xA>(x);  // row_bcast15 -> rows 1,3
  x += dpp_get<0x143, 0xC>(x);  // row_bcast31 -> rows 2,3 -> lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
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

#endif
