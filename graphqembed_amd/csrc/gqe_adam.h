// gqe_adam.h — the ONE definition of an Adam step of one element, shared by the eager optimiser pass, the lazy row
// launches (gqe_kernels.hip) and the replay inside the fused kernel's gather (gqe_fused.h).
#ifndef GQE_ADAM_H
#define GQE_ADAM_H
#include <hip/hip_runtime.h>

// One Adam step of one element, torch.optim.Adam's formulas (exp_avg.lerp_(g, 1-b1); exp_avg_sq = b2 v + (1-b2) g g;
// p -= step_size * exp_avg / (sqrt(exp_avg_sq) / bc2_sqrt + eps)).  Every operation is spelled out — fused
// multiply-adds where they are wanted, the hardware's v_sqrt_f32 / v_rcp_f32 (1 ulp) for the root and the two
// divisions — so that the eager pass and the replay of deferred steps (lazy rows, below) execute the SAME
// instruction sequence and agree bit for bit; left to the compiler, two call sites may contract differently.  The
// 1-ulp primitives keep a replayed step at ~15 instructions per element (the IEEE expansions are 3x that and sit
// on the critical path of a row that owes dozens of steps); against torch's IEEE result the update differs in the
// last bit or two, far inside the tolerance of every parity test (and of fp32 training itself).
__device__ __forceinline__ void gqe_adam1(float& p, float& m, float& v, float g, float step_size, float inv_bc2, float b1c,
                                      float b2, float b2c, float eps) {
  m = __fmaf_rn(b1c, __fsub_rn(g, m), m);
  v = __fmaf_rn(__fmul_rn(b2c, g), g, __fmul_rn(v, b2));
  const float den = __fmaf_rn(__builtin_amdgcn_sqrtf(v), inv_bc2, eps);
  p = __fmaf_rn(-step_size, __fmul_rn(m, __builtin_amdgcn_rcpf(den)), p);
}

#endif
