// gqe_fused_inst.hip — one translation unit per (decoder, MLP-intersection) variant of the fused kernel:
// compiled 6 times with -DGQE_DEC={0,1,2} -DGQE_MLP={0,1} so the variants build in parallel.
#include "gqe_fused.h"

#define GQE_CAT2(a, b, c) a##b##_##c
#define GQE_CAT(a, b, c) GQE_CAT2(a, b, c)

hipError_t GQE_CAT(gqe_launch_fused_, GQE_DEC, GQE_MLP)(const GqeFusedArgs& a) {
  return launch_fused_dm<GQE_DEC, (GQE_MLP != 0)>(a);
}
