// gqe_fused_inst.hip — one translation unit per (decoder, MLP-intersection, waves per workgroup) variant of the fused
// kernel: compiled with -DGQE_DEC={0,1,2} -DGQE_MLP={0,1} -DGQE_FW={16,8} so the variants build in parallel.
#include "gqe_fused.h"

#define GQE_CAT2(a, b, c, d) a##b##_##c##_w##d
#define GQE_CAT(a, b, c, d) GQE_CAT2(a, b, c, d)

hipError_t GQE_CAT(gqe_launch_fused_, GQE_DEC, GQE_MLP, GQE_FW)(const GqeFusedArgs& a) {
  return launch_fused_dm<GQE_DEC, (GQE_MLP != 0)>(a);
}
