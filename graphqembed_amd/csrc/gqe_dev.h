// gqe_dev.h — structs shared by the host side (gqe_host.cpp) and the kernels (gqe_kernels.hip).
#ifndef GQE_DEV_H
#define GQE_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gqe.h"

#define GQE_TQ 16            // queries per tile (= per workgroup)
#define GQE_WAVES 4          // wave64s per workgroup
#define GQE_THREADS 256
#define GQE_OPT_CHUNK 1024   // floats per optimiser chunk (256 threads x float4)
#define GQE_MAX_SEGS 96
#define GQE_GEMM_KCHUNK 128  // queries per pair-GEMM unit

// Device-side view of one batch (gqe_batch + launch geometry + scratch slots).
struct GqeDevBatch {
  int32_t qtype, B, n_anchors, idx_offset;
  int32_t tile_begin, out_offset, has_neg, n_final;
  int32_t n_hops[GQE_MAX_BRANCH];
  int32_t Bpad;
  int64_t target_table;
  int64_t anchor_table[GQE_MAX_BRANCH];
  int64_t hop_param[GQE_MAX_BRANCH][GQE_MAX_HOPS];
  int64_t final_param, pre_param, post_param;
  int64_t scratch_base;  // float offset of this batch's scratch rows in the workspace
  float margin, grad_scale, inv_B, loss_weight;
  // scratch slots (row blocks of Bpad x d floats); -1 = unused
  int32_t slot_x[GQE_MAX_BRANCH][GQE_MAX_HOPS];   // bilinear: input of hop h of branch i
  int32_t slot_gy[GQE_MAX_BRANCH][GQE_MAX_HOPS];  // bilinear: grad wrt output of hop h of branch i
  int32_t slot_e[GQE_MAX_BRANCH];                 // MLP: e_i (input of Pre)
  int32_t slot_gz[GQE_MAX_BRANCH];                // MLP: grad wrt Pre.e_i
  int32_t slot_hh, slot_gq;                       // MLP: h (input of Post), grad wrt q
  int32_t slot_fx, slot_fg;                       // bilinear final projection: input, grad wrt output
  int32_t slot_act[2][GQE_MAX_HOPS];              // bilinear chain: act_h of the +/- side
  int32_t slot_gact[2][GQE_MAX_HOPS];             // bilinear chain: grad wrt act_{h+1}
};

// dM += L^T R over K rows (L, R: float offsets into the workspace, rows of d floats)
struct GqeGemmJob {
  int64_t param_off, L_off, R_off;
  int32_t K, unit_begin, unit_end, pad;
};

struct GqeDevSeg {
  int64_t offset, numel, chunk_begin;
  float step_size, bc2_sqrt;
};

size_t gqe_fused_lds_bytes(int d);
hipError_t gqe_launch_fused(int dec, int mlp, int inter_min, bool bwd, int tiles, hipStream_t st, const GqeDevBatch* db,
                            int nb, const float* params, float* grads, float* ws, const int32_t* idx, int d,
                            float* losses, float* pos, float* neg);
hipError_t gqe_launch_pair_gemm(int n_units, hipStream_t st, const GqeGemmJob* jobs, const float* ws, float* grads, int d);
hipError_t gqe_launch_opt(int mode, hipStream_t st, const GqeDevSeg* segs, int n_segs, long long total_chunks, float* p,
                          float* g, float* m, float* v, float lr, float b1, float b2, float eps);

#endif
