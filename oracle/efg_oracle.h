/*
 * oracle/efg_oracle.h -- CPU restatement of the EFG hot path (TEST INFRASTRUCTURE ONLY).
 *
 * Nothing under oracle/ is on the product path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library, and only as the checker.
 * Each function cites the reference file:line (relative to /root/reference) whose
 * algorithm it restates.  Parity status per function is stated in efg_oracle.c.
 */
#ifndef EFG_ORACLE_H
#define EFG_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- voxelization (efg/operators/src/voxelize/voxelization_cpu.cpp) ---- */
void oracle_dynamic_voxelize(const float* points, int64_t n, int f, const float voxel_size[3],
                             const float coors_range[6], int32_t* coors /*[n,3] zyx*/);
int oracle_hard_voxelize(const float* points, int64_t n, int f, const float voxel_size[3],
                         const float coors_range[6], int max_points, int max_voxels,
                         float* voxels /*[max_voxels,max_points,f] pre-zeroed*/,
                         int32_t* coors /*[max_voxels,3] pre-zeroed*/,
                         int32_t* num_points_per_voxel /*[max_voxels] pre-zeroed*/);
/* VoxelMeanFeatureExtractor.forward (efg/modeling/readers/voxel_reader.py:14-19) */
void oracle_voxel_mean(const float* voxels, const int32_t* npv, int64_t m, int max_points, int f,
                       int nfeat, float* out /*[m,nfeat]*/);

/* ---- dynamic scatter (efg/operators/src/voxelize/scatter_points_cuda.cu) ---- */
/* reduce: 0 = sum, 1 = mean, 2 = max (reduce_t, scatter_points_cuda.cu:9).  Outputs sized
 * for the worst case (n rows); returns M. */
int64_t oracle_scatter_forward(const float* feats, const int32_t* coors, int64_t n, int c, int ndim,
                               int reduce, float* voxel_feats, int32_t* voxel_coors,
                               int32_t* point2voxel, int32_t* count);
void oracle_scatter_backward(float* grad_feats /*[n,c]*/, const float* grad_voxel, const float* feats,
                             const float* voxel_feats, const int32_t* point2voxel,
                             const int32_t* count, int64_t n, int64_t m, int c, int reduce);

/* ---- sparse convolution (third-party spconv; contract = SURVEY.md B.6) ---- */
/* indices are (b,z,y,x) int32.  Output order is canonical: ascending linear index
 * ((b*D+z)*H+y)*W+x over the OUTPUT grid.  nbr is [kvol][m_out] int32 (input row or -1),
 * kernel offset index = (kz*KH + ky)*KW + kx. */
int64_t oracle_spconv_out_indices(const int32_t* in_idx, int64_t m_in, int batch, const int in_shape[3],
                                  const int ksize[3], const int stride[3], const int pad[3],
                                  int32_t* out_idx /*cap rows*/, int64_t cap, int out_shape[3]);
void oracle_spconv_rulebook(const int32_t* in_idx, int64_t m_in, const int32_t* out_idx, int64_t m_out,
                            int batch, const int in_shape[3], const int ksize[3], const int stride[3],
                            const int pad[3], int32_t* nbr /*[kvol][m_out]*/);
/* weight layout [cout][kvol][cin] (spconv 2.x "KRSC").  bias may be NULL. */
void oracle_spconv_forward(const float* in_feat, int64_t m_in, int cin, const float* weight,
                           const float* bias, int cout, int kvol, const int32_t* nbr, int64_t m_out,
                           float* out_feat);
void oracle_spconv_dgrad(const float* grad_out, int64_t m_out, int cout, const float* weight, int cin,
                         int kvol, const int32_t* nbr, int64_t m_in, float* grad_in);
void oracle_spconv_wgrad(const float* in_feat, int64_t m_in, int cin, const float* grad_out,
                         int64_t m_out, int cout, int kvol, const int32_t* nbr, float* grad_w);
void oracle_sparse_to_dense(const float* feat, const int32_t* idx, int64_t m, int c, int batch,
                            const int shape[3], float* dense /*[b,c,d,h,w] pre-zeroed*/);

/* ---- box / multi-scale deformable attention (box_attn_kernel.cuh:34-184,274-349) ---- */
void oracle_msda_forward(const float* value, const int64_t* shapes, const int64_t* level_start,
                         const float* loc, const float* attn, int b, int s, int h, int d, int l,
                         int lq, int p, float* out /*[b,lq,h*d]*/);
void oracle_msda_backward(const float* value, const int64_t* shapes, const int64_t* level_start,
                          const float* loc, const float* attn, const float* grad_out, int b, int s,
                          int h, int d, int l, int lq, int p, float* grad_value, float* grad_loc,
                          float* grad_attn /* all pre-zeroed */);

/* ---- next row n1: rotated BEV IoU / NMS (efg/operators/src/iou3d_nms/iou3d_nms_kernel.cu) ---- */
float oracle_box_overlap(const float* box_a, const float* box_b);
float oracle_iou_bev(const float* box_a, const float* box_b);
void oracle_boxes_bev(const float* a, int na, const float* b, int nb, int mode, float* out);
int oracle_nms(const float* boxes, int n, float thresh, int rotated, int64_t* keep);

/* ---- next row "GPU matcher": scipy.optimize.linear_sum_assignment ($CQ/modules/matcher.py:89) ---- */
int oracle_lsap(const float* cost, int nq, int g_stride, int ng, int64_t* query_of_gt);

#ifdef __cplusplus
}
#endif
#endif
