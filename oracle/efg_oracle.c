/*
 * oracle/efg_oracle.c -- plain-C CPU restatement of the EFG Voxel-DETR/ConQueR hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Loaded by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the *checker*; never by the product path (efg_amd/ fails loudly when
 * its HIP library is missing instead of falling back to this file).
 *
 * Parity status (what pins each restatement):
 *   voxelize (dynamic + hard)  PINNED  against the reference's own voxelization_cpu.cpp built
 *                                      in place as oracle/_ref/libefg_ref.so, against the numba
 *                                      twin efg/geometry/point_cloud_ops.py:5-53 and against the
 *                                      committed golden vectors tests/golden/voxelize_*.npz.
 *   msda / box attention       PINNED  against ms_deform_attn_core_pytorch
 *                                      (efg/operators/ms_deform_attn.py:55-76) golden vectors
 *                                      tests/golden/msda_*.npz (fwd) and autograd through it (bwd).
 *   dynamic scatter            FORWARD PINNED (voxel set, counts, point membership, max exactly, sum / mean vs an fp64
 *                                      sum of the reference's own groups) against the reference's
 *                                      dynamic_point_to_voxel_cpu (scatter_points_cpu.cpp:62-119) built in place into
 *                                      oracle/_ref (tests/test_oracle_voxelize.py); the sorted output ORDER and the
 *                                      BACKWARD are restated from scatter_points_cuda.cu:209-352 only (no CPU path in
 *                                      the reference): PARITY UNPINNED for those two.
 *   sparse convolution         PARITY UNPINNED: spconv (traveller59/spconv, PyPI spconv-cu11x,
 *                                      version not pinned by the reference: README.md:26-27,
 *                                      sparse_net.py:6-11) is absent from /root/reference and from
 *                                      this image.  Restated from its published semantics
 *                                       and checked against dense
 *                                      torch.nn.functional.conv3d on the call-site geometries of
 *                                      efg/modeling/backbones/sparse_net.py:79-98,120-165,273-282.
 */
#include "efg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * voxelization
 * ---------------------------------------------------------------------------------------- */

/* grid_size[i] = round((range[3+i]-range[i]) / voxel_size[i]) in float arithmetic
 * (voxelization_cpu.cpp:119-122; operands are std::vector<float>). */
static void grid_size_of(const float vs[3], const float cr[6], int grid[3]) {
  for (int i = 0; i < 3; ++i) grid[i] = (int)roundf((cr[3 + i] - cr[i]) / vs[i]);
}

/* voxelization_cpu.cpp:7-41 (dynamic_voxelize_kernel).  Per point, axes x,y,z in order:
 * c = floor((p - min) / vs) in fp32; out of [0,grid) -> all three coords -1 (CPU encoding).
 * NaN / out-of-int-range values are undefined behaviour in the reference; we define them
 * as "outside" (SURVEY.md B.1). */
static int point_coor(const float* p, const float vs[3], const float cr[6], const int grid[3],
                      int coor_zyx[3]) {
  for (int j = 0; j < 3; ++j) {
    float v = floorf((p[j] - cr[j]) / vs[j]);
    if (!(v >= 0.0f && v < (float)grid[j])) return 0;
    coor_zyx[2 - j] = (int)v;
  }
  return 1;
}

void oracle_dynamic_voxelize(const float* points, int64_t n, int f, const float vs[3],
                             const float cr[6], int32_t* coors) {
  int grid[3];
  grid_size_of(vs, cr, grid);
  for (int64_t i = 0; i < n; ++i) {
    int c[3];
    if (point_coor(points + i * f, vs, cr, grid, c)) {
      coors[i * 3 + 0] = c[0];
      coors[i * 3 + 1] = c[1];
      coors[i * 3 + 2] = c[2];
    } else {
      coors[i * 3 + 0] = coors[i * 3 + 1] = coors[i * 3 + 2] = -1;
    }
  }
}

/* voxelization_cpu.cpp:43-99 (hard_voxelize_kernel) + :105-142 (hard_voxelize_cpu);
 * identical to the numba loop efg/geometry/point_cloud_ops.py:5-53.
 * The reference keeps a dense coor_to_voxelidx map filled with -1; we keep the same dense
 * map but store voxelidx+1 in lazily-zeroed calloc pages (0 = unseen) so that only touched
 * pages are ever faulted in. */
int oracle_hard_voxelize(const float* points, int64_t n, int f, const float vs[3], const float cr[6],
                         int max_points, int max_voxels, float* voxels, int32_t* coors,
                         int32_t* npv) {
  int grid[3];
  grid_size_of(vs, cr, grid);
  const size_t cells = (size_t)grid[0] * (size_t)grid[1] * (size_t)grid[2];
  int32_t* map = (int32_t*)calloc(cells ? cells : 1, sizeof(int32_t));
  if (!map) return -1;
  int voxel_num = 0;
  for (int64_t i = 0; i < n; ++i) {
    int c[3];
    if (!point_coor(points + i * f, vs, cr, grid, c)) continue; /* :71 */
    const size_t cell = ((size_t)c[0] * grid[1] + c[1]) * grid[0] + c[2];
    int voxelidx = map[cell] - 1;
    if (voxelidx == -1) { /* :76-87 */
      voxelidx = voxel_num;
      if (max_voxels != -1 && voxel_num >= max_voxels) break;
      voxel_num += 1;
      map[cell] = voxelidx + 1;
      coors[voxelidx * 3 + 0] = c[0];
      coors[voxelidx * 3 + 1] = c[1];
      coors[voxelidx * 3 + 2] = c[2];
    }
    const int num = npv[voxelidx]; /* :90-96 */
    if (max_points == -1 || num < max_points) {
      memcpy(voxels + ((size_t)voxelidx * max_points + num) * f, points + i * f, sizeof(float) * f);
      npv[voxelidx] += 1;
    }
  }
  free(map);
  return voxel_num;
}

/* efg/modeling/readers/voxel_reader.py:14-19: sum over the max_points slots (zero padded)
 * of the first nfeat features, divided by the (capped) point count.  torch.sum over dim=1
 * of 5 elements is a sequential fp32 sum. */
void oracle_voxel_mean(const float* voxels, const int32_t* npv, int64_t m, int max_points, int f,
                       int nfeat, float* out) {
  for (int64_t v = 0; v < m; ++v)
    for (int k = 0; k < nfeat; ++k) {
      float s = 0.0f;
      for (int p = 0; p < max_points; ++p) s += voxels[((size_t)v * max_points + p) * f + k];
      out[v * nfeat + k] = s / (float)npv[v];
    }
}

/* ------------------------------------------------------------------------------------------
 * dynamic scatter  (scatter_points_cuda.cu:209-352)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int64_t key;
  int64_t idx;
} key_idx_t;

static int cmp_key_idx(const void* a, const void* b) {
  const key_idx_t *x = (const key_idx_t*)a, *y = (const key_idx_t*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

int64_t oracle_scatter_forward(const float* feats, const int32_t* coors, int64_t n, int c, int ndim,
                               int reduce, float* voxel_feats, int32_t* voxel_coors,
                               int32_t* point2voxel, int32_t* count) {
  if (n == 0) return 0;
  /* :220 coor_space_dim = coors.max(0) + 1 (per call, over ALL rows incl. -1 rows) */
  int64_t dim[8];
  for (int j = 0; j < ndim; ++j) {
    int32_t mx = coors[j];
    for (int64_t i = 1; i < n; ++i)
      if (coors[i * ndim + j] > mx) mx = coors[i * ndim + j];
    dim[j] = (int64_t)mx + 1;
  }
  /* :70-81 coors_id_kernel: row-major linearisation, -1 as soon as a coord is negative */
  key_idx_t* ki = (key_idx_t*)malloc(sizeof(key_idx_t) * (size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    int64_t id = 0;
    for (int j = 0; j < ndim && id != -1; ++j) {
      id *= dim[j];
      int64_t t = coors[i * ndim + j];
      id = (t < 0) ? -1 : id + t;
    }
    ki[i].key = id;
    ki[i].idx = i;
  }
  /* :237 argsort; :84-98 segment heads; :250 cumsum; :251 scatter back to point order */
  qsort(ki, (size_t)n, sizeof(key_idx_t), cmp_key_idx);
  int64_t m = 0;
  int32_t cur = -1;
  for (int64_t x = 0; x < n; ++x) {
    if (x == 0) cur = (ki[0].key == -1) ? -1 : 0;
    else if (ki[x - 1].key < ki[x].key) cur += 1;
    point2voxel[ki[x].idx] = cur;
  }
  m = (int64_t)cur + 1;
  free(ki);
  /* :255-284 feats_reduce_kernel */
  for (int64_t v = 0; v < m; ++v) {
    count[v] = 0;
    for (int k = 0; k < c; ++k) voxel_feats[v * c + k] = (reduce == 2) ? -INFINITY : 0.0f;
  }
  for (int64_t i = 0; i < n; ++i) {
    const int32_t v = point2voxel[i];
    if (v < 0) continue;
    for (int j = 0; j < ndim; ++j) voxel_coors[(int64_t)v * ndim + j] = coors[i * ndim + j];
    if (reduce == 2) {
      for (int k = 0; k < c; ++k)
        voxel_feats[(int64_t)v * c + k] = fmaxf(voxel_feats[(int64_t)v * c + k], feats[i * c + k]);
    } else {
      if (reduce == 1) count[v] += 1; /* count only filled for MEAN (:121-123) */
      for (int k = 0; k < c; ++k) voxel_feats[(int64_t)v * c + k] += feats[i * c + k];
    }
  }
  if (reduce == 1)
    for (int64_t v = 0; v < m; ++v)
      for (int k = 0; k < c; ++k) voxel_feats[v * c + k] /= (float)count[v];
  return m;
}

/* scatter_points_cuda.cu:292-352.  max: gradient goes to the LOWEST point index whose value
 * equals the voxel max (atomicMin at :180-184). */
void oracle_scatter_backward(float* grad_feats, const float* grad_voxel, const float* feats,
                             const float* voxel_feats, const int32_t* point2voxel,
                             const int32_t* count, int64_t n, int64_t m, int c, int reduce) {
  memset(grad_feats, 0, sizeof(float) * (size_t)n * c);
  if (reduce == 0 || reduce == 1) {
    for (int64_t i = 0; i < n; ++i) {
      const int32_t v = point2voxel[i];
      if (v < 0) continue;
      for (int k = 0; k < c; ++k) {
        float g = grad_voxel[(int64_t)v * c + k];
        grad_feats[i * c + k] = (reduce == 1) ? g / (float)count[v] : g;
      }
    }
  } else {
    int64_t* from = (int64_t*)malloc(sizeof(int64_t) * (size_t)(m * c ? m * c : 1));
    for (int64_t e = 0; e < m * c; ++e) from[e] = n;
    for (int64_t i = 0; i < n; ++i) {
      const int32_t v = point2voxel[i];
      if (v < 0) continue;
      for (int k = 0; k < c; ++k)
        if (feats[i * c + k] == voxel_feats[(int64_t)v * c + k] && i < from[(int64_t)v * c + k])
          from[(int64_t)v * c + k] = i;
    }
    for (int64_t e = 0; e < m * c; ++e)
      if (from[e] < n) grad_feats[from[e] * c + (e % c)] = grad_voxel[e];
    free(from);
  }
}

/* ------------------------------------------------------------------------------------------
 * sparse convolution (contract: SURVEY.md B.6; call sites sparse_net.py:79-98,120-165,273-309)
 * ---------------------------------------------------------------------------------------- */
static int cmp_i64(const void* a, const void* b) {
  int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return x < y ? -1 : (x > y);
}

static void out_shape_of(const int in_shape[3], const int k[3], const int s[3], const int p[3],
                         int out_shape[3]) {
  for (int a = 0; a < 3; ++a) out_shape[a] = (in_shape[a] + 2 * p[a] - k[a]) / s[a] + 1;
}

/* Regular SparseConv3d: output site o is active iff some active input i and kernel offset d
 * satisfy i = o*s - p + d.  Canonical order = ascending linear index over the output grid. */
int64_t oracle_spconv_out_indices(const int32_t* in_idx, int64_t m_in, int batch, const int in_shape[3],
                                  const int ks[3], const int st[3], const int pd[3],
                                  int32_t* out_idx, int64_t cap, int out_shape[3]) {
  (void)batch;
  out_shape_of(in_shape, ks, st, pd, out_shape);
  const int kvol = ks[0] * ks[1] * ks[2];
  int64_t* keys = (int64_t*)malloc(sizeof(int64_t) * (size_t)(m_in * kvol + 1));
  int64_t nk = 0;
  for (int64_t i = 0; i < m_in; ++i) {
    const int32_t* c = in_idx + i * 4;
    for (int kz = 0; kz < ks[0]; ++kz)
      for (int ky = 0; ky < ks[1]; ++ky)
        for (int kx = 0; kx < ks[2]; ++kx) {
          const int kk[3] = {kz, ky, kx};
          int o[3], ok = 1;
          for (int a = 0; a < 3 && ok; ++a) {
            int t = c[1 + a] + pd[a] - kk[a];
            if (t < 0 || t % st[a] != 0) ok = 0;
            else {
              o[a] = t / st[a];
              if (o[a] >= out_shape[a]) ok = 0;
            }
          }
          if (ok)
            keys[nk++] = (((int64_t)c[0] * out_shape[0] + o[0]) * out_shape[1] + o[1]) * out_shape[2] + o[2];
        }
  }
  qsort(keys, (size_t)nk, sizeof(int64_t), cmp_i64);
  int64_t m = 0;
  for (int64_t j = 0; j < nk; ++j) {
    if (j && keys[j] == keys[j - 1]) continue;
    if (m < cap) {
      int64_t key = keys[j];
      out_idx[m * 4 + 3] = (int32_t)(key % out_shape[2]); key /= out_shape[2];
      out_idx[m * 4 + 2] = (int32_t)(key % out_shape[1]); key /= out_shape[1];
      out_idx[m * 4 + 1] = (int32_t)(key % out_shape[0]); key /= out_shape[0];
      out_idx[m * 4 + 0] = (int32_t)key;
    }
    ++m;
  }
  free(keys);
  return m;
}

static int cmp_key_idx_keyonly(const void* a, const void* b) {
  const key_idx_t *x = (const key_idx_t*)a, *y = (const key_idx_t*)b;
  return x->key < y->key ? -1 : (x->key > y->key);
}

/* nbr[k][o] = row of the active input at o*s - p + k, else -1.  Works for both conv kinds:
 * SubMConv3d passes out_idx == in_idx, stride 1, pad = ksize/2. */
void oracle_spconv_rulebook(const int32_t* in_idx, int64_t m_in, const int32_t* out_idx, int64_t m_out,
                            int batch, const int in_shape[3], const int ks[3], const int st[3],
                            const int pd[3], int32_t* nbr) {
  (void)batch;
  key_idx_t* tab = (key_idx_t*)malloc(sizeof(key_idx_t) * (size_t)(m_in + 1));
  for (int64_t i = 0; i < m_in; ++i) {
    const int32_t* c = in_idx + i * 4;
    tab[i].key = (((int64_t)c[0] * in_shape[0] + c[1]) * in_shape[1] + c[2]) * in_shape[2] + c[3];
    tab[i].idx = i;
  }
  qsort(tab, (size_t)m_in, sizeof(key_idx_t), cmp_key_idx);
  const int kvol = ks[0] * ks[1] * ks[2];
#pragma omp parallel for schedule(static)
  for (int64_t o = 0; o < m_out; ++o) {
    const int32_t* c = out_idx + o * 4;
    for (int kz = 0; kz < ks[0]; ++kz)
      for (int ky = 0; ky < ks[1]; ++ky)
        for (int kx = 0; kx < ks[2]; ++kx) {
          const int k = (kz * ks[1] + ky) * ks[2] + kx;
          const int iz = c[1] * st[0] - pd[0] + kz, iy = c[2] * st[1] - pd[1] + ky,
                    ix = c[3] * st[2] - pd[2] + kx;
          int32_t row = -1;
          if (iz >= 0 && iz < in_shape[0] && iy >= 0 && iy < in_shape[1] && ix >= 0 && ix < in_shape[2]) {
            key_idx_t q;
            q.key = (((int64_t)c[0] * in_shape[0] + iz) * in_shape[1] + iy) * in_shape[2] + ix;
            q.idx = 0;
            key_idx_t* hit = (key_idx_t*)bsearch(&q, tab, (size_t)m_in, sizeof(key_idx_t), cmp_key_idx_keyonly);
            if (hit) row = (int32_t)hit->idx;
          }
          nbr[(int64_t)k * m_out + o] = row;
        }
  }
  (void)kvol;
  free(tab);
}

/* out[o][co] = bias[co] + sum_k sum_ci W[co][k][ci] * in[nbr[k][o]][ci]; double accumulation
 * (the checker is deliberately more precise than fp32 so both sides are judged against it). */
void oracle_spconv_forward(const float* in_feat, int64_t m_in, int cin, const float* weight,
                           const float* bias, int cout, int kvol, const int32_t* nbr, int64_t m_out,
                           float* out_feat) {
  (void)m_in;
#pragma omp parallel for schedule(static)
  for (int64_t o = 0; o < m_out; ++o) {
    for (int co = 0; co < cout; ++co) {
      double acc = bias ? (double)bias[co] : 0.0;
      for (int k = 0; k < kvol; ++k) {
        const int32_t r = nbr[(int64_t)k * m_out + o];
        if (r < 0) continue;
        const float* x = in_feat + (int64_t)r * cin;
        const float* w = weight + ((int64_t)co * kvol + k) * cin;
        for (int ci = 0; ci < cin; ++ci) acc += (double)w[ci] * (double)x[ci];
      }
      out_feat[o * cout + co] = (float)acc;
    }
  }
}

/* grad_in[i][ci] = sum over pairs (k,o) with nbr[k][o]==i of sum_co W[co][k][ci]*grad_out[o][co] */
void oracle_spconv_dgrad(const float* grad_out, int64_t m_out, int cout, const float* weight, int cin,
                         int kvol, const int32_t* nbr, int64_t m_in, float* grad_in) {
  double* acc = (double*)calloc((size_t)(m_in * cin + 1), sizeof(double));
  for (int k = 0; k < kvol; ++k)
    for (int64_t o = 0; o < m_out; ++o) {
      const int32_t r = nbr[(int64_t)k * m_out + o];
      if (r < 0) continue;
      for (int co = 0; co < cout; ++co) {
        const double g = grad_out[o * cout + co];
        const float* w = weight + ((int64_t)co * kvol + k) * cin;
        double* a = acc + (int64_t)r * cin;
        for (int ci = 0; ci < cin; ++ci) a[ci] += g * (double)w[ci];
      }
    }
  for (int64_t e = 0; e < m_in * cin; ++e) grad_in[e] = (float)acc[e];
  free(acc);
}

/* grad_w[co][k][ci] = sum_o grad_out[o][co] * in[nbr[k][o]][ci] */
void oracle_spconv_wgrad(const float* in_feat, int64_t m_in, int cin, const float* grad_out,
                         int64_t m_out, int cout, int kvol, const int32_t* nbr, float* grad_w) {
  (void)m_in;
#pragma omp parallel for schedule(dynamic)
  for (int k = 0; k < kvol; ++k) {
    double* acc = (double*)calloc((size_t)cout * cin, sizeof(double));
    for (int64_t o = 0; o < m_out; ++o) {
      const int32_t r = nbr[(int64_t)k * m_out + o];
      if (r < 0) continue;
      const float* x = in_feat + (int64_t)r * cin;
      for (int co = 0; co < cout; ++co) {
        const double g = grad_out[o * cout + co];
        for (int ci = 0; ci < cin; ++ci) acc[co * cin + ci] += g * (double)x[ci];
      }
    }
    for (int co = 0; co < cout; ++co)
      for (int ci = 0; ci < cin; ++ci) grad_w[((int64_t)co * kvol + k) * cin + ci] = (float)acc[co * cin + ci];
    free(acc);
  }
}

/* SparseConvTensor.dense(): [B, C, D, H, W] (sparse_net.py:304) */
void oracle_sparse_to_dense(const float* feat, const int32_t* idx, int64_t m, int c, int batch,
                            const int shape[3], float* dense) {
  (void)batch;
  const int64_t dhw = (int64_t)shape[0] * shape[1] * shape[2];
  for (int64_t r = 0; r < m; ++r) {
    const int32_t* q = idx + r * 4;
    const int64_t sp = ((int64_t)q[1] * shape[1] + q[2]) * shape[2] + q[3];
    for (int ch = 0; ch < c; ++ch) dense[((int64_t)q[0] * c + ch) * dhw + sp] = feat[r * c + ch];
  }
}

/* ------------------------------------------------------------------------------------------
 * box / multi-scale deformable attention (box_attn_kernel.cuh:34-97,100-184,274-349,352-472;
 * the ms_deform_attn family ms_deform_im2col_cuda.cuh:238-... is the same math)
 * ---------------------------------------------------------------------------------------- */
void oracle_msda_forward(const float* value, const int64_t* shapes, const int64_t* level_start,
                         const float* loc, const float* attn, int b, int s, int h, int d, int l,
                         int lq, int p, float* out) {
  const int64_t total = (int64_t)b * lq * h;
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < total; ++t) {
    const int m = (int)(t % h);
    const int64_t bq = t / h;
    const int bi = (int)(bq / lq);
    const float* lw = loc + t * l * p * 2;
    const float* aw = attn + t * l * p;
    float* o = out + t * d;
    for (int c = 0; c < d; ++c) o[c] = 0.0f;
    for (int li = 0; li < l; ++li) {
      const int H = (int)shapes[li * 2], W = (int)shapes[li * 2 + 1];
      const float* v = value + ((int64_t)bi * s + level_start[li]) * h * d;
      for (int pi = 0; pi < p; ++pi) {
        const float loc_w = lw[(li * p + pi) * 2], loc_h = lw[(li * p + pi) * 2 + 1];
        const float wgt = aw[li * p + pi];
        const float h_im = loc_h * (float)H - 0.5f, w_im = loc_w * (float)W - 0.5f;
        if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue; /* :325-328 */
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - h_low, lwf = w_im - w_low, hh = 1 - lh, hw = 1 - lwf;
        const float w1 = hh * hw, w2 = hh * lwf, w3 = lh * hw, w4 = lh * lwf;
        for (int c = 0; c < d; ++c) {
          float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
          if (h_low >= 0 && w_low >= 0) v1 = v[((int64_t)h_low * W + w_low) * h * d + m * d + c];
          if (h_low >= 0 && w_high <= W - 1) v2 = v[((int64_t)h_low * W + w_high) * h * d + m * d + c];
          if (h_high <= H - 1 && w_low >= 0) v3 = v[((int64_t)h_high * W + w_low) * h * d + m * d + c];
          if (h_high <= H - 1 && w_high <= W - 1) v4 = v[((int64_t)h_high * W + w_high) * h * d + m * d + c];
          o[c] += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * wgt;
        }
      }
    }
  }
}

void oracle_msda_backward(const float* value, const int64_t* shapes, const int64_t* level_start,
                          const float* loc, const float* attn, const float* grad_out, int b, int s,
                          int h, int d, int l, int lq, int p, float* grad_value, float* grad_loc,
                          float* grad_attn) {
  const int64_t total = (int64_t)b * lq * h;
  /* grad_value accumulated in double for an order-independent checker */
  double* gv = (double*)calloc((size_t)b * s * h * d + 1, sizeof(double));
  for (int64_t t = 0; t < total; ++t) {
    const int m = (int)(t % h);
    const int64_t bq = t / h;
    const int bi = (int)(bq / lq);
    const float* lw = loc + t * l * p * 2;
    const float* aw = attn + t * l * p;
    const float* go = grad_out + t * d;
    for (int li = 0; li < l; ++li) {
      const int H = (int)shapes[li * 2], W = (int)shapes[li * 2 + 1];
      const int64_t base = ((int64_t)bi * s + level_start[li]) * h * d;
      const float* v = value + base;
      double* g = gv + base;
      for (int pi = 0; pi < p; ++pi) {
        const float loc_w = lw[(li * p + pi) * 2], loc_h = lw[(li * p + pi) * 2 + 1];
        const float wgt = aw[li * p + pi];
        const float h_im = loc_h * (float)H - 0.5f, w_im = loc_w * (float)W - 0.5f;
        if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - h_low, lwf = w_im - w_low, hh = 1 - lh, hw = 1 - lwf;
        const float w1 = hh * hw, w2 = hh * lwf, w3 = lh * hw, w4 = lh * lwf;
        double ga = 0, gw = 0, gh = 0;
        for (int c = 0; c < d; ++c) {
          const float top = go[c];
          const float tv = top * wgt;
          float v1 = 0, v2 = 0, v3 = 0, v4 = 0, ghw = 0, gww = 0;
          if (h_low >= 0 && w_low >= 0) {
            const int64_t q = ((int64_t)h_low * W + w_low) * h * d + m * d + c;
            v1 = v[q]; ghw -= hw * v1; gww -= hh * v1; g[q] += (double)(w1 * tv);
          }
          if (h_low >= 0 && w_high <= W - 1) {
            const int64_t q = ((int64_t)h_low * W + w_high) * h * d + m * d + c;
            v2 = v[q]; ghw -= lwf * v2; gww += hh * v2; g[q] += (double)(w2 * tv);
          }
          if (h_high <= H - 1 && w_low >= 0) {
            const int64_t q = ((int64_t)h_high * W + w_low) * h * d + m * d + c;
            v3 = v[q]; ghw += hw * v3; gww -= lh * v3; g[q] += (double)(w3 * tv);
          }
          if (h_high <= H - 1 && w_high <= W - 1) {
            const int64_t q = ((int64_t)h_high * W + w_high) * h * d + m * d + c;
            v4 = v[q]; ghw += lwf * v4; gww += lh * v4; g[q] += (double)(w4 * tv);
          }
          ga += (double)(top * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4));
          gw += (double)((float)W * gww * tv);
          gh += (double)((float)H * ghw * tv);
        }
        grad_attn[t * l * p + li * p + pi] = (float)ga;
        grad_loc[(t * l * p + li * p + pi) * 2] = (float)gw;
        grad_loc[(t * l * p + li * p + pi) * 2 + 1] = (float)gh;
      }
    }
  }
  for (int64_t e = 0; e < (int64_t)b * s * h * d; ++e) grad_value[e] = (float)gv[e];
  free(gv);
}

/* ------------------------------------------------------------------------------------------
 * "next" row n1: rotated BEV overlap / IoU and NMS  (efg/operators/src/iou3d_nms/
 * iou3d_nms_kernel.cu:34-239 box_overlap / iou_bev, :270-309 nms_kernel, :312-322 iou_normal,
 * iou3d_nms.cpp:82-128 sequential suppression).  PARITY UNPINNED: the reference's CPU twin
 * iou3d_cpu.cpp includes <cuda.h> / <cuda_runtime_api.h>, which this image lacks, so it cannot be
 * built here and the reference has no tests; restated from the device functions and cross-checked
 * against an independent polygon-clipping IoU in tests/test_oracle_iou3d.py.
 * ---------------------------------------------------------------------------------------- */
typedef struct { float x, y; } pt_t;
static const float kIouEps = 1e-8f;

static float cross3(pt_t p1, pt_t p2, pt_t p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
static float cross2(pt_t a, pt_t b) { return a.x * b.y - a.y * b.x; }

static int rect_cross(pt_t p1, pt_t p2, pt_t q1, pt_t q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

static int in_box2d(const float* box, pt_t p) { /* :54-64, MARGIN 1e-2 */
  const float MARGIN = 1e-2f;
  const float c = cosf(-box[6]), s = sinf(-box[6]);
  const float rx = (p.x - box[0]) * c + (p.y - box[1]) * (-s);
  const float ry = (p.x - box[0]) * s + (p.y - box[1]) * c;
  return fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN;
}

static int seg_intersection(pt_t p1, pt_t p0, pt_t q1, pt_t q0, pt_t* ans) { /* :66-97 */
  if (!rect_cross(p0, p1, q0, q1)) return 0;
  const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  const float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > kIouEps) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

static void box_corners(const float* b, pt_t c[5]) { /* :127-158 */
  const float hx = b[3] / 2, hy = b[4] / 2, cs = cosf(b[6]), sn = sinf(b[6]);
  const float xs[4] = {b[0] - hx, b[0] + hx, b[0] + hx, b[0] - hx};
  const float ys[4] = {b[1] - hy, b[1] - hy, b[1] + hy, b[1] + hy};
  for (int k = 0; k < 4; ++k) {
    c[k].x = (xs[k] - b[0]) * cs + (ys[k] - b[1]) * (-sn) + b[0];
    c[k].y = (xs[k] - b[0]) * sn + (ys[k] - b[1]) * cs + b[1];
  }
  c[4] = c[0];
}

float oracle_box_overlap(const float* box_a, const float* box_b) { /* :111-239 */
  pt_t ca[5], cb[5], pts[16], center = {0.f, 0.f};
  box_corners(box_a, ca);
  box_corners(box_b, cb);
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (seg_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], &pts[cnt])) {
        center.x += pts[cnt].x; center.y += pts[cnt].y; ++cnt;
      }
  for (int k = 0; k < 4; ++k) {
    if (in_box2d(box_a, cb[k])) { center.x += cb[k].x; center.y += cb[k].y; pts[cnt++] = cb[k]; }
    if (in_box2d(box_b, ca[k])) { center.x += ca[k].x; center.y += ca[k].y; pts[cnt++] = ca[k]; }
  }
  if (cnt == 0) return 0.0f; /* reference: 0/0 centre, empty area loop -> 0 */
  center.x /= cnt; center.y /= cnt;
  for (int j = 0; j < cnt - 1; ++j) /* bubble sort by angle, descending (:207-215) */
    for (int i = 0; i < cnt - j - 1; ++i)
      if (atan2f(pts[i].y - center.y, pts[i].x - center.x) > atan2f(pts[i + 1].y - center.y, pts[i + 1].x - center.x)) {
        pt_t t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
      }
  float area = 0;
  for (int k = 0; k < cnt - 1; ++k) {
    pt_t a = {pts[k].x - pts[0].x, pts[k].y - pts[0].y}, b = {pts[k + 1].x - pts[0].x, pts[k + 1].y - pts[0].y};
    area += cross2(a, b);
  }
  return fabsf(area) / 2.0f;
}

float oracle_iou_bev(const float* a, const float* b) { /* :241-248 */
  const float sa = a[3] * a[4], sb = b[3] * b[4], so = oracle_box_overlap(a, b);
  return so / fmaxf(sa + sb - so, kIouEps);
}

static float iou_normal(const float* a, const float* b) { /* :312-322 */
  const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  const float inter = fmaxf(right - left, 0.f) * fmaxf(bottom - top, 0.f);
  return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, kIouEps);
}

/* mode 0: overlap area, 1: IoU.  out [na, nb] */
void oracle_boxes_bev(const float* a, int na, const float* b, int nb, int mode, float* out) {
#pragma omp parallel for schedule(dynamic, 16)
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j)
      out[(int64_t)i * nb + j] = mode ? oracle_iou_bev(a + i * 7, b + j * 7) : oracle_box_overlap(a + i * 7, b + j * 7);
}

/* boxes already sorted by score (descending); greedy suppression exactly as nms_kernel + the host
 * loop of iou3d_nms.cpp:104-117: box j > i is removed when IoU(i, j) > thresh and i is kept.
 * rotated = 1: iou_bev, 0: iou_normal.  Returns the number kept; keep[] holds their indices. */
int oracle_nms(const float* boxes, int n, float thresh, int rotated, int64_t* keep) {
  unsigned char* removed = (unsigned char*)calloc((size_t)n + 1, 1);
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    if (removed[i]) continue;
    keep[kept++] = i;
    for (int j = i + 1; j < n; ++j)
      if (!removed[j]) {
        const float v = rotated ? oracle_iou_bev(boxes + i * 7, boxes + j * 7) : iou_normal(boxes + i * 7, boxes + j * 7);
        if (v > thresh) removed[j] = 1;
      }
  }
  free(removed);
  return kept;
}

/* ------------------------------------------------------------------------------------------
 * "next" row (GPU matcher): linear sum assignment.  The reference calls
 * scipy.optimize.linear_sum_assignment ($CQ/modules/matcher.py:1,89) -- a third-party dependency
 * (scipy, not pinned by the reference; 1.15.3 in this image) whose implementation is the
 * rectangular shortest-augmenting-path algorithm of D. F. Crouse, "On implementing 2D rectangular
 * assignment algorithms", IEEE TAES 52(4), 2016 (scipy/optimize/rectangular_lsap).  Restated here
 * including its tie-breaking (candidate columns are scanned in the order of the `remaining` list,
 * which starts REVERSED and is compacted by moving the last element into the freed slot).
 * PINNED against scipy itself in tests/test_oracle_lsap.py (random, tie-heavy integer and constant
 * matrices, both orientations).
 *   cost: row-major [nq][g_stride] floats, first ng columns valid.  The assignment runs with the
 *   smaller side as rows (scipy transposes when nq > ng) in fp64, same operation order.
 *   query_of_gt[g] = matched query, or -1 (only when ng > nq).
 * ---------------------------------------------------------------------------------------- */
int oracle_lsap(const float* cost, int nq, int g_stride, int ng, int64_t* query_of_gt) {
  for (int g = 0; g < ng; ++g) query_of_gt[g] = -1;
  if (nq == 0 || ng == 0) return 0;
  const int transposed = nq > ng; /* rows = GT, cols = queries */
  const int nr = transposed ? ng : nq, nc = transposed ? nq : ng;
#define COST(i, j) ((double)(transposed ? cost[(size_t)(j) * g_stride + (i)] : cost[(size_t)(i) * g_stride + (j)]))
  double* u = (double*)calloc(nr, sizeof(double));
  double* v = (double*)calloc(nc, sizeof(double));
  double* spc = (double*)malloc(sizeof(double) * nc);
  int* path = (int*)malloc(sizeof(int) * nc);
  int* col4row = (int*)malloc(sizeof(int) * nr);
  int* row4col = (int*)malloc(sizeof(int) * nc);
  int* remaining = (int*)malloc(sizeof(int) * nc);
  unsigned char* SR = (unsigned char*)malloc(nr);
  unsigned char* SC = (unsigned char*)malloc(nc);
  for (int i = 0; i < nr; ++i) col4row[i] = -1;
  for (int j = 0; j < nc; ++j) row4col[j] = -1;
  int ok = 1;
  for (int cur = 0; cur < nr && ok; ++cur) {
    double min_val = 0;
    int i = cur, num_remaining = nc, sink = -1;
    for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
    memset(SR, 0, nr);
    memset(SC, 0, nc);
    for (int j = 0; j < nc; ++j) spc[j] = INFINITY;
    while (sink == -1) {
      int index = -1;
      double lowest = INFINITY;
      SR[i] = 1;
      for (int it = 0; it < num_remaining; ++it) {
        const int j = remaining[it];
        const double r = min_val + COST(i, j) - u[i] - v[j];
        if (r < spc[j]) {
          path[j] = i;
          spc[j] = r;
        }
        if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
          lowest = spc[j];
          index = it;
        }
      }
      min_val = lowest;
      if (min_val == INFINITY) { ok = 0; break; } /* infeasible */
      const int j = remaining[index];
      if (row4col[j] == -1) sink = j; else i = row4col[j];
      SC[j] = 1;
      remaining[index] = remaining[--num_remaining];
    }
    if (!ok) break;
    u[cur] += min_val;
    for (int r = 0; r < nr; ++r)
      if (SR[r] && r != cur) u[r] += min_val - spc[col4row[r]];
    for (int j = 0; j < nc; ++j)
      if (SC[j]) v[j] -= min_val - spc[j];
    int j = sink;
    for (;;) {
      const int r = path[j];
      row4col[j] = r;
      const int t = col4row[r];
      col4row[r] = j;
      j = t;
      if (r == cur) break;
    }
  }
#undef COST
  if (ok) {
    if (transposed) for (int g = 0; g < nr; ++g) query_of_gt[g] = col4row[g];
    else for (int q = 0; q < nr; ++q) query_of_gt[col4row[q]] = q;
  }
  free(u); free(v); free(spc); free(path); free(col4row); free(row4col); free(remaining); free(SR); free(SC);
  return ok ? 0 : -1;
}
