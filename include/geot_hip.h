/*
 * geot_hip.h -- C ABI of libgeot_hip.so, the MI355X (gfx950) implementation of
 * GeoT's point-cloud sampling / grouping / interpolation hot path.
 *
 * Every entry point is a plain launcher: device pointers, sizes and a HIP
 * stream handle in, hipError_t (as int, 0 == hipSuccess) out.  No torch types,
 * no allocation, no host synchronisation, never exit(): safe to call from any
 * thread and to capture into a hipGraph.  `stream` is a hipStream_t passed as
 * void* (NULL = the legacy default stream).
 *
 * Each declaration cites the reference launcher / binding it replaces
 * (paths relative to the GeoT checkout).  Layouts: fp32 and int32 only, all
 * tensors contiguous, exactly as the reference's extensions require
 * (SURVEY.md section 8b).  Accumulating outputs (`*_grad`, FPS `temp`) must
 * arrive pre-filled (0 / 1e10) as in the reference; all other outputs are
 * written in full, so they may arrive uninitialised.
 *
 * Distances are un-contracted IEEE fp32, ((dx*dx)+(dy*dy))+(dz*dz), so integer
 * results are bit-identical to the CPU oracle (oracle/geot_oracle.c).
 */
#ifndef GEOT_HIP_H
#define GEOT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define GEOT_ABI_VERSION 7
#define GEOT_NTM_MAX_C 32   /* largest class count of the geot_ntm_* entry points */

/* ABI version / diagnostics. */
int geot_abi_version(void);
/* Which squared-distance arithmetic this build of the library uses (all index-producing ops):
 * 0 = un-contracted IEEE fp32 ((dx*dx)+(dy*dy))+(dz*dz) (default, libgeot_hip.so);
 * 1 = fma(dz,dz,fma(dy,dy,dx*dx)) (libgeot_hip_fma.so); 2 = fma(dz,dz,fma(dx,dx,dy*dy)) (libgeot_hip_fma_xy.so):
 * the two contractions nvcc -fmad=true may have made of the reference's source. */
int geot_distance_mode(void);
const char *geot_error_string(int hip_error);

/* ---- furthest point sampling -------------------------------------------
 * Dense batch.  Replaces
 *   pointnet2/_ext_src/src/sampling_gpu.cu:178-232 furthest_point_sampling_kernel_wrapper
 *     (bound as furthest_point_sampling, sampling.cpp:67-88): block_cap=512, skip_origin=1
 *   openpoints/cpp/pointnet2_batch/src/sampling_gpu.cu:218-260 (bound as
 *     furthest_point_sampling_wrapper, sampling.cpp:39-48):          block_cap=1024, skip_origin=0
 * xyz (b,n,3); temp (b,n) in/out, pre-filled 1e10; idxs (b,m) out.
 * block_cap selects the reference thread-block size whose reduction order
 * defines the tie rule; it is not our launch geometry. */
int geot_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp, int *idxs,
                                 int block_cap, int skip_origin, void *stream);

/* Offset-batched, optionally weighted.  Replaces
 *   pointops/src/sampling/sampling_cuda_kernel.cu:131-171 furthestsampling_cuda_launcher
 *   pointops/src/sampling/sampling_cuda_kernel.cu:309-349 furthestsampling_weights_cuda_launcher
 *   (declared extern "C" in pointops/src/sampling/sampling_cuda_kernel.h:9-29).
 * xyz (n_total,3); offset/new_offset (b) int32 inclusive prefix sums (device);
 * weights (n_total) or NULL; tmp (n_total) in/out, pre-filled 1e10;
 * idx (new_offset[b-1]) out, global indices.  n_max = largest segment. */
int geot_furthestsampling_offset(int b, int n_max, const float *xyz, const int *offset,
                                 const int *new_offset, const float *weights, float *tmp,
                                 int *idx, void *stream);

/* ---- gather ---------------------------------------------------------------
 * pointnet2/_ext_src/src/sampling_gpu.cu:25-33, 52-60
 * openpoints/cpp/pointnet2_batch/src/sampling_gpu.cu (gather_points_kernel_launcher_fast, _grad_) */
int geot_gather_points(int b, int c, int n, int m, const float *points, const int *idx, float *out,
                       void *stream);
int geot_gather_points_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                            float *grad_points, void *stream);
/* The same gradient without float atomics (one writer per element, bit-reproducible): see geot_group_points_grad_ws.
 * workspace: geot_scatter_grad_ws_floats(b, c, n, m, 1, 0) floats of scratch. */
int geot_gather_points_grad_ws(int b, int c, int n, int m, const float *grad_out, const int *idx,
                               float *grad_points, float *workspace, void *stream);

/* ---- ball query -------------------------------------------------------------
 * pointnet2/_ext_src/src/ball_query_gpu.cu:49-57 query_ball_point_kernel_wrapper
 * openpoints/cpp/pointnet2_batch/src/ball_query_gpu.cu (ball_query_kernel_launcher_fast)
 * new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample); written in full. */
int geot_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                    const float *xyz, int *idx, void *stream);
/* openpoints/cpp/pointops/src/ballquery/ballquery_cuda_kernel.cu:80-88 ballquery_launcher.
 * b = number of batch segments in offset/new_offset (the reference scans for it). */
int geot_ballquery_offset(int b, int m, float radius, int nsample, const float *xyz,
                          const float *new_xyz, const int *offset, const int *new_offset, int *idx,
                          void *stream);

/* ---- group (channels-first) -------------------------------------------------
 * pointnet2/_ext_src/src/group_points_gpu.cu:33-42, 69-78
 * openpoints/cpp/pointnet2_batch/src/group_points_gpu.cu (…_launcher_fast) */
int geot_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                      const int *idx, float *out, void *stream);
int geot_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                           const int *idx, float *grad_points, void *stream);

/* Same result as geot_group_points_grad / geot_three_interpolate_grad (up to fp32 summation order) WITHOUT float
 * atomics (geot_amd/csrc/tile_scatter.hip): the pairs of every tile of sources are sorted by target once per call, a
 * workgroup keeps the sums of (batch, 4 channels) x all targets in LDS and streams grad_out through it -- grad_out is
 * read once, every output element has one writer and one summation order: bit-reproducible, 3-4x faster.
 * workspace: geot_scatter_grad_ws_floats(b, c, targets, sources, slots, weighted) floats; contents irrelevant unless
 * geot_grad_ws_needs_zero says 1 (shapes the sorted form does not take -- more than 32768 targets per cloud -- fall
 * back to a channels-last atomic accumulation in the workspace, which must then arrive zero-filled). */
long long geot_scatter_grad_ws_floats(int b, int c, int m_targets, long long n_sources, int slots_per_source,
                                      int weighted);
int geot_group_points_grad_ws(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                              const int *idx, float *grad_points, float *workspace, void *stream);
/* 0 when the *_grad_ws entry points only use their workspace as scratch for these sizes (gradient as a
 * gather over a reverse index), 1 when they accumulate in it and it must arrive zero-filled. */
int geot_grad_ws_needs_zero(int b, int c, int m_targets, long long n_sources, int slots_per_source);
int geot_three_interpolate_grad_ws(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                   const float *weight, float *grad_points, float *workspace,
                                   void *stream);
/* As geot_three_interpolate_grad_ws, for a grad_points (and a workspace) that arrive UNINITIALISED: every element
 * of grad_points is written.  Where one thread owns an output element (the reverse-index gather) the result is
 * stored instead of added: no zero-fill pass and no read of the old value, 12 % of the op at (8, 1536, 24000 ->
 * 8192).  Replaces the torch.zeros + accumulate of pointnet2_utils.py:147-165's backward. */
int geot_three_interpolate_grad_out(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                    const float *weight, float *grad_points, float *workspace, void *stream);
/* PointnetFPModule front end (pointnet2/pointnet2_modules.py:619-626; openpoints' three_interpolation is the same
 * chain) without its temporaries.  geot_fp_weights: the inverse-distance weights from three_nn's SQUARED
 * distances, weight = r / ((r0 + r1) + r2), r = 1 / (sqrt(d2) + 1e-8), in one launch.  _into / _grad_from: the
 * interpolation writes (reads its gradient from) the first c channels of the wider (B, c + c_skip, n) tensor
 * that `torch.cat([interpolated, unknow_feats], dim=1)` would build; *_bstride = floats between batches there
 * (>= c * n).  Workspace of _grad_from as for _grad_ws. */
int geot_fp_weights(int b, int n, const float *dist2, float *weight, void *stream);
int geot_three_interpolate_into(int b, int c, int m, int n, const float *points, const int *idx,
                                const float *weight, float *out, long long out_bstride, void *stream);
int geot_three_interpolate_grad_from(int b, int c, int n, int m, const float *grad_out, long long grad_bstride,
                                     const int *idx, const float *weight, float *grad_points, float *workspace,
                                     void *stream);

/* Grid-accelerated variants of geot_knn_sorted / geot_three_nn (geot_amd/csrc/knn_grid.hip): identical
 * outputs, bit for bit, but only the cells around each query are visited (exact: the search widens until
 * the k-th distance is strictly inside the visited block).  workspace = geot_knn_grid_ws_bytes(b, nr)
 * bytes of 16-byte-aligned scratch (contents irrelevant on entry).  Falls back to the brute-force kernel
 * when workspace is NULL / too small or geot_knn_grid_eligible(b, nq, nr, k) is 0 (small problems, k > 64,
 * or GEOT_NN_IMPL=basic|wave in the environment; GEOT_NN_IMPL=grid forces the grid where it is valid). */
long long geot_knn_grid_ws_bytes(int b, int nr);
/* geot_ball_query through the same grid (cells of edge >= 1.0001 radius; identical output). */
int geot_ball_grid_eligible(int b, int n, int m, float radius, int nsample);
int geot_ball_query_ws(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                       int *idx, void *workspace, long long ws_bytes, void *stream);
int geot_knn_grid_eligible(int b, int nq, int nr, int k);
int geot_knn_sorted_ws(int b, int nq, int nr, int k, const float *query, const float *ref, int *idx,
                       float *dist2, void *workspace, long long ws_bytes, void *stream);
int geot_three_nn_ws(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                     void *workspace, long long ws_bytes, void *stream);

/* BatchNorm (+ ReLU) on channels-first (b, c, l) tensors as streaming passes -- the conv1x1 -> BatchNorm -> ReLU
 * stages of SharedMLP (pointnet2/pytorch_utils.py:8-117) and of the mini-PointNet Encoder (transformer.py:106-136) in
 * training mode.  The O(c) arithmetic between the passes (mean / variance / running statistics / SyncBatchNorm's
 * all-reduce) is the caller's (geot_amd/fused_norm.py).
 *   geot_bn_slices      S = number of slices a row is cut into for the partial sums (<= 32)
 *   geot_bn_stats       partial (b,c,S,2) = per-slice (sum x, sum x^2)
 *   geot_bn_apply       out = x * scale[c] + shift[c], clamped at 0 when relu
 *   geot_bn_bwd_reduce  partial (b,c,S,2) = (sum g, sum g xhat), g = dz masked by [x*scale+shift > 0] when relu,
 *                       xhat = (x - mean[c]) * rstd[c]
 *   geot_bn_bwd_apply   dx = k0[c] * (g - c1[c] - xhat * c2[c])
 * and the PointnetFPModule front end (pointnet2_modules.py:619-640) with the first 1x1 convolution moved in front of
 * the interpolation:  y (b,c,n) = sum_t weight[.,t] A[:, idx[.,t]] + Wb (c,cs) skip (b,cs,n),  A (b,c,m) = W_a
 * known_feats from the caller's GEMM, cs <= 8; partial (b,c,S',2) = per-slice (sum y, sum y^2) for the BatchNorm
 * behind it (S' = geot_fp_front_slices).  Rows of A must fit the LDS (m <= 36864). */
/* The O(c) arithmetic between the passes as launches of its own (13 and 7 torch launches per layer otherwise:
 * 2.5 % of the configs[2] step), split where SyncBatchNorm all-reduces:
 *   geot_bn_sums      sums (c,2) fp64 = sum over b and S of partial (b,c,S,2), fixed order
 *   geot_bn_finalize  n = count_dev ? *count_dev (device double: the all-reduced element count) : count;
 *                     mean = sums0/n, var = max(sums1/n - mean^2, 0) in fp64; mean, rstd = 1/sqrt(var+eps) as fp32;
 *                     scale = gamma*rstd, shift = beta - mean*scale (gamma / beta NULL: 1 / 0); running_mean / _var
 *                     (NULL: not tracked) <- (1-eaf)*old + eaf*(mean | var*n/max(n-1,1))  (torch batch_norm semantics);
 *                     pre_bias (NULL: none): the statistics are those of y while the layer normalises y + pre_bias[c]
 *                     (a convolution bias in front of a batch-statistics BatchNorm cancels in the output: the caller
 *                     skips the add, only the running mean sees it)
 *   geot_bn_bwd_coef  g_beta, g_gamma = fp32 of local_sums (this rank's sum g, sum g xhat); c1, c2 = sums/n (0 if n == 0) */
int geot_bn_sums(int b, int c, int s, const float *partial, double *sums, void *stream);
/* geot_bn_stats and geot_fp_front write statistics records (b,c,S,4) = (s1, s2, pivot, count): sums of (x - pivot) and
 * (x - pivot)^2 around the slice's first element, so that |mean| >> std costs no digits; geot_bn_sums_shifted turns every
 * record back into (sum x, sum x^2) in fp64 and adds them: sums (c,2).  (geot_bn_sums: the plain (b,c,S,2) partials of
 * the backward reduce.) */
int geot_bn_sums_shifted(int b, int c, int s, const float *partial, double *sums, void *stream);
int geot_bn_finalize(int c, const double *sums, double count, const double *count_dev, double eps, double eaf,
                     const float *gamma, const float *beta, const float *pre_bias, float *running_mean,
                     float *running_var, float *mean, float *rstd, float *scale, float *shift, void *stream);
int geot_bn_bwd_coef(int c, const double *local_sums, const double *sums, double count, const double *count_dev,
                     float *g_gamma, float *g_beta, float *c1, float *c2, void *stream);
int geot_bn_slices(int b, int c, int l);
int geot_bn_stats(int b, int c, int l, const float *x, float *partial, void *stream);
int geot_bn_apply(int b, int c, int l, int relu, const float *x, const float *scale, const float *shift, float *out,
                  void *stream);
int geot_bn_bwd_reduce(int b, int c, int l, int relu, const float *x, const float *dz, const float *scale,
                       const float *shift, const float *mean, const float *rstd, float *partial, void *stream);
int geot_bn_bwd_apply(int b, int c, int l, int relu, const float *x, const float *dz, const float *scale,
                      const float *shift, const float *mean, const float *rstd, const float *k0, const float *c1,
                      const float *c2, float *dx, void *stream);
/* Residual add + LayerNorm of the transformer blocks (transformer.py:41-104: x + drop_path(branch), then norm; the
 * position embedding added between blocks, :395-400) as one pass each way over row-major (rows, c) tensors,
 * c in {128, 256, 384, 512, 768, 1024} (geot_res_ln_supported):
 *   t = x + s[row / rows_per_sample] * y + extra   (y, s, extra may be NULL; s needs y),  z = LayerNorm(t; gamma, beta, eps)
 *   geot_res_ln       writes t (t_out NULL: not wanted), z, and the per-row mean / rstd
 *   geot_res_ln_grad  g = gt + LayerNorm'(gz) (gt / gz NULL: zero) = the gradient of x and of extra; gy_out (NULL: not
 *                     wanted) = s * g = the gradient of y; d gamma, d beta;  workspace: geot_res_ln_ws_floats floats. */
int geot_res_ln_supported(int c);
long long geot_res_ln_ws_floats(int rows, int c);
int geot_res_ln(int rows, int c, int rows_per_sample, float eps, const float *x, const float *y, const float *s,
                const float *extra, const float *gamma, const float *beta, float *t_out, float *z_out, float *mean,
                float *rstd, void *stream);
int geot_res_ln_grad(int rows, int c, int rows_per_sample, const float *gz, const float *gt, const float *t,
                     const float *mean, const float *rstd, const float *gamma, const float *s, float *g_out, float *gy_out,
                     float *dgamma, float *dbeta, float *workspace, void *stream);
/* Attention head split (transformer.py:70-72): the (b, n, 3, h, d) qkv projection -> out (3, b*h, n, d) = (q * scale, k,
 * v) contiguous for the batched GEMMs, d a multiple of 4; _grad: three (b*h, n, d) gradients (NULL: zero) back into
 * the (b, n, 3, h, d) gradient with d q scaled -- one launch each instead of a stack, a permuted copy and an
 * element-wise pass over the (b, h, n, n) score gradient. */
int geot_qkv_split(int b, int n, int h, int d, float scale, const float *qkv, float *out, void *stream);
int geot_qkv_split_grad(int b, int n, int h, int d, float scale, const float *gq, const float *gk, const float *gv,
                        float *grad_qkv, void *stream);
/* Gradient of a soft-max over the last dimension n in {64, 128, 256, 512, 1024} of row-major (rows, n): grad_in = y * (grad -
 * sum_j grad_j y_j) in one pass (the attention scores of transformer.py:75-77; torch runs grad * y as a pass of its own). */
int geot_softmax_grad(long long rows, int n, const float *grad, const float *y, float *grad_in, void *stream);
/* Poly-1 focal loss (openpoints/loss/build.py:183-258 Poly1FocalLoss; :799-892 Poly1FocalLoss_U_corr when `keep` is
 * given) on logits (b, c, n) with int64 class labels (b, n) -- no one-hot tensors, two launches forward, one backward:
 *   l = at * BCEwithlogits(x, y) * (1 - pt)^gamma + epsilon * (1 - pt)^(gamma + 1),  y = [label == c], pt = y p + (1-y)(1-p),
 *   at = alpha y + (1 - alpha)(1 - y) (alpha < 0: 1);  out2[0] = sum l / (b c n), or with keep (b, n) bytes:
 *   sum l keep / (c sum keep + 0.001);  out2[1] = 1 / that denominator (for _grad).  workspace: _ws_doubles doubles.
 *   _grad: grad_logits = upstream[0] * out2[1] * keep * dl/dx  (upstream: device scalar). */
long long geot_poly1_focal_ws_doubles(int b, int c, int n);
int geot_poly1_focal(int b, int c, int n, float alpha, float gamma, float epsilon, const float *logits,
                     const long long *labels, const unsigned char *keep, double *workspace, float *out2, void *stream);
int geot_poly1_focal_grad(int b, int c, int n, float alpha, float gamma, float epsilon, const float *logits,
                          const long long *labels, const unsigned char *keep, const float *out2, const float *upstream,
                          float *grad_logits, void *stream);
/* max over the n innermost elements of every row, x (rows, n) contiguous and 16-byte aligned, n a multiple of 4, <= 256:
 * out (rows), arg (rows) uint8 = the first maximum's slot (torch.max semantics); _grad writes dx (rows, n) in full.
 * (Encoder's max over a group's points, transformer.py:127-134; max over nsample of the SA modules.) */
int geot_segment_max(long long rows, int n, const float *x, float *out, unsigned char *arg, void *stream);
/* BatchNorm -> ReLU -> max over the n innermost elements of y (b, c, groups, n) without building the normalised tensor
 * (the last SharedMLP stage + max_pool2d of a SetAbstraction module in training mode, pointnet2_modules.py:57-66): the
 * affine map is monotone, so out = relu(scale_c sel + shift_c) with sel = the row's maximum (scale_c >= 0) or minimum;
 * sel and arg (slot of the first extremum) are kept for _grad:  dx_j = k0_c ([j == arg] gm - c1_c - xhat_j c2_c),
 * gm (b, c, groups) = the pooled output's gradient where the pre-activation was positive, c1 / c2 = the means of the
 * (sparse) gradient and of gradient * xhat (caller: geot_amd/fused_norm.py bn_relu_max).  n as geot_segment_max. */
int geot_bn_pool(int b, int c, int groups, int n, int relu, const float *y, const float *scale, const float *shift, float *out,
                 float *sel, unsigned char *arg, void *stream);
int geot_bn_pool_grad(int b, int c, int groups, int n, const float *y, const float *gm, const unsigned char *arg,
                      const float *mean, const float *rstd, const float *k0, const float *c1, const float *c2, float *dx,
                      void *stream);
/* out (rows) = the sum of every row, same shape rules: the gradient of a per-group term broadcast over the group's
 * points (Encoder, transformer.py:131-132: feature_global expanded over n) at streaming speed, fixed summation order. */
int geot_segment_sum(long long rows, int n, const float *x, float *out, void *stream);
/* out[r] (fp64) = sum of the n floats of row r, fixed order: the long few-row sums (1x1-conv bias gradients over (B, C, N),
 * per-channel input sums) without the semaphore memset torch's reduction issues (a memset node under capture). */
int geot_rowsum_f64(long long rows, int n, const float *x, double *out, void *stream);
/* partial (rows, S, j) = per-slice sums of a[row][.] * b[jj][.] for a (rows, l), b (j, l), j <= 8, S =
 * geot_rowdot_small_slices(rows, l): the weight gradient of a 1x1 convolution with a handful of input channels
 * (Encoder first_conv, transformer.py:110: Conv1d(3, 128) over all points of all groups), one streaming pass. */
/* out (cols) = column sums of the row-major x (rows, cols): the bias gradient of a Linear layer (fixed summation order;
 * workspace: geot_colsum_ws_floats floats). */
long long geot_colsum_ws_floats(int rows, int cols);
int geot_colsum(int rows, int cols, const float *x, float *out, float *workspace, void *stream);
int geot_rowdot_small_slices(int rows, int l);
int geot_rowdot_small(int rows, int l, int j, const float *a, const float *b, float *partial, void *stream);
int geot_segment_max_grad(long long rows, int n, const float *dy, const unsigned char *arg, float *dx, void *stream);
int geot_fp_front_slices(int b, int c, int m, int n);
int geot_fp_front(int b, int c, int m, int n, int cs, const float *A, const int *idx, const float *weight,
                  const float *skip, const float *Wb, float *y, float *partial, void *stream);

/* ---- the same FP front end and its BatchNorm on POINT-MAJOR activations (B, N, C), csrc/channels_last.hip ----------
 * Behaviour replaced: three_interpolate + concat + the first Conv2d/BatchNorm2d/ReLU of PointnetFPModule.mlp
 * (pointnet2/pointnet2_modules.py:619-640, pointnet2/_ext_src/src/interpolate_gpu.cu:88-146) and its gradient.
 * A point's C channels are one contiguous row; the 1x1 convolutions on either side are GEMMs and take the layout as a
 * transpose flag, so nothing is transposed in memory.  Needs C % 4 == 0 and C <= 4096 (geot_cl_tiles() >= 0).
 *   geot_cl_tiles(batches, rows_per_batch, c)  rows T of the (T, 2, c) partial-sum buffer of a bn_*_cl reduction over
 *                      batches x rows_per_batch rows;  geot_fp_front_cl_tiles(b, c, n, cs): the same for fp_front_cl
 *   geot_fp_front_cl   y_cl (b,n,c) = sum_t weight[b,e,t] a_cl[b, idx[b,e,t], :] + wb (c,cs) skip (b,cs,n)[:, e];
 *                      partial (T,2,c) = per-tile (sum y, sum y^2); order (b,n) or NULL = the sequence in which a
 *                      workgroup takes its points (values do not depend on it; the partial sums' rounding does)
 *   geot_bn_stats_cl / _apply_cl / _bwd_reduce_cl / _bwd_apply_cl   as geot_bn_* above on (rows, c) row-major
 *   geot_bn_sums_cl    sums (c,2) fp64 = sum over the T tiles of partial (T,2,c), fixed order
 *   geot_rix_build     reverse index of idx (b,L,nt) with values in [0,m): per target the (source, weight) pairs in
 *                      ascending pair order, into ws (geot_rix_ws_ints ints); weight NULL = unit weights; order (b,m)
 *                      or NULL = the sequence in which the gather takes the targets (a permutation per cloud)
 *   geot_gather_rows_csr_cl   out_cl (b,m,c) = sum over the pairs of target j of weight * g_cl[b, source, :]
 *                      (the gradient of a point-major gather; one writer per row, fixed order: reproducible); `order`
 *                      must be the one the index was built with */
int geot_cl_tiles(int batches, long long rows_per_batch, int c);
int geot_fp_front_cl_tiles(int b, int c, int n, int cs);
int geot_fp_front_cl(int b, int c, int m, int n, int cs, const float *a_cl, const int *idx, const float *weight,
                     const float *skip, const float *wb, const int *order, float *y_cl, float *partial, void *stream);
int geot_bn_stats_cl(long long rows, int c, const float *x, float *partial, void *stream);
int geot_bn_apply_cl(long long rows, int c, int relu, const float *x, const float *scale, const float *shift, float *out,
                     void *stream);
int geot_bn_bwd_reduce_cl(long long rows, int c, int relu, const float *x, const float *dz, const float *scale,
                          const float *shift, const float *mean, const float *rstd, float *partial, void *stream);
int geot_bn_bwd_apply_cl(long long rows, int c, int relu, const float *x, const float *dz, const float *scale,
                         const float *shift, const float *mean, const float *rstd, const float *k0, const float *c1,
                         const float *c2, float *dx, void *stream);
int geot_bn_sums_cl(int tiles, int c, const float *partial, double *sums, void *stream);
/* statistics records: fp_front_cl and bn_stats_cl accumulate SHIFTED sums (around each tile's first row) and write
 * (tiles, 3, c) = (s1, s2, pivot) followed by `tiles` row counts = geot_cl_stat_floats(tiles, c) floats;
 * geot_bn_sums_shifted_cl rebuilds (sum x, sum x^2) per tile in fp64 and adds them up: sums (c,2) */
long long geot_cl_stat_floats(int tiles, int c);
int geot_bn_sums_shifted_cl(int tiles, int c, const float *partial, double *sums, void *stream);
long long geot_rix_ws_ints(int b, long long L, int m, int nt);
int geot_rix_build(int b, int L, int m, int nt, const int *idx, const float *weight, const int *order, int *ws,
                   long long ws_ints, void *stream);
int geot_gather_rows_csr_cl(int b, int c, int L, int m, int nt, const float *g_cl, const int *ws, const int *order,
                            float *out_cl, void *stream);
/* The backward of [fp_front_cl -> BatchNorm (+ ReLU)] in two passes instead of four, the gradient gy of the BatchNorm's
 * input never written:  geot_bn_bwd_reduce_skip_cl = geot_bn_bwd_reduce_cl + the sums sum g skip_k, sum xhat skip_k that
 * give grad_wb = sum_e gy_e skip_e once the means are known (partial (T, 2 + 2 cs, c), T = geot_cl_tiles(1, b n, c);
 * geot_bn_sums_k_cl adds the T tiles up: sums (c, K) fp64);  geot_gather_rows_csr_bn_cl = geot_gather_rows_csr_cl of
 * gy = scale (g - c1 - xhat c2), formed on the fly from the rows of y_cl and dz_cl. */
int geot_bn_sums_k_cl(int tiles, int c, int k, const float *partial, double *sums, void *stream);
int geot_bn_bwd_reduce_skip_cl(int b, int n, int c, int cs, int relu, const float *x, const float *dz, const float *scale,
                               const float *shift, const float *mean, const float *rstd, const float *skip, float *partial,
                               void *stream);
/* grad_wb (c, cs) = scale_c (S1 - c1_c S2 - c2_c S3) from sums_k (c, 2 + 2 cs) of the pass above, c1 / c2 of
 * geot_bn_bwd_coef and s2 (cs) fp64 = the sum of every skip channel over all points */
int geot_fp_skip_wgrad_cl(int c, int cs, const double *sums_k, const float *scale, const float *c1, const float *c2,
                          const double *s2, float *gwb, void *stream);
int geot_gather_rows_csr_bn_cl(int b, int c, int L, int m, int nt, int relu, const float *y_cl, const float *dz_cl,
                               const float *scale, const float *shift, const float *mean, const float *rstd, const float *c1,
                               const float *c2, const int *ws, const int *order, float *out_cl, void *stream);

/* EdgeConv tail = the rest of DGCNN_Propagation's layer behind the (linear) 1x1 convolution
 * (openpoints/models/backbone/transformer.py:366-379: Conv2d -> GroupNorm(groups) -> LeakyReLU(slope) ->
 * max over the k neighbours), fused.  With P = W_d x_k (b,c,nk) and Q = (W_q - W_d) x_q (b,c,nq) from the caller's
 * GEMMs, y[b,:,i,j] = P[b,:,idx[b,i,j]] + Q[b,:,i] is the convolution's output; out (b,c,nq) =
 * max_j LeakyReLU(GroupNorm(y)).  The (b,c,nq,k) tensor is never materialised.  Saved for the gradient: ysel
 * (b,c,nq) the selected y, ysum (b,c,nq) = sum_j y, jsel (b,c,nq) uint8 the selected slot, stats (b,groups,2) =
 * (mean, 1/sqrt(var+eps)).  _grad writes grad_p (b,c,nk), grad_q (b,c,nq), grad_gamma (c), grad_beta (c) in full
 * (no atomics: deterministic).  workspace: geot_edgeconv_ws_bytes() bytes of scratch, contents irrelevant.
 * Needs slope >= 0, c % groups == 0, k <= 255 and rows that fit the LDS (nk <= 38400, nq <= 17066):
 * geot_edgeconv_eligible() tells; callers fall back to the composed ops otherwise. */
int geot_edgeconv_eligible(int b, int c, int nq, int nk, int k, int groups);
long long geot_edgeconv_ws_bytes(int b, int c, int nq, int nk, int k);
int geot_edgeconv_gn_max(int b, int c, int nq, int nk, int k, int groups, float eps, float slope, const float *P,
                         const float *Q, const int *idx, const float *gamma, const float *beta, float *out,
                         float *ysel, float *ysum, unsigned char *jsel, float *stats, void *workspace,
                         long long ws_bytes, void *stream);
int geot_edgeconv_gn_max_grad(int b, int c, int nq, int nk, int k, int groups, float slope, const float *P,
                              const float *Q, const int *idx, const float *gamma, const float *beta,
                              const float *ysel, const float *ysum, const unsigned char *jsel, const float *stats,
                              const float *grad_out, float *grad_p, float *grad_q, float *grad_gamma,
                              float *grad_beta, void *workspace, long long ws_bytes, void *stream);
/* The gradient's reverse index of idx (pairs grouped by target, ascending pair id) depends on the kNN graph alone:
 * geot_edgeconv_rix_build writes it into `rix` (geot_edgeconv_rix_ints ints) wherever the caller has the graph early;
 * geot_edgeconv_gn_max_grad_rix = geot_edgeconv_gn_max_grad with that index instead of idx (7 launches fewer on the
 * gradient's stream). */
long long geot_edgeconv_rix_ints(int b, int nq, int nk, int k);
int geot_edgeconv_rix_build(int b, int nq, int nk, int k, const int *idx, int *rix, long long rix_ints, void *stream);
int geot_edgeconv_gn_max_grad_rix(int b, int c, int nq, int nk, int k, int groups, float slope, const float *P,
                                  const float *Q, const int *rix, const float *gamma, const float *beta,
                                  const float *ysel, const float *ysum, const unsigned char *jsel, const float *stats,
                                  const float *grad_out, float *grad_p, float *grad_q, float *grad_gamma,
                                  float *grad_beta, void *workspace, long long ws_bytes, void *stream);

/* EdgeConv graph feature = DGCNN_Propagation.get_graph_feature
 * (openpoints/models/backbone/transformer.py:343-364: transpose + fancy-index gather + permute +
 * expand + cat), in one pass:  x_q (b,c,nq), x_k (b,c,nk), idx (b,nq,k) int32 neighbours in x_k ->
 * out (b,2c,nq,k) = cat(x_k[idx] - x_q, x_q).  _grad accumulates into grad_xq (b,c,nq) and grad_xk
 * (b,c,nk); workspace = b*nk*c zero-filled floats. */
int geot_graph_feature(int b, int c, int nq, int nk, int k, const float *x_q, const float *x_k,
                       const int *idx, float *out, void *stream);
int geot_graph_feature_grad(int b, int c, int nq, int nk, int k, const float *grad_out, const int *idx,
                            float *grad_xq, float *grad_xk, float *workspace, void *stream);

/* ---- three_nn / three_interpolate ------------------------------------------
 * pointnet2/_ext_src/src/interpolate_gpu.cu:64-71, 106-115, 148-157
 * openpoints/cpp/pointnet2_batch/src/interpolate_gpu.cu (…_launcher_fast)
 * dist2 holds SQUARED distances (callers take sqrt). */
int geot_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                  int *idx, void *stream);
int geot_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                           const float *weight, float *out, void *stream);
int geot_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                const float *weight, float *grad_points, void *stream);

/* ---- kNN --------------------------------------------------------------------
 * Heap-ordered, offset-batched: pointops/src/knnquery/knnquery_cuda_kernel.cu:111-116
 * knnquery_cuda_launcher (extern "C" in knnquery_cuda_kernel.h:9-17).
 * idx (m,nsample) global indices, dist2 (m,nsample) squared.  nsample <= 256. */
int geot_knnquery_heap(int b, int m, int nsample, const float *xyz, const float *new_xyz,
                       const int *offset, const int *new_offset, int *idx, float *dist2,
                       void *stream);
/* Sorted kNN in d <= 32 dimensions (feature_space_loss's neighbours among the 17-dim soft-max vectors,
 * utils/insT_loss.py:19: knn_point = cdist + topk there): query (b,nq,d), ref (b,nr,d), k <= 64; ascending
 * (d2, index); d2 accumulated over the dimensions in order, un-contracted. */
int geot_knn_sorted_nd(int b, int nq, int nr, int d, int k, const float *query, const float *ref, int *idx,
                       float *dist2, void *stream);
/* geot_knnquery_heap for uniform batches (every segment n_per support points, m_per queries, as
 * pointops.knn builds them): sorted (nsample+1)-NN from the grid search; queries whose first nsample+1
 * distances are strictly increasing have a unique answer and are copied, the rest (ties, too few
 * candidates) go through the literal heap.  Identical output.  workspace: geot_knnquery_heap_ws_bytes. */
long long geot_knnquery_heap_ws_bytes(int b, int n_per, int m_per, int nsample);
int geot_knnquery_heap_ws(int b, int n_per, int m_per, int nsample, const float *xyz, const float *new_xyz,
                          const int *offset, const int *new_offset, int *idx, float *dist2, void *workspace,
                          long long ws_bytes, void *stream);
/* Sorted brute-force kNN: the contract of the un-vendored knn_cuda.KNN
 * (openpoints/models/backbone/transformer.py:280,293,313,353) and of
 * knn_point = cdist + topk (openpoints/models/layers/knn.py:7-20).
 * query (b,nq,3), ref (b,nr,3) -> idx (b,nq,k) int32, dist2 (b,nq,k) squared,
 * ascending by (dist2, index); missing neighbours are (inf, 0).  k <= 256. */
int geot_knn_sorted(int b, int nq, int nr, int k, const float *query, const float *ref, int *idx,
                    float *dist2, void *stream);

/* ---- openpoints pointops, channels-last -----
 * openpoints/cpp/pointops/src/{grouping,interpolation,subtraction,aggregation}/…_cuda_kernel.cu */
int geot_grouping_cl(int m, int nsample, int c, const float *input, const int *idx, float *out,
                     void *stream);
int geot_grouping_cl_grad(int m, int nsample, int c, const float *grad_out, const int *idx,
                          float *grad_in, void *stream);
int geot_interpolation_cl(int n, int c, int k, const float *input, const int *idx,
                          const float *weight, float *out, void *stream);
int geot_interpolation_cl_grad(int n, int c, int k, const float *grad_out, const int *idx,
                               const float *weight, float *grad_in, void *stream);
int geot_subtraction_cl(int n, int nsample, int c, const float *in1, const float *in2,
                        const int *idx, float *out, void *stream);
int geot_subtraction_cl_grad(int n, int nsample, int c, const int *idx, const float *grad_out,
                             float *grad_in1, float *grad_in2, void *stream);
int geot_aggregation_cl(int n, int nsample, int c, int w_c, const float *input,
                        const float *position, const float *weight, const int *idx, float *out,
                        void *stream);
int geot_aggregation_cl_grad(int n, int nsample, int c, int w_c, const float *input,
                             const float *position, const float *weight, const int *idx,
                             const float *grad_out, float *grad_in, float *grad_position,
                             float *grad_weight, void *stream);

/* ---- fused SetAbstraction body ---------------
 * Replaces the chain QueryAndGroup (pointnet2/pointnet2_utils.py:343-358: two
 * grouping_operation + centre subtraction + cat) -> SharedMLP
 * (pointnet2/pytorch_utils.py:8-33: 1x1 Conv2d + BatchNorm + ReLU stack) ->
 * max_pool2d over nsample (pointnet2/pointnet2_modules.py:360-363), for inference
 * (BatchNorm folded into the conv by the caller).  fp32 MFMA.
 * xyz (b,n,3), new_xyz (b,npoint,3), features (b,c_feat,n) or NULL when c_feat==0,
 * idx (b,npoint,nsample) from ball_query, out (b, widths[nlayers-1], npoint).
 * widths = host array of the nlayers output widths (each <= 256, nlayers <= 4);
 * relu_mask bit l = ReLU after layer l.  params = device block, per layer
 * W^T zero-padded to [kp][cp] then bias [cp], with kp_0 = 3+c_feat rounded up to even,
 * kp_l = cp_{l-1}, cp = width rounded up to 32/64/128/256;
 * geot_sa_param_floats returns its length (or -1 for unsupported shapes).
 * nsample must be 8, 16 or a multiple of 32. */
int geot_sa_param_floats(int c_feat, int nlayers, const int *widths);
int geot_sa_group_mlp_max(int b, int n, int npoint, int nsample, int c_feat, const float *xyz,
                          const float *new_xyz, const float *features, const int *idx,
                          float xyz_scale, int nlayers, const int *widths, int relu_mask,
                          const float *params, float *out, void *stream);

/* ---- NTM: per-point instance-dependent transition matrix ------
 * c = class count, 1 <= c <= GEOT_NTM_MAX_C (32).  c == 17 (the reference's num_classes,
 * cfgs/tooth_semi/default.yaml:29) runs the kernels specialised for it (MFMA / LDS tiles of 17 x 17 rows); every
 * other count -- the reference builds its heads and losses for any nclasses (transformer.py:1104-1110,
 * insT_loss.py:62-67) -- runs run-time-C kernels with the same arithmetic.  Three entry points exist for c == 17
 * only and return hipErrorInvalidValue otherwise, each with a generic counterpart: geot_ntm_sig_t_mean_grad_w
 * (use _grad_raw + one GEMM), geot_ntm_threed_loss_fwd_graph / _grad_graph (use geot_ntm_threed_loss[_ord] +
 * geot_ntm_threed_loss_grad).
 *
 * sig_t_mean.forward (openpoints/models/backbone/transformer.py:1120-1131), fused:
 *   p (b,c,n) softmax probs, W (c,c,2c) = the c Linear(2c->c, bias=False) weights stacked
 *   [kk][out][in], cm (c,c) class means -> ins_T (b*n,c,c): row kk = clamp([p_i, cm[kk]] @ W[kk]^T,
 *   1e-5, 1-1e-5), L1-normalised.
 * _grad_raw: d loss / d (pre-clamp row) for a given d loss / d ins_T (the caller finishes the tiny
 *   weight-gradient GEMM, as nn.Linear's backward does in the reference). */
int geot_ntm_sig_t_mean(int b, int n, int c, const float *p, const float *W, const float *cm,
                        float *ins_T, void *stream);
int geot_ntm_sig_t_mean_grad_raw(int b, int n, int c, const float *p, const float *W, const float *cm,
                                 const float *grad_ins_T, float *grad_raw, void *stream);
/* Weight gradient of sig_t_mean without materialising d raw: grad_W (c,c,2c) += d loss / d W given
 * grad_ins_T (b*n,c,c); workspace = geot_ntm_sig_t_mean_ws_floats(b, n) floats of scratch. */
long long geot_ntm_sig_t_mean_ws_floats(int b, int n);
int geot_ntm_sig_t_mean_grad_w(int b, int n, int c, const float *p, const float *W, const float *cm,
                               const float *grad_ins_T, float *grad_W, float *workspace, void *stream);
/* Logit correction (examples/segmentation/train.py:549-552), fused:
 *   newT_i = L1-normalise(lam * ema_t + (1-lam) * ins_T_i); out[:, i] = logits[:, i]^T @ newT_i.
 *   logits/out (b,c,n), ins_T (b*n,c,c), ema_t (c,c).  _grad: grad_logits (b,c,n) and grad_ins_T are
 *   written in full, grad_ema_t (c,c) is accumulated into (pre-zero it). */
int geot_ntm_correct(int b, int n, int c, float lam, const float *logits, const float *ins_T,
                     const float *ema_t, float *out, void *stream);
int geot_ntm_correct_grad(int b, int n, int c, float lam, const float *logits, const float *ins_T,
                          const float *ema_t, const float *grad_out, float *grad_logits,
                          float *grad_ins_T, float *grad_ema_t, void *stream);
/* Class anchors, train.py:505-526: class_T (c,c) row cc = eta[b*, :, n*] of the point with the largest
 * eta[b, cc, n] (eta (b,c,n) fp32: the weak view's soft-max), the first maximum in flattened (b, n) order as
 * torch.argmax picks it; v_star (c, nullable) receives the maxima (the multi-rank exchange compares them). */
int geot_ntm_class_anchors(int b, int n, int c, const float *eta, float *class_T, float *v_star, void *stream);
/* Class-level transition block, train.py:505-557 once the anchor rows class_T (c,c) are gathered: Gaussian
 * tooth-adjacency prior from sigma (c) over the label projection proj (c) (train.py:48), blend, the three
 * `X / X.sum(1)` normalisations exactly as written there (column k divided by row-sum k), EMA.  c <= 32.
 * Writes ema_t_corr, ema_t_next, prior_T (c,c) and, when ema_t_keep is not NULL, a copy of ema_t there (a caller
 * that keeps ema_t in one persistent buffer overwrites it with ema_t_next, train.py:556-557, before backward).
 * _grad writes d/d sigma (c) given d/d ema_t_corr and d/d prior_T (either may be NULL); sigma is the only
 * learnable input.  geo_lambda / ema_decay are DOUBLES (ABI 6): the reference multiplies fp32 tensors by the Python
 * floats cfg.geo_lambma and (1 - cfg.geo_lambma) (train.py:533-534, 540-545), i.e. by fl32(0.999) and fl32(1 - 0.999);
 * from a float argument the complement could only be formed as 1.f - fl32(0.999), 4.7e-5 off -- which is the whole
 * sigma gradient's relative error, since that gradient is proportional to (1 - geo)(1 - decay). */
int geot_ntm_class_transition(int c, double geo_lambda, double ema_decay, const float *class_T, const float *sigma,
                              const float *ema_t, const float *proj, float *ema_t_corr, float *ema_t_next,
                              float *prior_T, float *ema_t_keep, void *stream);
int geot_ntm_class_transition_grad(int c, double geo_lambda, double ema_decay, const float *class_T,
                                   const float *sigma, const float *ema_t, const float *proj,
                                   const float *grad_ema_t_corr, const float *grad_prior_T, float *grad_sigma,
                                   void *stream);
/* _grad_ws: as _grad, with grad_ema_t reduced through geot_ntm_correct_ws_floats(b, n) floats of scratch. */
long long geot_ntm_correct_ws_floats(int b, int n);
int geot_ntm_correct_grad_ws(int b, int n, int c, float lam, const float *logits, const float *ins_T,
                             const float *ema_t, const float *grad_out, float *grad_logits,
                             float *grad_ins_T, float *grad_ema_t, float *workspace, void *stream);
/* threeD_space_loss (utils/insT_loss.py:68-110) over a given kNN graph:
 *   positions (b,n,3), labels (b,n) int32, ins_T (b*n,c,c), nbr (b,n,k) int32 local neighbour ids
 *   (the reference uses knn_point(k+1)[..., 1:]); per_point (b*n) = sum_j w_ij |T_i-T_j|^2 /
 *   (sum_j w_ij + 1e-3), w_ij = [label_i==label_j] exp(-|p_i-p_j|^2/(2 sigma^2)); loss = mean.
 *   _grad accumulates grad_scale * d(sum per_point)/d ins_T into grad_ins_T (pre-zero it);
 *   pass grad_scale = upstream_grad / (b*n).  k <= 64. */
int geot_ntm_threed_loss(int b, int n, int c, int k, float sigma, const float *positions,
                         const int *labels, const float *ins_T, const int *nbr, float *per_point,
                         void *stream);
int geot_ntm_threed_loss_grad(int b, int n, int c, int k, float sigma, float grad_scale,
                              const float *positions, const int *labels, const float *ins_T,
                              const int *nbr, float *grad_ins_T, void *stream);
/* _grad_ws: the same gradient through an atomic-free gather over the kNN graph and its reverse (built in
 * the workspace, geot_ntm_threed_loss_ws_bytes(b, n, k) bytes); 3x faster when labels are spatially
 * coherent, i.e. on real scans, where most edges are live.  Falls back to _grad without a workspace. */
long long geot_ntm_threed_loss_ws_bytes(int b, int n, int k);
int geot_ntm_threed_loss_grad_ws(int b, int n, int c, int k, float sigma, float grad_scale,
                                 const float *positions, const int *labels, const float *ins_T,
                                 const int *nbr, const int *order, float *grad_ins_T, void *workspace,
                                 long long ws_bytes, void *stream);
/* `order` (b*n int32 global point ids, or NULL) = the order in which points are PROCESSED; results do not
 * depend on it.  With geot_spatial_order's output consecutive waves work on spatial neighbours and the
 * gathered (1156-byte) rows of a point's graph neighbours are found in L2: 1.5x on randomly ordered scans. */
int geot_ntm_threed_loss_ord(int b, int n, int c, int k, float sigma, const float *positions,
                             const int *labels, const float *ins_T, const int *nbr, const int *order,
                             float *per_point, void *stream);
/* Forward that leaves the graph (reverse adjacency with fixed 64-slot lists + overflow list, edge weights,
 * normalisers) in `graph` (geot_ntm_threed_graph_bytes(b, n, k) bytes), and the backward that consumes it:
 * the backward is then the gather alone.  Same results as the entry points above, except that
 * _grad_graph WRITES grad_ins_T in full (it need not be zero-filled). */
long long geot_ntm_threed_graph_bytes(int b, int n, int k);
int geot_ntm_threed_loss_fwd_graph(int b, int n, int c, int k, float sigma, const float *positions,
                                   const int *labels, const float *ins_T, const int *nbr, const int *order,
                                   float *per_point, void *graph, long long graph_bytes, void *stream);
int geot_ntm_threed_loss_grad_graph(int b, int n, int c, int k, float grad_scale, const float *upstream,
                                    const float *ins_T, const int *nbr, const int *order, const void *graph,
                                    long long graph_bytes, float *grad_ins_T, void *stream);
/* upstream: optional DEVICE scalar multiplied into grad_scale (autograd's incoming gradient), so the host
 * does not have to read it back; NULL = 1. */
int geot_spatial_order(int b, int n, const float *xyz, int *order, void *workspace, long long ws_bytes,
                       void *stream); /* workspace: geot_knn_grid_ws_bytes(b, n) bytes, 16-byte aligned */
/* feature_space_loss (utils/insT_loss.py:9-58; disabled in the shipped cfg, use_feat_loss): same graph
 * kernel over feat_dim-dimensional features (b,n,feat_dim) with SIGNED weights
 * w_ij = (label_i == label_j ? +1 : -1) exp(-|f_i-f_j|^2/(2 sigma^2)) and no per-point normalisation:
 * per_point (b*n) = sum_j w_ij |T_i-T_j|^2; the reference's loss is sum(per_point) / (b*n*k).
 * _grad: pass grad_scale = upstream_grad / (b*n*k). */
int geot_ntm_feature_loss(int b, int n, int c, int k, int feat_dim, float sigma, const float *feats,
                          const int *labels, const float *ins_T, const int *nbr, float *per_point,
                          void *stream);
int geot_ntm_feature_loss_grad(int b, int n, int c, int k, int feat_dim, float sigma, float grad_scale,
                               const float *feats, const int *labels, const float *ins_T, const int *nbr,
                               float *grad_ins_T, void *stream);

/* ---- dataloader-side ops ---------------------------------------------------------------
 * geot_grid_subsampling replaces cpp_subsampling.compute (openpoints/cpp/subsampling/wrapper.cpp:58-285 ->
 * grid_subsampling/grid_subsampling.cpp:4-106): voxel size sample_dl, points (n,3), optional features
 * (n,fdim) and integer labels (n,ldim).  Outputs have room for n rows; *out_count (device int) receives the
 * number of voxels M; rows 0..M-1 are the voxel barycentres (fp32 sums in input order, bit-identical to the
 * reference), mean features and majority labels, by ascending voxel key (the reference emits the same rows in
 * its hash map's iteration order; on a label tie it takes the first maximum in that order, here the smallest
 * tied label).  ws: geot_grid_subsampling_ws_bytes(n) bytes of device scratch (sized for the current device:
 * -1 when the process has no GPU). */
long long geot_grid_subsampling_ws_bytes(int n);
int geot_grid_subsampling(int n, int fdim, int ldim, float sample_dl, const float *points, const float *features,
                          const int *labels, float *out_points, float *out_features, int *out_labels,
                          int *out_count, void *ws, long long ws_bytes, void *stream);
/* pc_norm of openpoints/dataset/tooth_semi/tooth_dataset.py:108-114: stats (4 floats, device) = centroid xyz
 * and scale = max row norm of the centred cloud.  ws: geot_pc_norm_ws_bytes() bytes. */
long long geot_pc_norm_ws_bytes(void);
int geot_pc_norm_stats(int n, const float *points, float *stats, void *ws, long long ws_bytes, void *stream);
/* tooth_dataset.py:132-147: out_points[i] = (points[selected[i]] - centroid) / scale (m,3); with labels (n)
 * also out_labels (m) int64 and class_weights (num_classes) = histogram of the gathered labels / m (inf -> 0).
 * selected NULL = identity (m == n).  hist_ws: num_classes + 1 ints; the last one is set non-zero when an
 * entry of selected was out of range (numpy raises IndexError there; the row is then read from point 0). */
int geot_cloud_sample(int n, int m, int num_classes, const float *points, const int *labels,
                      const long long *selected, const float *stats, float *out_points, long long *out_labels,
                      float *class_weights, int *hist_ws, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GEOT_HIP_H */
