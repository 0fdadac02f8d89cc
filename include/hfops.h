/*
 * hfops.h -- C ABI of libhfops.so: the HeteroFusionRCNN point-cloud hot path as
 * hand-written HIP kernels for MI355X (gfx950 / CDNA4).
 *
 * This is the drop-in boundary.  Each entry point replaces one launcher function that the
 * reference's TF custom-op wrappers declare `extern` in their .cpp and define in their .cu
 * (cited per function).  Argument lists keep the reference's order and meaning; the ABI adds
 *   - a trailing `hf_stream_t stream` (a hipStream_t; NULL = the default stream), and
 *   - an `int` status return (HF_OK / HF_E*), never exit() (the reference exit(-1)s at
 *     interpolate/tf_interpolate_g.cu:82-86 and bev_iou/bev_iou.cpp:14-22).
 *
 * Conventions (same as the reference op boundary):
 *   - every pointer is a DEVICE pointer to a dense row-major buffer owned by the caller;
 *   - nothing here allocates, frees, copies to the host or synchronises; all work is
 *     enqueued on `stream`.  Ops that need scratch take it from the caller
 *     (hf_*_workspace() says how much);
 *   - outputs that the reference zero-/one-fills with cudaMemset inside OpKernel::Compute
 *     (tf_grouping.cpp:204, tf_sampling.cpp:174, tf_interpolate.cpp:91-92,130,178,
 *     bev_iou.cpp:175-176, tf_cropping.cpp:171-176) are initialised INSIDE these functions;
 *   - shape/attribute violations that the reference rejects with OP_REQUIRES return
 *     HF_EINVAL and launch nothing;
 *   - stateless, re-entrant, safe to call from several host threads on different streams.
 *
 * Arithmetic contract: IEEE fp32, no FMA contraction, expressions evaluated in the order the
 * reference writes them; integer outputs are bit-exact with the reference semantics
 * (tie rules documented per function).
 */
#ifndef HFOPS_H
#define HFOPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HF_OK 0
#define HF_EINVAL (-1)     /* shape / attribute violation (OP_REQUIRES in the reference) */
#define HF_EHIP (-2)       /* a HIP runtime call or launch failed; see hf_last_hip_error() */
#define HF_EWORKSPACE (-3) /* caller-provided workspace missing or too small */

typedef void *hf_stream_t; /* hipStream_t */

/* library / error helpers */
const char *hf_version(void);
const char *hf_strerror(int status);
int hf_last_hip_error(void); /* hipError_t of the last HF_EHIP on this thread */

/* ------------------------------------------------------------------ sampling/ */

/* replaces farthestpointsamplingLauncher(b,n,m,inp,temp,out)  sampling/tf_sampling.cpp:94
 * (kernel sampling/tf_sampling_g.cu:105-170).  inp (b,n,3) -> out (b,m) int32; out[:,0] = 0.
 * Tie rule among equal max-min distances: smallest (k mod 512), then smallest k.
 * `temp` is the reference's (32,n) float scratch; this implementation keeps the running
 * distances on chip when n <= hf_fps_onchip_limit() and then ignores it (may be NULL);
 * above that limit it needs hf_fps_workspace(b,n) bytes there. */
int hf_farthest_point_sample(int b, int n, int m, const float *inp, float *temp, int *out, hf_stream_t stream);
/* The same op with the kernel forced (the tests run every kernel against the oracle): HF_FPS_PLAIN = points and running
 * distances in registers, `threads` = 256 / 512 / 1024 per cloud (0: by size); HF_FPS_BUCKET = the spatially bucketed kernel
 * with exact pruning, `threads` = 1024 (16 waves x 16 buckets) or anything else = 512 (8 waves x 32 buckets, the default);
 * HF_FPS_WAVE = one wave per cloud, no barrier per round (n <= 512: the RoI clouds of the second stage);
 * HF_FPS_AUTO = what hf_farthest_point_sample picks (one wave per cloud up to 512 points, bucketed from 8192).  Identical output
 * whatever the choice. */
#define HF_FPS_AUTO 0
#define HF_FPS_PLAIN 1
#define HF_FPS_BUCKET 2
#define HF_FPS_WAVE 3
int hf_farthest_point_sample_variant(int kernel, int threads, int b, int n, int m, const float *inp, float *temp, int *out,
                                     hf_stream_t stream);
size_t hf_fps_workspace(int b, int n);
int hf_fps_onchip_limit(void);

/* replaces gatherpointLauncher(b,n,m,inp,idx,out)  sampling/tf_sampling.cpp:125 (kernel :172-181) */
int hf_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out, hf_stream_t stream);

/* replaces cudaMemset + scatteraddpointLauncher(b,n,m,out_g,idx,inp_g)  sampling/tf_sampling.cpp:150,174 */
int hf_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx, float *inp_g, hf_stream_t stream);

/* ------------------------------------------------------------------ grouping/ */

/* replaces queryBallPointLauncher(b,n,m,radius,nsample,xyz1,xyz2,idx,pts_cnt)  grouping/tf_grouping.cpp:66
 * (kernel grouping/tf_grouping_g.cu:3-36).  xyz1 (b,n,3) data, xyz2 (b,m,3) queries ->
 * idx (b,m,nsample): the first nsample data indices, ascending, with
 * max(sqrtf(|q-p|^2),1e-20f) < radius; remaining slots repeat the first hit; a query with
 * no hit gets a row of zeros (undefined in the reference).  pts_cnt (b,m) = #hits (<= nsample),
 * may be NULL.  radius > 0 and nsample > 0 required (tf_grouping.cpp:70-74). */
int hf_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2,
                        int *idx, int *pts_cnt, hf_stream_t stream);

/* replaces groupPointLauncher(b,n,c,m,nsample,points,idx,out)  grouping/tf_grouping.cpp:142 (kernel :40-57) */
int hf_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out,
                   hf_stream_t stream);

/* replaces cudaMemset + groupPointGradLauncher(...)  grouping/tf_grouping.cpp:173,204 (kernel :61-78) */
int hf_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                        float *grad_points, hf_stream_t stream);

/* The CSR inverse of an index tensor, built with the geometry (coordinates only): idx holds `total` values per cloud, each in
 * [0, m) (group_point / kNN tables: total = queries * nsample; three_nn: total = 3 n).  offsets (b, m+1), entries (b, total):
 * for every target the flat positions that name it, ascending.  m <= 19968 (two LDS words per target).
 * hf_group_point_grad_gather: the gradient of hf_group_point / hf_group_point_into over that inverse -- every data point sums
 * the grad_out rows that name it in ascending (query, slot) order (the order of the reference's CPU loop,
 * grouping/test/query_ball_point.cpp:53-66) and its row is written once: no atomics, no zero fill, deterministic.
 * grad_out rows have stride `width`, the gathered columns start at `col` (width = c, col = 0 for plain group_point). */
int hf_index_inverse(int b, long long total, int m, const int *idx, int *offsets, int *entries, hf_stream_t stream);
int hf_group_point_grad_gather(int b, int n, int c, int m, int nsample, int width, int col, const float *grad_out,
                               const int *offsets, const int *entries, float *grad_points, hf_stream_t stream);

/* Fused query_ball_point + group_point(xyz) in ONE launch: the pair of calls at
 * hf/core/feature_extractors/pointnet_util.py:48-49,258-259.  Outputs are exactly those of the
 * two separate ops; grouped_xyz (b,m,nsample,3) optionally has the query subtracted
 * (`center` != 0: pointnet_util.py:50-52, same fp32 subtraction).  idx / pts_cnt may be NULL. */
int hf_query_ball_group_xyz(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2,
                            int center, int *idx, int *pts_cnt, float *grouped_xyz, hf_stream_t stream);

/* The same op pair with a caller-owned workspace (hf_ball_query_workspace(b, n) bytes, 16-byte aligned) and an explicit
 * kernel choice.  With a workspace the batched shapes of a train step (more workgroups than one round of the single-launch
 * kernel) build the cell structure of every cloud ONCE (counting sort by hashed cell) and answer all query tiles from it.
 * variant: HF_BQ_AUTO by shape, or one kernel forced (HF_BQ_CELL single launch, HF_BQ_BRUTEFORCE, HF_BQ_SORTED) -- the tests
 * run every kernel against the oracle this way.  grouped_xyz may be NULL (query_ball_point alone); outputs identical to
 * hf_query_ball_point / hf_query_ball_group_xyz whatever the variant. */
#define HF_BQ_AUTO 0
#define HF_BQ_CELL 1
#define HF_BQ_BRUTEFORCE 2
#define HF_BQ_SORTED 3
size_t hf_ball_query_workspace(int b, int n);
int hf_query_ball_group_xyz_ws(int variant, int b, int n, int m, float radius, int nsample, const float *xyz1,
                               const float *xyz2, int center, int *idx, int *pts_cnt, float *grouped_xyz, void *workspace,
                               size_t workspace_bytes, hf_stream_t stream);

/* replaces selectionSortLauncher(b,n,m,k,dist,outi,out)  grouping/tf_grouping.cpp:108 (kernel :83-123) */
int hf_select_top_k(int b, int n, int m, int k, const float *dist, int *outi, float *out, hf_stream_t stream);

/* ---------------------------------------------------------------- interpolate/ */

/* replaces three_nn_gpu(b,n,m,unknown,known,dist2,idx)  interpolate/tf_interpolate.cpp:60
 * (kernel interpolate/tf_interpolate_g.cu:22-65).  Three smallest (squared distance, index)
 * pairs per unknown point, earlier index first on ties; unused slots (m<3) = +inf / 0. */
int hf_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                hf_stream_t stream);

/* replaces three_interpolate_gpu(b,c,m,n,points,idx,weight,out)  tf_interpolate.cpp:99 (kernel :90-110)
 * channel-first: points (b,c,m) -> out (b,c,n) */
int hf_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx, const float *weight,
                         float *out, hf_stream_t stream);

/* replaces cudaMemset + three_interpolate_grad_gpu(b,c,n,m,grad_out,idx,weight,grad_points)
 * tf_interpolate.cpp:141,178 (kernel :133-155); channel-first */
int hf_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                              const float *weight, float *grad_points, hf_stream_t stream);

/* Channel-last forms = what the Python surface interpolate/tf_interpolate.py:26-49 exposes
 * (it transposes (b,m,c)->(b,c,m), calls the op, transposes back).  Same values, no transposes:
 * points (b,m,c) -> out (b,n,c);  grad_out (b,n,c) -> grad_points (b,m,c). */
int hf_three_interpolate_cl(int b, int m, int c, int n, const float *points, const int *idx, const float *weight,
                            float *out, hf_stream_t stream);
int hf_three_interpolate_cl_grad(int b, int n, int c, int m, const float *grad_out, const int *idx,
                                 const float *weight, float *grad_points, hf_stream_t stream);

/* -------------------------------------------------------------------- bev_iou/ */

/* replaces cudaMemset x2 + compute_bev_iou_gpu(num_a,boxes_a,num_b,boxes_b,ans_overlap,ans_iou)
 * bev_iou/bev_iou.cpp:144,175-177 (kernel bev_iou_g.cu:240-254).  boxes (N,5) [x1,y1,x2,y2,ry].
 * ans_overlap / ans_iou (num_a,num_b); either may be NULL.  num_a > 0, num_b > 0 required. */
int hf_compute_bev_iou(int num_a, const float *boxes_a, int num_b, const float *boxes_b, float *ans_overlap,
                       float *ans_iou, hf_stream_t stream);

/* replaces oriented_nms_gpu(boxes,mask,boxes_num,thresh)  bev_iou/bev_iou.cpp:40 (kernel bev_iou_g.cu:256-298)
 * mask (n, ceil(n/64)) uint64, every tile written as in the reference. */
int hf_nms_mask(const float *boxes, unsigned long long *mask, int boxes_num, float nms_overlap_thresh,
                hf_stream_t stream);

/* The whole OrientedNMSOp::Compute (bev_iou.cpp:60-116) without its cudaMalloc / blocking D2H /
 * host sweep / H2D: mask kernel + device-resident greedy sweep.  boxes are score-sorted (N,5);
 * keep (N) int32 = kept indices ascending, tail padded with keep[0]; *num_kept (device int,
 * may be NULL) = count before padding.  thresh >= 0, n > 0 required.
 * Workspace per frame (hf_oriented_nms_workspace(n) bytes, a multiple of 256; the base 16-byte aligned): the dense
 * mask (n * ceil(n/64) words), then -- each part padded to 256 bytes -- one counter per column block, one list of
 * 2048 (word, row) entries per column block (the nonzero words right of the diagonal, what the sweep reads), the
 * n transposed diagonal words, an 80-byte table entry per box (cos / sin / corners, computed once per call) and the n
 * mask words next to the diagonal.  About 1.5x the dense mask at n = 9000 (15.6 MB). */
size_t hf_oriented_nms_workspace(int n);
int hf_oriented_nms(const float *boxes, int n, float thresh, int *keep, int *num_kept, void *workspace,
                    size_t workspace_bytes, hf_stream_t stream);
/* The reference maps OrientedNMS over the frames of a batch with tf.map_fn (rpn_model.py:683-687), i.e. one
 * op call (and one host round trip) per frame.  Batched form: boxes (frames, n, 5), keep (frames, n),
 * num_kept (frames) or NULL, workspace = frames * hf_oriented_nms_workspace(n) bytes; one launch pair. */
int hf_oriented_nms_batched(int frames, const float *boxes, int n, float thresh, int *keep, int *num_kept,
                            void *workspace, size_t workspace_bytes, hf_stream_t stream);

/* ------------------------------------------------------------------- cropping/ */

/* replaces cudaMemset x6 + pccropandsample_gpu(...)  cropping/tf_cropping.cpp:100,171-180
 * (kernel cropping/tf_cropping_g.cu:43-132).  `mask`/`crop_mask`/`non_empty_box` are 1-byte bools.
 * Deterministic: the points of each box are taken in ascending point index (the reference's
 * order is atomic-arrival order; ascending is the member of its output set that a
 * single-thread launch produces).  resize > 0 required; box_ind values must lie in [0,batch). */
int hf_pc_crop_and_sample(const float *pts, const float *fts, const float *intensities, const unsigned char *mask,
                          const float *boxes, const int *box_ind, int num_boxes, int batch, int npts, int resize,
                          int channel, int intensity_channel, float *crop_pts, float *crop_fts,
                          float *crop_intensities, unsigned char *crop_mask, int *crop_ind,
                          unsigned char *non_empty_box, hf_stream_t stream);

/* replaces cudaMemset + pccropandsamplegradfts_gpu(box_ind,crop_ind,grad_crop_fts,num_boxes,npts,resize,channel,grad_fts)
 * cropping/tf_cropping.cpp:192,225.  `batch` added so the zero fill can happen here. */
int hf_pc_crop_and_sample_grad_fts(const int *box_ind, const int *crop_ind, const float *grad_crop_fts,
                                   int num_boxes, int batch, int npts, int resize, int channel, float *grad_fts,
                                   hf_stream_t stream);

/* ------------------------------------------------- callers of the path: the grouped-point MLP */

/* k nearest data points of every query: the knn_point of grouping/tf_grouping.py:62-95 (and PointCNN's
 * knn_indices_general, hf/core/pointfly.py:185-212) without the dense (b,m,n) distance matrix.
 * xyz1 (b,n,3) data, xyz2 (b,m,3) queries -> val (b,m,k) squared distances ascending, idx (b,m,k) int32;
 * equal distances: lower index first (tf.nn.top_k's rule).  1 <= k <= min(n, 67). */
int hf_knn_point(int b, int n, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx,
                 hf_stream_t stream);

/* Training-mode batch norm (+ optional ReLU) over channel-last rows x (rows, c): the
 * tf_util.batch_norm_template + tf.nn.relu pair inside tf_util.conv2d
 * (hf/core/feature_extractors/tf_util.py:190-203,554-581; decay = 1 - momentum, epsilon = eps).
 * Batch statistics are reduced deterministically (per-block fp32 partials, fp64 final reduction).
 * running_mean / running_var (may be NULL) are updated as (1-momentum)*running + momentum*batch
 * (biased variance, as TensorFlow's moments).  save_mean / save_invstd (c floats each) feed the backward pass.
 * `relu` (all BatchNorm entry points that take it): bit 0 = ReLU after the normalisation; bit 1 = ELU applied to x on
 * load, before the statistics and the normalisation (pointfly.dense / conv2d: linear -> ELU -> batch_normalization,
 * hf/core/pointfly.py:371-497): 0 none, 1 BN+ReLU, 2 ELU+BN.
 * workspace: hf_bn_workspace(rows, c) bytes of device scratch. */
size_t hf_bn_workspace(long long rows, int c);
int hf_bn_relu_fwd_train(long long rows, int c, const float *x, const float *gamma, const float *beta, float eps,
                         float momentum, float *running_mean, float *running_var, int relu, float *y,
                         float *save_mean, float *save_invstd, void *workspace, size_t workspace_bytes,
                         hf_stream_t stream);
/* inference: y = relu?(gamma*invstd*(x-mean)+beta) with caller-provided mean / invstd */
int hf_bn_relu_fwd_eval(long long rows, int c, const float *x, const float *gamma, const float *beta,
                        const float *mean, const float *invstd, int relu, float *y, hf_stream_t stream);
/* backward of hf_bn_relu_fwd_train: dx (rows,c), dgamma (c), dbeta (c); the ReLU mask is recomputed from x.
 * dx_colsum (c floats, may be NULL) receives the column sums of dx: the bias gradient of the Linear / 1x1
 * convolution that produced x, for free in the same pass. */
int hf_bn_relu_bwd(long long rows, int c, const float *x, const float *dy, const float *gamma, const float *beta,
                   const float *save_mean, const float *save_invstd, int relu, float *dx, float *dgamma,
                   float *dbeta, float *dx_colsum, void *workspace, size_t workspace_bytes, hf_stream_t stream);
/* The same two passes with ROW STRIDES on the normalised tensor: y (forward) / dy (backward) may be a column slice of a wider
 * buffer (ld floats per row, ld >= c; 16-byte rows when c % 4 == 0).  That is how a BatchNorm output lands directly inside
 * the concat the reference builds next (pointcnn.py:104, :349: [lifted | gathered], [x-conv | skip]) and how its gradient is
 * read out of the concat's gradient in place: no tf.concat copy in either direction. */
int hf_bn_relu_fwd_train_ld(long long rows, int c, const float *x, const float *gamma, const float *beta, float eps,
                            float momentum, float *running_mean, float *running_var, int relu, float *y, long long ldy,
                            float *save_mean, float *save_invstd, void *workspace, size_t workspace_bytes, hf_stream_t stream);
int hf_bn_relu_bwd_ld(long long rows, int c, const float *x, const float *dy, long long lddy, const float *gamma,
                      const float *beta, const float *save_mean, const float *save_invstd, int relu, float *dx, float *dgamma,
                      float *dbeta, float *dx_colsum, void *workspace, size_t workspace_bytes, hf_stream_t stream);

/* knn_point (see hf_knn_point) with a scratch buffer: the data points of every cloud are binned into a 2-D grid over
 * its two widest axes first; each query then searches rings of cells around its own and stops once everything outside
 * the visited block is provably farther than its k-th best distance.  Same outputs as hf_knn_point (ties to the lower
 * index).  hf_knn_workspace returns 0 for n > 65536; hf_knn_point_sorted then runs hf_knn_point. */
size_t hf_knn_workspace(int b, int n);
int hf_knn_point_sorted(int b, int n, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx,
                        void *workspace, size_t workspace_bytes, hf_stream_t stream);

/* The concat of sample_and_group (pointnet_util.py:58-60) in one pass: out (b, m, nsample, width) rows are
 * [grouped_xyz (3), points[idx] (c), zeros], or with xyz_last != 0 the multi-scale module's order
 * [points[idx] (c), grouped_xyz (3), zeros] (pointnet_util.py:264); width >= 3 + c, width % 4 == 0.  Replaces
 * group_point + concat.  hf_group_concat_grad scatters the feature columns of grad_out back to grad_points (b, n, c). */
/* group_point into / its gradient out of a column slice of wider rows: out (b, m, nsample, width)[..., col : col + c] =
 * points[idx]; the other columns are not touched.  For concatenations whose other part is produced elsewhere (PointCNN's
 * [lifted coordinates | gathered features], pointcnn.py:96-99).  16-byte copies when c, width, col are multiples of 4. */
int hf_group_point_into(int b, int n, int c, int m, int nsample, int width, int col, const float *points, const int *idx,
                        float *out, hf_stream_t stream);
int hf_group_point_grad_from(int b, int n, int c, int m, int nsample, int width, int col, const float *grad_out,
                             const int *idx, float *grad_points, hf_stream_t stream);
int hf_group_concat(int b, int n, int c, int m, int nsample, int width, int xyz_last, const float *grouped_xyz,
                    const float *points, const int *idx, float *out, hf_stream_t stream);
int hf_group_concat_grad(int b, int n, int c, int m, int nsample, int width, int xyz_last, const float *grad_out,
                         const int *idx, float *grad_points, hf_stream_t stream);

/* three_nn (tf_interpolate.cpp:68-75) with a scratch buffer: the known points of every cloud are binned into a 2-D
 * grid first, each unknown point then searches rings of cells around its own (the k = 3 case of hf_knn_point_sorted).
 * Same outputs as hf_three_nn, bit for bit (ties to the lower index, +inf / 0 when fewer than 3 known points).
 * hf_three_nn_workspace returns 0 for m > 65536; hf_three_nn_sorted then runs hf_three_nn. */
size_t hf_three_nn_workspace(int b, int m);
int hf_three_nn_sorted(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                       void *workspace, size_t workspace_bytes, hf_stream_t stream);

/* Inverse of a three_nn index (tf_interpolate.cpp:68-75 produces idx): for every known point the (unknown point,
 * slot) pairs that reference it, as CSR.  offsets (b, m+1) int32, entries (b, 3n) int32 holding unknown*3+slot in
 * ascending order inside each bucket; idx values outside [0, m) are dropped.  m <= 8192. */
int hf_three_nn_inverse(int b, int n, int m, const int *idx, int *offsets, int *entries, hf_stream_t stream);
/* ThreeInterpolateGrad (tf_interpolate.cpp:39-48, 150-167) in gather form over that inverse, channel-last:
 * grad_out (b,n,c), weight (b,n,3) -> grad_points (b,m,c).  No atomics; sums in the order of the reference's
 * sequential loop (ascending unknown point, then slot). */
int hf_three_interpolate_cl_grad_gather(int b, int n, int c, int m, const float *grad_out, const float *weight,
                                        const int *offsets, const int *entries, float *grad_points, hf_stream_t stream);

/* batch statistics only (the first half of hf_bn_relu_fwd_train): mean / invstd of x (rows, c) over rows, running
 * estimates updated.  For callers that apply the normalisation elsewhere (fused into a pooling or a GEMM operand load). */
int hf_bn_stats(long long rows, int c, const float *x, float eps, float momentum, float *running_mean, float *running_var,
                float *save_mean, float *save_invstd, void *workspace, size_t workspace_bytes, hf_stream_t stream);

/* three_interpolate written into the concat of pointnet_fp_module (pointnet_util.py:311-313): out (b, n, width) rows are
 * [interpolated (c), skip (c1), zeros]; width >= c + c1, width % 4 == 0; skip (b, n, c1) may be NULL when c1 == 0.
 * hf_three_interpolate_concat_grad: gradient w.r.t. points from the first c columns of grad_out (b, n, width), gather
 * form over the inverse index (see hf_three_nn_inverse). */
int hf_three_interpolate_concat(int b, int m, int c, int n, int c1, int width, const float *points, const int *idx,
                                const float *weight, const float *skip, float *out, hf_stream_t stream);
int hf_three_interpolate_concat_grad(int b, int n, int c, int m, int width, const float *grad_out, const float *weight,
                                     const int *offsets, const int *entries, float *grad_points, hf_stream_t stream);

/* BN + ReLU + max over the k rows of every group, fused: the tail of a set-abstraction MLP
 * (tf_util.conv2d(..., bn=True) then tf.reduce_max(axis=[2]), pointnet_util.py:156-176).  z is (groups*k, c)
 * pre-BN; pooled is (groups, c).  training != 0: batch statistics are computed here (and the running estimates
 * updated) and written to mean / invstd, argmax (groups, c) bytes records the winning row of each maximum
 * (k <= 255); training == 0: mean / invstd are inputs, argmax may be NULL.  The normalised (groups*k, c)
 * activation is never materialised. */
int hf_bn_relu_maxpool_fwd(long long groups, int k, int c, const float *z, const float *gamma, const float *beta,
                           int training, float eps, float momentum, float *running_mean, float *running_var,
                           float *mean, float *invstd, float *pooled, unsigned char *argmax, void *workspace,
                           size_t workspace_bytes, hf_stream_t stream);
/* backward: dpooled (groups, c) -> dz (groups*k, c), dgamma, dbeta, and optionally the column sums of dz */
int hf_bn_relu_maxpool_bwd(long long groups, int k, int c, const float *z, const float *dpooled,
                           const unsigned char *argmax, const float *gamma, const float *beta, const float *save_mean,
                           const float *save_invstd, float *dz, float *dgamma, float *dbeta, float *dz_colsum,
                           void *workspace, size_t workspace_bytes, hf_stream_t stream);

/* Weight gradient of tf_util.conv2d([1,1]) (tf_util.py:180-203) on channel-last rows: grad_weight (cout, cin) =
 * grad_z^T x, fp32 MFMA, reduction over rows split into chunks whose partial tiles are summed in a fixed order.
 * in_gamma != NULL: x is the previous layer's pre-BN output and relu(bn(x)) is applied while it is staged
 * (in_gamma/in_beta/in_mean/in_invstd are that layer's (cin,) parameters and batch statistics). */
size_t hf_linear_wgrad_workspace(long long rows, int cout, int cin);
int hf_linear_wgrad(long long rows, int cout, int cin, const float *grad_z, const float *x, const float *in_gamma,
                    const float *in_beta, const float *in_mean, const float *in_invstd, float *grad_weight,
                    void *workspace, size_t workspace_bytes, hf_stream_t stream);

/* tf_util.conv2d([1,1], bn=True) up to the batch statistics, one pass (tf_util.py:180-203, 554-581):
 * z (rows, cout) = act(x) weight^T + bias on the fp32 MFMA, and the BatchNorm statistics of z (mean, invstd; running
 * estimates updated) from the accumulators.  in_gamma != NULL: x is the PREVIOUS layer's pre-BN output and
 * act = relu(bn(.)) with that layer's (cin,) parameters / statistics is applied while x is staged, so the normalised
 * activation between two layers is not read back; in_gamma == NULL: x is used as is.  x_act (rows, cin), optional
 * with in_gamma: the activated input is also stored (the weight gradient of this layer needs it in training).
 * cout <= 256, cin <= 1024. */
size_t hf_linear_bn_fwd_workspace(int cout);
int hf_linear_bn_fwd(long long rows, int cin, int cout, const float *x, const float *in_gamma, const float *in_beta,
                     const float *in_mean, const float *in_invstd, float *x_act, const float *weight, const float *bias,
                     float *z,
                     float eps, float momentum, float *running_mean, float *running_var, float *mean, float *invstd,
                     void *workspace, size_t workspace_bytes, hf_stream_t stream);

/* The same pass for PointCNN's layer order, pf.dense = linear (no bias) -> ELU -> BatchNorm (pointfly.py:480-497): z is the
 * pre-activation output, the statistics are those of elu(z), and with in_gamma != NULL x is the previous dense layer's
 * pre-activation output with a (elu(x) - mean) + beta applied while it is staged (x_act: stored as well). */
int hf_linear_elu_bn_fwd(long long rows, int cin, int cout, const float *x, const float *in_gamma, const float *in_beta,
                         const float *in_mean, const float *in_invstd, float *x_act, const float *weight, float *z, float eps,
                         float momentum, float *running_mean, float *running_var, float *mean, float *invstd, void *workspace,
                         size_t workspace_bytes, hf_stream_t stream);
/* ... and its input gradient: dx (rows, cin) = dz (rows, cout) W with the BatchNorm-backward sums of the dense layer BELOW
 * (p_dgamma = sum dx * xhat, p_dbeta = sum dx, xhat from elu(z_prev)) taken from the accumulators; workspace as
 * hf_linear_bn_bwd_workspace(cin). */
int hf_linear_elu_bn_bwd(long long rows, int cout, int cin, const float *dz, const float *weight_t, float *dx, const float *z_prev,
                         const float *p_gamma, const float *p_beta, const float *p_mean, const float *p_invstd, float *p_dgamma,
                         float *p_dbeta, void *workspace, size_t workspace_bytes, hf_stream_t stream);

/* The lifting chain of an X-Conv (pointcnn.py:96-99), two pf.dense layers on the local coordinates: x3 (rows, 3) ->
 * y0 = BN0(elu(x3 W0^T)) (c0 channels) -> z1 = y0 W1^T (c1 channels; the caller normalises elu(z1) with mean1 / invstd1).
 * The first layer's output is never stored: its statistics come from one pass over x3, and the second GEMM rebuilds y0 from x3
 * while it stages its operand.  w0 (c0, 3), w1 (c1, c0); batch statistics and running estimates of both layers are produced.
 * hf_lift_elu_bn_bwd: from dz1 (rows, c1) = the gradient w.r.t. z1: grad_w1 (c1, c0), dgamma0 / dbeta0 (c0), and grad_w0_t (3, c0) =
 * the TRANSPOSE of the first layer's weight gradient; dy0 = dz1 W1 (w1_t = W1^T as (c0, c1)) is rebuilt in the accumulators of
 * two passes and never written.  c0 <= 160 (a multiple of 4), c1 <= 256. */
size_t hf_lift_elu_bn_fwd_workspace(int c0, int c1);
int hf_lift_elu_bn_fwd(long long rows, int c0, int c1, const float *x3, const float *w0, const float *gamma0, const float *beta0,
                       float eps0, float momentum0, float *running_mean0, float *running_var0, float *mean0, float *invstd0,
                       const float *w1, float *z1, float eps1, float momentum1, float *running_mean1, float *running_var1,
                       float *mean1, float *invstd1, void *workspace, size_t workspace_bytes, hf_stream_t stream);
/* inference form: the first layer normalised with the GIVEN mean0 / invstd0 (its running estimates), z1 only; workspace as
 * hf_lift_elu_bn_fwd_workspace */
int hf_lift_elu_fwd_eval(long long rows, int c0, int c1, const float *x3, const float *w0, const float *gamma0, const float *beta0,
                         const float *mean0, const float *invstd0, const float *w1, float *z1, void *workspace, size_t workspace_bytes,
                         hf_stream_t stream);
/* the same with the SECOND layer's inference constants given too: y1 = gamma1 * invstd1 * (elu(z1) - mean1) + beta1 leaves the GEMM's
 * epilogue, the normalisation pass of that layer does not run (two-stage inference: the lifting layers of the RoI clouds) */
int hf_lift_elu_fwd_eval_bn(long long rows, int c0, int c1, const float *x3, const float *w0, const float *gamma0, const float *beta0,
                            const float *mean0, const float *invstd0, const float *w1, const float *gamma1, const float *beta1,
                            const float *mean1, const float *invstd1, float *y1, void *workspace, size_t workspace_bytes,
                            hf_stream_t stream);
size_t hf_lift_elu_bn_bwd_workspace(long long rows, int c0, int c1);
int hf_lift_elu_bn_bwd(long long rows, int c0, int c1, const float *x3, const float *w0, const float *gamma0, const float *beta0,
                       const float *mean0, const float *invstd0, const float *dz1, const float *w1_t, float *grad_w0_t,
                       float *grad_w1, float *dgamma0, float *dbeta0, void *workspace, size_t workspace_bytes, hf_stream_t stream);

/* BatchNorm (training mode) with the DROPOUT that follows it fused in -- pointfly's dense -> dropout of the PointCNN fc layers and of
 * the RPN box head (hf/core/feature_extractors/pointcnn.py:371-384, hf/core/models/rpn_model.py:556-568: tf.layers.dropout keeps an
 * element with probability 1 - rate and scales it by 1 / (1 - rate)).  y = dropout(bn(act(x))).  The keep decision of an element
 * is a counter hash of (seed of the call, element index), evaluated in the apply pass and again in the two backward passes: no
 * mask tensor, no pass of its own in either direction.  drop_state (device, two 64-bit words: base seed, forward calls so far) belongs
 * to the layer and is advanced on the device (so a captured step draws new masks at every replay); seed_out (device, one word)
 * receives this call's seed and is what hf_bn_dropout_bwd takes.  salt: mixed into the seed (the caller's rank: ranks share the
 * base seed after the parameter broadcast).  0 <= rate < 1, resolved to 1 / 65536.  Other arguments as hf_bn_relu_fwd_train / _bwd. */
int hf_bn_dropout_fwd_train(long long rows, int c, const float *x, const float *gamma, const float *beta, float eps, float momentum,
                            float *running_mean, float *running_var, int relu, float rate, unsigned long long salt,
                            unsigned long long *drop_state, unsigned long long *seed_out, float *y, float *save_mean,
                            float *save_invstd, void *workspace, size_t workspace_bytes, hf_stream_t stream);
int hf_bn_dropout_bwd(long long rows, int c, const float *x, const float *dy, const float *gamma, const float *beta,
                      const float *save_mean, const float *save_invstd, int relu, float rate, const unsigned long long *seed,
                      float *dx, float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes, hf_stream_t stream);

/* Input gradient of a Linear with a handful of outputs (the segmentation head of the RPN, hf/core/models/rpn_model.py: dense to
 * classes + 1 logits): dx (rows, cin) = g (rows, cout) w (cout, cin), cout <= 4, one streaming pass. */
int hf_narrow_linear_dx(long long rows, int cin, int cout, const float *g, const float *w, float *dx, hf_stream_t stream);

/* The second half of hf_bn_relu_bwd alone: dx from dy, x and ALREADY KNOWN dgamma / dbeta (no reduction pass). */
int hf_bn_relu_bwd_dx(long long rows, int c, const float *x, const float *dy, const float *gamma, const float *beta,
                      const float *save_mean, const float *save_invstd, const float *dgamma, const float *dbeta, int relu,
                      float *dx, hf_stream_t stream);
/* Input gradient of tf_util.conv2d([1,1], bn=True) on the fp32 MFMA: dx (rows, cin) = dz (rows, cout) W, weight_t = W^T
 * as (cin, cout).  z != NULL: the first operand is dy, the gradient w.r.t. this layer's activation, and dz is rebuilt
 * from (dy, z, gamma, beta, mean, invstd, dgamma, dbeta) while it is staged (and stored to dz_out if given); z == NULL:
 * the first operand is dz itself.  z_prev != NULL: dgamma / dbeta of the layer below (p_*: its parameters and
 * statistics, z_prev (rows, cin) its pre-BN output) are reduced from the accumulators into p_dgamma / p_dbeta.
 * dx may be NULL when only dz_out / the sums are wanted.  cin <= 256, cout <= 256 with z. */
size_t hf_linear_bn_bwd_workspace(int cin);
int hf_linear_bn_bwd(long long rows, int cout, int cin, const float *dy_or_dz, const float *z, const float *gamma,
                     const float *beta, const float *mean, const float *invstd, const float *dgamma, const float *dbeta,
                     float *dz_out, const float *weight_t, float *dx, const float *z_prev, const float *p_gamma,
                     const float *p_beta, const float *p_mean, const float *p_invstd, float *p_dgamma, float *p_dbeta,
                     void *workspace, size_t workspace_bytes, hf_stream_t stream);

/* ---- glue either side of the ops ---- */

/* The LiDAR -> image fusion step in one pass: hf/core/projection.py:5-32 (tf_rect_to_image: homogeneous point times
 * the 3x4 P2 matrix, divide by depth) + hf/core/models/rpn_model.py:227-235 (tf.cast to int32, tf.gather_nd of the
 * image feature map at [b, v, u]).  pts (b,p,3), calib (b,3,4), img (b,h,w,c) -> out (b,p,c); a pixel outside the
 * image gives zeros (tf.gather_nd on GPU).  pix (b,p,2) [u,v], optional, is what the gradient needs.
 * hf_project_gather_grad: grad_img (b,h,w,c), zero-filled here, += grad_out rows at their pixels. */
int hf_project_gather(int b, int p, int h, int w, int c, const float *pts, const float *calib, const float *img, float *out,
                      int *pix, hf_stream_t stream);
int hf_project_gather_grad(int b, int p, int h, int w, int c, const float *grad_out, const int *pix, float *grad_img,
                           hf_stream_t stream);
/* The 'concat' fusion of the RPN / RCNN with its path drop (hf/core/models/rpn_model.py:515-546, rcnn_model.py:560-590):
 * out (rows, c1+c2) = [a * masks[0] | b * masks[1]]; masks = two floats ON THE DEVICE (this step's path-drop decision,
 * create_path_drop_masks, rpn_model.py:1130-1193) or NULL for (1, 1).  _grad: grad_a = grad_out[:, :c1] * masks[0],
 * grad_b = grad_out[:, c1:] * masks[1]; either may be NULL. */
int hf_fuse_concat(long long rows, int c1, int c2, const float *a, const float *b, const float *masks, float *out,
                   hf_stream_t stream);
int hf_fuse_concat_grad(long long rows, int c1, int c2, const float *grad_out, const float *masks, float *grad_a,
                        float *grad_b, hf_stream_t stream);
/* The RPN loss (hf/core/models/rpn_model.py:1040-1128 with hf/core/losses.py:131-226) in two passes.  rows = B*P points;
 * seg_logits (rows, k+1); head (rows, k, D), D = 4 nbx + 2 nbt + 4 laid out as _parse_rpn_output slices it (:870-935);
 * label (rows) int32: 0 background, 1..k; the targets exactly as hf_bin_box_encode writes them (bin_x / res_x / bin_z / res_z
 * (rows, k), bin_theta / res_theta / res_y (rows), res_size (rows, 3)): the labelled class's entries are picked here
 * (:733-776).  hf_rpn_loss_fwd -> out5 = [segmentation, bin classification, regression, #foreground, total loss]
 * (focal alpha 0.25 gamma 2 on the clipped softmax, x seg_weight / rows; softmax cross-entropy of the three bin groups and
 * smooth-L1 of the true bin's residuals, y and the sizes over foreground points, / max(#fg, 1)).  hf_rpn_loss_bwd -> the
 * gradients w.r.t. seg_logits and head times *upstream (a device scalar); grad_head is zero-filled here.
 * workspace: hf_rpn_loss_workspace() bytes. */
size_t hf_rpn_loss_workspace(void);
int hf_rpn_loss_fwd(long long rows, int k, int nbx, int nbt, const float *seg_logits, const float *head, const int *label,
                    const int *bin_x, const float *res_x, const int *bin_z, const float *res_z, const int *bin_theta,
                    const float *res_theta, const float *res_y, const float *res_size, float seg_weight, float cls_weight,
                    float reg_weight, float *out5, void *workspace, size_t workspace_bytes, hf_stream_t stream);
int hf_rpn_loss_bwd(long long rows, int k, int nbx, int nbt, const float *seg_logits, const float *head, const int *label,
                    const int *bin_x, const float *res_x, const int *bin_z, const float *res_z, const int *bin_theta,
                    const float *res_theta, const float *res_y, const float *res_size, float seg_weight, float cls_weight,
                    float reg_weight, const float *out5, const float *upstream, float *grad_seg, float *grad_head,
                    hf_stream_t stream);
/* hf/core/bin_based_box3d_encoder.py:9-139 (tf_decode) for `rows` reference points x k classes: rows = B*p in the RPN
 * (ref_theta NULL = the constant 0), the RoI count in the RCNN.  Per (row, class) inputs are (rows, k[, 3]) arrays;
 * ss / deltas (k,) the per-class XZ search range and bin length; boxes (rows, k, 7) = [x, y, z, l, w, h, ry]. */
int hf_bin_box_decode(long long rows, int k, const float *ref_pts, const float *ref_theta, const int *bin_x,
                      const float *res_x_norm, const int *bin_z, const float *res_z_norm, const int *bin_theta,
                      const float *res_theta_norm, const float *res_y, const float *res_size_norm, const float *mean_sizes,
                      const float *ss, const float *deltas, float r, float delta_theta, float *boxes, hf_stream_t stream);
/* bin_based_box3d_encoder.py:142-269 (tf_encode); rcnn = 0 / 1 selects the rank-3 (RPN) / rank-2 (RCNN) orientation
 * rule.  hi_xz (k,) = float32(2*Ss - 1e-3), hi_theta = float32(2*R - 1e-3), half_delta_theta = float32(0.5*DELTA_THETA):
 * constants the reference forms in Python doubles before TensorFlow casts them. */
int hf_bin_box_encode(long long rows, int k, int rcnn, const float *ref_pts, const float *ref_theta, const float *boxes,
                      const float *mean_sizes, const float *ss, const float *deltas, const float *hi_xz, float r,
                      float hi_theta, float delta_theta, float half_delta_theta, int *bin_x, float *res_x_norm, int *bin_z,
                      float *res_z_norm, int *bin_theta, float *res_theta_norm, float *res_y, float *res_size_norm,
                      hf_stream_t stream);

/* The decoding block of the RPN / RCNN heads in one pass (hf/core/models/rpn_model.py:870-935 parse, :593-605 argmax,
 * :248-290 residual gather, :609-639 mean sizes + tf_decode, :237-246 class gather).  head (rows, k, d) with
 * d = 2*nbx + 2*nbz + 2*nbt + 4 laid out [bin_x logits | res_x_norms | bin_z logits | res_z_norms | bin_theta logits |
 * res_theta_norms | res_y | res_size_norm(3)]; mean_sizes_k (k, 3).  cls == NULL: boxes (rows, k, 7); cls (rows,) int32:
 * boxes (rows, 7), the decoded box of each row's class (rows whose cls is outside [0, k) are left untouched). */
int hf_bin_head_decode(long long rows, int k, int nbx, int nbz, int nbt, const float *head, const float *ref_pts,
                       const float *ref_theta, const float *mean_sizes_k, const float *ss, const float *deltas, float r,
                       float delta_theta, const int *cls, float *boxes, hf_stream_t stream);

/* ------------------------------------------------------------------- PointCNN X-Conv products (callers)
 * hf_xconv_apply       replaces tf.matmul(X, nn_fts_input)  hf/core/feature_extractors/pointcnn.py:133
 *                      x (rows,k,k), f (rows,k,c) -> out (rows,k,c): out[r][i][:] = sum_j x[r][i][j] * f[r][j][:]; k in {4, 8}
 * hf_xconv_apply_grad  its gradients: grad_x (rows,k,k) and / or grad_f (rows,k,c) (either may be NULL)
 * hf_depthwise_k       replaces pf.depthwise_conv2d(.., (1,K)) and the depthwise half of pf.separable_conv2d(.., (1,K))
 *                      on a width-K input (pointfly.py:437-457; pointcnn.py:104-131): x (rows,k,c), w (k,c,m) in
 *                      TensorFlow's depthwise filter layout (1,K,C,M) -> y (rows, c*m), y[r][ch*m+mm] = sum_w x[r][w][ch]*w[w][ch][mm]
 * hf_depthwise_k_grad  grad_x (rows,k,c) and / or grad_w (k,c,m) (zero-filled here, accumulated with atomics)
 * (k, m) supported: k = 8 with m in {1,2,3,4,8}, k = 4 with m in {1,4}; anything else: HF_EINVAL. */
int hf_xconv_apply(long long rows, int k, int c, const float *x, const float *f, float *out, hf_stream_t stream);
int hf_xconv_apply_grad(long long rows, int k, int c, const float *x, const float *f, const float *grad_out, float *grad_x,
                        float *grad_f, hf_stream_t stream);
int hf_depthwise_k(long long rows, int k, int c, int m, const float *x, const float *w, float *y, hf_stream_t stream);
/* hf_xconv_apply followed by hf_depthwise_k in ONE pass (pointcnn.py:133-140): out (rows, c*m) from x (rows,k,k),
 * f (rows,k,c), wd (k,c,m) without the (rows,k,c) product in memory; bit-identical to the two calls.  (k, m): k = 8 with
 * m in 1..4, and (4,1), (4,4), (12,1), (12,2) (the RCNN's layers, rcnn_multiclass.config:157-186).
 * hf_xconv_depthwise_grad: grad_x (rows,k,k), grad_f (rows,k,c), grad_wd (k,c,m; zero-filled here, atomics), any may be NULL. */
int hf_xconv_depthwise(long long rows, int k, int c, int m, const float *x, const float *f, const float *wd, float *out,
                       hf_stream_t stream);
int hf_xconv_depthwise_grad(long long rows, int k, int c, int m, const float *x, const float *f, const float *wd,
                            const float *grad_out, float *grad_x, float *grad_f, float *grad_wd, hf_stream_t stream);
int hf_depthwise_k_grad(long long rows, int k, int c, int m, const float *x, const float *w, const float *grad_y,
                        float *grad_x, float *grad_w, hf_stream_t stream);
/* the same with a workspace (hf_depthwise_k_grad_workspace bytes): the row chunks' partial weight gradients are written out and
 * added in a fixed order instead of meeting in atomics on the same k*c*m addresses -- deterministic, and 4x faster for the
 * 8-channel layers of the X-transform where those atomics were the whole cost */
size_t hf_depthwise_k_grad_workspace(long long rows, int k, int c, int m);
int hf_depthwise_k_grad_ws(long long rows, int k, int c, int m, const float *x, const float *w, const float *grad_y, float *grad_x,
                           float *grad_w, void *workspace, size_t workspace_bytes, hf_stream_t stream);
/* The same pass with F_* = [F_delta | gathered features] (pointcnn.py:124-127: tf.gather_nd of the previous layer's features
 * at the neighbour indices, concatenated behind the lifted coordinates) NOT materialised: channels [0, c0) are read from
 * f_delta (rows, k, c0), channels [c0, c0+c1) from the feature table fts (b, n_src, c1) through the neighbour table
 * idx (b, rows_per_cloud, k) (indices into the row's own cloud, as hf_group_point takes them); rows = b * rows_per_cloud.
 * c0 must be a multiple of 64 (it is C/4 of the previous layer in every
 * shipped configuration).  Results are bit-identical to hf_group_point_into + hf_xconv_depthwise on the concatenation.
 * hf_xconv_depthwise_gather_grad: grad_x (rows,k,k), grad_f_delta (rows,k,c0), grad_wd (k,c0+c1,m),
 * grad_fts (b*n_src, c1) -- the last one in gather form through the CSR inverse of the neighbour table (offsets (b, n_src+1),
 * entries (b, rows_per_cloud*k) from hf_index_inverse): every table row is written once, summed in ascending (row, slot) order
 * like hf_group_point_grad_gather; no atomics, no zero fill.  Any gradient may be NULL.  workspace
 * (hf_xconv_depthwise_gather_grad_workspace bytes, may be NULL): with it the gathered block's gradient is written once and summed
 * per table row (faster; without it it is rebuilt per table row from grad_out: no extra memory, same values), and the row
 * chunks' partial depthwise-weight gradients are added in a fixed order instead of meeting in atomics (deterministic grad_wd). */
size_t hf_xconv_depthwise_gather_grad_workspace(int b, int rows_per_cloud, int k, int c0, int c1, int m);
int hf_xconv_depthwise_gather(int b, int n_src, int rows_per_cloud, int k, int c0, int c1, int m, const float *x,
                              const float *f_delta, const float *fts, const int *idx, const float *wd, float *out,
                              hf_stream_t stream);
int hf_xconv_depthwise_gather_grad(int b, int n_src, int rows_per_cloud, int k, int c0, int c1, int m, const float *x,
                                   const float *f_delta, const float *fts, const int *idx, const float *wd,
                                   const float *grad_out, const int *offsets, const int *entries, float *grad_x,
                                   float *grad_f_delta, float *grad_fts, float *grad_wd, void *workspace, size_t workspace_bytes,
                                   hf_stream_t stream);

/* The FIRST layer of a set-abstraction MLP on neighbourhoods read in place: sample_and_group
 * (hf/core/feature_extractors/pointnet_util.py:42-64) materialises new_points (B,M,K,C+3) = [grouped_xyz - centre | points[idx]] and
 * the first tf_util.conv2d (:156-160) reads it back; here the operand rows are assembled from (points, idx, grouped_xyz) while
 * they are staged for the MFMA tiles, so that tensor never exists.  rows = B*M*K, rows_per_cloud = M*K, points (B,n_src,c_feat)
 * (NULL when c_feat == 0), idx (rows) int32, grouped_xyz (rows,3) = query_ball_group's centred coordinates.
 * Column layout of the assembled operand, which `weight` (cout, cin) / `grad_weight` follow:
 *   [feature 0 .. c_feat-1, zeros up to cfp = round_up(c_feat, 4) | x, y, z, 0]      cin = cfp + 4
 * (the caller permutes / pads the layer's weight once per step: a few KB).  Otherwise as hf_linear_bn_fwd / hf_linear_wgrad:
 * z (rows,cout) = operand weight^T + bias with the batch statistics of z from the accumulators; grad_weight = grad_z^T operand. */
int hf_linear_bn_fwd_gather(long long rows, int c_feat, int cout, const float *points, int n_src, long long rows_per_cloud,
                            const int *idx, const float *grouped_xyz, const float *weight, const float *bias, float *z, float eps,
                            float momentum, float *running_mean, float *running_var, float *mean, float *invstd, void *workspace,
                            size_t workspace_bytes, hf_stream_t stream);
int hf_linear_wgrad_gather(long long rows, int cout, int c_feat, const float *grad_z, const float *points, int n_src,
                           long long rows_per_cloud, const int *idx, const float *grouped_xyz, float *grad_weight,
                           void *workspace, size_t workspace_bytes, hf_stream_t stream);

/* ------------------------------------------------------------------ the optimizer step of the train step */

/* tf.train.AdamOptimizer.apply_gradients over EVERY parameter tensor in one launch (hf/core/trainer.py:71,
 * hf/builders/optimizer_builder.py:59-64; the framework form is one multi-tensor launch per 4 KB of kernel arguments).
 * table: one entry per tensor, ON THE DEVICE; chunk_map: num_chunks pairs (tensor, chunk) of hf_adam_chunk() elements each,
 * on the device; step: one float on the device = the number of THIS step (1 for the first; the caller advances it);
 * grad_scale multiplies every gradient on load (1 / world after a summing all-reduce).
 * mode 0: p -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)   (TensorFlow's placement of epsilon)
 * mode 1: p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)  (torch.optim.Adam's) */
typedef struct hf_adam_entry {
    float *param;
    const float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    long long numel;
} hf_adam_entry;
int hf_adam_chunk(void);
int hf_adam_multi(int num_chunks, const hf_adam_entry *table, const int *chunk_map, const float *step, float lr, float beta1,
                  float beta2, float eps, float grad_scale, int mode, hf_stream_t stream);

/* ------------------------------------------------------------------ feeding a captured step */

/* n (<= hf_copy_multi_max()) device-to-device copies in ONE launch: dst[i] <- src[i], bytes[i] each (the three arrays live on the
 * HOST; they travel in the kernel arguments).  The feed_dict of one sess.run (hf/core/trainer.py:216-226) becomes, for a captured
 * step, the refresh of its static input slots: ~40 tensors per step (points, labels, the neighbour tables of every level) that
 * the framework would copy with one launch each.  Overlapping ranges are not supported. */
int hf_copy_multi_max(void);
int hf_copy_multi(int n, void *const *dst, const void *const *src, const long long *bytes, hf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HFOPS_H */
