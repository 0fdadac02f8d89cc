// ref_gpu_driver.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref/libhfref_gpu.so).
//
// extern "C" entry points around the reference's own launcher functions, which are
// compiled unmodified from /root/reference/*/*.cu by oracle/Makefile.  The prototypes
// below are the declarations the reference's op wrappers use:
//   sampling/tf_sampling.cpp:94,125,150   grouping/tf_grouping.cpp:66,142,173
//   bev_iou/bev_iou.cpp:40,144            cropping/tf_cropping.cpp:100,192
// All pointers are device pointers.  The reference launches on the legacy default
// stream; each wrapper brackets the launch with a device synchronise so callers on any
// stream see completed results.
#include <hip/hip_runtime.h>

void farthestpointsamplingLauncher(int b, int n, int m, const float *inp, float *temp, int *out);
void gatherpointLauncher(int b, int n, int m, const float *inp, const int *idx, float *out);
void scatteraddpointLauncher(int b, int n, int m, const float *out_g, const int *idx, float *inp_g);
void queryBallPointLauncher(int b, int n, int m, float radius, int nsample, const float *xyz1,
                            const float *xyz2, int *idx, int *pts_cnt);
void groupPointLauncher(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out);
void groupPointGradLauncher(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                            float *grad_points);
void compute_bev_iou_gpu(const int num_a, const float *boxes_a, const int num_b, const float *boxes_b,
                         float *ans_overlap, float *ans_iou);
void oriented_nms_gpu(const float *boxes, unsigned long long *mask, int boxes_num, float nms_overlap_thresh);
void pccropandsample_gpu(const float *pts, const float *fts, const float *intensities, const bool *mask,
                         const float *boxes, const int *box_ind, int num_boxes, int batch, int npts, int resize,
                         int channel, int intensity_channel, float *crop_pts, float *crop_fts,
                         float *crop_intensities, bool *crop_mask, int *crop_ind, bool *non_empty_box);

#define HFREF_SYNC_CALL(call)                                  \
    do {                                                       \
        if (hipDeviceSynchronize() != hipSuccess) return -1;   \
        call;                                                  \
        if (hipGetLastError() != hipSuccess) return -2;        \
        if (hipDeviceSynchronize() != hipSuccess) return -3;   \
        return 0;                                              \
    } while (0)

extern "C" {

int hfref_farthest_point_sample(int b, int n, int m, const float *inp, float *temp, int *out)
{
    HFREF_SYNC_CALL(farthestpointsamplingLauncher(b, n, m, inp, temp, out));
}
int hfref_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out)
{
    HFREF_SYNC_CALL(gatherpointLauncher(b, n, m, inp, idx, out));
}
int hfref_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx, float *inp_g)
{
    HFREF_SYNC_CALL(scatteraddpointLauncher(b, n, m, out_g, idx, inp_g));
}
int hfref_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2,
                           int *idx, int *pts_cnt)
{
    HFREF_SYNC_CALL(queryBallPointLauncher(b, n, m, radius, nsample, xyz1, xyz2, idx, pts_cnt));
}
int hfref_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out)
{
    HFREF_SYNC_CALL(groupPointLauncher(b, n, c, m, nsample, points, idx, out));
}
int hfref_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                           float *grad_points)
{
    HFREF_SYNC_CALL(groupPointGradLauncher(b, n, c, m, nsample, grad_out, idx, grad_points));
}
int hfref_compute_bev_iou(int num_a, const float *boxes_a, int num_b, const float *boxes_b, float *ans_overlap,
                          float *ans_iou)
{
    HFREF_SYNC_CALL(compute_bev_iou_gpu(num_a, boxes_a, num_b, boxes_b, ans_overlap, ans_iou));
}
int hfref_nms_mask(const float *boxes, unsigned long long *mask, int n, float thresh)
{
    HFREF_SYNC_CALL(oriented_nms_gpu(boxes, mask, n, thresh));
}
int hfref_pc_crop_and_sample(const float *pts, const float *fts, const float *intensities, const bool *mask,
                             const float *boxes, const int *box_ind, int num_boxes, int batch, int npts,
                             int resize, int channel, int intensity_channel, float *crop_pts, float *crop_fts,
                             float *crop_intensities, bool *crop_mask, int *crop_ind, bool *non_empty_box)
{
    HFREF_SYNC_CALL(pccropandsample_gpu(pts, fts, intensities, mask, boxes, box_ind, num_boxes, batch, npts,
                                        resize, channel, intensity_channel, crop_pts, crop_fts, crop_intensities,
                                        crop_mask, crop_ind, non_empty_box));
}

}  // extern "C"
