// bev_iou.hip -- rotated-BEV overlap / IoU matrix and oriented NMS for gfx950.
//
// Replaces bev_iou/bev_iou_g.cu:8-298 and the host half of OrientedNMSOp::Compute
// (bev_iou/bev_iou.cpp:60-116: cudaMalloc, blocking D2H of the mask, host greedy sweep, H2D).
//
// Structure (both the IoU matrix and the NMS mask kernel):
//   phase 1  every pair gets a trig-free bounding-circle test.  Far pairs are exactly 0 in the
//            reference too (no edge crossing, no corner inside: bev_iou_g.cu:150-176 leave
//            cnt = 0 -> area 0), so they are written as zeros straight away;
//   phase 2  the few surviving pairs are compacted into an LDS queue and the expensive polygon
//            clip (16 edge tests, atan2 sort, shoelace) runs on dense lanes instead of on
//            1-2 live lanes per wave.
// The clip itself follows the reference operation by operation (fp32, no contraction); cos/sin
// are evaluated once per box per pair-side (cos(-a) == cos(a), sin(-a) == -sin(a) exactly).
#include <math.h>

#include "hf_common.h"

namespace hf {

struct Pt { float x, y; };

constexpr float kIouEps = 1e-8f;  // bev_iou_g.cu:7

// bev_iou_g.cu:33-35
__device__ __forceinline__ float cross3(Pt p1, Pt p2, Pt p0)
{
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

// bev_iou_g.cu:62-91 (rect pre-check :37-43 inlined)
__device__ __forceinline__ bool seg_intersection(Pt p1, Pt p0, Pt q1, Pt q0, Pt &ans)
{
    const bool rect = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
                      fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
    if (!rect) return false;
    const float s1 = cross3(q0, p1, p0);
    const float s2 = cross3(p1, q1, p0);
    const float s3 = cross3(p0, q1, q0);
    const float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > kIouEps) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

// check_in_box2d, bev_iou_g.cu:45-60, with cos(-a)=ac, sin(-a)=-as passed in
__device__ __forceinline__ bool in_box2d(const float *box, float ac, float as_neg, Pt p)
{
    const float MARGIN = 1e-5f;
    const float cx = (box[0] + box[2]) / 2, cy = (box[1] + box[3]) / 2;
    const float rx = (p.x - cx) * ac + (p.y - cy) * as_neg + cx;
    const float ry = -(p.x - cx) * as_neg + (p.y - cy) * ac + cy;
    return rx > box[0] - MARGIN && rx < box[2] + MARGIN && ry > box[1] - MARGIN && ry < box[3] + MARGIN;
}

// rotate_around_center, bev_iou_g.cu:92-96
__device__ __forceinline__ Pt rot_center(Pt c, float ac, float as, Pt p)
{
    Pt r;
    r.x = (p.x - c.x) * ac + (p.y - c.y) * as + c.x;
    r.y = -(p.x - c.x) * as + (p.y - c.y) * ac + c.y;
    return r;
}

// box_overlap, bev_iou_g.cu:102-206
__device__ float box_overlap(const float *a, const float *b)
{
    const float a_x1 = a[0], a_y1 = a[1], a_x2 = a[2], a_y2 = a[3], a_angle = a[4];
    const float b_x1 = b[0], b_y1 = b[1], b_x2 = b[2], b_y2 = b[3], b_angle = b[4];
    const Pt ca = { (a_x1 + a_x2) / 2, (a_y1 + a_y2) / 2 };
    const Pt cb = { (b_x1 + b_x2) / 2, (b_y1 + b_y2) / 2 };
    const float acs = cosf(a_angle), asn = sinf(a_angle);
    const float bcs = cosf(b_angle), bsn = sinf(b_angle);
    Pt A[5], B[5];
    A[0] = rot_center(ca, acs, asn, Pt{ a_x1, a_y1 });
    A[1] = rot_center(ca, acs, asn, Pt{ a_x2, a_y1 });
    A[2] = rot_center(ca, acs, asn, Pt{ a_x2, a_y2 });
    A[3] = rot_center(ca, acs, asn, Pt{ a_x1, a_y2 });
    A[4] = A[0];
    B[0] = rot_center(cb, bcs, bsn, Pt{ b_x1, b_y1 });
    B[1] = rot_center(cb, bcs, bsn, Pt{ b_x2, b_y1 });
    B[2] = rot_center(cb, bcs, bsn, Pt{ b_x2, b_y2 });
    B[3] = rot_center(cb, bcs, bsn, Pt{ b_x1, b_y2 });
    B[4] = B[0];

    Pt cp[24];
    float ang[24];
    Pt ctr = { 0.f, 0.f };
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Pt x;
            if (seg_intersection(A[i + 1], A[i], B[j + 1], B[j], x)) {
                ctr.x = ctr.x + x.x; ctr.y = ctr.y + x.y;
                cp[cnt++] = x;
            }
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (in_box2d(a, acs, -asn, B[k])) {
            ctr.x = ctr.x + B[k].x; ctr.y = ctr.y + B[k].y;
            cp[cnt++] = B[k];
        }
        if (in_box2d(b, bcs, -bsn, A[k])) {
            ctr.x = ctr.x + A[k].x; ctr.y = ctr.y + A[k].y;
            cp[cnt++] = A[k];
        }
    }
    if (cnt < 3) return 0.0f;  // fewer than 3 points: the fan below sums nothing but zeros
    ctr.x /= cnt;
    ctr.y /= cnt;
    // the reference re-evaluates atan2 inside every comparison (point_cmp :98-100); it is a pure
    // function of the point, so evaluate once per point and carry it through the swaps.
    for (int k = 0; k < cnt; ++k) ang[k] = atan2f(cp[k].y - ctr.y, cp[k].x - ctr.x);
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (ang[i] > ang[i + 1]) {
                const Pt tp = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = tp;
                const float ta = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = ta;
            }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k) {
        const float ux = cp[k].x - cp[0].x, uy = cp[k].y - cp[0].y;
        const float vx = cp[k + 1].x - cp[0].x, vy = cp[k + 1].y - cp[0].y;
        area += ux * vy - uy * vx;
    }
    return fabsf(area) / 2.0f;
}

// iou_bev(box_a, box_b, s_overlap), bev_iou_g.cu:208-215
__device__ __forceinline__ float iou_from_overlap(const float *a, const float *b, float s)
{
    const float sa = (a[2] - a[0]) * (a[3] - a[1]);
    const float sb = (b[2] - b[0]) * (b[3] - b[1]);
    return s / fmaxf(sa + sb - s, kIouEps);
}

// Conservative "cannot touch" test, no trig: centres further apart than the sum of the boxes'
// half-perimeter bounds (>= circumradius) plus a slack that dwarfs MARGIN=1e-5 and fp32 rounding
// at these magnitudes.  true => the reference computes exactly 0 overlap / 0 IoU.
__device__ __forceinline__ bool surely_disjoint(const float *a, const float *b)
{
    const float cax = (a[0] + a[2]) * 0.5f, cay = (a[1] + a[3]) * 0.5f;
    const float cbx = (b[0] + b[2]) * 0.5f, cby = (b[1] + b[3]) * 0.5f;
    const float ra = (fabsf(a[2] - a[0]) + fabsf(a[3] - a[1])) * 0.5f;
    const float rb = (fabsf(b[2] - b[0]) + fabsf(b[3] - b[1])) * 0.5f;
    const float mag = fabsf(cax) + fabsf(cay) + fabsf(cbx) + fabsf(cby) + ra + rb;
    const float reach = ra + rb + 1e-3f + 1e-5f * mag;
    const float dx = cax - cbx, dy = cay - cby;
    return dx * dx + dy * dy > reach * reach;  // NaN/inf inputs compare false -> full path
}

// ---------------------------------------------------------------- IoU matrix
constexpr int kIouThreads = 256;
constexpr int kIouPerThread = 8;
constexpr int kIouChunk = kIouThreads * kIouPerThread;  // pairs per block

__global__ __launch_bounds__(kIouThreads) void bev_iou_kernel(int num_a, const float *__restrict__ boxes_a, int num_b,
                                                              const float *__restrict__ boxes_b,
                                                              float *__restrict__ ans_overlap,
                                                              float *__restrict__ ans_iou)
{
    __shared__ int queue[kIouChunk];
    __shared__ int qcount;
    const long long total = static_cast<long long>(num_a) * num_b;
    const long long base = static_cast<long long>(blockIdx.x) * kIouChunk;
    if (threadIdx.x == 0) qcount = 0;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kIouPerThread; ++it) {
        const int local = it * kIouThreads + threadIdx.x;
        const long long e = base + local;
        if (e < total) {
            const int ia = static_cast<int>(e / num_b), ib = static_cast<int>(e - static_cast<long long>(ia) * num_b);
            if (surely_disjoint(boxes_a + static_cast<size_t>(ia) * 5, boxes_b + static_cast<size_t>(ib) * 5)) {
                if (ans_overlap) ans_overlap[e] = 0.0f;
                if (ans_iou) ans_iou[e] = 0.0f;
            } else {
                queue[atomicAdd(&qcount, 1)] = local;
            }
        }
    }
    __syncthreads();
    const int nq = qcount;
    for (int q = threadIdx.x; q < nq; q += kIouThreads) {
        const long long e = base + queue[q];
        const int ia = static_cast<int>(e / num_b), ib = static_cast<int>(e - static_cast<long long>(ia) * num_b);
        float a[5], b[5];
#pragma unroll
        for (int d = 0; d < 5; ++d) { a[d] = boxes_a[static_cast<size_t>(ia) * 5 + d]; b[d] = boxes_b[static_cast<size_t>(ib) * 5 + d]; }
        const float s = box_overlap(a, b);
        if (ans_overlap) ans_overlap[e] = s;
        if (ans_iou) ans_iou[e] = iou_from_overlap(a, b, s);
    }
}

// ---------------------------------------------------------------- NMS mask
// one block per (row tile, col tile) of 64x64 pairs; mask word (row, col tile) bit j set iff
// iou(row, col*64+j) > thresh, j > row inside the diagonal tile (bev_iou_g.cu:256-298).
constexpr int kNmsThreads = 256;

template <bool UPPER_ONLY>
__global__ __launch_bounds__(kNmsThreads) void nms_mask_kernel(int n, float thresh, const float *__restrict__ boxes,
                                                               unsigned long long *__restrict__ mask)
{
    const int row_t = blockIdx.y, col_t = blockIdx.x;
    if (UPPER_ONLY && col_t < row_t) return;  // never read by the sweep (bev_iou.cpp:100-103 starts at nblock)
    __shared__ float rb[64 * 5], cbx[64 * 5];
    __shared__ unsigned long long words[64];
    __shared__ int queue[64 * 64];
    __shared__ int qcount;
    const int t = threadIdx.x;
    const int row_size = min(n - row_t * 64, 64), col_size = min(n - col_t * 64, 64);
    for (int e = t; e < row_size * 5; e += kNmsThreads) rb[e] = boxes[static_cast<size_t>(row_t) * 64 * 5 + e];
    for (int e = t; e < col_size * 5; e += kNmsThreads) cbx[e] = boxes[static_cast<size_t>(col_t) * 64 * 5 + e];
    if (t < 64) words[t] = 0ull;
    if (t == 0) qcount = 0;
    __syncthreads();
    for (int e = t; e < 64 * 64; e += kNmsThreads) {
        const int r = e >> 6, c = e & 63;
        const bool valid = r < row_size && c < col_size && !(row_t == col_t && c <= r);
        if (valid && !surely_disjoint(rb + r * 5, cbx + c * 5)) queue[atomicAdd(&qcount, 1)] = e;
    }
    __syncthreads();
    const int nq = qcount;
    for (int q = t; q < nq; q += kNmsThreads) {
        const int e = queue[q];
        const int r = e >> 6, c = e & 63;
        const float s = box_overlap(rb + r * 5, cbx + c * 5);
        if (iou_from_overlap(rb + r * 5, cbx + c * 5, s) > thresh) atomicOr(&words[r], 1ull << c);
    }
    __syncthreads();
    const int col_blocks = (n + 63) / 64;
    if (t < row_size) mask[(static_cast<size_t>(row_t) * 64 + t) * col_blocks + col_t] = words[t];
}

// ---------------------------------------------------------------- greedy sweep on the device
// bev_iou.cpp:87-112.  One 256-thread block: wave 0 resolves the 64 boxes of a column block
// sequentially (their mutual suppression sits in the diagonal mask words, held one per lane),
// then all waves OR the kept rows into the running removal words of the later column blocks.
constexpr int kSweepThreads = 256;

__global__ __launch_bounds__(kSweepThreads) void nms_sweep_kernel(int n, const unsigned long long *__restrict__ mask,
                                                                  int *__restrict__ keep, int *__restrict__ num_kept)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long *remv = reinterpret_cast<unsigned long long *>(smem_raw);  // col_blocks words
    __shared__ unsigned long long kept_bits;
    __shared__ int kept_total;
    const int t = threadIdx.x;
    const int cb = (n + 63) / 64;
    for (int w = t; w < cb; w += kSweepThreads) remv[w] = 0ull;
    if (t == 0) kept_total = 0;
    __syncthreads();
    for (int blk = 0; blk < cb; ++blk) {
        if (t < 64) {
            const int i = blk * 64 + t;
            const unsigned long long diag = i < n ? mask[static_cast<size_t>(i) * cb + blk] : 0ull;
            unsigned long long word = remv[blk];
            const int lim = min(64, n - blk * 64);
            unsigned long long kb = 0ull;
            for (int l = 0; l < lim; ++l) {  // wave-uniform scalar loop
                if (!((word >> l) & 1ull)) {
                    kb |= 1ull << l;
                    const unsigned lo = __builtin_amdgcn_readlane(static_cast<int>(diag & 0xffffffffu), l);
                    const unsigned hi = __builtin_amdgcn_readlane(static_cast<int>(diag >> 32), l);
                    word |= (static_cast<unsigned long long>(hi) << 32) | lo;
                }
            }
            const int before = kept_total;
            if ((kb >> t) & 1ull) keep[before + __builtin_popcountll(kb & ((1ull << t) - 1ull))] = i;
            if (t == 0) { kept_bits = kb; kept_total = before + __builtin_popcountll(kb); }
        }
        __syncthreads();
        const unsigned long long kb = kept_bits;
        for (int w = blk + 1 + t; w < cb; w += kSweepThreads) {
            unsigned long long acc = remv[w];
            unsigned long long bits = kb;
            while (bits) {
                const int l = __builtin_ctzll(bits);
                bits &= bits - 1ull;
                acc |= mask[(static_cast<size_t>(blk) * 64 + l) * cb + w];
            }
            remv[w] = acc;
        }
        __syncthreads();
    }
    const int kept = kept_total;
    // pad with keep[0]; box 0 is always kept (nothing precedes it), bev_iou.cpp:110-112
    for (int p = kept + t; p < n; p += kSweepThreads) keep[p] = 0;
    if (t == 0 && num_kept) *num_kept = kept;
}

}  // namespace hf

using namespace hf;

HF_API int hf_compute_bev_iou(int num_a, const float *boxes_a, int num_b, const float *boxes_b, float *ans_overlap,
                              float *ans_iou, hf_stream_t stream)
{
    // ComputeBevIOUOp: N > 0, M > 0, (N,5) / (M,5)  (bev_iou.cpp:156-157)
    if (num_a <= 0 || num_b <= 0 || !boxes_a || !boxes_b) return HF_EINVAL;
    if (!ans_overlap && !ans_iou) return HF_OK;
    const long long total = static_cast<long long>(num_a) * num_b;
    const long long blocks = (total + kIouChunk - 1) / kIouChunk;
    if (blocks > 0x7fffffffLL) return HF_EINVAL;
    hipLaunchKernelGGL(bev_iou_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kIouThreads), 0, as_stream(stream),
                       num_a, boxes_a, num_b, boxes_b, ans_overlap, ans_iou);
    return launch_status();
}

HF_API int hf_nms_mask(const float *boxes, unsigned long long *mask, int boxes_num, float nms_overlap_thresh,
                       hf_stream_t stream)
{
    if (boxes_num <= 0 || !boxes || !mask) return HF_EINVAL;
    const int cb = (boxes_num + 63) / 64;
    if (cb > 65535) return HF_EINVAL;
    hipLaunchKernelGGL((nms_mask_kernel<false>), dim3(cb, cb), dim3(kNmsThreads), 0, as_stream(stream), boxes_num,
                       nms_overlap_thresh, boxes, mask);
    return launch_status();
}

HF_API size_t hf_oriented_nms_workspace(int n)
{
    if (n <= 0) return 0;
    const size_t cb = (static_cast<size_t>(n) + 63) / 64;
    return sizeof(unsigned long long) * static_cast<size_t>(n) * cb;
}

HF_API int hf_oriented_nms(const float *boxes, int n, float thresh, int *keep, int *num_kept, void *workspace,
                           size_t workspace_bytes, hf_stream_t stream)
{
    // OrientedNMSOp: nms_threshold >= 0 (bev_iou.cpp:52), N > 0 (:65)
    if (n <= 0 || !(thresh >= 0.0f) || !boxes || !keep) return HF_EINVAL;
    if (!workspace || workspace_bytes < hf_oriented_nms_workspace(n)) return HF_EWORKSPACE;
    const int cb = (n + 63) / 64;
    const size_t lds = sizeof(unsigned long long) * static_cast<size_t>(cb);
    if (cb > 65535 || lds > 150 * 1024) return HF_EINVAL;
    hipStream_t st = as_stream(stream);
    unsigned long long *mask = static_cast<unsigned long long *>(workspace);
    hipLaunchKernelGGL((nms_mask_kernel<true>), dim3(cb, cb), dim3(kNmsThreads), 0, st, n, thresh, boxes, mask);
    int rc = launch_status();
    if (rc != HF_OK) return rc;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&nms_sweep_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(kSweepThreads), lds, st, n, mask, keep, num_kept);
    return launch_status();
}
