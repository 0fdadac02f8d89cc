// bev_iou.hip -- rotated-BEV overlap / IoU matrix and oriented NMS for gfx950.
//
// Replaces bev_iou/bev_iou_g.cu:8-298 and the host half of OrientedNMSOp::Compute
// (bev_iou/bev_iou.cpp:60-116: cudaMalloc, blocking D2H of the mask, host greedy sweep, H2D).
//
// Structure (both the IoU matrix and the NMS mask kernel):
//   phase 0  one workgroup per 64 x 64 tile of pairs; cos/sin, rotated corners, centre and radius are
//            computed ONCE per box into LDS (the reference redoes them for every pair);
//   phase 1  every pair gets a trig-free bounding-circle test.  Far pairs are exactly 0 in the
//            reference too (no edge crossing, no corner inside: bev_iou_g.cu:150-176 leave
//            cnt = 0 -> area 0), so they are written as zeros straight away;
//   phase 2  the few surviving pairs are compacted into an LDS queue; on dense lanes they get a
//            separating-axis test (again exact-zero safe) and only then the expensive polygon clip
//            (16 edge tests, atan2 sort, shoelace).
// The clip itself follows the reference operation by operation (fp32, no contraction); cos/sin
// are evaluated once per box per pair-side (cos(-a) == cos(a), sin(-a) == -sin(a) exactly).
#include <math.h>

#include <algorithm>

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

// Everything about one box that does not depend on its partner: evaluated once per box per tile
// (the reference recomputes cos/sin and the rotated corners for every pair, bev_iou_g.cu:130-140).
struct BoxPre {
    Pt cor[4];              // rotated corners, order of bev_iou_g.cu:118-128
    float cs, sn;           // cos(angle), sin(angle)
    float box[5];           // x1, y1, x2, y2, angle
    float cx, cy, rad;      // centre and a bound on the circumradius (half perimeter), for the first filter
};

__device__ __forceinline__ void box_precompute(const float *b, BoxPre &o)
{
#pragma unroll
    for (int d = 0; d < 5; ++d) o.box[d] = b[d];
    const Pt c = { (b[0] + b[2]) / 2, (b[1] + b[3]) / 2 };
    o.cs = cosf(b[4]);
    o.sn = sinf(b[4]);
    o.cor[0] = rot_center(c, o.cs, o.sn, Pt{ b[0], b[1] });
    o.cor[1] = rot_center(c, o.cs, o.sn, Pt{ b[2], b[1] });
    o.cor[2] = rot_center(c, o.cs, o.sn, Pt{ b[2], b[3] });
    o.cor[3] = rot_center(c, o.cs, o.sn, Pt{ b[0], b[3] });
    o.cx = c.x;
    o.cy = c.y;
    o.rad = (fabsf(b[2] - b[0]) + fabsf(b[3] - b[1])) * 0.5f;
}

// first filter, 6 LDS words per pair: centres further apart than the two radius bounds plus a slack that
// dwarfs MARGIN = 1e-5 and fp32 rounding.  true => the reference computes exactly 0 (no edge crossing, no
// corner inside: bev_iou_g.cu:150-176 leave cnt = 0).
__device__ __forceinline__ bool circles_apart(const BoxPre &a, const BoxPre &b)
{
    const float mag = fabsf(a.cx) + fabsf(a.cy) + fabsf(b.cx) + fabsf(b.cy) + a.rad + b.rad;
    const float reach = a.rad + b.rad + 1e-3f + 1e-5f * mag;
    const float dx = a.cx - b.cx, dy = a.cy - b.cy;
    return dx * dx + dy * dy > reach * reach;  // NaN/inf compare false -> next filter
}

// second filter (survivors of the first only): separating-axis test over the four edge directions of the
// rotated corners, accepted only when the gap along some axis exceeds the same kind of slack.
__device__ __forceinline__ bool surely_disjoint(const BoxPre &a, const BoxPre &b)
{
    float mag = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        mag = fmaxf(mag, fmaxf(fmaxf(fabsf(a.cor[k].x), fabsf(a.cor[k].y)), fmaxf(fabsf(b.cor[k].x), fabsf(b.cor[k].y))));
    const float slack = 1e-3f + 1e-5f * mag;
    bool sep = false;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const BoxPre &p = which == 0 ? a : b;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // axis = edge direction cor[e+1] - cor[e] (not normalised; gaps are compared scaled by its length)
            const float ux = p.cor[e + 1].x - p.cor[e].x, uy = p.cor[e + 1].y - p.cor[e].y;
            const float len = fabsf(ux) + fabsf(uy);  // >= |u|: makes the required gap larger, never smaller
            float amin = INFINITY, amax = -INFINITY, bmin = INFINITY, bmax = -INFINITY;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float pa = a.cor[k].x * ux + a.cor[k].y * uy;
                const float pb = b.cor[k].x * ux + b.cor[k].y * uy;
                amin = fminf(amin, pa); amax = fmaxf(amax, pa);
                bmin = fminf(bmin, pb); bmax = fmaxf(bmax, pb);
            }
            const float gap = fmaxf(bmin - amax, amin - bmax);
            if (gap > (slack + 4e-6f * mag) * len && len > 0.f) sep = true;  // projections round at ~mag*|u|*1e-6
        }
    }
    return sep;  // NaN / inf inputs compare false -> full path
}

// box_overlap, bev_iou_g.cu:102-206, on precomputed corners
__device__ float box_overlap(const BoxPre &pa, const BoxPre &pb)
{
    Pt A[5], B[5];
#pragma unroll
    for (int k = 0; k < 4; ++k) { A[k] = pa.cor[k]; B[k] = pb.cor[k]; }
    A[4] = A[0];
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
        if (in_box2d(pa.box, pa.cs, -pa.sn, B[k])) {
            ctr.x = ctr.x + B[k].x; ctr.y = ctr.y + B[k].y;
            cp[cnt++] = B[k];
        }
        if (in_box2d(pb.box, pb.cs, -pb.sn, A[k])) {
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

// ---------------------------------------------------------------- 64 x 64 pair tiles
constexpr int kTileThreads = 256;

struct TileShared {
    BoxPre ra[64], cb[64];
    int queue[64 * 64];
    unsigned long long words[64];
    int qcount;
};

__device__ __forceinline__ void tile_stage(TileShared &sh, const float *boxes_r, int row0, int row_size,
                                           const float *boxes_c, int col0, int col_size)
{
    const int t = threadIdx.x;
    if (t < 64) {
        if (t < row_size) box_precompute(boxes_r + static_cast<size_t>(row0 + t) * 5, sh.ra[t]);
        sh.words[t] = 0ull;
    } else if (t < 128) {
        const int c = t - 64;
        if (c < col_size) box_precompute(boxes_c + static_cast<size_t>(col0 + c) * 5, sh.cb[c]);
    }
    if (t == 128) sh.qcount = 0;
}

// IoU matrix: one workgroup per 64 x 64 tile of (a, b) pairs
__global__ __launch_bounds__(kTileThreads) void bev_iou_kernel(int num_a, const float *__restrict__ boxes_a, int num_b,
                                                               const float *__restrict__ boxes_b,
                                                               float *__restrict__ ans_overlap,
                                                               float *__restrict__ ans_iou)
{
    __shared__ TileShared sh;
    const int t = threadIdx.x;
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    const int row_size = min(num_a - row0, 64), col_size = min(num_b - col0, 64);
    tile_stage(sh, boxes_a, row0, row_size, boxes_b, col0, col_size);
    __syncthreads();
    // filter 1 (all pairs, cheap): bounding circles
    for (int e = t; e < 64 * 64; e += kTileThreads) {
        const int r = e >> 6, c = e & 63;
        if (r < row_size && c < col_size) {
            if (circles_apart(sh.ra[r], sh.cb[c])) {
                const size_t o = static_cast<size_t>(row0 + r) * num_b + col0 + c;
                if (ans_overlap) ans_overlap[o] = 0.0f;
                if (ans_iou) ans_iou[o] = 0.0f;
            } else {
                sh.queue[atomicAdd(&sh.qcount, 1)] = e;
            }
        }
    }
    __syncthreads();
    // filter 2 (survivors only, dense lanes): separating axes; then the clip on what is left
    const int nq = sh.qcount;
    for (int q = t; q < nq; q += kTileThreads) {
        const int e = sh.queue[q];
        const int r = e >> 6, c = e & 63;
        float s = 0.0f, iou = 0.0f;
        if (!surely_disjoint(sh.ra[r], sh.cb[c])) {
            s = box_overlap(sh.ra[r], sh.cb[c]);
            iou = iou_from_overlap(sh.ra[r].box, sh.cb[c].box, s);
        }
        const size_t o = static_cast<size_t>(row0 + r) * num_b + col0 + c;
        if (ans_overlap) ans_overlap[o] = s;
        if (ans_iou) ans_iou[o] = iou;
    }
}

// ---------------------------------------------------------------- NMS mask
// one block per (row tile, col tile) of 64x64 pairs; mask word (row, col tile) bit j set iff
// iou(row, col*64+j) > thresh, j > row inside the diagonal tile (bev_iou_g.cu:256-298).
constexpr int kNmsThreads = kTileThreads;

template <bool UPPER_ONLY>
__global__ __launch_bounds__(kNmsThreads) void nms_mask_kernel(int n, float thresh, const float *__restrict__ boxes,
                                                               unsigned long long *__restrict__ mask)
{
    const int row_t = blockIdx.y, col_t = blockIdx.x;
    if (UPPER_ONLY && col_t < row_t) return;  // never read by the sweep (bev_iou.cpp:100-103 starts at nblock)
    // blockIdx.z = frame of a batched call: every frame has its own (n,5) boxes and (n, ceil(n/64)) mask
    boxes += static_cast<size_t>(blockIdx.z) * n * 5;
    mask += static_cast<size_t>(blockIdx.z) * n * ((n + 63) / 64);
    __shared__ TileShared sh;
    const int t = threadIdx.x;
    const int row_size = min(n - row_t * 64, 64), col_size = min(n - col_t * 64, 64);
    tile_stage(sh, boxes, row_t * 64, row_size, boxes, col_t * 64, col_size);
    __syncthreads();
    for (int e = t; e < 64 * 64; e += kNmsThreads) {
        const int r = e >> 6, c = e & 63;
        const bool valid = r < row_size && c < col_size && !(row_t == col_t && c <= r);
        if (valid && !circles_apart(sh.ra[r], sh.cb[c])) sh.queue[atomicAdd(&sh.qcount, 1)] = e;
    }
    __syncthreads();
    const int nq = sh.qcount;
    for (int q = t; q < nq; q += kNmsThreads) {
        const int e = sh.queue[q];
        const int r = e >> 6, c = e & 63;
        if (surely_disjoint(sh.ra[r], sh.cb[c])) continue;
        const float s = box_overlap(sh.ra[r], sh.cb[c]);
        if (iou_from_overlap(sh.ra[r].box, sh.cb[c].box, s) > thresh) atomicOr(&sh.words[r], 1ull << c);
    }
    __syncthreads();
    const int col_blocks = (n + 63) / 64;
    if (t < row_size) mask[(static_cast<size_t>(row_t) * 64 + t) * col_blocks + col_t] = sh.words[t];
}

// ---------------------------------------------------------------- greedy sweep on the device
// bev_iou.cpp:87-112 without the host.  One 1024-thread workgroup walks the column blocks in order.
// For column block `blk` the removal word is PULLED: remv = OR over every box kept so far of
// mask[box][blk] (the kept list is in LDS; all threads OR their share, then a tree reduction) -- the
// host loop pushes each kept row into all later words instead, which serialises on one thread.  Wave 0
// then resolves the 64 boxes of the block against each other from the diagonal words (one per lane).
constexpr int kSweepThreads = 1024;

__global__ __launch_bounds__(kSweepThreads) void nms_sweep_kernel(int n, const unsigned long long *__restrict__ mask,
                                                                  int *__restrict__ keep, int *__restrict__ num_kept)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int *kept_list = reinterpret_cast<int *>(smem_raw);  // up to n kept box indices
    // blockIdx.x = frame of a batched call
    mask += static_cast<size_t>(blockIdx.x) * n * ((n + 63) / 64);
    keep += static_cast<size_t>(blockIdx.x) * n;
    if (num_kept) num_kept += blockIdx.x;
    __shared__ unsigned long long red[kSweepThreads / 64];
    __shared__ unsigned long long remv_word;
    __shared__ int kept_total;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cb = (n + 63) / 64;
    if (t == 0) kept_total = 0;
    __syncthreads();
    for (int blk = 0; blk < cb; ++blk) {
        // ---- pull: which boxes of this block are already suppressed by earlier kept boxes ----
        const int kt = kept_total;
        unsigned long long acc = 0ull;
        for (int i = t; i < kt; i += kSweepThreads) acc |= mask[static_cast<size_t>(kept_list[i]) * cb + blk];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const unsigned lo = __shfl_xor(static_cast<unsigned>(acc), d);
            const unsigned hi = __shfl_xor(static_cast<unsigned>(acc >> 32), d);
            acc |= (static_cast<unsigned long long>(hi) << 32) | lo;
        }
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (t == 0) {
            unsigned long long r = 0ull;
#pragma unroll
            for (int w = 0; w < kSweepThreads / 64; ++w) r |= red[w];
            remv_word = r;
        }
        __syncthreads();
        // ---- wave 0: sequential resolution inside the block ----
        if (t < 64) {
            const int i = blk * 64 + t;
            const unsigned long long diag = i < n ? mask[static_cast<size_t>(i) * cb + blk] : 0ull;
            unsigned long long word = remv_word;
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
            if ((kb >> t) & 1ull) {
                const int pos = before + __builtin_popcountll(kb & ((1ull << t) - 1ull));
                kept_list[pos] = i;
                keep[pos] = i;
            }
            if (t == 0) kept_total = before + __builtin_popcountll(kb);
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
    const int gy = (num_a + 63) / 64, gx = (num_b + 63) / 64;
    if (gy > 65535) {
        // very tall matrices: walk the rows in slabs of 65535 tiles
        for (int y0 = 0; y0 < gy; y0 += 65535) {
            const int rows0 = y0 * 64;
            const int na = std::min(num_a - rows0, 65535 * 64);
            hipLaunchKernelGGL(bev_iou_kernel, dim3(gx, (na + 63) / 64), dim3(kTileThreads), 0, as_stream(stream), na,
                               boxes_a + static_cast<size_t>(rows0) * 5, num_b, boxes_b,
                               ans_overlap ? ans_overlap + static_cast<size_t>(rows0) * num_b : nullptr,
                               ans_iou ? ans_iou + static_cast<size_t>(rows0) * num_b : nullptr);
        }
        return launch_status();
    }
    hipLaunchKernelGGL(bev_iou_kernel, dim3(gx, gy), dim3(kTileThreads), 0, as_stream(stream), num_a, boxes_a, num_b,
                       boxes_b, ans_overlap, ans_iou);
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
    return hf_oriented_nms_batched(1, boxes, n, thresh, keep, num_kept, workspace, workspace_bytes, stream);
}

HF_API int hf_oriented_nms_batched(int frames, const float *boxes, int n, float thresh, int *keep, int *num_kept,
                                   void *workspace, size_t workspace_bytes, hf_stream_t stream)
{
    // OrientedNMSOp: nms_threshold >= 0 (bev_iou.cpp:52), N > 0 (:65)
    if (frames <= 0 || frames > 65535 || n <= 0 || !(thresh >= 0.0f) || !boxes || !keep) return HF_EINVAL;
    if (!workspace || workspace_bytes < static_cast<size_t>(frames) * hf_oriented_nms_workspace(n)) return HF_EWORKSPACE;
    const int cb = (n + 63) / 64;
    const size_t lds = sizeof(int) * static_cast<size_t>(n);  // kept list
    if (cb > 65535 || lds > 140 * 1024) return HF_EINVAL;     // n <= ~35 000 boxes (pre_nms_size is 9000)
    hipStream_t st = as_stream(stream);
    unsigned long long *mask = static_cast<unsigned long long *>(workspace);
    hipLaunchKernelGGL((nms_mask_kernel<true>), dim3(cb, cb, frames), dim3(kNmsThreads), 0, st, n, thresh, boxes, mask);
    int rc = launch_status();
    if (rc != HF_OK) return rc;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&nms_sweep_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(frames), dim3(kSweepThreads), lds, st, n, mask, keep, num_kept);
    return launch_status();
}
