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
#include <stdlib.h>

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

// box_overlap, bev_iou_g.cu:102-206, on precomputed corners -- SIXTEEN lanes per pair.
// As one thread per pair the clip is a dependent chain of ~2000 instructions (16 edge tests with divisions, up to 16
// atan2f, a sort, a fan) that sets the latency of a whole tile.  Here the 16 lanes of a group split it:
//   lane s           tests edge pair (i, j) = (s >> 2, s & 3)                       -> candidate point s      (:150-164)
//   lanes 0..7       test one corner each: B[k] inside A (even), A[k] inside B (odd) -> candidate point 16 + s (:166-176)
//   every lane       adds up the valid points IN THE REFERENCE'S ORDER (slots 0..23 ascending) -> the same centroid bits
//   point owners     atan2f of their point(s); rank = valid points with a smaller angle, or an equal angle and a smaller
//                    slot (= the position the reference's stable bubble sort gives the point, :181-190)
//   lanes 0..13      one fan term each (points of rank r, r+1 against rank 0);  the terms are then summed in rank
//                    order, so the area has the reference's rounding sequence (:192-197).
// Scratch per group in LDS: 24 points, 24 angles, 24 sorted slots, 16 terms.  The lanes of a group sit in one wave: the
// LDS executes a wave's accesses in order, a compiler barrier between the steps is all the synchronisation needed.
constexpr int kClipSlots = 24;
struct ClipScratch {
    float px[kClipSlots], py[kClipSlots], ang[kClipSlots];
    int sorted[kClipSlots];
    float term[16];
};

__device__ __forceinline__ void group_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// all 16 lanes of a group call this together (sub = lane & 15); every lane returns the area
__device__ float box_overlap_group(const BoxPre &pa, const BoxPre &pb, ClipScratch &cs, int sub)
{
    // ---- candidate points ----
    const int i = sub >> 2, j = sub & 3;
    Pt x1;
    const bool hit1 = seg_intersection(pa.cor[(i + 1) & 3], pa.cor[i], pb.cor[(j + 1) & 3], pb.cor[j], x1);
    bool hit2 = false;
    Pt x2 = { 0.f, 0.f };
    if (sub < 8) {
        const int k = sub >> 1;
        if (sub & 1) { x2 = pa.cor[k]; hit2 = in_box2d(pb.box, pb.cs, -pb.sn, x2); }   // A[k] inside B
        else { x2 = pb.cor[k]; hit2 = in_box2d(pa.box, pa.cs, -pa.sn, x2); }          // B[k] inside A
    }
    const int row = (threadIdx.x & 63) & ~15;   // first lane of my group inside the wave
    const unsigned m1 = static_cast<unsigned>(__ballot(hit1) >> row) & 0xffffu;
    const unsigned m2 = static_cast<unsigned>(__ballot(hit2) >> row) & 0xffu;
    const unsigned valid = m1 | (m2 << 16);
    const int cnt = __builtin_popcount(valid);
    if (cnt < 3) return 0.0f;   // group-uniform: fewer than 3 points, the fan sums nothing but zeros
    if (hit1) { cs.px[sub] = x1.x; cs.py[sub] = x1.y; }
    if (hit2) { cs.px[16 + sub] = x2.x; cs.py[16 + sub] = x2.y; }
    group_sync();
    // ---- centroid: the same sequence of additions as the reference ----
    Pt ctr = { 0.f, 0.f };
    for (unsigned m = valid; m; m &= m - 1u) {
        const int sl = __builtin_ctz(m);
        ctr.x = ctr.x + cs.px[sl]; ctr.y = ctr.y + cs.py[sl];
    }
    ctr.x /= cnt;
    ctr.y /= cnt;
    // ---- angles (point_cmp :98-100 evaluates atan2 of the point minus the centre) ----
    float a1 = 0.f, a2 = 0.f;
    if (hit1) { a1 = atan2f(x1.y - ctr.y, x1.x - ctr.x); cs.ang[sub] = a1; }
    if (hit2) { a2 = atan2f(x2.y - ctr.y, x2.x - ctr.x); cs.ang[16 + sub] = a2; }
    group_sync();
    // ---- position of my point(s) after the reference's stable sort ----
    int r1 = 0, r2 = 0;
    for (unsigned m = valid; m; m &= m - 1u) {
        const int sl = __builtin_ctz(m);
        const float aj = cs.ang[sl];
        r1 += (aj < a1 || (aj == a1 && sl < sub)) ? 1 : 0;
        r2 += (aj < a2 || (aj == a2 && sl < 16 + sub)) ? 1 : 0;
    }
    if (hit1) cs.sorted[r1] = sub;
    if (hit2) cs.sorted[r2] = 16 + sub;
    group_sync();
    // ---- fan terms, then their sum in rank order ----
    const int r = sub + 1;
    if (r <= cnt - 2) {
        const int s0 = cs.sorted[0], sk = cs.sorted[r], sn = cs.sorted[r + 1];
        const float x0 = cs.px[s0], y0 = cs.py[s0];
        const float ux = cs.px[sk] - x0, uy = cs.py[sk] - y0;
        const float vx = cs.px[sn] - x0, vy = cs.py[sn] - y0;
        cs.term[r] = ux * vy - uy * vx;
    }
    group_sync();
    float area = 0.f;
    for (int k = 1; k <= cnt - 2; ++k) area += cs.term[k];   // (the reference's k = 0 term is exactly 0)
    group_sync();   // the scratch is reused by the group's next pair
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
    unsigned short queue[64 * 64];    // pairs that passed the bounding-circle filter
    unsigned short queue2[64 * 64];   // ... and the separating-axis filter
    unsigned long long words[64];
    int qcount, q2count;
    ClipScratch clip[kTileThreads / 16];
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
    if (t == 128) { sh.qcount = 0; sh.q2count = 0; }
}

// IoU matrix: one workgroup per 64 x 64 tile of (a, b) pairs
__global__ __launch_bounds__(kTileThreads, 4) void bev_iou_kernel(int num_a, const float *__restrict__ boxes_a, int num_b,
                                                               const float *__restrict__ boxes_b,
                                                               float *__restrict__ ans_overlap,
                                                               float *__restrict__ ans_iou, int stop)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TileShared &sh = *reinterpret_cast<TileShared *>(smem_raw);
    const int t = threadIdx.x;
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    const int row_size = min(num_a - row0, 64), col_size = min(num_b - col0, 64);
    if (stop == 1) return;   // diagnostics only (HF_BEV_STOP): outputs invalid
    tile_stage(sh, boxes_a, row0, row_size, boxes_b, col0, col_size);
    __syncthreads();
    if (stop == 2) return;
    // filter 1 (all pairs, cheap): bounding circles.  A thread owns four consecutive columns of a row: when all four are
    // far apart (the usual case) the zeros leave as one 16-byte store per output
    const bool vec = (num_b & 3) == 0;   // rows of the outputs are 16-byte aligned (the launcher checks the base pointers)
    for (int e4 = t; e4 < 64 * 16; e4 += kTileThreads) {
        const int r = e4 >> 4, c0 = (e4 & 15) * 4;
        if (r >= row_size || c0 >= col_size) continue;
        bool apart[4];
        bool all_apart = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            apart[i] = c0 + i < col_size ? circles_apart(sh.ra[r], sh.cb[c0 + i]) : true;
            all_apart = all_apart && apart[i];
        }
        const size_t o = static_cast<size_t>(row0 + r) * num_b + col0 + c0;
        if (all_apart && vec && c0 + 4 <= col_size) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ans_overlap) *reinterpret_cast<float4 *>(ans_overlap + o) = z;
            if (ans_iou) *reinterpret_cast<float4 *>(ans_iou + o) = z;
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (c0 + i >= col_size) continue;
            if (apart[i]) {
                if (ans_overlap) ans_overlap[o + i] = 0.0f;
                if (ans_iou) ans_iou[o + i] = 0.0f;
            } else {
                sh.queue[atomicAdd(&sh.qcount, 1)] = static_cast<unsigned short>((r << 6) | (c0 + i));
            }
        }
    }
    __syncthreads();
    if (stop == 3) return;
    // filter 2 (survivors only, dense lanes): separating axes -> exact zeros or the second queue
    const int nq = sh.qcount;
    for (int q = t; q < nq; q += kTileThreads) {
        const int e = sh.queue[q];
        const int r = e >> 6, c = e & 63;
        if (surely_disjoint(sh.ra[r], sh.cb[c])) {
            const size_t o = static_cast<size_t>(row0 + r) * num_b + col0 + c;
            if (ans_overlap) ans_overlap[o] = 0.0f;
            if (ans_iou) ans_iou[o] = 0.0f;
        } else {
            sh.queue2[atomicAdd(&sh.q2count, 1)] = static_cast<unsigned short>(e);
        }
    }
    __syncthreads();
    // the clip on what is left: sixteen lanes per pair, sixteen pairs per pass
    if (stop == 4) return;
    const int nq2 = sh.q2count;
    const int grp = t >> 4, sub = t & 15;
    for (int q0 = 0; q0 < nq2; q0 += kTileThreads / 16) {
        const int q = q0 + grp;
        if (q >= nq2) continue;   // whole groups leave together
        const int e = sh.queue2[q];
        const int r = e >> 6, c = e & 63;
        const float s = box_overlap_group(sh.ra[r], sh.cb[c], sh.clip[grp], sub);
        if (sub == 0) {
            const size_t o = static_cast<size_t>(row0 + r) * num_b + col0 + c;
            if (ans_overlap) ans_overlap[o] = s;
            if (ans_iou) ans_iou[o] = iou_from_overlap(sh.ra[r].box, sh.cb[c].box, s);
        }
    }
}

// ---------------------------------------------------------------- NMS mask
// one block per (row tile, col tile) of 64x64 pairs; mask word (row, col tile) bit j set iff
// iou(row, col*64+j) > thresh, j > row inside the diagonal tile (bev_iou_g.cu:256-298).
constexpr int kNmsThreads = kTileThreads;

template <bool UPPER_ONLY>
__global__ __launch_bounds__(kNmsThreads, 4) void nms_mask_kernel(int n, float thresh, const float *__restrict__ boxes,
                                                               unsigned long long *__restrict__ mask)
{
    const int row_t = blockIdx.y, col_t = blockIdx.x;
    if (UPPER_ONLY && col_t < row_t) return;  // never read by the sweep (bev_iou.cpp:100-103 starts at nblock)
    // blockIdx.z = frame of a batched call: every frame has its own (n,5) boxes and (n, ceil(n/64)) mask
    boxes += static_cast<size_t>(blockIdx.z) * n * 5;
    mask += static_cast<size_t>(blockIdx.z) * n * ((n + 63) / 64);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TileShared &sh = *reinterpret_cast<TileShared *>(smem_raw);
    const int t = threadIdx.x;
    const int row_size = min(n - row_t * 64, 64), col_size = min(n - col_t * 64, 64);
    tile_stage(sh, boxes, row_t * 64, row_size, boxes, col_t * 64, col_size);
    __syncthreads();
    for (int e = t; e < 64 * 64; e += kNmsThreads) {
        const int r = e >> 6, c = e & 63;
        const bool valid = r < row_size && c < col_size && !(row_t == col_t && c <= r);
        if (valid && !circles_apart(sh.ra[r], sh.cb[c])) sh.queue[atomicAdd(&sh.qcount, 1)] = static_cast<unsigned short>(e);
    }
    __syncthreads();
    const int nq = sh.qcount;
    for (int q = t; q < nq; q += kNmsThreads) {
        const int e = sh.queue[q];
        if (!surely_disjoint(sh.ra[e >> 6], sh.cb[e & 63])) sh.queue2[atomicAdd(&sh.q2count, 1)] = static_cast<unsigned short>(e);
    }
    __syncthreads();
    {
        const int nq2 = sh.q2count;
        const int grp = t >> 4, sub = t & 15;
        for (int q0 = 0; q0 < nq2; q0 += kNmsThreads / 16) {
            const int q = q0 + grp;
            if (q >= nq2) continue;   // whole groups leave together
            const int e = sh.queue2[q];
            const int r = e >> 6, c = e & 63;
            const float s = box_overlap_group(sh.ra[r], sh.cb[c], sh.clip[grp], sub);
            if (sub == 0 && iou_from_overlap(sh.ra[r].box, sh.cb[c].box, s) > thresh) atomicOr(&sh.words[r], 1ull << c);
        }
    }
    __syncthreads();
    const int col_blocks = (n + 63) / 64;
    if (t < row_size) mask[(static_cast<size_t>(row_t) * 64 + t) * col_blocks + col_t] = sh.words[t];
}

// ---------------------------------------------------------------- greedy sweep on the device
// bev_iou.cpp:87-112 without the host.  One 1024-thread workgroup per frame walks the column blocks in order, the
// removal words remv[col block] live in LDS.  Per block of 64 boxes:
//   resolve  wave 0: lane l holds the diagonal word of box blk*64+l; a scalar loop visits only the boxes that are still
//            alive (find-first-set on the alive mask), keeps each and ORs its diagonal word into the alive mask's
//            complement -- the host loop's `if (!(remv[nblock] & 1 << inblock))`;
//   push     all threads: every kept box of the block ORs its mask row, words blk+1 .. cb-1, into remv[] (the host
//            loop's `remv[j] |= p[j]`): thread -> (one of four row groups, word), rows are read with coalesced 8-byte
//            loads, combined in registers and merged with one LDS atomic per thread and word.
// (The round-1 form PULLED instead: for every block it re-read one word of every box kept so far, 8-byte reads at a
// 1 KB stride -- 0.95 ms of the 1.14 ms at 9000 boxes.)
constexpr int kSweepThreads = 1024;

__global__ __launch_bounds__(kSweepThreads) void nms_sweep_kernel(int n, const unsigned long long *__restrict__ mask,
                                                                  int *__restrict__ keep, int *__restrict__ num_kept)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long *remv = reinterpret_cast<unsigned long long *>(smem_raw);   // cb words
    // blockIdx.x = frame of a batched call
    const int cb = (n + 63) / 64;
    mask += static_cast<size_t>(blockIdx.x) * n * cb;
    keep += static_cast<size_t>(blockIdx.x) * n;
    if (num_kept) num_kept += blockIdx.x;
    __shared__ unsigned long long kept_bits[2];
    __shared__ int kept_total;
    __shared__ int klist[2][64];   // the kept boxes of a block (lane numbers), in order; double-buffered by block parity
    const int t = threadIdx.x;
    for (int w = t; w < cb; w += kSweepThreads) remv[w] = 0ull;
    if (t == 0) { kept_total = 0; kept_bits[0] = 0ull; kept_bits[1] = 0ull; }
    __syncthreads();
    // Step blk: wave 0 resolves block blk while waves 1..15 push the rows kept in block blk-1 into words blk+1.. .
    // Word blk of those rows -- the one wave 0 needs right now -- is fetched by wave 0 itself.
    // wave 0 keeps two words per lane one step ahead of their use (they do not depend on any decision): the diagonal
    // word of its box in the next block and word blk+1 of its box in the current block
    unsigned long long diag = 0ull, prevw = 0ull;
    if (t < 64) diag = t < n ? mask[static_cast<size_t>(t) * cb] : 0ull;
    for (int blk = 0; blk <= cb; ++blk) {
        const int par = blk & 1;
        if (t < 64) {
            if (blk < cb) {
                unsigned long long diag_n = 0ull, prevw_n = 0ull;
                if (blk + 1 < cb) {
                    const int in = (blk + 1) * 64 + t, ic = blk * 64 + t;
                    diag_n = in < n ? mask[static_cast<size_t>(in) * cb + blk + 1] : 0ull;
                    prevw_n = ic < n ? mask[static_cast<size_t>(ic) * cb + blk + 1] : 0ull;
                }
                const unsigned long long kprev = blk > 0 ? kept_bits[par ^ 1] : 0ull;
                if ((kprev >> t) & 1ull) atomicOr(&remv[blk], prevw);
                const int i = blk * 64 + t;
                const int lim = min(64, n - blk * 64);
                const unsigned long long valid = lim == 64 ? ~0ull : ((1ull << lim) - 1ull);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                asm volatile("" ::: "memory");
                unsigned long long dead = remv[blk];   // suppressed by boxes kept in earlier blocks
                unsigned long long kb = 0ull;
                unsigned long long alive = ~dead & valid;
                while (alive) {   // wave-uniform scalar loop over the boxes that survive
                    const int l = __builtin_ctzll(alive);
                    kb |= 1ull << l;
                    const unsigned lo = __builtin_amdgcn_readlane(static_cast<int>(diag & 0xffffffffu), l);
                    const unsigned hi = __builtin_amdgcn_readlane(static_cast<int>(diag >> 32), l);
                    dead |= (static_cast<unsigned long long>(hi) << 32) | lo;   // bits <= l of a diagonal word are never set
                    alive = ~dead & valid & ~((2ull << l) - 1ull);
                }
                const int before = kept_total;
                if ((kb >> t) & 1ull) {
                    const int o = __builtin_popcountll(kb & ((1ull << t) - 1ull));
                    keep[before + o] = i;
                    klist[par][o] = t;
                }
                if (t == 0) { kept_total = before + __builtin_popcountll(kb); kept_bits[par] = kb; }
                diag = diag_n;
                prevw = prevw_n;
            }
        } else if (blk > 0) {
            // rows kept in block blk-1 -> words blk+1 .. cb-1; thread -> (row group g of 4, word w): consecutive threads
            // read consecutive words of a row (coalesced); group g takes the kept boxes number g, g+4, ...
            const unsigned long long kb = kept_bits[par ^ 1];
            const int nw = cb - blk - 1;
            if (kb != 0ull && nw > 0) {
                const int u = t - 64;
                const int g = u / 240;
                const int nk = __builtin_popcountll(kb);
                const int *kl = klist[par ^ 1];
                for (int w = u - g * 240; w < nw; w += 240) {
                    unsigned long long acc = 0ull;
                    for (int o = g; o < nk; o += 4)
                        acc |= mask[static_cast<size_t>((blk - 1) * 64 + kl[o]) * cb + blk + 1 + w];
                    if (acc) atomicOr(&remv[blk + 1 + w], acc);
                }
            }
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

// the pair-tile kernels keep their state (and the clip's point arrays) in dynamic LDS: ~33 KB, four workgroups per CU
template <typename K>
static void tile_lds_attr(K kernel)
{
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(sizeof(TileShared)));
        done = true;
    }
}

HF_API int hf_compute_bev_iou(int num_a, const float *boxes_a, int num_b, const float *boxes_b, float *ans_overlap,
                              float *ans_iou, hf_stream_t stream)
{
    // ComputeBevIOUOp: N > 0, M > 0, (N,5) / (M,5)  (bev_iou.cpp:156-157)
    if (num_a <= 0 || num_b <= 0 || !boxes_a || !boxes_b) return HF_EINVAL;
    if (!ans_overlap && !ans_iou) return HF_OK;
    const int gy = (num_a + 63) / 64, gx = (num_b + 63) / 64;
    tile_lds_attr(&bev_iou_kernel);
    const char *stop_env = getenv("HF_BEV_STOP");   // diagnostics only: early exit after a phase (outputs invalid)
    const int stop = stop_env && stop_env[0] ? atoi(stop_env) : 0;
    if ((ans_overlap && reinterpret_cast<uintptr_t>(ans_overlap) % 16 != 0) || (ans_iou && reinterpret_cast<uintptr_t>(ans_iou) % 16 != 0))
        return HF_EINVAL;   // the outputs are written with 16-byte stores (every allocator returns at least that alignment)
    if (gy > 65535) {
        // very tall matrices: walk the rows in slabs of 65535 tiles
        for (int y0 = 0; y0 < gy; y0 += 65535) {
            const int rows0 = y0 * 64;
            const int na = std::min(num_a - rows0, 65535 * 64);
            hipLaunchKernelGGL(bev_iou_kernel, dim3(gx, (na + 63) / 64), dim3(kTileThreads), sizeof(TileShared), as_stream(stream), na,
                               boxes_a + static_cast<size_t>(rows0) * 5, num_b, boxes_b,
                               ans_overlap ? ans_overlap + static_cast<size_t>(rows0) * num_b : nullptr,
                               ans_iou ? ans_iou + static_cast<size_t>(rows0) * num_b : nullptr, stop);
        }
        return launch_status();
    }
    hipLaunchKernelGGL(bev_iou_kernel, dim3(gx, gy), dim3(kTileThreads), sizeof(TileShared), as_stream(stream), num_a, boxes_a, num_b,
                       boxes_b, ans_overlap, ans_iou, stop);
    return launch_status();
}

HF_API int hf_nms_mask(const float *boxes, unsigned long long *mask, int boxes_num, float nms_overlap_thresh,
                       hf_stream_t stream)
{
    if (boxes_num <= 0 || !boxes || !mask) return HF_EINVAL;
    const int cb = (boxes_num + 63) / 64;
    if (cb > 65535) return HF_EINVAL;
    tile_lds_attr(&nms_mask_kernel<false>);
    hipLaunchKernelGGL((nms_mask_kernel<false>), dim3(cb, cb), dim3(kNmsThreads), sizeof(TileShared), as_stream(stream), boxes_num,
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
    const size_t lds = sizeof(unsigned long long) * static_cast<size_t>(cb);  // removal words
    if (cb > 16384) return HF_EINVAL;     // n <= ~1 000 000 boxes (pre_nms_size is 9000)
    hipStream_t st = as_stream(stream);
    unsigned long long *mask = static_cast<unsigned long long *>(workspace);
    tile_lds_attr(&nms_mask_kernel<true>);
    hipLaunchKernelGGL((nms_mask_kernel<true>), dim3(cb, cb, frames), dim3(kNmsThreads), sizeof(TileShared), st, n, thresh, boxes, mask);
    int rc = launch_status();
    if (rc != HF_OK) return rc;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&nms_sweep_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(frames), dim3(kSweepThreads), lds, st, n, mask, keep, num_kept);
    return launch_status();
}
