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
#include <type_traits>

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
struct __attribute__((aligned(16))) BoxPre {
    float cx, cy, rad, mag; // centre, the circumradius (half diagonal), |cx| + |cy| + rad: the first filter
                            // reads these four with one 16-byte LDS access
    Pt cor[4];              // rotated corners, order of bev_iou_g.cu:118-128
    float cs, sn;           // cos(angle), sin(angle)
    float box[5];           // x1, y1, x2, y2, angle
    float pad_;
};

static_assert(sizeof(BoxPre) == 80, "the NMS workspace table and its 16-byte copies assume 80 bytes per box");

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
    // half the diagonal: the distance from the centre to every rotated corner (rotation keeps it; its fp32 rounding is covered
    // by the filters' slack of 1e-3 + 1e-5 * mag).  Round 4: was the half perimeter (w + h) / 2, which is 30 % longer for a
    // 3.9 x 1.6 box -- 1.7 x as many pairs survived the circle filter and went through the separating-axis test
    const float bw = b[2] - b[0], bh = b[3] - b[1];
    o.rad = 0.5f * sqrtf(bw * bw + bh * bh) * 1.000001f;
    o.mag = fabsf(o.cx) + fabsf(o.cy) + o.rad;
}

// first filter, 6 LDS words per pair: centres further apart than the two radius bounds plus a slack that
// dwarfs MARGIN = 1e-5 and fp32 rounding.  true => the reference computes exactly 0 (no edge crossing, no
// corner inside: bev_iou_g.cu:150-176 leave cnt = 0).
__device__ __forceinline__ bool circles_apart(const BoxPre &a, const BoxPre &b)
{
    const float mag = a.mag + b.mag;
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

// Third filter, NMS only (the mask needs `iou > thresh`, not the value): an upper bound on the overlap area that costs a few dozen
// instructions instead of the ~600 x 8 lanes of the clip.  For an edge direction u of box P (side length L along u, L' across it),
// A n B lies inside P n {x : proj_u(x) in proj_u(other box)}: a rectangle of area (overlap of the two projections) x L' -- and
// inside the intersection of P's two such slabs, the rectangle (overlap along u) x (overlap along u').  The
// smallest of these six bounds (per box: two slabs and their product), padded by the same kind of slack as the other filters, bounds the reference's
// overlap from above; iou is increasing in the overlap, so `bound / (sa + sb - bound) < 0.98 thresh` means the reference's
// `iou > thresh` (bev_iou_g.cu:208-215, :283) is false and the mask bit is 0 either way.  At the RPN's threshold of 0.8 this
// removes nearly every pair that survived the separating-axis test (two boxes that merely touch); at 0.01 it removes nothing
// and costs ~40 instructions per surviving pair.  Degenerate boxes (a side <= 0, NaN / inf anywhere) compare false: full path.
__device__ __forceinline__ bool iou_surely_below(const BoxPre &a, const BoxPre &b, float thresh)
{
    const float wa = a.box[2] - a.box[0], ha = a.box[3] - a.box[1], wb = b.box[2] - b.box[0], hb = b.box[3] - b.box[1];
    if (!(wa > 0.f && ha > 0.f && wb > 0.f && hb > 0.f)) return false;
    float mag = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        mag = fmaxf(mag, fmaxf(fmaxf(fabsf(a.cor[k].x), fabsf(a.cor[k].y)), fmaxf(fabsf(b.cor[k].x), fabsf(b.cor[k].y))));
    const float slack = 1e-3f + 1e-5f * mag;   // metres along the axis: rounding of corners and projections, MARGIN = 1e-5
    float ub = INFINITY;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const BoxPre &p = which == 0 ? a : b;
        const float w = which == 0 ? wa : wb, h = which == 0 ? ha : hb;
        float both = 1.0f;   // in P's own frame A n B also lies inside the rectangle [overlap along x] x [overlap along y]
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // cor[0] -> cor[1] runs along the box's own x axis (length w), cor[1] -> cor[2] along its y axis (length h)
            const float ux = p.cor[e + 1].x - p.cor[e].x, uy = p.cor[e + 1].y - p.cor[e].y;
            const float along = e == 0 ? w : h, across = e == 0 ? h : w;
            float amin = INFINITY, amax = -INFINITY, bmin = INFINITY, bmax = -INFINITY;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float pa = a.cor[k].x * ux + a.cor[k].y * uy;
                const float pb = b.cor[k].x * ux + b.cor[k].y * uy;
                amin = fminf(amin, pa); amax = fmaxf(amax, pa);
                bmin = fminf(bmin, pb); bmax = fmaxf(bmax, pb);
            }
            // projections are scaled by |u| = along (up to rounding): overlap length in metres, padded
            const float ov = fmaxf(fminf(amax, bmax) - fmaxf(amin, bmin), 0.f) / along + slack;
            ub = fminf(ub, ov * (across + slack));
            both *= ov;
        }
        ub = fminf(ub, both);
    }
    const float sa = wa * ha, sb = wb * hb;
    const float den = sa + sb - ub;
    return den > 0.f && ub < 0.98f * thresh * den;   // NaN / inf anywhere: false
}

// box_overlap, bev_iou_g.cu:102-206, on precomputed corners -- EIGHT lanes per pair.
// As one thread per pair the clip is a dependent chain of ~2000 instructions (16 edge tests with divisions, up to 24
// atan2f, a sort, a fan) that sets the latency of a whole tile; sixteen lanes per pair (the first cooperative form) cut
// the chain but kept every compute unit's vector ALUs busy with mostly idle lanes: the clip of ~25 surviving pairs of a
// tile was 8 us of a 20 us kernel.  Eight lanes per pair: half the lane-instructions per pair and 32 pairs per pass of a
// 256-thread workgroup.  The lanes of a group split the work:
//   lane s           tests edge pairs (i, j) = (s >> 2, s & 3) and (2 + (s >> 2), s & 3) -> candidate points s, s + 8  (:150-164)
//   lane s           tests one corner: B[s >> 1] inside A (s even), A[s >> 1] inside B (s odd) -> candidate point 16 + s (:166-176)
//   every lane       adds up the valid points IN THE REFERENCE'S ORDER (slots 0..23 ascending) -> the same centroid bits
//   lane s           takes the s-th VALID point (s+8-th ...): its atan2f; rank = valid points with a smaller angle, or an
//                    equal angle and a smaller slot (= the position the reference's stable bubble sort gives it, :181-190)
//   every lane       up to three fan terms (points of rank r, r+1 against rank 0), r = s+1, s+9, s+17; the terms are
//                    then summed in rank order, so the area has the reference's rounding sequence (:192-197).
// Scratch per group in LDS: 24 points, 24 angles, 24 sorted slots, 24 terms.  The lanes of a group sit in one wave: the
// LDS executes a wave's accesses in order, a compiler barrier between the steps is all the synchronisation needed.
constexpr int kClipSlots = 24;
constexpr int kClipLanes = 8;
struct ClipScratch {
    float px[kClipSlots], py[kClipSlots];
    union { float ang[kClipSlots]; float term[kClipSlots]; };   // the angles are dead once every point has its rank
    unsigned char sorted[kClipSlots];
};

__device__ __forceinline__ void group_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// all 8 lanes of a group call this together (sub = lane & 7); every lane returns the area
__device__ float box_overlap_group(const BoxPre &pa, const BoxPre &pb, ClipScratch &cs, int sub, int stop = 0)
{
    // ---- candidate points: slots sub, sub + 8 (edge crossings), 16 + sub (a corner inside the other box) ----
    const int j = sub & 3, i0 = sub >> 2, i1 = i0 + 2;
    const Pt qa = pb.cor[(j + 1) & 3], qb = pb.cor[j];
    Pt x[3];
    bool hit[3];
    hit[0] = seg_intersection(pa.cor[(i0 + 1) & 3], pa.cor[i0], qa, qb, x[0]);
    hit[1] = seg_intersection(pa.cor[(i1 + 1) & 3], pa.cor[i1], qa, qb, x[1]);
    {
        const int k = sub >> 1;
        if (sub & 1) { x[2] = pa.cor[k]; hit[2] = in_box2d(pb.box, pb.cs, -pb.sn, x[2]); }   // A[k] inside B
        else { x[2] = pb.cor[k]; hit[2] = in_box2d(pa.box, pa.cs, -pa.sn, x[2]); }          // B[k] inside A
    }
    const int row = (threadIdx.x & 63) & ~(kClipLanes - 1);   // first lane of my group inside the wave
    const unsigned valid = (static_cast<unsigned>(__ballot(hit[0]) >> row) & 0xffu) |
                           ((static_cast<unsigned>(__ballot(hit[1]) >> row) & 0xffu) << 8) |
                           ((static_cast<unsigned>(__ballot(hit[2]) >> row) & 0xffu) << 16);
    const int cnt = __builtin_popcount(valid);
    if (cnt < 3) return 0.0f;   // group-uniform: fewer than 3 points, the fan sums nothing but zeros
    if (stop == 5) return 1.0f;   // diagnostics only
#pragma unroll
    for (int p = 0; p < 3; ++p)
        if (hit[p]) { cs.px[sub + 8 * p] = x[p].x; cs.py[sub + 8 * p] = x[p].y; }
    group_sync();
    // ---- centroid: the same sequence of additions as the reference ----
    Pt ctr = { 0.f, 0.f };
    for (unsigned m = valid; m; m &= m - 1u) {
        const int sl = __builtin_ctz(m);
        ctr.x = ctr.x + cs.px[sl]; ctr.y = ctr.y + cs.py[sl];
    }
    ctr.x /= cnt;
    ctr.y /= cnt;
    if (stop == 6) return ctr.x;
    // ---- angles (point_cmp :98-100 evaluates atan2 of the point minus the centre) ----
    // The valid points (6 of the 24 slots on average) are dealt out again, one per lane: lane s takes the s-th valid
    // slot (and the (s+8)-th ... when there are more than eight), so a wave runs atan2f once, not once per slot a lane owns.
    auto nth_valid = [&](int p) -> int {
        unsigned m = valid;
        for (int k = 0; k < p; ++k) m &= m - 1u;
        return __builtin_ctz(m);
    };
    for (int p = sub; p < cnt; p += kClipLanes) {
        const int sl = nth_valid(p);
        cs.ang[sl] = atan2f(cs.py[sl] - ctr.y, cs.px[sl] - ctr.x);
    }
    group_sync();
    if (stop == 7) return cs.ang[0];
    // ---- position of my point(s) after the reference's stable sort ----
    for (int p = sub; p < cnt; p += kClipLanes) {
        const int sl = nth_valid(p);
        const float am = cs.ang[sl];
        int rk = 0;
        for (unsigned m = valid; m; m &= m - 1u) {
            const int o = __builtin_ctz(m);
            const float aj = cs.ang[o];
            rk += (aj < am || (aj == am && o < sl)) ? 1 : 0;
        }
        cs.sorted[rk] = sl;
    }
    group_sync();
    // ---- fan terms, then their sum in rank order ----
    {
        const int s0 = cs.sorted[0];
        const float x0 = cs.px[s0], y0 = cs.py[s0];
        for (int r = sub + 1; r <= cnt - 2; r += kClipLanes) {
            const int sk = cs.sorted[r], sn = cs.sorted[r + 1];
            const float ux = cs.px[sk] - x0, uy = cs.py[sk] - y0;
            const float vx = cs.px[sn] - x0, vy = cs.py[sn] - y0;
            cs.term[r] = ux * vy - uy * vx;
        }
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

template <int ROWS, int THREADS>
struct TileSharedT {
    BoxPre ra[ROWS], cb[64];
    union {   // the first queue is dead once the second one is complete (a barrier later the clip starts)
        unsigned short queue[ROWS * 64];           // pairs that passed the bounding-circle filter: row << 6 | col
        ClipScratch clip[THREADS / kClipLanes];    // one per group of lanes
    };
    unsigned short queue2[ROWS * 64];              // ... and the separating-axis filter
    unsigned long long words[64];                  // NMS: the mask words of the tile's rows
    unsigned long long apart[ROWS];                // IoU: bit 16 i + k of word r = pair (r, 4 k + i) is an exact zero by the circle filter
    unsigned long long late[ROWS];                 // IoU: the same for the zeros of the separating-axis filter
    int qcount, q2count, arrived;
};
#ifndef HF_NMS_THREADS
#define HF_NMS_THREADS 512   // 256 / 384 / 512 / 768 threads: 9000 clustered boxes 317 / 281 / 257 / 267 us, uniform 180 / 190 / 183 / 231 us
#endif
typedef TileSharedT<64, HF_NMS_THREADS> TileShared;   // the square tiles of the NMS mask
// compute_bev_iou: 96 x 64 tiles, 512 threads (46 KB of LDS, <= 64 VGPRs: three workgroups per CU).  70 000 rows are 730
// workgroups = ONE round of the 768 resident slots (64-row tiles were 1094 workgroups, and the 70 of the second round
// doubled the kernel's duration), and 96 rows are exactly three steps of the circle filter (measured in one run,
// scripts/probes/bev_variant.py: 64 / 72 / 80 / 96 / 104 / 112 / 128 rows = 15.6 / 15.5 / 15.6 / 14.0 / 15.1 / 15.7 / 19.2 us).  Waves 0..5 clip (48 groups of eight lanes: one pass for almost every tile), waves 6..7 write the zeros
// (measured in one run, scripts/probes/bev_variant.py: 1 / 2 / 3 / 4 storing waves = 18.8 / 15.7 / 17.1 / 17.7 us).
#ifndef HF_IOU_ROWS
#define HF_IOU_ROWS 96
#endif
constexpr int kIouRows = HF_IOU_ROWS;
#ifndef HF_IOU_THREADS
#define HF_IOU_THREADS 512
#endif
constexpr int kIouThreads = HF_IOU_THREADS;
#ifndef HF_IOU_STORE_WAVES
#define HF_IOU_STORE_WAVES 2
#endif
constexpr int kIouStoreWaves = HF_IOU_STORE_WAVES;   // a wave keeps only a handful of stores in flight: one storing wave per
                                                      // workgroup drains the zeros at 4 TB/s, two at 5
constexpr int kIouClipThreads = kIouThreads - 64 * kIouStoreWaves;
typedef TileSharedT<kIouRows, kIouThreads> IouShared;

template <int ROWS, int THREADS>
__device__ __forceinline__ void tile_stage(TileSharedT<ROWS, THREADS> &sh, const float *boxes_r, int row0, int row_size,
                                           const float *boxes_c, int col0, int col_size)
{
    const int t = threadIdx.x;
    // threads 0..63: the columns; threads 64..: the rows
    static_assert(ROWS + 64 <= THREADS, "one thread per box");
    if (t < 64) {
        if (t < col_size) box_precompute(boxes_c + static_cast<size_t>(col0 + t) * 5, sh.cb[t]);
        sh.words[t] = 0ull;
    } else if (t - 64 < ROWS) {
        if (t - 64 < row_size) box_precompute(boxes_r + static_cast<size_t>(row0 + t - 64) * 5, sh.ra[t - 64]);
        sh.apart[t - 64] = 0ull;
        sh.late[t - 64] = 0ull;
    }
    if (t == THREADS - 1) { sh.qcount = 0; sh.q2count = 0; sh.arrived = 0; }
}

// IoU matrix: one workgroup per 96 x 64 tile of (a, b) pairs.
// 99 % of the pairs are exact zeros by the first filter, and writing them (8 bytes per pair, both outputs) is the floor of
// the kernel: ~6 us of store drain at 70 000 x 64.  A wave that issues those stores is held by the back-pressure of the
// write path for that long -- so ONE wave (the last) issues them all, after the filters, while the other five clip the
// ~30 pairs of the tile that do overlap.  (When every wave stored its share before the clip, the clip started after the
// drain: 18 us.)
__global__ __launch_bounds__(kIouThreads, 8) void bev_iou_kernel(int num_a, const float *__restrict__ boxes_a, int num_b,
                                                              const float *__restrict__ boxes_b,
                                                              float *__restrict__ ans_overlap,
                                                              float *__restrict__ ans_iou, int stop)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    IouShared &sh = *reinterpret_cast<IouShared *>(smem_raw);
    const int t = threadIdx.x;
    const int row0 = blockIdx.y * kIouRows, col0 = blockIdx.x * 64;
    const int row_size = min(num_a - row0, kIouRows), col_size = min(num_b - col0, 64);
    if (stop == 1) return;   // diagnostics only (HF_BEV_STOP): outputs invalid
    tile_stage(sh, boxes_a, row0, row_size, boxes_b, col0, col_size);
    __syncthreads();
    if (stop == 2) return;
    // filter 1 (all pairs, cheap): bounding circles.  A thread owns FOUR FIXED columns (their centre / radius / magnitude
    // stay in registers) and walks the rows 32 apart: a quarter of a 16-byte LDS read and a handful of packed instructions per pair.
    // Exact zeros are recorded in the tile's bitmap, the rest goes to the queue.
    {
        const int c0 = (t & 15) * 4;
        // reach = (a.rad + 1e-5 a.mag) + (b.rad + 1e-3 + 1e-5 b.mag): the column half is folded once per thread, the row
        // half once per row; pairs go two at a time through the packed fp32 instructions
        typedef float f2v __attribute__((ext_vector_type(2)));
        f2v bx[2], by[2], bs[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 q = *reinterpret_cast<const float4 *>(&sh.cb[min(c0 + i, 63)]);
            bx[i >> 1][i & 1] = q.x; by[i >> 1][i & 1] = q.y; bs[i >> 1][i & 1] = q.z + 1e-3f + 1e-5f * q.w;
        }
        // a wave covers four rows per step; the trip count is wave-uniform (the ballots below need every lane).  Full
        // tiles (all of them when the matrix is a multiple of 80 x 64) run without the per-row / per-column validity masks.
        auto filter_rows = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            static_assert((kIouRows + kIouThreads / 16 - 1) / (kIouThreads / 16) <= 8, "four bits per row step in one register");
            unsigned surv_all = 0u;   // my pairs that go on to the next filter: four bits per row step
            int iter = 0;
            for (int rb = (t >> 6) * 4; rb < row_size; rb += kIouThreads / 16) {
                const int r = rb + ((t >> 4) & 3);
                const bool live = FULL || (r < row_size && c0 < col_size);
                const float4 a = *reinterpret_cast<const float4 *>(&sh.ra[live ? r : 0]);   // cx, cy, rad, mag
                const float as = a.z + 1e-5f * a.w;
                unsigned m16[4];       // bit k = pair (r, 4 k + i) is far apart: the ballot bits of the 16 lanes of my row
                unsigned surv = 0u;    // my pairs that go on to the next filter
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // = circles_apart(): centres further apart than the radius bounds plus a slack that dwarfs MARGIN and
                    // fp32 rounding; NaN / inf compare false -> next filter
                    const f2v reach = bs[h] + as;
                    const f2v dx = bx[h] - a.x, dy = by[h] - a.y;
                    const f2v d2 = __builtin_elementwise_fma(dx, dx, dy * dy);
                    const f2v r2 = reach * reach;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int i = 2 * h + u;
                        const bool in = FULL || (live && c0 + i < col_size);
                        const bool far = in && d2[u] > r2[u];
                        if (in && !far) surv |= 1u << i;
                        const unsigned long long bal = __ballot(far);
                        const unsigned half = (t & 32) ? static_cast<unsigned>(bal >> 32) : static_cast<unsigned>(bal);
                        m16[i] = (half >> (t & 16)) & 0xffffu;
                    }
                }
                surv_all |= surv << (4 * iter);
                ++iter;
                if (live && (t & 15) == 0)
                    sh.apart[r] = (static_cast<unsigned long long>(m16[2] | (m16[3] << 16)) << 32) | (m16[0] | (m16[1] << 16));
            }
            // the survivors of all my rows in one go (about one pair in a hundred): one trip to the queue counter per wave
            for (; surv_all; surv_all &= surv_all - 1u) {
                const int b = __builtin_ctz(surv_all);
                const int r = (t >> 6) * 4 + (kIouThreads / 16) * (b >> 2) + ((t >> 4) & 3);
                sh.queue[atomicAdd(&sh.qcount, 1)] = static_cast<unsigned short>((r << 6) | (c0 + (b & 3)));
            }
        };
        if (row_size == kIouRows && col_size == 64) filter_rows(std::true_type{});
        else filter_rows(std::false_type{});
    }
    __syncthreads();   // the only barrier after the staging: from here on the storing wave and the clipping waves part
    if (stop == 3) return;
    if (t >= kIouClipThreads) {
        // the zeros of filter 1: lane -> (one of four rows, four columns); a full nibble leaves as one 16-byte store per
        // output (rows of the outputs are 16-byte aligned when num_b is a multiple of 4; the launcher checks the bases)
        if (stop == 8) return;   // diagnostics only: no zeros written
        const bool vec = (num_b & 3) == 0;
        const int l = t & 63, k = l & 15, c0 = k * 4, sw = (t - kIouClipThreads) >> 6;
        constexpr int kStep = 4 * kIouStoreWaves;   // rows between two visits of a lane
        // wait for the separating-axis filter of the clipping waves (well under a microsecond): its zeros then leave with
        // the rest, in full 16-byte stores where the whole nibble is zero
        while (__hip_atomic_load(&sh.arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < kIouClipThreads / 64) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        if (c0 < col_size) {
            constexpr int kBatch = 5;   // rows per lane and batch: the bitmap words first, then the stores back to back
            for (int rb = (l >> 4) + 4 * sw; rb < row_size; rb += kStep * kBatch) {
                unsigned long long w[kBatch];
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    const int r = min(rb + kStep * j, kIouRows - 1);
                    w[j] = (sh.apart[r] | sh.late[r]) >> k;
                }
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    const int r = rb + kStep * j;
                    if (r >= row_size) break;
                    const unsigned nib = static_cast<unsigned>(w[j] & 1ull) | (static_cast<unsigned>(w[j] >> 15) & 2u) |
                                         (static_cast<unsigned>(w[j] >> 30) & 4u) | (static_cast<unsigned>(w[j] >> 45) & 8u);
                    const size_t o = static_cast<size_t>(row0 + r) * num_b + col0 + c0;
                    if (nib == 15u && vec) {
                        typedef float f4v __attribute__((ext_vector_type(4)));
                        const f4v z = { 0.f, 0.f, 0.f, 0.f };
                        if (ans_overlap) __builtin_nontemporal_store(z, reinterpret_cast<f4v *>(ans_overlap + o));
                        if (ans_iou) __builtin_nontemporal_store(z, reinterpret_cast<f4v *>(ans_iou + o));
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if ((nib >> i) & 1u) {
                                if (ans_overlap) ans_overlap[o + i] = 0.0f;
                                if (ans_iou) ans_iou[o + i] = 0.0f;
                            }
                    }
                }
            }
        }
        return;
    }
    // filter 2 (survivors only, dense lanes, the clipping waves): separating axes -> exact zeros or the second queue
    const int nq = sh.qcount;
    for (int q = t; q < nq; q += kIouClipThreads) {
        const int e = sh.queue[q];
        const int r = e >> 6, c = e & 63;
        if (surely_disjoint(sh.ra[r], sh.cb[c])) atomicOr(&sh.late[r], 1ull << (16 * (c & 3) + (c >> 2)));   // written by the storing waves
        else sh.queue2[atomicAdd(&sh.q2count, 1)] = static_cast<unsigned short>(e);
    }
    // the clipping waves meet here (the sixth is busy storing and must not be waited for): an arrival counter in LDS.
    // All five are resident and all five get here, so the wait is bounded.
    // Only LDS traffic has to be ordered: a workgroup-scope fence would also wait for this wave's global stores (the
    // zeros of filter 2), and those sit behind the storing waves' megabytes in the write path.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((t & 63) == 0) __hip_atomic_fetch_add(&sh.arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(&sh.arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < kIouClipThreads / 64) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    if (stop == 4) return;
    // the clip on what is left: eight lanes per pair, 40 pairs per pass.  The results of the first two passes wait in
    // registers until the group has clipped its last pair: their stores, too, could stall the wave while the zeros drain.
    const int nq2 = sh.q2count;
    const int grp = t / kClipLanes, sub = t % kClipLanes;
    int held_e[2] = { -1, -1 };
    float held_s[2] = { 0.f, 0.f };
    auto write_pair = [&](int e, float sv) {
        const int r = e >> 6, c = e & 63;
        const size_t o = static_cast<size_t>(row0 + r) * num_b + col0 + c;
        if (ans_overlap) ans_overlap[o] = sv;
        if (ans_iou) ans_iou[o] = iou_from_overlap(sh.ra[r].box, sh.cb[c].box, sv);
    };
    int pass = 0;
    for (int q0 = 0; q0 < nq2; q0 += kIouClipThreads / kClipLanes, ++pass) {
        const int q = q0 + grp;
        if (q >= nq2) break;   // whole groups leave together
        const int e = sh.queue2[q];
        const float sv = box_overlap_group(sh.ra[e >> 6], sh.cb[e & 63], sh.clip[grp], sub, stop);
        if (sub == 0) {
            if (pass == 0) { held_e[0] = e; held_s[0] = sv; }
            else if (pass == 1) { held_e[1] = e; held_s[1] = sv; }
            else write_pair(e, sv);
        }
    }
    if (sub == 0) {
        if (held_e[0] >= 0) write_pair(held_e[0], held_s[0]);
        if (held_e[1] >= 0) write_pair(held_e[1], held_s[1]);
    }
}

// ---------------------------------------------------------------- NMS mask
// one block per (row tile, col tile) of 64x64 pairs; mask word (row, col tile) bit j set iff
// iou(row, col*64+j) > thresh, j > row inside the diagonal tile (bev_iou_g.cu:256-298).
constexpr int kNmsThreads = HF_NMS_THREADS;

// An entry of a column block's list: a nonzero mask word right of the diagonal and the row it belongs to.
struct __attribute__((aligned(16))) NmsEntry {
    unsigned long long word;
    int row, pad;
};
constexpr int kNmsListCap = 2048;   // entries per column block (a column that overflows is swept from the dense mask)

// per-frame workspace of hf_oriented_nms:
// [mask n*cb words][counts cb ints, padded][lists cb * kNmsListCap entries][transposed diagonal words, n][BoxPre table, n]
// [next words, n: mask word (row, row block + 1)]
__host__ __device__ inline size_t nms_ws_counts_offset(int n) { return (sizeof(unsigned long long) * static_cast<size_t>(n) * ((n + 63) / 64) + 255) & ~static_cast<size_t>(255); }
__host__ __device__ inline size_t nms_ws_lists_offset(int n) { return nms_ws_counts_offset(n) + ((sizeof(int) * static_cast<size_t>((n + 63) / 64) + 255) & ~static_cast<size_t>(255)); }
__host__ __device__ inline size_t nms_ws_diagt_offset(int n) { return nms_ws_lists_offset(n) + sizeof(NmsEntry) * static_cast<size_t>((n + 63) / 64) * kNmsListCap; }
__host__ __device__ inline size_t nms_ws_boxpre_offset(int n) { return nms_ws_diagt_offset(n) + ((sizeof(unsigned long long) * static_cast<size_t>(n) + 255) & ~static_cast<size_t>(255)); }
__host__ __device__ inline size_t nms_ws_nextw_offset(int n) { return nms_ws_boxpre_offset(n) + ((sizeof(BoxPre) * static_cast<size_t>(n) + 255) & ~static_cast<size_t>(255)); }
__host__ __device__ inline size_t nms_ws_bytes(int n) { return nms_ws_nextw_offset(n) + ((sizeof(unsigned long long) * static_cast<size_t>(n) + 255) & ~static_cast<size_t>(255)); }

// UPPER_ONLY = the form hf_oriented_nms launches: `mask` is the frame-0 workspace (stride ws_stride bytes per frame),
// tiles below the diagonal are skipped, and every nonzero word right of the diagonal is also appended to the list of
// its column block.
// the per-box part of the tile kernels once per box (hf_oriented_nms: 10 011 tiles at 9000 boxes would each redo the
// cos / sin / corners of their 128 boxes)
__global__ __launch_bounds__(256) void nms_boxpre_kernel(int n, const float *__restrict__ boxes, unsigned char *__restrict__ ws,
                                                         size_t ws_stride)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    boxes += static_cast<size_t>(blockIdx.y) * n * 5;
    BoxPre *table = reinterpret_cast<BoxPre *>(ws + static_cast<size_t>(blockIdx.y) * ws_stride + nms_ws_boxpre_offset(n));
    BoxPre p;
    box_precompute(boxes + static_cast<size_t>(i) * 5, p);
    p.pad_ = 0.f;
    table[i] = p;
}

template <bool UPPER_ONLY>
__global__ __launch_bounds__(kNmsThreads, HF_NMS_THREADS >= 512 ? 8 : 4) void nms_mask_kernel(int n, float thresh, const float *__restrict__ boxes,
                                                               unsigned long long *__restrict__ mask, size_t ws_stride)
{
    const int row_t = blockIdx.y, col_t = blockIdx.x;
    if (UPPER_ONLY && col_t < row_t) return;  // never read by the sweep (bev_iou.cpp:100-103 starts at nblock)
    // blockIdx.z = frame of a batched call: every frame has its own (n,5) boxes and workspace / (n, ceil(n/64)) mask
    boxes += static_cast<size_t>(blockIdx.z) * n * 5;
    unsigned char *ws = reinterpret_cast<unsigned char *>(mask) + static_cast<size_t>(blockIdx.z) * ws_stride;
    mask = reinterpret_cast<unsigned long long *>(ws);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TileShared &sh = *reinterpret_cast<TileShared *>(smem_raw);
    const int t = threadIdx.x;
    const int row_size = min(n - row_t * 64, 64), col_size = min(n - col_t * 64, 64);
    if (UPPER_ONLY) {
        // the boxes' precomputed parts come from the table (80 bytes per box, copied 16 bytes per thread and step)
        const float4 *table = reinterpret_cast<const float4 *>(ws + nms_ws_boxpre_offset(n));
        constexpr int kVec = sizeof(BoxPre) / 16;
        float4 *dst_c = reinterpret_cast<float4 *>(sh.cb), *dst_r = reinterpret_cast<float4 *>(sh.ra);
        for (int i = t; i < col_size * kVec; i += kNmsThreads) dst_c[i] = table[static_cast<size_t>(col_t) * 64 * kVec + i];
        for (int i = t; i < row_size * kVec; i += kNmsThreads) dst_r[i] = table[static_cast<size_t>(row_t) * 64 * kVec + i];
        if (t < 64) sh.words[t] = 0ull;
        if (t == kNmsThreads - 1) { sh.qcount = 0; sh.q2count = 0; sh.arrived = 0; }
    } else {
        tile_stage(sh, boxes, row_t * 64, row_size, boxes, col_t * 64, col_size);
    }
    __syncthreads();
#if defined(HF_NMS_PHASE_STOP)
    if (HF_NMS_PHASE_STOP == 1) return;   // diagnostic variant builds only (scripts/probes/build_bev_variant.sh): phase costs
#endif
    // filter 1: bounding circles, four fixed columns per thread in registers, rows kNmsThreads / 16 apart; pairs go two at a time
    // through the packed fp32 instructions (as in bev_iou_kernel), and a thread's survivors are pushed in one go at the end -- the
    // per-pair form ran eight exec-masked branch regions with an LDS atomic each per thread (~130 instructions; this: ~80)
    {
        const int c0 = (t & 15) * 4;
        typedef float f2v __attribute__((ext_vector_type(2)));
        f2v bx[2], by[2], bs[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 q = *reinterpret_cast<const float4 *>(&sh.cb[min(c0 + i, 63)]);
            bx[i >> 1][i & 1] = q.x; by[i >> 1][i & 1] = q.y; bs[i >> 1][i & 1] = q.z + 1e-3f + 1e-5f * q.w;
        }
        constexpr int kRowStep = kNmsThreads / 16;
        static_assert(64 / kRowStep <= 8, "four bits per row step in one register");
        unsigned surv = 0u;   // my pairs that go on: bit 4 * step + column
#pragma unroll
        for (int it = 0; it < 64 / kRowStep; ++it) {
            const int r = (t >> 4) + it * kRowStep;
            const bool live = r < row_size && c0 < col_size;
            const float4 a = *reinterpret_cast<const float4 *>(&sh.ra[live ? r : 0]);   // cx, cy, rad, mag
            const float as = a.z + 1e-5f * a.w;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // = circles_apart(): NaN / inf compare false -> next filter
                const f2v reach = bs[h] + as;
                const f2v dx = bx[h] - a.x, dy = by[h] - a.y;
                const f2v d2 = __builtin_elementwise_fma(dx, dx, dy * dy);
                const f2v r2 = reach * reach;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int c = c0 + 2 * h + u;
                    const bool valid = live && c < col_size && !(row_t == col_t && c <= r);
                    if (valid && !(d2[u] > r2[u])) surv |= 1u << (4 * it + 2 * h + u);
                }
            }
        }
        for (; surv; surv &= surv - 1u) {
            const int b = __builtin_ctz(surv);
            const int r = (t >> 4) + (b >> 2) * kRowStep;
            sh.queue[atomicAdd(&sh.qcount, 1)] = static_cast<unsigned short>((r << 6) | (c0 + (b & 3)));
        }
    }
    __syncthreads();
#if defined(HF_NMS_PHASE_STOP)
    if (HF_NMS_PHASE_STOP == 2) return;
#endif
    const int nq = sh.qcount;
    for (int q = t; q < nq; q += kNmsThreads) {
        const int e = sh.queue[q];
        if (!surely_disjoint(sh.ra[e >> 6], sh.cb[e & 63]) && !iou_surely_below(sh.ra[e >> 6], sh.cb[e & 63], thresh))
            sh.queue2[atomicAdd(&sh.q2count, 1)] = static_cast<unsigned short>(e);
    }
    __syncthreads();
#if defined(HF_NMS_PHASE_STOP)
    if (HF_NMS_PHASE_STOP == 3) return;
#endif
    {
#if defined(HF_NMS_PHASE_STOP)
        const int nq2 = HF_NMS_PHASE_STOP == 4 ? 0 : sh.q2count;   // 4: no clip at all
#else
        const int nq2 = sh.q2count;
#endif
        const int grp = t / kClipLanes, sub = t % kClipLanes;
        for (int q0 = 0; q0 < nq2; q0 += kNmsThreads / kClipLanes) {
            const int q = q0 + grp;
            if (q >= nq2) continue;   // whole groups leave together
            const int e = sh.queue2[q];
            const int r = e >> 6, c = e & 63;
            const float s = box_overlap_group(sh.ra[r], sh.cb[c], sh.clip[grp], sub);
            if (sub == 0 && iou_from_overlap(sh.ra[r].box, sh.cb[c].box, s) > thresh) atomicOr(&sh.words[r], 1ull << c);
        }
    }
    __syncthreads();
#if defined(HF_NMS_PHASE_STOP)
    if (HF_NMS_PHASE_STOP == 5) return;   // everything but the stores
#endif
    const int col_blocks = (n + 63) / 64;
    if (t < row_size) {
        const unsigned long long w = sh.words[t];
        // hf_oriented_nms clears the dense mask with one memset and its tiles store their NONZERO words only (64 eight-byte stores
        // to 64 different lines per tile: 640 000 partial-line writes at 9000 boxes, nearly all of them zeros)
        if (!UPPER_ONLY || w != 0ull) mask[(static_cast<size_t>(row_t) * 64 + t) * col_blocks + col_t] = w;
        if (UPPER_ONLY && col_t == row_t + 1)   // the word next to the diagonal: the sweep's resolving wave adds these itself
            reinterpret_cast<unsigned long long *>(ws + nms_ws_nextw_offset(n))[row_t * 64 + t] = w;
        if (UPPER_ONLY && col_t > row_t + 1 && w != 0ull) {
            int *counts = reinterpret_cast<int *>(ws + nms_ws_counts_offset(n));
            NmsEntry *lists = reinterpret_cast<NmsEntry *>(ws + nms_ws_lists_offset(n));
            const int slot = atomicAdd(&counts[col_t], 1);
            if (slot < kNmsListCap) lists[static_cast<size_t>(col_t) * kNmsListCap + slot] = NmsEntry{ w, row_t * 64 + t, 0 };
        }
    }
    if (UPPER_ONLY && col_t == row_t && t < col_size) {
        // the diagonal tile, transposed: bit r of word c = box row_t*64+r suppresses box row_t*64+c ("who kills me"),
        // the form the sweep resolves a block with
        unsigned long long tw = 0ull;
        for (int r = 0; r < 64; ++r) tw |= ((sh.words[r] >> t) & 1ull) << r;
        reinterpret_cast<unsigned long long *>(ws + nms_ws_diagt_offset(n))[row_t * 64 + t] = tw;
    }
}

// ---------------------------------------------------------------- greedy sweep on the device
// bev_iou.cpp:87-112 without the host.  One 1024-thread workgroup per frame walks the column blocks in order; the kept
// bits of every block live in LDS.  The dense mask is 8 n^2/64 bytes (10 MB at 9000 boxes) of which almost every word is
// zero, and ONE compute unit cannot stream it (measured: ~17 GB/s -> 0.25 ms for the upper triangle however deep the
// prefetch).  So the mask kernel also files every nonzero word two or more blocks right of the diagonal under its
// column block, writes the words NEXT to the diagonal to an array of their own and the diagonal tile transposed.
// Step blk, one barrier per step:
//   gather   waves 1..15: the entries of column block blk+1 (requested kSweepLead steps earlier, they depend on no
//            decision); their rows lie in blocks <= blk-1, all decided: an entry whose row was kept adds its word to the
//            block's removal word (an LDS atomic) -- the host loop's `remv[j] |= p[j]`, transposed;
//   resolve  wave 0, meanwhile: the boxes kept in block blk-1 add their next words to block blk's removal word; then
//            lane l holds the TRANSPOSED diagonal word of box blk*64+l (which boxes of the block suppress it).
//            kept = alive and no kept suppressor: iterated for the whole wave at once until nothing changes (one ballot
//            per round, a round per link of the longest suppression chain) -- the host loop's
//            `if (!(remv[nblock] & 1 << inblock))` without its 64 dependent steps.
// A column block whose list overflowed (> kNmsListCap nonzero words: thousands of boxes overlapping the same 64) is
// gathered from the dense mask instead: every kept row so far is asked for its word of that column.
// (Round 1 PULLED like that for every block: 0.95 ms of 1.14 ms.  Pushing kept rows into all later words, even with the
// loads issued four steps ahead of the decision, stayed at 0.25 ms: the single CU's memory parallelism.)
constexpr int kSweepThreads = 1024;
constexpr int kSweepGatherers = kSweepThreads - 64;
constexpr int kSweepLead = 4;   // steps between the request of a column's first entries and their use

// The streaming sweep.  The barrier form above pays one 1024-thread barrier plus ~5 dependent LDS round trips per column block
// (0.55 us x 141 blocks at 9000 boxes) although what a block needs from memory -- the entries filed under its column, its
// diagonal and next words -- depends on NO decision.  Here waves 1..15 only MOVE list entries from memory into an LDS ring, as
// far ahead as the ring allows (a flag per column says "complete"), and wave 0 alone walks the blocks: for block b it ORs the
// words of the ring entries whose row was kept (rows <= b-2: all decided), adds the next words of the boxes kept in block b-1,
// resolves the block by the whole-wave fixed point, publishes how far the ring has been consumed.  No workgroup barrier inside
// the walk; every wait is on an LDS word that another wave is bound to write (producers fill columns in increasing order, space
// in the ring is granted in column order, the consumer takes columns in order), so every wave reaches the end.
constexpr int kStreamMaxBlocks = 1024;      // 65 536 boxes: keptw + cnts + pref + two flag arrays = 24 KB
constexpr int kStreamRing = 4096;           // list entries in the ring (64 KB, a power of two): two full columns
constexpr int kStreamWords = 64;            // blocks whose (diagonal, next) word pairs sit in LDS ahead of wave 0 (64 KB)
constexpr int kStreamWordWaves = 8;         // waves 1..8 stream those words, waves 9..15 the list entries
constexpr int kStreamMaxAverage = 32;       // entries per column block on average up to which one consuming wave keeps up (measured:
                                            // 9000 clustered boxes at threshold 0.01 hold 86 per column and run 260 us streaming against 222;
                                            // at the RPN's 0.8 the lists of uniform and clustered boxes alike hold ~20 entries IN TOTAL)

__device__ __forceinline__ int lds_load_i32(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Wave 0 issues NO global load: a first version prefetched its diagonal / next words into registers eight blocks ahead, and the
// compiler's wait-count insertion (loops with data-dependent trip counts around the uses) put an `s_waitcnt vmcnt(0)` into every
// step -- each block then paid a full memory round trip behind its own `keep[]` store (0.39 us per block, empty lists or not).
// Waves 1..8 stream those word pairs into LDS instead, waves 9..15 the list entries.
__device__ __forceinline__ void sweep_stream(int n, int cb, const int *__restrict__ counts, const NmsEntry *__restrict__ lists,
                             const unsigned long long *__restrict__ diagt, const unsigned long long *__restrict__ nextw,
                             int *__restrict__ keep, unsigned char *smem_raw, int *kept_total, int *consumed_pref, int *scan_ws)
{
    typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
    // the 16-byte arrays first (smem_raw is 16-byte aligned): no integer arithmetic on the pointers -- a round trip through
    // uintptr_t made the compiler forget that these are LDS addresses and read them with FLAT loads (hundreds of cycles each)
    ull2 *words = reinterpret_cast<ull2 *>(smem_raw);                                        // kStreamWords * 64
    NmsEntry *ring = reinterpret_cast<NmsEntry *>(words + kStreamWords * 64);                // kStreamRing
    unsigned long long *keptw = reinterpret_cast<unsigned long long *>(ring + kStreamRing);  // cb   kept bits per block
    int *cnts = reinterpret_cast<int *>(keptw + cb);                                         // cb   entries filed under the block
    int *pref = cnts + cb;                                                                   // cb + 1   entries before the block
    int *ready = pref + cb + 1;                                                              // cb   the block's entries are in the ring
    int *wready = ready + cb;                                                                // cb   the block's word pairs are in LDS
    int *consumed_blk = scan_ws + 30;                                                        // blocks wave 0 is done with
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // exclusive prefix of the counts (cb <= 1024 = one value per thread)
    {
        const int c = t < cb ? counts[t] : 0;
        int total = 0;
        const int ex = block_exclusive_scan(c, scan_ws, &total);
        if (t < cb) { cnts[t] = c; pref[t] = ex; ready[t] = 0; wready[t] = 0; keptw[t] = 0ull; }
        if (t == 0) { pref[cb] = total; *kept_total = 0; *consumed_pref = 0; *consumed_blk = 0; }
    }
    __syncthreads();
    if (wave >= 1 && wave <= kStreamWordWaves) {
        // ---------------- the (diagonal, next) words of every block: a wave moves four blocks per memory round trip, eight waves
        // take turns (one wave alone fed wave 0 a block per 0.37 us -- exactly what the walk then took) ----------------
        for (int b0 = 4 * (wave - 1); b0 < cb; b0 += 4 * kStreamWordWaves) {
            while (b0 + 4 - lds_load_i32(consumed_blk) > kStreamWords) __builtin_amdgcn_s_sleep(2);
            asm volatile("" ::: "memory");
            unsigned long long d[4], x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int blk = b0 + j, i = blk * 64 + lane;
                d[j] = (blk < cb && i < n) ? diagt[i] : 0ull;
                x[j] = (blk >= 1 && blk < cb) ? nextw[i - 64] : 0ull;   // boxes of block blk-1 all exist
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (b0 + j < cb) words[((b0 + j) & (kStreamWords - 1)) * 64 + lane] = ull2{ d[j], x[j] };
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < 4 && b0 + lane < cb) __hip_atomic_store(&wready[b0 + lane], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    if (wave > kStreamWordWaves) {
        // ---------------- list entries: the remaining waves, a column each, in increasing order ----------------
        for (int col = wave - 1 - kStreamWordWaves; col < cb; col += kSweepThreads / 64 - 1 - kStreamWordWaves) {
            const int cnt = cnts[col];
            if (cnt == 0) continue;   // the consumer does not wait for an empty column
            const int base = pref[col];
            while (base + cnt - lds_load_i32(consumed_pref) > kStreamRing) __builtin_amdgcn_s_sleep(2);
            asm volatile("" ::: "memory");
            const NmsEntry *src = lists + static_cast<size_t>(col) * kNmsListCap;
            typedef unsigned u4v __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<NmsEntry *>(src), 0, cnt * 16, 0x00020000);
            for (int k0 = 0; k0 < cnt; k0 += 64 * 4) {   // four 16-byte loads in flight per lane (past the end: zeros, not stored)
                u4v e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (k0 + 64 * j + lane) * 16, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + 64 * j + lane;
                    if (k < cnt) *reinterpret_cast<u4v *>(&ring[(base + k) & (kStreamRing - 1)]) = e[j];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's ring writes are done (the LDS keeps a wave's order)
            if (lane == 0) __hip_atomic_store(&ready[col], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    // ---------------- consumer: wave 0.  Per block: word pair (LDS) -> [ring entries -> kept bits of their rows] -> ballots ----------------
    __builtin_amdgcn_s_setprio(3);
    int total_kept = 0;
    unsigned long long kb_prev = 0ull;   // the kept bits of the block before (wave-uniform)
    // One wave runs ~100 instructions per block back to back, so every LDS round trip inside the block's dependent chain shows
    // (0.37 us per block with four of them).  What the NEXT block needs from LDS and depends on no decision -- its flag, word
    // pair, entry count and ring offset -- is therefore requested at the top of the current block and has arrived when it is
    // used.  The flag is read BEFORE the words (the LDS keeps a wave's order, the producer wrote the words before the flag):
    // a set flag vouches for the words read after it; a clear one means wait and read again.
    // (a taken branch costs a lone wave more than a handful of instructions: the walk is written branch-lean -- the peek past the
    // last block re-reads the last block, uniform values are stored by every lane, the keep[] store is a range-checked buffer
    // store whose unkept lanes aim past the end)
    auto peek = [&](int blk, int &flag, ull2 &w, int &cnt, int &base) {
        const int bk = min(blk, cb - 1);
        flag = lds_load_i32(&wready[bk]);
        w = words[(bk & (kStreamWords - 1)) * 64 + lane];
        cnt = cnts[bk];
        base = pref[bk];
    };
    const __amdgpu_buffer_rsrc_t rkeep = __builtin_amdgcn_make_buffer_rsrc(keep, 0, n * 4, 0x00020000);
    int flag_n, cnt_n, base_n;
    ull2 w_n;
    peek(0, flag_n, w_n, cnt_n, base_n);
    for (int blk = 0; blk < cb; ++blk) {
        int flag = flag_n;
        ull2 w = w_n;
        const int cnt = cnt_n, base = base_n;
        if (flag == 0) {   // uniform: the producers are behind (start of the kernel)
            while (lds_load_i32(&wready[blk]) == 0) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            w = words[(blk & (kStreamWords - 1)) * 64 + lane];
        }
        peek(blk + 1, flag_n, w_n, cnt_n, base_n);
        const int lim = min(64, n - blk * 64);
        const unsigned long long valid = lim == 64 ? ~0ull : ((1ull << lim) - 1ull);
        // (1) the words of the column's entries whose row was kept
        unsigned long long acc = 0ull;
        if (cnt > 0) {
            while (lds_load_i32(&ready[blk]) == 0) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            for (int k = lane; k < cnt; k += 64) {
                const NmsEntry e = ring[(base + k) & (kStreamRing - 1)];
                if ((keptw[e.row >> 6] >> (e.row & 63)) & 1ull) acc |= e.word;
            }
        }
        // (2) the next words of the boxes kept in block blk-1
        if ((kb_prev >> lane) & 1ull) acc |= w.y;
        // OR over the wave: few lanes hold anything
        unsigned long long remv = 0ull;
        for (unsigned long long m = __ballot(acc != 0ull); m; m &= m - 1ull) {
            const int src = __builtin_ctzll(m);
            const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<unsigned>(acc)), src));
            const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<unsigned>(acc >> 32)), src));
            remv |= (static_cast<unsigned long long>(hi) << 32) | lo;
        }
        // (3) resolve: kept[l] = alive[l] and no kept killer (fixed point over the wave, as in the barrier form)
        const unsigned long long killers = lane < lim ? w.x : 0ull;
        const unsigned long long alive = ~remv & valid;
        // two rounds unconditionally (no suppression inside the block -- the usual case -- is confirmed by the second), then until stable
        unsigned long long kb = alive & ~__ballot((killers & alive) != 0ull);
        unsigned long long nextk = alive & ~__ballot((killers & kb) != 0ull);
        while (nextk != kb) {
            kb = nextk;
            nextk = alive & ~__ballot((killers & kb) != 0ull);
        }
        __builtin_amdgcn_raw_buffer_store_b32(static_cast<unsigned>(blk * 64 + lane), rkeep,
                                              ((kb >> lane) & 1ull) ? (total_kept + mask_prefix(kb)) * 4 : -16, 0, 0);
        total_kept += __builtin_popcountll(kb);
        kb_prev = kb;
        // every lane stores the same values: no exec juggling, no branch
        keptw[blk] = kb;   // read by this wave's later steps: the LDS keeps a wave's accesses in order
        __hip_atomic_store(consumed_pref, base + cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // the ring up to here is free
        __hip_atomic_store(consumed_blk, blk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (lane == 0) *kept_total = total_kept;
}

__global__ __launch_bounds__(kSweepThreads) void nms_sweep_kernel(int n, int chunk_blocks, const unsigned char *__restrict__ ws_base,
                                                                  size_t ws_stride, int *__restrict__ keep,
                                                                  int *__restrict__ num_kept)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
    const int cb = (n + 63) / 64;
    // blockIdx.x = frame of a batched call
    const unsigned char *ws = ws_base + static_cast<size_t>(blockIdx.x) * ws_stride;
    const unsigned long long *mask = reinterpret_cast<const unsigned long long *>(ws);
    const int *counts = reinterpret_cast<const int *>(ws + nms_ws_counts_offset(n));
    const NmsEntry *lists = reinterpret_cast<const NmsEntry *>(ws + nms_ws_lists_offset(n));
    const unsigned long long *diagt = reinterpret_cast<const unsigned long long *>(ws + nms_ws_diagt_offset(n));
    const unsigned long long *nextw = reinterpret_cast<const unsigned long long *>(ws + nms_ws_nextw_offset(n));
    keep += static_cast<size_t>(blockIdx.x) * n;
    if (num_kept) num_kept += blockIdx.x;
    ull2 *dn = reinterpret_cast<ull2 *>(smem_raw);                                  // chunk_blocks * 64: (transposed diagonal word, next word)
    unsigned long long *keptw = reinterpret_cast<unsigned long long *>(dn + chunk_blocks * 64);   // cb: kept bits per block
    unsigned long long *remv = keptw + cb;                                          // cb: removal word per block
    int *cnts = reinterpret_cast<int *>(remv + cb);                                 // cb: entries filed under the block
    __shared__ int kept_total;
    __shared__ int any_overflow, consumed_pref, scan_ws[32];
    const int t = threadIdx.x;
    // ---- the streaming form (round 4): no barrier per column block.  Taken when no column's list overflowed and the per-block
    // arrays plus the entry ring fit in LDS (cb <= kStreamMaxBlocks = 65 536 boxes); else the barrier form below. ----
    if (t == 0) { any_overflow = 0; scan_ws[31] = 0; }
    __syncthreads();
    if (cb <= kStreamMaxBlocks) {
        // not taken either when the lists are long on average: one wave then tests more entries per block than the fifteen
        // gatherers of the barrier form do in a step
        int mine = 0;
        for (int w = t; w < cb; w += kSweepThreads) {
            const int c = counts[w];
            if (c > kNmsListCap) any_overflow = 1;
            mine += c;
        }
        if (mine > kStreamMaxAverage * cb / 4) any_overflow = 1;            // a thread's share alone says "long" (cheap early out)
        else if (mine > 0 && atomicAdd(&scan_ws[31], mine) + mine > kStreamMaxAverage * cb) any_overflow = 1;
    }
    __syncthreads();
    if (cb <= kStreamMaxBlocks && !any_overflow) {
        sweep_stream(n, cb, counts, lists, diagt, nextw, keep, smem_raw, &kept_total, &consumed_pref, scan_ws);
        __syncthreads();
        const int kept_s = kept_total;
        for (int p = kept_s + t; p < n; p += kSweepThreads) keep[p] = 0;   // pad with keep[0] (bev_iou.cpp:110-112)
        if (t == 0 && num_kept) *num_kept = kept_s;
        return;
    }
    for (int w = t; w < cb; w += kSweepThreads) { remv[w] = 0ull; keptw[w] = 0ull; cnts[w] = counts[w]; }
    if (t == 0) kept_total = 0;
    // every chunk_blocks steps all threads load the word pairs of the next chunk: wave 0 never waits for memory
    auto load_chunk = [&](int c0) {
        const int i0 = c0 * 64, i1 = min(n, (c0 + chunk_blocks) * 64);
        for (int i = i0 + t; i < i1; i += kSweepThreads) dn[i - i0] = ull2{ diagt[i], (i >> 6) + 1 < cb ? nextw[i] : 0ull };
    };
    __syncthreads();
    // Two loops with the same barriers (one per step): the resolving wave and the gatherers share nothing but LDS.
    // Step s: wave 0 decides block s while the gatherers already collect column s+1 from the rows decided before.
    if (t < 64) {
        __builtin_amdgcn_s_setprio(3);
        for (int blk = 0; blk < cb; ++blk) {
            if (blk % chunk_blocks == 0) {
                load_chunk(blk);
                __syncthreads();
            }
            const int i = blk * 64 + t;
            const int lim = min(64, n - blk * 64);
            const unsigned long long valid = lim == 64 ? ~0ull : ((1ull << lim) - 1ull);
            const int slot = (blk % chunk_blocks) * 64 + t;
            // the boxes kept in block blk-1 suppress through their NEXT word (the lists start two blocks to the right)
            if (blk > 0 && ((keptw[blk - 1] >> t) & 1ull)) {
                const unsigned long long nw = (blk % chunk_blocks) != 0 ? dn[slot - 64].y : nextw[i - 64];
                if (nw != 0ull) atomicOr(&remv[blk], nw);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            asm volatile("" ::: "memory");
            // who (of this block) suppresses my box if kept: bits below my lane only
            const unsigned long long killers = t < lim ? dn[slot].x : 0ull;
            const unsigned long long alive = ~remv[blk] & valid;   // not suppressed by boxes kept in earlier blocks
            // kept[l] = alive[l] and no kept killer: its unique solution is the fixed point of the whole-wave update below
            // (position l is final after l+1 rounds; a chain of suppressions is rarely deeper than a few boxes)
            unsigned long long kb = alive;
            for (;;) {
                const unsigned long long next = alive & ~__ballot((killers & kb) != 0ull);
                if (next == kb) break;   // wave-uniform
                kb = next;
            }
            const int before = kept_total;
            if ((kb >> t) & 1ull) keep[before + __builtin_popcountll(kb & ((1ull << t) - 1ull))] = i;
            if (t == 0) { kept_total = before + __builtin_popcountll(kb); keptw[blk] = kb; }
            __syncthreads();   // block blk is decided, column blk+1 has everything from blocks < blk
        }
    } else {
        const int u = t - 64;
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        // the first entry of a column for this thread, requested kSweepLead steps ahead through a range-checked buffer
        // load (a column with fewer entries, or past the last one, returns zeros: row 0 / word 0 adds nothing)
        auto request = [&](int col) -> u4v {
            const bool live = col < cb;   // uniform
            const int cnt = live ? min(cnts[col], kNmsListCap) : 0;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<NmsEntry *>(lists + static_cast<size_t>(live ? col : 0) * kNmsListCap), 0, cnt * 16, 0x00020000);
            return __builtin_amdgcn_raw_buffer_load_b128(rs, u * 16, 0, 0);
        };
        auto kept_row = [&](int row) -> bool { return (keptw[row >> 6] >> (row & 63)) & 1ull; };
        u4v e0 = request(1), e1 = request(2), e2 = request(3), e3 = request(4);
        static_assert(kSweepLead == 4, "four entry registers rotate below");
        // step blk gathers column blk+1: its entries come from row blocks <= blk-1, all decided
        auto gstep = [&](int blk, u4v &e) {
            if (blk % chunk_blocks == 0) {
                load_chunk(blk);
                __syncthreads();
            }
            const int col = blk + 1;
            if (col < cb) {
                const int cnt = cnts[col];   // uniform
                unsigned long long acc = 0ull;
                if (cnt <= kNmsListCap) {
                    const unsigned long long w0 = (static_cast<unsigned long long>(e.y) << 32) | e.x;
                    if (w0 != 0ull && kept_row(static_cast<int>(e.z))) acc = w0;
                    for (int k = u + kSweepGatherers; k < cnt; k += kSweepGatherers) {   // long lists: the rest, on demand
                        const NmsEntry x = lists[static_cast<size_t>(col) * kNmsListCap + k];
                        if (kept_row(x.row)) acc |= x.word;
                    }
                } else {
                    // overflow: the dense mask, one word per kept row of the blocks before blk
                    for (int i = u; i < blk * 64; i += kSweepGatherers)
                        if (kept_row(i)) acc |= mask[static_cast<size_t>(i) * cb + col];
                }
                if (acc != 0ull) atomicOr(&remv[col], acc);   // few lanes hold anything: cheaper than a wave reduction first
            }
            e = request(col + kSweepLead);
            __syncthreads();
        };
        for (int base = 0; base < cb; base += 4) {
            gstep(base, e0);
            if (base + 1 < cb) gstep(base + 1, e1);
            if (base + 2 < cb) gstep(base + 2, e2);
            if (base + 3 < cb) gstep(base + 3, e3);
        }
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
static int tile_lds_attr(K kernel, size_t bytes = sizeof(TileShared))
{
    return ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), bytes);
}

HF_API int hf_compute_bev_iou(int num_a, const float *boxes_a, int num_b, const float *boxes_b, float *ans_overlap,
                              float *ans_iou, hf_stream_t stream)
{
    // ComputeBevIOUOp: N > 0, M > 0, (N,5) / (M,5)  (bev_iou.cpp:156-157)
    if (num_a <= 0 || num_b <= 0 || !boxes_a || !boxes_b) return HF_EINVAL;
    if (!ans_overlap && !ans_iou) return HF_OK;
    const int gy = (num_a + kIouRows - 1) / kIouRows, gx = (num_b + 63) / 64;
    if (const int lrc = tile_lds_attr(&bev_iou_kernel, sizeof(IouShared)); lrc != HF_OK) return lrc;
    const int stop = HF_DIAG_INT("HF_BEV_STOP", 0);   // diagnostic builds only: early exit after a phase (outputs invalid)
    if ((ans_overlap && reinterpret_cast<uintptr_t>(ans_overlap) % 16 != 0) || (ans_iou && reinterpret_cast<uintptr_t>(ans_iou) % 16 != 0))
        return HF_EINVAL;   // the outputs are written with 16-byte stores (every allocator returns at least that alignment)
    if (gy > 65535) {
        // very tall matrices: walk the rows in slabs of 65535 tiles
        for (int y0 = 0; y0 < gy; y0 += 65535) {
            const int rows0 = y0 * kIouRows;
            const int na = std::min(num_a - rows0, 65535 * kIouRows);
            hipLaunchKernelGGL(bev_iou_kernel, dim3(gx, (na + kIouRows - 1) / kIouRows), dim3(kIouThreads), sizeof(IouShared), as_stream(stream), na,
                               boxes_a + static_cast<size_t>(rows0) * 5, num_b, boxes_b,
                               ans_overlap ? ans_overlap + static_cast<size_t>(rows0) * num_b : nullptr,
                               ans_iou ? ans_iou + static_cast<size_t>(rows0) * num_b : nullptr, stop);
        }
        return launch_status();
    }
    hipLaunchKernelGGL(bev_iou_kernel, dim3(gx, gy), dim3(kIouThreads), sizeof(IouShared), as_stream(stream), num_a, boxes_a, num_b,
                       boxes_b, ans_overlap, ans_iou, stop);
    return launch_status();
}

HF_API int hf_nms_mask(const float *boxes, unsigned long long *mask, int boxes_num, float nms_overlap_thresh,
                       hf_stream_t stream)
{
    if (boxes_num <= 0 || !boxes || !mask) return HF_EINVAL;
    const int cb = (boxes_num + 63) / 64;
    if (cb > 65535) return HF_EINVAL;
    if (const int lrc = tile_lds_attr(&nms_mask_kernel<false>); lrc != HF_OK) return lrc;
    hipLaunchKernelGGL((nms_mask_kernel<false>), dim3(cb, cb), dim3(kNmsThreads), sizeof(TileShared), as_stream(stream), boxes_num,
                       nms_overlap_thresh, boxes, mask, static_cast<size_t>(0));
    return launch_status();
}

HF_API size_t hf_oriented_nms_workspace(int n)
{
    if (n <= 0) return 0;
    return nms_ws_bytes(n);   // dense mask + per-column-block counts + per-column-block lists of nonzero words
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
    if (reinterpret_cast<uintptr_t>(workspace) % 16 != 0) return HF_EINVAL;   // the list entries are read 16 bytes at a time
    const int cb = (n + 63) / 64;
    // LDS: kept bits + removal words per block, and the diagonal words of a chunk of blocks (all of them up to 8192 boxes)
    const int chunk_blocks = std::min(cb, 64);   // (transposed diagonal, next) word pairs of 4096 boxes: 64 KB
    size_t lds = sizeof(unsigned long long) * (2 * static_cast<size_t>(cb) + 2 * 64 * static_cast<size_t>(chunk_blocks)) + sizeof(int) * static_cast<size_t>(cb);
    if (cb > 8192 || lds > 160 * 1024 - 256) return HF_EINVAL;   // n <= 524 288 boxes (pre_nms_size is 9000)
    if (cb <= kStreamMaxBlocks)   // the streaming sweep: per-block arrays + the ring of list entries
        lds = std::max(lds, sizeof(unsigned long long) * cb + sizeof(int) * (4 * static_cast<size_t>(cb) + 1) + 16 + 16 * 64 * static_cast<size_t>(kStreamWords) +
                                sizeof(NmsEntry) * kStreamRing);
    hipStream_t st = as_stream(stream);
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    const size_t ws_stride = nms_ws_bytes(n);
    // the dense mask (its tiles store nonzero words only) and the list counters behind it start at zero: one strided memset over
    // [mask | counters] of every frame's workspace (the counters follow the mask, padded to 256 bytes)
    int rc = hip_status(hipMemset2DAsync(ws, ws_stride, 0, nms_ws_counts_offset(n) + sizeof(int) * static_cast<size_t>(cb), frames, st));
    if (rc != HF_OK) return rc;
    hipLaunchKernelGGL(nms_boxpre_kernel, dim3(div_up(n, 256), frames), dim3(256), 0, st, n, boxes, ws, ws_stride);
    if (const int lrc = tile_lds_attr(&nms_mask_kernel<true>); lrc != HF_OK) return lrc;
    hipLaunchKernelGGL((nms_mask_kernel<true>), dim3(cb, cb, frames), dim3(kNmsThreads), sizeof(TileShared), st, n, thresh, boxes,
                       reinterpret_cast<unsigned long long *>(ws), ws_stride);
    rc = launch_status();
    if (rc != HF_OK) return rc;
    if (HF_DIAG_INT("HF_NMS_STOP", 0) == 1) return HF_OK;   // diagnostic builds only: mask kernel alone (keep[] is not written)
    rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&nms_sweep_kernel), lds);
    if (rc != HF_OK) return rc;
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(frames), dim3(kSweepThreads), lds, st, n, chunk_blocks, ws, ws_stride, keep, num_kept);
    return launch_status();
}
