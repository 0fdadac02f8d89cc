// ballquery.hip -- query_ball_point (+ fused group_point on xyz) for gfx950: the cell kernel (main path).
//
// Replaces grouping/tf_grouping_g.cu:3-36 (query_ball_point_gpu) and, fused, :40-57 (group_point_gpu on the
// coordinates) plus the `grouped_xyz -= new_xyz` of pointnet_util.py:56 of the reference.
//
// Semantics kept bit for bit: query j scans the data points in ascending index, a point is a hit iff
// max(sqrtf(s),1e-20f) < radius  <=>  s < T (T computed exactly on the host, grouping.hip:ball_threshold) with
// s = (x2-x1)^2+(y2-y1)^2+(z2-z1)^2, no FMA; the row holds the first nsample hits padded with the first one;
// pts_cnt = min(hits, nsample); a row without hits is all zeros.
//
// Design.  A brute-force scan is b*m*n pair tests (537 M at the headline shape) for a ball that holds a handful
// of points.  One workgroup (1024 threads) owns QPW CONSECUTIVE queries of one cloud -- consecutive in the
// query index, so every output of the workgroup (idx rows, grouped rows, counts) is one contiguous range.
// Space is cut into cubic cells of width cs = 2.2 radius in a FIXED frame: cell(v) = round(v / cs) per axis, read off
// the mantissa of one fma (no bounding box, no data-dependent grid: nothing to reduce before the first useful
// instruction).  A cell is hashed to one of 131072 bits: word = (cy + A cx + B cz) mod 4096, bit = cx mod 32
// (8 VALU instructions per point: 3 fma, 2 mad_u24, 1 and, the LDS read, 1 bfe, 1 merge).  The workgroup
//   1  loads its queries, clears the bitmap and the list heads, and requests the cloud (16 points per thread stay
//      in registers);
//   2  marks, in a 16 KB bitmap in LDS, the <= 8 cells that each query's padded box [q - rp, q + rp] touches
//      (rp > radius; cs >= 2 rp, so <= 2 cells per axis);
//   3  tests every point of the cloud against the bitmap (one 4-byte LDS read per point): a point whose bit is
//      clear cannot be a hit of any of these queries (cell() is monotone, so a point inside a query's padded box
//      computes one of that query's cells; hash collisions only add harmless candidates).  One LDS atomic per
//      WAVE reserves slots for the survivors (a few hundred of 16384);
//   4  the lanes that hold survivors re-read their coordinates (the register arrays cannot be indexed at run time;
//      the latency hides behind the other waves' step 3) and push each onto the list of its cell (4096 list heads,
//      direct-mapped by 12 hash bits);
//   5  every query (G = 1024 / QPW lanes each, one lane per cell) walks the lists of its <= 8 cells, applies the
//      reference's exact fp32 test, and keeps the nsample SMALLEST data indices in a sorted LDS row (= the first
//      nsample hits of the reference's ascending scan);
//   6  every wave writes the rows of its own queries with 16-byte stores (no workgroup barrier after step 5:
//      rows leave as soon as their wave is done).
// If the survivors do not fit the LDS buffer (dense clouds) the cloud is walked in index order in chunks of 2048
// points, each chunk binned / searched as above into the same rows; the walk stops as soon as every
// row is full, because later chunks hold larger indices only (the reference's `break`).
// A query whose coordinates are so large against the radius that its padded box could span three cells switches its
// workgroup to the exhaustive mode (every point a candidate of every query): correct, slow, never seen on real clouds.
// Results do not depend on the cell geometry, on the arrival order of the atomics or on the chunking: the hit
// set is decided by the same `s < T` on every candidate and the order by the data index alone.
#include <math.h>

#include <type_traits>

#include "bq_common.h"

namespace hf {

constexpr int kCellSegPoints = 16384;   // points per register segment: NT threads hold 16384 / NT points each
constexpr int kCellWords = 4096;        // bitmap words, 32 hashed cells each (2048 / 4096 / 8192 words: 69.4 / 66.7 / 67.8 us at 80 clouds,
                                        // 10.8 / 10.8 / 10.9 us at 8)
constexpr int kCellHeads = 4096;        // list heads, direct-mapped by 12 hash bits
// key = 4 (cy + A cx + B cz) + (quarter-cell fraction of y): bits 2..13 = bitmap word, the bit in the word = cx mod 32
// (x is a long axis of a LiDAR scene in the camera frame and in the sensor frame alike; A, B: fewest false candidates
// on KITTI-sized scenes at radii 0.1 .. 2 in both frames, as good as a full 16-bit multiplicative hash).
// List head = the low 11 bits of that word index | (cx & 1) << 11.  The 8 cells of a query never share a list: the two x sides differ
// in cx & 1, and within one the word offsets {0, 1, B, B + 1} are distinct.
constexpr unsigned kCellHashA = 97u;
constexpr unsigned kCellHashB = 75u;
constexpr unsigned kCellWordMask = (kCellWords - 1) << 2;   // byte offset of the bitmap word inside the key
constexpr int kCellChunk = 2048;        // dense path: points per flush (<= cap)
constexpr int kCellSlotBits = 12;       // row entry = (data index << 12) | slot in the LDS candidate buffer
constexpr int kCellIdxBits = 19;        // candidate word = data index | (next slot + 1) << 19: n <= 2^19, cap < 2^12

struct CellShared {
    int nc;                             // candidates appended so far
    int exh;                            // exhaustive mode (a query's padded box may span three cells)
    int flag;                           // block-wide AND (all_threads)
};

// LDS word at a byte offset from the start of the workgroup's allocation.  The kernel has no static LDS object, so
// its dynamic region starts at 0 (cell_launch checks the attribute): the address needs no base added per access.
constexpr unsigned kCellBitsOffset = 64;
__device__ __forceinline__ unsigned lds_u32(unsigned byte_off)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) unsigned *>(static_cast<size_t>(byte_off));
}

// Diagnostic build only (-DHF_QBP_STAMPS, scripts/probes/qbp_stamps.py): wave 0 of every workgroup stamps the
// shader clock at the phase boundaries into a buffer of its own.  The product library is built without it.
#ifdef HF_QBP_STAMPS
__device__ unsigned long long g_qbp_stamps[4096 * 16];
#define HF_STAMP(i)                                                                                                   \
    do {                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        unsigned long long ts_;                                                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_)::"memory");                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        if (threadIdx.x == 0 && blockIdx.x + gridDim.x * blockIdx.y < 4096)                                          \
            g_qbp_stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (i)] = ts_;                                     \
    } while (0)
#define HF_STAMP_REAL(i, thr)                                                                                         \
    do {                                                                                                              \
        unsigned long long ts_;                                                                                       \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_)::"memory");                               \
        if (threadIdx.x == (thr) && blockIdx.x + gridDim.x * blockIdx.y < 4096)                                      \
            g_qbp_stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (i)] = ts_;                                     \
    } while (0)
#define HF_STAMP_LAST(i)                                                                                              \
    do {                                                                                                              \
        unsigned long long ts_;                                                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_)::"memory");                                   \
        if (threadIdx.x == blockDim.x - 1 && blockIdx.x + gridDim.x * blockIdx.y < 4096)                             \
            g_qbp_stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (i)] = ts_;                                     \
    } while (0)
#else
#define HF_STAMP(i) do { } while (0)
#define HF_STAMP_LAST(i) do { } while (0)
#define HF_STAMP_REAL(i, thr) do { } while (0)
#endif

template <bool GROUP, int SM, int NT, bool A4>
__global__ __launch_bounds__(NT) void qbp_cell_kernel(int n, int m, int qpw, int glog, float radius,
                                                                float thresh, float inv_cs, int nsample,
                                                                int ns_shift, int cap, int stop,
                                                                const float *__restrict__ xyz1,
                                                                const float *__restrict__ xyz2, int center,
                                                                int *__restrict__ idx, int *__restrict__ pts_cnt,
                                                                float *__restrict__ grouped)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    CellShared &sh = *reinterpret_cast<CellShared *>(smem_raw);
    unsigned *bits = reinterpret_cast<unsigned *>(smem_raw + 64);      // kCellWords
    int *head = reinterpret_cast<int *>(bits + kCellWords);            // kCellHeads   newest candidate of the list, -1: none
    float4 *qbuf = reinterpret_cast<float4 *>(head + kCellHeads);      // qpw   (x, y, z, -)
    float4 *cand = qbuf + qpw;                                         // cap   (x, y, z, index | next << 19)
    int *rows = reinterpret_cast<int *>(cand + cap);                   // qpw * rs   nsample smallest, ascending
    const int rs = ((nsample + 3) & ~3) + 4;                           // 16-byte rows, stride = 4 banks mod 32
    int *hits = rows + qpw * rs;                                       // qpw   min(total hits, nsample)
    int *stage = hits + qpw;                                           // 2 NT   hand-off inside a wave (8-byte aligned)
    constexpr int kCellThreads = NT;
    constexpr int kCellPPT = kCellSegPoints / NT;
    constexpr int kCellSeg = kCellSegPoints;
    constexpr int kCellChunkSlots = kCellChunk / NT;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar: buffer descriptors are built from it
    const int G = 1 << glog;
    const int sub = t & (G - 1);
    const int gbase = lane & ~(G - 1);
    // grid = (clouds, query tiles): linear workgroup id = cloud + b * tile; workgroups are dealt round-robin over
    // the 8 XCDs, so with b a multiple of 8 all tiles of a cloud read it from ONE XCD's L2 (speed only).
    const int bb = blockIdx.x, j0 = blockIdx.y * qpw;
    const int nq = min(qpw, m - j0);
    const float *p1 = xyz1 + static_cast<size_t>(bb) * n * 3;
    const float *p2 = xyz2 + (static_cast<size_t>(bb) * m + j0) * 3;
    if (stop == -1) return;
    HF_STAMP_REAL(12, 0);
    HF_STAMP(0);

    // ---------------- 1: my query first (vmcnt retires in order), then the cloud ----------------
    const int qi = t >> glog;            // the query this lane serves in steps 5 and 6
    const bool qlive = qi < nq;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    const __amdgpu_buffer_rsrc_t rcloud = make_rsrc(p1, static_cast<unsigned>(n) * 12u);
    {
        const P3 q = load_p3(make_rsrc(p2, static_cast<unsigned>(nq) * 12u), qi);   // qi >= nq: zeros
        qx = q.x; qy = q.y; qz = q.z;
    }
    // the first half of the cloud is requested now: the texture path streams it in (196 KB per workgroup at
    // 64 bytes per clock is ~3000 cycles) while the bitmap is built; the second half follows the marking
    float px[kCellPPT], py[kCellPPT], pz[kCellPPT];
    constexpr int kEarly = kCellPPT / 2;
    // points u0 .. u1-1 of this thread in the segment that starts at seg0.
    // A4 (the cloud starts on a 16-byte boundary and n is a multiple of 4): slot u = point 4 (t + (u/4) NT) + u%4, i.e. a
    // thread owns FOUR CONSECUTIVE points = 48 bytes = three aligned 16-byte loads (three texture-path requests for four
    // points instead of four; scripts/probes/l2_read_probe.hip).  Otherwise slot u = point u NT + t, one 12-byte load each.
    auto load_points = [&](auto u0_tag, auto u1_tag, int seg0) {
        constexpr int U0 = decltype(u0_tag)::value, U1 = decltype(u1_tag)::value;
        if constexpr (A4) {
            static_assert(U0 % 4 == 0 && U1 % 4 == 0, "whole quads");
            typedef unsigned u4v __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int q = U0 / 4; q < U1 / 4; ++q) {
                const int byte0 = (seg0 + 4 * (q * kCellThreads + t)) * 12;   // past the end: zeros, masked out later
                const u4v a = __builtin_amdgcn_raw_buffer_load_b128(rcloud, byte0, 0, 0);
                const u4v b = __builtin_amdgcn_raw_buffer_load_b128(rcloud, byte0 + 16, 0, 0);
                const u4v c = __builtin_amdgcn_raw_buffer_load_b128(rcloud, byte0 + 32, 0, 0);
                const int u = 4 * q;
                px[u] = __uint_as_float(a.x); py[u] = __uint_as_float(a.y); pz[u] = __uint_as_float(a.z);
                px[u + 1] = __uint_as_float(a.w); py[u + 1] = __uint_as_float(b.x); pz[u + 1] = __uint_as_float(b.y);
                px[u + 2] = __uint_as_float(b.z); py[u + 2] = __uint_as_float(b.w); pz[u + 2] = __uint_as_float(c.x);
                px[u + 3] = __uint_as_float(c.y); py[u + 3] = __uint_as_float(c.z); pz[u + 3] = __uint_as_float(c.w);
            }
        } else {
#pragma unroll
            for (int u = U0; u < U1; ++u) {
                const P3 p = load_p3(rcloud, static_cast<unsigned>(seg0 + u * kCellThreads + t));   // past the end: zeros, masked out later
                px[u] = p.x; py[u] = p.y; pz[u] = p.z;
            }
        }
    };
    HF_STAMP(9);
    {
        typedef int i4 __attribute__((ext_vector_type(4)));
        for (int i = t; i < kCellWords / 4; i += NT) reinterpret_cast<i4 *>(bits)[i] = i4{ 0, 0, 0, 0 };
        for (int i = t; i < kCellHeads / 4; i += NT) reinterpret_cast<i4 *>(head)[i] = i4{ -1, -1, -1, -1 };
        if (t == 0) { sh.nc = 0; sh.exh = 0; }
    }
    HF_STAMP(10);
    __syncthreads();
    // the cloud requests go out behind the barrier: the query's latency is covered by issuing them
    load_points(std::integral_constant<int, 0>{}, std::integral_constant<int, kEarly>{}, 0);
    HF_STAMP(1);

    // cell of v along one axis = round(v / cs), read off the mantissa of fma(v, 1/cs, 1.5 * 2^23): one instruction,
    // monotone in v; exact while |v / cs| < 2^22 (queries beyond 2^21 cells switch to the exhaustive mode)
    auto cellc = [&](float v) -> unsigned { return __float_as_uint(__builtin_fmaf(v, inv_cs, 12582912.0f)); };
    // the y axis in quarter cells (same trick, 4 / cs): its cell is the FLOOR of a quarter of that, the two fraction bits
    // ride along in the low bits of the key where nothing reads them.  Exact while |v / cs| < 2^20.
    const float inv_cs4 = 4.0f * inv_cs;
    auto cellq = [&](float v) -> unsigned { return __float_as_uint(__builtin_fmaf(v, inv_cs4, 12582912.0f)); };
    // ---------------- 2: mark the cells my query's padded box touches ----------------
    // key of the low-corner cell (bits 0..13); bits 16..18: the box reaches into the next cell along x / y / z;
    // bits 20..24: cx of the low corner mod 32
    unsigned qhash = 0u;
    auto corner_key = [&](int c) -> unsigned {
        return qhash + 4u * ((c & 1) * kCellHashA + ((c >> 1) & 1) + ((c >> 2) & 1) * kCellHashB);
    };
    auto corner_cx = [&](int c) -> unsigned { return (qhash >> 20) + (c & 1); };   // low 5 bits
    auto head_of = [&](unsigned key, unsigned cx) -> unsigned { return ((key >> 2) & 2047u) | ((cx & 1u) << 11); };
    {
        // pad: > radius plus the fp32 rounding of q -/+ rp
        const float qmax = fmaxf(fabsf(qx), fmaxf(fabsf(qy), fabsf(qz)));
        const float rp = radius * 1.001f + 1e-6f * qmax;
        const unsigned lx = cellc(qx - rp), ly = cellq(qy - rp) >> 2, lz = cellc(qz - rp);
        const unsigned hx = cellc(qx + rp) - lx, hy = (cellq(qy + rp) >> 2) - ly, hz = cellc(qz + rp) - lz;
        // cs >= 2 rp with slack: 0 <= h <= 1.  Otherwise (coordinates ~1e5 radii away from the origin) -> exhaustive mode
        if (qlive && (2.0f * rp * inv_cs > 0.995f || ((hx | hy | hz) & ~1u) || !(qmax * inv_cs < 524288.0f))) sh.exh = 1;
        qhash = mad_u24(lz, 4u * kCellHashB, mad_u24(lx, 4u * kCellHashA, ly << 2)) & kCellWordMask;
        qhash |= static_cast<unsigned>((hx & 1) | ((hy & 1) << 1) | ((hz & 1) << 2)) << 16;
        qhash |= (lx & 31u) << 20;
        for (int c = sub; c < 8; c += G) {   // G >= 8: one cell per lane; G = 4: two
            if (qlive && (c & ~(qhash >> 16)) == 0) {
                atomicOr(&bits[(corner_key(c) & kCellWordMask) >> 2], 1u << (corner_cx(c) & 31u));
            }
        }
        if (qlive && sub == 0) qbuf[qi] = make_float4(qx, qy, qz, 0.f);
    }
    HF_STAMP(2);
    __syncthreads();
    HF_STAMP(3);
    if (stop == -2) return;
    const bool exh = sh.exh != 0;   // uniform
    load_points(std::integral_constant<int, kEarly>{}, std::integral_constant<int, kCellPPT>{}, 0);

    // key of the cell of a point: the raw float bits go into the multiply-adds (24-bit operands; only bits 0..13 of the
    // result are used and those depend on the low 14 bits of the operands alone); cx = raw bits of the x cell
    auto point_key = [&](float x, float y, float z, unsigned &cx) -> unsigned {
        cx = cellc(x);
        return mad_u24(cellc(z), 4u * kCellHashB, mad_u24(cx, 4u * kCellHashA, cellq(y)));
    };
    // NP points per thread (k = k0 + u * 1024, valid below lim):
    //   pass 1  bitmap lookups, four in flight, -> one mask bit per point (no divergence);
    //   slots   one LDS atomic per WAVE reserves the wave's candidate slots (DPP prefix of the lane counts);
    //   pass 2  every lane walks the set bits of its mask and records the data index in its slots.
    // quad_tag: slot u = point k0 + 4 (u/4) NT + u%4 (the A4 layout, k0 = seg0 + 4 t); else k0 + u NT (k0 = seg0 + t)
    auto add_points = [&](auto np_tag, auto quad_tag, const float *x, const float *y, const float *z, int k0, int lim) {
        constexpr int NP = decltype(np_tag)::value;
        constexpr bool QUAD = decltype(quad_tag)::value;
        auto kof = [&](int u) -> int { return QUAD ? k0 + 4 * (u >> 2) * kCellThreads + (u & 3) : k0 + u * kCellThreads; };
        unsigned mask = 0u;
#pragma unroll
        for (int u0 = 0; u0 < NP; u0 += 4) {
            unsigned cx[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (u0 + i < NP) w[i] = lds_u32(kCellBitsOffset + (point_key(x[u0 + i], y[u0 + i], z[u0 + i], cx[i]) & kCellWordMask));
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (u0 + i < NP) mask |= __builtin_amdgcn_ubfe(w[i], cx[i], 1u) << (u0 + i);
            __builtin_amdgcn_sched_barrier(0);   // finish a group before the next: the waits on the cloud loads stay incremental
        }
        HF_STAMP(14);
        if (exh) mask = NP >= 32 ? ~0u : (1u << (NP & 31)) - 1u;
        // valid points: kof(u) < lim; in the quad layout lim is a multiple of 4, so quads are valid or not as a whole
        const int nvalid = lim > k0 ? (QUAD ? min(NP, 4 * ((lim - k0 + 4 * kCellThreads - 1) / (4 * kCellThreads)))
                                            : min(NP, (lim - k0 + kCellThreads - 1) / kCellThreads)) : 0;
        mask &= nvalid >= 32 ? ~0u : (1u << (nvalid & 31)) - 1u;
        const int mine = __builtin_popcount(mask);
        const int inc = wave_inclusive_scan_i32(mine);
        const int wtot = __builtin_amdgcn_readlane(inc, 63);
        if (wtot == 0) return;   // wave-uniform
        int base = 0;
        if (lane == 0) base = atomicAdd(&sh.nc, wtot);
        int slot = __builtin_amdgcn_readfirstlane(base) + inc - mine;
        HF_STAMP(15);
        // pass 2 (few lanes, few rounds): re-read the coordinates of up to four candidates at once -- the register
        // arrays cannot be indexed by a run-time u -- and push each onto the list of its cell.  The latency of the
        // re-reads hides behind the other waves' pass 1; no separate link step, no barrier for it.
        while (mask) {
            int kk[4];
            P3 pp[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                kk[i] = -1;
                if (mask) {
                    const int u = __builtin_ctz(mask);
                    mask &= mask - 1u;
                    kk[i] = kof(u);
                    pp[i] = load_p3(rcloud, static_cast<unsigned>(kk[i]));
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (kk[i] >= 0) {
                    if (slot < cap) {
                        unsigned cx;
                        const unsigned key = point_key(pp[i].x, pp[i].y, pp[i].z, cx);
                        const unsigned h = exh ? 0u : head_of(key, cx);
                        const int old = atomicExch(&head[h], slot);
                        cand[slot] = make_float4(pp[i].x, pp[i].y, pp[i].z, __int_as_float(kk[i] | ((old + 1) << kCellIdxBits)));
                    }
                    ++slot;
                }
            }
        }
    };
    // block-wide AND (rare paths only; __syncthreads_and would add a static LDS object in front of the dynamic region)
    auto all_threads = [&](bool pred) -> bool {
        __syncthreads();
        if (t == 0) sh.flag = 1;
        __syncthreads();
        if (!pred) sh.flag = 0;
        __syncthreads();
        return sh.flag != 0;
    };

    // ---- step 5 on the current candidate lists; called by all threads together ----
    int myhits = 0;   // total hits of my query (owner lane: sub == 0)
    int mylen = 0;
    int *const row = rows + qi * rs;
    // owner lane: one more hit, keep the nsample smallest (sorted ascending)
    auto insert = [&](int kk) {
        ++myhits;
        if (mylen < nsample || kk < row[mylen - 1]) {
            int pos = mylen < nsample ? mylen++ : mylen - 1;
            while (pos > 0 && row[pos - 1] > kk) { row[pos] = row[pos - 1]; --pos; }
            row[pos] = kk;
        }
    };
    // lane `sub` of the query's group walks the lists of cells sub, sub + G, ... < 8 (cell = x side, y side, z side bits)
    auto first_list = [&](int &cur, int &cnext) {
        cur = -1;
        cnext = sub;
        if (exh) { cnext = 8; cur = (qlive && sub == 0) ? head[0] : -1; }
        else if (G >= 8) {   // one cell per lane: no queue of lists
            cnext = 8;
            if (qlive && sub < 8 && (sub & ~(qhash >> 16)) == 0)
                cur = head[head_of(corner_key(sub), corner_cx(sub))];
        }
    };
    auto next_list = [&](int &cur, int &cnext) {
        while (cur < 0 && cnext < 8) {
            if (qlive && (cnext & ~(qhash >> 16)) == 0)
                cur = head[head_of(corner_key(cnext), corner_cx(cnext))];
            cnext += G;
        }
    };
    // one list element: exact test; returns the row entry (index << 12 | slot) or -1, and advances
    auto visit = [&](int &cur) -> int {
        const float4 c = cand[cur];
        const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
        const float s2 = dx * dx + dy * dy + dz * dz;
        const int w = __float_as_int(c.w);
        const int k = ((w & ((1 << kCellIdxBits) - 1)) << kCellSlotBits) | cur;
        cur = static_cast<int>(static_cast<unsigned>(w) >> kCellIdxBits) - 1;
        return s2 < thresh ? k : -1;
    };
    auto search = [&]() {
        // fast form: every lane keeps its first two hits in registers, no traffic between lanes while walking
        int cur, cnext;
        first_list(cur, cnext);
        next_list(cur, cnext);
        int h0 = -1, h1 = -1, nh = 0;
        while (__ballot(cur >= 0) != 0ull) {   // wave-uniform
            if (cur >= 0) {
                const int k = visit(cur);
                if (k >= 0) {
                    if (nh == 0) h0 = k; else if (nh == 1) h1 = k;
                    ++nh;
                }
                next_list(cur, cnext);
            }
        }
        if (__ballot(nh > 2) == 0ull) {
            // the common case: hand the (at most two) hits of every lane to the owner of the group
            typedef int i2 __attribute__((ext_vector_type(2)));
            i2 *stage2 = reinterpret_cast<i2 *>(stage);
            const unsigned long long anyhit = __ballot(nh > 0);
            if (anyhit == 0ull) return;   // wave-uniform
            if (nh > 0) stage2[t] = i2{ h0, h1 };
            // the owner reads OTHER lanes' slots: per thread these are provably different addresses, so nothing but a
            // full compiler barrier keeps the reads behind the write (the LDS itself executes a wave's accesses in order)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            if (sub == 0) {
                unsigned bitsq = static_cast<unsigned>(anyhit >> gbase) & (G >= 8 ? 0xffu : 0xfu);   // only lanes sub < 8 search
                while (bitsq) {
                    const int src = __builtin_ctz(bitsq);
                    bitsq &= bitsq - 1u;
                    const i2 v = stage2[t + src];
                    insert(v.x);
                    if (v.y >= 0) insert(v.y);
                }
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            return;
        }
        // a lane met more than two hits (dense data): walk again, every hit goes to the owner as it is found
        first_list(cur, cnext);
        next_list(cur, cnext);
        while (__ballot(cur >= 0) != 0ull) {   // wave-uniform
            int k = -1;
            if (cur >= 0) {
                k = visit(cur);
                next_list(cur, cnext);
            }
            const unsigned long long bal = __ballot(k >= 0);
            if (bal == 0ull) continue;  // wave-uniform
            if (k >= 0) stage[2 * t] = k;   // the same 8-byte slot per lane as the fast form (other waves may be in either)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (sub == 0) {
                unsigned bitsq = static_cast<unsigned>(bal >> gbase) & (G >= 8 ? 0xffu : 0xfu);   // only lanes sub < 8 search
                while (bitsq) {
                    const int src = __builtin_ctz(bitsq);
                    bitsq &= bitsq - 1u;
                    insert(stage[2 * (t + src)]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    };

    // ---------------- 3 + 4 + 5 over the cloud, one register segment (16384 points) at a time ----------------
    int nflush = 0;
    for (int seg0 = 0; seg0 < n; seg0 += kCellSeg) {
        if (seg0 > 0) {
            // every row already full: later segments hold larger indices only
            if (all_threads(!qlive || sub != 0 || myhits >= nsample)) break;
            typedef int i4 __attribute__((ext_vector_type(4)));
            for (int i = t; i < kCellHeads / 4; i += NT) reinterpret_cast<i4 *>(head)[i] = i4{ -1, -1, -1, -1 };
            if (t == 0) sh.nc = 0;
            load_points(std::integral_constant<int, 0>{}, std::integral_constant<int, kCellPPT>{}, seg0);
            __syncthreads();
        }
        add_points(std::integral_constant<int, kCellPPT>{}, std::integral_constant<bool, A4>{}, px, py, pz, A4 ? seg0 + 4 * t : seg0 + t, n);
        HF_STAMP(4);
        __syncthreads();
        HF_STAMP(5);
        if (stop == 1) return;
        const int nc = sh.nc;
        if (nc <= cap) {
            // common case: every candidate of the segment fits -> one search
            if (nc > 0) {
                HF_STAMP(6);
                if (stop == 2) return;
                search();
                HF_STAMP(7);
                ++nflush;
            }
        } else {
            // dense cloud: walk the segment in index order, 2048 points per flush (<= cap candidates)
            const int segend = min(n, seg0 + kCellSeg);
            for (int base = seg0; base < segend; base += kCellChunkSlots * kCellThreads) {
                __syncthreads();   // previous search done with the lists
                typedef int i4 __attribute__((ext_vector_type(4)));
                for (int i = t; i < kCellHeads / 4; i += NT) reinterpret_cast<i4 *>(head)[i] = i4{ -1, -1, -1, -1 };
                if (t == 0) sh.nc = 0;
                float cx[kCellChunkSlots], cy[kCellChunkSlots], cz[kCellChunkSlots];
#pragma unroll
                for (int s = 0; s < kCellChunkSlots; ++s) {
                    const int k = base + s * kCellThreads + t;
                    const P3 p = load_p3(rcloud, static_cast<unsigned>(k));
                    cx[s] = p.x; cy[s] = p.y; cz[s] = p.z;
                }
                __syncthreads();
                add_points(std::integral_constant<int, kCellChunkSlots>{}, std::false_type{}, cx, cy, cz, base + t, segend);
                __syncthreads();
                search();
                nflush += 2;   // rows may refer to overwritten slots: step 6 reads the cloud
                if (all_threads(!qlive || sub != 0 || myhits >= nsample)) break;
            }
        }
    }
    if (stop == 4) return;

    // ---------------- 6: every wave writes the rows of its own queries ----------------
    if (qlive && sub == 0) hits[qi] = min(myhits, nsample);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool from_lds = nflush == 1;                     // uniform: `cand` still holds every row's candidates
    const int qw0 = wave << (6 - glog);                    // first query slot of this wave
    const int nqw = min(64 >> glog, nq - qw0);             // its live queries (consecutive j)
    if (nqw <= 0) return;
    const size_t jbase = static_cast<size_t>(bb) * m + j0 + qw0;
    if (pts_cnt) {
        const __amdgpu_buffer_rsrc_t rc = make_rsrc(pts_cnt + jbase, static_cast<unsigned>(nqw) * 4u);
        store_i32<SM>(rc, lane, hits[qw0 + (lane < nqw ? lane : 0)]);   // lanes >= nqw: dropped by the range check
    }
    const int total_e = nqw * nsample;
    const __amdgpu_buffer_rsrc_t ri = make_rsrc(idx ? idx + jbase * nsample : nullptr, idx ? static_cast<unsigned>(total_e) * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(GROUP ? grouped + jbase * nsample * 3 : nullptr, GROUP ? static_cast<unsigned>(total_e) * 12u : 0u);
    const int slot_mask = (1 << kCellSlotBits) - 1;
    // coordinates of a row entry (-1: no hit, the row is all index 0), minus the centre
    auto coords = [&](int pk, const float4 &cq, float o[3]) {
        float vx, vy, vz;
        if (from_lds && pk >= 0) { const float4 s4 = cand[pk & slot_mask]; vx = s4.x; vy = s4.y; vz = s4.z; }
        else { const P3 s3 = load_p3(rcloud, static_cast<unsigned>(pk < 0 ? 0 : (pk >> kCellSlotBits))); vx = s3.x; vy = s3.y; vz = s3.z; }
        if (center) { vx = vx - cq.x; vy = vy - cq.y; vz = vz - cq.z; }
        o[0] = vx; o[1] = vy; o[2] = vz;
    };
    // one (query, column) pair per lane and pass: every store instruction of the wave covers one contiguous range
    // (256 bytes of idx, 768 bytes of grouped_xyz) -> whole lines leave the chip, nothing to merge
    for (int e0 = lane; e0 < total_e; e0 += 256) {
        // four elements per lane: all the LDS reads first, then the stores
        int pk[4];
        float v[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = e0 + 64 * i;
            const int ee = e < total_e ? e : lane;
            const int q = ns_shift >= 0 ? (ee >> ns_shift) : (ee / nsample);
            const int c = ee - q * nsample;
            const int h = hits[qw0 + q];
            pk[i] = h == 0 ? -1 : rows[(qw0 + q) * rs + (c < h ? c : 0)];
            if (GROUP) coords(pk[i], qbuf[qw0 + q], v[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = e0 + 64 * i;   // e >= total_e: dropped by the range check of the descriptor
            if (idx) store_i32<SM>(ri, e, pk[i] < 0 ? 0 : (pk[i] >> kCellSlotBits));
            if (GROUP) store_p3<SM>(rg, e, v[i][0], v[i][1], v[i][2]);
        }
    }
    HF_STAMP(8);
    HF_STAMP_LAST(11);
    HF_STAMP_REAL(13, blockDim.x - 1);
}

static size_t cell_lds_bytes(int nsample, int qpw, int cap, int nt)
{
    const int rs = ((nsample + 3) & ~3) + 4;
    return 64 + sizeof(unsigned) * kCellWords + sizeof(int) * kCellHeads + sizeof(float4) * (static_cast<size_t>(qpw) + cap) +
           sizeof(int) * (static_cast<size_t>(qpw) * rs + qpw + 2 * nt);
}

template <bool GRP, int SM, int NT, bool A4>
static int cell_launch(dim3 grid, size_t lds, hipStream_t st, int n, int m, int qpw, int glog, float radius, float thresh,
                        float inv_cs, int nsample, int ns_shift, int cap, int stop, const float *xyz1, const float *xyz2,
                        int center, int *idx, int *pts_cnt, float *grouped)
{
    static int static_lds = -1;
    if (static_lds < 0) {
        hipFuncAttributes fa;
        static_lds = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&qbp_cell_kernel<GRP, SM, NT, A4>)) == hipSuccess
                         ? static_cast<int>(fa.sharedSizeBytes) : 1;
    }
    if (static_lds != 0) return HF_EINVAL;   // lds_u32 assumes the dynamic region starts at LDS address 0
    const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(&qbp_cell_kernel<GRP, SM, NT, A4>), lds);
    if (lrc != HF_OK) return lrc;
    hipLaunchKernelGGL((qbp_cell_kernel<GRP, SM, NT, A4>), grid, dim3(NT), lds, st, n, m, qpw, glog, radius, thresh, inv_cs,
                       nsample, ns_shift, cap, stop, xyz1, xyz2, center, idx, pts_cnt, grouped);
    return launch_status();
}

// returns HF_EINVAL when the shape is outside the cell kernel's range (the caller then takes an older kernel)
int launch_ball_query_cell(int b, int n, int m, float radius, float thresh, int nsample, const float *xyz1,
                           const float *xyz2, int center, int *idx, int *pts_cnt, float *grouped, hipStream_t st)
{
    if (nsample > 128 || !(radius < 3.0e18f) || !(radius > 1.0e-18f) || b > 65535 || n > (1 << kCellIdxBits)) return HF_EINVAL;
    // workgroup size: 1024 threads, 16 points each.  HF_QBP_NT=512 selects the 8-wave form with 32 points per thread
    // (diagnostics only: measured slower, profiles/r02_qbp_cell_notes.md)
    const int nt = HF_DIAG_INT("HF_QBP_NT", 1024) == 512 ? 512 : 1024;
    // queries per workgroup: rows of nsample ints in LDS; G = nt / qpw lanes per query, 4 <= G <= 64
    int qpw = nsample <= 32 ? 128 : (nsample <= 64 ? 64 : 32);
    // batched launches (the clouds of a geometry group): 256 queries per workgroup halve the per-query share of the cloud
    // scan as soon as that still fills every CU (80 clouds: 91.8 -> 69.1 us; 8 clouds: 11.7 -> 16.4 us, so not there)
    if (nsample <= 32 && nt == 1024 && static_cast<long long>(b) * div_up(m, 256) >= kNumCU) qpw = 256;
    while (qpw > nt / 64 && static_cast<long long>(b) * div_up(m, qpw) < kNumCU) qpw >>= 1;
    qpw = HF_DIAG_INT("HF_QBP_QPW", qpw);            // diagnostics only
    if (qpw < nt / 64 || qpw > 256 || qpw > nt / 4 || (qpw & (qpw - 1))) return HF_EINVAL;
    int glog = 0;
    while ((nt >> glog) > qpw) ++glog;
    const int cap = HF_DIAG_INT("HF_QBP_CAP", 2048);  // diagnostics only (>= 2048: one dense chunk always fits)
    if (cap < kCellChunk || cap >= (1 << kCellSlotBits)) return HF_EINVAL;
    if (div_up(m, qpw) > 65535) return HF_EINVAL;
    const int stop = HF_DIAG_INT("HF_QBP_STOP", 0);   // diagnostics only: early exit after a phase (outputs invalid)
    // stores: write-through when the whole grid is one round of workgroups (11.07 vs 11.27 us at 8 clouds), non-temporal
    // for larger grids (69.6 vs 71.9 us at 80 clouds); HF_QBP_STORE = 0 plain / 1 non-temporal / 2 write-through: diagnostics
    const int sm = HF_DIAG_INT("HF_QBP_STORE", static_cast<long long>(b) * div_up(m, qpw) <= kNumCU ? 2 : 1);
    const size_t lds = cell_lds_bytes(nsample, qpw, cap, nt);
    const float inv_cs = 1.0f / (2.2f * radius);   // cell width 2.2 radius >= 2 (radius + pad)
    int ns_shift = -1;
    if ((nsample & (nsample - 1)) == 0) { ns_shift = 0; while ((1 << ns_shift) < nsample) ++ns_shift; }
    dim3 grid(b, div_up(m, qpw));
    // four consecutive points per thread through 16-byte loads when every cloud starts on a 16-byte boundary
    // (HF_QBP_A4=0: diagnostics, the 12-byte form)
    const bool a4 = (n % 4 == 0) && (reinterpret_cast<uintptr_t>(xyz1) % 16 == 0) && HF_DIAG_INT("HF_QBP_A4", 1) != 0;
#define HF_CELL_ARGS grid, lds, st, n, m, qpw, glog, radius, thresh, inv_cs, nsample, ns_shift, cap, stop, xyz1, xyz2, center, idx, pts_cnt, grouped
#define HF_CELL_DISPATCH(GRP, SMODE)                                                                                   \
    do {                                                                                                              \
        if (nt == 1024) return a4 ? cell_launch<GRP, SMODE, 1024, true>(HF_CELL_ARGS) : cell_launch<GRP, SMODE, 1024, false>(HF_CELL_ARGS); \
        else return cell_launch<GRP, SMODE, 512, false>(HF_CELL_ARGS);                                                \
    } while (0)
    if (grouped) {
        if (sm == 0) HF_CELL_DISPATCH(true, 0);
        else if (sm == 2) HF_CELL_DISPATCH(true, 2);
        else HF_CELL_DISPATCH(true, 1);
    } else {
        if (sm == 0) HF_CELL_DISPATCH(false, 0);
        else if (sm == 2) HF_CELL_DISPATCH(false, 2);
        else HF_CELL_DISPATCH(false, 1);
    }
#undef HF_CELL_DISPATCH
#undef HF_CELL_ARGS
}

}  // namespace hf

#ifdef HF_QBP_STAMPS
extern "C" __attribute__((visibility("default"))) int hf_debug_qbp_stamps(unsigned long long *host, int nwg)
{
    return static_cast<int>(hipMemcpyFromSymbol(host, HIP_SYMBOL(hf::g_qbp_stamps), sizeof(unsigned long long) * 16 * nwg));
}
#endif
