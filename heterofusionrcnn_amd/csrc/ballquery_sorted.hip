// ballquery_sorted.hip -- query_ball_point (+ fused group_point on xyz) for gfx950, the cell-sorted path: the cell structure
// of a cloud is built ONCE per (cloud, radius) and shared by all its query tiles.
//
// Replaces grouping/tf_grouping_g.cu:3-36 (query_ball_point_gpu) and, fused, :40-57 (group_point_gpu on the coordinates)
// plus the `grouped_xyz -= new_xyz` of pointnet_util.py:56, like ballquery.hip, with the same semantics bit for bit:
// hit iff s < T (T exact on the host, grouping.hip:ball_threshold), s = dx^2 + dy^2 + dz^2 without FMA; the row holds the
// first nsample hits in ascending data index, padded with the first; pts_cnt = min(hits, nsample); no hit -> zeros.
//
// Why a second path.  The single-launch kernel (ballquery.hip) lets every 128- or 256-query tile re-hash the whole cloud
// (16384 points, 196 KB from L2) to find its ~600 candidates: 16 .. 32 scans per cloud.  That is the right trade for one
// round of workgroups (8 clouds: nothing to amortise a second launch over), not for the batched launches of the train step
// (the 80 clouds of a geometry group), where the redundant scans are most of the time.  Here:
//   bq_build_kernel   one workgroup per cloud: a counting sort of the points by hashed cell (cell = round(v / cs) per axis in a
//                     FIXED frame, cs = 2.2 radius, read off the mantissa of one fma; bucket = (cy + 73 cx + 1187 cz) mod H,
//                     H = 1024 .. 16384 by cloud size).  LDS histogram with returning atomics (the return value is the rank
//                     inside the bucket), one block scan, then every point goes to start[bucket] + rank as (x, y, z, index).
//                     Output in the caller's workspace: sorted float4[n] + start u32[H + 1] per cloud.
//   bq_query_kernel   8 lanes per query, 8 queries per wave, 32 per workgroup (no workgroup barrier at all: a wave owns its
//                     queries from the first load to the last store).  Lane c of a query reads the bucket range of cell c of
//                     the <= 8 cells its padded box [q - rp, q + rp] touches (cs >= 2 rp: <= 2 cells per axis; the 8 buckets of
//                     a query are distinct by construction of the hash), tests up to four candidates per trip with the
//                     reference's exact fp32 expression, and hands its hits to the query's first lane, which keeps the
//                     nsample SMALLEST data indices sorted in an LDS row (= the first nsample hits of the reference's
//                     ascending scan; the order inside a bucket does not matter).  Then the wave writes its 8 rows: every
//                     store instruction covers one contiguous range (256 B of idx, 768 B of grouped_xyz).
// A query whose coordinates are so large against the radius that its padded box could span three cells walks the whole
// sorted array instead (every point a candidate: correct, slow, never seen on real clouds).
// Results depend neither on the hash, nor on the arrival order of the atomics: the hit set is decided by `s < T` on every
// candidate and the order by the data index alone.
#include <math.h>

#include <type_traits>

#include "bq_common.h"

namespace hf {

constexpr int kSortThreads = 1024;
constexpr unsigned kSortHashX = 73u, kSortHashZ = 1187u;   // bucket = (cy + 73 cx + 1187 cz) mod H: the 8 subset sums of
                                                            // {1, 73, 1187} are distinct mod every H >= 1024
constexpr int kQueryThreads = 256;
constexpr int kHitCap = 8;             // hits per query kept with their coordinates in LDS (the few-hit path of the query kernel)
constexpr int kPackedMaxN = 32768;     // up to here: table entry = first | count << 16, row entry = index << 16 | position
constexpr int kPackBits = 16;          // row entry = data index << 16 | position in the sorted array (n <= 32768), else the index

__host__ __device__ inline size_t bq_sorted_stride(int n, int hbits)
{
    return (16u * static_cast<size_t>(n) + 4u * ((static_cast<size_t>(1) << hbits) + 4u) + 15u) / 16u * 16u;
}

static int bq_hbits(int n) { return n <= 1024 ? 10 : (n <= 4096 ? 12 : 14); }

// cell of v along one axis = round(v / cs), read off the mantissa of fma(v, 1/cs, 1.5 * 2^23): one instruction, monotone
// in v; exact while |v / cs| < 2^22 (queries beyond 2^19 cells take the exhaustive walk)
__device__ __forceinline__ unsigned cell_of(float v, float inv_cs) { return __float_as_uint(__builtin_fmaf(v, inv_cs, 12582912.0f)); }
// only the low hbits <= 14 bits of the key are used; they depend on the low 14 bits of the operands alone, so the raw float
// bits go into the 24-bit multiply-adds
__device__ __forceinline__ unsigned cell_key(unsigned cx, unsigned cy, unsigned cz)
{
    return mad_u24(cz, kSortHashZ, mad_u24(cx, kSortHashX, cy));
}

// One cloud is sorted by S workgroups, each owning a contiguous range of HS = H / S buckets: every workgroup reads the whole
// cloud (the keys of all points tell it how many fall below its range = where its part of the sorted array starts), but
// only the points of its own buckets go through the histogram, the scan and the scatter.  No communication between the
// workgroups.  PPT > 0: n <= PPT * 1024, the points stay in registers between the histogram (returning LDS atomics: the
// return value is the rank inside the bucket) and the scatter, which goes through LDS in pieces of kSortChunk entries so that
// the sorted array leaves in whole lines (a workgroup that scatters 16 bytes per request is bound by its request rate: 16 us
// for a 16384-point cloud); PPT = 0: any n, the cloud is read twice and scattered through cursor atomics (any order inside a
// bucket is a valid one).  The scan: a wave per contiguous segment of counters, lane-consecutive words, DPP scans, the carry
// in a scalar (a thread-per-16-counters scan cost 7.8 us in bank conflicts and partial-line stores).
// Measured alternatives (profiles/r03_bq_sorted_notes.md): own points parked in LDS instead of registers, 8 workgroups per
// cloud, three per CU: 23 us for 80 clouds against 16 us (every workgroup still hashes all 16384 points, and the extra
// ballots / slot atomics cost more than the registers saved).
constexpr int kSortChunk = 6016;      // 64 KB of counters (one workgroup per cloud) + 128 B + 94 KB of chunk < 160 KB of LDS
template <int PPT, int HB, int S>
__global__ __launch_bounds__(kSortThreads) void bq_build_kernel(int n, int packed, int stop, float inv_cs, const float *__restrict__ xyz1,
                                                                 unsigned char *__restrict__ ws, size_t ws_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned bq_hist[];   // HS counters | 32 ints (scans) | chunk float4[kSortChunk]
    constexpr int H = 1 << HB, HS = H / S;
    static_assert(HS >= kSortThreads && (S & (S - 1)) == 0 && HS * 4 <= kSortChunk * 16, "64 counters per wave and trip; counts fit the chunk area");
    const int t = threadIdx.x, bb = blockIdx.x, sidx = blockIdx.y;
    const int lane = t & 63, wave = t >> 6;
    int *wsum = reinterpret_cast<int *>(bq_hist + HS);
    float4 *chunk = reinterpret_cast<float4 *>(bq_hist + HS + 32);
    auto chunk_counts = [&](int a) -> unsigned & { return reinterpret_cast<unsigned *>(chunk)[a]; };   // scratch until the scatter
    const __amdgpu_buffer_rsrc_t rcloud = make_rsrc(xyz1 + static_cast<size_t>(bb) * n * 3, static_cast<unsigned>(n) * 12u);
    float4 *sorted = reinterpret_cast<float4 *>(ws + static_cast<size_t>(bb) * ws_stride);
    unsigned *start = reinterpret_cast<unsigned *>(sorted + n);
    for (int i = t; i < HS; i += kSortThreads) bq_hist[i] = 0u;
    __syncthreads();
    constexpr int NP = PPT > 0 ? PPT : 1;
    float px[NP], py[NP], pz[NP];
    unsigned kr[NP];                      // my points of this workgroup's buckets: local bucket | rank inside it << 16; else ~0
    int nbelow = 0;                       // my points that fall into the buckets of workgroups 0 .. sidx - 1
    auto key_of = [&](const P3 &p) -> unsigned { return cell_key(cell_of(p.x, inv_cs), cell_of(p.y, inv_cs), cell_of(p.z, inv_cs)) & (H - 1); };
    if constexpr (PPT > 0) {
        static_assert(PPT * kSortThreads <= 65536 && HS <= 65536, "bucket and rank share a word");
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const P3 p = load_p3(rcloud, static_cast<unsigned>(u * kSortThreads + t));   // past the end: zeros, masked below
            px[u] = p.x; py[u] = p.y; pz[u] = p.z;
        }
        if (stop == 1) { if (px[0] == 1.2345e-30f) start[0] = 0u; return; }   // loads only
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const unsigned key = key_of(P3{ px[u], py[u], pz[u] });
            const int owner = static_cast<int>(key / HS);
            kr[u] = 0xffffffffu;
            if (u * kSortThreads + t < n) {
                if (owner < sidx) ++nbelow;
                else if (owner == sidx) kr[u] = (key & (HS - 1)) | (atomicAdd(&bq_hist[key & (HS - 1)], 1u) << 16);
            }
        }
    } else {
        for (int k = t; k < n; k += kSortThreads) {
            const unsigned kk = key_of(load_p3(rcloud, static_cast<unsigned>(k)));
            const int owner = static_cast<int>(kk / HS);
            if (owner < sidx) ++nbelow;
            else if (owner == sidx) atomicAdd(&bq_hist[kk & (HS - 1)], 1u);
        }
    }
    __syncthreads();
    if (stop == 2) { if (nbelow == -7) start[0] = 0u; return; }             // + keys and histogram atomics
    // exclusive scan of my HS counters.  Wave w owns the contiguous segment [w SEG, (w + 1) SEG): lane-consecutive words per
    // access (no bank conflicts, whole lines to the bucket table), a DPP scan per 64 counters, the carry in a scalar.
    constexpr int SEG = HS / 16, IT = SEG / 64;
    unsigned carry = 0u;
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {
        const int a = wave * SEG + i * 64 + lane;
        const unsigned cnt = bq_hist[a];
        const unsigned inc = static_cast<unsigned>(wave_inclusive_scan_i32(static_cast<int>(cnt)));
        bq_hist[a] = carry + inc - cnt;                             // exclusive, relative to the wave's segment
        chunk_counts(a) = cnt;
        carry += static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(inc), 63));
    }
    if (lane == 0) wsum[wave] = static_cast<int>(carry);
    int lo = 0;
    (void)block_exclusive_scan(nbelow, wsum + 16, &lo);             // lo = where my part of the sorted array starts (2 barriers)
    unsigned base = 0u;
    int own_count = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const unsigned x = static_cast<unsigned>(wsum[w]);
        if (w < wave) base += x;
        own_count += static_cast<int>(x);
    }
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {
        const int a = wave * SEG + i * 64 + lane;
        const unsigned ex = bq_hist[a] + base;
        bq_hist[a] = ex;                                            // relative to lo
        // packed (n <= 32768): first entry | entries << 16 -- one load per cell in the query kernel instead of two
        start[sidx * HS + a] = (static_cast<unsigned>(lo) + ex) | (packed ? chunk_counts(a) << 16 : 0u);
    }
    if (sidx == S - 1 && t == kSortThreads - 1) start[H] = static_cast<unsigned>(n);
    __syncthreads();
    if (stop == 3) return;                                                  // + scans and the bucket table
    if constexpr (PPT > 0) {
        // (bucket, rank) -> position inside my part
#pragma unroll
        for (int u = 0; u < PPT; ++u)
            if (kr[u] != 0xffffffffu) kr[u] = bq_hist[kr[u] & 0xffffu] + (kr[u] >> 16);
        __syncthreads();                                                    // the counts' scratch copy lives in the chunk area
        for (int cb = 0; cb < own_count; cb += kSortChunk) {   // uniform
#pragma unroll
            for (int u = 0; u < PPT; ++u) {
                const unsigned rel = kr[u] - static_cast<unsigned>(cb);   // 0xffffffff - cb stays >= kSortChunk
                if (rel < static_cast<unsigned>(kSortChunk))
                    chunk[rel] = make_float4(px[u], py[u], pz[u], __int_as_float(u * kSortThreads + t));
            }
            __syncthreads();
            if (stop == 4) return;                                          // + first chunk staged in LDS
            const int cnt = min(kSortChunk, own_count - cb);
            for (int i = t; i < cnt; i += kSortThreads) sorted[lo + cb + i] = chunk[i];
            __syncthreads();
        }
    } else {
        for (int k = t; k < n; k += kSortThreads) {
            const P3 p = load_p3(rcloud, static_cast<unsigned>(k));
            const unsigned kk = key_of(p);
            if (static_cast<int>(kk / HS) == sidx)
                sorted[static_cast<unsigned>(lo) + atomicAdd(&bq_hist[kk & (HS - 1)], 1u)] = make_float4(p.x, p.y, p.z, __int_as_float(k));
        }
    }
}

// G lanes per query (8, 4 or 2: 8 / G cells per lane), 64 / G queries per wave, kQueryThreads / G per workgroup.  Fewer lanes
// per query = more queries in flight per CU: the kernel is a chain of dependent memory round trips (query -> bucket bounds ->
// candidates -> coordinates -> stores), so what fills the machine is the number of independent chains, not lanes per chain;
// the loads of a lane's cells are issued together.
template <bool GROUP, int SM, int G>
__global__ __launch_bounds__(kQueryThreads) void bq_query_kernel(int n, int m, int hbits, float radius, float thresh, float inv_cs,
                                                                  int nsample, int ns_shift, int pshift,
                                                                  const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                                  int center, const unsigned char *__restrict__ ws,
                                                                  size_t ws_stride, int *__restrict__ idx,
                                                                  int *__restrict__ pts_cnt, float *__restrict__ grouped)
{
    extern __shared__ __attribute__((aligned(16))) int bq_smem[];
    constexpr int CPL = 8 / G;                                          // cells per lane
    constexpr int QPW = kQueryThreads / G;                              // queries per workgroup
    constexpr int QPV = 64 / G;                                         // queries per wave
    constexpr int GLOG = G == 8 ? 3 : (G == 4 ? 2 : 1);
    const int rs = ((nsample + 3) & ~3) + 4;                            // 16-byte rows, stride = 4 banks mod 32
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 *stage4 = reinterpret_cast<u4 *>(bq_smem);                       // 2 * threads   hand-off inside a wave: (x, y, z, index) per hit
    int *stage = bq_smem;                                               //               ... or one / two row entries per lane
    u4 *hl = stage4 + 2 * kQueryThreads;                                // QPW * kHitCap   the hits of a query, ascending index (few-hit path)
    int *rows = reinterpret_cast<int *>(hl + QPW * kHitCap);            // QPW * rs   nsample smallest entries, ascending (general path)
    int *hits = rows + QPW * rs;                                        // QPW        min(total hits, nsample)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int sub = t & (G - 1), gbase = lane & ~(G - 1);
    // grid = (clouds, query tiles): linear workgroup id = cloud + b * tile; workgroups are dealt round-robin over the 8 XCDs,
    // so with b a multiple of 8 all tiles of a cloud (and its build workgroups) share ONE XCD's L2 (speed only)
    const int bb = blockIdx.x, j0 = blockIdx.y * QPW;
    const int nq = min(QPW, m - j0);
    const int qi = t >> GLOG;
    const bool qlive = qi < nq;
    const float *p2 = xyz2 + (static_cast<size_t>(bb) * m + j0) * 3;
    const __amdgpu_buffer_rsrc_t rcloud = make_rsrc(xyz1 + static_cast<size_t>(bb) * n * 3, static_cast<unsigned>(n) * 12u);
    const unsigned char *wsb = ws + static_cast<size_t>(bb) * ws_stride;
    const __amdgpu_buffer_rsrc_t rsorted = make_rsrc(wsb, static_cast<unsigned>(n) * 16u);
    const unsigned *start = reinterpret_cast<const unsigned *>(wsb + static_cast<size_t>(n) * 16u);

    const P3 q = load_p3(make_rsrc(p2, static_cast<unsigned>(nq) * 12u), static_cast<unsigned>(qi));   // qi >= nq: zeros
    const float qx = q.x, qy = q.y, qz = q.z;
    // the bucket ranges of my cells: cell c = x side | y side << 1 | z side << 2 of the padded box; lane `sub` takes cells
    // sub, sub + G, ...
    int s[CPL], s1[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) { s[j] = 0; s1[j] = 0; }
    {
        const float qmax = fmaxf(fabsf(qx), fmaxf(fabsf(qy), fabsf(qz)));
        const float rp = radius * 1.001f + 1e-6f * qmax;                // pad: > radius plus the fp32 rounding of q -/+ rp
        const unsigned lx = cell_of(qx - rp, inv_cs), ly = cell_of(qy - rp, inv_cs), lz = cell_of(qz - rp, inv_cs);
        const unsigned hx = cell_of(qx + rp, inv_cs) - lx, hy = cell_of(qy + rp, inv_cs) - ly, hz = cell_of(qz + rp, inv_cs) - lz;
        // cs >= 2 rp with slack: 0 <= h <= 1.  Otherwise (coordinates ~1e5 radii away from the origin) -> exhaustive walk
        const bool exh = 2.0f * rp * inv_cs > 0.995f || ((hx | hy | hz) & ~1u) || !(qmax * inv_cs < 524288.0f);
        if (qlive) {
            if (exh) {
                const int piece = (n + G - 1) / G;                      // cell slot 0 of every lane walks a G-th of the array
                s[0] = min(n, sub * piece);
                s1[0] = min(n, s[0] + piece);
            } else {
                const unsigned hmask = (hx & 1u) | ((hy & 1u) << 1) | ((hz & 1u) << 2);
                const unsigned base = cell_key(lx, ly, lz);
                unsigned lo[CPL], hi[CPL];
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    const unsigned c = static_cast<unsigned>(sub + j * G);
                    const unsigned key = (base + (c & 1u) * kSortHashX + ((c >> 1) & 1u) + ((c >> 2) & 1u) * kSortHashZ) & ((1u << hbits) - 1u);
                    const bool on = (c & ~hmask) == 0u;
                    lo[j] = on ? start[key] : 0u;
                    hi[j] = (on && !pshift) ? start[key + 1] : 0u;
                }
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    s[j] = static_cast<int>(pshift ? (lo[j] & 0xffffu) : lo[j]);
                    s1[j] = static_cast<int>(pshift ? (lo[j] & 0xffffu) + (lo[j] >> 16) : hi[j]);
                }
            }
        }
    }
    int s0[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) s0[j] = s[j];
    // exact test of one sorted entry
    auto is_hit = [&](const u4 &c) -> bool {
        const float dx = qx - __uint_as_float(c.x), dy = qy - __uint_as_float(c.y), dz = qz - __uint_as_float(c.z);
        const float s2 = dx * dx + dy * dy + dz * dz;
        return s2 < thresh;
    };
    // row entry of the general path: index << 16 | position in the sorted array (n <= 32768), else the index
    auto entry_of = [&](const u4 &c, int pos) -> int {
        const int k = static_cast<int>(c.w);
        return pshift ? ((k << kPackBits) | pos) : k;
    };
    auto test = [&](const u4 &c, int pos) -> int { return is_hit(c) ? entry_of(c, pos) : -1; };

    int myhits = 0, mylen = 0;        // owner lane (sub == 0): total hits of my query, entries in its row
    int *const row = rows + qi * rs;
    auto insert = [&](int kk) {       // one more hit: keep the nsample smallest (sorted ascending)
        ++myhits;
        if (mylen < nsample || kk < row[mylen - 1]) {
            int pos = mylen < nsample ? mylen++ : mylen - 1;
            while (pos > 0 && row[pos - 1] > kk) { row[pos] = row[pos - 1]; --pos; }
            row[pos] = kk;
        }
    };
    auto pending = [&]() -> bool {
        bool p = false;
#pragma unroll
        for (int j = 0; j < CPL; ++j) p = p || (s[j] < s1[j]);
        return p;
    };
    // ---- walk: up to 4 / CPL .. 4 candidates of every cell per trip; every lane keeps its first two hits in registers ----
    constexpr int PER = CPL >= 4 ? 2 : 4;      // candidates per cell and trip (the loads of a trip are all in flight together)
    u4 hv0 = { 0u, 0u, 0u, 0u }, hv1 = { 0u, 0u, 0u, 0u };
    int hp0 = 0, hp1 = 0, nh = 0;
    while (__ballot(pending()) != 0ull) {   // wave-uniform
        u4 c[CPL][PER];
#pragma unroll
        for (int j = 0; j < CPL; ++j)
#pragma unroll
            for (int i = 0; i < PER; ++i)
                if (s[j] + i < s1[j]) c[j][i] = __builtin_amdgcn_raw_buffer_load_b128(rsorted, (s[j] + i) * 16, 0, 0);
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                if (s[j] + i < s1[j]) {
                    if (is_hit(c[j][i])) {
                        if (nh == 0) { hv0 = c[j][i]; hp0 = s[j] + i; } else if (nh == 1) { hv1 = c[j][i]; hp1 = s[j] + i; }
                        ++nh;
                    }
                }
            }
            s[j] += PER;
        }
    }
    const int qw0 = wave * QPV;                            // first query slot of this wave
    const int nqw = min(QPV, nq - qw0);                    // its live queries (consecutive j)
    const size_t jbase = static_cast<size_t>(bb) * m + j0 + qw0;
    const int total_e = nqw * nsample;
    const __amdgpu_buffer_rsrc_t ri = make_rsrc(idx ? idx + jbase * nsample : nullptr, idx ? static_cast<unsigned>(max(total_e, 0)) * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(GROUP ? grouped + jbase * nsample * 3 : nullptr, GROUP ? static_cast<unsigned>(max(total_e, 0)) * 12u : 0u);
    const unsigned long long few = __ballot(nh > 2);
    if (few == 0ull) {
        // ---- few-hit path (the usual case): every hit travels with its coordinates through LDS to the owner lane of its query,
        // which keeps up to kHitCap of them sorted by index; the rows are then written from LDS alone ----
        const unsigned long long anyhit = __ballot(nh > 0);
        // the hits travel ALREADY CENTRED (x - qx, y - qy, z - qz, index): the lane that found a hit holds its query's centre, and the
        // same fp32 subtraction done here once per hit is not done once per output element in the store loop below (which was
        // 260 of the wave's ~400 vector instructions: centre fetched through three cross-lane reads and subtracted per element,
        // a select against the empty-row fallback per component -- the kernel is vector-issue-bound, profiles/r04_bq_query_pmc.txt)
        auto centred = [&](u4 v) -> u4 {
            if (GROUP && center) {
                v.x = __float_as_uint(__uint_as_float(v.x) - qx);
                v.y = __float_as_uint(__uint_as_float(v.y) - qy);
                v.z = __float_as_uint(__uint_as_float(v.z) - qz);
            }
            return v;
        };
        if (nh > 0) { stage4[2 * t] = centred(hv0); if (nh > 1) stage4[2 * t + 1] = centred(hv1); }
        if (nh > 0) stage[8 * t + 7] = nh > 1 ? static_cast<int>(hv1.w) : -1;    // .w of the second slot: -1 = one hit only
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        int total = 0;
        if (sub == 0 && qlive) {
            u4 *const mine = hl + qi * kHitCap;
            unsigned bitsq = static_cast<unsigned>(anyhit >> gbase) & ((1u << G) - 1u);
            while (bitsq) {
                const int src = __builtin_ctz(bitsq);
                bitsq &= bitsq - 1u;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const u4 v = stage4[2 * (t + src) + w];
                    if (w == 1 && static_cast<int>(v.w) < 0) break;
                    if (total < kHitCap) {
                        int pos = total;
                        while (pos > 0 && static_cast<int>(mine[pos - 1].w) > static_cast<int>(v.w)) { mine[pos] = mine[pos - 1]; --pos; }
                        mine[pos] = v;
                    }
                    ++total;
                }
            }
            hits[qi] = min(total, nsample);
            if (GROUP && total == 0) {
                // a row without hits is index 0 everywhere: its coordinates are point 0 of the cloud (minus the centre); filed as the
                // row's entry 0, so that the store loop needs no special case (rare: a query that is no point of its own cloud)
                const P3 p0 = load_p3(rcloud, 0u);
                mine[0] = centred(u4{ __float_as_uint(p0.x), __float_as_uint(p0.y), __float_as_uint(p0.z), 0u });
            } else if (!GROUP && total == 0) {
                mine[0] = u4{ 0u, 0u, 0u, 0u };
            }
        }
        if (__ballot(total > kHitCap) == 0ull) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            if (nqw <= 0) return;
            if (pts_cnt) {
                const __amdgpu_buffer_rsrc_t rc = make_rsrc(pts_cnt + jbase, static_cast<unsigned>(nqw) * 4u);
                store_i32<SM>(rc, lane, hits[qw0 + (lane < nqw ? lane : 0)]);   // lanes >= nqw: dropped by the range check
            }
            // four elements per lane and pass: the LDS reads first, then the stores; every store instruction of the wave covers one
            // contiguous range (256 bytes of idx, 768 bytes of grouped_xyz).  POW2: nsample is a power of two (every shipped
            // config): element -> (query, column) is a shift and a mask, decided once for the loop
            auto write_rows = [&](auto pow2_tag) {
                constexpr bool POW2 = decltype(pow2_tag)::value;
                for (int e0 = lane; e0 - lane < total_e; e0 += 256) {
                    u4 hv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = e0 + 64 * i;
                        const int ee = e < total_e ? e : 0;
                        const int qq = POW2 ? (ee >> ns_shift) : (ee / nsample);
                        const int col = POW2 ? (ee & (nsample - 1)) : (ee - qq * nsample);
                        const int h = max(hits[qw0 + qq], 1);                   // an empty row reads its entry 0 (point 0, index 0)
                        hv[i] = hl[(qw0 + qq) * kHitCap + (col < h ? col : 0)];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = e0 + 64 * i;   // e >= total_e: dropped by the range check of the descriptor
                        if (idx) store_i32<SM>(ri, e, static_cast<int>(hv[i].w));
                        if (GROUP) store_p3<SM>(rg, e, __uint_as_float(hv[i].x), __uint_as_float(hv[i].y), __uint_as_float(hv[i].z));
                    }
                }
            };
            if (ns_shift >= 0) write_rows(std::true_type{}); else write_rows(std::false_type{});
            return;
        }
        // some query of this wave has more than kHitCap hits: the general path below (rows of entries; the staging area is
        // reused, every lane of the wave is past its reads of it)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    }
    if (few == 0ull) {
        // at most two hits per lane: hand their row entries to the owner of the group
        typedef int i2 __attribute__((ext_vector_type(2)));
        i2 *stage2 = reinterpret_cast<i2 *>(stage);
        const unsigned long long anyhit = __ballot(nh > 0);
        if (anyhit != 0ull) {   // wave-uniform
            if (nh > 0) stage2[4 * t] = i2{   /* inside this lane's own 32 bytes of the staging area: other waves may be in the few-hit path */  entry_of(hv0, hp0), nh > 1 ? entry_of(hv1, hp1) : -1 };
            // the owner reads OTHER lanes' slots: nothing but a full compiler barrier keeps the reads behind the write (the
            // LDS itself executes a wave's accesses in order)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            if (sub == 0) {
                unsigned bitsq = static_cast<unsigned>(anyhit >> gbase) & ((1u << G) - 1u);
                while (bitsq) {
                    const int src = __builtin_ctz(bitsq);
                    bitsq &= bitsq - 1u;
                    const i2 v = stage2[4 * (t + src)];
                    insert(v.x);
                    if (v.y >= 0) insert(v.y);
                }
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        // a lane met more than two hits (dense data): walk again, every hit goes to the owner as it is found
#pragma unroll
        for (int j = 0; j < CPL; ++j) s[j] = s0[j];
        while (__ballot(pending()) != 0ull) {   // wave-uniform
            int k = -1;
            bool done = false;
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                if (!done && s[j] < s1[j]) {
                    const u4 c = __builtin_amdgcn_raw_buffer_load_b128(rsorted, s[j] * 16, 0, 0);
                    k = test(c, s[j]);
                    ++s[j];
                    done = true;
                }
            }
            const unsigned long long bal = __ballot(k >= 0);
            if (bal == 0ull) continue;   // wave-uniform
            if (k >= 0) stage[8 * t] = k;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (sub == 0) {
                unsigned bitsq = static_cast<unsigned>(bal >> gbase) & ((1u << G) - 1u);
                while (bitsq) {
                    const int src = __builtin_ctz(bitsq);
                    bitsq &= bitsq - 1u;
                    insert(stage[8 * (t + src)]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ---------------- every wave writes the rows of its own queries ----------------
    if (qlive && sub == 0) hits[qi] = min(myhits, nsample);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (nqw <= 0) return;
    if (pts_cnt) {
        const __amdgpu_buffer_rsrc_t rc = make_rsrc(pts_cnt + jbase, static_cast<unsigned>(nqw) * 4u);
        store_i32<SM>(rc, lane, hits[qw0 + (lane < nqw ? lane : 0)]);   // lanes >= nqw: dropped by the range check
    }
    for (int e0 = lane; e0 - lane < total_e; e0 += 256) {   // wave-uniform trip count: the centre comes from another lane
        // four elements per lane: all the reads first, then the stores; every store instruction of the wave covers one
        // contiguous range (256 bytes of idx, 768 bytes of grouped_xyz)
        int pk[4];
        float v[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = e0 + 64 * i;
            const int ee = e < total_e ? e : 0;
            const int qq = ns_shift >= 0 ? (ee >> ns_shift) : (ee / nsample);
            const int col = ee - qq * nsample;
            const int h = hits[qw0 + qq];
            pk[i] = h == 0 ? -1 : rows[(qw0 + qq) * rs + (col < h ? col : 0)];
            if (GROUP) {
                float vx, vy, vz;
                if (pshift && pk[i] >= 0) {
                    const u4 c = __builtin_amdgcn_raw_buffer_load_b128(rsorted, (pk[i] & ((1 << kPackBits) - 1)) * 16, 0, 0);
                    vx = __uint_as_float(c.x); vy = __uint_as_float(c.y); vz = __uint_as_float(c.z);
                } else {
                    const P3 s3 = load_p3(rcloud, static_cast<unsigned>(pk[i] < 0 ? 0 : pk[i]));
                    vx = s3.x; vy = s3.y; vz = s3.z;
                }
                if (center) {
                    // the centre of row qq: lane G qq of this wave holds it (all G lanes of a query do)
                    const float cx = __shfl(qx, (qq * G) & 63), cy = __shfl(qy, (qq * G) & 63), cz = __shfl(qz, (qq * G) & 63);
                    vx = vx - cx; vy = vy - cy; vz = vz - cz;
                }
                v[i][0] = vx; v[i][1] = vy; v[i][2] = vz;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = e0 + 64 * i;   // e >= total_e: dropped by the range check of the descriptor
            if (idx) store_i32<SM>(ri, e, pk[i] < 0 ? 0 : (pshift ? (pk[i] >> kPackBits) : pk[i]));
            if (GROUP) store_p3<SM>(rg, e, v[i][0], v[i][1], v[i][2]);
        }
    }
}

size_t ball_query_sorted_workspace(int b, int n)
{
    if (b <= 0 || n <= 0) return 0;
    return static_cast<size_t>(b) * bq_sorted_stride(n, bq_hbits(n));
}

template <int PPT, int HB, int S>
static int build_launch(int b, int n, float inv_cs, const float *xyz1, unsigned char *ws, size_t stride, hipStream_t st)
{
    const size_t lds = sizeof(unsigned) * ((static_cast<size_t>(1) << HB) / S + 32) + sizeof(float4) * kSortChunk;
    const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(&bq_build_kernel<PPT, HB, S>), lds);
    if (lrc != HF_OK) return lrc;
    // stop: diagnostic builds only (early exit after a phase, outputs invalid); the product passes the constant 0
    hipLaunchKernelGGL((bq_build_kernel<PPT, HB, S>), dim3(b, S), dim3(kSortThreads), lds, st, n, n <= kPackedMaxN ? 1 : 0,
                       HF_DIAG_INT("HF_BQ_STOP", 0), inv_cs, xyz1, ws, stride);
    return launch_status();
}

// returns HF_EINVAL when the shape is outside this path's range (the caller then takes another kernel)
int launch_ball_query_sorted(int b, int n, int m, float radius, float thresh, int nsample, const float *xyz1, const float *xyz2,
                             int center, int *idx, int *pts_cnt, float *grouped, void *workspace, size_t workspace_bytes,
                             hipStream_t st)
{
    if (nsample > 128 || !(radius < 3.0e18f) || !(radius > 1.0e-18f) || b > 65535 || n > (1 << 26)) return HF_EINVAL;
    if (!workspace || workspace_bytes < ball_query_sorted_workspace(b, n)) return HF_EWORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) % 16 != 0) return HF_EINVAL;
    const int hbits = bq_hbits(n);
    const size_t stride = bq_sorted_stride(n, hbits);
    const float inv_cs = 1.0f / (2.2f * radius);   // cell width 2.2 radius >= 2 (radius + pad)
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    int rc;
    // workgroups per cloud: as many as still fit ONE round of the chip (a 1024-thread workgroup with its registers owns a CU)
    const int split = HF_DIAG_INT("HF_BQ_SPLIT", b * 4 <= kNumCU ? 4 : (b * 2 <= kNumCU ? 2 : 1));
    if (n <= 1024) rc = build_launch<1, 10, 1>(b, n, inv_cs, xyz1, ws, stride, st);
    else if (n <= 4096) rc = split >= 2 ? build_launch<4, 12, 2>(b, n, inv_cs, xyz1, ws, stride, st) : build_launch<4, 12, 1>(b, n, inv_cs, xyz1, ws, stride, st);
    else if (n <= 16384) rc = split >= 4 ? build_launch<16, 14, 4>(b, n, inv_cs, xyz1, ws, stride, st)
                              : (split == 2 ? build_launch<16, 14, 2>(b, n, inv_cs, xyz1, ws, stride, st) : build_launch<16, 14, 1>(b, n, inv_cs, xyz1, ws, stride, st));
    else rc = build_launch<0, 14, 4>(b, n, inv_cs, xyz1, ws, stride, st);
    if (rc != HF_OK) return rc;
    int ns_shift = -1;
    if ((nsample & (nsample - 1)) == 0) { ns_shift = 0; while ((1 << ns_shift) < nsample) ++ns_shift; }
    const int pshift = n <= kPackedMaxN ? 1 : 0;
    // lanes per query: 8 in the product; HF_BQ_G = 4 / 2: diagnostic builds only
    const long long nq_total = static_cast<long long>(b) * m;
    (void)nq_total;
    const int g = HF_DIAG_INT("HF_BQ_G", 8);   // measured (profiles/r03_bq_sorted_notes.md): 4 and 2 lanes per query are no faster
    if (g != 8 && g != 4 && g != 2) return HF_EINVAL;
    const int qpw = kQueryThreads / g;
    if (div_up(m, qpw) > 65535) return HF_EINVAL;
    const int rs = ((nsample + 3) & ~3) + 4;
    const size_t lds = 16 * (2 * static_cast<size_t>(kQueryThreads) + static_cast<size_t>(qpw) * kHitCap) + sizeof(int) * (static_cast<size_t>(qpw) * rs + qpw);
    const dim3 grid(b, div_up(m, qpw));
#define HF_BQ_LAUNCH(GRP, GG)                                                                                                     \
    do {                                                                                                                          \
        const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(&bq_query_kernel<GRP, 1, GG>), lds);                    \
        if (lrc != HF_OK) return lrc;                                                                                             \
        hipLaunchKernelGGL((bq_query_kernel<GRP, 1, GG>), grid, dim3(kQueryThreads), lds, st, n, m, hbits, radius, thresh, inv_cs, \
                           nsample, ns_shift, pshift, xyz1, xyz2, center, ws, stride, idx, pts_cnt, grouped);                     \
    } while (0)
    if (grouped) {
        if (g == 8) HF_BQ_LAUNCH(true, 8); else if (g == 4) HF_BQ_LAUNCH(true, 4); else HF_BQ_LAUNCH(true, 2);
    } else {
        if (g == 8) HF_BQ_LAUNCH(false, 8); else if (g == 4) HF_BQ_LAUNCH(false, 4); else HF_BQ_LAUNCH(false, 2);
    }
#undef HF_BQ_LAUNCH
    return launch_status();
}

}  // namespace hf
