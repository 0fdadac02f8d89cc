// sampling.hip -- farthest_point_sample / gather_point (+grad) for gfx950.
//
// Replaces sampling/tf_sampling_g.cu:105-192 of the reference (launchers :199-211).
//
// FPS design (MI355X-first, not the reference's 512-thread / global-temp layout):
//   * one 1024-thread workgroup (16 waves, 4 per SIMD) per cloud; every point and its running
//     min-distance live in VGPRs for the whole sampling (16 points/thread at n = 16384), so a
//     round touches no memory except one 16-entry LDS exchange;
//   * per round: VALU update of the thread's points, DPP wave arg-max, one LDS slot per wave
//     (double-buffered -> a single s_barrier per round), every wave then reduces the 16 slots
//     redundantly and picks up the winner's coordinates from the same slot;
//   * the reference's tie rule (smallest k mod 512, then smallest k -- it falls out of its
//     512-thread strided scan + left-biased tree, tf_sampling_g.cu:146,158) is reproduced with
//     an explicit tie key, so the result does not depend on this kernel's own layout.
// The op is latency/VALU-bound on one CU per cloud (m-1 strictly sequential rounds), not
// HBM-bound; DESIGN.md reports us/round.
#include <math.h>

#include "hf_common.h"

namespace hf {

// smaller key wins a distance tie
__device__ __forceinline__ unsigned fps_tiekey(int k) { return (static_cast<unsigned>(k & 511) << 22) | static_cast<unsigned>(k >> 9); }

constexpr int kFpsMaxPoints = 16384;  // on-chip limit: points held in registers
constexpr int kFpsMaxWaves = 16;

struct __attribute__((aligned(32))) FpsSlot {
    int dist;   // float bits of the wave's best running distance (>= 0), -1 = no valid point
    int k;      // its point index
    float x, y, z;
    int pad[3];
};

template <int N> struct FpsVec { typedef float type __attribute__((ext_vector_type(N))); };
template <> struct FpsVec<1> { typedef float type; };
template <int N> __device__ __forceinline__ float vec_get(const typename FpsVec<N>::type &v, int i) { return v[i]; }
template <> __device__ __forceinline__ float vec_get<1>(const float &v, int) { return v; }
template <int N> __device__ __forceinline__ void vec_set(typename FpsVec<N>::type &v, int i, float f) { v[i] = f; }
template <> __device__ __forceinline__ void vec_set<1>(float &v, int, float f) { v = f; }
template <int N> struct FpsIVec { typedef int type __attribute__((ext_vector_type(N))); };
template <> struct FpsIVec<1> { typedef int type; };
template <int N> __device__ __forceinline__ int ivec_get(const typename FpsIVec<N>::type &v, int i) { return v[i]; }
template <> __device__ __forceinline__ int ivec_get<1>(const int &v, int) { return v; }
template <int N> __device__ __forceinline__ void ivec_set(typename FpsIVec<N>::type &v, int i, int f) { v[i] = f; }
template <> __device__ __forceinline__ void ivec_set<1>(int &v, int, int f) { v = f; }

// block-level pick among the per-wave slots; every wave computes the same winner.  Lanes replicate
// slot (lane & (NW-1)) in each DPP row, so 4-step row reductions suffice; the winner's record is then
// taken from the winning lane's registers (v_readlane) instead of a second, dependent LDS read.  The
// tie-key reduction only runs when two waves really post the same distance.
struct FpsPick { int k; float x, y, z; };

template <int NW>
__device__ __forceinline__ FpsPick fps_pick_slot(const FpsSlot *slots, int lane)
{
    const int sl = lane & (NW - 1);
    const int4 head = *reinterpret_cast<const int4 *>(&slots[sl]);  // dist, k, x, y
    const float sz = slots[sl].z;
    const int sd = head.x, sk = head.y;
    const unsigned gbest = row_max_u32(static_cast<unsigned>(sd + 1));
    unsigned long long win = __ballot(static_cast<unsigned>(sd + 1) == gbest) & ((1ull << NW) - 1ull);
    if (__builtin_popcountll(win) > 1) {  // wave-uniform, rare: exact distance tie between waves
        const unsigned key = static_cast<unsigned>(sd + 1) == gbest ? fps_tiekey(sk) : 0xffffffffu;
        const unsigned gkey = row_min_u32(key);
        win = __ballot(key == gkey) & ((1ull << NW) - 1ull);
    }
    const int wl = __builtin_ctzll(win);
    FpsPick r;
    r.k = __builtin_amdgcn_readlane(sk, wl);
    r.x = __int_as_float(__builtin_amdgcn_readlane(head.z, wl));
    r.y = __int_as_float(__builtin_amdgcn_readlane(head.w, wl));
    r.z = readlane_f(sz, wl);
    return r;
}

// slot i of thread t holds point fps_slot_point<PPT,NT>(t,i).  Inside a thread the slots are ordered by
// (k mod 512, k) so that "first slot wins" is the reference's tie rule: with NT >= 512 every slot of a
// thread shares k mod 512; with NT = 256 the first half of the slots has residue t, the second t+256.
template <int PPT, int NT>
__device__ __forceinline__ int fps_slot_point(int t, int i)
{
    if constexpr (NT >= 512 || PPT == 1) {
        return t + i * NT;
    } else {
        static_assert(NT == 256 && PPT % 2 == 0, "slot order implemented for 256-thread workgroups");
        constexpr int H = PPT / 2;
        return t + 256 * (i / H) + 512 * (i % H);
    }
}

template <int PPT, int NT>
__global__ __launch_bounds__(NT) void fps_onchip_kernel(int n, int m, const float *__restrict__ xyz,
                                                        int *__restrict__ out)
{
    constexpr int kFpsWaves = NT / 64;
    __shared__ FpsSlot slots[2][kFpsWaves];
    // the sampled indices are staged in LDS and written out once at the end: a global store per round
    // would make every round's s_barrier wait for its vmcnt(0) (an L2 round trip on the critical path)
    __shared__ unsigned short picked[PPT * NT];
    typedef typename FpsVec<PPT>::type vec_t;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const float *pts = xyz + static_cast<size_t>(blockIdx.x) * n * 3;
    int *o = out + static_cast<size_t>(blockIdx.x) * m;

    // x/y/z as ext vectors: a wave-uniform dynamic index lowers to s_set_gpr_idx (no scratch)
    vec_t x, y, z;
    int td[PPT];  // running min distance as float bits (>= +0: int order == float order); -1 = no point
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = fps_slot_point<PPT, NT>(t, i);
        const bool ok = k < n;
        const int kk = ok ? k : 0;
        vec_set<PPT>(x, i, pts[kk * 3 + 0]);
        vec_set<PPT>(y, i, pts[kk * 3 + 1]);
        vec_set<PPT>(z, i, pts[kk * 3 + 2]);
        td[i] = ok ? __float_as_int(1e38f) : -1;  // min() keeps -1 forever; -1 never beats a real point
    }

    float x1 = pts[0], y1 = pts[1], z1 = pts[2];
    const int mm = m < PPT * NT ? m : PPT * NT;  // rounds beyond n only repeat point 0 (all distances are 0)
    if (t == 0) picked[0] = 0;

    for (int j = 1; j < mm; ++j) {
        int best = -1, bi = 0;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const float dx = vec_get<PPT>(x, i) - x1, dy = vec_get<PPT>(y, i) - y1, dz = vec_get<PPT>(z, i) - z1;
            const float d = dx * dx + dy * dy + dz * dz;
            const int s = min(__float_as_int(d), td[i]);  // == fminf on non-negative floats
            td[i] = s;
            if (s > best) { best = s; bi = i; }  // strict: the first slot in (k mod 512, k) order wins
        }
        // wave arg-max; an exact tie between lanes (rare) is resolved with the explicit tie key
        const int wbest = static_cast<int>(wave_max_u32(static_cast<unsigned>(best + 1))) - 1;
        unsigned long long tied = __ballot(best == wbest);
        if (__builtin_popcountll(tied) > 1) {
            const unsigned key = best == wbest ? fps_tiekey(fps_slot_point<PPT, NT>(t, bi)) : 0xffffffffu;
            const unsigned kmin = wave_min_u32(key);
            tied = __ballot(key == kmin);
        }
        const int wl = __builtin_ctzll(tied);
        const int wi = __builtin_amdgcn_readlane(bi, wl);
        const float cx = vec_get<PPT>(x, wi), cy = vec_get<PPT>(y, wi), cz = vec_get<PPT>(z, wi);
        FpsSlot *cur = slots[j & 1];
        if (lane == wl) {
            cur[wave].dist = wbest;
            cur[wave].k = fps_slot_point<PPT, NT>(t, wi);
            cur[wave].x = cx;
            cur[wave].y = cy;
            cur[wave].z = cz;
        }
        __syncthreads();
        const FpsPick w = fps_pick_slot<kFpsWaves>(cur, lane);
        x1 = w.x;
        y1 = w.y;
        z1 = w.z;
        if (t == 0) picked[j] = static_cast<unsigned short>(w.k);
    }
    __syncthreads();
    for (int j = t; j < m; j += NT) o[j] = j < mm ? picked[j] : 0;
}

// ------------------------------------------------------------------------------------------
// One WAVE per cloud (clouds of at most 512 points: the RoI clouds of the second stage, 800 per batch, 512 -> 128 -> 32 -> 8
// points; the coarsest encoder levels of the RPN).  A round of the workgroup kernel above is an LDS exchange and a barrier
// between four or more waves (~5 us per round with eight workgroups per CU: 0.64 ms for 800 x 512 -> 128); a single wave holds
// all 512 points (8 per lane) and needs neither: the round is the lanes' updates, one DPP arg-max and three v_readlane.  Same
// arithmetic, same tie rule (k < 512: the smaller k wins; within a lane the slots are in ascending k and the first one wins).
// ------------------------------------------------------------------------------------------
constexpr int kFpsWaveMaxPoints = 512;
constexpr int kFpsWaveThreads = 256;   // four independent clouds per workgroup

template <int PPT>
__global__ __launch_bounds__(kFpsWaveThreads) void fps_wave_kernel(int b, int n, int m, const float *__restrict__ xyz, int *__restrict__ out)
{
    typedef typename FpsVec<PPT>::type vec_t;
    const int lane = threadIdx.x & 63;
    const int cloud = blockIdx.x * (kFpsWaveThreads / 64) + (threadIdx.x >> 6);
    if (cloud >= b) return;   // whole waves leave; nothing below synchronises across waves
    const float *pts = xyz + static_cast<size_t>(cloud) * n * 3;
    int *o = out + static_cast<size_t>(cloud) * m;
    vec_t x, y, z;
    int td[PPT];  // running min distance as float bits (>= +0: int order == float order); -1 = no point
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = lane + 64 * i;
        const bool ok = k < n;
        const int kk = ok ? k : 0;
        vec_set<PPT>(x, i, pts[kk * 3 + 0]);
        vec_set<PPT>(y, i, pts[kk * 3 + 1]);
        vec_set<PPT>(z, i, pts[kk * 3 + 2]);
        td[i] = ok ? __float_as_int(1e38f) : -1;
    }
    float x1 = pts[0], y1 = pts[1], z1 = pts[2];
    const int mm = m < PPT * 64 ? m : PPT * 64;  // rounds beyond n only repeat point 0 (all distances are 0)
    if (lane == 0) o[0] = 0;
    for (int j = 1; j < mm; ++j) {
        int best = -1, bi = 0;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const float dx = vec_get<PPT>(x, i) - x1, dy = vec_get<PPT>(y, i) - y1, dz = vec_get<PPT>(z, i) - z1;
            const float d = dx * dx + dy * dy + dz * dz;
            const int sdist = min(__float_as_int(d), td[i]);  // == fminf on non-negative floats
            td[i] = sdist;
            if (sdist > best) { best = sdist; bi = i; }  // strict: the first slot (the smaller k) wins
        }
        const int wbest = static_cast<int>(wave_max_u32(static_cast<unsigned>(best + 1))) - 1;
        unsigned long long tied = __ballot(best == wbest);
        if (__builtin_popcountll(tied) > 1) {   // wave-uniform, rare: exact distance tie between lanes
            const unsigned key = best == wbest ? fps_tiekey(lane + 64 * bi) : 0xffffffffu;
            const unsigned kmin = wave_min_u32(key);
            tied = __ballot(key == kmin);
        }
        const int wl = __builtin_ctzll(tied);
        const int wi = __builtin_amdgcn_readlane(bi, wl);
        x1 = readlane_f(vec_get<PPT>(x, wi), wl);
        y1 = readlane_f(vec_get<PPT>(y, wi), wl);
        z1 = readlane_f(vec_get<PPT>(z, wi), wl);
        if (lane == 0) o[j] = wl + 64 * wi;
    }
    for (int j = mm + lane; j < m; j += 64) o[j] = 0;
}

// ------------------------------------------------------------------------------------------
// Bucketed FPS (main path for larger clouds).  Same arithmetic, same winner every round; the
// difference is which points get touched.
//
// A new sample p only lowers the running distance of points closer to p than their current value.
// Late in the sampling that is a small neighbourhood, yet the plain kernel above updates all n
// points every round.  Here the cloud is first counting-sorted by Morton cell (prologue, in LDS)
// so that every (wave, slot) pair holds a spatially compact BUCKET of 64 points with a bounding
// box and a cached maximum running distance.  Per round, lanes 0..PPT-1 of each wave test their
// wave's PPT buckets at once:  if  dist2(p, bbox) * 0.99999 > bucket_max  every point of the bucket
// has d >= its running distance, min(d, td) is a no-op and the bucket is skipped -- exactly.
// (fp32: each computed d is >= the computed box bound * (1 - 8 ulp); 0.99999 leaves 40x slack.)
// Only touched buckets recompute their maximum; a wave whose buckets were all skipped re-posts
// its cached candidate.  The block-level pick and the tie rule are those of the plain kernel.
// ------------------------------------------------------------------------------------------
// Diagnostic build only (-DHF_FPS_STAMPS, scripts/probes/fps_stamps.py): wave 0 of every workgroup sums the shader
// cycles of the sections of a round.  The product library is built without it.
#ifdef HF_FPS_STAMPS
__device__ unsigned long long g_fps_stamps[64 * 8];
#define HF_FPS_STAMP(i)                                                                                               \
    do {                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        unsigned long long ts_;                                                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_)::"memory");                                   \
        stamp_sum[i] += ts_ - stamp_last;                                                                             \
        stamp_last = ts_;                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    } while (0)
#else
#define HF_FPS_STAMP(i)
#endif

constexpr int kFpsCellBits = 12;
constexpr int kFpsCells = 1 << kFpsCellBits;

struct FpsBucketShared {
    unsigned bmin[3], bmax[3];
    int wsum[kFpsMaxWaves];
    FpsSlot slots[2][kFpsMaxWaves];
};

template <int PPT, int NT>
__global__ __launch_bounds__(NT) void fps_bucket_kernel(int n, int m, const float *__restrict__ xyz,
                                                        int *__restrict__ out)
{
    constexpr int kFpsThreads = NT, kFpsWaves = NT / 64;
    static_assert(PPT <= 32, "one bucket per lane of the low half-wave at most (32-bit touch mask)");
    __shared__ FpsBucketShared sh;
    __shared__ int cellstart[kFpsCells];
    __shared__ unsigned short order[kFpsThreads * PPT];   // sorted position -> original point index
    __shared__ unsigned short picked[kFpsThreads * PPT];  // sampled indices, written out once at the end

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const float *pts = xyz + static_cast<size_t>(blockIdx.x) * n * 3;
    int *o = out + static_cast<size_t>(blockIdx.x) * m;

    // ---------------- prologue 1: bounding box, cell of every point ----------------
    if (t < 3) { sh.bmin[t] = 0xffffffffu; sh.bmax[t] = 0u; }
    if (t < kFpsMaxWaves) sh.wsum[t] = 0;   // block_exclusive_scan sums 16 wave slots; a 512-thread group fills 8
    for (int c = t; c < kFpsCells; c += kFpsThreads) cellstart[c] = 0;
    __syncthreads();
    {
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (int k = t; k < n; k += kFpsThreads) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float v = pts[k * 3 + d];
                lo[d] = fminf(lo[d], v);
                hi[d] = fmaxf(hi[d], v);
            }
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const unsigned a = wave_min_u32(f2ord(lo[d]));
            const unsigned b = wave_max_u32(f2ord(hi[d]));
            if (lane == 0) { atomicMin(&sh.bmin[d], a); atomicMax(&sh.bmax[d], b); }
        }
    }
    __syncthreads();
    float glo[3], gext[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { glo[d] = ord2f(sh.bmin[d]); gext[d] = ord2f(sh.bmax[d]) - glo[d]; }
    // split the 12 cell bits over the axes so that cells are as cubic as possible
    int bits[3] = { 0, 0, 0 };
    {
        float cs[3] = { gext[0], gext[1], gext[2] };
        for (int b = 0; b < kFpsCellBits; ++b) {
            int d = 0;
            if (cs[1] > cs[d]) d = 1;
            if (cs[2] > cs[d]) d = 2;
            if (d == 0) { bits[0]++; cs[0] *= 0.5f; } else if (d == 1) { bits[1]++; cs[1] *= 0.5f; } else { bits[2]++; cs[2] *= 0.5f; }
        }
    }
    float cscale[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) cscale[d] = gext[d] > 0.0f ? static_cast<float>(1 << bits[d]) / gext[d] : 0.0f;
    auto cell_of = [&](float px, float py, float pz) -> int {
        int c[3];
        const float v[3] = { px, py, pz };
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            int q = static_cast<int>((v[d] - glo[d]) * cscale[d]);
            const int top = (1 << bits[d]) - 1;
            c[d] = q < 0 ? 0 : (q > top ? top : q);
        }
        // generalised Morton interleave (axes drop out when their bits run out)
        int code = 0, pos = 0;
        for (int b = 0; b < kFpsCellBits; ++b) {
#pragma unroll
            for (int d = 0; d < 3; ++d)
                if (b < bits[d]) { code |= ((c[d] >> b) & 1) << pos; ++pos; }
        }
        return code;
    };
    for (int k = t; k < n; k += kFpsThreads) atomicAdd(&cellstart[cell_of(pts[k * 3], pts[k * 3 + 1], pts[k * 3 + 2])], 1);
    __syncthreads();
    // ---------------- prologue 2: scan the cells, scatter the point indices ----------------
    {
        constexpr int CPT = kFpsCells / kFpsThreads;
        int c[CPT], sum = 0;
#pragma unroll
        for (int u = 0; u < CPT; ++u) { c[u] = cellstart[t * CPT + u]; sum += c[u]; }
        int run = block_exclusive_scan(sum, sh.wsum, nullptr);
#pragma unroll
        for (int u = 0; u < CPT; ++u) { cellstart[t * CPT + u] = run; run += c[u]; }
    }
    __syncthreads();
    for (int k = t; k < n; k += kFpsThreads) {
        const int pos = atomicAdd(&cellstart[cell_of(pts[k * 3], pts[k * 3 + 1], pts[k * 3 + 2])], 1);
        order[pos] = static_cast<unsigned short>(k);
    }
    __syncthreads();

    // ---------------- prologue 3: load the buckets; bucket g = i*kFpsWaves + wave sits in slot i of this wave ----------------
    // x / y / z / running distance as ext vectors: the rounds below index them with a wave-uniform slot number taken from a bit
    // mask (s_set_gpr_idx / v_movrel).  A chain of `if (mask >> i & 1)` over all PPT slots cost a scalar branch per slot and
    // loop -- 32 branches a round, most of a round's time (scripts/probes/fps_stamps.py).  Measured and dropped: a binary tree of
    // branches to per-slot code (instruction-cache misses: 2.0 us per round), the running distance in LDS with the lanes that
    // hold a bucket's maximum cached per bucket (0.92), 32 slots as two 16-wide halves behind a branch (1.18).
    typename FpsVec<PPT>::type x, y, z;
    typename FpsIVec<PPT>::type td;
    // lane i (< PPT) keeps the box and the maximum running distance of bucket i
    float bx0 = 0.f, bx1 = 0.f, by0 = 0.f, by1 = 0.f, bz0 = 0.f, bz1 = 0.f;
    int bmaxv = -1;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = (i * kFpsWaves + wave) * 64 + lane;
        const bool ok = p < n;
        const int k = ok ? order[p] : 0;
        const float px = pts[k * 3 + 0], py = pts[k * 3 + 1], pz = pts[k * 3 + 2];
        vec_set<PPT>(x, i, px);
        vec_set<PPT>(y, i, py);
        vec_set<PPT>(z, i, pz);
        ivec_set<PPT>(td, i, ok ? __float_as_int(1e38f) : -1);  // -1 is never raised by min(), never wins
        const unsigned lx = wave_min_u32(ok ? f2ord(px) : 0xffffffffu), hx = wave_max_u32(ok ? f2ord(px) : 0u);
        const unsigned ly = wave_min_u32(ok ? f2ord(py) : 0xffffffffu), hy = wave_max_u32(ok ? f2ord(py) : 0u);
        const unsigned lz = wave_min_u32(ok ? f2ord(pz) : 0xffffffffu), hz = wave_max_u32(ok ? f2ord(pz) : 0u);
        const bool any = __ballot(ok) != 0ull;
        if (lane == i) {
            bx0 = ord2f(lx); bx1 = ord2f(hx); by0 = ord2f(ly); by1 = ord2f(hy); bz0 = ord2f(lz); bz1 = ord2f(hz);
            bmaxv = any ? __float_as_int(1e38f) : -1;
        }
    }

    float x1 = pts[0], y1 = pts[1], z1 = pts[2];
    const int mm = m < kFpsThreads * PPT ? m : kFpsThreads * PPT;
    if (t == 0) picked[0] = 0;
    // this wave's current candidate (wave-uniform values)
    int c_dist = -1, c_k = 0;
    float c_x = 0.f, c_y = 0.f, c_z = 0.f;

#ifdef HF_FPS_STAMPS
    unsigned long long stamp_sum[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif
    for (int j = 1; j < mm; ++j) {
        HF_FPS_STAMP(0);
        // ---- which of my wave's buckets can change? (lanes 0..PPT-1, one bucket each) ----
        const float ex = fmaxf(fmaxf(bx0 - x1, x1 - bx1), 0.0f);
        const float ey = fmaxf(fmaxf(by0 - y1, y1 - by1), 0.0f);
        const float ez = fmaxf(fmaxf(bz0 - z1, z1 - bz1), 0.0f);
        const float lb = ex * ex + ey * ey + ez * ez;
        const bool need = bmaxv >= 0 && !(lb * 0.99999f > __int_as_float(bmaxv));
        const unsigned mask = static_cast<unsigned>(__ballot(need));
        HF_FPS_STAMP(1);
        if (mask != 0u) {
            for (unsigned todo = mask; todo != 0u; todo &= todo - 1u) {
                const int i = __builtin_ctz(todo);  // wave-uniform
                const float dx = vec_get<PPT>(x, i) - x1, dy = vec_get<PPT>(y, i) - y1, dz = vec_get<PPT>(z, i) - z1;
                const float d = dx * dx + dy * dy + dz * dz;
                const int s = min(__float_as_int(d), ivec_get<PPT>(td, i));
                ivec_set<PPT>(td, i, s);
                const int nb = static_cast<int>(wave_max_u32(static_cast<unsigned>(s + 1))) - 1;
                if (lane == i) bmaxv = nb;
            }
            HF_FPS_STAMP(2);
            // ---- this wave's best point: max distance, then the reference's tie key ----
            const int wbest = static_cast<int>(wave_max_u32(static_cast<unsigned>(bmaxv + 1))) - 1;
            c_dist = wbest;
            if (wbest >= 0) {
                unsigned bestkey = 0xffffffffu;
                for (unsigned eq = static_cast<unsigned>(__ballot(bmaxv == wbest)); eq != 0u; eq &= eq - 1u) {
                    const int i = __builtin_ctz(eq);
                    const bool match = ivec_get<PPT>(td, i) == wbest;
                    const int korig = order[(i * kFpsWaves + wave) * 64 + lane];
                    unsigned long long cand = __ballot(match);
                    unsigned kmin;
                    if (__builtin_popcountll(cand) > 1) {  // rare: the same distance twice inside a bucket
                        const unsigned key = match ? fps_tiekey(korig) : 0xffffffffu;
                        kmin = wave_min_u32(key);
                        cand = __ballot(key == kmin);
                    } else {
                        kmin = fps_tiekey(__builtin_amdgcn_readlane(korig, __builtin_ctzll(cand)));
                    }
                    if (kmin < bestkey) {
                        bestkey = kmin;
                        const int L = __builtin_ctzll(cand);
                        c_k = __builtin_amdgcn_readlane(korig, L);
                        c_x = readlane_f(vec_get<PPT>(x, i), L);
                        c_y = readlane_f(vec_get<PPT>(y, i), L);
                        c_z = readlane_f(vec_get<PPT>(z, i), L);
                    }
                }
            }
        }
        HF_FPS_STAMP(3);
        FpsSlot *cur = sh.slots[j & 1];
        if (lane == 0) {
            cur[wave].dist = c_dist;
            cur[wave].k = c_k;
            cur[wave].x = c_x;
            cur[wave].y = c_y;
            cur[wave].z = c_z;
        }
        HF_FPS_STAMP(4);
        __syncthreads();
        HF_FPS_STAMP(5);
        const FpsPick w = fps_pick_slot<kFpsWaves>(cur, lane);
        x1 = w.x;
        y1 = w.y;
        z1 = w.z;
        if (t == 0) picked[j] = static_cast<unsigned short>(w.k);
        HF_FPS_STAMP(6);
    }
#ifdef HF_FPS_STAMPS
    if (t == 0 && blockIdx.x < 64)
        for (int i = 0; i < 8; ++i) g_fps_stamps[blockIdx.x * 8 + i] = stamp_sum[i];
#endif
    __syncthreads();
    for (int j = t; j < m; j += kFpsThreads) o[j] = j < mm ? picked[j] : 0;
}

// Fallback for clouds larger than the on-chip limit: running distances in caller scratch
// (b,n) floats, points re-read from global (L2-resident).  Same selection rule.
__global__ __launch_bounds__(1024) void fps_scratch_kernel(int n, int m, const float *__restrict__ xyz,
                                                           float *__restrict__ temp, int *__restrict__ out)
{
    constexpr int kFpsThreads = 1024, kFpsWaves = 16;
    __shared__ FpsSlot slots[2][kFpsWaves];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const float *pts = xyz + static_cast<size_t>(blockIdx.x) * n * 3;
    float *td = temp + static_cast<size_t>(blockIdx.x) * n;
    int *o = out + static_cast<size_t>(blockIdx.x) * m;
    for (int k = t; k < n; k += kFpsThreads) td[k] = 1e38f;
    float x1 = pts[0], y1 = pts[1], z1 = pts[2];
    if (t == 0) o[0] = 0;
    for (int j = 1; j < m; ++j) {
        int best = -1, bk = 0;
        float bx = 0, by = 0, bz = 0;
        for (int k = t; k < n; k += kFpsThreads) {
            const float px = pts[k * 3 + 0], py = pts[k * 3 + 1], pz = pts[k * 3 + 2];
            const float dx = px - x1, dy = py - y1, dz = pz - z1;
            const float d = dx * dx + dy * dy + dz * dz;
            const float d2 = fminf(d, td[k]);
            td[k] = d2;
            const int s = __float_as_int(d2);
            if (s > best) { best = s; bk = k; bx = px; by = py; bz = pz; }
        }
        const int wbest = static_cast<int>(wave_max_u32(static_cast<unsigned>(best + 1))) - 1;
        const unsigned long long tied = __ballot(best == wbest);
        const int wl = __builtin_ctzll(tied);
        const int par = j & 1;
        if (lane == wl) {
            slots[par][wave].dist = wbest;
            slots[par][wave].k = bk;
            slots[par][wave].x = bx;
            slots[par][wave].y = by;
            slots[par][wave].z = bz;
        }
        __syncthreads();
        const FpsPick w = fps_pick_slot<kFpsWaves>(slots[par], lane);
        const int old = w.k;
        x1 = w.x;
        y1 = w.y;
        z1 = w.z;
        if (t == 0) o[j] = old;
    }
}

// gather_point: out[b,j,:] = inp[b,idx[b,j],:]   (tf_sampling_g.cu:172-181)
__global__ void gather_point_kernel(int n, int m, long long total, const float *__restrict__ inp,
                                    const int *__restrict__ idx, float *__restrict__ out)
{
    // one thread per output float: coalesced 4-byte stores, 12-byte segments gathered through L2
    for (long long e = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long row = e / 3;
        const int c = static_cast<int>(e - row * 3);
        const long long bb = row / m;
        const int a = idx[row];
        out[e] = inp[(bb * n + a) * 3 + c];
    }
}

// gather_point grad: atomicAdd scatter (tf_sampling_g.cu:183-192), target zeroed by the caller
__global__ void gather_point_grad_kernel(int n, int m, long long total, const float *__restrict__ out_g,
                                         const int *__restrict__ idx, float *__restrict__ inp_g)
{
    for (long long e = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long row = e / 3;
        const int c = static_cast<int>(e - row * 3);
        const long long bb = row / m;
        const int a = idx[row];
        atomicAdd(&inp_g[(bb * n + a) * 3 + c], out_g[e]);
    }
}

template <int PPT, int NT>
static int launch_fps_bucket(int b, int n, int m, const float *inp, int *out, hipStream_t st)
{
    hipLaunchKernelGGL((fps_bucket_kernel<PPT, NT>), dim3(b), dim3(NT), 0, st, n, m, inp, out);
    return launch_status();
}

// bucketed kernel: NT threads, one 64-point bucket per (wave, slot), PPT = ceil(n / NT) slots rounded up to a power of two
template <int NT>
static int launch_fps_bucket_nt(int b, int n, int m, const float *inp, int *out, hipStream_t st)
{
    const int ppt = div_up(n, NT);
    if (ppt <= 1) return launch_fps_bucket<1, NT>(b, n, m, inp, out, st);
    if (ppt <= 2) return launch_fps_bucket<2, NT>(b, n, m, inp, out, st);
    if (ppt <= 4) return launch_fps_bucket<4, NT>(b, n, m, inp, out, st);
    if (ppt <= 8) return launch_fps_bucket<8, NT>(b, n, m, inp, out, st);
    if (ppt <= 16) return launch_fps_bucket<16, NT>(b, n, m, inp, out, st);
    if constexpr (NT <= 512) {
        if (ppt <= 32) return launch_fps_bucket<32, NT>(b, n, m, inp, out, st);
    }
    return HF_EINVAL;
}

template <int PPT, int NT>
static int launch_fps_plain(int b, int n, int m, const float *inp, int *out, hipStream_t st)
{
    hipLaunchKernelGGL((fps_onchip_kernel<PPT, NT>), dim3(b), dim3(NT), 0, st, n, m, inp, out);
    return launch_status();
}

// plain kernel: NT threads, PPT = ceil(n / NT) rounded up to a power of two
template <int NT>
static int launch_fps_plain_nt(int b, int n, int m, const float *inp, int *out, hipStream_t st)
{
    const int ppt = div_up(n, NT);
    if (ppt <= 1) return launch_fps_plain<1, NT>(b, n, m, inp, out, st);
    if (ppt <= 2) return launch_fps_plain<2, NT>(b, n, m, inp, out, st);
    if (ppt <= 4) return launch_fps_plain<4, NT>(b, n, m, inp, out, st);
    if (ppt <= 8) return launch_fps_plain<8, NT>(b, n, m, inp, out, st);
    if (ppt <= 16) return launch_fps_plain<16, NT>(b, n, m, inp, out, st);
    if constexpr (NT <= 512) {
        if (ppt <= 32) return launch_fps_plain<32, NT>(b, n, m, inp, out, st);
    }
    return HF_EINVAL;
}

// mode: HF_FPS_AUTO by size, HF_FPS_PLAIN / HF_FPS_BUCKET forced (hf_farthest_point_sample_variant: the tests run both
// kernels against the oracle); nt: workgroup size of the plain kernel (0 = by size)
static int launch_fps_onchip(int b, int n, int m, const float *inp, int *out, hipStream_t st, int mode = HF_FPS_AUTO, int nt = 0)
{
    if (mode == HF_FPS_WAVE || (mode == HF_FPS_AUTO && n <= kFpsWaveMaxPoints)) {
        if (n > kFpsWaveMaxPoints) return HF_EINVAL;
        const dim3 grid(div_up(b, kFpsWaveThreads / 64));
        const int ppt = div_up(n, 64);
        if (ppt <= 1) hipLaunchKernelGGL((fps_wave_kernel<1>), grid, dim3(kFpsWaveThreads), 0, st, b, n, m, inp, out);
        else if (ppt <= 2) hipLaunchKernelGGL((fps_wave_kernel<2>), grid, dim3(kFpsWaveThreads), 0, st, b, n, m, inp, out);
        else if (ppt <= 4) hipLaunchKernelGGL((fps_wave_kernel<4>), grid, dim3(kFpsWaveThreads), 0, st, b, n, m, inp, out);
        else hipLaunchKernelGGL((fps_wave_kernel<8>), grid, dim3(kFpsWaveThreads), 0, st, b, n, m, inp, out);
        return launch_status();
    }
    // bucket pruning pays once there are enough rounds to amortise its prologue
    const bool bucket = mode == 2 || (mode == 0 && n >= 8192 && m >= 256);
    if (bucket) {
        // 8 waves of 16 / 32 buckets each: 0.83 us per round at 16384 points against 0.88 with 16 waves (the block-level pick and
        // the bucket test are repeated by every wave)
        if (nt != 1024 && n <= 512 * 32) return launch_fps_bucket_nt<512>(b, n, m, inp, out, st);
        return launch_fps_bucket_nt<1024>(b, n, m, inp, out, st);
    }
    if (nt == 0) nt = n <= 4096 ? 256 : 512;  // fewer, fatter waves: measured faster at every size
    if (nt == 256 && n <= 256 * 16) return launch_fps_plain_nt<256>(b, n, m, inp, out, st);
    if (nt <= 512 && n <= 512 * 32) return launch_fps_plain_nt<512>(b, n, m, inp, out, st);
    return launch_fps_plain_nt<1024>(b, n, m, inp, out, st);
}

}  // namespace hf

using namespace hf;

#ifdef HF_FPS_STAMPS
extern "C" __attribute__((visibility("default"))) int hf_debug_fps_stamps(unsigned long long *host, int nwg)
{
    return static_cast<int>(hipMemcpyFromSymbol(host, HIP_SYMBOL(hf::g_fps_stamps), sizeof(unsigned long long) * 8 * nwg));
}
#endif

HF_API int hf_fps_onchip_limit(void) { return kFpsMaxPoints; }

HF_API size_t hf_fps_workspace(int b, int n)
{
    if (b <= 0 || n <= hf_fps_onchip_limit()) return 0;
    return sizeof(float) * static_cast<size_t>(b) * static_cast<size_t>(n);
}

HF_API int hf_farthest_point_sample(int b, int n, int m, const float *inp, float *temp, int *out, hf_stream_t stream)
{
    // FarthestPointSampleOp: npoint > 0 (tf_sampling.cpp:100), (b,n,3) input (:106)
    if (b < 0 || n <= 0 || m <= 0 || !inp || !out) return HF_EINVAL;
    if (b == 0) return HF_OK;
    hipStream_t st = as_stream(stream);
    if (n <= kFpsMaxPoints) return launch_fps_onchip(b, n, m, inp, out, st);
    if (!temp) return HF_EWORKSPACE;
    hipLaunchKernelGGL(fps_scratch_kernel, dim3(b), dim3(1024), 0, st, n, m, inp, temp, out);
    return launch_status();
}

HF_API int hf_farthest_point_sample_variant(int kernel, int threads, int b, int n, int m, const float *inp, float *temp, int *out,
                                            hf_stream_t stream)
{
    if (b < 0 || n <= 0 || m <= 0 || !inp || !out) return HF_EINVAL;
    if (kernel < HF_FPS_AUTO || kernel > HF_FPS_WAVE || (threads != 0 && threads != 256 && threads != 512 && threads != 1024))
        return HF_EINVAL;
    if (b == 0) return HF_OK;
    if (n > kFpsMaxPoints) return hf_farthest_point_sample(b, n, m, inp, temp, out, stream);
    if (kernel == HF_FPS_WAVE && n > kFpsWaveMaxPoints) return HF_EINVAL;
    if (kernel == HF_FPS_PLAIN && threads != 0 && n > threads * (threads == 1024 ? 16 : 32)) return HF_EINVAL;
    if (kernel == HF_FPS_PLAIN && threads == 256 && n > 256 * 16) return HF_EINVAL;
    return launch_fps_onchip(b, n, m, inp, out, as_stream(stream), kernel, threads);
}

HF_API int hf_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out, hf_stream_t stream)
{
    if (b < 0 || n <= 0 || m < 0 || !inp || !idx || !out) return HF_EINVAL;
    const long long total = static_cast<long long>(b) * m * 3;
    if (total == 0) return HF_OK;
    const int block = 256;
    const int grid = static_cast<int>(std::min<long long>((total + block - 1) / block, kNumCU * 8));
    hipLaunchKernelGGL(gather_point_kernel, dim3(grid), dim3(block), 0, as_stream(stream), n, m, total, inp, idx, out);
    return launch_status();
}

HF_API int hf_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx, float *inp_g,
                                hf_stream_t stream)
{
    if (b < 0 || n <= 0 || m < 0 || !out_g || !idx || !inp_g) return HF_EINVAL;
    hipStream_t st = as_stream(stream);
    if (b == 0) return HF_OK;
    int rc = hip_status(hipMemsetAsync(inp_g, 0, sizeof(float) * static_cast<size_t>(b) * n * 3, st));
    if (rc != HF_OK) return rc;
    const long long total = static_cast<long long>(b) * m * 3;
    if (total == 0) return HF_OK;
    const int block = 256;
    const int grid = static_cast<int>(std::min<long long>((total + block - 1) / block, kNumCU * 8));
    hipLaunchKernelGGL(gather_point_grad_kernel, dim3(grid), dim3(block), 0, st, n, m, total, out_g, idx, inp_g);
    return launch_status();
}
