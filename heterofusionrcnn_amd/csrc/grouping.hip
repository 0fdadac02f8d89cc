// grouping.hip -- brute-force query_ball_point (fallback; the main kernel is in ballquery.hip) / group_point (+grad) /
// group_concat / knn_point / select_top_k for gfx950, and the ball-query entry points.
//
// Replaces grouping/tf_grouping_g.cu:3-123 of the reference (launchers :125-141).
//
// The hit test of the reference is  max(sqrtf(s),1e-20f) < radius  with
// s = (x2-x1)^2+(y2-y1)^2+(z2-z1)^2 (tf_grouping_g.cu:24-25).  sqrtf is correctly rounded and
// monotone, so there is one float T with  hit <=> s < T ; the host computes T exactly
// (ball_threshold) and the kernels never take a square root.
#include <math.h>

#include <algorithm>

#include "hf_common.h"

namespace hf {

// smallest float T such that !(max(sqrtf(T),1e-20f) < radius); hit <=> s < T for s >= 0
static float ball_threshold(float radius)
{
    if (!(radius > 1e-20f)) return 0.0f;  // max(.,1e-20f) >= 1e-20f >= radius: never a hit
    if (isinf(radius)) return INFINITY;   // every finite s hits
    float t = radius * radius;
    if (isinf(t)) t = 3.4028234664e38f;
    while (t > 0.0f && sqrtf(t) >= radius) t = nextafterf(t, 0.0f);      // now sqrtf(t) < radius (or t == 0)
    while (sqrtf(nextafterf(t, INFINITY)) < radius) t = nextafterf(t, INFINITY);
    if (sqrtf(t) < radius) t = nextafterf(t, INFINITY);                   // first non-hit
    return t;
}

// ------------------------------------------------------------------------------------------
// Brute-force ball query (fallback path): one thread per query, data streamed through LDS tiles
// in ascending index so "first nsample in index order" is the scan order; hits are collected in
// an LDS row buffer and the block then writes idx / pts_cnt / grouped_xyz fully coalesced.
// ------------------------------------------------------------------------------------------
constexpr int kQbThreads = 256;
constexpr int kQbTile = 1024;  // data points per LDS tile (12 KB)

template <bool GROUP>
__global__ __launch_bounds__(kQbThreads) void qbp_bruteforce_kernel(int n, int m, float thresh, int nsample,
                                                                    const float *__restrict__ xyz1,
                                                                    const float *__restrict__ xyz2, int center,
                                                                    int *__restrict__ idx, int *__restrict__ pts_cnt,
                                                                    float *__restrict__ grouped)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *tile = reinterpret_cast<float *>(smem_raw);                 // kQbTile*3 floats
    int *rows = reinterpret_cast<int *>(tile + kQbTile * 3);           // kQbThreads * rs ints
    int *cnts = rows + kQbThreads * (nsample | 1);                     // kQbThreads ints
    const int rs = nsample | 1;                                        // odd row stride: conflict-free columns

    const int t = threadIdx.x;
    const int bb = blockIdx.y;
    const int j0 = blockIdx.x * kQbThreads;
    const int j = j0 + t;
    const float *p1 = xyz1 + static_cast<size_t>(bb) * n * 3;
    const float *p2 = xyz2 + static_cast<size_t>(bb) * m * 3;

    const bool live = j < m;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) { qx = p2[j * 3 + 0]; qy = p2[j * 3 + 1]; qz = p2[j * 3 + 2]; }
    int cnt = live ? 0 : nsample;
    int *myrow = rows + t * rs;

    for (int base = 0; base < n; base += kQbTile) {
        const int tn = min(kQbTile, n - base);
        __syncthreads();  // previous tile fully consumed
        for (int e = t; e < tn * 3; e += kQbThreads) tile[e] = p1[static_cast<size_t>(base) * 3 + e];
        // stop streaming once every query of the block is full
        const int all_full = __syncthreads_and(cnt >= nsample);
        if (all_full) break;
        if (cnt < nsample) {
            for (int k = 0; k < tn; ++k) {
                const float x1 = tile[k * 3 + 0], y1 = tile[k * 3 + 1], z1 = tile[k * 3 + 2];
                const float dx = qx - x1, dy = qy - y1, dz = qz - z1;
                const float s = dx * dx + dy * dy + dz * dz;
                if (s < thresh) {
                    myrow[cnt] = base + k;
                    if (++cnt == nsample) break;
                }
            }
        }
    }
    cnts[t] = live ? cnt : 0;
    __syncthreads();

    // ---- coalesced epilogue: rows j0..j0+nrow-1 are contiguous in every output ----
    const int nrow = min(kQbThreads, m - j0);
    if (pts_cnt && t < nrow) pts_cnt[static_cast<size_t>(bb) * m + j0 + t] = cnts[t];
    const int total = nrow * nsample;
    if (idx) {
        int *o = idx + (static_cast<size_t>(bb) * m + j0) * nsample;
        for (int e = t; e < total; e += kQbThreads) {
            const int r = e / nsample, c = e - r * nsample;
            const int h = cnts[r];
            o[e] = h == 0 ? 0 : rows[r * rs + (c < h ? c : 0)];
        }
    }
    if (GROUP) {
        float *g = grouped + (static_cast<size_t>(bb) * m + j0) * nsample * 3;
        for (int e = t; e < total * 3; e += kQbThreads) {
            const int rc = e / 3, d = e - rc * 3;
            const int r = rc / nsample, c = rc - r * nsample;
            const int h = cnts[r];
            const int k = h == 0 ? 0 : rows[r * rs + (c < h ? c : 0)];
            float v = p1[k * 3 + d];
            if (center) v = v - p2[(j0 + r) * 3 + d];
            g[e] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// group_point: out[b,j,k,:] = points[b, idx[b,j,k], :]   (tf_grouping_g.cu:40-57)
// Row copies; VEC floats per lane so that both the gathered source row segment and the
// destination are contiguous (16-byte accesses when c % 4 == 0).
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ void group_point_kernel(int n, int c, long long rows_per_batch, long long nrows,
                                   const float *__restrict__ points, const int *__restrict__ idx,
                                   float *__restrict__ out)
{
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int cv = c / VEC;  // vectors per row
    const long long total = nrows * cv;
    for (long long e = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long row = e / cv;
        const int l = static_cast<int>(e - row * cv);
        const long long bb = row / rows_per_batch;
        const int ii = idx[row];
        const vec_t v = *reinterpret_cast<const vec_t *>(points + (bb * n + ii) * c + l * VEC);
        *reinterpret_cast<vec_t *>(out + row * c + l * VEC) = v;
    }
}

// The same op when c / 4 is a power of two (c = 16 .. 1024): no integer division (lane -> row by a shift, the batch is
// grid.y), ROWS rows per thread in flight (index loads, then ROWS independent 16-byte gathers before the first store),
// nontemporal stores for the write-once output so that the feature table stays in the XCD's L2.
template <int ROWS>
__global__ __launch_bounds__(256) void group_point_pow2_kernel(int nbatch, int n, int c, int cv_shift, int rows_per_batch,
                                                               const float *__restrict__ points,
                                                               const int *__restrict__ idx, float *__restrict__ out)
{
    typedef float v4 __attribute__((ext_vector_type(4)));
    // linear workgroup id = cloud + b * chunk: workgroups are dealt round-robin over the 8 XCDs, so with b a multiple of 8
    // every workgroup of a cloud runs on ONE XCD and that cloud's table stays in its 4 MB L2 (speed only)
    const int bb = blockIdx.x % nbatch;
    const unsigned gxb = gridDim.x / nbatch;
    const unsigned tid = (blockIdx.x / nbatch) * 256u + threadIdx.x;
    const int l = static_cast<int>(tid & ((1u << cv_shift) - 1u));
    const int rstride = static_cast<int>((gxb * 256u) >> cv_shift);
    const float *base = points + static_cast<size_t>(bb) * n * c + l * 4;
    const int *ib = idx + static_cast<size_t>(bb) * rows_per_batch;
    float *ob = out + static_cast<size_t>(bb) * rows_per_batch * c + l * 4;
    for (int row0 = static_cast<int>(tid >> cv_shift); row0 < rows_per_batch; row0 += rstride * ROWS) {
        v4 v[ROWS];
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int row = row0 + i * rstride;
            v[i] = *reinterpret_cast<const v4 *>(base + static_cast<size_t>(ib[row < rows_per_batch ? row : row0]) * c);
        }
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int row = row0 + i * rstride;
            if (row < rows_per_batch) __builtin_nontemporal_store(v[i], reinterpret_cast<v4 *>(ob + static_cast<size_t>(row) * c));
        }
    }
}

// group_point grad: atomicAdd scatter (tf_grouping_g.cu:61-78); target zeroed by the caller
__global__ void group_point_grad_kernel(int n, int c, long long rows_per_batch, long long nrows,
                                        const float *__restrict__ grad_out, const int *__restrict__ idx,
                                        float *__restrict__ grad_points)
{
    const long long total = nrows * c;
    for (long long e = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long row = e / c;
        const int l = static_cast<int>(e - row * c);
        const long long bb = row / rows_per_batch;
        const int ii = idx[row];
        atomicAdd(grad_points + (bb * n + ii) * c + l, grad_out[e]);
    }
}

// group_point gradient in gather form: for every data point the rows of grad_out that name it (CSR inverse of idx, built
// with the geometry: hf_index_inverse) are summed in ascending (query, slot) order -- the order of the reference's sequential
// CPU loop (grouping/test/query_ball_point.cpp:53-66) -- and the point's gradient row is written ONCE: no atomics, no zero
// fill, deterministic.  c / VEC lanes per data point; grad_out rows have stride `width`, the gathered columns start at `col`.
template <int VEC>
__global__ void group_point_grad_gather_kernel(int n, int c, int width, int col, int cv, long long rows_per_batch, long long data_rows,
                                               const float *__restrict__ grad_out, const int *__restrict__ offsets,
                                               const int *__restrict__ entries, float *__restrict__ grad_points)
{
    const int rows_per_block = blockDim.x / cv;
    const int cvec = threadIdx.x % cv, rsub = threadIdx.x / cv;
    if (rsub >= rows_per_block) return;
    for (long long row = blockIdx.x * static_cast<long long>(rows_per_block) + rsub; row < data_rows;
         row += static_cast<long long>(gridDim.x) * rows_per_block) {
        const long long bb = row / n;
        const int k = static_cast<int>(row - bb * n);
        const int *off = offsets + bb * (n + 1);
        const int *ent = entries + bb * rows_per_batch;
        const float *go = grad_out + bb * rows_per_batch * width + col + cvec * VEC;
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
        const int lo = off[k], hi = off[k + 1];
        for (int a = lo; a < hi; ++a) {
            const float *src = go + static_cast<long long>(ent[a]) * width;
            if constexpr (VEC == 4) {
                const float4 v = *reinterpret_cast<const float4 *>(src);
                acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
            } else {
                acc[0] += src[0];
            }
        }
        float *dst = grad_points + row * c + cvec * VEC;
        if constexpr (VEC == 4) *reinterpret_cast<float4 *>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        else dst[0] = acc[0];
    }
}

// ------------------------------------------------------------------------------------------
// group_concat: the concat of sample_and_group (pointnet_util.py:58-60) written directly,
//   out[b,j,k,:] = [ grouped_xyz[b,j,k,0:3], points[b, idx[b,j,k], 0:c], 0 ... ]   (row width `width` >= 3 + c)
// or, xyz_last (the multi-scale module's order, pointnet_util.py:264): [ points[...], grouped_xyz[...], 0 ... ]
// instead of group_point into a temporary followed by a concat that reads and writes everything again.  The zero
// columns pad rows to a multiple of 4 floats so that the GEMM that consumes them can use 16-byte accesses
// (3 + C is never a multiple of 4 for the usual C = 64, 128).  One float4 of output per lane.
// ------------------------------------------------------------------------------------------
__global__ void group_concat_kernel(int n, int c, int w4, int xyz_col, int feat_col, long long rows_per_batch,
                                    long long nrows, const float *__restrict__ gxyz,
                                    const float *__restrict__ points, const int *__restrict__ idx,
                                    float *__restrict__ out)
{
    const long long total = nrows * w4;
    for (long long e = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long row = e / w4;
        const int q = static_cast<int>(e - row * w4);
        const long long bb = row / rows_per_batch;
        const float *src = points + (bb * n + idx[row]) * c;
        const float *g = gxyz + row * 3;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = q * 4 + i;
            const int jf = j - feat_col, jx = j - xyz_col;
            v[i] = (jf >= 0 && jf < c) ? src[jf] : ((jx >= 0 && jx < 3) ? g[jx] : 0.0f);
        }
        *reinterpret_cast<float4 *>(out + e * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// gradient w.r.t. points: the feature columns 3..3+c of every row scattered back (target zeroed by the caller)
__global__ void group_concat_grad_kernel(int n, int c, int width, int feat_col, long long rows_per_batch, long long nrows,
                                         const float *__restrict__ grad_out, const int *__restrict__ idx,
                                         float *__restrict__ grad_points)
{
    const long long total = nrows * c;
    for (long long e = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long row = e / c;
        const int l = static_cast<int>(e - row * c);
        const long long bb = row / rows_per_batch;
        atomicAdd(grad_points + (bb * n + idx[row]) * c + l, grad_out[row * width + feat_col + l]);
    }
}

// ------------------------------------------------------------------------------------------
// knn_point: the k nearest data points of every query, ascending (distance, index).
// Caller side of the path: the reference builds a dense (b,m,n) distance
// matrix |q|^2 - 2 q.p + |p|^2 with a batched matmul and takes tf.nn.top_k of its negation
// (grouping/tf_grouping.py:62-95; hf/core/pointfly.py:185-212 does the same for PointCNN: 1 GiB per
// frame at n = m = 16384).  Here: one thread per query, data streamed through LDS tiles, the k best
// kept in a sorted per-thread LDS column; a candidate costs one compare unless it enters the list.
// Distances are the direct (qx-px)^2+(qy-py)^2+(qz-pz)^2 (more accurate than the expanded form);
// ties go to the lower index, tf.nn.top_k's documented rule.
// ------------------------------------------------------------------------------------------
constexpr int kKnnThreads = 256;
constexpr int kKnnTile = 1024;

// k = 4 / 8 / 12 / 16 (every neighbourhood size of the shipped configs: the X-Conv K of the RPN is 8, of the RCNN 4 / 8 / 12) on
// SMALL clouds -- the second stage's 800 RoI clouds of 512 / 128 / 32 points per batch: one thread per query, the k best
// (distance, index) pairs in REGISTERS (sorted insertion as K select pairs, taken only when a candidate beats the current worst:
// ~K ln(n / K) times per query), the cloud staged through LDS as float4 so that a candidate is ONE broadcast read.  The kernel
// below keeps its lists in LDS columns and walks them with a data-dependent loop (353 us for 800 x 512 x 512, k = 4), binning +
// ring search pays only on large clouds (327 us).  Same ranking: (q - p)^2 summed per axis in fp32 without FMA, candidates in
// ascending index, a strictly smaller distance displaces -- ties to the lower index.
template <int K>
__global__ __launch_bounds__(kKnnThreads) void knn_small_kernel(int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                                float *__restrict__ val, int *__restrict__ idx)
{
    constexpr int kTile = 1024;
    __shared__ float4 tile[kTile];
    const int t = threadIdx.x, bb = blockIdx.y;
    const int j = blockIdx.x * kKnnThreads + t;
    const float *p1 = xyz1 + static_cast<size_t>(bb) * n * 3;
    const float *p2 = xyz2 + static_cast<size_t>(bb) * m * 3;
    const bool live = j < m;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) { qx = p2[j * 3]; qy = p2[j * 3 + 1]; qz = p2[j * 3 + 2]; }
    float bd[K];
    int bi[K];
#pragma unroll
    for (int s = 0; s < K; ++s) { bd[s] = INFINITY; bi[s] = 0x7fffffff; }
    for (int base = 0; base < n; base += kTile) {
        const int tn = min(kTile, n - base);
        __syncthreads();
        for (int e = t; e < tn; e += kKnnThreads) {
            const float *p = p1 + static_cast<size_t>(base + e) * 3;
            tile[e] = make_float4(p[0], p[1], p[2], 0.f);
        }
        __syncthreads();
        if (!live) continue;
        for (int c = 0; c < tn; ++c) {
            const float4 pt = tile[c];   // the same address in every lane: one broadcast read
            const float dx = qx - pt.x, dy = qy - pt.y, dz = qz - pt.z;
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < bd[K - 1]) {  // strict: indices ascend, an equal distance never displaces (NaN never enters)
                float cd = d;
                int ci = base + c;
                bool shift = false;   // from the insertion point on, every entry moves down one place (equal distances included)
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    const bool lt = shift || cd < bd[s];
                    shift = lt;
                    const float od = bd[s];
                    const int oi = bi[s];
                    bd[s] = lt ? cd : od; bi[s] = lt ? ci : oi;
                    cd = lt ? od : cd; ci = lt ? oi : ci;
                }
            }
        }
    }
    if (live) {
        float *ov = val + (static_cast<size_t>(bb) * m + j) * K;
        int *oi = idx + (static_cast<size_t>(bb) * m + j) * K;
#pragma unroll
        for (int s = 0; s < K; ++s) {
            ov[s] = bd[s];
            oi[s] = bi[s] == 0x7fffffff ? 0 : bi[s];
        }
    }
}

// PARTS lanes share one query: lane `part` takes the tile points c == part (mod PARTS) and keeps its own
// sorted list; part 0 then merges the PARTS lists by (distance, index).  More parts = more waves in flight
// when there are few queries (4096 queries x 8 clouds are only 512 waves with one thread per query).
template <int PARTS>
__global__ __launch_bounds__(kKnnThreads) void knn_kernel(int n, int m, int k, const float *__restrict__ xyz1,
                                                          const float *__restrict__ xyz2, float *__restrict__ val,
                                                          int *__restrict__ idx)
{
    constexpr int QB = kKnnThreads / PARTS;  // queries per workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *tile = reinterpret_cast<float *>(smem_raw);        // kKnnTile*3
    float *bd = tile + kKnnTile * 3;                          // k * kKnnThreads, column per thread
    int *bi = reinterpret_cast<int *>(bd + static_cast<size_t>(k) * kKnnThreads);
    const int t = threadIdx.x, bb = blockIdx.y;
    const int ql = t % QB, part = t / QB;
    const int j = blockIdx.x * QB + ql;
    const float *p1 = xyz1 + static_cast<size_t>(bb) * n * 3;
    const float *p2 = xyz2 + static_cast<size_t>(bb) * m * 3;
    const bool live = j < m;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) { qx = p2[j * 3]; qy = p2[j * 3 + 1]; qz = p2[j * 3 + 2]; }
    for (int s = 0; s < k; ++s) { bd[s * kKnnThreads + t] = INFINITY; bi[s * kKnnThreads + t] = 0x7fffffff; }
    float worst = INFINITY;  // bd[k-1]
    for (int base = 0; base < n; base += kKnnTile) {
        const int tn = min(kKnnTile, n - base);
        __syncthreads();
        for (int e = t; e < tn * 3; e += kKnnThreads) tile[e] = p1[static_cast<size_t>(base) * 3 + e];
        __syncthreads();
        if (!live) continue;
        for (int c = part; c < tn; c += PARTS) {
            const float dx = qx - tile[c * 3], dy = qy - tile[c * 3 + 1], dz = qz - tile[c * 3 + 2];
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < worst) {  // strict: inside one part indices ascend, an equal distance never displaces
                int pos = k - 1;
                while (pos > 0 && bd[(pos - 1) * kKnnThreads + t] > d) {
                    bd[pos * kKnnThreads + t] = bd[(pos - 1) * kKnnThreads + t];
                    bi[pos * kKnnThreads + t] = bi[(pos - 1) * kKnnThreads + t];
                    --pos;
                }
                bd[pos * kKnnThreads + t] = d;
                bi[pos * kKnnThreads + t] = base + c;
                worst = bd[(k - 1) * kKnnThreads + t];
            }
        }
    }
    __syncthreads();
    if (PARTS > 1 && live && part == 0) {
        // merge the other parts' lists into mine, ordered by (distance, index)
        for (int p = 1; p < PARTS; ++p) {
            const int ot = p * QB + ql;
            for (int s = 0; s < k; ++s) {
                const float d = bd[s * kKnnThreads + ot];
                const int id = bi[s * kKnnThreads + ot];
                const float wd = bd[(k - 1) * kKnnThreads + t];
                const int wi = bi[(k - 1) * kKnnThreads + t];
                if (!(d < wd || (d == wd && id < wi))) break;  // the rest of that list is no better
                int pos = k - 1;
                while (pos > 0) {
                    const float pd = bd[(pos - 1) * kKnnThreads + t];
                    const int pi = bi[(pos - 1) * kKnnThreads + t];
                    if (!(pd > d || (pd == d && pi > id))) break;
                    bd[pos * kKnnThreads + t] = pd;
                    bi[pos * kKnnThreads + t] = pi;
                    --pos;
                }
                bd[pos * kKnnThreads + t] = d;
                bi[pos * kKnnThreads + t] = id;
            }
        }
    }
    if (live && part == 0) {
        float *ov = val + (static_cast<size_t>(bb) * m + j) * k;
        int *oi = idx + (static_cast<size_t>(bb) * m + j) * k;
        for (int s = 0; s < k; ++s) {
            const int id = bi[s * kKnnThreads + t];
            ov[s] = bd[s * kKnnThreads + t];
            oi[s] = id == 0x7fffffff ? 0 : id;
        }
    }
}

// select_top_k (SelectionSort op, tf_grouping_g.cu:83-123): one thread per (b,m) row, as the
// reference; unused by every shipped config (kept for surface completeness).
__global__ void select_top_k_kernel(int n, long long nrows, int k, const float *__restrict__ dist,
                                    int *__restrict__ outi, float *__restrict__ out)
{
    for (long long r = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; r < nrows;
         r += static_cast<long long>(gridDim.x) * blockDim.x) {
        const float *src = dist + r * n;
        float *pd = out + r * n;
        int *pi = outi + r * n;
        for (int s = 0; s < n; ++s) { pd[s] = src[s]; pi[s] = s; }
        for (int s = 0; s < k && s < n; ++s) {
            int mn = s;
            for (int u = s + 1; u < n; ++u)
                if (pd[u] < pd[mn]) mn = u;
            if (mn != s) {
                const float tf = pd[mn]; pd[mn] = pd[s]; pd[s] = tf;
                const int ti = pi[mn]; pi[mn] = pi[s]; pi[s] = ti;
            }
        }
    }
}

// group_point writing into / its gradient reading from a COLUMN SLICE of wider rows: out[row][col .. col+c) of rows of
// `width` floats.  PointCNN concatenates the lifted coordinates with the gathered features of the previous layer
// (pointcnn.py:96-99): gathering straight into the concat buffer saves the copy of the larger part of it, and the gradient
// of the concat is read in place.
template <int VEC>
__global__ void group_point_into_kernel(int n, int c, int width, int col, long long rows_per_batch, long long nrows,
                                        const float *__restrict__ points, const int *__restrict__ idx, float *__restrict__ out)
{
    const int cv = c / VEC;
    const long long total = nrows * cv;
    for (long long e = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long row = e / cv;
        const int l = static_cast<int>(e - row * cv) * VEC;
        const long long bb = row / rows_per_batch;
        const float *src = points + (bb * n + idx[row]) * c + l;
        float *dst = out + row * width + col + l;
        if (VEC == 4) *reinterpret_cast<float4 *>(dst) = *reinterpret_cast<const float4 *>(src);
        else dst[0] = src[0];
    }
}

__global__ void group_point_grad_from_kernel(int n, int c, int width, int col, long long rows_per_batch, long long nrows,
                                             const float *__restrict__ grad_out, const int *__restrict__ idx,
                                             float *__restrict__ grad_points)
{
    const long long total = nrows * c;
    for (long long e = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long row = e / c;
        const int l = static_cast<int>(e - row * c);
        const long long bb = row / rows_per_batch;
        atomicAdd(grad_points + (bb * n + idx[row]) * c + l, grad_out[row * width + col + l]);
    }
}

static int grid_for(long long work_items, int block)
{
    long long g = (work_items + block - 1) / block;
    const long long cap = static_cast<long long>(kNumCU) * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return static_cast<int>(g);
}

static int launch_ball_query_bruteforce(int b, int n, int m, float thresh, int nsample, const float *xyz1,
                                        const float *xyz2, int center, int *idx, int *pts_cnt, float *grouped,
                                        hipStream_t st)
{
    const size_t lds = sizeof(float) * kQbTile * 3 + sizeof(int) * kQbThreads * (static_cast<size_t>(nsample | 1) + 1);
    if (lds > 160 * 1024) return HF_EINVAL;  // nsample > ~148: not used by any config
    dim3 grid(div_up(m, kQbThreads), b);
    if (grouped) {
        if (const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(&qbp_bruteforce_kernel<true>), lds); lrc != HF_OK) return lrc;
        hipLaunchKernelGGL((qbp_bruteforce_kernel<true>), grid, dim3(kQbThreads), lds, st, n, m, thresh, nsample, xyz1,
                           xyz2, center, idx, pts_cnt, grouped);
    } else {
        if (const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(&qbp_bruteforce_kernel<false>), lds); lrc != HF_OK) return lrc;
        hipLaunchKernelGGL((qbp_bruteforce_kernel<false>), grid, dim3(kQbThreads), lds, st, n, m, thresh, nsample,
                           xyz1, xyz2, center, idx, pts_cnt, grouped);
    }
    return launch_status();
}

// variant: HF_BQ_AUTO picks by shape; the others force one kernel (tests cover every path this way).
//   auto: with a workspace, 32 clouds and up, and more than one round of the single-launch kernel's workgroups (the batched
//         launches of a train step) -> the cell-sorted pair of kernels (ballquery_sorted.hip: the cell structure is built once
//         per cloud; measured at 80 clouds 52-57 us against 70 us (16384 points, r 0.5), 20 against 31 us (4096 points),
//         18 against 20 us (1024 points); equal at 32 clouds; 23 against 16 us at 16 clouds);
//         else the single-launch cell kernel (ballquery.hip); shapes outside both ranges (nsample > 128, n > 2^19 points per
//         cloud, infinite radius) take the brute-force kernel.
static int launch_ball_query(int variant, int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2,
                             int center, int *idx, int *pts_cnt, float *grouped, void *workspace, size_t workspace_bytes,
                             hipStream_t st)
{
    const float thresh = ball_threshold(radius);
    if (variant < HF_BQ_AUTO || variant > HF_BQ_SORTED) return HF_EINVAL;
    if (variant == HF_BQ_SORTED)
        return launch_ball_query_sorted(b, n, m, radius, thresh, nsample, xyz1, xyz2, center, idx, pts_cnt, grouped, workspace,
                                        workspace_bytes, st);
    if (variant == HF_BQ_AUTO && workspace && b >= 32 && static_cast<long long>(b) * div_up(m, 128) > kNumCU) {
        const int rc = launch_ball_query_sorted(b, n, m, radius, thresh, nsample, xyz1, xyz2, center, idx, pts_cnt, grouped, workspace,
                                                workspace_bytes, st);
        if (rc != HF_EINVAL) return rc;
    }
    if (variant != HF_BQ_BRUTEFORCE) {
        const int rc = launch_ball_query_cell(b, n, m, radius, thresh, nsample, xyz1, xyz2, center, idx, pts_cnt, grouped, st);
        if (rc != HF_EINVAL || variant == HF_BQ_CELL) return rc;
    }
    return launch_ball_query_bruteforce(b, n, m, thresh, nsample, xyz1, xyz2, center, idx, pts_cnt, grouped, st);
}

// ------------------------------------------------------------------------------------------
// knn_point on a 2-D grid with an outward ring search.
// The tiled kernel above evaluates all n*m pairs (2.1 G for the 16384 x 16384 x 8 neighbour search of the PointCNN
// RPN).  Here the data points of a cloud are first binned into G x G cells over its two widest axes (counting sort
// in LDS, one workgroup per cloud).  A query visits the cells at Chebyshev distance 0, 1, 2, ... from its own cell;
// after ring R every unvisited point lies outside the (2R+1)^2 block of cells, i.e. at least `gap` away along one of
// the two axes, where gap is the distance from the query to the nearest side of the block that still has cells
// beyond it.  The search stops when (gap - margin)^2 > the current k-th best distance: fp32 d = dx^2 + dy^2 + dz^2
// is >= each of its terms, and `margin` (1e-3 of a cell) covers the rounding of the cell assignment.  Insertion
// compares (distance, index) lexicographically = ties to the lower index, the tiled kernel's rule, so the visiting
// order does not matter.
// Per-cloud workspace: float4 pts[n] (x, y, z, index) in cell order | int cellstart[G*G+1] | float params[8]
// {amin, ascale, bmin, bscale, cell width a, cell width b, axis a, axis b}.
// ------------------------------------------------------------------------------------------
constexpr int kKnnBinThreads = 1024;
constexpr int kKnnMaxGrid = 64;          // G <= 64: 4096 cells of counters in LDS
constexpr int kKnnBinMaxData = 65536;

struct KnnWs {
    float4 *pts;
    int *cellstart;
    float *params;
};

__host__ __device__ inline size_t knn_ws_stride(int n, int g)
{
    const size_t bytes = sizeof(float4) * static_cast<size_t>(n) + sizeof(int) * (static_cast<size_t>(g) * g + 1 + 8);
    return (bytes + 15) / 16 * 16;
}

__device__ __forceinline__ KnnWs knn_ws(void *base, int frame, int n, int g)
{
    unsigned char *p = static_cast<unsigned char *>(base) + knn_ws_stride(n, g) * frame;
    KnnWs w;
    w.pts = reinterpret_cast<float4 *>(p);
    w.cellstart = reinterpret_cast<int *>(w.pts + n);
    w.params = reinterpret_cast<float *>(w.cellstart + g * g + 1);
    return w;
}

// the same expression in the binning and in the search: monotone non-decreasing in v, NaN -> 0
__device__ __forceinline__ int knn_cell1(float v, float vmin, float scale, int g)
{
    const float f = (v - vmin) * scale;
    const int c = f >= 0.0f ? (f < 1.0e9f ? static_cast<int>(f) : g - 1) : 0;
    return c > g - 1 ? g - 1 : c;
}

__global__ __launch_bounds__(kKnnBinThreads) void knn_bin_kernel(int n, int g, const float *__restrict__ xyz1,
                                                                 void *__restrict__ workspace)
{
    __shared__ unsigned bmin[3], bmax[3];
    __shared__ int wsum[16];
    __shared__ int count[kKnnMaxGrid * kKnnMaxGrid];
    const int t = threadIdx.x, lane = t & 63, bb = blockIdx.x;
    const float *p = xyz1 + static_cast<size_t>(bb) * n * 3;
    const KnnWs w = knn_ws(workspace, bb, n, g);
    const int ncell = g * g;
    if (t < 3) { bmin[t] = 0xffffffffu; bmax[t] = 0u; }
    for (int c = t; c < ncell; c += kKnnBinThreads) count[c] = 0;
    __syncthreads();
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int i = t; i < n; i += kKnnBinThreads) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = p[i * 3 + d];
            if (fabsf(v) < INFINITY) { lo[d] = fminf(lo[d], v); hi[d] = fmaxf(hi[d], v); }  // the box of the finite coordinates
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const unsigned a = wave_min_u32(f2ord(lo[d])), b = wave_max_u32(f2ord(hi[d]));
        if (lane == 0) { atomicMin(&bmin[d], a); atomicMax(&bmax[d], b); }
    }
    __syncthreads();
    float ext[3], vmin[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        vmin[d] = ord2f(bmin[d]);
        const float e = ord2f(bmax[d]) - vmin[d];
        ext[d] = (e > 0.0f && e < INFINITY) ? e : 0.0f;  // no finite point on this axis / zero extent: one cell
        if (!(fabsf(vmin[d]) < INFINITY)) vmin[d] = 0.0f;
    }
    // the two widest axes
    int narrow = 0;
    if (ext[1] < ext[narrow]) narrow = 1;
    if (ext[2] < ext[narrow]) narrow = 2;
    const int axa = narrow == 0 ? 1 : 0, axb = narrow == 2 ? 1 : 2;
    const float sa = ext[axa] > 0.0f ? static_cast<float>(g) / ext[axa] : 0.0f;
    const float sb = ext[axb] > 0.0f ? static_cast<float>(g) / ext[axb] : 0.0f;
    auto cell_of = [&](int i) -> int {
        return knn_cell1(p[i * 3 + axb], vmin[axb], sb, g) * g + knn_cell1(p[i * 3 + axa], vmin[axa], sa, g);
    };
    for (int i = t; i < n; i += kKnnBinThreads) atomicAdd(&count[cell_of(i)], 1);
    __syncthreads();
    int carry = 0;
    for (int base = 0; base < ncell; base += kKnnBinThreads) {
        const int c = base + t;
        const int v = c < ncell ? count[c] : 0;
        int tot;
        const int ex = block_exclusive_scan(v, wsum, &tot);
        if (c < ncell) {
            w.cellstart[c] = carry + ex;
            count[c] = carry + ex;  // becomes the scatter cursor
        }
        carry += tot;
    }
    if (t == 0) {
        w.cellstart[ncell] = carry;
        w.params[0] = vmin[axa]; w.params[1] = sa; w.params[2] = vmin[axb]; w.params[3] = sb;
        w.params[4] = sa > 0.0f ? ext[axa] / static_cast<float>(g) : INFINITY;   // cell widths
        w.params[5] = sb > 0.0f ? ext[axb] / static_cast<float>(g) : INFINITY;
        w.params[6] = __int_as_float(axa); w.params[7] = __int_as_float(axb);
    }
    __syncthreads();
    for (int i = t; i < n; i += kKnnBinThreads) {
        const int pos = atomicAdd(&count[cell_of(i)], 1);
        w.pts[pos] = make_float4(p[i * 3], p[i * 3 + 1], p[i * 3 + 2], __int_as_float(i));
    }
}

__global__ __launch_bounds__(kKnnThreads) void knn_grid_kernel(int n, int m, int k, int g, const float *__restrict__ xyz2,
                                                               void *__restrict__ workspace, float *__restrict__ val,
                                                               int *__restrict__ idx)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *bd = reinterpret_cast<float *>(smem_raw);                       // k * kKnnThreads, column per thread
    int *bi = reinterpret_cast<int *>(bd + static_cast<size_t>(k) * kKnnThreads);
    const int t = threadIdx.x, bb = blockIdx.y;
    const int j = blockIdx.x * kKnnThreads + t;
    if (j >= m) return;
    const KnnWs w = knn_ws(workspace, bb, n, g);
    const float *q = xyz2 + (static_cast<size_t>(bb) * m + j) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    for (int s = 0; s < k; ++s) { bd[s * kKnnThreads + t] = INFINITY; bi[s * kKnnThreads + t] = 0x7fffffff; }
    float worst = INFINITY;
    int worst_i = 0x7fffffff;
    auto visit_cell = [&](int c) {
        const int e = w.cellstart[c + 1];
        for (int i = w.cellstart[c]; i < e; ++i) {
            const float4 p = w.pts[i];
            const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
            const float d = dx * dx + dy * dy + dz * dz;
            const int ci = __float_as_int(p.w);
            // an infinite distance never enters the list (the tiled kernel's strict `d < worst` against its +inf
            // sentinel): such slots keep the sentinel index
            if (d < worst || (d == worst && ci < worst_i && d < INFINITY)) {
                int pos = k - 1;
                while (pos > 0) {
                    const float pd = bd[(pos - 1) * kKnnThreads + t];
                    const int pi = bi[(pos - 1) * kKnnThreads + t];
                    if (!(pd > d || (pd == d && pi > ci))) break;
                    bd[pos * kKnnThreads + t] = pd;
                    bi[pos * kKnnThreads + t] = pi;
                    --pos;
                }
                bd[pos * kKnnThreads + t] = d;
                bi[pos * kKnnThreads + t] = ci;
                worst = bd[(k - 1) * kKnnThreads + t];
                worst_i = bi[(k - 1) * kKnnThreads + t];
            }
        }
    };
    const float amin = w.params[0], sa = w.params[1], bmin = w.params[2], sb = w.params[3];
    const float wa = w.params[4], wb = w.params[5];
    const int axa = __float_as_int(w.params[6]), axb = __float_as_int(w.params[7]);
    const float qa = axa == 0 ? qx : (axa == 1 ? qy : qz), qb = axb == 0 ? qx : (axb == 1 ? qy : qz);
    const int ca = knn_cell1(qa, amin, sa, g), cb = knn_cell1(qb, bmin, sb, g);
    const float margin_a = 1e-3f * wa, margin_b = 1e-3f * wb;
    for (int ring = 0; ring < g; ++ring) {
        const int a0 = ca - ring, a1 = ca + ring, b0 = cb - ring, b1 = cb + ring;
        if (ring == 0) {
            visit_cell(cb * g + ca);
        } else {
            // top and bottom rows of the ring, then the two side columns without their corners
            for (int a = (a0 < 0 ? 0 : a0); a <= (a1 > g - 1 ? g - 1 : a1); ++a) {
                if (b0 >= 0) visit_cell(b0 * g + a);
                if (b1 <= g - 1) visit_cell(b1 * g + a);
            }
            for (int b = (b0 + 1 < 0 ? 0 : b0 + 1); b <= (b1 - 1 > g - 1 ? g - 1 : b1 - 1); ++b) {
                if (a0 >= 0) visit_cell(b * g + a0);
                if (a1 <= g - 1) visit_cell(b * g + a1);
            }
        }
        // distance from the query to the nearest side of the visited block that still has cells beyond it
        float gap = INFINITY;
        if (a0 > 0) gap = fminf(gap, (qa - (amin + static_cast<float>(a0) * wa)) - margin_a);
        if (a1 < g - 1) gap = fminf(gap, ((amin + static_cast<float>(a1 + 1) * wa) - qa) - margin_a);
        if (b0 > 0) gap = fminf(gap, (qb - (bmin + static_cast<float>(b0) * wb)) - margin_b);
        if (b1 < g - 1) gap = fminf(gap, ((bmin + static_cast<float>(b1 + 1) * wb) - qb) - margin_b);
        if (gap == INFINITY) break;                       // the block covers the whole grid
        if (gap > 0.0f && gap * gap > worst) break;       // nothing outside the block can enter the list
    }
    float *ov = val + (static_cast<size_t>(bb) * m + j) * k;
    int *oi = idx + (static_cast<size_t>(bb) * m + j) * k;
    for (int s = 0; s < k; ++s) {
        const int id = bi[s * kKnnThreads + t];
        ov[s] = bd[s * kKnnThreads + t];
        oi[s] = id == 0x7fffffff ? 0 : id;  // an unfilled slot (all remaining distances infinite / NaN): index 0, as above
    }
}

// The same search for k = 3 / 4 / 8 / 12 / 16 (three_nn and every neighbourhood size of the shipped configs) with the list in
// REGISTERS and the memory round trips batched.  The kernel above walks a cell point by point -- one dependent 16-byte load per
// candidate, plus two bound loads per cell -- ~100 serial round trips per query at k = 8: it is latency-bound (221 us for 8 x 16384
// queries on 4096 points).  Here: the cells of one grid ROW are consecutive in the sorted array, so the top and the bottom row of a
// ring are each ONE contiguous range (two bound loads, all four issued together); a range is scanned four candidates per trip (four
// loads in flight); the list is K (distance, index) register pairs, sorted insertion as K select pairs.  Same candidates, same
// (distance, index) order, same stop rule: same output.
template <int K>
__global__ __launch_bounds__(kKnnThreads) void knn_grid_reg_kernel(int n, int m, int g, const float *__restrict__ xyz2,
                                                                   void *__restrict__ workspace, float *__restrict__ val,
                                                                   int *__restrict__ idx)
{
    const int t = threadIdx.x, bb = blockIdx.y;
    const int j = blockIdx.x * kKnnThreads + t;
    if (j >= m) return;
    const KnnWs w = knn_ws(workspace, bb, n, g);
    const float *q = xyz2 + (static_cast<size_t>(bb) * m + j) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    float bd[K];
    int bi[K];
#pragma unroll
    for (int s = 0; s < K; ++s) { bd[s] = INFINITY; bi[s] = 0x7fffffff; }
    auto offer = [&](const float4 &p) __attribute__((always_inline)) {
        const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
        const float d = dx * dx + dy * dy + dz * dz;
        const int ci = __float_as_int(p.w);
        // an infinite distance never enters the list (the tiled kernel's strict `d < worst` against its +inf sentinel)
        if (d < bd[K - 1] || (d == bd[K - 1] && ci < bi[K - 1] && d < INFINITY)) {
            float cd = d;
            int cc = ci;
            bool shift = false;   // from the insertion point on every entry moves down one place
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const bool lt = shift || cd < bd[s] || (cd == bd[s] && cc < bi[s]);
                shift = lt;
                const float od = bd[s];
                const int oi = bi[s];
                bd[s] = lt ? cd : od; bi[s] = lt ? cc : oi;
                cd = lt ? od : cd; cc = lt ? oi : cc;
            }
        }
    };
    // sorted entries i0 .. i1-1, four per trip (past the end: the last entry again, not offered)
    auto scan = [&](int i0, int i1) __attribute__((always_inline)) {   // (outlined, the lists would live in scratch)
        for (int i = i0; i < i1; i += 4) {
            float4 p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = w.pts[min(i + u, i1 - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u < i1) offer(p[u]);
        }
    };
    const float amin = w.params[0], sa = w.params[1], bmin = w.params[2], sb = w.params[3];
    const float wa = w.params[4], wb = w.params[5];
    const int axa = __float_as_int(w.params[6]), axb = __float_as_int(w.params[7]);
    const float qa = axa == 0 ? qx : (axa == 1 ? qy : qz), qb = axb == 0 ? qx : (axb == 1 ? qy : qz);
    const int ca = knn_cell1(qa, amin, sa, g), cb = knn_cell1(qb, bmin, sb, g);
    const float margin_a = 1e-3f * wa, margin_b = 1e-3f * wb;
    for (int ring = 0; ring < g; ++ring) {
        const int a0 = ca - ring, a1 = ca + ring, b0 = cb - ring, b1 = cb + ring;
        const int alo = a0 < 0 ? 0 : a0, ahi = a1 > g - 1 ? g - 1 : a1;
        if (ring == 0) {
            const int c = cb * g + ca;
            scan(w.cellstart[c], w.cellstart[c + 1]);
        } else {
            // top and bottom rows of the ring: one contiguous range each (their four bounds requested together)
            const bool top = b0 >= 0, bot = b1 <= g - 1;
            const int t0 = top ? w.cellstart[b0 * g + alo] : 0, t1 = top ? w.cellstart[b0 * g + ahi + 1] : 0;
            const int u0 = bot ? w.cellstart[b1 * g + alo] : 0, u1 = bot ? w.cellstart[b1 * g + ahi + 1] : 0;
            scan(t0, t1);
            scan(u0, u1);
            // the two side columns without their corners: single cells, the bounds of a row's pair requested together
            for (int b = (b0 + 1 < 0 ? 0 : b0 + 1); b <= (b1 - 1 > g - 1 ? g - 1 : b1 - 1); ++b) {
                const bool left = a0 >= 0, right = a1 <= g - 1;
                const int l0 = left ? w.cellstart[b * g + a0] : 0, l1 = left ? w.cellstart[b * g + a0 + 1] : 0;
                const int r0 = right ? w.cellstart[b * g + a1] : 0, r1 = right ? w.cellstart[b * g + a1 + 1] : 0;
                scan(l0, l1);
                scan(r0, r1);
            }
        }
        // distance from the query to the nearest side of the visited block that still has cells beyond it
        float gap = INFINITY;
        if (a0 > 0) gap = fminf(gap, (qa - (amin + static_cast<float>(a0) * wa)) - margin_a);
        if (a1 < g - 1) gap = fminf(gap, ((amin + static_cast<float>(a1 + 1) * wa) - qa) - margin_a);
        if (b0 > 0) gap = fminf(gap, (qb - (bmin + static_cast<float>(b0) * wb)) - margin_b);
        if (b1 < g - 1) gap = fminf(gap, ((bmin + static_cast<float>(b1 + 1) * wb) - qb) - margin_b);
        if (gap == INFINITY) break;                       // the block covers the whole grid
        if (gap > 0.0f && gap * gap > bd[K - 1]) break;   // nothing outside the block can enter the list
    }
    float *ov = val + (static_cast<size_t>(bb) * m + j) * K;
    int *oi = idx + (static_cast<size_t>(bb) * m + j) * K;
#pragma unroll
    for (int s = 0; s < K; ++s) {
        ov[s] = bd[s];
        oi[s] = bi[s] == 0x7fffffff ? 0 : bi[s];  // an unfilled slot (all remaining distances infinite / NaN): index 0, as above
    }
}

static int knn_grid_for(int n)
{
    int g = 4;
    while (g < kKnnMaxGrid && g * g * 4 < n) g <<= 1;
    return g;
}

}  // namespace hf

using namespace hf;

HF_API int hf_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2,
                               int *idx, int *pts_cnt, hf_stream_t stream)
{
    // QueryBallPointGpuOp: radius > 0, nsample > 0 (tf_grouping.cpp:70-74); (b,n,3)/(b,m,3) (:79-85)
    if (!(radius > 0.0f) || nsample <= 0 || b < 0 || n <= 0 || m < 0 || !xyz1 || !xyz2 || !idx) return HF_EINVAL;
    if (b == 0 || m == 0) return HF_OK;
    return launch_ball_query(HF_BQ_AUTO, b, n, m, radius, nsample, xyz1, xyz2, 0, idx, pts_cnt, nullptr, nullptr, 0, as_stream(stream));
}

HF_API int hf_query_ball_group_xyz(int b, int n, int m, float radius, int nsample, const float *xyz1,
                                   const float *xyz2, int center, int *idx, int *pts_cnt, float *grouped_xyz,
                                   hf_stream_t stream)
{
    if (!(radius > 0.0f) || nsample <= 0 || b < 0 || n <= 0 || m < 0 || !xyz1 || !xyz2 || !grouped_xyz)
        return HF_EINVAL;
    if (b == 0 || m == 0) return HF_OK;
    return launch_ball_query(HF_BQ_AUTO, b, n, m, radius, nsample, xyz1, xyz2, center, idx, pts_cnt, grouped_xyz, nullptr, 0,
                             as_stream(stream));
}

HF_API size_t hf_ball_query_workspace(int b, int n) { return ball_query_sorted_workspace(b, n); }

HF_API int hf_query_ball_group_xyz_ws(int variant, int b, int n, int m, float radius, int nsample, const float *xyz1,
                                      const float *xyz2, int center, int *idx, int *pts_cnt, float *grouped_xyz,
                                      void *workspace, size_t workspace_bytes, hf_stream_t stream)
{
    if (!(radius > 0.0f) || nsample <= 0 || b < 0 || n <= 0 || m < 0 || !xyz1 || !xyz2 || (!grouped_xyz && !idx)) return HF_EINVAL;
    if (b == 0 || m == 0) return HF_OK;
    return launch_ball_query(variant, b, n, m, radius, nsample, xyz1, xyz2, center, idx, pts_cnt, grouped_xyz, workspace,
                             workspace_bytes, as_stream(stream));
}

HF_API int hf_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out,
                          hf_stream_t stream)
{
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0 || !points || !idx || !out) return HF_EINVAL;
    const long long nrows = static_cast<long long>(b) * m * nsample;
    if (nrows == 0) return HF_OK;
    const long long rpb = static_cast<long long>(m) * nsample;
    const int block = 256;
    hipStream_t st = as_stream(stream);
    const bool al16 = (reinterpret_cast<uintptr_t>(points) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    const int cv = c / 4;
    if (c % 4 == 0 && al16 && cv >= 4 && cv <= 256 && (cv & (cv - 1)) == 0 && b <= 65535 && rpb * cv < (1LL << 31)) {
        int cv_shift = 0;
        while ((1 << cv_shift) < cv) ++cv_shift;
        long long gx = (rpb * cv + 256 * 4 - 1) / (256 * 4);
        const long long cap = std::max<long long>(1, (kNumCU * 8) / b);
        if (gx > cap) gx = cap;
        hipLaunchKernelGGL((group_point_pow2_kernel<4>), dim3(static_cast<unsigned>(gx * b)), dim3(256), 0, st, b, n, c, cv_shift,
                           static_cast<int>(rpb), points, idx, out);
    } else if (c % 4 == 0 && al16)
        hipLaunchKernelGGL((group_point_kernel<4>), dim3(grid_for(nrows * (c / 4), block)), dim3(block), 0, st, n, c,
                           rpb, nrows, points, idx, out);
    else
        hipLaunchKernelGGL((group_point_kernel<1>), dim3(grid_for(nrows * c, block)), dim3(block), 0, st, n, c, rpb,
                           nrows, points, idx, out);
    return launch_status();
}

HF_API int hf_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                               float *grad_points, hf_stream_t stream)
{
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0 || !grad_out || !idx || !grad_points) return HF_EINVAL;
    if (b == 0) return HF_OK;
    hipStream_t st = as_stream(stream);
    int rc = hip_status(hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(b) * n * c, st));
    if (rc != HF_OK) return rc;
    const long long nrows = static_cast<long long>(b) * m * nsample;
    if (nrows == 0) return HF_OK;
    const int block = 256;
    hipLaunchKernelGGL(group_point_grad_kernel, dim3(grid_for(nrows * c, block)), dim3(block), 0, st, n, c,
                       static_cast<long long>(m) * nsample, nrows, grad_out, idx, grad_points);
    return launch_status();
}

HF_API int hf_group_point_into(int b, int n, int c, int m, int nsample, int width, int col, const float *points, const int *idx,
                               float *out, hf_stream_t stream)
{
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0 || col < 0 || width < col + c || !points || !idx || !out) return HF_EINVAL;
    const long long nrows = static_cast<long long>(b) * m * nsample;
    if (nrows == 0) return HF_OK;
    const long long rpb = static_cast<long long>(m) * nsample;
    const int block = 256;
    hipStream_t st = as_stream(stream);
    const bool vec = c % 4 == 0 && width % 4 == 0 && col % 4 == 0 && reinterpret_cast<uintptr_t>(points) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(out) % 16 == 0;
    if (vec)
        hipLaunchKernelGGL((group_point_into_kernel<4>), dim3(grid_for(nrows * (c / 4), block)), dim3(block), 0, st, n, c, width, col,
                           rpb, nrows, points, idx, out);
    else
        hipLaunchKernelGGL((group_point_into_kernel<1>), dim3(grid_for(nrows * c, block)), dim3(block), 0, st, n, c, width, col, rpb,
                           nrows, points, idx, out);
    return launch_status();
}

HF_API int hf_group_point_grad_from(int b, int n, int c, int m, int nsample, int width, int col, const float *grad_out,
                                    const int *idx, float *grad_points, hf_stream_t stream)
{
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0 || col < 0 || width < col + c || !grad_out || !idx || !grad_points)
        return HF_EINVAL;
    if (b == 0) return HF_OK;
    hipStream_t st = as_stream(stream);
    int rc = hip_status(hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(b) * n * c, st));
    if (rc != HF_OK) return rc;
    const long long nrows = static_cast<long long>(b) * m * nsample;
    if (nrows == 0) return HF_OK;
    const int block = 256;
    hipLaunchKernelGGL(group_point_grad_from_kernel, dim3(grid_for(nrows * c, block)), dim3(block), 0, st, n, c, width, col,
                       static_cast<long long>(m) * nsample, nrows, grad_out, idx, grad_points);
    return launch_status();
}

HF_API int hf_group_point_grad_gather(int b, int n, int c, int m, int nsample, int width, int col, const float *grad_out,
                                      const int *offsets, const int *entries, float *grad_points, hf_stream_t stream)
{
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0 || col < 0 || width < col + c || !offsets || !grad_points) return HF_EINVAL;
    const long long data_rows = static_cast<long long>(b) * n;
    if (data_rows == 0) return HF_OK;
    const long long rpb = static_cast<long long>(m) * nsample;
    if (rpb > 0 && (!grad_out || !entries)) return HF_EINVAL;
    const bool vec4 = c % 4 == 0 && width % 4 == 0 && col % 4 == 0 &&
                      ((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(grad_points)) % 16 == 0);
    const int cv = vec4 ? c / 4 : c;
    if (cv > 1024) return HF_EINVAL;
    int block = 256;
    if (cv > block) block = ((cv + 63) / 64) * 64;
    const int rows_per_block = block / cv;
    long long grid = (data_rows + rows_per_block - 1) / rows_per_block;
    if (grid > kNumCU * 64ll) grid = kNumCU * 64ll;
    if (vec4)
        hipLaunchKernelGGL((group_point_grad_gather_kernel<4>), dim3(static_cast<unsigned>(grid)), dim3(block), 0, as_stream(stream), n, c,
                           width, col, cv, rpb, data_rows, grad_out, offsets, entries, grad_points);
    else
        hipLaunchKernelGGL((group_point_grad_gather_kernel<1>), dim3(static_cast<unsigned>(grid)), dim3(block), 0, as_stream(stream), n, c,
                           width, col, cv, rpb, data_rows, grad_out, offsets, entries, grad_points);
    return launch_status();
}

HF_API int hf_group_concat(int b, int n, int c, int m, int nsample, int width, int xyz_last, const float *grouped_xyz,
                           const float *points, const int *idx, float *out, hf_stream_t stream)
{
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0 || width < 3 + c || width % 4 != 0) return HF_EINVAL;
    const long long nrows = static_cast<long long>(b) * m * nsample;
    if (nrows == 0) return HF_OK;  // empty tensors carry null pointers
    if (!grouped_xyz || !points || !idx || !out || reinterpret_cast<uintptr_t>(out) % 16 != 0) return HF_EINVAL;
    const int block = 256;
    hipLaunchKernelGGL(group_concat_kernel, dim3(grid_for(nrows * (width / 4), block)), dim3(block), 0, as_stream(stream), n,
                       c, width / 4, xyz_last ? c : 0, xyz_last ? 0 : 3, static_cast<long long>(m) * nsample, nrows, grouped_xyz,
                       points, idx, out);
    return launch_status();
}

HF_API int hf_group_concat_grad(int b, int n, int c, int m, int nsample, int width, int xyz_last, const float *grad_out,
                                const int *idx, float *grad_points, hf_stream_t stream)
{
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0 || width < 3 + c || !grad_points) return HF_EINVAL;
    if (b == 0) return HF_OK;
    hipStream_t st = as_stream(stream);
    int rc = hip_status(hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(b) * n * c, st));
    if (rc != HF_OK) return rc;
    const long long nrows = static_cast<long long>(b) * m * nsample;
    if (nrows == 0) return HF_OK;  // empty tensors carry null pointers
    if (!grad_out || !idx) return HF_EINVAL;
    const int block = 256;
    hipLaunchKernelGGL(group_concat_grad_kernel, dim3(grid_for(nrows * c, block)), dim3(block), 0, st, n, c, width,
                       xyz_last ? 0 : 3, static_cast<long long>(m) * nsample, nrows, grad_out, idx, grad_points);
    return launch_status();
}

HF_API int hf_knn_point(int b, int n, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx,
                        hf_stream_t stream)
{
    if (k <= 0 || b < 0 || n <= 0 || m < 0 || k > n || !xyz1 || !xyz2 || !val || !idx) return HF_EINVAL;
    if (b == 0 || m == 0) return HF_OK;
    const size_t lds = sizeof(float) * kKnnTile * 3 + (sizeof(float) + sizeof(int)) * static_cast<size_t>(k) * kKnnThreads;
    if (lds > 150 * 1024 || b > 65535) return HF_EINVAL;  // k <= 67 (the reference uses 8..32)
    hipStream_t st = as_stream(stream);
    // enough waves to fill the chip: one thread per query when there are many queries, else 4 or 16 lanes per query
    const long long queries = static_cast<long long>(b) * m;
#define HF_KNN_LAUNCH(P)                                                                                              \
    do {                                                                                                              \
        if (const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(&knn_kernel<P>), lds); lrc != HF_OK) return lrc; \
        hipLaunchKernelGGL((knn_kernel<P>), dim3(div_up(m, kKnnThreads / P), b), dim3(kKnnThreads), lds, st, n, m, k,   \
                           xyz1, xyz2, val, idx);                                                                     \
    } while (0)
    // small clouds with one of the shipped neighbourhood sizes: the register-list kernel (n <= 4096: beyond that the tile loop of
    // one thread per query is slower than the 4- / 16-lane forms below)
    if (n <= 4096 && (k == 4 || k == 8 || k == 12 || k == 16) && static_cast<long long>(div_up(m, kKnnThreads)) * b >= 256) {
        const dim3 grid(div_up(m, kKnnThreads), b);
        if (k == 4) hipLaunchKernelGGL((knn_small_kernel<4>), grid, dim3(kKnnThreads), 0, st, n, m, xyz1, xyz2, val, idx);
        else if (k == 8) hipLaunchKernelGGL((knn_small_kernel<8>), grid, dim3(kKnnThreads), 0, st, n, m, xyz1, xyz2, val, idx);
        else if (k == 12) hipLaunchKernelGGL((knn_small_kernel<12>), grid, dim3(kKnnThreads), 0, st, n, m, xyz1, xyz2, val, idx);
        else hipLaunchKernelGGL((knn_small_kernel<16>), grid, dim3(kKnnThreads), 0, st, n, m, xyz1, xyz2, val, idx);
        return launch_status();
    }
    if (queries >= 256 * 1024) HF_KNN_LAUNCH(1);
    else if (queries >= 64 * 1024) HF_KNN_LAUNCH(4);
    else HF_KNN_LAUNCH(16);
#undef HF_KNN_LAUNCH
    return launch_status();
}

namespace hf {
size_t knn_grid_workspace(int b, int n)
{
    if (b <= 0 || n <= 0 || n > kKnnBinMaxData) return 0;
    return knn_ws_stride(n, knn_grid_for(n)) * static_cast<size_t>(b);
}

int launch_knn_grid(int b, int n, int m, int k, const float *data, const float *queries, float *val, int *idx,
                    void *workspace, hipStream_t st)
{
    const size_t lds = (sizeof(float) + sizeof(int)) * static_cast<size_t>(k) * kKnnThreads;
    if (lds > 150 * 1024 || b > 65535 || reinterpret_cast<uintptr_t>(workspace) % 16 != 0) return HF_EINVAL;  // k <= 75
    const int g = knn_grid_for(n);
    hipLaunchKernelGGL(knn_bin_kernel, dim3(b), dim3(kKnnBinThreads), 0, st, n, g, data, workspace);
    {
        const dim3 grid(div_up(m, kKnnThreads), b);
#define HF_KNN_REG(KK) hipLaunchKernelGGL((knn_grid_reg_kernel<KK>), grid, dim3(kKnnThreads), 0, st, n, m, g, queries, workspace, val, idx)
        switch (k) {   // the register-list kernel for three_nn and the shipped neighbourhood sizes
        case 3: HF_KNN_REG(3); return launch_status();
        case 4: HF_KNN_REG(4); return launch_status();
        case 8: HF_KNN_REG(8); return launch_status();
        case 12: HF_KNN_REG(12); return launch_status();
        case 16: HF_KNN_REG(16); return launch_status();
        default: break;
        }
#undef HF_KNN_REG
    }
    if (const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(&knn_grid_kernel), lds); lrc != HF_OK) return lrc;
    hipLaunchKernelGGL(knn_grid_kernel, dim3(div_up(m, kKnnThreads), b), dim3(kKnnThreads), lds, st, n, m, k, g, queries,
                       workspace, val, idx);
    return launch_status();
}
}  // namespace hf

HF_API size_t hf_knn_workspace(int b, int n)
{
    return knn_grid_workspace(b, n);  // 0: larger clouds use the tiled all-pairs kernel
}

HF_API int hf_knn_point_sorted(int b, int n, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx,
                               void *workspace, size_t workspace_bytes, hf_stream_t stream)
{
    if (k <= 0 || b < 0 || n <= 0 || m < 0 || k > n || !xyz1 || !xyz2 || !val || !idx) return HF_EINVAL;
    if (b == 0 || m == 0) return HF_OK;
    const size_t need = hf_knn_workspace(b, n);
    if (need == 0) return hf_knn_point(b, n, m, k, xyz1, xyz2, val, idx, stream);
    if (!workspace || workspace_bytes < need) return HF_EWORKSPACE;
    return launch_knn_grid(b, n, m, k, xyz1, xyz2, val, idx, workspace, as_stream(stream));
}

HF_API int hf_select_top_k(int b, int n, int m, int k, const float *dist, int *outi, float *out, hf_stream_t stream)
{
    // SelectionSortGpuOp: k > 0 (tf_grouping.cpp:113), (b,m,n) dist (:118)
    if (k <= 0 || b < 0 || n <= 0 || m < 0 || !dist || !outi || !out) return HF_EINVAL;
    const long long nrows = static_cast<long long>(b) * m;
    if (nrows == 0) return HF_OK;
    hipLaunchKernelGGL(select_top_k_kernel, dim3(grid_for(nrows, 64)), dim3(64), 0, as_stream(stream), n, nrows, k,
                       dist, outi, out);
    return launch_status();
}
