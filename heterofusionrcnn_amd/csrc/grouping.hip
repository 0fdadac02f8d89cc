// grouping.hip -- query_ball_point / group_point (+grad) / fused ball-group / select_top_k for gfx950.
//
// Replaces grouping/tf_grouping_g.cu:3-123 of the reference (launchers :125-141).
//
// The hit test of the reference is  max(sqrtf(s),1e-20f) < radius  with
// s = (x2-x1)^2+(y2-y1)^2+(z2-z1)^2 (tf_grouping_g.cu:24-25).  sqrtf is correctly rounded and
// monotone, so there is one float T with  hit <=> s < T ; the host computes T exactly
// (ball_threshold) and the kernels never take a square root.
#include <math.h>

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

static int grid_for(long long work_items, int block)
{
    long long g = (work_items + block - 1) / block;
    const long long cap = static_cast<long long>(kNumCU) * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return static_cast<int>(g);
}

static int launch_ball_query(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2,
                             int center, int *idx, int *pts_cnt, float *grouped, hipStream_t st)
{
    const float thresh = ball_threshold(radius);
    const size_t lds = sizeof(float) * kQbTile * 3 + sizeof(int) * kQbThreads * (static_cast<size_t>(nsample | 1) + 1);
    if (lds > 160 * 1024) return HF_EINVAL;  // nsample > ~148: not used by any config
    dim3 grid(div_up(m, kQbThreads), b);
    if (grouped) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qbp_bruteforce_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        hipLaunchKernelGGL((qbp_bruteforce_kernel<true>), grid, dim3(kQbThreads), lds, st, n, m, thresh, nsample, xyz1,
                           xyz2, center, idx, pts_cnt, grouped);
    } else {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qbp_bruteforce_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        hipLaunchKernelGGL((qbp_bruteforce_kernel<false>), grid, dim3(kQbThreads), lds, st, n, m, thresh, nsample,
                           xyz1, xyz2, center, idx, pts_cnt, grouped);
    }
    return launch_status();
}

}  // namespace hf

using namespace hf;

HF_API int hf_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2,
                               int *idx, int *pts_cnt, hf_stream_t stream)
{
    // QueryBallPointGpuOp: radius > 0, nsample > 0 (tf_grouping.cpp:70-74); (b,n,3)/(b,m,3) (:79-85)
    if (!(radius > 0.0f) || nsample <= 0 || b < 0 || n <= 0 || m < 0 || !xyz1 || !xyz2 || !idx) return HF_EINVAL;
    if (b == 0 || m == 0) return HF_OK;
    return launch_ball_query(b, n, m, radius, nsample, xyz1, xyz2, 0, idx, pts_cnt, nullptr, as_stream(stream));
}

HF_API int hf_query_ball_group_xyz(int b, int n, int m, float radius, int nsample, const float *xyz1,
                                   const float *xyz2, int center, int *idx, int *pts_cnt, float *grouped_xyz,
                                   hf_stream_t stream)
{
    if (!(radius > 0.0f) || nsample <= 0 || b < 0 || n <= 0 || m < 0 || !xyz1 || !xyz2 || !grouped_xyz)
        return HF_EINVAL;
    if (b == 0 || m == 0) return HF_OK;
    return launch_ball_query(b, n, m, radius, nsample, xyz1, xyz2, center, idx, pts_cnt, grouped_xyz,
                             as_stream(stream));
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
    if (c % 4 == 0 && al16)
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
