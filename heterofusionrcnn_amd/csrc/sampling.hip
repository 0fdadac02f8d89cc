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
#include "hf_common.h"

namespace hf {

// smaller key wins a distance tie
__device__ __forceinline__ unsigned fps_tiekey(int k) { return (static_cast<unsigned>(k & 511) << 22) | static_cast<unsigned>(k >> 9); }

constexpr int kFpsThreads = 1024;
constexpr int kFpsWaves = kFpsThreads / kWave;
constexpr int kFpsMaxPPT = 16;

struct FpsSlot {
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

// block-level pick among the 16 per-wave slots; every lane of every wave computes the same winner.
// lanes replicate slot (lane & 15) in each DPP row, so 4-step row reductions suffice.
__device__ __forceinline__ int fps_pick_slot(const FpsSlot *slots, int lane)
{
    const int sl = lane & (kFpsWaves - 1);
    const int sd = slots[sl].dist;
    const int sk = slots[sl].k;
    const unsigned gbest = row_max_u32(static_cast<unsigned>(sd + 1));
    const unsigned key = static_cast<unsigned>(sd + 1) == gbest ? fps_tiekey(sk) : 0xffffffffu;
    const unsigned gkey = row_min_u32(key);
    const unsigned long long win = __ballot(key == gkey);
    return __builtin_ctzll(win) & (kFpsWaves - 1);
}

template <int PPT, bool FULL>
__global__ __launch_bounds__(kFpsThreads) void fps_onchip_kernel(int n, int m, const float *__restrict__ xyz,
                                                                 int *__restrict__ out)
{
    __shared__ FpsSlot slots[2][kFpsWaves];
    typedef typename FpsVec<PPT>::type vec_t;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const float *pts = xyz + static_cast<size_t>(blockIdx.x) * n * 3;
    int *o = out + static_cast<size_t>(blockIdx.x) * m;

    // x/y/z as ext vectors: a wave-uniform dynamic index lowers to s_set_gpr_idx (no scratch)
    vec_t x, y, z;
    int td[PPT];  // running min distance as float bits; all values >= +0 so int order == float order
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = t + i * kFpsThreads;
        const int kk = (FULL || k < n) ? k : 0;
        vec_set<PPT>(x, i, pts[kk * 3 + 0]);
        vec_set<PPT>(y, i, pts[kk * 3 + 1]);
        vec_set<PPT>(z, i, pts[kk * 3 + 2]);
        td[i] = __float_as_int(1e38f);
    }
    int nvalid = PPT;  // slots [0,nvalid) of this thread hold real points
    if (!FULL) nvalid = t < n ? (n - 1 - t) / kFpsThreads + 1 : 0;

    float x1 = pts[0], y1 = pts[1], z1 = pts[2];
    if (t == 0) o[0] = 0;

    for (int j = 1; j < m; ++j) {
        int best = -1, bi = 0;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const float dx = vec_get<PPT>(x, i) - x1, dy = vec_get<PPT>(y, i) - y1, dz = vec_get<PPT>(z, i) - z1;
            const float d = dx * dx + dy * dy + dz * dz;
            int s = min(__float_as_int(d), td[i]);  // == fminf on non-negative floats
            td[i] = s;
            if (!FULL) s = i < nvalid ? s : -1;
            if (s > best) { best = s; bi = i; }  // strict: the smallest k wins inside a thread
        }
        // wave arg-max; among equal lanes the lowest lane has the smallest (k mod 512)
        const int wbest = static_cast<int>(wave_max_u32(static_cast<unsigned>(best + 1))) - 1;
        const unsigned long long tied = __ballot(best == wbest);
        const int wl = __builtin_ctzll(tied);
        const int wi = __builtin_amdgcn_readlane(bi, wl);
        const float cx = vec_get<PPT>(x, wi), cy = vec_get<PPT>(y, wi), cz = vec_get<PPT>(z, wi);
        FpsSlot *cur = slots[j & 1];
        if (lane == wl) {
            cur[wave].dist = wbest;
            cur[wave].k = t + wi * kFpsThreads;
            cur[wave].x = cx;
            cur[wave].y = cy;
            cur[wave].z = cz;
        }
        __syncthreads();
        const int ws = fps_pick_slot(cur, lane);
        x1 = cur[ws].x;
        y1 = cur[ws].y;
        z1 = cur[ws].z;
        if (t == 0) o[j] = cur[ws].k;
    }
}

// Fallback for clouds larger than the on-chip limit: running distances in caller scratch
// (b,n) floats, points re-read from global (L2-resident).  Same selection rule.
__global__ __launch_bounds__(kFpsThreads) void fps_scratch_kernel(int n, int m, const float *__restrict__ xyz,
                                                                  float *__restrict__ temp, int *__restrict__ out)
{
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
        const int ws = fps_pick_slot(slots[par], lane);
        const int old = slots[par][ws].k;
        x1 = slots[par][ws].x;
        y1 = slots[par][ws].y;
        z1 = slots[par][ws].z;
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

template <int PPT>
static int launch_fps_onchip(int b, int n, int m, const float *inp, int *out, hipStream_t st)
{
    if (n == PPT * kFpsThreads)
        hipLaunchKernelGGL((fps_onchip_kernel<PPT, true>), dim3(b), dim3(kFpsThreads), 0, st, n, m, inp, out);
    else
        hipLaunchKernelGGL((fps_onchip_kernel<PPT, false>), dim3(b), dim3(kFpsThreads), 0, st, n, m, inp, out);
    return launch_status();
}

}  // namespace hf

using namespace hf;

HF_API int hf_fps_onchip_limit(void) { return kFpsMaxPPT * kFpsThreads; }

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
    const int ppt = div_up(n, kFpsThreads);
    if (ppt <= 1) return launch_fps_onchip<1>(b, n, m, inp, out, st);
    if (ppt <= 2) return launch_fps_onchip<2>(b, n, m, inp, out, st);
    if (ppt <= 4) return launch_fps_onchip<4>(b, n, m, inp, out, st);
    if (ppt <= 8) return launch_fps_onchip<8>(b, n, m, inp, out, st);
    if (ppt <= 16) return launch_fps_onchip<16>(b, n, m, inp, out, st);
    if (!temp) return HF_EWORKSPACE;
    hipLaunchKernelGGL(fps_scratch_kernel, dim3(b), dim3(kFpsThreads), 0, st, n, m, inp, temp, out);
    return launch_status();
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
