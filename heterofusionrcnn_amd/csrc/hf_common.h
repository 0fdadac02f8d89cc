// hf_common.h -- shared helpers for the libhfops.so HIP sources (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hfops.h"

#define HF_API extern "C" __attribute__((visibility("default")))

namespace hf {

extern thread_local int g_last_hip_error;

inline hipStream_t as_stream(hf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int hip_status(hipError_t e)
{
    if (e == hipSuccess) return HF_OK;
    g_last_hip_error = static_cast<int>(e);
    return HF_EHIP;
}

// hf_api.cpp: opt a kernel in to `bytes` of dynamic LDS (> 48 KB needs hipFuncAttributeMaxDynamicSharedMemorySize) ON THE
// CURRENT DEVICE.  The grant is remembered per (device, kernel): a process that drives several devices raises the limit on
// each of them, and a launch path pays one hipGetDevice + a table lookup instead of an attribute call.  HF_EHIP if the
// runtime refuses.  Call it on every launch path whose LDS request can exceed 48 KB.
int ensure_dynamic_lds(const void *kernel, size_t bytes);

// BatchNorm partial-sum layout shared by mlp.hip and gemm.hip: partial[ch * kBnMaxBlocks + blk] = sum,
// partial[(c + ch) * kBnMaxBlocks + blk] = sum of squares, blk < nblk <= kBnMaxBlocks
constexpr int kBnMaxBlocks = 2048;
// mlp.hip: fp64 reduction of the partials -> batch mean / invstd, running statistics update
void launch_bn_stats_finalize(long long rows, int c, int nblk, const float *partial, float eps, float momentum,
                              float *running_mean, float *running_var, float *save_mean, float *save_invstd,
                              hipStream_t st);
// mlp.hip: fp64 reduction of BN-backward partials (sum dh, sum dh*xhat) -> dbeta, dgamma
void launch_bn_bwd_finalize(int c, int nblk, const float *partial, float *dgamma, float *dbeta, hipStream_t st);
// out[e] = sum over chunks of partial[chunk * total + e], chunks added in a fixed order (gemm.hip)
void launch_partial_reduce(int total, int chunks, const float *partial, float *out, hipStream_t st);

// grouping.hip: k nearest data points of every query on a 2-D grid (binning + ring search), any k <= 75; fewer than k
// data points leave (+inf, index 0) in the unfilled slots.  workspace: knn_grid_workspace(b, n) bytes, 16-byte aligned.
size_t knn_grid_workspace(int b, int n);
int launch_knn_grid(int b, int n, int m, int k, const float *data, const float *queries, float *val, int *idx,
                    void *workspace, hipStream_t st);

// ballquery.hip: the cell ball-query kernel; HF_EINVAL = shape outside its range (caller falls back)
int launch_ball_query_cell(int b, int n, int m, float radius, float thresh, int nsample, const float *xyz1,
                           const float *xyz2, int center, int *idx, int *pts_cnt, float *grouped, hipStream_t st);

// ballquery_sorted.hip: the cell-sorted pair of kernels (build once per cloud + query); HF_EINVAL = shape outside its range,
// HF_EWORKSPACE = workspace missing / too small (ball_query_sorted_workspace(b, n) bytes, 16-byte aligned)
size_t ball_query_sorted_workspace(int b, int n);
int launch_ball_query_sorted(int b, int n, int m, float radius, float thresh, int nsample, const float *xyz1, const float *xyz2,
                             int center, int *idx, int *pts_cnt, float *grouped, void *workspace, size_t workspace_bytes,
                             hipStream_t st);

// Diagnostic knobs (phase exits, forced tile shapes, ...) exist only in builds with -DHF_DIAG (scripts/probes/build_diag.sh);
// the product library reads no environment variable on any launch path: HF_DIAG_INT(name, dflt) is the constant dflt.
#ifdef HF_DIAG
#include <stdlib.h>
inline int hf_diag_env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e && e[0] ? atoi(e) : dflt;
}
#define HF_DIAG_INT(name, dflt) hf_diag_env_int(name, dflt)
#else
#define HF_DIAG_INT(name, dflt) (dflt)
#endif

// status of the launch that was just enqueued (no synchronisation)
inline int launch_status() { return hip_status(hipGetLastError()); }

constexpr int kWave = 64;        // CDNA wavefront
constexpr int kNumCU = 256;      // MI355X
constexpr int kNumXCD = 8;

inline int div_up(long long a, long long b) { return static_cast<int>((a + b - 1) / b); }

// ---- wave64 cross-lane primitives (DPP) ----
// old = 0 + bound_ctrl lets hipcc fold the move into v_max_u32_dpp / v_min_u32_dpp.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xf, 0xf, true));
}

// max / min over each 16-lane row; every lane of the row receives the result
__device__ __forceinline__ unsigned row_max_u32(unsigned v)
{
    v = max(v, dpp_u32<0xB1>(v));    // quad_perm [1,0,3,2]
    v = max(v, dpp_u32<0x4E>(v));    // quad_perm [2,3,0,1]
    v = max(v, dpp_u32<0x124>(v));   // row_ror:4
    v = max(v, dpp_u32<0x128>(v));   // row_ror:8
    return v;
}
__device__ __forceinline__ unsigned row_min_u32(unsigned v)
{
    v = min(v, dpp_u32<0xB1>(v));
    v = min(v, dpp_u32<0x4E>(v));
    v = min(v, dpp_u32<0x124>(v));
    v = min(v, dpp_u32<0x128>(v));
    return v;
}

// max / min over the 64 lanes of a wave, returned wave-uniform (SGPR): 4 DPP steps, then the
// four row results are combined on the scalar unit.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    v = row_max_u32(v);
    const unsigned a = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 0));
    const unsigned b = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 16));
    const unsigned c = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 32));
    const unsigned d = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 48));
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    v = row_min_u32(v);
    const unsigned a = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 0));
    const unsigned b = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 16));
    const unsigned c = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 32));
    const unsigned d = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 48));
    return min(min(a, b), min(c, d));
}

__device__ __forceinline__ unsigned f2ord(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone float -> uint
}
__device__ __forceinline__ float ord2f(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// tf.nn.elu and its slope (pointfly.py:480-497: the activation of every PointCNN layer, applied BEFORE the batch norm)
__device__ __forceinline__ float elu_fwd(float x) { return x > 0.0f ? x : expm1f(x); }
__device__ __forceinline__ float elu_slope(float x) { return x > 0.0f ? 1.0f : expf(x); }

// exclusive prefix sum of one int per thread over a 1024-thread workgroup; `wsum` = 16 ints of LDS
__device__ __forceinline__ int block_exclusive_scan(int v, int *wsum, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int x = wsum[w];
        if (w < wave) base += x;
        tot += x;
    }
    __syncthreads();  // wsum reusable
    if (total) *total = tot;
    return base + inc - v;
}

// inclusive prefix sum over the 64 lanes of a wave: Hillis-Steele inside each 16-lane row (row_shr, lanes shifted in
// from outside the row read 0), then row_bcast:15 / row_bcast:31 carry the row totals across rows.  6 DPP adds,
// no LDS crossbar (a __shfl_up chain is 6 ds_bpermute round trips).
__device__ __forceinline__ int wave_inclusive_scan_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

__device__ __forceinline__ int lane_id() { return static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))); }

// number of set bits of `mask` below this lane
__device__ __forceinline__ int mask_prefix(unsigned long long mask)
{
    return static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mask >> 32),
                                                      __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mask), 0u)));
}

__device__ __forceinline__ float readlane_f(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

}  // namespace hf
