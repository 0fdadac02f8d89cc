// xconv.hip -- the two per-point products of PointCNN's X-Conv (hf/core/feature_extractors/pointcnn.py:16-151) for gfx950.
//
//   X-transform apply   F_X = X x F_*      pointcnn.py:133  tf.matmul(X, nn_fts_input): per representative point a (K,K)
//                                          matrix times the (K,C) block of lifted + gathered neighbour features
//   depthwise (1,K)     pointfly.py:437-457 depthwise_conv2d / the depthwise half of separable_conv2d with a (1,K) window
//                                          over a width-K input: out[c*M + m] = sum_w in[w][c] * W[w][c][m]
//
// In the reference both are library calls on tiny operands: a batched GEMM with 131 072 batches of 8x8 @ 8xC, and a
// depthwise convolution whose window covers its whole input.  The framework route costs a third of the PointCNN RPN
// step (batched-GEMM kernels at ~1.4 ms per layer, strided copies around every einsum).  Both are memory-bound by
// nature: every operand is read once and every result written once, the K*K (or K*M) coefficients of a point live in
// scalar registers / LDS.  fp32, multiply then add in index order (no FMA contraction), like the rest of the library.
#include "hf_common.h"

namespace hf {

// The M consecutive floats a (row, channel) pair owns in a (rows, C*M) tensor.  As M scalar accesses a wave's instruction touches
// 64 x 4 bytes at a stride of 4 M bytes -- every cache line of the row is visited M times (the K = 4, M = 4 layer of the RCNN wrote
// its 2.4 GB at 1.2 TB/s that way); as ONE 8- / 16-byte access per lane the instruction covers a contiguous range.  `base` is
// the tensor's first element (a kernel argument: the alignment test is wave-uniform).
template <int M>
__device__ __forceinline__ void store_m(float *base, size_t elem, const float (&v)[M])
{
    float *p = base + elem;
    if constexpr (M == 4 || M == 8) {
        if ((reinterpret_cast<uintptr_t>(base) & 15) == 0) {
#pragma unroll
            for (int q = 0; q < M / 4; ++q) reinterpret_cast<float4 *>(p)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            return;
        }
    } else if constexpr (M == 2) {
        if ((reinterpret_cast<uintptr_t>(base) & 7) == 0) { *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]); return; }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) p[m] = v[m];
}
template <int M>
__device__ __forceinline__ void load_m(const float *base, size_t elem, float (&v)[M])
{
    const float *p = base + elem;
    if constexpr (M == 4 || M == 8) {
        if ((reinterpret_cast<uintptr_t>(base) & 15) == 0) {
#pragma unroll
            for (int q = 0; q < M / 4; ++q) {
                const float4 t = reinterpret_cast<const float4 *>(p)[q];
                v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
            }
            return;
        }
    } else if constexpr (M == 2) {
        if ((reinterpret_cast<uintptr_t>(base) & 7) == 0) { const float2 t = *reinterpret_cast<const float2 *>(p); v[0] = t.x; v[1] = t.y; return; }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) v[m] = p[m];
}


constexpr int kXcThreads = 256;

// out[r][i][ch] = sum_j X[r][i][j] * F[r][j][ch]   (TRANSPOSED: sum_j X[r][j][i] * F[r][j][ch], the dF of the backward)
// One wave per row at a time: the row's K*K coefficients are wave-uniform (scalar loads), lanes walk the channels.
template <int K, bool TRANSPOSED>
__global__ __launch_bounds__(kXcThreads) void xconv_apply_kernel(long long rows, int c, const float *__restrict__ x,
                                                                const float *__restrict__ f, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long long wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)) +
                           static_cast<long long>(blockIdx.x) * (kXcThreads / 64);
    const long long nwaves = static_cast<long long>(gridDim.x) * (kXcThreads / 64);
    for (long long r = wave; r < rows; r += nwaves) {
        const float *xr = x + r * (K * K);
        float coef[K][K];
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int j = 0; j < K; ++j) coef[i][j] = TRANSPOSED ? xr[j * K + i] : xr[i * K + j];
        const float *fr = f + r * K * c;
        float *orow = out + r * K * c;
        for (int ch = lane; ch < c; ch += 64) {
            float v[K];
#pragma unroll
            for (int j = 0; j < K; ++j) v[j] = fr[static_cast<size_t>(j) * c + ch];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float acc = coef[i][0] * v[0];
#pragma unroll
                for (int j = 1; j < K; ++j) acc = acc + coef[i][j] * v[j];
                orow[static_cast<size_t>(i) * c + ch] = acc;
            }
        }
    }
}

// dX[r][i][j] = sum_ch dO[r][i][ch] * F[r][j][ch]: a wave per row; every lane accumulates the K*K products of its
// channels, the 64 partial matrices meet in LDS (stride K*K + 1: conflict-free both ways) and lane p sums entry p.
template <int K>
__global__ __launch_bounds__(kXcThreads) void xconv_dx_kernel(long long rows, int c, const float *__restrict__ grad_out,
                                                             const float *__restrict__ f, float *__restrict__ grad_x)
{
    extern __shared__ float lds[];
    constexpr int KK = K * K, STRIDE = KK + 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float *mine = lds + static_cast<size_t>(w) * 64 * STRIDE;
    const long long wave = w + static_cast<long long>(blockIdx.x) * (kXcThreads / 64);
    const long long nwaves = static_cast<long long>(gridDim.x) * (kXcThreads / 64);
    for (long long r = wave; r < rows; r += nwaves) {
        const float *gr = grad_out + r * K * c;
        const float *fr = f + r * K * c;
        float acc[K][K];
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int j = 0; j < K; ++j) acc[i][j] = 0.f;
        for (int ch = lane; ch < c; ch += 64) {
            float g[K], v[K];
#pragma unroll
            for (int j = 0; j < K; ++j) { g[j] = gr[static_cast<size_t>(j) * c + ch]; v[j] = fr[static_cast<size_t>(j) * c + ch]; }
#pragma unroll
            for (int i = 0; i < K; ++i)
#pragma unroll
                for (int j = 0; j < K; ++j) acc[i][j] = acc[i][j] + g[i] * v[j];
        }
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int j = 0; j < K; ++j) mine[lane * STRIDE + i * K + j] = acc[i][j];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int p = lane; p < KK; p += 64) {
            float s = 0.f;
            for (int l = 0; l < 64; ++l) s = s + mine[l * STRIDE + p];
            grad_x[r * KK + p] = s;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// depthwise (1,K): y[r][ch*M + m] = sum_w x[r][w][ch] * W[w][ch][m]; one thread per (row, channel)
template <int K, int M>
__global__ __launch_bounds__(kXcThreads) void depthwise_fwd_kernel(long long rows, int c, const float *__restrict__ x,
                                                                  const float *__restrict__ wgt, float *__restrict__ y)
{
    const long long total = rows * c;
    for (long long g = static_cast<long long>(blockIdx.x) * kXcThreads + threadIdx.x; g < total;
         g += static_cast<long long>(gridDim.x) * kXcThreads) {
        const long long r = g / c;
        const int ch = static_cast<int>(g - r * c);
        float acc[M];
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m] = 0.f;
#pragma unroll
        for (int w = 0; w < K; ++w) {
            const float xv = x[(r * K + w) * c + ch];
#pragma unroll
            for (int m = 0; m < M; ++m) acc[m] = acc[m] + xv * wgt[(static_cast<size_t>(w) * c + ch) * M + m];
        }
        store_m<M>(y, static_cast<size_t>(g) * M, acc);
    }
}

// dx[r][w][ch] = sum_m dy[r][ch*M + m] * W[w][ch][m]
template <int K, int M>
__global__ __launch_bounds__(kXcThreads) void depthwise_dx_kernel(long long rows, int c, const float *__restrict__ grad_y,
                                                                 const float *__restrict__ wgt, float *__restrict__ grad_x)
{
    const long long total = rows * c;
    for (long long g = static_cast<long long>(blockIdx.x) * kXcThreads + threadIdx.x; g < total;
         g += static_cast<long long>(gridDim.x) * kXcThreads) {
        const long long r = g / c;
        const int ch = static_cast<int>(g - r * c);
        float gy[M];
        load_m<M>(grad_y, static_cast<size_t>(g) * M, gy);
#pragma unroll
        for (int w = 0; w < K; ++w) {
            float acc = 0.f;
#pragma unroll
            for (int m = 0; m < M; ++m) acc = acc + gy[m] * wgt[(static_cast<size_t>(w) * c + ch) * M + m];
            grad_x[(r * K + w) * c + ch] = acc;
        }
    }
}

// Narrow layers (the X-transformation's own depthwise steps, pointcnn.py:107-131: c = K channels, depth multiplier K): with the
// grid a multiple of c threads, a thread keeps ONE channel for all its rows, and its K x M weights are loaded once into registers.
// The kernels above re-read them through the vector memory path for every row -- 64 loads per (row, channel) at K = M = 8: 85 us
// forward / 150 us input gradient for 131 072 rows x 8 channels (33 MB in, 33 MB out).  Same sums in the same order.
template <int K, int M, bool DX>
__global__ __launch_bounds__(kXcThreads) void depthwise_narrow_kernel(long long rows, int c, const float *__restrict__ in,
                                                                     const float *__restrict__ wgt, float *__restrict__ out)
{
    const long long tid = static_cast<long long>(blockIdx.x) * kXcThreads + threadIdx.x;
    const long long nthreads = static_cast<long long>(gridDim.x) * kXcThreads;   // a multiple of c (the launcher sees to it)
    const int ch = static_cast<int>(tid % c);
    float wr[K][M];
#pragma unroll
    for (int w = 0; w < K; ++w)
#pragma unroll
        for (int m = 0; m < M; ++m) wr[w][m] = wgt[(static_cast<size_t>(w) * c + ch) * M + m];
    for (long long r = tid / c; r < rows; r += nthreads / c) {
        if (!DX) {
            // y[r][ch*M + m] = sum_w x[r][w][ch] * W[w][ch][m]
            float acc[M];
#pragma unroll
            for (int m = 0; m < M; ++m) acc[m] = 0.f;
#pragma unroll
            for (int w = 0; w < K; ++w) {
                const float xv = in[(r * K + w) * c + ch];
#pragma unroll
                for (int m = 0; m < M; ++m) acc[m] = acc[m] + xv * wr[w][m];
            }
            store_m<M>(out, static_cast<size_t>(r * c + ch) * M, acc);
        } else {
            // dx[r][w][ch] = sum_m dy[r][ch*M + m] * W[w][ch][m]
            float gy[M];
            load_m<M>(in, static_cast<size_t>(r * c + ch) * M, gy);
#pragma unroll
            for (int w = 0; w < K; ++w) {
                float acc = 0.f;
#pragma unroll
                for (int m = 0; m < M; ++m) acc = acc + gy[m] * wr[w][m];
                out[(r * K + w) * c + ch] = acc;
            }
        }
    }
}

// grid for the narrow kernels: whole multiples of c threads, enough workgroups for the rows
static int narrow_grid(long long rows, int c)
{
    // kXcThreads * g must be a multiple of c: g a multiple of c / gcd(c, kXcThreads)
    int a = c, b = kXcThreads;
    while (b) { const int tmp = a % b; a = b; b = tmp; }
    const int unit = c / a;
    long long g = (rows * c + kXcThreads - 1) / kXcThreads;
    const long long cap = static_cast<long long>(kNumCU) * 8;
    if (g > cap) g = cap;
    g = (g + unit - 1) / unit * unit;
    return static_cast<int>(g);
}
constexpr int kNarrowMaxC = 64;

// dW[w][ch][m] = sum_r x[r][w][ch] * dy[r][ch*M + m]: thread (row chunk, channel) sums its rows in registers, then one
// atomic per coefficient and chunk (grad_w zero-filled by the entry point).  The order of the chunks is not fixed: the
// sum is reproducible to fp32 rounding, not bit for bit.
//
// `partial` != NULL (hf_depthwise_k_grad_ws): every chunk writes its K*c*M sums to partial[chunk][...] instead and a second
// kernel adds the chunks in a fixed order -- 512 chunks x 512 atomics on the same 16 cache lines were 50 us of a 60 us call
// for the X-transform's 8-channel layers, whatever the number of rows; deterministic as a bonus.
template <int K, int M>
__global__ __launch_bounds__(kXcThreads) void depthwise_dw_kernel(long long rows, int c, int rows_per_chunk,
                                                                 const float *__restrict__ x, const float *__restrict__ grad_y,
                                                                 float *__restrict__ grad_w, float *__restrict__ partial)
{
    // narrow layers (the X-transform has 8 channels): the threads a block has beyond the channels split the chunk's rows,
    // their partial sums meet in LDS, and the block issues ONE atomic per coefficient (8 x 8 x 8 coefficients hit by a
    // thousand blocks x 32 row slots each were 16 M atomics on 512 addresses: 7.6 ms)
    extern __shared__ float red[];   // [K * M][cw] when nrs > 1
    const int cw = c < kXcThreads ? c : kXcThreads, nrs = kXcThreads / cw;
    const int t = static_cast<int>(threadIdx.x);
    const int cl = t % cw, ch = blockIdx.x * cw + cl, rs = t / cw;
    const bool live = ch < c && rs < nrs;
    if (nrs > 1) {
        for (int i = t; i < K * M * cw; i += kXcThreads) red[i] = 0.f;
        __syncthreads();
    }
    const long long r0 = static_cast<long long>(blockIdx.y) * rows_per_chunk + rs;
    const long long r1 = static_cast<long long>(blockIdx.y + 1) * rows_per_chunk < rows ? static_cast<long long>(blockIdx.y + 1) * rows_per_chunk : rows;
    float acc[K][M];
#pragma unroll
    for (int w = 0; w < K; ++w)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[w][m] = 0.f;
    if (live) {
        for (long long r = r0; r < r1; r += nrs) {
            float gy[M];
            load_m<M>(grad_y, static_cast<size_t>(r * c + ch) * M, gy);
#pragma unroll
            for (int w = 0; w < K; ++w) {
                const float xv = x[(r * K + w) * c + ch];
#pragma unroll
                for (int m = 0; m < M; ++m) acc[w][m] = acc[w][m] + xv * gy[m];
            }
        }
    }
    float *mine = partial ? partial + static_cast<size_t>(blockIdx.y) * K * c * M : nullptr;
    if (nrs == 1) {
        if (live) {
#pragma unroll
            for (int w = 0; w < K; ++w)
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const size_t e = (static_cast<size_t>(w) * c + ch) * M + m;
                    if (mine) mine[e] = acc[w][m];
                    else atomicAdd(&grad_w[e], acc[w][m]);
                }
        }
        return;
    }
    // the row slots of a channel meet: inside a wave by xor-shuffles (power-of-two channel counts up to 32: the lanes of a
    // wave are 64 / cw row slots x cw channels), then one LDS slot per wave.  LDS atomics here -- 64 per thread, eight lanes of
    // a wave on every address -- were 13 us of serialised LDS work per block, the whole cost of the 8-channel layers.
    const bool shuffle = (cw & (cw - 1)) == 0 && cw <= 32;
    if (shuffle) {
        const int lane = t & 63, wave = t >> 6;
        __syncthreads();   // the zero fill above is not needed on this path, but every thread must be past it
#pragma unroll
        for (int w = 0; w < K; ++w)
#pragma unroll
            for (int m = 0; m < M; ++m) {
                float v = live ? acc[w][m] : 0.f;
                for (int off = cw; off < 64; off <<= 1) v += __shfl_xor(v, off);
                if (lane < cw) red[(wave * K * M + w * M + m) * cw + lane] = v;
            }
        __syncthreads();
        for (int i = t; i < K * M * cw; i += kXcThreads) {
            float v = red[i];
#pragma unroll
            for (int wv = 1; wv < kXcThreads / 64; ++wv) v += red[wv * K * M * cw + i];
            const int wm = i / cw, c2 = blockIdx.x * cw + i % cw;
            if (c2 < c) {
                const size_t e = (static_cast<size_t>(wm / M) * c + c2) * M + wm % M;
                if (mine) mine[e] = v;
                else atomicAdd(&grad_w[e], v);
            }
        }
        return;
    }
    if (live) {
#pragma unroll
        for (int w = 0; w < K; ++w)
#pragma unroll
            for (int m = 0; m < M; ++m) atomicAdd(&red[(w * M + m) * cw + cl], acc[w][m]);
    }
    __syncthreads();
    for (int i = t; i < K * M * cw; i += kXcThreads) {
        const int wm = i / cw, c2 = blockIdx.x * cw + i % cw;
        if (c2 < c) {
            const size_t e = (static_cast<size_t>(wm / M) * c + c2) * M + wm % M;
            if (mine) mine[e] = red[i];
            else atomicAdd(&grad_w[e], red[i]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// X-transform apply and the depthwise step of the separable convolution in one pass (pointcnn.py:133-140):
//   out[r][ch*M + m] = sum_k (sum_j X[r][k][j] * F[r][j][ch]) * Wd[k][ch][m]
// F_X = X x F_* (rows x K x C, the largest tensor of an X-Conv: 1.3 GB for the decoder layers at 131 072 points) is never
// written: the K products of a (row, channel) stay in registers.  Same multiply / add order as hf_xconv_apply followed by
// hf_depthwise_k, so the results are bit-identical to the two-kernel route.
// Mapping: a lane owns ONE channel for the whole kernel (its K x M depthwise weights live in registers, blockIdx.y picks
// the 64-channel chunk), the four waves of a block walk the rows of a chunk; the row's K x K matrix is wave-uniform.
//
// GATHER: F_* = [F_delta | F gathered] (pointcnn.py:124-127 concat of the lifted coordinates with the neighbours' features) is
// not materialised either: channels below c0 come from `f` = F_delta (rows x K x c0), the others from the feature table
// `fts` (clouds x n_src rows x c - c0) through the neighbour table `idx` (rows x K, indices inside the row's cloud).  c0 is a multiple of 64, so a block's 64-channel
// chunk lies on one side.  The gathered rows (K x the table, 1.1 GB for the last decoder layers) were written by
// group_point and read back here; now the table (134 MB, L2 / MALL resident) is read in place.
struct XcSource {
    const float *base;   // + the lane's channel
    long long stride;
    bool gathered;
};

__device__ __forceinline__ XcSource xc_source(int ch, int c, int c0, const float *f, const float *fts, const int *idx)
{
    XcSource s;
    if (idx == nullptr) { s.base = f + ch; s.stride = c; s.gathered = false; }
    else if (ch - (ch & 63) < c0) { s.base = f + ch; s.stride = c0; s.gathered = false; }
    else { s.base = fts + (ch - c0); s.stride = c - c0; s.gathered = true; }
    return s;
}

template <int K, int M, bool GATHER>
__global__ __launch_bounds__(kXcThreads) void xconv_dw_fwd_kernel(long long rows, int c, int c0, int rows_per_block,
                                                                 const float *__restrict__ x, const float *__restrict__ f,
                                                                 const float *__restrict__ fts, const int *__restrict__ idx, int n_src,
                                                                 int rows_per_cloud, const float *__restrict__ wd, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int ch = blockIdx.y * 64 + lane;
    const bool live = ch < c;
    const XcSource src = xc_source(live ? ch : blockIdx.y * 64, c, c0, f, fts, GATHER ? idx : nullptr);
    float w[K][M];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int m = 0; m < M; ++m) w[k][m] = live ? wd[(static_cast<size_t>(k) * c + ch) * M + m] : 0.f;
    const long long r0 = static_cast<long long>(blockIdx.x) * rows_per_block;
    const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    for (long long r = r0 + wave; r < r1; r += kXcThreads / 64) {
        const float *xr = x + r * (K * K);
        float fv[K];
        const long long tbase = (GATHER && src.gathered) ? static_cast<long long>(static_cast<unsigned>(r) / static_cast<unsigned>(rows_per_cloud)) * n_src : 0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const long long srow = (GATHER && src.gathered) ? tbase + idx[r * K + j] : r * K + j;
            fv[j] = live ? src.base[srow * src.stride] : 0.f;
        }
        float o[M];
#pragma unroll
        for (int m = 0; m < M; ++m) o[m] = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float t = xr[k * K] * fv[0];
#pragma unroll
            for (int j = 1; j < K; ++j) t = t + xr[k * K + j] * fv[j];
#pragma unroll
            for (int m = 0; m < M; ++m) o[m] = o[m] + t * w[k][m];
        }
        if (live) store_m<M>(out, static_cast<size_t>(r * c + ch) * M, o);
    }
}

// gradients w.r.t. F and Wd, same mapping: per (row, channel) the gradient of F_X (K values), F_X itself (recomputed) and
// the K input gradients stay in registers; a lane sums its channel's K x M weight gradients over its rows and adds them
// to grad_wd with one atomic per coefficient (grad_wd zero-filled by the entry point)
// GATHER: grad_f is the gradient of F_delta only (rows x K x c0); the gathered channels' gradient is either written to
// grad_gathered (rows x K x c1, then summed per table row by hf_group_point_grad_gather) or rebuilt per table row by
// xconv_dw_bwd_fts_kernel
template <int K, int M, bool GATHER>
__global__ __launch_bounds__(kXcThreads) void xconv_dw_bwd_fw_kernel(long long rows, int c, int c0, int rows_per_block,
                                                                    const float *__restrict__ x, const float *__restrict__ f,
                                                                    const float *__restrict__ fts, const int *__restrict__ idx, int n_src,
                                                                    int rows_per_cloud, const float *__restrict__ wd,
                                                                    const float *__restrict__ grad_out, float *__restrict__ grad_f,
                                                                    float *__restrict__ grad_gathered, float *__restrict__ grad_wd,
                                                                    float *__restrict__ wd_partial)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int ch = blockIdx.y * 64 + lane;
    const bool live = ch < c;
    const XcSource src = xc_source(live ? ch : blockIdx.y * 64, c, c0, f, fts, GATHER ? idx : nullptr);
    if (GATHER && src.gathered && !grad_wd && !grad_gathered) return;   // nothing of this chunk is asked for (no partial to write either)
    float w[K][M], gw[K][M];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int m = 0; m < M; ++m) { w[k][m] = live ? wd[(static_cast<size_t>(k) * c + ch) * M + m] : 0.f; gw[k][m] = 0.f; }
    const long long r0 = static_cast<long long>(blockIdx.x) * rows_per_block;
    const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    for (long long r = r0 + wave; r < r1; r += kXcThreads / 64) {
        const float *xr = x + r * (K * K);
        float fv[K], g[M], gfx[K];
        const long long tbase = (GATHER && src.gathered) ? static_cast<long long>(static_cast<unsigned>(r) / static_cast<unsigned>(rows_per_cloud)) * n_src : 0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const long long srow = (GATHER && src.gathered) ? tbase + idx[r * K + j] : r * K + j;
            fv[j] = live ? src.base[srow * src.stride] : 0.f;
        }
        load_m<M>(grad_out, static_cast<size_t>(r * c + (live ? ch : 0)) * M, g);
        if (!live) {
#pragma unroll
            for (int m = 0; m < M; ++m) g[m] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float a = 0.f;
#pragma unroll
            for (int m = 0; m < M; ++m) a = a + g[m] * w[k][m];          // = hf_depthwise_k_grad's dx
            gfx[k] = a;
            float t = xr[k * K] * fv[0];                                  // F_X recomputed
#pragma unroll
            for (int j = 1; j < K; ++j) t = t + xr[k * K + j] * fv[j];
#pragma unroll
            for (int m = 0; m < M; ++m) gw[k][m] = gw[k][m] + t * g[m];   // = hf_depthwise_k_grad's dw
        }
        float *gdst = (GATHER && src.gathered) ? (grad_gathered ? grad_gathered + (ch - c0) : nullptr) : (grad_f ? grad_f + ch : nullptr);
        if (live && gdst) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                float a = xr[j] * gfx[0];                                  // = hf_xconv_apply_grad's dF (X transposed)
#pragma unroll
                for (int k = 1; k < K; ++k) a = a + xr[k * K + j] * gfx[k];
                gdst[(r * K + j) * src.stride] = a;
            }
        }
    }
    if (grad_wd) {
        // the waves of a block meet in LDS first: one global atomic per coefficient and BLOCK (per wave it was 8.4 M atomics on
        // 2080 addresses for the first encoder layer: 870 us for 140 us of memory traffic)
        __shared__ float red[K * M][64];
        for (int i = threadIdx.x; i < K * M * 64; i += kXcThreads) (&red[0][0])[i] = 0.f;
        __syncthreads();
        if (live) {
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int m = 0; m < M; ++m) atomicAdd(&red[k * M + m][lane], gw[k][m]);
        }
        __syncthreads();
        // wd_partial: the row chunks' sums are written out (wd_partial[row chunk][K*c*M]) and added in a fixed order by a second
        // kernel; else one global atomic per coefficient and block on grad_wd (zero-filled by the entry point)
        float *mine = wd_partial ? wd_partial + static_cast<size_t>(blockIdx.x) * K * c * M : nullptr;
        for (int i = threadIdx.x; i < K * M * 64; i += kXcThreads) {
            const int km = i >> 6, c2 = blockIdx.y * 64 + (i & 63);
            if (c2 < c) {
                const size_t e = (static_cast<size_t>(km / M) * c + c2) * M + km % M;
                if (mine) mine[e] = red[km][i & 63];
                else atomicAdd(&grad_wd[e], red[km][i & 63]);
            }
        }
    }
}

// gradient w.r.t. X: dX[r][k][j] = sum_ch dF_X[r][k][ch] * F[r][j][ch] with dF_X rebuilt from grad_out and Wd on the fly;
// a wave per row, the 64 partial K x K matrices of the lanes meet in LDS (as xconv_dx_kernel)
template <int K, int M, bool GATHER>
__global__ __launch_bounds__(kXcThreads) void xconv_dw_bwd_x_kernel(long long rows, int c, int c0, const float *__restrict__ f,
                                                                   const float *__restrict__ fts, const int *__restrict__ idx, int n_src,
                                                                   int rows_per_cloud, const float *__restrict__ wd,
                                                                   const float *__restrict__ grad_out, float *__restrict__ grad_x)
{
    // the 64 partial K x K matrices of a wave's lanes are first summed inside each quad of lanes (two DPP adds per entry),
    // the 16 quad sums meet in LDS: a quarter of the LDS of the one-slot-per-lane form (4.2 KB per wave: 9 blocks per CU
    // instead of 2, which is what the gathered loads need to stay in flight) and a quarter of its reads
    extern __shared__ float lds[];
    constexpr int KK = K * K, STRIDE = KK + 1, QUADS = 16;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float *mine = lds + static_cast<size_t>(w) * QUADS * STRIDE;
    const long long wave = w + static_cast<long long>(blockIdx.x) * (kXcThreads / 64);
    const long long nwaves = static_cast<long long>(gridDim.x) * (kXcThreads / 64);
    const int c1 = c - c0;
    for (long long r = wave; r < rows; r += nwaves) {
        float acc[K][K];
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int j = 0; j < K; ++j) acc[i][j] = 0.f;
        long long trow[K];   // the K table rows of this row's neighbours, once per row (wave-uniform)
        if constexpr (GATHER) {
            const long long tbase = static_cast<long long>(static_cast<unsigned>(r) / static_cast<unsigned>(rows_per_cloud)) * n_src;
#pragma unroll
            for (int j = 0; j < K; ++j) trow[j] = tbase + idx[r * K + j];
        }
        for (int ch = lane; ch < c; ch += 64) {
            float fv[K], g[M];
            if constexpr (GATHER) {
                // selects, no branch around the loads
                const bool gathered = ch >= c0;
                const float *base = gathered ? fts + (ch - c0) : f + ch;
                const long long stride = gathered ? c1 : c0;
#pragma unroll
                for (int j = 0; j < K; ++j) fv[j] = base[(gathered ? trow[j] : r * K + j) * stride];
            } else {
#pragma unroll
                for (int j = 0; j < K; ++j) fv[j] = f[(r * K + j) * c + ch];
            }
            load_m<M>(grad_out, static_cast<size_t>(r * c + ch) * M, g);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float a = 0.f;
#pragma unroll
                for (int m = 0; m < M; ++m) a = a + g[m] * wd[(static_cast<size_t>(k) * c + ch) * M + m];
#pragma unroll
                for (int j = 0; j < K; ++j) acc[k][j] = acc[k][j] + a * fv[j];
            }
        }
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int j = 0; j < K; ++j) {
                float v = acc[i][j];
                v = v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xb1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
                v = v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4e, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
                if ((lane & 3) == 0) mine[(lane >> 2) * STRIDE + i * K + j] = v;
            }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int p = lane; p < KK; p += 64) {
            float s = 0.f;
#pragma unroll
            for (int l = 0; l < QUADS; ++l) s = s + mine[l * STRIDE + p];
            grad_x[r * KK + p] = s;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// gradient w.r.t. the gathered feature table (GATHER mode): table row s collects, in ascending (row, slot) order (the CSR
// lists of hf_index_inverse, as hf_group_point_grad_gather sums them), the dF of every neighbour slot that read it,
//   dF[r][j][ch] = sum_k X[r][k][j] * (sum_m grad_out[r][ch*M+m] * Wd[k][ch][m]),
// rebuilt from grad_out on the fly with the multiply / add order of xconv_dw_bwd_fw_kernel: the (rows x K x c1) gradient
// of the gathered block is never written.  Same mapping as the forward: a lane owns a channel of the table, the waves of a
// block walk a chunk of table rows; list entries and the X column are wave-uniform.
template <int K, int M>
__global__ __launch_bounds__(kXcThreads) void xconv_dw_bwd_fts_kernel(long long src_rows, int n_src, long long rows_per_cloud,
                                                                     int c, int c0, int rows_per_block,
                                                                     const float *__restrict__ x, const float *__restrict__ wd,
                                                                     const float *__restrict__ grad_out,
                                                                     const int *__restrict__ offsets, const int *__restrict__ entries,
                                                                     float *__restrict__ grad_fts)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int c1 = c - c0, cf = blockIdx.y * 64 + lane, ch = c0 + cf;
    const bool live = cf < c1;
    float w[K][M];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int m = 0; m < M; ++m) w[k][m] = live ? wd[(static_cast<size_t>(k) * c + ch) * M + m] : 0.f;
    const long long s0 = static_cast<long long>(blockIdx.x) * rows_per_block;
    const long long s1 = s0 + rows_per_block < src_rows ? s0 + rows_per_block : src_rows;
    long long bb = s0 / n_src;
    for (long long s = s0 + wave; s < s1; s += kXcThreads / 64) {
        while (s >= (bb + 1) * n_src) ++bb;
        const int pt = static_cast<int>(s - bb * n_src);
        const int *off = offsets + bb * (n_src + 1);
        const int *ent = entries + bb * rows_per_cloud * K;
        const int lo = off[pt], hi = off[pt + 1];
        float acc = 0.f;
        // four list entries per trip: their loads are issued together (a dependent scalar load -> row -> vector load chain per
        // entry kept one load in flight per wave); summed in list order, slots past the end add zero
        for (int e = lo; e < hi; e += 4) {
            float a4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool valid = e + u < hi;
                const int slot = ent[valid ? e + u : lo];
                const long long r = bb * rows_per_cloud + slot / K;
                const int j = slot % K;
                const float *xr = x + r * (K * K);
                float g[M], gfx[K];
                load_m<M>(grad_out, static_cast<size_t>(r * c + (live ? ch : 0)) * M, g);
                if (!live) {
#pragma unroll
                    for (int m = 0; m < M; ++m) g[m] = 0.f;
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    float a = 0.f;
#pragma unroll
                    for (int m = 0; m < M; ++m) a = a + g[m] * w[k][m];
                    gfx[k] = a;
                }
                float a = xr[j] * gfx[0];
#pragma unroll
                for (int k = 1; k < K; ++k) a = a + xr[k * K + j] * gfx[k];
                a4[u] = valid ? a : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += a4[u];
        }
        if (live) grad_fts[s * c1 + cf] = acc;
    }
}

static int grid_for(long long items, int per_block)
{
    const long long blocks = (items + per_block - 1) / per_block;
    return static_cast<int>(blocks < 1 ? 1 : (blocks > 8 * kNumCU ? 8 * kNumCU : blocks));
}

template <int K>
static int launch_apply(long long rows, int c, const float *x, const float *f, float *out, bool transposed, hipStream_t st)
{
    const int grid = grid_for(rows, kXcThreads / 64);
    if (transposed) hipLaunchKernelGGL((xconv_apply_kernel<K, true>), dim3(grid), dim3(kXcThreads), 0, st, rows, c, x, f, out);
    else hipLaunchKernelGGL((xconv_apply_kernel<K, false>), dim3(grid), dim3(kXcThreads), 0, st, rows, c, x, f, out);
    return launch_status();
}

template <int K>
static int launch_dx(long long rows, int c, const float *grad_out, const float *f, float *grad_x, hipStream_t st)
{
    const size_t lds = sizeof(float) * (kXcThreads / 64) * 64 * (K * K + 1);
    const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(&xconv_dx_kernel<K>), lds);
    if (lrc != HF_OK) return lrc;
    hipLaunchKernelGGL((xconv_dx_kernel<K>), dim3(grid_for(rows, kXcThreads / 64)), dim3(kXcThreads), lds, st, rows, c, grad_out, f,
                       grad_x);
    return launch_status();
}

}  // namespace hf

using namespace hf;

HF_API int hf_xconv_apply(long long rows, int k, int c, const float *x, const float *f, float *out, hf_stream_t stream)
{
    if (rows < 0 || c <= 0 || !x || !f || !out) return HF_EINVAL;
    if (rows == 0) return HF_OK;
    switch (k) {
    case 4: return launch_apply<4>(rows, c, x, f, out, false, as_stream(stream));
    case 8: return launch_apply<8>(rows, c, x, f, out, false, as_stream(stream));
    default: return HF_EINVAL;   // K of the shipped configs is 8 (rpn_multiclass.config:67-112); other K: the caller's GEMM route
    }
}

HF_API int hf_xconv_apply_grad(long long rows, int k, int c, const float *x, const float *f, const float *grad_out,
                               float *grad_x, float *grad_f, hf_stream_t stream)
{
    if (rows < 0 || c <= 0 || !x || !f || !grad_out || (!grad_x && !grad_f)) return HF_EINVAL;
    if (rows == 0) return HF_OK;
    hipStream_t st = as_stream(stream);
    int rc = HF_OK;
#define HF_XC_CASE(KK)                                                                                                 \
    case KK:                                                                                                           \
        if (grad_f) rc = launch_apply<KK>(rows, c, x, grad_out, grad_f, true, st);                                     \
        if (rc == HF_OK && grad_x) rc = launch_dx<KK>(rows, c, grad_out, f, grad_x, st);                               \
        return rc;
    switch (k) {
        HF_XC_CASE(4)
        HF_XC_CASE(8)
    default: return HF_EINVAL;
    }
#undef HF_XC_CASE
}

// (k, m) pairs instantiated: the depthwise steps of the X-transform are (K, K) -> K*K with m = K; the separable
// convolutions use depth multipliers 1..4 (pointcnn.py:259-265)
#define HF_DW_DISPATCH(CALL)                                                                                           \
    if (k == 8 && m == 1) { CALL(8, 1) } else if (k == 8 && m == 2) { CALL(8, 2) } else if (k == 8 && m == 3) { CALL(8, 3) }      \
    else if (k == 8 && m == 4) { CALL(8, 4) } else if (k == 8 && m == 8) { CALL(8, 8) }                                 \
    else if (k == 4 && m == 1) { CALL(4, 1) } else if (k == 4 && m == 4) { CALL(4, 4) }                                 \
    else return HF_EINVAL;

HF_API int hf_depthwise_k(long long rows, int k, int c, int m, const float *x, const float *w, float *y, hf_stream_t stream)
{
    if (rows < 0 || c <= 0 || !x || !w || !y) return HF_EINVAL;
    if (rows == 0) return HF_OK;
    if (c <= kNarrowMaxC && k * m <= 64) {   // narrow layers: the thread's weights in registers
        const int ng = narrow_grid(rows, c);
#define HF_DW_NFWD(KK, MM) hipLaunchKernelGGL((depthwise_narrow_kernel<KK, MM, false>), dim3(ng), dim3(kXcThreads), 0, as_stream(stream), rows, c, x, w, y);
        HF_DW_DISPATCH(HF_DW_NFWD)
#undef HF_DW_NFWD
        return launch_status();
    }
    const int grid = grid_for(rows * c, kXcThreads);
#define HF_DW_FWD(KK, MM) hipLaunchKernelGGL((depthwise_fwd_kernel<KK, MM>), dim3(grid), dim3(kXcThreads), 0, as_stream(stream), rows, c, x, w, y);
    HF_DW_DISPATCH(HF_DW_FWD)
#undef HF_DW_FWD
    return launch_status();
}

static void dw_chunks(long long rows, int c, int &cblocks, int &rows_per_chunk, unsigned &nchunks)
{
    // enough chunks to fill the chip, few enough that the per-chunk tail stays a footnote; at least 64 rows per chunk
    cblocks = c < kXcThreads ? 1 : div_up(c, kXcThreads);
    long long chunks = (2 * kNumCU + cblocks - 1) / cblocks;
    if (chunks > (rows + 63) / 64) chunks = (rows + 63) / 64;
    if (chunks < 1) chunks = 1;
    if (chunks > rows) chunks = rows;
    if (chunks > 65535) chunks = 65535;
    rows_per_chunk = static_cast<int>((rows + chunks - 1) / chunks);
    nchunks = static_cast<unsigned>((rows + rows_per_chunk - 1) / rows_per_chunk);
}

HF_API size_t hf_depthwise_k_grad_workspace(long long rows, int k, int c, int m)
{
    if (rows <= 0 || k <= 0 || c <= 0 || m <= 0) return 0;
    int cblocks, rpc;
    unsigned nchunks;
    dw_chunks(rows, c, cblocks, rpc, nchunks);
    return sizeof(float) * static_cast<size_t>(nchunks) * k * c * m;
}

static int depthwise_k_grad_impl(long long rows, int k, int c, int m, const float *x, const float *w, const float *grad_y,
                                 float *grad_x, float *grad_w, float *partial, hipStream_t st)
{
    if (grad_w && (!partial || rows == 0)) {
        const int rc = hip_status(hipMemsetAsync(grad_w, 0, sizeof(float) * static_cast<size_t>(k) * c * m, st));
        if (rc != HF_OK) return rc;
    }
    if (rows == 0) return HF_OK;
    if (grad_x && c <= kNarrowMaxC && k * m <= 64) {
        const int ng = narrow_grid(rows, c);
#define HF_DW_NDX(KK, MM) hipLaunchKernelGGL((depthwise_narrow_kernel<KK, MM, true>), dim3(ng), dim3(kXcThreads), 0, st, rows, c, grad_y, w, grad_x);
        HF_DW_DISPATCH(HF_DW_NDX)
#undef HF_DW_NDX
        const int rc = launch_status();
        if (rc != HF_OK) return rc;
    } else if (grad_x) {
        const int grid = grid_for(rows * c, kXcThreads);
#define HF_DW_DX(KK, MM) hipLaunchKernelGGL((depthwise_dx_kernel<KK, MM>), dim3(grid), dim3(kXcThreads), 0, st, rows, c, grad_y, w, grad_x);
        HF_DW_DISPATCH(HF_DW_DX)
#undef HF_DW_DX
        const int rc = launch_status();
        if (rc != HF_OK) return rc;
    }
    if (grad_w) {
        int cblocks, rows_per_chunk;
        unsigned nchunks;
        dw_chunks(rows, c, cblocks, rows_per_chunk, nchunks);
        // block-level reduction when a block holds >= 2 row slots (one slot per wave on the shuffle path)
        const size_t lds = c <= kXcThreads / 2 ? sizeof(float) * static_cast<size_t>(k) * m * c * (((c & (c - 1)) == 0 && c <= 32) ? kXcThreads / 64 : 1) : 0;
        if (lds > 48 * 1024) return HF_EINVAL;
        const dim3 grid(cblocks, nchunks);
#define HF_DW_DW(KK, MM) hipLaunchKernelGGL((depthwise_dw_kernel<KK, MM>), grid, dim3(kXcThreads), lds, st, rows, c, rows_per_chunk, x, grad_y, grad_w, partial);
        HF_DW_DISPATCH(HF_DW_DW)
#undef HF_DW_DW
        if (partial) launch_partial_reduce(k * c * m, static_cast<int>(nchunks), partial, grad_w, st);
        return launch_status();
    }
    return HF_OK;
}

HF_API int hf_depthwise_k_grad(long long rows, int k, int c, int m, const float *x, const float *w, const float *grad_y,
                               float *grad_x, float *grad_w, hf_stream_t stream)
{
    if (rows < 0 || c <= 0 || !x || !w || !grad_y || (!grad_x && !grad_w)) return HF_EINVAL;
    return depthwise_k_grad_impl(rows, k, c, m, x, w, grad_y, grad_x, grad_w, nullptr, as_stream(stream));
}

HF_API int hf_depthwise_k_grad_ws(long long rows, int k, int c, int m, const float *x, const float *w, const float *grad_y,
                                  float *grad_x, float *grad_w, void *workspace, size_t workspace_bytes, hf_stream_t stream)
{
    if (rows < 0 || c <= 0 || !x || !w || !grad_y || (!grad_x && !grad_w)) return HF_EINVAL;
    if (grad_w && rows > 0 && (!workspace || workspace_bytes < hf_depthwise_k_grad_workspace(rows, k, c, m))) return HF_EWORKSPACE;
    return depthwise_k_grad_impl(rows, k, c, m, x, w, grad_y, grad_x, grad_w, grad_w ? static_cast<float *>(workspace) : nullptr,
                                 as_stream(stream));
}

// (k, m) supported by the fused X-apply + depthwise kernels: k = 8 with m = 1..4 (the separable convolutions of pointcnn.py:259-265
// under rpn_multiclass.config), and the RCNN's (4, 1), (4, 4), (12, 1), (12, 2)
#define HF_XDW_DISPATCH(CALL)                                                                                          \
    if (k == 8 && m == 1) { CALL(8, 1) } else if (k == 8 && m == 2) { CALL(8, 2) } else if (k == 8 && m == 3) { CALL(8, 3) }      \
    else if (k == 8 && m == 4) { CALL(8, 4) }                                                                          \
    else if (k == 4 && m == 1) { CALL(4, 1) } else if (k == 4 && m == 4) { CALL(4, 4) }   /* rcnn_multiclass.config:157-186: K = 4, 8, 12, 12 */ \
    else if (k == 12 && m == 1) { CALL(12, 1) } else if (k == 12 && m == 2) { CALL(12, 2) }                             \
    else return HF_EINVAL;

static void xdw_grid(long long rows, int c, dim3 &grid, int &rows_per_block, int blocks_per_cu = 8)
{
    const int cchunks = div_up(c, 64);
    long long rchunks = (blocks_per_cu * kNumCU + cchunks - 1) / cchunks;   // ~8 blocks per CU in total (forward, dX)
    if (rchunks > (rows + 3) / 4) rchunks = (rows + 3) / 4;
    if (rchunks < 1) rchunks = 1;
    rows_per_block = static_cast<int>((rows + rchunks - 1) / rchunks);
    grid = dim3(static_cast<unsigned>((rows + rows_per_block - 1) / rows_per_block), cchunks);
}

static int xdw_forward(long long rows, int k, int c, int c0, int m, const float *x, const float *f, const float *fts, const int *idx,
                       int n_src, int rows_per_cloud, const float *wd, float *out, hipStream_t st)
{
    dim3 grid;
    int rpb;
    xdw_grid(rows, c, grid, rpb);
#define HF_XDW_FWD(KK, MM)                                                                                             \
    if (idx) hipLaunchKernelGGL((xconv_dw_fwd_kernel<KK, MM, true>), grid, dim3(kXcThreads), 0, st, rows, c, c0, rpb, x, f, fts, idx, n_src, rows_per_cloud, wd, out); \
    else hipLaunchKernelGGL((xconv_dw_fwd_kernel<KK, MM, false>), grid, dim3(kXcThreads), 0, st, rows, c, c0, rpb, x, f, fts, idx, n_src, rows_per_cloud, wd, out);
    HF_XDW_DISPATCH(HF_XDW_FWD)
#undef HF_XDW_FWD
    return launch_status();
}

// grid of xconv_dw_bwd_fw_kernel: every block ends in one sum per weight coefficient (an atomic on the same k*c*m addresses, or a
// row of the partial buffer): 4 blocks per CU, and at least 32 rows per block (one frame per GPU: the deep layers have a few
// hundred rows)
static void xdw_bwd_grid(long long rows, int c, dim3 &grid, int &rpb)
{
    xdw_grid(rows, c, grid, rpb, 4);
    if (rpb < 32 && rows > 32) { rpb = 32; grid.x = static_cast<unsigned>((rows + rpb - 1) / rpb); }
}

static int xdw_backward(long long rows, int k, int c, int c0, int m, const float *x, const float *f, const float *fts, const int *idx,
                        int n_src, int rows_per_cloud, const float *wd, const float *grad_out, float *grad_x, float *grad_f, float *grad_gathered,
                        float *grad_wd, hipStream_t st, float *wd_partial = nullptr)
{
    if (grad_wd && (!wd_partial || rows == 0)) {
        const int rc = hip_status(hipMemsetAsync(grad_wd, 0, sizeof(float) * static_cast<size_t>(k) * c * m, st));
        if (rc != HF_OK) return rc;
    }
    if (rows == 0) return HF_OK;
    if (!grad_wd) wd_partial = nullptr;
    if (grad_f || grad_wd || grad_gathered) {
        dim3 grid;
        int rpb;
        xdw_bwd_grid(rows, c, grid, rpb);
#define HF_XDW_BFW(KK, MM)                                                                                             \
        if (idx) hipLaunchKernelGGL((xconv_dw_bwd_fw_kernel<KK, MM, true>), grid, dim3(kXcThreads), 0, st, rows, c, c0, rpb, x, f, fts, idx, n_src, rows_per_cloud, wd, grad_out, grad_f, grad_gathered, grad_wd, wd_partial); \
        else hipLaunchKernelGGL((xconv_dw_bwd_fw_kernel<KK, MM, false>), grid, dim3(kXcThreads), 0, st, rows, c, c0, rpb, x, f, fts, idx, n_src, rows_per_cloud, wd, grad_out, grad_f, grad_gathered, grad_wd, wd_partial);
        HF_XDW_DISPATCH(HF_XDW_BFW)
#undef HF_XDW_BFW
        if (wd_partial) launch_partial_reduce(k * c * m, static_cast<int>(grid.x), wd_partial, grad_wd, st);
        const int rc = launch_status();
        if (rc != HF_OK) return rc;
    }
    if (grad_x) {
        const size_t lds = sizeof(float) * (kXcThreads / 64) * 16 * (static_cast<size_t>(k) * k + 1);   // <= 37 KB at K = 12
#define HF_XDW_BX(KK, MM)                                                                                              \
        if (idx) hipLaunchKernelGGL((xconv_dw_bwd_x_kernel<KK, MM, true>), dim3(grid_for(rows, kXcThreads / 64)), dim3(kXcThreads), lds, st, rows, c, c0, f, fts, idx, n_src, rows_per_cloud, wd, grad_out, grad_x); \
        else hipLaunchKernelGGL((xconv_dw_bwd_x_kernel<KK, MM, false>), dim3(grid_for(rows, kXcThreads / 64)), dim3(kXcThreads), lds, st, rows, c, c0, f, fts, idx, n_src, rows_per_cloud, wd, grad_out, grad_x);
        HF_XDW_DISPATCH(HF_XDW_BX)
#undef HF_XDW_BX
        return launch_status();
    }
    return HF_OK;
}

HF_API int hf_xconv_depthwise(long long rows, int k, int c, int m, const float *x, const float *f, const float *wd, float *out,
                              hf_stream_t stream)
{
    if (rows < 0 || c <= 0 || !x || !f || !wd || !out) return HF_EINVAL;
    if (rows == 0) return HF_OK;
    return xdw_forward(rows, k, c, c, m, x, f, nullptr, nullptr, 0, 1, wd, out, as_stream(stream));
}

HF_API int hf_xconv_depthwise_grad(long long rows, int k, int c, int m, const float *x, const float *f, const float *wd,
                                   const float *grad_out, float *grad_x, float *grad_f, float *grad_wd, hf_stream_t stream)
{
    if (rows < 0 || c <= 0 || !x || !f || !wd || !grad_out || (!grad_x && !grad_f && !grad_wd)) return HF_EINVAL;
    return xdw_backward(rows, k, c, c, m, x, f, nullptr, nullptr, 0, 1, wd, grad_out, grad_x, grad_f, nullptr, grad_wd, as_stream(stream));
}

HF_API int hf_xconv_depthwise_gather(int b, int n_src, int rows_per_cloud, int k, int c0, int c1, int m, const float *x,
                                     const float *f_delta, const float *fts, const int *idx, const float *wd, float *out,
                                     hf_stream_t stream)
{
    if (b < 0 || n_src <= 0 || rows_per_cloud < 0 || c0 <= 0 || c0 % 64 != 0 || c1 <= 0) return HF_EINVAL;
    const long long rows = static_cast<long long>(b) * rows_per_cloud;
    if (rows == 0) return HF_OK;
    if (rows > 0x7fffffffll || !x || !f_delta || !fts || !idx || !wd || !out) return HF_EINVAL;
    return xdw_forward(rows, k, c0 + c1, c0, m, x, f_delta, fts, idx, n_src, rows_per_cloud, wd, out, as_stream(stream));
}

// [gradient of the gathered block: rows x k x c1][partial depthwise-weight gradients: row chunks x k x (c0+c1) x m]
static size_t xdw_gathered_bytes(int b, int rows_per_cloud, int k, int c1)
{
    return (sizeof(float) * static_cast<size_t>(b) * rows_per_cloud * k * c1 + 255) & ~static_cast<size_t>(255);
}

HF_API size_t hf_xconv_depthwise_gather_grad_workspace(int b, int rows_per_cloud, int k, int c0, int c1, int m)
{
    if (b <= 0 || rows_per_cloud <= 0 || k <= 0 || c0 <= 0 || c1 <= 0 || m <= 0) return 0;
    dim3 grid;
    int rpb;
    xdw_bwd_grid(static_cast<long long>(b) * rows_per_cloud, c0 + c1, grid, rpb);
    return xdw_gathered_bytes(b, rows_per_cloud, k, c1) + sizeof(float) * static_cast<size_t>(grid.x) * k * (c0 + c1) * m;
}

HF_API int hf_xconv_depthwise_gather_grad(int b, int n_src, int rows_per_cloud, int k, int c0, int c1, int m, const float *x,
                                          const float *f_delta, const float *fts, const int *idx, const float *wd,
                                          const float *grad_out, const int *offsets, const int *entries, float *grad_x,
                                          float *grad_f_delta, float *grad_fts, float *grad_wd, void *workspace,
                                          size_t workspace_bytes, hf_stream_t stream)
{
    if (b < 0 || n_src <= 0 || rows_per_cloud < 0 || c0 <= 0 || c0 % 64 != 0 || c1 <= 0 || !x || !f_delta || !fts || !idx || !wd ||
        !grad_out || static_cast<long long>(b) * rows_per_cloud > 0x7fffffffll ||
        (!grad_x && !grad_f_delta && !grad_fts && !grad_wd))
        return HF_EINVAL;
    if (grad_fts && (!offsets || !entries)) return HF_EINVAL;
    if (workspace && workspace_bytes < hf_xconv_depthwise_gather_grad_workspace(b, rows_per_cloud, k, c0, c1, m)) return HF_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    const long long rows = static_cast<long long>(b) * rows_per_cloud;
    const int c = c0 + c1;
    // with a workspace the gathered block's gradient is written once (rows x k x c1) and summed per table row by
    // hf_group_point_grad_gather (240 us + 230 us of extra writes at the last decoder layer); without one it is rebuilt per
    // table row from grad_out (no extra memory, 890 us there)
    float *gathered = (grad_fts && workspace) ? static_cast<float *>(workspace) : nullptr;
    float *wd_partial = workspace ? reinterpret_cast<float *>(static_cast<unsigned char *>(workspace) + xdw_gathered_bytes(b, rows_per_cloud, k, c1)) : nullptr;
    if (grad_x || grad_f_delta || grad_wd || gathered) {
        const int rc = xdw_backward(rows, k, c, c0, m, x, f_delta, fts, idx, n_src, rows_per_cloud, wd, grad_out, grad_x, grad_f_delta, gathered,
                                    grad_wd, st, wd_partial);
        if (rc != HF_OK) return rc;
    }
    if (grad_fts) {
        const long long src_rows = static_cast<long long>(b) * n_src;
        if (src_rows == 0) return HF_OK;
        if (rows == 0) return hip_status(hipMemsetAsync(grad_fts, 0, sizeof(float) * static_cast<size_t>(src_rows) * c1, st));
        if (gathered) return hf_group_point_grad_gather(b, n_src, c1, rows_per_cloud, k, c1, 0, gathered, offsets, entries, grad_fts, stream);
        dim3 grid;
        int rpb;
        xdw_grid(src_rows, c1, grid, rpb);
#define HF_XDW_FTS(KK, MM) hipLaunchKernelGGL((xconv_dw_bwd_fts_kernel<KK, MM>), grid, dim3(kXcThreads), 0, st, src_rows, n_src, static_cast<long long>(rows_per_cloud), c, c0, rpb, x, wd, grad_out, offsets, entries, grad_fts);
        HF_XDW_DISPATCH(HF_XDW_FTS)
#undef HF_XDW_FTS
        return launch_status();
    }
    return HF_OK;
}
