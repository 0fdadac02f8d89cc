// gemm.hip -- fp32 MFMA GEMMs of the shared-MLP layers for gfx950.
//
// Caller side of the hot path: tf_util.conv2d([1,1]) on R = B*M*K grouped rows
// (hf/core/feature_extractors/tf_util.py:180-203) is a tall-skinny GEMM, R ~ 1e4..1e6 rows against 4..384
// channels, and so are its two gradients.  All three are built on v_mfma_f32_32x32x2_f32: f32 in, f32
// accumulate, bit-for-bit a k-ordered fmaf chain, so the arithmetic type of the path stays fp32.
//
// Lane maps of the 32x32x2 form (lane l): A[i = l&31][k = l>>5], B[k = l>>5][j = l&31];
// D register g holds row i = 8*(g>>2) + 4*(l>>5) + (g&3), column j = l&31.
//
//   wgrad   dW (Cout, Cin) = G^T X, the reduction runs over rows.  Both operands are read in their row-major
//           global layout and staged in LDS as they are: lane (l&31) walks channels, (l>>5) picks the row of a
//           pair, so LDS reads are conflict-free and no transpose exists anywhere.  Rows are cut into chunks,
//           one workgroup per (output tile, chunk); partial tiles are summed in a fixed order by a second
//           kernel (deterministic, no atomics).  X may be given as the previous layer's pre-BN output: the
//           BN affine + ReLU is then applied while staging ("activation on load").
//   forward Z (R, Cout) = act(X) W^T + b plus the per-channel sum and sum of squares of Z (the BatchNorm batch
//           statistics) from the accumulators, so the separate statistics pass over Z disappears; with activation
//           on load the previous layer's normalised output is never written either.  A workgroup owns 128 rows
//           and all Cout columns (<= 256): wave w holds rows 32w..32w+31 as Cout/32 accumulator tiles.
#include <math.h>
#include <stdint.h>

#include "hf_common.h"

namespace hf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kGemmThreads = 256;
constexpr int kGemmRowsPerStage = 32;  // rows staged in LDS per step = 16 MFMA k-pairs

// four consecutive floats of row `row`, zero outside [0, row_end) x [0, ncols).  Branch-free: an out-of-range
// access reads a safe address and is replaced by zero afterwards.  VEC: ncols % 4 == 0 and a 16-byte aligned base.
template <bool VEC>
__device__ __forceinline__ float4 load4_guarded(const float *__restrict__ base, long long row, long long row_end,
                                                int col, int ncols)
{
    const bool row_ok = row < row_end;
    const float *p = base + (row_ok ? row : 0) * ncols;
    if constexpr (VEC) {
        const bool ok = row_ok && col < ncols;
        const float4 v = *reinterpret_cast<const float4 *>(p + (ok ? col : 0));
        return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        float4 v;
        const bool k0 = row_ok && col < ncols, k1 = row_ok && col + 1 < ncols, k2 = row_ok && col + 2 < ncols,
                   k3 = row_ok && col + 3 < ncols;
        const float x = p[k0 ? col : 0], y = p[k1 ? col + 1 : 0], z = p[k2 ? col + 2 : 0], w = p[k3 ? col + 3 : 0];
        v.x = k0 ? x : 0.f; v.y = k1 ? y : 0.f; v.z = k2 ? z : 0.f; v.w = k3 ? w : 0.f;
        return v;
    }
}

// The first layer of a set-abstraction MLP reads NEIGHBOURHOODS: row r = (cloud, query, slot) of the grouped tensor
// [points[cloud, idx[r], :] | grouped_xyz[r, :]] that sample_and_group builds (pointnet_util.py:42-64) and the first
// tf_util.conv2d reads back.  With a GatherSrc the operand is assembled from its two sources while it is staged, so that
// (B, M, K, C+3) tensor never exists.  Column layout of the assembled operand (the caller permutes the
// weight's columns to match): [features 0..cf-1, zero-padded to cfp (a multiple of 4) | x, y, z, 0]  ->  cin = cfp + 4.
struct GatherSrc {
    const float *points;        // (B, n_src, cf); unused when cf == 0
    const int *idx;             // (rows) = flattened (B, M, K) neighbour table
    const float *gxyz;          // (rows, 3) coordinates relative to the query (query_ball_group's third output)
    int cf, cfp, n_src;
    unsigned rows_per_cloud;    // M * K
    bool vec;                   // cf % 4 == 0 and `points` 16-byte aligned
};

// four consecutive operand columns col .. col+3 (col a multiple of 4) of row `row`; zero outside
__device__ __forceinline__ float4 load4_gathered(const GatherSrc &g, long long row, long long row_end, int col)
{
    const bool row_ok = row < row_end;
    const long long r = row_ok ? row : 0;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < g.cfp) {
        const unsigned cloud = static_cast<unsigned>(r) / g.rows_per_cloud;      // rows < 2^32 (checked by the launcher)
        const float *p = g.points + (static_cast<size_t>(cloud) * g.n_src + g.idx[r]) * g.cf;
        if (g.vec) {
            const bool ok = row_ok && col < g.cf;
            const float4 w = *reinterpret_cast<const float4 *>(p + (ok ? col : 0));
            if (ok) v = w;
        } else {
            const bool k0 = row_ok && col < g.cf, k1 = row_ok && col + 1 < g.cf, k2 = row_ok && col + 2 < g.cf, k3 = row_ok && col + 3 < g.cf;
            const float x = p[k0 ? col : 0], y = p[k1 ? col + 1 : 0], z = p[k2 ? col + 2 : 0], w = p[k3 ? col + 3 : 0];
            v.x = k0 ? x : 0.f; v.y = k1 ? y : 0.f; v.z = k2 ? z : 0.f; v.w = k3 ? w : 0.f;
        }
    } else if (col == g.cfp) {
        const float *q = g.gxyz + r * 3;
        const float x = q[0], y = q[1], z = q[2];
        if (row_ok) { v.x = x; v.y = y; v.z = z; }
    }
    return v;
}

// per-thread BN affine + ReLU of the four columns a thread stages: y = max(a (x - mu) + beta, 0) -- the difference first, as the
// reference graph forms it: a x + (beta - mu a) loses |mu| / sigma digits around y = 0 and flips three times as many ReLU masks
// against an fp64 evaluation (profiles/r03_grad_noise.md)
struct ColAct {
    float a[4], c[4], mu[4];
    bool on;
};

__device__ __forceinline__ ColAct make_col_act(int col, int ncols, const float *gamma, const float *beta,
                                               const float *mean, const float *invstd)
{
    ColAct f;
    f.on = gamma != nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f.a[i] = 1.f;
        f.c[i] = 0.f;
        f.mu[i] = 0.f;
        if (f.on && col + i < ncols) {
            f.a[i] = gamma[col + i] * invstd[col + i];
            f.c[i] = beta[col + i];
            f.mu[i] = mean[col + i];
        }
    }
    return f;
}

__device__ __forceinline__ float4 apply_col_act(const ColAct &f, float4 v, bool inside)
{
    if (f.on && inside) {  // rows / columns outside the matrix must stay zero
        v.x = fmaxf(f.a[0] * (v.x - f.mu[0]) + f.c[0], 0.f);
        v.y = fmaxf(f.a[1] * (v.y - f.mu[1]) + f.c[1], 0.f);
        v.z = fmaxf(f.a[2] * (v.z - f.mu[2]) + f.c[2], 0.f);
        v.w = fmaxf(f.a[3] * (v.w - f.mu[3]) + f.c[3], 0.f);
    }
    return v;
}

// ------------------------------------------------------------------------------------------
// wgrad: partial[chunk][n][k] = sum over the chunk's rows of G[r][n] * act(X)[r][k]
// WM / WN: 32x32 MFMA tiles per wave along Cout / Cin; a workgroup is 2 x 2 waves -> tile 64*WM x 64*WN.
// ------------------------------------------------------------------------------------------
template <int WM, int WN, bool VEC, bool GATHER = false>
__global__ __launch_bounds__(kGemmThreads) void wgrad_kernel(long long rows, int cout, int cin, int mtiles,
                                                             long long rows_per_chunk, const float *__restrict__ G,
                                                             const float *__restrict__ X,
                                                             const float *__restrict__ in_gamma,
                                                             const float *__restrict__ in_beta,
                                                             const float *__restrict__ in_mean,
                                                             const float *__restrict__ in_invstd,
                                                             float *__restrict__ partial, GatherSrc gs)
{
    constexpr int TM = 64 * WM, TN = 64 * WN;
    constexpr int GS = TM + 32, XS = TN + 32;  // LDS row strides: the two row-halves of a wave land on disjoint banks
    constexpr int GC4 = TM / 4, XC4 = TN / 4;  // float4 per staged row
    constexpr int GPASS = kGemmRowsPerStage * GC4 / kGemmThreads, XPASS = kGemmRowsPerStage * XC4 / kGemmThreads;
    constexpr int GROWS = kGemmThreads / GC4, XROWS = kGemmThreads / XC4;  // rows covered per pass
    __shared__ float Gs[kGemmRowsPerStage * GS];
    __shared__ float Xs[kGemmRowsPerStage * XS];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int tile_m = blockIdx.x % mtiles, tile_n = blockIdx.x / mtiles;
    const long long r0 = blockIdx.y * rows_per_chunk;
    const long long r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;

    const int gcol = tile_m * TM + (t % GC4) * 4, grow = t / GC4;
    const int xcol = tile_n * TN + (t % XC4) * 4, xrow = t / XC4;
    const ColAct act = make_col_act(xcol, cin, in_gamma, in_beta, in_mean, in_invstd);

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

    float4 gr[GPASS], xr[XPASS];
    auto fetch = [&](long long rt) {
#pragma unroll
        for (int p = 0; p < GPASS; ++p) gr[p] = load4_guarded<VEC>(G, rt + grow + p * GROWS, r1, gcol, cout);
#pragma unroll
        for (int p = 0; p < XPASS; ++p)
            xr[p] = GATHER ? load4_gathered(gs, rt + xrow + p * XROWS, r1, xcol) : load4_guarded<VEC>(X, rt + xrow + p * XROWS, r1, xcol, cin);
    };
    fetch(r0);
    for (long long rt = r0; rt < r1; rt += kGemmRowsPerStage) {
#pragma unroll
        for (int p = 0; p < GPASS; ++p)
            *reinterpret_cast<float4 *>(&Gs[(grow + p * GROWS) * GS + (t % GC4) * 4]) = gr[p];
        // the activation is applied here, not at fetch time: the loads stay in flight across the MFMA loop
#pragma unroll
        for (int p = 0; p < XPASS; ++p)
            *reinterpret_cast<float4 *>(&Xs[(xrow + p * XROWS) * XS + (t % XC4) * 4]) =
                apply_col_act(act, xr[p], rt + xrow + p * XROWS < r1);
        __syncthreads();
        if (rt + kGemmRowsPerStage < r1) fetch(rt + kGemmRowsPerStage);  // in flight during the MFMAs below
        const float *ga = Gs + (lane >> 5) * GS + wm * 32 * WM + (lane & 31);
        const float *xb = Xs + (lane >> 5) * XS + wn * 32 * WN + (lane & 31);
        float a[2][WM], b[2][WN];  // operands of the next row pair are read while this pair's MFMAs run
#pragma unroll
        for (int i = 0; i < WM; ++i) a[0][i] = ga[i * 32];
#pragma unroll
        for (int j = 0; j < WN; ++j) b[0][j] = xb[j * 32];
#pragma unroll
        for (int s = 0; s < kGemmRowsPerStage / 2; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < kGemmRowsPerStage / 2) {
#pragma unroll
                for (int i = 0; i < WM; ++i) a[nxt][i] = ga[2 * (s + 1) * GS + i * 32];
#pragma unroll
                for (int j = 0; j < WN; ++j) b[nxt][j] = xb[2 * (s + 1) * XS + j * 32];
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    float *out = partial + static_cast<size_t>(blockIdx.y) * cout * cin;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int k = tile_n * TN + wn * 32 * WN + j * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int n = tile_m * TM + wm * 32 * WM + i * 32 + 8 * (g >> 2) + 4 * (lane >> 5) + (g & 3);
                if (n < cout && k < cin) out[static_cast<size_t>(n) * cin + k] = acc[i][j][g];
            }
        }
}

// dW[e] = sum over chunks in a fixed order: 64 consecutive elements x 16 chunk groups per workgroup
// (group g sums chunks g, g+16, ... ascending; the 16 group sums are then added ascending)
constexpr int kWredGroups = 16;
__global__ __launch_bounds__(64 * kWredGroups) void wgrad_reduce_kernel(int total, int chunks,
                                                                        const float *__restrict__ partial,
                                                                        float *__restrict__ dw)
{
    __shared__ float red[kWredGroups][64];
    const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    float s = 0.f;
    if (e < total) {
        int c = grp;
        for (; c + 3 * kWredGroups < chunks; c += 4 * kWredGroups) {  // four independent loads in flight
            const float v0 = partial[static_cast<size_t>(c) * total + e];
            const float v1 = partial[static_cast<size_t>(c + kWredGroups) * total + e];
            const float v2 = partial[static_cast<size_t>(c + 2 * kWredGroups) * total + e];
            const float v3 = partial[static_cast<size_t>(c + 3 * kWredGroups) * total + e];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; c < chunks; c += kWredGroups) s += partial[static_cast<size_t>(c) * total + e];
    }
    red[grp][el] = s;
    __syncthreads();
    if (grp == 0 && e < total) {
        float r = red[0][el];
#pragma unroll
        for (int g = 1; g < kWredGroups; ++g) r += red[g][el];
        dw[e] = r;
    }
}

// ------------------------------------------------------------------------------------------
// forward: Z = act(X) W^T + bias, partial BN statistics of Z per workgroup
// ------------------------------------------------------------------------------------------
constexpr int kFwdRows = 128;   // rows per workgroup tile
constexpr int kFwdKC = 32;      // input channels per LDS stage
constexpr int kFwdLS = kFwdKC + 4;  // LDS row stride (keeps float4 stores aligned; 2-way read conflicts are noise here)
constexpr int kFwdMaxCin = 1024;

template <int NT, bool VEC, bool GATHER = false>
__global__ __launch_bounds__(kGemmThreads) __attribute__((amdgpu_waves_per_eu(2))) void linear_fwd_kernel(long long rows, int cin, int cout, long long ntiles,
                                                                  const float *__restrict__ X,
                                                                  const float *__restrict__ in_gamma,
                                                                  const float *__restrict__ in_beta,
                                                                  const float *__restrict__ in_mean,
                                                                  const float *__restrict__ in_invstd,
                                                                  const float *__restrict__ W,
                                                                  const float *__restrict__ bias, float *__restrict__ Z,
                                                                  float *__restrict__ Hout, float *__restrict__ partial, int elu,
                                                                  GatherSrc gs)
{
    // elu: the layer order of PointCNN's dense (pointfly.py:480-497), linear -> ELU -> BatchNorm: the activation on load is
    // a (elu(x) - mu) + beta (no clamp) and the statistics in the epilogue are those of elu(z)
    __shared__ float As[kFwdRows * kFwdLS];
    __shared__ float Bs[NT * 32 * kFwdLS];
    __shared__ float red[4][NT * 32][2];
    __shared__ float sc[kFwdMaxCin], sh[kFwdMaxCin], smu[kFwdMaxCin];   // scale, beta, mean

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool act = in_gamma != nullptr;
    if (act) {
        for (int k = t; k < cin; k += kGemmThreads) {
            const float a = in_gamma[k] * in_invstd[k];
            sc[k] = a;
            sh[k] = in_beta[k];
            smu[k] = in_mean[k];
        }
    }
    __syncthreads();

    const int k4 = (t & 7) * 4, srow = t >> 3;  // staging: 8 threads cover the 32 channels of a row, 32 rows per pass
    float s1[NT], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { s1[nt] = 0.f; s2[nt] = 0.f; }

    float4 ar[4], br[NT];
    auto fetch = [&](long long r0, int kc) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            ar[p] = GATHER ? load4_gathered(gs, r0 + srow + 32 * p, rows, kc + k4) : load4_guarded<VEC>(X, r0 + srow + 32 * p, rows, kc + k4, cin);
#pragma unroll
        for (int p = 0; p < NT; ++p) br[p] = load4_guarded<VEC>(W, srow + 32 * p, cout, kc + k4, cin);
    };
    if (blockIdx.x < ntiles) fetch(static_cast<long long>(blockIdx.x) * kFwdRows, 0);
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long row0 = tile * kFwdRows;
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[nt][g] = 0.f;

        for (int kc = 0; kc < cin; kc += kFwdKC) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float4 v = ar[p];
                if (act) {  // rows / channels outside the matrix stay zero
                    const bool rin = row0 + srow + 32 * p < rows;
                    const int k = kc + k4;
                    if (elu) {
                        v.x = (rin && k < cin) ? sc[k] * (elu_fwd(v.x) - smu[k]) + sh[k] : 0.f;
                        v.y = (rin && k + 1 < cin) ? sc[k + 1] * (elu_fwd(v.y) - smu[k + 1]) + sh[k + 1] : 0.f;
                        v.z = (rin && k + 2 < cin) ? sc[k + 2] * (elu_fwd(v.z) - smu[k + 2]) + sh[k + 2] : 0.f;
                        v.w = (rin && k + 3 < cin) ? sc[k + 3] * (elu_fwd(v.w) - smu[k + 3]) + sh[k + 3] : 0.f;
                    } else {
                        v.x = (rin && k < cin) ? fmaxf(sc[k] * (v.x - smu[k]) + sh[k], 0.f) : 0.f;
                        v.y = (rin && k + 1 < cin) ? fmaxf(sc[k + 1] * (v.y - smu[k + 1]) + sh[k + 1], 0.f) : 0.f;
                        v.z = (rin && k + 2 < cin) ? fmaxf(sc[k + 2] * (v.z - smu[k + 2]) + sh[k + 2], 0.f) : 0.f;
                        v.w = (rin && k + 3 < cin) ? fmaxf(sc[k + 3] * (v.w - smu[k + 3]) + sh[k + 3], 0.f) : 0.f;
                    }
                    if (Hout && rin && k < cin) {  // the activated input, kept for the weight gradient
                        float *h = Hout + (row0 + srow + 32 * p) * cin + k;
                        if constexpr (VEC) {
                            *reinterpret_cast<float4 *>(h) = v;
                        } else {
                            h[0] = v.x;
                            if (k + 1 < cin) h[1] = v.y;
                            if (k + 2 < cin) h[2] = v.z;
                            if (k + 3 < cin) h[3] = v.w;
                        }
                    }
                }
                *reinterpret_cast<float4 *>(&As[(srow + 32 * p) * kFwdLS + k4]) = v;
            }
#pragma unroll
            for (int p = 0; p < NT; ++p) *reinterpret_cast<float4 *>(&Bs[(srow + 32 * p) * kFwdLS + k4]) = br[p];
            __syncthreads();
            // the next stage -- of this tile, or the first one of the workgroup's next tile -- is in flight during
            // the MFMAs and the output stores below
            if (kc + kFwdKC < cin) fetch(row0, kc + kFwdKC);
            else if (tile + gridDim.x < ntiles) fetch((tile + gridDim.x) * kFwdRows, 0);
            const float *ap = As + (32 * wave + (lane & 31)) * kFwdLS + (lane >> 5);
            const float *bp = Bs + (lane & 31) * kFwdLS + (lane >> 5);
#pragma unroll 4
            for (int s = 0; s < kFwdKC / 2; ++s) {
                const float a = ap[2 * s];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[nt * 32 * kFwdLS + 2 * s], acc[nt], 0, 0, 0);
            }
            __syncthreads();
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = nt * 32 + (lane & 31);
            const bool cin_range = col < cout;
            const float bv = (bias && cin_range) ? bias[col] : 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const long long row = row0 + 32 * wave + 8 * (g >> 2) + 4 * (lane >> 5) + (g & 3);
                const float v = acc[nt][g] + bv;
                if (cin_range && row < rows) {
                    Z[row * cout + col] = v;
                    const float sv = elu ? elu_fwd(v) : v;
                    s1[nt] += sv;
                    s2[nt] += sv * sv;
                }
            }
        }
    }
    // per-workgroup column sums: the two row-halves of a wave, then the four waves in a fixed order
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        s1[nt] += __shfl_xor(s1[nt], 32);
        s2[nt] += __shfl_xor(s2[nt], 32);
        if (lane < 32) {
            red[wave][nt * 32 + lane][0] = s1[nt];
            red[wave][nt * 32 + lane][1] = s2[nt];
        }
    }
    __syncthreads();
    for (int col = t; col < cout; col += kGemmThreads) {
        float a = red[0][col][0], b = red[0][col][1];
#pragma unroll
        for (int w = 1; w < 4; ++w) { a += red[w][col][0]; b += red[w][col][1]; }
        partial[static_cast<size_t>(col) * kBnMaxBlocks + blockIdx.x] = a;
        partial[static_cast<size_t>(cout + col) * kBnMaxBlocks + blockIdx.x] = b;
    }
}

// ------------------------------------------------------------------------------------------
// backward (input gradient): DX (R, Cin) = DZ (R, Cout) W, the same tall-skinny shape as the forward with the roles
// (X, W) -> (DZ, W^T).  Two things ride along:
//   * DZ on load (FROM_DY): the operand is not read but rebuilt while it is staged, from the gradient DY w.r.t. this
//     layer's activation, its pre-BN output Z and its BN constants -- dz = a (dh - dbeta/R - xhat dgamma/R) with
//     dh = dy where the ReLU was active -- and written out once for the weight gradient.  The separate BN-backward
//     "dx" pass disappears.
//   * the BN-backward sums of the layer BELOW (its dbeta = sum dh', dgamma = sum dh' xhat', dh' = DX masked by that
//     layer's ReLU) are taken from the accumulators in the epilogue, reading that layer's pre-BN output once.  The
//     separate BN-backward "reduce" pass disappears.
// WT is W transposed, (Cin, Cout) row-major (a few KB, transposed by the caller).
// ------------------------------------------------------------------------------------------
constexpr int kBwdMaxK = 256;  // Cout of the layer (the K dimension here): six per-channel tables in LDS

template <int NT, bool VEC, bool FROM_DY>
__global__ __launch_bounds__(kGemmThreads) __attribute__((amdgpu_waves_per_eu(2))) void linear_bwd_kernel(
    long long rows, int kdim, int ncols, long long ntiles, const float *__restrict__ DY, const float *__restrict__ Z,
    const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ dgamma, const float *__restrict__ dbeta,
    float *__restrict__ DZ_out, const float *__restrict__ WT, float *__restrict__ DX, const float *__restrict__ Zprev,
    const float *__restrict__ p_gamma, const float *__restrict__ p_beta, const float *__restrict__ p_mean,
    const float *__restrict__ p_invstd, float *__restrict__ partial, int elu)
{
    // elu (without FROM_DY): the layer below is linear -> ELU -> BatchNorm; DX is the gradient w.r.t. its normalised output, so
    // its BN-backward sums are sum DX and sum DX * xhat with xhat from elu(z_prev), no mask
    __shared__ float As[kFwdRows * kFwdLS];
    __shared__ float Bs[NT * 32 * kFwdLS];
    __shared__ float red[4][NT * 32][2];
    __shared__ float ta[FROM_DY ? kBwdMaxK : 1], tsh[FROM_DY ? kBwdMaxK : 1], tmu[FROM_DY ? kBwdMaxK : 1],
        tis[FROM_DY ? kBwdMaxK : 1], tc1[FROM_DY ? kBwdMaxK : 1], tc2[FROM_DY ? kBwdMaxK : 1];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (FROM_DY) {
        const float inv_r = 1.0f / static_cast<float>(rows);
        for (int k = t; k < kdim; k += kGemmThreads) {
            const float a = gamma[k] * invstd[k];
            ta[k] = a;
            tsh[k] = beta[k];
            tmu[k] = mean[k];
            tis[k] = invstd[k];
            tc1[k] = dbeta[k] * inv_r;
            tc2[k] = dgamma[k] * inv_r;
        }
    }
    __syncthreads();
    const bool sums = Zprev != nullptr;
    // BN constants of the layer below for this lane's output columns
    float pa[NT], psh[NT], pmu[NT], pis[NT], s1[NT], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = nt * 32 + (lane & 31);
        pa[nt] = 0.f; psh[nt] = 0.f; pmu[nt] = 0.f; pis[nt] = 0.f; s1[nt] = 0.f; s2[nt] = 0.f;
        if (sums && col < ncols) {
            pa[nt] = p_gamma[col] * p_invstd[col];
            psh[nt] = p_beta[col];
            pmu[nt] = p_mean[col];
            pis[nt] = p_invstd[col];
        }
    }

    const int k4 = (t & 7) * 4, srow = t >> 3;
    float4 ar[4], zr[4], br[NT];
    auto fetch = [&](long long r0, int kc) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            ar[p] = load4_guarded<VEC>(DY, r0 + srow + 32 * p, rows, kc + k4, kdim);
            if (FROM_DY) zr[p] = load4_guarded<VEC>(Z, r0 + srow + 32 * p, rows, kc + k4, kdim);
        }
#pragma unroll
        for (int p = 0; p < NT; ++p) br[p] = load4_guarded<VEC>(WT, srow + 32 * p, ncols, kc + k4, kdim);
    };
    if (blockIdx.x < ntiles) fetch(static_cast<long long>(blockIdx.x) * kFwdRows, 0);
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long row0 = tile * kFwdRows;
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[nt][g] = 0.f;

        for (int kc = 0; kc < kdim; kc += kFwdKC) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float4 v = ar[p];
                if (FROM_DY) {
                    const bool rin = row0 + srow + 32 * p < rows;
                    const int k = kc + k4;
                    float d[4] = { v.x, v.y, v.z, v.w };
                    const float zz[4] = { zr[p].x, zr[p].y, zr[p].z, zr[p].w };
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (rin && k + i < kdim) {
                            const float a = ta[k + i];
                            const float dh = (a * (zz[i] - tmu[k + i]) + tsh[k + i] > 0.0f) ? d[i] : 0.0f;
                            const float xhat = (zz[i] - tmu[k + i]) * tis[k + i];
                            d[i] = a * (dh - tc1[k + i] - xhat * tc2[k + i]);
                        } else {
                            d[i] = 0.0f;
                        }
                    }
                    v = make_float4(d[0], d[1], d[2], d[3]);
                    if (DZ_out && rin && k < kdim) {
                        float *o = DZ_out + (row0 + srow + 32 * p) * kdim + k;
                        if constexpr (VEC) {
                            *reinterpret_cast<float4 *>(o) = v;
                        } else {
                            o[0] = v.x;
                            if (k + 1 < kdim) o[1] = v.y;
                            if (k + 2 < kdim) o[2] = v.z;
                            if (k + 3 < kdim) o[3] = v.w;
                        }
                    }
                }
                *reinterpret_cast<float4 *>(&As[(srow + 32 * p) * kFwdLS + k4]) = v;
            }
#pragma unroll
            for (int p = 0; p < NT; ++p) *reinterpret_cast<float4 *>(&Bs[(srow + 32 * p) * kFwdLS + k4]) = br[p];
            __syncthreads();
            if (kc + kFwdKC < kdim) fetch(row0, kc + kFwdKC);
            else if (tile + gridDim.x < ntiles) fetch((tile + gridDim.x) * kFwdRows, 0);
            const float *ap = As + (32 * wave + (lane & 31)) * kFwdLS + (lane >> 5);
            const float *bp = Bs + (lane & 31) * kFwdLS + (lane >> 5);
#pragma unroll 4
            for (int s = 0; s < kFwdKC / 2; ++s) {
                const float a = ap[2 * s];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[nt * 32 * kFwdLS + 2 * s], acc[nt], 0, 0, 0);
            }
            __syncthreads();
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = nt * 32 + (lane & 31);
            if (col >= ncols) continue;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const long long row = row0 + 32 * wave + 8 * (g >> 2) + 4 * (lane >> 5) + (g & 3);
                if (row >= rows) continue;
                const float v = acc[nt][g];
                if (DX) DX[row * ncols + col] = v;
                if (sums) {
                    const float zp = Zprev[row * ncols + col];
                    if (elu) {
                        s1[nt] += v;
                        s2[nt] += v * ((elu_fwd(zp) - pmu[nt]) * pis[nt]);
                    } else {
                        const float dh = (pa[nt] * (zp - pmu[nt]) + psh[nt] > 0.0f) ? v : 0.0f;
                        s1[nt] += dh;
                        s2[nt] += dh * ((zp - pmu[nt]) * pis[nt]);
                    }
                }
            }
        }
    }
    if (!sums) return;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        s1[nt] += __shfl_xor(s1[nt], 32);
        s2[nt] += __shfl_xor(s2[nt], 32);
        if (lane < 32) {
            red[wave][nt * 32 + lane][0] = s1[nt];
            red[wave][nt * 32 + lane][1] = s2[nt];
        }
    }
    __syncthreads();
    for (int col = t; col < ncols; col += kGemmThreads) {
        float a = red[0][col][0], b = red[0][col][1];
#pragma unroll
        for (int w = 1; w < 4; ++w) { a += red[w][col][0]; b += red[w][col][1]; }
        partial[static_cast<size_t>(col) * kBnMaxBlocks + blockIdx.x] = a;
        partial[static_cast<size_t>(ncols + col) * kBnMaxBlocks + blockIdx.x] = b;
    }
}

// ------------------------------------------------------------------------------------------
// The lifting chain of an X-Conv (pointcnn.py:96-99): two pf.dense layers on the K local coordinates of every point,
//   y0 = BN0(elu(W0 x)),  x (rows, 3);      z1 = y0 W1^T,  out = BN1(elu(z1)).
// The first layer has THREE input channels: its output (rows x C0, 268 MB at a million rows) is cheaper to recompute from x
// wherever it is needed than to write once and read four times.  Every kernel below forms z0 the same way (lift_z), so the
// forward statistics, the operand of the second GEMM, the weight gradient and both backward passes see the same bits.
// ------------------------------------------------------------------------------------------
constexpr int kLiftMaxC = 256;
struct LiftTab {
    float wx[kLiftMaxC], wy[kLiftMaxC], wz[kLiftMaxC], a[kLiftMaxC], mu[kLiftMaxC], be[kLiftMaxC], is[kLiftMaxC];
};

__device__ __forceinline__ void lift_tab_load(LiftTab &T, int c, const float *__restrict__ w0, const float *__restrict__ gamma,
                                              const float *__restrict__ beta, const float *__restrict__ mean,
                                              const float *__restrict__ invstd)
{
    for (int k = threadIdx.x; k < c; k += blockDim.x) {
        T.wx[k] = w0[3 * k];
        T.wy[k] = w0[3 * k + 1];
        T.wz[k] = w0[3 * k + 2];
        T.a[k] = gamma[k] * invstd[k];
        T.mu[k] = mean[k];
        T.be[k] = beta[k];
        T.is[k] = invstd[k];
    }
}
__device__ __forceinline__ float lift_z(float wx, float wy, float wz, float x0, float x1, float x2) { return (x0 * wx + x1 * wy) + x2 * wz; }
// the first layer's ELU is evaluated five times per element instead of once, and expm1f (~30 instructions) made the lift
// kernels VALU-bound (the forward wrote its 268 MB in 119 us); exp(x) - 1 on the hardware exponential is TensorFlow's own
// formula for tf.nn.elu (Eigen: x < 0 ? exp(x) - 1 : x) and two instructions
__device__ __forceinline__ float lift_elu(float x) { return x > 0.0f ? x : __expf(x) - 1.0f; }
__device__ __forceinline__ float lift_elu_slope(float x) { return x > 0.0f ? 1.0f : __expf(x); }
__device__ __forceinline__ float lift_y(const LiftTab &T, int k, float x0, float x1, float x2)
{
    return T.a[k] * (lift_elu(lift_z(T.wx[k], T.wy[k], T.wz[k], x0, x1, x2)) - T.mu[k]) + T.be[k];
}

// batch statistics of elu(W0 x): a lane owns a channel, the waves of a block walk a chunk of rows (x is wave-uniform)
__global__ __launch_bounds__(256) void lift_stats_kernel(long long rows, int c, long long rows_per_block,
                                                         const float *__restrict__ x3, const float *__restrict__ w0,
                                                         float *__restrict__ partial)
{
    __shared__ float red[4][64][2];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int ch = blockIdx.y * 64 + lane;
    const bool live = ch < c;
    const float wx = live ? w0[3 * ch] : 0.f, wy = live ? w0[3 * ch + 1] : 0.f, wz = live ? w0[3 * ch + 2] : 0.f;
    const long long r0 = blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float s = 0.f, q = 0.f;
    for (long long r = r0 + wave; r < r1; r += 4) {
        const float e = lift_elu(lift_z(wx, wy, wz, x3[3 * r], x3[3 * r + 1], x3[3 * r + 2]));
        s += e;
        q += e * e;
    }
    red[wave][lane][0] = s;
    red[wave][lane][1] = q;
    __syncthreads();
    if (wave == 0 && live) {
        float a = red[0][lane][0], b = red[0][lane][1];
#pragma unroll
        for (int w = 1; w < 4; ++w) { a += red[w][lane][0]; b += red[w][lane][1]; }
        partial[static_cast<size_t>(ch) * kBnMaxBlocks + blockIdx.x] = a;
        partial[static_cast<size_t>(c + ch) * kBnMaxBlocks + blockIdx.x] = b;
    }
}

// z1 = y0 W1^T with y0 generated from x while the operand is staged; statistics of elu(z1) from the accumulators
template <int NT>
__global__ __launch_bounds__(kGemmThreads) __attribute__((amdgpu_waves_per_eu(2))) void lift_linear_fwd_kernel(
    long long rows, int c0, int cout, long long ntiles, const float *__restrict__ x3, const float *__restrict__ w0,
    const float *__restrict__ gamma0, const float *__restrict__ beta0, const float *__restrict__ mean0,
    const float *__restrict__ invstd0, const float *__restrict__ W, float *__restrict__ Z, float *__restrict__ partial,
    const float *__restrict__ gamma1, const float *__restrict__ beta1, const float *__restrict__ mean1, const float *__restrict__ invstd1)
{
    __shared__ float As[kFwdRows * kFwdLS];
    __shared__ float Bs[NT * 32 * kFwdLS];
    __shared__ float red[4][NT * 32][2];
    __shared__ LiftTab T;
    // inference with the second layer's running statistics given (hf_lift_elu_fwd_eval_bn): the epilogue stores
    // gamma1 * invstd1 * (elu(z1) - mean1) + beta1 instead of z1 -- the normalisation pass of that layer (a read and a write of the
    // rows x C1 tensor: 0.32 ms for the second stage's 1.6 M x 128 lifting layer) disappears
    const bool out_bn = gamma1 != nullptr;
    float oa[NT], ob[NT], om[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = nt * 32 + (threadIdx.x & 31);
        const bool okc = out_bn && col < cout;
        oa[nt] = okc ? gamma1[col] * invstd1[col] : 0.f;
        ob[nt] = okc ? beta1[col] : 0.f;
        om[nt] = okc ? mean1[col] : 0.f;
    }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    lift_tab_load(T, c0, w0, gamma0, beta0, mean0, invstd0);
    __syncthreads();

    const int k4 = (t & 7) * 4, srow = t >> 3;
    float s1[NT], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { s1[nt] = 0.f; s2[nt] = 0.f; }
    float xr[4][3], xn[4][3];
    float4 br[NT];
    auto fetch_x = [&](long long r0, float (&dst)[4][3]) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const long long row = r0 + srow + 32 * p;
            const bool ok = row < rows;
            const float *px = x3 + (ok ? row : 0) * 3;
            const float a = px[0], b = px[1], c = px[2];
            dst[p][0] = ok ? a : 0.f; dst[p][1] = ok ? b : 0.f; dst[p][2] = ok ? c : 0.f;
        }
    };
    auto fetch_b = [&](int kc) {
#pragma unroll
        for (int p = 0; p < NT; ++p) br[p] = load4_guarded<true>(W, srow + 32 * p, cout, kc + k4, c0);
    };
    if (blockIdx.x < ntiles) { fetch_x(static_cast<long long>(blockIdx.x) * kFwdRows, xr); fetch_b(0); }
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long row0 = tile * kFwdRows;
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[nt][g] = 0.f;
        for (int kc = 0; kc < c0; kc += kFwdKC) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const bool rin = row0 + srow + 32 * p < rows;
                const int k = kc + k4;
                float4 v;
                v.x = (rin && k < c0) ? lift_y(T, k, xr[p][0], xr[p][1], xr[p][2]) : 0.f;
                v.y = (rin && k + 1 < c0) ? lift_y(T, k + 1, xr[p][0], xr[p][1], xr[p][2]) : 0.f;
                v.z = (rin && k + 2 < c0) ? lift_y(T, k + 2, xr[p][0], xr[p][1], xr[p][2]) : 0.f;
                v.w = (rin && k + 3 < c0) ? lift_y(T, k + 3, xr[p][0], xr[p][1], xr[p][2]) : 0.f;
                *reinterpret_cast<float4 *>(&As[(srow + 32 * p) * kFwdLS + k4]) = v;
            }
#pragma unroll
            for (int p = 0; p < NT; ++p) *reinterpret_cast<float4 *>(&Bs[(srow + 32 * p) * kFwdLS + k4]) = br[p];
            __syncthreads();
            if (kc + kFwdKC < c0) fetch_b(kc + kFwdKC);
            else if (tile + gridDim.x < ntiles) { fetch_x((tile + gridDim.x) * kFwdRows, xn); fetch_b(0); }
            const float *ap = As + (32 * wave + (lane & 31)) * kFwdLS + (lane >> 5);
            const float *bp = Bs + (lane & 31) * kFwdLS + (lane >> 5);
#pragma unroll 4
            for (int s = 0; s < kFwdKC / 2; ++s) {
                const float a = ap[2 * s];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[nt * 32 * kFwdLS + 2 * s], acc[nt], 0, 0, 0);
            }
            __syncthreads();
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int d = 0; d < 3; ++d) xr[p][d] = xn[p][d];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = nt * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const long long row = row0 + 32 * wave + 8 * (g >> 2) + 4 * (lane >> 5) + (g & 3);
                const float v = acc[nt][g];
                if (col < cout && row < rows) {
                    const float sv = lift_elu(v);   // 32 expm1f per lane and tile cost as much as the tile's MFMAs; |difference| <= 6e-8
                    Z[row * cout + col] = out_bn ? oa[nt] * (sv - om[nt]) + ob[nt] : v;   // the apply pass's own expression
                    s1[nt] += sv;
                    s2[nt] += sv * sv;
                }
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        s1[nt] += __shfl_xor(s1[nt], 32);
        s2[nt] += __shfl_xor(s2[nt], 32);
        if (lane < 32) {
            red[wave][nt * 32 + lane][0] = s1[nt];
            red[wave][nt * 32 + lane][1] = s2[nt];
        }
    }
    __syncthreads();
    for (int col = t; col < cout; col += kGemmThreads) {
        float a = red[0][col][0], b = red[0][col][1];
#pragma unroll
        for (int w = 1; w < 4; ++w) { a += red[w][col][0]; b += red[w][col][1]; }
        partial[static_cast<size_t>(col) * kBnMaxBlocks + blockIdx.x] = a;
        partial[static_cast<size_t>(cout + col) * kBnMaxBlocks + blockIdx.x] = b;
    }
}

// dW1 partial tiles = dz1^T y0 with y0 generated from x while it is staged (the wgrad_kernel above with its X operand rebuilt)
template <int WM, int WN>
__global__ __launch_bounds__(kGemmThreads) void lift_wgrad_kernel(long long rows, int cout, int c0, int mtiles,
                                                                  long long rows_per_chunk, const float *__restrict__ G,
                                                                  const float *__restrict__ x3, const float *__restrict__ w0,
                                                                  const float *__restrict__ gamma0, const float *__restrict__ beta0,
                                                                  const float *__restrict__ mean0, const float *__restrict__ invstd0,
                                                                  float *__restrict__ partial)
{
    constexpr int TM = 64 * WM, TN = 64 * WN;
    constexpr int GS = TM + 32, XS = TN + 32;
    constexpr int GC4 = TM / 4, XC4 = TN / 4;
    constexpr int GPASS = kGemmRowsPerStage * GC4 / kGemmThreads, XPASS = kGemmRowsPerStage * XC4 / kGemmThreads;
    constexpr int GROWS = kGemmThreads / GC4, XROWS = kGemmThreads / XC4;
    __shared__ float Gs[kGemmRowsPerStage * GS];
    __shared__ float Xs[kGemmRowsPerStage * XS];
    __shared__ LiftTab T;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int tile_m = blockIdx.x % mtiles, tile_n = blockIdx.x / mtiles;
    const long long r0 = blockIdx.y * rows_per_chunk;
    const long long r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
    lift_tab_load(T, c0, w0, gamma0, beta0, mean0, invstd0);
    __syncthreads();

    const int gcol = tile_m * TM + (t % GC4) * 4, grow = t / GC4;
    const int xcol = tile_n * TN + (t % XC4) * 4, xrow = t / XC4;
    const bool vec = cout % 4 == 0 && reinterpret_cast<uintptr_t>(G) % 16 == 0;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

    float4 gr[GPASS];
    float xr[XPASS][3];
    auto fetch = [&](long long rt) {
#pragma unroll
        for (int p = 0; p < GPASS; ++p)
            gr[p] = vec ? load4_guarded<true>(G, rt + grow + p * GROWS, r1, gcol, cout) : load4_guarded<false>(G, rt + grow + p * GROWS, r1, gcol, cout);
#pragma unroll
        for (int p = 0; p < XPASS; ++p) {
            const long long row = rt + xrow + p * XROWS;
            const bool ok = row < r1;
            const float *px = x3 + (ok ? row : 0) * 3;
            const float a = px[0], b = px[1], c = px[2];
            xr[p][0] = ok ? a : 0.f; xr[p][1] = ok ? b : 0.f; xr[p][2] = ok ? c : 0.f;
        }
    };
    fetch(r0);
    for (long long rt = r0; rt < r1; rt += kGemmRowsPerStage) {
#pragma unroll
        for (int p = 0; p < GPASS; ++p)
            *reinterpret_cast<float4 *>(&Gs[(grow + p * GROWS) * GS + (t % GC4) * 4]) = gr[p];
#pragma unroll
        for (int p = 0; p < XPASS; ++p) {
            const bool rin = rt + xrow + p * XROWS < r1;
            float4 v;
            v.x = (rin && xcol < c0) ? lift_y(T, xcol, xr[p][0], xr[p][1], xr[p][2]) : 0.f;
            v.y = (rin && xcol + 1 < c0) ? lift_y(T, xcol + 1, xr[p][0], xr[p][1], xr[p][2]) : 0.f;
            v.z = (rin && xcol + 2 < c0) ? lift_y(T, xcol + 2, xr[p][0], xr[p][1], xr[p][2]) : 0.f;
            v.w = (rin && xcol + 3 < c0) ? lift_y(T, xcol + 3, xr[p][0], xr[p][1], xr[p][2]) : 0.f;
            *reinterpret_cast<float4 *>(&Xs[(xrow + p * XROWS) * XS + (t % XC4) * 4]) = v;
        }
        __syncthreads();
        if (rt + kGemmRowsPerStage < r1) fetch(rt + kGemmRowsPerStage);
        const float *ga = Gs + (lane >> 5) * GS + wm * 32 * WM + (lane & 31);
        const float *xb = Xs + (lane >> 5) * XS + wn * 32 * WN + (lane & 31);
        float a[2][WM], b[2][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) a[0][i] = ga[i * 32];
#pragma unroll
        for (int j = 0; j < WN; ++j) b[0][j] = xb[j * 32];
#pragma unroll
        for (int s = 0; s < kGemmRowsPerStage / 2; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < kGemmRowsPerStage / 2) {
#pragma unroll
                for (int i = 0; i < WM; ++i) a[nxt][i] = ga[2 * (s + 1) * GS + i * 32];
#pragma unroll
                for (int j = 0; j < WN; ++j) b[nxt][j] = xb[2 * (s + 1) * XS + j * 32];
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    float *out = partial + static_cast<size_t>(blockIdx.y) * cout * c0;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int k = tile_n * TN + wn * 32 * WN + j * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int n = tile_m * TM + wm * 32 * WM + i * 32 + 8 * (g >> 2) + 4 * (lane >> 5) + (g & 3);
                if (n < cout && k < c0) out[static_cast<size_t>(n) * c0 + k] = acc[i][j][g];
            }
        }
}

// dy0 = dz1 W1 is formed in the accumulators and never written.  PASS 0: the BatchNorm-backward sums of the first layer
// (sum dy0, sum dy0 * xhat0) -> partial[2][c0][blocks].  PASS 1 (dgamma0 / dbeta0 known): dz0 = a0 (dy0 - dbeta0/R - xhat0 dgamma0/R)
// * elu'(z0) and the first layer's weight gradient dW0[c][d] = sum_r dz0[r][c] x[r][d] -> partial[3][c0][blocks].
// z0 and xhat0 are rebuilt from x in the epilogue (16 rows per lane).
template <int NT, int PASS>
__global__ __launch_bounds__(kGemmThreads) __attribute__((amdgpu_waves_per_eu(2))) void lift_linear_bwd_kernel(
    long long rows, int kdim, int c0, long long ntiles, const float *__restrict__ DZ, const float *__restrict__ WT,
    const float *__restrict__ x3, const float *__restrict__ w0, const float *__restrict__ gamma0, const float *__restrict__ mean0,
    const float *__restrict__ invstd0, const float *__restrict__ dgamma0, const float *__restrict__ dbeta0,
    float *__restrict__ partial)
{
    constexpr int NV = PASS == 0 ? 2 : 3;
    __shared__ float As[kFwdRows * kFwdLS];
    __shared__ float Bs[NT * 32 * kFwdLS];
    __shared__ float red[4][NT * 32][NV];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // constants of this lane's output columns (channels of the first layer)
    float wx[NT], wy[NT], wz[NT], pa[NT], pmu[NT], pis[NT], c1[NT], c2[NT], acc_s[NT][NV];
    const float inv_r = 1.0f / static_cast<float>(rows);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = nt * 32 + (lane & 31);
        const bool ok = col < c0;
        wx[nt] = ok ? w0[3 * col] : 0.f; wy[nt] = ok ? w0[3 * col + 1] : 0.f; wz[nt] = ok ? w0[3 * col + 2] : 0.f;
        pmu[nt] = ok ? mean0[col] : 0.f; pis[nt] = ok ? invstd0[col] : 0.f;
        pa[nt] = (PASS == 1 && ok) ? gamma0[col] * invstd0[col] : 0.f;
        c1[nt] = (PASS == 1 && ok) ? dbeta0[col] * inv_r : 0.f;
        c2[nt] = (PASS == 1 && ok) ? dgamma0[col] * inv_r : 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) acc_s[nt][v] = 0.f;
    }
    const bool vec = kdim % 4 == 0 && reinterpret_cast<uintptr_t>(DZ) % 16 == 0 && reinterpret_cast<uintptr_t>(WT) % 16 == 0;
    const int k4 = (t & 7) * 4, srow = t >> 3;
    float4 ar[4], br[NT];
    auto fetch = [&](long long r0, int kc) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            ar[p] = vec ? load4_guarded<true>(DZ, r0 + srow + 32 * p, rows, kc + k4, kdim) : load4_guarded<false>(DZ, r0 + srow + 32 * p, rows, kc + k4, kdim);
#pragma unroll
        for (int p = 0; p < NT; ++p)
            br[p] = vec ? load4_guarded<true>(WT, srow + 32 * p, c0, kc + k4, kdim) : load4_guarded<false>(WT, srow + 32 * p, c0, kc + k4, kdim);
    };
    if (blockIdx.x < ntiles) fetch(static_cast<long long>(blockIdx.x) * kFwdRows, 0);
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long row0 = tile * kFwdRows;
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[nt][g] = 0.f;
        for (int kc = 0; kc < kdim; kc += kFwdKC) {
#pragma unroll
            for (int p = 0; p < 4; ++p) *reinterpret_cast<float4 *>(&As[(srow + 32 * p) * kFwdLS + k4]) = ar[p];
#pragma unroll
            for (int p = 0; p < NT; ++p) *reinterpret_cast<float4 *>(&Bs[(srow + 32 * p) * kFwdLS + k4]) = br[p];
            __syncthreads();
            if (kc + kFwdKC < kdim) fetch(row0, kc + kFwdKC);
            else if (tile + gridDim.x < ntiles) fetch((tile + gridDim.x) * kFwdRows, 0);
            const float *ap = As + (32 * wave + (lane & 31)) * kFwdLS + (lane >> 5);
            const float *bp = Bs + (lane & 31) * kFwdLS + (lane >> 5);
#pragma unroll 4
            for (int s = 0; s < kFwdKC / 2; ++s) {
                const float a = ap[2 * s];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[nt * 32 * kFwdLS + 2 * s], acc[nt], 0, 0, 0);
            }
            __syncthreads();
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const long long row = row0 + 32 * wave + 8 * (g >> 2) + 4 * (lane >> 5) + (g & 3);
            if (row >= rows) continue;
            const float x0 = x3[3 * row], x1 = x3[3 * row + 1], x2 = x3[3 * row + 2];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (nt * 32 + (lane & 31) >= c0) continue;
                const float v = acc[nt][g];
                const float z0 = lift_z(wx[nt], wy[nt], wz[nt], x0, x1, x2);
                const float xhat = (lift_elu(z0) - pmu[nt]) * pis[nt];
                if (PASS == 0) {
                    acc_s[nt][0] += v;
                    acc_s[nt][1] += v * xhat;
                } else {
                    const float dz0 = pa[nt] * (v - c1[nt] - xhat * c2[nt]) * lift_elu_slope(z0);
                    acc_s[nt][0] += dz0 * x0;
                    acc_s[nt][1] += dz0 * x1;
                    acc_s[nt][NV - 1] += dz0 * x2;
                }
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            acc_s[nt][v] += __shfl_xor(acc_s[nt][v], 32);
            if (lane < 32) red[wave][nt * 32 + lane][v] = acc_s[nt][v];
        }
    __syncthreads();
    for (int col = t; col < c0; col += kGemmThreads) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float a = red[0][col][v];
#pragma unroll
            for (int w = 1; w < 4; ++w) a += red[w][col][v];
            partial[(static_cast<size_t>(v) * c0 + col) * kBnMaxBlocks + blockIdx.x] = a;
        }
    }
}

// out[row] = sum over blocks of partial[row][blk] (fp64 tree, a workgroup per row)
__global__ __launch_bounds__(256) void partial_rows_sum_kernel(int nblk, const float *__restrict__ partial, float *__restrict__ out)
{
    __shared__ double red[256];
    const int t = threadIdx.x;
    double a = 0.0;
    const float *p = partial + static_cast<size_t>(blockIdx.x) * kBnMaxBlocks;
    for (int i = t; i < nblk; i += 256) a += static_cast<double>(p[i]);
    red[t] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) red[t] += red[t + w];
        __syncthreads();
    }
    if (t == 0) out[blockIdx.x] = static_cast<float>(red[0]);
}

struct WgradPlan {
    int wm, wn, mtiles, ntiles, chunks;
    long long rows_per_chunk;
};

static WgradPlan wgrad_plan(long long rows, int cout, int cin)
{
    WgradPlan p;
    p.wm = cout > 64 ? 2 : 1;
    p.wn = cin > 64 ? 2 : 1;
    p.mtiles = div_up(cout, 64 * p.wm);
    p.ntiles = div_up(cin, 64 * p.wn);
    // ~3 workgroups per CU in total; a chunk is at least 256 rows and a multiple of the 32-row stage
    long long want = static_cast<long long>(kNumCU) * 3 / (p.mtiles * p.ntiles);
    if (want < 1) want = 1;
    long long rpc = (rows + want - 1) / want;
    if (rpc < 256) rpc = 256;
    rpc = (rpc + kGemmRowsPerStage - 1) / kGemmRowsPerStage * kGemmRowsPerStage;
    p.rows_per_chunk = rpc;
    p.chunks = static_cast<int>((rows + rpc - 1) / rpc);
    return p;
}

// Persistent tile loops: as many workgroups as are resident at once (LDS / VGPR bound per accumulator-tile count),
// times HF_GEMM_ROUNDS (default 2: whole rounds, and a tail that costs half as much when sampling pins a few CUs).
static int resident_grid(int nt, long long ntiles)
{
    static const int per_cu[9] = { 0, 5, 4, 3, 3, 2, 2, 2, 2 };
    static const int rounds = HF_DIAG_INT("HF_GEMM_ROUNDS", 2) < 1 ? 1 : HF_DIAG_INT("HF_GEMM_ROUNDS", 2);
    long long g = static_cast<long long>(kNumCU) * per_cu[nt < 1 ? 1 : (nt > 8 ? 8 : nt)] * rounds;
    if (g > kBnMaxBlocks) g = kBnMaxBlocks;
    if (g > ntiles) g = ntiles;
    return static_cast<int>(g);
}

void launch_partial_reduce(int total, int chunks, const float *partial, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(div_up(total, 64)), dim3(64 * kWredGroups), 0, st, total, chunks, partial, out);
}

static bool vec4_ok(const void *p, int ncols) { return ncols % 4 == 0 && reinterpret_cast<uintptr_t>(p) % 16 == 0; }

}  // namespace hf

using namespace hf;

HF_API size_t hf_linear_wgrad_workspace(long long rows, int cout, int cin)
{
    if (rows <= 0 || cout <= 0 || cin <= 0) return 0;
    const WgradPlan p = wgrad_plan(rows, cout, cin);
    return sizeof(float) * static_cast<size_t>(p.chunks) * cout * cin;
}

static GatherSrc no_gather() { return GatherSrc{ nullptr, nullptr, nullptr, 0, 0, 0, 1u, false }; }

// the A operand of a gathering launch; HF_EINVAL when the arguments do not describe one
static int make_gather(long long rows, int c_feat, const float *points, int n_src, long long rows_per_cloud, const int *idx,
                       const float *gxyz, GatherSrc *g)
{
    if (rows <= 0 || rows > 0x7fffffffll || c_feat < 0 || c_feat > kFwdMaxCin - 4 || n_src <= 0 || rows_per_cloud <= 0 ||
        rows_per_cloud > rows || rows % rows_per_cloud != 0 || !idx || !gxyz || (c_feat > 0 && !points))
        return HF_EINVAL;
    g->points = points; g->idx = idx; g->gxyz = gxyz;
    g->cf = c_feat; g->cfp = (c_feat + 3) & ~3; g->n_src = n_src;
    g->rows_per_cloud = static_cast<unsigned>(rows_per_cloud);
    g->vec = c_feat > 0 && c_feat % 4 == 0 && reinterpret_cast<uintptr_t>(points) % 16 == 0;
    return HF_OK;
}

static int linear_wgrad_impl(long long rows, int cout, int cin, const float *grad_z, const float *x, const float *in_gamma,
                             const float *in_beta, const float *in_mean, const float *in_invstd, float *grad_weight,
                             void *workspace, size_t workspace_bytes, hf_stream_t stream, const GatherSrc *gather)
{
    if (rows <= 0 || cout <= 0 || cin <= 0 || cout > 4096 || cin > 4096 || !grad_z || (!x && !gather) || !grad_weight) return HF_EINVAL;
    if (in_gamma && (!in_beta || !in_mean || !in_invstd)) return HF_EINVAL;
    if (!workspace || workspace_bytes < hf_linear_wgrad_workspace(rows, cout, cin)) return HF_EWORKSPACE;
    const WgradPlan p = wgrad_plan(rows, cout, cin);
    if (p.chunks > 65535) return HF_EINVAL;
    hipStream_t st = as_stream(stream);
    float *partial = static_cast<float *>(workspace);
    const dim3 grid(p.mtiles * p.ntiles, p.chunks);
    const bool vec = vec4_ok(grad_z, cout) && (gather || vec4_ok(x, cin));
    const GatherSrc gs = gather ? *gather : no_gather();
#define HF_WGRAD(M, N, V, GA)                                                                                           \
    hipLaunchKernelGGL((wgrad_kernel<M, N, V, GA>), grid, dim3(kGemmThreads), 0, st, rows, cout, cin, p.mtiles,          \
                       p.rows_per_chunk, grad_z, x, in_gamma, in_beta, in_mean, in_invstd, partial, gs)
#define HF_WGRAD_V(M, N)                                                                                                \
    do {                                                                                                                \
        if (gather) { if (vec) HF_WGRAD(M, N, true, true); else HF_WGRAD(M, N, false, true); }                          \
        else if (vec) HF_WGRAD(M, N, true, false); else HF_WGRAD(M, N, false, false);                                   \
    } while (0)
    if (p.wm == 2 && p.wn == 2) HF_WGRAD_V(2, 2);
    else if (p.wm == 2) HF_WGRAD_V(2, 1);
    else if (p.wn == 2) HF_WGRAD_V(1, 2);
    else HF_WGRAD_V(1, 1);
#undef HF_WGRAD_V
#undef HF_WGRAD
    const int total = cout * cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(div_up(total, 64)), dim3(64 * kWredGroups), 0, st, total, p.chunks, partial,
                       grad_weight);
    return launch_status();
}

HF_API int hf_linear_wgrad(long long rows, int cout, int cin, const float *grad_z, const float *x, const float *in_gamma,
                           const float *in_beta, const float *in_mean, const float *in_invstd, float *grad_weight,
                           void *workspace, size_t workspace_bytes, hf_stream_t stream)
{
    return linear_wgrad_impl(rows, cout, cin, grad_z, x, in_gamma, in_beta, in_mean, in_invstd, grad_weight, workspace, workspace_bytes,
                             stream, nullptr);
}

HF_API int hf_linear_wgrad_gather(long long rows, int cout, int c_feat, const float *grad_z, const float *points, int n_src,
                                  long long rows_per_cloud, const int *idx, const float *grouped_xyz, float *grad_weight,
                                  void *workspace, size_t workspace_bytes, hf_stream_t stream)
{
    GatherSrc g;
    if (const int rc = make_gather(rows, c_feat, points, n_src, rows_per_cloud, idx, grouped_xyz, &g); rc != HF_OK) return rc;
    return linear_wgrad_impl(rows, cout, g.cfp + 4, grad_z, nullptr, nullptr, nullptr, nullptr, nullptr, grad_weight, workspace,
                             workspace_bytes, stream, &g);
}

HF_API size_t hf_linear_bn_fwd_workspace(int cout)
{
    return cout > 0 ? sizeof(float) * 2 * static_cast<size_t>(cout) * kBnMaxBlocks : 0;
}

static int linear_bn_fwd_impl(long long rows, int cin, int cout, const float *x, const float *in_gamma, const float *in_beta,
                              const float *in_mean, const float *in_invstd, float *x_act, const float *weight,
                              const float *bias, float *z, float eps, float momentum, float *running_mean, float *running_var, float *mean, float *invstd,
                              void *workspace, size_t workspace_bytes, hf_stream_t stream, int elu, const GatherSrc *gather = nullptr)
{
    if (rows <= 0 || cin <= 0 || cout <= 0 || cin > kFwdMaxCin || cout > 256 || (!x && !gather) || !weight || !z || !mean || !invstd)
        return HF_EINVAL;
    if (in_gamma && (!in_beta || !in_mean || !in_invstd)) return HF_EINVAL;
    if (x_act && !in_gamma) return HF_EINVAL;
    if (!workspace || workspace_bytes < hf_linear_bn_fwd_workspace(cout)) return HF_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    float *partial = static_cast<float *>(workspace);
    const long long ntiles = (rows + kFwdRows - 1) / kFwdRows;
    const bool vec = (gather || vec4_ok(x, cin)) && vec4_ok(weight, cin) && (!x_act || vec4_ok(x_act, cin));
    const int nt = div_up(cout, 32);
    const int nblk = resident_grid(nt, ntiles);
    const GatherSrc gs = gather ? *gather : no_gather();
#define HF_FWD(N, V, GA)                                                                                                \
    hipLaunchKernelGGL((linear_fwd_kernel<N, V, GA>), dim3(nblk), dim3(kGemmThreads), 0, st, rows, cin, cout, ntiles, x, \
                       in_gamma, in_beta, in_mean, in_invstd, weight, bias, z, x_act, partial, elu, gs)
#define HF_FWD_V(N)                                                                                                     \
    case N:                                                                                                             \
        if (gather) { if (vec) HF_FWD(N, true, true); else HF_FWD(N, false, true); }                                    \
        else if (vec) HF_FWD(N, true, false); else HF_FWD(N, false, false);                                             \
        break
    switch (nt) {
        HF_FWD_V(1); HF_FWD_V(2); HF_FWD_V(3); HF_FWD_V(4); HF_FWD_V(5); HF_FWD_V(6); HF_FWD_V(7); HF_FWD_V(8);
        default: return HF_EINVAL;
    }
#undef HF_FWD_V
#undef HF_FWD
    launch_bn_stats_finalize(rows, cout, nblk, partial, eps, momentum, running_mean, running_var, mean, invstd, st);
    return launch_status();
}

HF_API int hf_linear_bn_fwd(long long rows, int cin, int cout, const float *x, const float *in_gamma, const float *in_beta,
                            const float *in_mean, const float *in_invstd, float *x_act, const float *weight,
                            const float *bias, float *z, float eps, float momentum, float *running_mean, float *running_var, float *mean, float *invstd,
                            void *workspace, size_t workspace_bytes, hf_stream_t stream)
{
    return linear_bn_fwd_impl(rows, cin, cout, x, in_gamma, in_beta, in_mean, in_invstd, x_act, weight, bias, z, eps, momentum, running_mean,
                              running_var, mean, invstd, workspace, workspace_bytes, stream, 0);
}

HF_API int hf_linear_bn_fwd_gather(long long rows, int c_feat, int cout, const float *points, int n_src, long long rows_per_cloud,
                                   const int *idx, const float *grouped_xyz, const float *weight, const float *bias, float *z, float eps,
                                   float momentum, float *running_mean, float *running_var, float *mean, float *invstd, void *workspace,
                                   size_t workspace_bytes, hf_stream_t stream)
{
    GatherSrc g;
    if (const int rc = make_gather(rows, c_feat, points, n_src, rows_per_cloud, idx, grouped_xyz, &g); rc != HF_OK) return rc;
    return linear_bn_fwd_impl(rows, g.cfp + 4, cout, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, weight, bias, z, eps, momentum,
                              running_mean, running_var, mean, invstd, workspace, workspace_bytes, stream, 0, &g);
}

HF_API int hf_linear_elu_bn_fwd(long long rows, int cin, int cout, const float *x, const float *in_gamma, const float *in_beta,
                                const float *in_mean, const float *in_invstd, float *x_act, const float *weight, float *z, float eps,
                                float momentum, float *running_mean, float *running_var, float *mean, float *invstd, void *workspace,
                                size_t workspace_bytes, hf_stream_t stream)
{
    return linear_bn_fwd_impl(rows, cin, cout, x, in_gamma, in_beta, in_mean, in_invstd, x_act, weight, nullptr, z, eps, momentum, running_mean,
                              running_var, mean, invstd, workspace, workspace_bytes, stream, 1);
}

HF_API size_t hf_linear_bn_bwd_workspace(int cin)
{
    return cin > 0 ? sizeof(float) * 2 * static_cast<size_t>(cin) * kBnMaxBlocks : 0;
}

static int linear_bn_bwd_impl(long long rows, int cout, int cin, const float *dy_or_dz, const float *z, const float *gamma,
                              const float *beta, const float *mean, const float *invstd, const float *dgamma,
                              const float *dbeta, float *dz_out, const float *weight_t, float *dx, const float *z_prev,
                              const float *p_gamma, const float *p_beta, const float *p_mean, const float *p_invstd,
                              float *p_dgamma, float *p_dbeta, void *workspace, size_t workspace_bytes, hf_stream_t stream, int elu)
{
    if (rows <= 0 || cout <= 0 || cin <= 0 || cin > 256 || !dy_or_dz || !weight_t) return HF_EINVAL;
    const bool from_dy = z != nullptr;
    if (from_dy && (cout > kBwdMaxK || !gamma || !beta || !mean || !invstd || !dgamma || !dbeta)) return HF_EINVAL;
    if (!from_dy && dz_out) return HF_EINVAL;
    const bool sums = z_prev != nullptr;
    if (sums && (!p_gamma || !p_beta || !p_mean || !p_invstd || !p_dgamma || !p_dbeta)) return HF_EINVAL;
    if (sums && (!workspace || workspace_bytes < hf_linear_bn_bwd_workspace(cin))) return HF_EWORKSPACE;
    if (!dx && !sums && !dz_out) return HF_EINVAL;
    hipStream_t st = as_stream(stream);
    float *partial = static_cast<float *>(workspace);
    const long long ntiles = (rows + kFwdRows - 1) / kFwdRows;
    const int nblk = resident_grid(div_up(cin, 32), ntiles);
    const bool vec = vec4_ok(dy_or_dz, cout) && vec4_ok(weight_t, cout) && (!from_dy || vec4_ok(z, cout)) &&
                     (!dz_out || vec4_ok(dz_out, cout));
    const int nt = div_up(cin, 32);
#define HF_BWD(N, V, F)                                                                                                 \
    hipLaunchKernelGGL((linear_bwd_kernel<N, V, F>), dim3(nblk), dim3(kGemmThreads), 0, st, rows, cout, cin, ntiles,     \
                       dy_or_dz, z, gamma, beta, mean, invstd, dgamma, dbeta, dz_out, weight_t, dx, z_prev, p_gamma,   \
                       p_beta, p_mean, p_invstd, partial, elu)
#define HF_BWD_N(N)                                                                                                     \
    case N:                                                                                                             \
        if (vec && from_dy) HF_BWD(N, true, true);                                                                      \
        else if (vec) HF_BWD(N, true, false);                                                                           \
        else if (from_dy) HF_BWD(N, false, true);                                                                       \
        else HF_BWD(N, false, false);                                                                                   \
        break
    switch (nt) {
        HF_BWD_N(1); HF_BWD_N(2); HF_BWD_N(3); HF_BWD_N(4); HF_BWD_N(5); HF_BWD_N(6); HF_BWD_N(7); HF_BWD_N(8);
        default: return HF_EINVAL;
    }
#undef HF_BWD_N
#undef HF_BWD
    if (sums) launch_bn_bwd_finalize(cin, nblk, partial, p_dgamma, p_dbeta, st);
    return launch_status();
}

HF_API int hf_linear_bn_bwd(long long rows, int cout, int cin, const float *dy_or_dz, const float *z, const float *gamma,
                            const float *beta, const float *mean, const float *invstd, const float *dgamma,
                            const float *dbeta, float *dz_out, const float *weight_t, float *dx, const float *z_prev,
                            const float *p_gamma, const float *p_beta, const float *p_mean, const float *p_invstd,
                            float *p_dgamma, float *p_dbeta, void *workspace, size_t workspace_bytes, hf_stream_t stream)
{
    return linear_bn_bwd_impl(rows, cout, cin, dy_or_dz, z, gamma, beta, mean, invstd, dgamma, dbeta, dz_out, weight_t, dx, z_prev, p_gamma,
                              p_beta, p_mean, p_invstd, p_dgamma, p_dbeta, workspace, workspace_bytes, stream, 0);
}

HF_API int hf_linear_elu_bn_bwd(long long rows, int cout, int cin, const float *dz, const float *weight_t, float *dx, const float *z_prev,
                                const float *p_gamma, const float *p_beta, const float *p_mean, const float *p_invstd, float *p_dgamma,
                                float *p_dbeta, void *workspace, size_t workspace_bytes, hf_stream_t stream)
{
    if (!z_prev) return HF_EINVAL;
    return linear_bn_bwd_impl(rows, cout, cin, dz, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, weight_t, dx, z_prev,
                              p_gamma, p_beta, p_mean, p_invstd, p_dgamma, p_dbeta, workspace, workspace_bytes, stream, 1);
}

static const float *const kNoF = nullptr;

HF_API size_t hf_lift_elu_bn_fwd_workspace(int c0, int c1)
{
    if (c0 <= 0 || c1 <= 0) return 0;
    return sizeof(float) * 2 * static_cast<size_t>(c0 > c1 ? c0 : c1) * kBnMaxBlocks;
}

HF_API int hf_lift_elu_bn_fwd(long long rows, int c0, int c1, const float *x3, const float *w0, const float *gamma0, const float *beta0,
                              float eps0, float momentum0, float *running_mean0, float *running_var0, float *mean0, float *invstd0,
                              const float *w1, float *z1, float eps1, float momentum1, float *running_mean1, float *running_var1,
                              float *mean1, float *invstd1, void *workspace, size_t workspace_bytes, hf_stream_t stream)
{
    if (rows <= 0 || c0 <= 0 || c0 > kLiftMaxC || c0 % 4 != 0 || c1 <= 0 || c1 > 256 || !x3 || !w0 || !gamma0 || !beta0 || !mean0 ||
        !invstd0 || !w1 || !z1 || !mean1 || !invstd1 || reinterpret_cast<uintptr_t>(w1) % 16 != 0)
        return HF_EINVAL;
    if (!workspace || workspace_bytes < hf_lift_elu_bn_fwd_workspace(c0, c1)) return HF_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    float *partial = static_cast<float *>(workspace);
    long long nblk0 = (rows + 255) / 256;
    if (nblk0 > kBnMaxBlocks) nblk0 = kBnMaxBlocks;
    const long long rpb = (rows + nblk0 - 1) / nblk0;
    nblk0 = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(lift_stats_kernel, dim3(static_cast<unsigned>(nblk0), div_up(c0, 64)), dim3(256), 0, st, rows, c0, rpb, x3, w0,
                       partial);
    launch_bn_stats_finalize(rows, c0, static_cast<int>(nblk0), partial, eps0, momentum0, running_mean0, running_var0, mean0, invstd0, st);
    const long long ntiles = (rows + kFwdRows - 1) / kFwdRows;
    const int nt = div_up(c1, 32);
    const int nblk = resident_grid(nt, ntiles);
#define HF_LIFT_FWD(N)                                                                                                  \
    case N:                                                                                                             \
        hipLaunchKernelGGL((lift_linear_fwd_kernel<N>), dim3(nblk), dim3(kGemmThreads), 0, st, rows, c0, c1, ntiles, x3, w0,  \
                           gamma0, beta0, mean0, invstd0, w1, z1, partial, kNoF, kNoF, kNoF, kNoF);                     \
        break
    switch (nt) {
        HF_LIFT_FWD(1); HF_LIFT_FWD(2); HF_LIFT_FWD(3); HF_LIFT_FWD(4); HF_LIFT_FWD(5); HF_LIFT_FWD(6); HF_LIFT_FWD(7); HF_LIFT_FWD(8);
        default: return HF_EINVAL;
    }
#undef HF_LIFT_FWD
    launch_bn_stats_finalize(rows, c1, nblk, partial, eps1, momentum1, running_mean1, running_var1, mean1, invstd1, st);
    return launch_status();
}

// inference: the first layer normalised with GIVEN statistics (its running estimates), no statistics pass; the second GEMM's batch
// statistics land in the workspace and are ignored
static int lift_elu_fwd_eval_impl(long long rows, int c0, int c1, const float *x3, const float *w0, const float *gamma0, const float *beta0,
                                  const float *mean0, const float *invstd0, const float *w1, const float *gamma1, const float *beta1,
                                  const float *mean1, const float *invstd1, float *z1, void *workspace, size_t workspace_bytes,
                                  hf_stream_t stream)
{
    if (rows <= 0 || c0 <= 0 || c0 > kLiftMaxC || c0 % 4 != 0 || c1 <= 0 || c1 > 256 || !x3 || !w0 || !gamma0 || !beta0 || !mean0 ||
        !invstd0 || !w1 || !z1 || reinterpret_cast<uintptr_t>(w1) % 16 != 0)
        return HF_EINVAL;
    if (!workspace || workspace_bytes < hf_lift_elu_bn_fwd_workspace(c0, c1)) return HF_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    float *partial = static_cast<float *>(workspace);
    const long long ntiles = (rows + kFwdRows - 1) / kFwdRows;
    const int nt = div_up(c1, 32);
    const int nblk = resident_grid(nt, ntiles);
#define HF_LIFT_FWD(N)                                                                                                  \
    case N:                                                                                                             \
        hipLaunchKernelGGL((lift_linear_fwd_kernel<N>), dim3(nblk), dim3(kGemmThreads), 0, st, rows, c0, c1, ntiles, x3, w0,  \
                           gamma0, beta0, mean0, invstd0, w1, z1, partial, gamma1, beta1, mean1, invstd1);              \
        break
    switch (nt) {
        HF_LIFT_FWD(1); HF_LIFT_FWD(2); HF_LIFT_FWD(3); HF_LIFT_FWD(4); HF_LIFT_FWD(5); HF_LIFT_FWD(6); HF_LIFT_FWD(7); HF_LIFT_FWD(8);
        default: return HF_EINVAL;
    }
#undef HF_LIFT_FWD
    return launch_status();
}

HF_API int hf_lift_elu_fwd_eval(long long rows, int c0, int c1, const float *x3, const float *w0, const float *gamma0, const float *beta0,
                                const float *mean0, const float *invstd0, const float *w1, float *z1, void *workspace,
                                size_t workspace_bytes, hf_stream_t stream)
{
    return lift_elu_fwd_eval_impl(rows, c0, c1, x3, w0, gamma0, beta0, mean0, invstd0, w1, nullptr, nullptr, nullptr, nullptr, z1, workspace,
                                  workspace_bytes, stream);
}

HF_API int hf_lift_elu_fwd_eval_bn(long long rows, int c0, int c1, const float *x3, const float *w0, const float *gamma0, const float *beta0,
                                   const float *mean0, const float *invstd0, const float *w1, const float *gamma1, const float *beta1,
                                   const float *mean1, const float *invstd1, float *y1, void *workspace, size_t workspace_bytes,
                                   hf_stream_t stream)
{
    if (!gamma1 || !beta1 || !mean1 || !invstd1) return HF_EINVAL;
    return lift_elu_fwd_eval_impl(rows, c0, c1, x3, w0, gamma0, beta0, mean0, invstd0, w1, gamma1, beta1, mean1, invstd1, y1, workspace,
                                  workspace_bytes, stream);
}

HF_API size_t hf_lift_elu_bn_bwd_workspace(long long rows, int c0, int c1)
{
    if (rows <= 0 || c0 <= 0 || c1 <= 0) return 0;
    return hf_linear_wgrad_workspace(rows, c1, c0) + sizeof(float) * 3 * static_cast<size_t>(c0) * kBnMaxBlocks;
}

HF_API int hf_lift_elu_bn_bwd(long long rows, int c0, int c1, const float *x3, const float *w0, const float *gamma0, const float *beta0,
                              const float *mean0, const float *invstd0, const float *dz1, const float *w1_t, float *grad_w0_t,
                              float *grad_w1, float *dgamma0, float *dbeta0, void *workspace, size_t workspace_bytes,
                              hf_stream_t stream)
{
    if (rows <= 0 || c0 <= 0 || c0 > 160 || c0 % 4 != 0 || c1 <= 0 || c1 > 256 || !x3 || !w0 || !gamma0 || !beta0 || !mean0 || !invstd0 ||
        !dz1 || !w1_t || !grad_w0_t || !grad_w1 || !dgamma0 || !dbeta0)
        return HF_EINVAL;
    if (!workspace || workspace_bytes < hf_lift_elu_bn_bwd_workspace(rows, c0, c1)) return HF_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    // dW1 = dz1^T y0
    const WgradPlan p = wgrad_plan(rows, c1, c0);
    if (p.chunks > 65535) return HF_EINVAL;
    float *wpartial = static_cast<float *>(workspace);
    float *spartial = reinterpret_cast<float *>(static_cast<unsigned char *>(workspace) + hf_linear_wgrad_workspace(rows, c1, c0));
    const dim3 wgrid(p.mtiles * p.ntiles, p.chunks);
#define HF_LIFT_WG(M, N)                                                                                                \
    hipLaunchKernelGGL((lift_wgrad_kernel<M, N>), wgrid, dim3(kGemmThreads), 0, st, rows, c1, c0, p.mtiles, p.rows_per_chunk, \
                       dz1, x3, w0, gamma0, beta0, mean0, invstd0, wpartial)
    if (p.wm == 2 && p.wn == 2) HF_LIFT_WG(2, 2);
    else if (p.wm == 2) HF_LIFT_WG(2, 1);
    else if (p.wn == 2) HF_LIFT_WG(1, 2);
    else HF_LIFT_WG(1, 1);
#undef HF_LIFT_WG
    const int total = c1 * c0;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(div_up(total, 64)), dim3(64 * kWredGroups), 0, st, total, p.chunks, wpartial, grad_w1);
    // the first layer's BatchNorm-backward sums, then its weight gradient: dy0 = dz1 W1 rebuilt in the accumulators both times
    const long long ntiles = (rows + kFwdRows - 1) / kFwdRows;
    const int nt = div_up(c0, 32);
    const int nblk = resident_grid(nt, ntiles);
#define HF_LIFT_BWD(N, P)                                                                                               \
    hipLaunchKernelGGL((lift_linear_bwd_kernel<N, P>), dim3(nblk), dim3(kGemmThreads), 0, st, rows, c1, c0, ntiles, dz1, w1_t, x3, w0, \
                       gamma0, mean0, invstd0, dgamma0, dbeta0, spartial)
#define HF_LIFT_BWD_N(P)                                                                                                \
    switch (nt) {                                                                                                       \
    case 1: HF_LIFT_BWD(1, P); break;                                                                                   \
    case 2: HF_LIFT_BWD(2, P); break;                                                                                   \
    case 3: HF_LIFT_BWD(3, P); break;                                                                                   \
    case 4: HF_LIFT_BWD(4, P); break;                                                                                   \
    case 5: HF_LIFT_BWD(5, P); break;                                                                                   \
    default: return HF_EINVAL;                                                                                          \
    }
    HF_LIFT_BWD_N(0)
    launch_bn_bwd_finalize(c0, nblk, spartial, dgamma0, dbeta0, st);
    HF_LIFT_BWD_N(1)
#undef HF_LIFT_BWD_N
#undef HF_LIFT_BWD
    hipLaunchKernelGGL(partial_rows_sum_kernel, dim3(3 * c0), dim3(256), 0, st, nblk, spartial, grad_w0_t);
    return launch_status();
}
