// mlp.hip -- training-mode BatchNorm + ReLU on channel-last rows (R, C) for gfx950.
//
// Caller side of the hot path: the grouped-point MLP of a set-abstraction
// level is tf_util.conv2d([1,1]) = matmul + bias, tf.contrib.layers.batch_norm(decay .9, eps 1e-3,
// not fused) and ReLU (hf/core/feature_extractors/tf_util.py:180-203,554-581) applied to
// R = B*M*K rows of 32..256 channels.  At R = 10^6 the three BN passes and the separate ReLU are
// pure HBM traffic; the framework kernels they map to run at 0.8-1.4 TB/s on MI355X (profiles/).
// Here:
//   forward   stats pass (read x once; per-block fp32 partial sums, reduced in fp64 -> deterministic,
//             no atomics) + one apply pass  y = relu(a*x + b)            (read x, write y)
//   backward  reduce pass (read dy, x: sum dh, sum dh*xhat, the ReLU mask recomputed from x)
//             + one dx pass                                              (read dy, x, write dx)
// Thread geometry: a thread owns VEC consecutive channels for its whole life (scale/shift live in
// registers) and strides over rows; 16-byte accesses when C % 4 == 0.
#include <math.h>

#include "hf_common.h"

namespace hf {


struct BnGeom {
    int vec;       // floats per thread access (4 or 1)
    int cv;        // vectors per row
    int rpb;       // rows handled concurrently by one block
    int threads;   // cv * rpb
    int nblk;
    long long rows_per_block;
};

constexpr int kBnRowsPerThread = 8;

static BnGeom bn_geom(long long rows, int c)
{
    BnGeom g;
    g.vec = (c % 4 == 0) ? 4 : 1;
    g.cv = c / g.vec;
    g.rpb = g.cv >= 256 ? 1 : 256 / g.cv;
    g.threads = g.cv * g.rpb;
    // >= kBnRowsPerThread rows per thread before another block is opened: a 16384 x 64 tensor is 128 blocks (one partial line each
    // per channel), not 1024 one-row-per-thread blocks whose partials (2 x 64 floats at an 8 KB stride) outweigh the tensor
    const long long per_block = static_cast<long long>(g.rpb) * HF_DIAG_INT("HF_BN_ROWS_PER_THREAD", kBnRowsPerThread);
    long long nblk = (rows + per_block - 1) / per_block;
    if (nblk > kBnMaxBlocks) nblk = kBnMaxBlocks;
    if (nblk < 1) nblk = 1;
    long long rpbk = (rows + nblk - 1) / nblk;
    rpbk = (rpbk + g.rpb - 1) / g.rpb * g.rpb;
    g.rows_per_block = rpbk;
    g.nblk = static_cast<int>((rows + rpbk - 1) / rpbk);
    return g;
}

template <int VEC> struct VecT { typedef float type __attribute__((ext_vector_type(VEC))); };
template <> struct VecT<1> { typedef float type; };

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type ldv(const float *p) { return *reinterpret_cast<const typename VecT<VEC>::type *>(p); }
// NT: a streaming load (a template parameter: as a run-time select between two loads of one address the compiler merged them
// into a plain load).  The passes over tensors that cannot stay in the 256 MB Infinity Cache between the pass that
// brings them in and the pass that reads them again (forward: one tensor above ~200 MB; backward: x and dy together) read 20-25 %
// faster when their loads do not allocate there (1 M x 64: forward 168 -> 136 us, backward 273 -> 223 us); smaller tensors ARE
// re-read from the cache by the second pass and lose 5 % with streaming loads (scripts/probes/bn_stream_timing.py)
template <int VEC, bool NT>
__device__ __forceinline__ typename VecT<VEC>::type ldv_s(const float *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const typename VecT<VEC>::type *>(p));
    else return ldv<VEC>(p);
}
template <int VEC>
__device__ __forceinline__ void stv(float *p, typename VecT<VEC>::type v) { *reinterpret_cast<typename VecT<VEC>::type *>(p) = v; }
template <int VEC> __device__ __forceinline__ float vget(const typename VecT<VEC>::type &v, int i) { return v[i]; }
template <> __device__ __forceinline__ float vget<1>(const float &v, int) { return v; }
template <int VEC> __device__ __forceinline__ void vset(typename VecT<VEC>::type &v, int i, float f) { v[i] = f; }
template <> __device__ __forceinline__ void vset<1>(float &v, int, float f) { v = f; }

// block-level sum over the `rpb` row-lanes that share a channel vector; result valid where rsub == 0
template <int VEC>
__device__ __forceinline__ void reduce_rows(float (&a)[VEC], float (&b)[VEC], int cv, int rpb, int cvec, int rsub,
                                            float *smem)
{
    // smem: rpb * cv * 2 * VEC floats
    float *mine = smem + (static_cast<size_t>(rsub) * cv + cvec) * 2 * VEC;
#pragma unroll
    for (int i = 0; i < VEC; ++i) { mine[i] = a[i]; mine[VEC + i] = b[i]; }
    __syncthreads();
    if (rsub == 0) {
        for (int r = 1; r < rpb; ++r) {
            const float *o = smem + (static_cast<size_t>(r) * cv + cvec) * 2 * VEC;
#pragma unroll
            for (int i = 0; i < VEC; ++i) { a[i] += o[i]; b[i] += o[VEC + i]; }
        }
    }
}

// `relu` argument of the BatchNorm entry points: bit 0 = ReLU after the normalisation (tf_util.conv2d), bit 1 = ELU BEFORE it
// (pointfly.dense / conv2d: linear -> ELU -> batch_normalization, hf/core/pointfly.py:371-497): the statistics, the
// normalisation and the backward pass then see elu(x), computed on load -- the activation never exists in memory.
constexpr int kBnRelu = 1, kBnEluIn = 2;

// The streaming passes below keep kBnUnroll rows of a thread in flight: one 16-byte load per thread and iteration, waited for at
// once (what the plain loop compiled to), left every wave idle for a full memory round trip per row -- 2.3-3 TB/s on the
// million-row tensors of the X-Conv layers; with four (eight in the two-operand passes) requests per thread outstanding the
// passes stream (scripts/probes/bn_stream_timing.py).  The ELU on load is exp(x) - 1 on the hardware exponential -- TensorFlow's
// own formula for tf.nn.elu (relu_op_functor.h: features.exp() - 1) -- three instructions instead of expm1f's 25 per element, which
// at 64 channels x 10^6 rows was as long as the memory time itself; its slope is the same exponential (EluGrad: (y + 1) dy).
constexpr int kBnUnroll = 4;

// elu(x) and, on request, its slope, branch-free (select, no exec masking around the exponential)
__device__ __forceinline__ float elu_stream(float x) { return x > 0.0f ? x : __builtin_amdgcn_exp2f(x * 1.44269504088896340736f) - 1.0f; }
__device__ __forceinline__ float elu_stream(float x, float &slope)
{
    const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
    slope = x > 0.0f ? 1.0f : e;
    return x > 0.0f ? x : e - 1.0f;
}

// Dropout fused into the normalisation (pointfly's dense -> dropout, pointcnn.py:371-384, rpn_model.py:556-568; tf.layers.dropout
// keeps an element with probability 1 - rate and scales it by 1 / (1 - rate)).  The framework form is a pass of its own in each
// direction (read, write, plus a mask tensor); here the keep decision is a counter hash of (seed of the call, element index),
// evaluated in the apply pass and AGAIN in the two backward passes: no mask in memory, no extra pass.  16 random bits per element
// (the rate is resolved to 1 / 65536); one 32-bit hash feeds two elements.
struct BnDrop {
    const unsigned long long *seed;   // device: the seed of this forward call (written by the statistics finalize kernel)
    unsigned thresh;                  // keep iff bits >= thresh, thresh = round(rate * 65536)
    float scale;                      // 1 / (1 - rate)
};

__device__ __forceinline__ unsigned mix32(unsigned h)   // a full-avalanche 32-bit finalizer (two multiplies)
{
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}

// multipliers (scale or 0) of the VEC elements of vector number vidx = row * cv + cvec
template <int VEC>
__device__ __forceinline__ void drop_multipliers(unsigned long long vidx, unsigned s0, unsigned s1, unsigned thresh, float scale, float (&m)[VEC])
{
    const unsigned h1 = mix32((static_cast<unsigned>(vidx) ^ s0) + mix32(static_cast<unsigned>(vidx >> 32) ^ s1));
    m[0] = (h1 & 0xffffu) >= thresh ? scale : 0.0f;
    if constexpr (VEC == 4) {
        const unsigned h2 = mix32(h1 ^ 0x68bc21ebu);
        m[1] = (h1 >> 16) >= thresh ? scale : 0.0f;
        m[2] = (h2 & 0xffffu) >= thresh ? scale : 0.0f;
        m[3] = (h2 >> 16) >= thresh ? scale : 0.0f;
    }
}

// A thread's rows are rsub, rsub + rpb, ... < local (rows of the block); operands are addressed as a uniform block base plus a 32-bit
// element offset (bn_geom keeps rows_per_block x row stride below 2^31).
// partial[blk][0][c] = sum x, partial[blk][1][c] = sum x^2 over the block's rows
template <int VEC, bool ELU, bool NT>
__global__ void bn_stats_kernel(long long rows, int c, int cv, int rpb, long long rows_per_block,
                                const float *__restrict__ x, float *__restrict__ partial)
{
    extern __shared__ float smem[];
    typedef typename VecT<VEC>::type V;
    const int t = threadIdx.x;
    const int cvec = t % cv, rsub = t / cv;
    const long long r0 = blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float s[VEC], q[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { s[i] = 0.f; q[i] = 0.f; }
    auto add = [&](const V &v) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) { float f = vget<VEC>(v, i); if (ELU) f = elu_stream(f); s[i] += f; q[i] += f * f; }
    };
    const int local = static_cast<int>(r1 - r0);
    const float *bx = x + r0 * c;
    const int sx = rpb * c;
    int ox = rsub * c + cvec * VEC, row = rsub;
    for (; row + (kBnUnroll - 1) * rpb < local; row += kBnUnroll * rpb, ox += kBnUnroll * sx) {
        V v[kBnUnroll];
#pragma unroll
        for (int u = 0; u < kBnUnroll; ++u) v[u] = ldv_s<VEC, NT>(bx + ox + u * sx);
#pragma unroll
        for (int u = 0; u < kBnUnroll; ++u) add(v[u]);
    }
    for (; row < local; row += rpb, ox += sx) add(ldv_s<VEC, NT>(bx + ox));
    reduce_rows<VEC>(s, q, cv, rpb, cvec, rsub, smem);
    if (rsub == 0) {
        // partial[which][channel][block]: the finalize kernel reads one channel's partials contiguously
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            partial[(static_cast<size_t>(cvec * VEC + i)) * kBnMaxBlocks + blockIdx.x] = s[i];
            partial[(static_cast<size_t>(c + cvec * VEC + i)) * kBnMaxBlocks + blockIdx.x] = q[i];
        }
    }
}

// Sums of one channel's two partial rows in fp64, fixed order (deterministic), no atomics.  Few partials (nblk <= kBnFinalizeWideFrom):
// one WAVE per channel, four channels per 256-thread workgroup, combined across the lanes by shuffles.  Many (the million-row
// tensors: up to 2048 per row): one 256-thread WORKGROUP per channel -- a wave alone walked 32 dependent-latency steps per lane
// (8-9 us per finalize launch, 120 launches per step); four waves with all of a lane's <= 8 loads requested up front take a
// third of that.  Valid in every lane of the channel's first wave on return (`owner`).
constexpr int kBnFinalizeChannels = 4;
constexpr int kBnFinalizeWideFrom = 256;
__host__ __device__ inline bool bn_finalize_wide(int nblk) { return nblk > kBnFinalizeWideFrom; }
static inline int bn_finalize_grid(int c, int nblk) { return bn_finalize_wide(nblk) ? c : (c + kBnFinalizeChannels - 1) / kBnFinalizeChannels; }
__device__ __forceinline__ int bn_finalize_channel(int nblk)
{
    return bn_finalize_wide(nblk) ? static_cast<int>(blockIdx.x) : static_cast<int>(blockIdx.x * kBnFinalizeChannels + (threadIdx.x >> 6));
}

// returns true in the lanes that hold the result (all lanes of the channel's first wave)
__device__ __forceinline__ bool bn_reduce_channel(const float *__restrict__ partial, int c, int nblk, int ch, double &s,
                                                  double &q)
{
    __shared__ double cross[2][kBnFinalizeChannels];
    const bool wide = bn_finalize_wide(nblk);   // uniform over the grid
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int first = wide ? static_cast<int>(threadIdx.x) : lane, stride = wide ? 64 * kBnFinalizeChannels : 64;
    double a = 0.0, b = 0.0;
    const float *p0 = partial + static_cast<size_t>(ch) * kBnMaxBlocks;
    const float *p1 = partial + static_cast<size_t>(c + ch) * kBnMaxBlocks;
    constexpr int kBatch = 8;
    for (int i = first; i < nblk; i += stride * kBatch) {
        float va[kBatch], vb[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int k = i + u * stride;
            va[u] = k < nblk ? p0[k] : 0.0f;
            vb[u] = k < nblk ? p1[k] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) { a += static_cast<double>(va[u]); b += static_cast<double>(vb[u]); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    if (wide) {
        if (lane == 0) { cross[0][wave] = a; cross[1][wave] = b; }
        __syncthreads();
        a = cross[0][0]; b = cross[1][0];
#pragma unroll
        for (int w = 1; w < kBnFinalizeChannels; ++w) { a += cross[0][w]; b += cross[1][w]; }
    }
    s = a;
    q = b;
    return !wide || wave == 0;
}

// batch statistics, running statistics: running = (1-m)*running + m*batch with the BIASED batch variance -- what
// tf.contrib.layers.batch_norm(fused=False) accumulates from tf.nn.moments (hf/core/feature_extractors/tf_util.py:571-581)
// and tf.layers.batch_normalization does too (torch would use the Bessel-corrected variance here)
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(long long rows, int c, int nblk,
                                                                const float *__restrict__ partial, float eps,
                                                                float momentum, float *__restrict__ running_mean,
                                                                float *__restrict__ running_var,
                                                                float *__restrict__ save_mean,
                                                                float *__restrict__ save_invstd,
                                                                unsigned long long *__restrict__ drop_state,
                                                                unsigned long long *__restrict__ seed_out,
                                                                unsigned long long salt)
{
    if (drop_state && blockIdx.x == 0 && threadIdx.x == 0) {
        // one more forward call of this layer: its seed (splitmix64 of base seed, caller's salt and call number) for the apply pass
        // that follows and for the backward passes
        const unsigned long long calls = drop_state[1] + 1ull;
        drop_state[1] = calls;
        unsigned long long z = drop_state[0] + salt * 0xbf58476d1ce4e5b9ull + calls * 0x9e3779b97f4a7c15ull;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        seed_out[0] = z ^ (z >> 31);
    }
    const int ch = bn_finalize_channel(nblk);
    if (ch >= c) return;   // whole waves (narrow form) or whole workgroups (wide form) leave together
    double s, q;
    if (!bn_reduce_channel(partial, c, nblk, ch, s, q) || (threadIdx.x & 63) != 0) return;
    const double mean = s / static_cast<double>(rows);
    double var = q / static_cast<double>(rows) - mean * mean;
    if (var < 0.0) var = 0.0;
    save_mean[ch] = static_cast<float>(mean);
    save_invstd[ch] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
    if (running_mean) running_mean[ch] = (1.0f - momentum) * running_mean[ch] + momentum * static_cast<float>(mean);
    if (running_var) running_var[ch] = (1.0f - momentum) * running_var[ch] + momentum * static_cast<float>(var);
}

// y = relu?(gamma*invstd*(x-mean) + beta)
template <int VEC, bool ELU, bool DROP, bool NT>
__global__ void bn_apply_kernel(long long rows, int c, int cv, int rpb, long long rows_per_block,
                                const float *__restrict__ x, const float *__restrict__ gamma,
                                const float *__restrict__ beta, const float *__restrict__ mean,
                                const float *__restrict__ invstd, int relu, float *__restrict__ y, long long ldy, BnDrop drop)
{
    typedef typename VecT<VEC>::type V;
    const int t = threadIdx.x;
    const int cvec = t % cv, rsub = t / cv;
    float a[VEC], b[VEC], mu[VEC];   // y = a (x - mu) + beta: the difference first, as the reference graph forms it
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const int ch = cvec * VEC + i;
        a[i] = gamma[ch] * invstd[ch];
        b[i] = beta[ch];
        mu[i] = mean[ch];
    }
    const long long r0 = blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    unsigned ds0 = 0u, ds1 = 0u;
    if (DROP) { const unsigned long long sd = *drop.seed; ds0 = static_cast<unsigned>(sd); ds1 = static_cast<unsigned>(sd >> 32); }
    auto one = [&](const V &v, int row) -> V {
        V o;
        float m[VEC];
        if (DROP) drop_multipliers<VEC>(static_cast<unsigned long long>(r0 + row) * cv + cvec, ds0, ds1, drop.thresh, drop.scale, m);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float xv = vget<VEC>(v, i);
            if (ELU) xv = elu_stream(xv);
            float h = a[i] * (xv - mu[i]) + b[i];
            if (relu & kBnRelu) h = fmaxf(h, 0.0f);
            if (DROP) h = h * m[i];
            vset<VEC>(o, i, h);
        }
        return o;
    };
    const int local = static_cast<int>(r1 - r0);
    const float *bx = x + r0 * c;
    float *by = y + r0 * ldy;                         // ldy >= c: the row may be a slice of a wider (concat) buffer
    const int sx = rpb * c, sy = rpb * static_cast<int>(ldy);
    int ox = rsub * c + cvec * VEC, oy = rsub * static_cast<int>(ldy) + cvec * VEC, row = rsub;
    for (; row + (kBnUnroll - 1) * rpb < local; row += kBnUnroll * rpb, ox += kBnUnroll * sx, oy += kBnUnroll * sy) {
        V v[kBnUnroll];
#pragma unroll
        for (int u = 0; u < kBnUnroll; ++u) v[u] = ldv_s<VEC, NT>(bx + ox + u * sx);
#pragma unroll
        for (int u = 0; u < kBnUnroll; ++u) stv<VEC>(by + oy + u * sy, one(v[u], row + u * rpb));
    }
    for (; row < local; row += rpb, ox += sx, oy += sy) stv<VEC>(by + oy, one(ldv_s<VEC, NT>(bx + ox), row));
}

// partial[blk][0][c] = sum dh, partial[blk][1][c] = sum dh*xhat  (dh = dy masked by the ReLU of a*x+b)
template <int VEC, bool ELU, bool DROP, bool NT>
__global__ void bn_bwd_reduce_kernel(long long rows, int c, int cv, int rpb, long long rows_per_block,
                                     const float *__restrict__ x, const float *__restrict__ dy,
                                     const float *__restrict__ gamma, const float *__restrict__ beta,
                                     const float *__restrict__ mean, const float *__restrict__ invstd, int relu,
                                     float *__restrict__ partial, long long lddy, BnDrop drop)
{
    extern __shared__ float smem[];
    typedef typename VecT<VEC>::type V;
    const int t = threadIdx.x;
    const int cvec = t % cv, rsub = t / cv;
    float a[VEC], b[VEC], mu[VEC], is[VEC], s1[VEC], s2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const int ch = cvec * VEC + i;
        mu[i] = mean[ch]; is[i] = invstd[ch];
        a[i] = gamma[ch] * is[i];
        b[i] = beta[ch];
        s1[i] = 0.f; s2[i] = 0.f;
    }
    const long long r0 = blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    unsigned ds0 = 0u, ds1 = 0u;
    if (DROP) { const unsigned long long sd = *drop.seed; ds0 = static_cast<unsigned>(sd); ds1 = static_cast<unsigned>(sd >> 32); }
    auto add = [&](const V &v, const V &g, int row) {
        float m[VEC];
        if (DROP) drop_multipliers<VEC>(static_cast<unsigned long long>(r0 + row) * cv + cvec, ds0, ds1, drop.thresh, drop.scale, m);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float xv = vget<VEC>(v, i);
            if (ELU) xv = elu_stream(xv);
            float dh = vget<VEC>(g, i);
            if (DROP) dh = dh * m[i];
            if ((relu & kBnRelu) && !(a[i] * (xv - mu[i]) + b[i] > 0.0f)) dh = 0.0f;
            s1[i] += dh;
            s2[i] += dh * ((xv - mu[i]) * is[i]);
        }
    };
    const int local = static_cast<int>(r1 - r0);
    const float *bx = x + r0 * c, *bg = dy + r0 * lddy;
    const int sx = rpb * c, sg = rpb * static_cast<int>(lddy);
    int ox = rsub * c + cvec * VEC, og = rsub * static_cast<int>(lddy) + cvec * VEC, row = rsub;
    for (; row + (kBnUnroll - 1) * rpb < local; row += kBnUnroll * rpb, ox += kBnUnroll * sx, og += kBnUnroll * sg) {
        V v[kBnUnroll], g[kBnUnroll];
#pragma unroll
        for (int u = 0; u < kBnUnroll; ++u) { v[u] = ldv_s<VEC, NT>(bx + ox + u * sx); g[u] = ldv_s<VEC, NT>(bg + og + u * sg); }
#pragma unroll
        for (int u = 0; u < kBnUnroll; ++u) add(v[u], g[u], row + u * rpb);
    }
    for (; row < local; row += rpb, ox += sx, og += sg) add(ldv_s<VEC, NT>(bx + ox), ldv_s<VEC, NT>(bg + og), row);
    reduce_rows<VEC>(s1, s2, cv, rpb, cvec, rsub, smem);
    if (rsub == 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            partial[(static_cast<size_t>(cvec * VEC + i)) * kBnMaxBlocks + blockIdx.x] = s1[i];
            partial[(static_cast<size_t>(c + cvec * VEC + i)) * kBnMaxBlocks + blockIdx.x] = s2[i];
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(int c, int nblk, const float *__restrict__ partial,
                                                              float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    const int ch = bn_finalize_channel(nblk);
    if (ch >= c) return;
    double s1, s2;
    if (!bn_reduce_channel(partial, c, nblk, ch, s1, s2) || (threadIdx.x & 63) != 0) return;
    dbeta[ch] = static_cast<float>(s1);
    dgamma[ch] = static_cast<float>(s2);
}

// dx = gamma*invstd*(dh - dbeta/R - xhat*dgamma/R)
template <int VEC, bool ELU, bool DROP, bool NT>
__global__ void bn_bwd_dx_kernel(long long rows, int c, int cv, int rpb, long long rows_per_block,
                                 const float *__restrict__ x, const float *__restrict__ dy,
                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                 const float *__restrict__ mean, const float *__restrict__ invstd,
                                 const float *__restrict__ dgamma, const float *__restrict__ dbeta, int relu,
                                 float *__restrict__ dx, float *__restrict__ colsum_partial, long long lddy, BnDrop drop)
{
    extern __shared__ float smem[];
    typedef typename VecT<VEC>::type V;
    const int t = threadIdx.x;
    const int cvec = t % cv, rsub = t / cv;
    const float inv_r = 1.0f / static_cast<float>(rows);
    float a[VEC], b[VEC], mu[VEC], is[VEC], c1[VEC], c2[VEC], cs[VEC], unused[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const int ch = cvec * VEC + i;
        mu[i] = mean[ch]; is[i] = invstd[ch];
        a[i] = gamma[ch] * is[i];
        b[i] = beta[ch];
        c1[i] = dbeta[ch] * inv_r;
        c2[i] = dgamma[ch] * inv_r;
        cs[i] = 0.f; unused[i] = 0.f;
    }
    const long long r0 = blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    unsigned ds0 = 0u, ds1 = 0u;
    if (DROP) { const unsigned long long sd = *drop.seed; ds0 = static_cast<unsigned>(sd); ds1 = static_cast<unsigned>(sd >> 32); }
    auto one = [&](const V &v, const V &g, int row) -> V {
        V o;
        float m[VEC];
        if (DROP) drop_multipliers<VEC>(static_cast<unsigned long long>(r0 + row) * cv + cvec, ds0, ds1, drop.thresh, drop.scale, m);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float slope = 1.0f;
            float xv = vget<VEC>(v, i);
            if (ELU) xv = elu_stream(xv, slope);
            float dh = vget<VEC>(g, i);
            if (DROP) dh = dh * m[i];
            if ((relu & kBnRelu) && !(a[i] * (xv - mu[i]) + b[i] > 0.0f)) dh = 0.0f;
            const float xhat = (xv - mu[i]) * is[i];
            float d = a[i] * (dh - c1[i] - xhat * c2[i]);
            if (ELU) d = d * slope;
            cs[i] += d;
            vset<VEC>(o, i, d);
        }
        return o;
    };
    const int local = static_cast<int>(r1 - r0);
    const float *bx = x + r0 * c, *bg = dy + r0 * lddy;
    float *bo = dx + r0 * c;
    const int sx = rpb * c, sg = rpb * static_cast<int>(lddy);
    int ox = rsub * c + cvec * VEC, og = rsub * static_cast<int>(lddy) + cvec * VEC, row = rsub;
    for (; row + (kBnUnroll - 1) * rpb < local; row += kBnUnroll * rpb, ox += kBnUnroll * sx, og += kBnUnroll * sg) {
        V v[kBnUnroll], g[kBnUnroll];
#pragma unroll
        for (int u = 0; u < kBnUnroll; ++u) { v[u] = ldv_s<VEC, NT>(bx + ox + u * sx); g[u] = ldv_s<VEC, NT>(bg + og + u * sg); }
#pragma unroll
        for (int u = 0; u < kBnUnroll; ++u) stv<VEC>(bo + ox + u * sx, one(v[u], g[u], row + u * rpb));
    }
    for (; row < local; row += rpb, ox += sx, og += sg) stv<VEC>(bo + ox, one(ldv_s<VEC, NT>(bx + ox), ldv_s<VEC, NT>(bg + og), row));
    if (colsum_partial) {  // column sums of dx = the bias gradient of the Linear that produced x
        reduce_rows<VEC>(cs, unused, cv, rpb, cvec, rsub, smem);
        if (rsub == 0) {
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                colsum_partial[(static_cast<size_t>(cvec * VEC + i)) * kBnMaxBlocks + blockIdx.x] = cs[i];
        }
    }
}

__global__ __launch_bounds__(256) void bn_colsum_finalize_kernel(int c, int nblk, const float *__restrict__ partial,
                                                                 float *__restrict__ colsum)
{
    const int ch = bn_finalize_channel(nblk);
    if (ch >= c) return;
    double s, q;
    if (bn_reduce_channel(partial, c, nblk, ch, s, q) && (threadIdx.x & 63) == 0) colsum[ch] = static_cast<float>(s);
}

// ------------------------------------------------------------------------------------------
// Short tensors (rows <= kBnSmallMaxRows, C % 4 == 0): ONE launch per direction instead of three.  At one frame per GPU the
// step holds ~45 such BatchNorms (the K*K-channel X-transformation layers and everything below 4096 points): three launches of
// 5-9 us each for a tensor that fits a few CUs' registers.  A workgroup owns `cvw` channel vectors (<= 128 bytes of a row) and
// ALL rows: every thread keeps its <= 8 rows in registers, the sums meet in LDS, the same registers are normalised and stored.
// No inter-workgroup traffic, no partials in memory.
// ------------------------------------------------------------------------------------------
// the single-launch kernels run ~4 waves per CU on a handful of CUs, so the activation's instruction count shows: exp(x) - 1 on
// the hardware exponential (TensorFlow's own formula for tf.nn.elu, relu_op_functor.h; the lifting kernels of gemm.hip use it
// too) and its gradient from the OUTPUT, (y + 1) dy for y <= 0, as TensorFlow's EluGrad does -- one v_exp_f32 per element
__device__ __forceinline__ float elu_hw(float x) { return x > 0.0f ? x : __expf(x) - 1.0f; }
__device__ __forceinline__ float elu_slope_from_output(float y) { return y > 0.0f ? 1.0f : y + 1.0f; }

constexpr int kBnSmallRowsPerThread = 8;
constexpr int kBnSmallMaxRows = 4096;

struct BnSmallGeom { int cvw, threads, rpb, nwg; };

static bool bn_small_geom(long long rows, int c, BnSmallGeom &g)
{
    if (c % 4 != 0 || rows > HF_DIAG_INT("HF_BN_SMALL_ROWS", kBnSmallMaxRows)) return false;
    const int cv = c / 4;
    int cvw = HF_DIAG_INT("HF_BN_SMALL_CVW", 8);
    // measured (scripts/probes/bn_small_timing.py, profiles/r04_bn_small_timing.txt): 256-thread workgroups win up to 1024 rows, 512 beyond;
    // 1024-thread workgroups (a whole CU's registers in the backward kernel) lose to both, and at 4096 rows x > 128 channels the
    // line re-fetch of the narrow channel slices (every workgroup touches every row's 128-byte line) costs more than two launches
    if (rows > 2048 && c > 128) return false;
    const int tmax = HF_DIAG_INT("HF_BN_SMALL_THREADS", rows <= 1024 ? 256 : 512);
    while (cvw > 1 && (cvw > cv || static_cast<long long>(cvw) * rows > tmax * kBnSmallRowsPerThread)) cvw >>= 1;
    long long want = rows < tmax / cvw ? rows : tmax / cvw;      // row lanes
    int threads = static_cast<int>((want * cvw + 63) / 64 * 64);
    if (threads > tmax) threads = tmax;
    g.cvw = cvw;
    g.threads = threads;
    g.rpb = threads / cvw;
    g.nwg = (cv + cvw - 1) / cvw;
    return static_cast<long long>(g.rpb) * kBnSmallRowsPerThread >= rows;
}

// sums of v[0..7] over the lanes that share a channel vector (stride cvw inside a wave), over the waves through `part`, in fp64;
// on return the lanes t = cl*8 + j of wave 0 (t < cvw*8) hold the total of component j of channel vector cl
__device__ __forceinline__ double bn_small_reduce(float (&v)[8], int cvw, float (*part)[8][8])
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6;
    for (int off = cvw; off < 64; off <<= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += __shfl_xor(v[j], off);
    }
    if (lane < cvw) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[wave][lane][j] = v[j];
    }
    __syncthreads();
    double acc = 0.0;
    if (t < cvw * 8) {
        for (int w = 0; w < nw; ++w) acc += static_cast<double>(part[w][t >> 3][t & 7]);
    }
    return acc;
}

__global__ __launch_bounds__(1024) void bn_small_fwd_kernel(int rows, int c, int cvw, int rpb, const float *__restrict__ x,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           float eps, float momentum, float *__restrict__ running_mean,
                                                           float *__restrict__ running_var, int relu, float *__restrict__ y,
                                                           long long ldy, float *__restrict__ save_mean,
                                                           float *__restrict__ save_invstd)
{
    __shared__ float part[16][8][8];
    __shared__ float coef[8][12];
    typedef VecT<4>::type f4;
    const int t = threadIdx.x;
    const int cl = t % cvw, rsub = t / cvw;
    const int cv = c >> 2;
    const int cvec = blockIdx.x * cvw + cl;
    const bool chan_ok = cvec < cv;
    const int cvec_c = chan_ok ? cvec : cv - 1;
    f4 xv[kBnSmallRowsPerThread];
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int i = 0; i < kBnSmallRowsPerThread; ++i) {
        const int r = rsub + i * rpb;
        const bool ok = chan_ok && r < rows;
        const f4 v = ldv<4>(x + static_cast<long long>(r < rows ? r : rows - 1) * c + cvec_c * 4);   // always a valid address
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float f = v[j];
            if (relu & kBnEluIn) f = elu_hw(f);
            f = ok ? f : 0.f;
            xv[i][j] = f;
            acc[j] += f;
            acc[4 + j] += f * f;
        }
    }
    const double tot = bn_small_reduce(acc, cvw, part);
    if (t < cvw * 8) {
        const double other = __shfl_xor(tot, 4);      // lanes j < 4 hold sum x and fetch sum x^2 from lane j + 4
        const int j = t & 7, ch = (blockIdx.x * cvw + (t >> 3)) * 4 + (j & 3);
        if (j < 4 && ch < c) {
            const double mean = tot / static_cast<double>(rows);
            double var = other / static_cast<double>(rows) - mean * mean;
            if (var < 0.0) var = 0.0;
            const float is = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
            save_mean[ch] = static_cast<float>(mean);
            save_invstd[ch] = is;
            if (running_mean) running_mean[ch] = (1.0f - momentum) * running_mean[ch] + momentum * static_cast<float>(mean);
            if (running_var) running_var[ch] = (1.0f - momentum) * running_var[ch] + momentum * static_cast<float>(var);
            coef[t >> 3][j] = gamma[ch] * is;
            coef[t >> 3][4 + j] = static_cast<float>(mean);
            coef[t >> 3][8 + j] = beta[ch];
        }
    }
    __syncthreads();
    if (!chan_ok) return;
    float a[4], mu[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = coef[cl][j]; mu[j] = coef[cl][4 + j]; b[j] = coef[cl][8 + j]; }
#pragma unroll
    for (int i = 0; i < kBnSmallRowsPerThread; ++i) {
        const int r = rsub + i * rpb;
        if (r < rows) {
            f4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float h = a[j] * (xv[i][j] - mu[j]) + b[j];
                if (relu & kBnRelu) h = fmaxf(h, 0.0f);
                o[j] = h;
            }
            stv<4>(y + static_cast<long long>(r) * ldy + cvec * 4, o);
        }
    }
}

__global__ __launch_bounds__(1024) void bn_small_bwd_kernel(int rows, int c, int cvw, int rpb, const float *__restrict__ x,
                                                           const float *__restrict__ dy, long long lddy,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           const float *__restrict__ mean, const float *__restrict__ invstd,
                                                           int relu, float *__restrict__ dx, float *__restrict__ dgamma,
                                                           float *__restrict__ dbeta, float *__restrict__ dx_colsum)
{
    __shared__ float part[16][8][8];
    __shared__ float tot_s[8][8];
    typedef VecT<4>::type f4;
    const int t = threadIdx.x;
    const int cl = t % cvw, rsub = t / cvw;
    const int cv = c >> 2;
    const int cvec = blockIdx.x * cvw + cl;
    const bool chan_ok = cvec < cv;
    const int cvec_c = chan_ok ? cvec : cv - 1;
    float a[4], b[4], mu[4], is[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ch = cvec_c * 4 + j;
        mu[j] = mean[ch]; is[j] = invstd[ch];
        a[j] = gamma[ch] * is[j];
        b[j] = beta[ch];
    }
    f4 xv[kBnSmallRowsPerThread], gv[kBnSmallRowsPerThread];    // x behind the ELU (if any); dh (dy behind the ReLU mask)
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int i = 0; i < kBnSmallRowsPerThread; ++i) {
        const int r = rsub + i * rpb;
        const bool ok = chan_ok && r < rows;
        const long long rc = r < rows ? r : rows - 1;
        const f4 xr = ldv<4>(x + rc * c + cvec_c * 4);
        const f4 g = ldv<4>(dy + rc * lddy + cvec_c * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xe = (relu & kBnEluIn) ? elu_hw(xr[j]) : xr[j];
            xv[i][j] = xe;
            float dh = ok ? g[j] : 0.f;
            if ((relu & kBnRelu) && !(a[j] * (xe - mu[j]) + b[j] > 0.0f)) dh = 0.0f;
            gv[i][j] = dh;
            acc[j] += dh;
            acc[4 + j] += dh * ((xe - mu[j]) * is[j]);
        }
    }
    const double tot = bn_small_reduce(acc, cvw, part);
    if (t < cvw * 8) {
        const int j = t & 7, ch = (blockIdx.x * cvw + (t >> 3)) * 4 + (j & 3);
        const float f = static_cast<float>(tot);
        tot_s[t >> 3][j] = f;
        if (ch < c) { if (j < 4) dbeta[ch] = f; else dgamma[ch] = f; }
    }
    __syncthreads();
    const float inv_r = 1.0f / static_cast<float>(rows);
    float c1[4], c2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { c1[j] = tot_s[cl][j] * inv_r; c2[j] = tot_s[cl][4 + j] * inv_r; acc[j] = 0.f; acc[4 + j] = 0.f; }
#pragma unroll
    for (int i = 0; i < kBnSmallRowsPerThread; ++i) {
        const int r = rsub + i * rpb;
        const bool ok = chan_ok && r < rows;
        f4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xe = xv[i][j];
            const float xhat = (xe - mu[j]) * is[j];
            float d = a[j] * (gv[i][j] - c1[j] - xhat * c2[j]);
            if (relu & kBnEluIn) d = d * elu_slope_from_output(xe);
            o[j] = d;
            acc[j] += ok ? d : 0.f;
        }
        if (ok) stv<4>(dx + static_cast<long long>(r) * c + cvec * 4, o);
    }
    if (dx_colsum) {   // column sums of dx = the bias gradient of the Linear that produced x (uniform branch)
        const double cs = bn_small_reduce(acc, cvw, part);      // its barrier also orders the reuse of `part`
        if (t < cvw * 8) {
            const int j = t & 7, ch = (blockIdx.x * cvw + (t >> 3)) * 4 + j;
            if (j < 4 && ch < c) dx_colsum[ch] = static_cast<float>(cs);
        }
    }
}

// ------------------------------------------------------------------------------------------
// BN + ReLU + max over the K rows of a group, fused (the tail of a set-abstraction MLP:
// tf_util.conv2d(..., bn=True) followed by tf.reduce_max(axis=[2]), pointnet_util.py:156-176).
// The normalised activation (G*K, C) is never written: the forward pass reads z once and keeps the
// per-(group, channel) maximum and its row; the backward pass rebuilds the one-hot dy from that row.
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ void bn_pool_fwd_kernel(long long groups, int k, int c, int cv, int rpb, long long groups_per_block,
                                   const float *__restrict__ z, const float *__restrict__ gamma,
                                   const float *__restrict__ beta, const float *__restrict__ mean,
                                   const float *__restrict__ invstd, float *__restrict__ pooled,
                                   unsigned char *__restrict__ argmax)
{
    const int t = threadIdx.x;
    const int cvec = t % cv, rsub = t / cv;
    float a[VEC], b[VEC], mu[VEC];   // y = a (x - mu) + beta: the difference first, as the reference graph forms it
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const int ch = cvec * VEC + i;
        a[i] = gamma[ch] * invstd[ch];
        b[i] = beta[ch];
        mu[i] = mean[ch];
    }
    const long long g0 = blockIdx.x * groups_per_block;
    const long long g1 = g0 + groups_per_block < groups ? g0 + groups_per_block : groups;
    for (long long g = g0 + rsub; g < g1; g += rpb) {
        float best[VEC];
        int bk[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { best[i] = -1.0f; bk[i] = 0; }  // relu output is >= 0
        const float *zg = z + (g * k) * c + cvec * VEC;
        for (int kk = 0; kk < k; ++kk) {
            const typename VecT<VEC>::type v = ldv<VEC>(zg + static_cast<long long>(kk) * c);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float h = fmaxf(a[i] * (vget<VEC>(v, i) - mu[i]) + b[i], 0.0f);
                if (h > best[i]) { best[i] = h; bk[i] = kk; }  // first maximum wins
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            pooled[g * c + cvec * VEC + i] = best[i];
            if (argmax) argmax[g * c + cvec * VEC + i] = static_cast<unsigned char>(bk[i]);
        }
    }
}

// sparse reduction for the BN backward: dh is dpooled at the arg-max row (if the ReLU is active there), else 0
template <int VEC>
__global__ void bn_pool_bwd_reduce_kernel(long long groups, int k, int c, int cv, int rpb, long long groups_per_block,
                                          const float *__restrict__ z, const float *__restrict__ dpooled,
                                          const unsigned char *__restrict__ argmax, const float *__restrict__ gamma,
                                          const float *__restrict__ beta, const float *__restrict__ mean,
                                          const float *__restrict__ invstd, float *__restrict__ partial)
{
    extern __shared__ float smem[];
    const int t = threadIdx.x;
    const int cvec = t % cv, rsub = t / cv;
    float a[VEC], b[VEC], mu[VEC], is[VEC], s1[VEC], s2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const int ch = cvec * VEC + i;
        mu[i] = mean[ch]; is[i] = invstd[ch];
        a[i] = gamma[ch] * is[i];
        b[i] = beta[ch];
        s1[i] = 0.f; s2[i] = 0.f;
    }
    const long long g0 = blockIdx.x * groups_per_block;
    const long long g1 = g0 + groups_per_block < groups ? g0 + groups_per_block : groups;
    for (long long g = g0 + rsub; g < g1; g += rpb) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const int ch = cvec * VEC + i;
            const int kk = argmax[g * c + ch];
            const float xv = z[(g * k + kk) * c + ch];
            float dh = dpooled[g * c + ch];
            if (!(a[i] * (xv - mu[i]) + b[i] > 0.0f)) dh = 0.0f;
            s1[i] += dh;
            s2[i] += dh * ((xv - mu[i]) * is[i]);
        }
    }
    reduce_rows<VEC>(s1, s2, cv, rpb, cvec, rsub, smem);
    if (rsub == 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            partial[(static_cast<size_t>(cvec * VEC + i)) * kBnMaxBlocks + blockIdx.x] = s1[i];
            partial[(static_cast<size_t>(c + cvec * VEC + i)) * kBnMaxBlocks + blockIdx.x] = s2[i];
        }
    }
}

// dense dz over all G*K rows: dz = gamma*invstd*(dh - dbeta/R - xhat*dgamma/R), dh one-hot per (group, channel)
template <int VEC>
__global__ void bn_pool_bwd_dx_kernel(long long rows, int k, int c, int cv, int rpb, long long rows_per_block,
                                      const float *__restrict__ z, const float *__restrict__ dpooled,
                                      const unsigned char *__restrict__ argmax, const float *__restrict__ gamma,
                                      const float *__restrict__ beta, const float *__restrict__ mean,
                                      const float *__restrict__ invstd, const float *__restrict__ dgamma,
                                      const float *__restrict__ dbeta, float *__restrict__ dz,
                                      float *__restrict__ colsum_partial)
{
    extern __shared__ float smem[];
    const int t = threadIdx.x;
    const int cvec = t % cv, rsub = t / cv;
    const float inv_r = 1.0f / static_cast<float>(rows);
    float a[VEC], b[VEC], mu[VEC], is[VEC], c1[VEC], c2[VEC], cs[VEC], unused[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const int ch = cvec * VEC + i;
        mu[i] = mean[ch]; is[i] = invstd[ch];
        a[i] = gamma[ch] * is[i];
        b[i] = beta[ch];
        c1[i] = dbeta[ch] * inv_r;
        c2[i] = dgamma[ch] * inv_r;
        cs[i] = 0.f; unused[i] = 0.f;
    }
    const long long r0 = blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    for (long long r = r0 + rsub; r < r1; r += rpb) {
        const long long g = r / k;
        const int kk = static_cast<int>(r - g * k);
        const typename VecT<VEC>::type v = ldv<VEC>(z + r * c + cvec * VEC);
        const typename VecT<VEC>::type dp = ldv<VEC>(dpooled + g * c + cvec * VEC);
        typename VecT<VEC>::type o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float xv = vget<VEC>(v, i);
            float dh = 0.0f;
            if (argmax[g * c + cvec * VEC + i] == kk && a[i] * (xv - mu[i]) + b[i] > 0.0f) dh = vget<VEC>(dp, i);
            const float xhat = (xv - mu[i]) * is[i];
            const float d = a[i] * (dh - c1[i] - xhat * c2[i]);
            cs[i] += d;
            vset<VEC>(o, i, d);
        }
        stv<VEC>(dz + r * c + cvec * VEC, o);
    }
    if (colsum_partial) {
        reduce_rows<VEC>(cs, unused, cv, rpb, cvec, rsub, smem);
        if (rsub == 0) {
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                colsum_partial[(static_cast<size_t>(cvec * VEC + i)) * kBnMaxBlocks + blockIdx.x] = cs[i];
        }
    }
}

void launch_bn_stats_finalize(long long rows, int c, int nblk, const float *partial, float eps, float momentum,
                              float *running_mean, float *running_var, float *save_mean, float *save_invstd, hipStream_t st)
{
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(bn_finalize_grid(c, nblk)), dim3(64 * kBnFinalizeChannels), 0, st, rows, c, nblk, partial, eps, momentum,
                       running_mean, running_var, save_mean, save_invstd, static_cast<unsigned long long *>(nullptr), static_cast<unsigned long long *>(nullptr), 0ull);
}

void launch_bn_bwd_finalize(int c, int nblk, const float *partial, float *dgamma, float *dbeta, hipStream_t st)
{
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(bn_finalize_grid(c, nblk)), dim3(64 * kBnFinalizeChannels), 0, st, c, nblk, partial, dgamma, dbeta);
}

// dx (rows, cin) = g (rows, cout) W (cout, cin) for a HANDFUL of outputs (the segmentation head, rpn_model.py: 256 -> classes + 1).
// As a GEMM its inner dimension is cout <= 4 (the library: 490 us at 131 072 rows); as scaled additions in the framework it was
// cout passes over dx.  Here: one pass, a thread owns VEC channels (its cout x VEC weights in registers) and strides over rows.
constexpr int kNarrowMaxOut = 4;
template <int VEC>
__global__ void narrow_linear_dx_kernel(long long rows, int cin, int cout, int cv, int rpb, long long rows_per_block,
                                        const float *__restrict__ g, const float *__restrict__ w, float *__restrict__ dx)
{
    typedef typename VecT<VEC>::type V;
    const int t = threadIdx.x;
    const int cvec = t % cv, rsub = t / cv;
    float wr[kNarrowMaxOut][VEC];
#pragma unroll
    for (int j = 0; j < kNarrowMaxOut; ++j)
#pragma unroll
        for (int i = 0; i < VEC; ++i) wr[j][i] = j < cout ? w[static_cast<size_t>(j) * cin + cvec * VEC + i] : 0.0f;
    const long long r0 = blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    for (long long r = r0 + rsub; r < r1; r += rpb) {
        float gv[kNarrowMaxOut];
#pragma unroll
        for (int j = 0; j < kNarrowMaxOut; ++j) gv[j] = j < cout ? g[r * cout + j] : 0.0f;
        V o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            // the framework's order: g_0 w_0, then + g_j w_j for j = 1 .. cout-1
            float acc = gv[0] * wr[0][i];
#pragma unroll
            for (int j = 1; j < kNarrowMaxOut; ++j)
                if (j < cout) acc = acc + gv[j] * wr[j][i];
            vset<VEC>(o, i, acc);
        }
        stv<VEC>(dx + r * cin + cvec * VEC, o);
    }
}

// the streaming passes by vector width, activation-on-load and fused dropout (drop == nullptr: none)
template <template <int, bool, bool> class L, typename... A>
static void bn_dispatch(const BnGeom &g, bool elu, bool drop, A... a)
{
    if (g.vec == 4) {
        if (elu) { if (drop) L<4, true, true>::go(a...); else L<4, true, false>::go(a...); }
        else { if (drop) L<4, false, true>::go(a...); else L<4, false, false>::go(a...); }
    } else {
        if (elu) { if (drop) L<1, true, true>::go(a...); else L<1, true, false>::go(a...); }
        else { if (drop) L<1, false, true>::go(a...); else L<1, false, false>::go(a...); }
    }
}

static const BnDrop kNoDrop = { nullptr, 0u, 1.0f };

static void launch_bn_stats(const BnGeom &g, hipStream_t st, long long rows, int c, const float *x, float *partial, bool elu, int nt)
{
    const size_t lds = sizeof(float) * static_cast<size_t>(g.threads) * 2 * g.vec;
#define HF_BN_L(V, E, N) hipLaunchKernelGGL((bn_stats_kernel<V, E, N>), dim3(g.nblk), dim3(g.threads), lds, st, rows, c, g.cv, g.rpb, g.rows_per_block, x, partial)
    if (g.vec == 4 && nt) { if (elu) HF_BN_L(4, true, true); else HF_BN_L(4, false, true); }   // streaming loads: 16-byte form only
    else if (g.vec == 4) { if (elu) HF_BN_L(4, true, false); else HF_BN_L(4, false, false); }
    else { if (elu) HF_BN_L(1, true, false); else HF_BN_L(1, false, false); }
#undef HF_BN_L
}

template <int V, bool E, bool D> struct BnApplyL {
    static void go(const BnGeom &g, hipStream_t st, long long rows, int c, const float *x, const float *gamma, const float *beta,
                   const float *mean, const float *invstd, int relu, float *y, long long ldy, BnDrop drop, int nt)
    {
        if constexpr (V == 4 && !D) {
            if (nt) { hipLaunchKernelGGL((bn_apply_kernel<4, E, false, true>), dim3(g.nblk), dim3(g.threads), 0, st, rows, c, g.cv, g.rpb, g.rows_per_block, x, gamma,
                           beta, mean, invstd, relu, y, ldy, drop); return; }
        }
        hipLaunchKernelGGL((bn_apply_kernel<V, E, D, false>), dim3(g.nblk), dim3(g.threads), 0, st, rows, c, g.cv, g.rpb, g.rows_per_block, x, gamma,
                           beta, mean, invstd, relu, y, ldy, drop);
    }
};
static void launch_bn_apply(const BnGeom &g, hipStream_t st, long long rows, int c, const float *x, const float *gamma, const float *beta,
                            const float *mean, const float *invstd, int relu, float *y, long long ldy, int nt, BnDrop drop = kNoDrop)
{
    bn_dispatch<BnApplyL>(g, (relu & kBnEluIn) != 0, drop.seed != nullptr, g, st, rows, c, x, gamma, beta, mean, invstd, relu, y, ldy, drop, nt);
}

template <int V, bool E, bool D> struct BnReduceL {
    static void go(const BnGeom &g, hipStream_t st, long long rows, int c, const float *x, const float *dy, const float *gamma,
                   const float *beta, const float *mean, const float *invstd, int relu, float *partial, long long lddy, BnDrop drop, int nt)
    {
        const size_t lds = sizeof(float) * static_cast<size_t>(g.threads) * 2 * g.vec;
        if constexpr (V == 4 && !D) {
            if (nt) { hipLaunchKernelGGL((bn_bwd_reduce_kernel<4, E, false, true>), dim3(g.nblk), dim3(g.threads), lds, st, rows, c, g.cv, g.rpb, g.rows_per_block, x, dy,
                           gamma, beta, mean, invstd, relu, partial, lddy, drop); return; }
        }
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<V, E, D, false>), dim3(g.nblk), dim3(g.threads), lds, st, rows, c, g.cv, g.rpb, g.rows_per_block, x, dy,
                           gamma, beta, mean, invstd, relu, partial, lddy, drop);
    }
};
static void launch_bn_bwd_reduce(const BnGeom &g, hipStream_t st, long long rows, int c, const float *x, const float *dy, const float *gamma,
                                 const float *beta, const float *mean, const float *invstd, int relu, float *partial, long long lddy,
                                 int nt, BnDrop drop = kNoDrop)
{
    bn_dispatch<BnReduceL>(g, (relu & kBnEluIn) != 0, drop.seed != nullptr, g, st, rows, c, x, dy, gamma, beta, mean, invstd, relu, partial, lddy, drop, nt);
}

template <int V, bool E, bool D> struct BnDxL {
    static void go(const BnGeom &g, hipStream_t st, long long rows, int c, const float *x, const float *dy, const float *gamma,
                   const float *beta, const float *mean, const float *invstd, const float *dgamma, const float *dbeta, int relu, float *dx,
                   float *colsum_partial, long long lddy, BnDrop drop, int nt)
    {
        const size_t lds = sizeof(float) * static_cast<size_t>(g.threads) * 2 * g.vec;
        if constexpr (V == 4 && !D) {
            if (nt) { hipLaunchKernelGGL((bn_bwd_dx_kernel<4, E, false, true>), dim3(g.nblk), dim3(g.threads), lds, st, rows, c, g.cv, g.rpb, g.rows_per_block, x, dy,
                           gamma, beta, mean, invstd, dgamma, dbeta, relu, dx, colsum_partial, lddy, drop); return; }
        }
        hipLaunchKernelGGL((bn_bwd_dx_kernel<V, E, D, false>), dim3(g.nblk), dim3(g.threads), lds, st, rows, c, g.cv, g.rpb, g.rows_per_block, x, dy,
                           gamma, beta, mean, invstd, dgamma, dbeta, relu, dx, colsum_partial, lddy, drop);
    }
};
static void launch_bn_bwd_dx(const BnGeom &g, hipStream_t st, long long rows, int c, const float *x, const float *dy, const float *gamma,
                             const float *beta, const float *mean, const float *invstd, const float *dgamma, const float *dbeta, int relu,
                             float *dx, float *colsum_partial, long long lddy, int nt, BnDrop drop = kNoDrop)
{
    bn_dispatch<BnDxL>(g, (relu & kBnEluIn) != 0, drop.seed != nullptr, g, st, rows, c, x, dy, gamma, beta, mean, invstd, dgamma, dbeta, relu, dx,
                       colsum_partial, lddy, drop, nt);
}

// streaming loads for the passes whose tensors cannot stay in the Infinity Cache until they are read again (see ldv)
constexpr long long kBnNtBytes = 200ll << 20;
static int bn_nt_fwd(long long rows, int c) { return static_cast<long long>(sizeof(float)) * rows * c > kBnNtBytes ? 1 : 0; }
static int bn_nt_bwd(long long rows, int c) { return 2ll * static_cast<long long>(sizeof(float)) * rows * c > kBnNtBytes ? 1 : 0; }

static bool aligned16(const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

}  // namespace hf

using namespace hf;

HF_API size_t hf_bn_workspace(long long rows, int c)
{
    if (rows <= 0 || c <= 0) return 0;
    return sizeof(float) * 2 * static_cast<size_t>(c) * kBnMaxBlocks;
}

// vec = 4 needs 16-byte rows on every strided operand too
static bool ld_ok4(long long ld) { return ld % 4 == 0; }
// the streaming kernels address a block's rows with 32-bit element offsets (rows of a block + one unrolled step, times the row stride)
static bool bn_offsets_fit(const BnGeom &g, long long ld)
{
    return (g.rows_per_block + static_cast<long long>(kBnUnroll + 1) * g.rpb) * ld < (1ll << 31);
}

HF_API int hf_bn_relu_fwd_train_ld(long long rows, int c, const float *x, const float *gamma, const float *beta, float eps,
                                   float momentum, float *running_mean, float *running_var, int relu, float *y, long long ldy,
                                   float *save_mean, float *save_invstd, void *workspace, size_t workspace_bytes,
                                   hf_stream_t stream)
{
    if (rows <= 0 || c <= 0 || c > 4096 || ldy < c || !x || !gamma || !beta || !y || !save_mean || !save_invstd) return HF_EINVAL;
    if (!workspace || workspace_bytes < hf_bn_workspace(rows, c)) return HF_EWORKSPACE;
    BnGeom g = bn_geom(rows, c);
    if (g.vec == 4 && !(aligned16(x) && aligned16(y) && ld_ok4(ldy))) return HF_EINVAL;
    if (!bn_offsets_fit(g, ldy)) return HF_EINVAL;
    hipStream_t st = as_stream(stream);
    BnSmallGeom sg;
    if (bn_small_geom(rows, c, sg)) {
        hipLaunchKernelGGL(bn_small_fwd_kernel, dim3(sg.nwg), dim3(sg.threads), 0, st, static_cast<int>(rows), c, sg.cvw, sg.rpb, x,
                           gamma, beta, eps, momentum, running_mean, running_var, relu, y, ldy, save_mean, save_invstd);
        return launch_status();
    }
    float *partial = static_cast<float *>(workspace);
    launch_bn_stats(g, st, rows, c, x, partial, (relu & kBnEluIn) != 0, bn_nt_fwd(rows, c));
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(bn_finalize_grid(c, g.nblk)), dim3(64 * kBnFinalizeChannels), 0, st, rows, c, g.nblk, partial, eps,
                       momentum, running_mean, running_var, save_mean, save_invstd, static_cast<unsigned long long *>(nullptr), static_cast<unsigned long long *>(nullptr), 0ull);
    launch_bn_apply(g, st, rows, c, x, gamma, beta, save_mean, save_invstd, relu, y, ldy, bn_nt_fwd(rows, c));
    return launch_status();
}

HF_API int hf_bn_relu_fwd_train(long long rows, int c, const float *x, const float *gamma, const float *beta, float eps,
                                float momentum, float *running_mean, float *running_var, int relu, float *y,
                                float *save_mean, float *save_invstd, void *workspace, size_t workspace_bytes,
                                hf_stream_t stream)
{
    return hf_bn_relu_fwd_train_ld(rows, c, x, gamma, beta, eps, momentum, running_mean, running_var, relu, y, c, save_mean,
                                   save_invstd, workspace, workspace_bytes, stream);
}

HF_API int hf_bn_stats(long long rows, int c, const float *x, float eps, float momentum, float *running_mean,
                       float *running_var, float *save_mean, float *save_invstd, void *workspace, size_t workspace_bytes,
                       hf_stream_t stream)
{
    if (rows <= 0 || c <= 0 || c > 4096 || !x || !save_mean || !save_invstd) return HF_EINVAL;
    if (!workspace || workspace_bytes < hf_bn_workspace(rows, c)) return HF_EWORKSPACE;
    BnGeom g = bn_geom(rows, c);
    if (g.vec == 4 && !aligned16(x)) return HF_EINVAL;
    hipStream_t st = as_stream(stream);
    float *partial = static_cast<float *>(workspace);
    launch_bn_stats(g, st, rows, c, x, partial, false, bn_nt_fwd(rows, c));
    launch_bn_stats_finalize(rows, c, g.nblk, partial, eps, momentum, running_mean, running_var, save_mean, save_invstd, st);
    return launch_status();
}

HF_API int hf_bn_relu_fwd_eval(long long rows, int c, const float *x, const float *gamma, const float *beta,
                               const float *mean, const float *invstd, int relu, float *y, hf_stream_t stream)
{
    if (rows <= 0 || c <= 0 || c > 4096 || !x || !gamma || !beta || !mean || !invstd || !y) return HF_EINVAL;
    BnGeom g = bn_geom(rows, c);
    if (g.vec == 4 && !(aligned16(x) && aligned16(y))) return HF_EINVAL;
    hipStream_t st = as_stream(stream);
    launch_bn_apply(g, st, rows, c, x, gamma, beta, mean, invstd, relu, y, static_cast<long long>(c), bn_nt_fwd(rows, c));
    return launch_status();
}

HF_API int hf_bn_relu_bwd_ld(long long rows, int c, const float *x, const float *dy, long long lddy, const float *gamma,
                             const float *beta, const float *save_mean, const float *save_invstd, int relu, float *dx,
                             float *dgamma, float *dbeta, float *dx_colsum, void *workspace, size_t workspace_bytes,
                             hf_stream_t stream)
{
    if (rows <= 0 || c <= 0 || c > 4096 || lddy < c || !x || !dy || !gamma || !beta || !save_mean || !save_invstd || !dx || !dgamma ||
        !dbeta)
        return HF_EINVAL;
    if (!workspace || workspace_bytes < hf_bn_workspace(rows, c)) return HF_EWORKSPACE;
    BnGeom g = bn_geom(rows, c);
    if (g.vec == 4 && !(aligned16(x) && aligned16(dy) && aligned16(dx) && ld_ok4(lddy))) return HF_EINVAL;
    if (!bn_offsets_fit(g, lddy)) return HF_EINVAL;
    hipStream_t st = as_stream(stream);
    BnSmallGeom sg;
    if (bn_small_geom(rows, c, sg)) {
        hipLaunchKernelGGL(bn_small_bwd_kernel, dim3(sg.nwg), dim3(sg.threads), 0, st, static_cast<int>(rows), c, sg.cvw, sg.rpb, x, dy,
                           lddy, gamma, beta, save_mean, save_invstd, relu, dx, dgamma, dbeta, dx_colsum);
        return launch_status();
    }
    float *partial = static_cast<float *>(workspace);
    launch_bn_bwd_reduce(g, st, rows, c, x, dy, gamma, beta, save_mean, save_invstd, relu, partial, lddy, bn_nt_bwd(rows, c));
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(bn_finalize_grid(c, g.nblk)), dim3(64 * kBnFinalizeChannels), 0, st, c, g.nblk, partial, dgamma, dbeta);
    float *cpart = dx_colsum ? partial : nullptr;  // the reduce partials were consumed by the finalize kernel above
    launch_bn_bwd_dx(g, st, rows, c, x, dy, gamma, beta, save_mean, save_invstd, dgamma, dbeta, relu, dx, cpart, lddy, bn_nt_bwd(rows, c));
    if (dx_colsum)
        hipLaunchKernelGGL(bn_colsum_finalize_kernel, dim3(bn_finalize_grid(c, g.nblk)), dim3(64 * kBnFinalizeChannels), 0, st, c, g.nblk, cpart, dx_colsum);
    return launch_status();
}

HF_API int hf_bn_relu_bwd(long long rows, int c, const float *x, const float *dy, const float *gamma, const float *beta,
                          const float *save_mean, const float *save_invstd, int relu, float *dx, float *dgamma,
                          float *dbeta, float *dx_colsum, void *workspace, size_t workspace_bytes, hf_stream_t stream)
{
    return hf_bn_relu_bwd_ld(rows, c, x, dy, c, gamma, beta, save_mean, save_invstd, relu, dx, dgamma, dbeta, dx_colsum, workspace,
                             workspace_bytes, stream);
}

static bool drop_args(float rate, BnDrop &d, const unsigned long long *seed)
{
    if (!(rate >= 0.0f) || !(rate < 1.0f) || !seed) return false;
    d.seed = seed;
    d.thresh = static_cast<unsigned>(rate * 65536.0f + 0.5f);
    if (d.thresh > 65535u) d.thresh = 65535u;
    d.scale = 1.0f / (1.0f - rate);
    return true;
}

HF_API int hf_bn_dropout_fwd_train(long long rows, int c, const float *x, const float *gamma, const float *beta, float eps, float momentum,
                                   float *running_mean, float *running_var, int relu, float rate, unsigned long long salt,
                                   unsigned long long *drop_state, unsigned long long *seed_out, float *y, float *save_mean,
                                   float *save_invstd, void *workspace, size_t workspace_bytes, hf_stream_t stream)
{
    if (rows <= 0 || c <= 0 || c > 4096 || !x || !gamma || !beta || !y || !save_mean || !save_invstd || !drop_state || !seed_out) return HF_EINVAL;
    BnDrop d;
    if (!drop_args(rate, d, seed_out)) return HF_EINVAL;
    if (!workspace || workspace_bytes < hf_bn_workspace(rows, c)) return HF_EWORKSPACE;
    BnGeom g = bn_geom(rows, c);
    if (g.vec == 4 && !(aligned16(x) && aligned16(y))) return HF_EINVAL;
    if (!bn_offsets_fit(g, c)) return HF_EINVAL;
    hipStream_t st = as_stream(stream);
    float *partial = static_cast<float *>(workspace);
    launch_bn_stats(g, st, rows, c, x, partial, (relu & kBnEluIn) != 0, bn_nt_fwd(rows, c));
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(bn_finalize_grid(c, g.nblk)), dim3(64 * kBnFinalizeChannels), 0, st, rows, c, g.nblk, partial, eps,
                       momentum, running_mean, running_var, save_mean, save_invstd, drop_state, seed_out, salt);
    launch_bn_apply(g, st, rows, c, x, gamma, beta, save_mean, save_invstd, relu, y, static_cast<long long>(c), bn_nt_fwd(rows, c), d);
    return launch_status();
}

HF_API int hf_bn_dropout_bwd(long long rows, int c, const float *x, const float *dy, const float *gamma, const float *beta,
                             const float *save_mean, const float *save_invstd, int relu, float rate, const unsigned long long *seed,
                             float *dx, float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes, hf_stream_t stream)
{
    if (rows <= 0 || c <= 0 || c > 4096 || !x || !dy || !gamma || !beta || !save_mean || !save_invstd || !dx || !dgamma || !dbeta) return HF_EINVAL;
    BnDrop d;
    if (!drop_args(rate, d, seed)) return HF_EINVAL;
    if (!workspace || workspace_bytes < hf_bn_workspace(rows, c)) return HF_EWORKSPACE;
    BnGeom g = bn_geom(rows, c);
    if (g.vec == 4 && !(aligned16(x) && aligned16(dy) && aligned16(dx))) return HF_EINVAL;
    if (!bn_offsets_fit(g, c)) return HF_EINVAL;
    hipStream_t st = as_stream(stream);
    float *partial = static_cast<float *>(workspace);
    launch_bn_bwd_reduce(g, st, rows, c, x, dy, gamma, beta, save_mean, save_invstd, relu, partial, static_cast<long long>(c), bn_nt_bwd(rows, c), d);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(bn_finalize_grid(c, g.nblk)), dim3(64 * kBnFinalizeChannels), 0, st, c, g.nblk, partial, dgamma, dbeta);
    launch_bn_bwd_dx(g, st, rows, c, x, dy, gamma, beta, save_mean, save_invstd, dgamma, dbeta, relu, dx, nullptr, static_cast<long long>(c), bn_nt_bwd(rows, c), d);
    return launch_status();
}

HF_API int hf_narrow_linear_dx(long long rows, int cin, int cout, const float *g, const float *w, float *dx, hf_stream_t stream)
{
    if (rows <= 0 || cin <= 0 || cin > 4096 || cout <= 0 || cout > kNarrowMaxOut || !g || !w || !dx) return HF_EINVAL;
    BnGeom gm = bn_geom(rows, cin);
    if (gm.vec == 4 && !aligned16(dx)) return HF_EINVAL;
    hipStream_t st = as_stream(stream);
    if (gm.vec == 4)
        hipLaunchKernelGGL((narrow_linear_dx_kernel<4>), dim3(gm.nblk), dim3(gm.threads), 0, st, rows, cin, cout, gm.cv, gm.rpb, gm.rows_per_block, g, w, dx);
    else
        hipLaunchKernelGGL((narrow_linear_dx_kernel<1>), dim3(gm.nblk), dim3(gm.threads), 0, st, rows, cin, cout, gm.cv, gm.rpb, gm.rows_per_block, g, w, dx);
    return launch_status();
}

HF_API int hf_bn_relu_bwd_dx(long long rows, int c, const float *x, const float *dy, const float *gamma, const float *beta,
                             const float *save_mean, const float *save_invstd, const float *dgamma, const float *dbeta,
                             int relu, float *dx, hf_stream_t stream)
{
    if (rows <= 0 || c <= 0 || c > 4096 || !x || !dy || !gamma || !beta || !save_mean || !save_invstd || !dx || !dgamma ||
        !dbeta)
        return HF_EINVAL;
    BnGeom g = bn_geom(rows, c);
    if (g.vec == 4 && !(aligned16(x) && aligned16(dy) && aligned16(dx))) return HF_EINVAL;
    hipStream_t st = as_stream(stream);
    launch_bn_bwd_dx(g, st, rows, c, x, dy, gamma, beta, save_mean, save_invstd, dgamma, dbeta, relu, dx, nullptr, static_cast<long long>(c), bn_nt_bwd(rows, c));
    return launch_status();
}

HF_API int hf_bn_relu_maxpool_fwd(long long groups, int k, int c, const float *z, const float *gamma, const float *beta,
                                  int training, float eps, float momentum, float *running_mean, float *running_var,
                                  float *mean, float *invstd, float *pooled, unsigned char *argmax, void *workspace,
                                  size_t workspace_bytes, hf_stream_t stream)
{
    if (groups <= 0 || k <= 0 || k > 255 || c <= 0 || c > 4096 || !z || !gamma || !beta || !mean || !invstd || !pooled)
        return HF_EINVAL;
    const long long rows = groups * k;
    hipStream_t st = as_stream(stream);
    if (training) {
        if (!argmax) return HF_EINVAL;
        if (!workspace || workspace_bytes < hf_bn_workspace(rows, c)) return HF_EWORKSPACE;
        BnGeom g = bn_geom(rows, c);
        if (g.vec == 4 && !aligned16(z)) return HF_EINVAL;
        float *partial = static_cast<float *>(workspace);
        launch_bn_stats(g, st, rows, c, z, partial, false, bn_nt_fwd(rows, c));
        hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(bn_finalize_grid(c, g.nblk)), dim3(64 * kBnFinalizeChannels), 0, st, rows, c, g.nblk, partial, eps, momentum,
                           running_mean, running_var, mean, invstd, static_cast<unsigned long long *>(nullptr), static_cast<unsigned long long *>(nullptr), 0ull);
    }
    BnGeom gg = bn_geom(groups, c);
    if (gg.vec == 4 && !aligned16(z)) return HF_EINVAL;
    if (gg.vec == 4)
        hipLaunchKernelGGL((bn_pool_fwd_kernel<4>), dim3(gg.nblk), dim3(gg.threads), 0, st, groups, k, c, gg.cv, gg.rpb,
                           gg.rows_per_block, z, gamma, beta, mean, invstd, pooled, argmax);
    else
        hipLaunchKernelGGL((bn_pool_fwd_kernel<1>), dim3(gg.nblk), dim3(gg.threads), 0, st, groups, k, c, gg.cv, gg.rpb,
                           gg.rows_per_block, z, gamma, beta, mean, invstd, pooled, argmax);
    return launch_status();
}

HF_API int hf_bn_relu_maxpool_bwd(long long groups, int k, int c, const float *z, const float *dpooled,
                                  const unsigned char *argmax, const float *gamma, const float *beta,
                                  const float *save_mean, const float *save_invstd, float *dz, float *dgamma,
                                  float *dbeta, float *dz_colsum, void *workspace, size_t workspace_bytes,
                                  hf_stream_t stream)
{
    if (groups <= 0 || k <= 0 || k > 255 || c <= 0 || c > 4096 || !z || !dpooled || !argmax || !gamma || !beta ||
        !save_mean || !save_invstd || !dz || !dgamma || !dbeta)
        return HF_EINVAL;
    const long long rows = groups * k;
    if (!workspace || workspace_bytes < hf_bn_workspace(rows, c)) return HF_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    float *partial = static_cast<float *>(workspace);
    BnGeom gg = bn_geom(groups, c);
    BnGeom g = bn_geom(rows, c);
    if (g.vec == 4 && !(aligned16(z) && aligned16(dz) && aligned16(dpooled))) return HF_EINVAL;
    const size_t lds_g = sizeof(float) * static_cast<size_t>(gg.threads) * 2 * gg.vec;
    const size_t lds_r = sizeof(float) * static_cast<size_t>(g.threads) * 2 * g.vec;
    if (gg.vec == 4)
        hipLaunchKernelGGL((bn_pool_bwd_reduce_kernel<4>), dim3(gg.nblk), dim3(gg.threads), lds_g, st, groups, k, c, gg.cv,
                           gg.rpb, gg.rows_per_block, z, dpooled, argmax, gamma, beta, save_mean, save_invstd, partial);
    else
        hipLaunchKernelGGL((bn_pool_bwd_reduce_kernel<1>), dim3(gg.nblk), dim3(gg.threads), lds_g, st, groups, k, c, gg.cv,
                           gg.rpb, gg.rows_per_block, z, dpooled, argmax, gamma, beta, save_mean, save_invstd, partial);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(bn_finalize_grid(c, gg.nblk)), dim3(64 * kBnFinalizeChannels), 0, st, c, gg.nblk, partial, dgamma, dbeta);
    float *cpart = dz_colsum ? partial : nullptr;
    if (g.vec == 4)
        hipLaunchKernelGGL((bn_pool_bwd_dx_kernel<4>), dim3(g.nblk), dim3(g.threads), lds_r, st, rows, k, c, g.cv, g.rpb,
                           g.rows_per_block, z, dpooled, argmax, gamma, beta, save_mean, save_invstd, dgamma, dbeta, dz,
                           cpart);
    else
        hipLaunchKernelGGL((bn_pool_bwd_dx_kernel<1>), dim3(g.nblk), dim3(g.threads), lds_r, st, rows, k, c, g.cv, g.rpb,
                           g.rows_per_block, z, dpooled, argmax, gamma, beta, save_mean, save_invstd, dgamma, dbeta, dz,
                           cpart);
    if (dz_colsum) hipLaunchKernelGGL(bn_colsum_finalize_kernel, dim3(bn_finalize_grid(c, g.nblk)), dim3(64 * kBnFinalizeChannels), 0, st, c, g.nblk, cpart, dz_colsum);
    return launch_status();
}
