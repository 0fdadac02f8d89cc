// optim.hip -- the Adam update of ALL parameter tensors in one launch, and the refresh of a captured step's input slots in one
// launch (gfx950).
//
// The train step of hf/core/trainer.py:71 ends in tf.train.AdamOptimizer.apply_gradients (hf/builders/optimizer_builder.py:59-64).
// As framework calls that was seven multi-tensor launches per step (the tensor list travels in kernel arguments, 4 KB at a time)
// at 1.2 TB/s -- 0.32 ms of a 9 ms step at one frame per GPU.  Here the tensor list lives in device memory: one table entry per
// parameter tensor (parameter, gradient, first and second moment, length) and one map entry per 16 K-element chunk; a workgroup
// looks up its chunk and streams it with 16-byte accesses where the four pointers allow.  The gradients are read where autograd
// left them (no gather copy); with several ranks the averaging factor 1 / world rides along as `grad_scale`.
#include <math.h>

#include "hf_common.h"

namespace hf {

constexpr int kAdamChunk = 16384;     // elements per workgroup
constexpr int kAdamThreads = 256;

struct AdamEntry {                    // == hf_adam_entry (hfops.h)
    float *p;
    const float *g;
    float *m;
    float *v;
    long long n;
};
static_assert(sizeof(AdamEntry) == sizeof(hf_adam_entry), "table layout");

// mode 0: TensorFlow's form  p -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)           (tf.train.AdamOptimizer)
// mode 1: torch.optim.Adam's  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ __launch_bounds__(kAdamThreads) void adam_multi_kernel(const AdamEntry *__restrict__ table, const int2 *__restrict__ map,
                                                                  const float *__restrict__ step, float lr, float b1, float b2,
                                                                  float eps, float grad_scale, int mode)
{
    const int2 where = map[blockIdx.x];                 // (tensor, chunk)
    const AdamEntry e = table[where.x];
    const long long o0 = static_cast<long long>(where.y) * kAdamChunk;
    const long long n = e.n - o0 < kAdamChunk ? e.n - o0 : kAdamChunk;
    const float t = *step;                              // this step's number (1, 2, ...): the caller advanced it
    const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
    const float sq2 = sqrtf(bc2);
    const float a = mode == 0 ? lr * sq2 / bc1 : lr / bc1;
    const float inv_sq2 = 1.0f / sq2;
    // the four pointers come out of a table in memory: tell the compiler they are GLOBAL addresses (else it emits FLAT accesses)
    typedef __attribute__((address_space(1))) float gfloat;
    gfloat *p = (gfloat *)(e.p + o0);
    const gfloat *g = (const gfloat *)(e.g + o0);
    gfloat *m = (gfloat *)(e.m + o0);
    gfloat *v = (gfloat *)(e.v + o0);
    auto upd = [&](auto &pp, float gg, auto &mm, auto &vv) {
        gg *= grad_scale;
        mm = b1 * mm + (1.0f - b1) * gg;
        vv = b2 * vv + (1.0f - b2) * gg * gg;
        const float d = mode == 0 ? sqrtf(vv) + eps : sqrtf(vv) * inv_sq2 + eps;
        pp -= a * mm / d;
    };
    const bool al = ((reinterpret_cast<uintptr_t>(e.p + o0) | reinterpret_cast<uintptr_t>(e.g + o0) | reinterpret_cast<uintptr_t>(e.m + o0) |
                      reinterpret_cast<uintptr_t>(e.v + o0)) & 15) == 0;
    if (al) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const long long n4 = n >> 2;
        for (long long i = threadIdx.x; i < n4; i += kAdamThreads) {
            typedef __attribute__((address_space(1))) f4 gf4;
            f4 pv = ((gf4 *)p)[i], mv = ((gf4 *)m)[i], vv = ((gf4 *)v)[i];
            const f4 gv = ((const gf4 *)g)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) { float a_ = pv[k], m_ = mv[k], v_ = vv[k]; upd(a_, gv[k], m_, v_); pv[k] = a_; mv[k] = m_; vv[k] = v_; }
            ((gf4 *)p)[i] = pv;
            ((gf4 *)m)[i] = mv;
            ((gf4 *)v)[i] = vv;
        }
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += kAdamThreads) upd(p[i], g[i], m[i], v[i]);
    } else {
        for (long long i = threadIdx.x; i < n; i += kAdamThreads) upd(p[i], g[i], m[i], v[i]);
    }
}

// ---- several device-to-device copies in one launch (the refresh of a captured step's input slots) ----
constexpr int kCopyMax = 64;            // 3 x 8 x 64 + 4 x 65 = 1.8 KB of kernel arguments
constexpr int kCopyChunk = 64 * 1024;   // bytes per workgroup
constexpr int kCopyThreads = 256;

struct CopyList {
    void *dst[kCopyMax];
    const void *src[kCopyMax];
    long long bytes[kCopyMax];
    int first[kCopyMax + 1];            // first workgroup of every copy (prefix sums of their chunk counts); first[n .. kCopyMax] = all
};

// One workgroup per 64 KB chunk of SOME copy: a flat grid over all chunks, the copy found by a binary search of the prefix table
// (scalar loads of kernel arguments).  The first form launched (chunks of the longest copy) x (copies) workgroups -- 844 x 50 for a
// step whose inputs hold one 55 MB feature map and fifty small tables: 42 000 workgroups, all but ~1000 of them leaving at once,
// 97 us for 60 MB.
__global__ __launch_bounds__(kCopyThreads) void copy_multi_kernel(CopyList list)
{
    int lo = 0, hi = kCopyMax - 1;      // largest `which` with first[which] <= blockIdx.x (first[0] = 0)
#pragma unroll
    for (int step = 0; step < 6; ++step) {   // 2^6 = kCopyMax
        const int mid = (lo + hi + 1) >> 1;
        if (list.first[mid] <= static_cast<int>(blockIdx.x)) lo = mid; else hi = mid - 1;
    }
    const int which = lo;
    const long long total = list.bytes[which];
    const long long o0 = static_cast<long long>(static_cast<int>(blockIdx.x) - list.first[which]) * kCopyChunk;
    if (o0 >= total) return;
    const long long n = total - o0 < kCopyChunk ? total - o0 : kCopyChunk;
    typedef __attribute__((address_space(1))) unsigned char gbyte;
    gbyte *d = (gbyte *)(static_cast<unsigned char *>(list.dst[which]) + o0);
    const gbyte *s = (const gbyte *)(static_cast<const unsigned char *>(list.src[which]) + o0);
    if (((reinterpret_cast<uintptr_t>(list.dst[which]) | reinterpret_cast<uintptr_t>(list.src[which])) & 15) == 0) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(1))) u4 gu4;
        const long long n16 = n >> 4;
        for (long long i = threadIdx.x; i < n16; i += kCopyThreads) ((gu4 *)d)[i] = ((const gu4 *)s)[i];
        for (long long i = (n16 << 4) + threadIdx.x; i < n; i += kCopyThreads) d[i] = s[i];
    } else {
        for (long long i = threadIdx.x; i < n; i += kCopyThreads) d[i] = s[i];
    }
}

}  // namespace hf

using namespace hf;

HF_API int hf_copy_multi_max(void) { return kCopyMax; }

HF_API int hf_copy_multi(int n, void *const *dst, const void *const *src, const long long *bytes, hf_stream_t stream)
{
    if (n < 0 || n > kCopyMax) return HF_EINVAL;
    if (n == 0) return HF_OK;
    if (!dst || !src || !bytes) return HF_EINVAL;
    static_assert(kCopyMax == 64, "the kernel's binary search takes six steps");
    CopyList list;
    long long chunks = 0;
    for (int i = 0; i < n; ++i) {
        if (bytes[i] < 0 || (bytes[i] > 0 && (!dst[i] || !src[i]))) return HF_EINVAL;
        list.dst[i] = dst[i];
        list.src[i] = src[i];
        list.bytes[i] = bytes[i];
        list.first[i] = static_cast<int>(chunks);
        chunks += (bytes[i] + kCopyChunk - 1) / kCopyChunk;
        if (chunks > 0x7fffffffLL) return HF_EINVAL;
    }
    for (int i = n; i <= kCopyMax; ++i) list.first[i] = static_cast<int>(chunks);
    for (int i = n; i < kCopyMax; ++i) { list.dst[i] = nullptr; list.src[i] = nullptr; list.bytes[i] = 0; }
    if (chunks == 0) return HF_OK;
    hipLaunchKernelGGL(copy_multi_kernel, dim3(static_cast<unsigned>(chunks)), dim3(kCopyThreads), 0, as_stream(stream), list);
    return launch_status();
}

HF_API int hf_adam_chunk(void) { return kAdamChunk; }

HF_API int hf_adam_multi(int num_chunks, const hf_adam_entry *table, const int *chunk_map, const float *step, float lr, float beta1,
                         float beta2, float eps, float grad_scale, int mode, hf_stream_t stream)
{
    if (num_chunks < 0 || (mode != 0 && mode != 1) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return HF_EINVAL;
    if (num_chunks == 0) return HF_OK;
    if (!table || !chunk_map || !step) return HF_EINVAL;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(num_chunks), dim3(kAdamThreads), 0, as_stream(stream),
                       reinterpret_cast<const AdamEntry *>(table), reinterpret_cast<const int2 *>(chunk_map), step, lr, beta1, beta2, eps,
                       grad_scale, mode);
    return launch_status();
}
