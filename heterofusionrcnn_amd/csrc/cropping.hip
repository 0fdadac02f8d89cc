// cropping.hip -- pc_crop_and_sample (+ feature grad) for gfx950.
//
// Replaces cropping/tf_cropping_g.cu:7-168 and the six cudaMemsets of
// PcCropAndSampleGpuOp::Compute (cropping/tf_cropping.cpp:171-176).
//
// The reference appends inside points with atomicInc on a shared counter, so the order of a
// box's points is whatever order the atomics land in (tf_cropping_g.cu:85).  Here the order is
// defined: ascending point index -- what that kernel produces when launched with one thread.
//   phase 1  each of the block's waves scans a contiguous quarter of the cloud and compacts the
//            indices of inside points in order (ballot + prefix popcount, no atomics, no
//            barriers); concatenating the per-wave lists in wave order is ascending order;
//   phase 2  all threads copy rows for the `resize` output slots, slot s reading list entry
//            s (s < cnt) or (s - cnt) mod cnt (the reference's cyclic padding, :108-126);
//            empty boxes get zeros / non_empty = false.  Every output element is written exactly
//            once, so no memset passes are needed.
// Round 4: both phases keep several independent loads of a thread in flight.  The first form scanned 64 points per trip (three
// loads, wait, ballot: 64 dependent round trips per wave) and copied one 16-byte piece per thread and trip behind an integer
// division by the channel count -- 131 us for 512 boxes x 512 points x 288 channels = 2.3 TB/s of output with two 256-thread
// workgroups per CU.  Now: 1024 threads per box, four 64-point chunks (twelve loads) per trip of the scan, four pieces per trip of the
// copy with (slot, piece) advanced incrementally.
#include "hf_common.h"

namespace hf {

#ifndef HF_CROP_THREADS
#define HF_CROP_THREADS 1024   // 256 / 512 / 1024 threads per box: 107 / 88-93 / 82 us at 512 x 512 x 288 (scripts/probes/crop_timing.py, one run)
#endif
#ifndef HF_CROP_SCAN
#define HF_CROP_SCAN 4
#endif
#ifndef HF_CROP_COPY
#define HF_CROP_COPY 4
#endif
constexpr int kCropThreads = HF_CROP_THREADS;
constexpr int kCropWaves = kCropThreads / kWave;

// tf_cropping_g.cu:3-5
__device__ __forceinline__ float dot3(float x1, float y1, float z1, float x2, float y2, float z2)
{
    return x1 * x2 + y1 * y2 + z1 * z2;
}

struct CropBox {
    float ux, uy, uz, vx, vy, vz, wx, wy, wz;
    float u1, u2, v1, v4, w1, w5;
};

// the point-independent half of is_point_inside, tf_cropping_g.cu:11-34
__device__ __forceinline__ CropBox make_crop_box(const float *bb)
{
    const float p1x = bb[0], p1y = bb[8], p1z = bb[16];
    const float p2x = bb[1], p2y = bb[9], p2z = bb[17];
    const float p4x = bb[3], p4y = bb[11], p4z = bb[19];
    const float p5x = bb[4], p5y = bb[12], p5z = bb[20];
    CropBox c;
    c.ux = p2x - p1x; c.uy = p2y - p1y; c.uz = p2z - p1z;
    c.vx = p4x - p1x; c.vy = p4y - p1y; c.vz = p4z - p1z;
    c.wx = p5x - p1x; c.wy = p5y - p1y; c.wz = p5z - p1z;
    c.u1 = dot3(c.ux, c.uy, c.uz, p1x, p1y, p1z);
    c.u2 = dot3(c.ux, c.uy, c.uz, p2x, p2y, p2z);
    c.v1 = dot3(c.vx, c.vy, c.vz, p1x, p1y, p1z);
    c.v4 = dot3(c.vx, c.vy, c.vz, p4x, p4y, p4z);
    c.w1 = dot3(c.wx, c.wy, c.wz, p1x, p1y, p1z);
    c.w5 = dot3(c.wx, c.wy, c.wz, p5x, p5y, p5z);
    return c;
}

// tf_cropping_g.cu:24-40
__device__ __forceinline__ bool inside(const CropBox &c, float px, float py, float pz)
{
    const float ud = dot3(c.ux, c.uy, c.uz, px, py, pz);
    const float vd = dot3(c.vx, c.vy, c.vz, px, py, pz);
    const float wd = dot3(c.wx, c.wy, c.wz, px, py, pz);
    return c.u1 < ud && ud < c.u2 && c.v1 < vd && vd < c.v4 && c.w1 < wd && wd < c.w5;
}

__global__ __launch_bounds__(kCropThreads) void crop_kernel(
    const float *__restrict__ pts_data, const float *__restrict__ fts_data, const float *__restrict__ int_data,
    const unsigned char *__restrict__ mask_data, const float *__restrict__ boxes, const int *__restrict__ box_ind,
    int num_boxes, int npts, int resize, int channel, int ichannel, float *__restrict__ crop_pts,
    float *__restrict__ crop_fts, float *__restrict__ crop_int, unsigned char *__restrict__ crop_mask,
    int *__restrict__ crop_ind, unsigned char *__restrict__ non_empty)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int *lists = reinterpret_cast<int *>(smem_raw);        // kCropWaves * resize
    int *src = lists + kCropWaves * resize;                // resize: source point of every output slot
    __shared__ int wave_cnt[kCropWaves];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int bx = blockIdx.x; bx < num_boxes; bx += gridDim.x) {
        const CropBox cbx = make_crop_box(boxes + static_cast<size_t>(bx) * 24);
        const int bch = box_ind[bx];
        const float *P = pts_data + static_cast<size_t>(bch) * npts * 3;

        // ---- phase 1: ordered compaction, one contiguous range per wave ----
        const int per = (npts + kCropWaves - 1) / kCropWaves;
        const int lo = wave * per, hi = min(npts, lo + per);
        int cnt = 0;  // wave-uniform
        int *mine = lists + wave * resize;
        constexpr int kScan = HF_CROP_SCAN;   // 64-point chunks per trip: their loads go out together
        for (int p0 = lo; p0 < hi && cnt < resize; p0 += 64 * kScan) {
            float x[kScan], y[kScan], z[kScan];
#pragma unroll
            for (int u = 0; u < kScan; ++u) {
                const int p = min(p0 + 64 * u + lane, hi - 1);   // past the end: a valid address, masked below
                x[u] = P[p * 3]; y[u] = P[p * 3 + 1]; z[u] = P[p * 3 + 2];
            }
#pragma unroll
            for (int u = 0; u < kScan; ++u) {
                const int p = p0 + 64 * u + lane;
                const bool in = p < hi && inside(cbx, x[u], y[u], z[u]);
                const unsigned long long m = __ballot(in);
                const int pos = cnt + mask_prefix(m);
                if (in && pos < resize) mine[pos] = p;
                cnt += __builtin_popcountll(m);
            }
        }
        if (lane == 0) wave_cnt[wave] = min(cnt, resize);
        __syncthreads();

        int offs[kCropWaves + 1];
        offs[0] = 0;
#pragma unroll
        for (int w = 0; w < kCropWaves; ++w) offs[w + 1] = offs[w] + wave_cnt[w];
        const int total = min(offs[kCropWaves], resize);

        // source point per output slot (-1 = empty box)
        for (int s = t; s < resize; s += kCropThreads) {
            int e = -1;
            if (total > 0) {
                const int q = s < total ? s : (s - total) % total;
                int w = 0;
#pragma unroll
                for (int ww = 1; ww < kCropWaves; ++ww)
                    if (q >= offs[ww]) w = ww;
                e = lists[w * resize + (q - offs[w])];
            }
            src[s] = e;
        }
        __syncthreads();

        // ---- phase 2: row copies ----
        const size_t ob = static_cast<size_t>(bx) * resize;
        const float *F = fts_data + static_cast<size_t>(bch) * npts * channel;
        const float *I = int_data + static_cast<size_t>(bch) * npts * ichannel;
        const unsigned char *M = mask_data + static_cast<size_t>(bch) * npts;
        for (int s = t; s < resize; s += kCropThreads) {
            const int e = src[s];
            crop_ind[ob + s] = e < 0 ? 0 : e;
            crop_mask[ob + s] = e < 0 ? 0 : M[e];
        }
        for (int i = t; i < resize * 3; i += kCropThreads) {
            const int s = i / 3, d = i - s * 3;
            const int e = src[s];
            crop_pts[ob * 3 + i] = e < 0 ? 0.0f : P[e * 3 + d];
        }
        for (int i = t; i < resize * ichannel; i += kCropThreads) {
            const int s = i / ichannel, d = i - s * ichannel;
            const int e = src[s];
            crop_int[ob * ichannel + i] = e < 0 ? 0.0f : I[static_cast<size_t>(e) * ichannel + d];
        }
        if ((channel & 3) == 0) {
            const int cv = channel >> 2;
            const float4 *F4 = reinterpret_cast<const float4 *>(F);
            float4 *O4 = reinterpret_cast<float4 *>(crop_fts + ob * channel);
            // piece i = (slot i / cv, 16-byte piece i % cv); a thread's pieces are kCropThreads apart: (slot, piece) advance by
            // (kCropThreads / cv, kCropThreads % cv) with one carry -- no division inside the loop
            const int total4 = resize * cv, ds = kCropThreads / cv, dd = kCropThreads % cv;
            constexpr int kCopy = HF_CROP_COPY;
            int i = t, sl = t / cv, d = t - sl * cv;
            for (; i + (kCopy - 1) * kCropThreads < total4; i += kCopy * kCropThreads) {
                float4 v[kCopy];
#pragma unroll
                for (int u = 0; u < kCopy; ++u) {
                    const int e = src[sl];
                    v[u] = e < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : F4[static_cast<size_t>(e) * cv + d];
                    sl += ds; d += dd;
                    if (d >= cv) { d -= cv; ++sl; }
                }
#pragma unroll
                for (int u = 0; u < kCopy; ++u) O4[i + u * kCropThreads] = v[u];
            }
            for (; i < total4; i += kCropThreads) {
                const int e = src[sl];
                O4[i] = e < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : F4[static_cast<size_t>(e) * cv + d];
                sl += ds; d += dd;
                if (d >= cv) { d -= cv; ++sl; }
            }
        } else {
            for (int i = t; i < resize * channel; i += kCropThreads) {
                const int s = i / channel, d = i - s * channel;
                const int e = src[s];
                crop_fts[ob * channel + i] = e < 0 ? 0.0f : F[static_cast<size_t>(e) * channel + d];
            }
        }
        if (t == 0) non_empty[bx] = total > 0 ? 1 : 0;
        __syncthreads();  // lists / src reused by the next box
    }
}

// PcCropAndSampleGradFts, tf_cropping_g.cu:134-150; target zeroed by the caller
__global__ void crop_grad_fts_kernel(const int *__restrict__ box_ind, const int *__restrict__ crop_ind,
                                     const float *__restrict__ grad_crop_fts, long long total, int npts, int resize,
                                     int channel, float *__restrict__ grad_fts)
{
    for (long long e = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long row = e / channel;  // (box, slot)
        const int c = static_cast<int>(e - row * channel);
        const int bx = static_cast<int>(row / resize);
        const int k = crop_ind[row];
        atomicAdd(grad_fts + (static_cast<size_t>(box_ind[bx]) * npts + k) * channel + c, grad_crop_fts[e]);
    }
}

}  // namespace hf

using namespace hf;

HF_API int hf_pc_crop_and_sample(const float *pts, const float *fts, const float *intensities,
                                 const unsigned char *mask, const float *boxes, const int *box_ind, int num_boxes,
                                 int batch, int npts, int resize, int channel, int intensity_channel, float *crop_pts,
                                 float *crop_fts, float *crop_intensities, unsigned char *crop_mask, int *crop_ind,
                                 unsigned char *non_empty_box, hf_stream_t stream)
{
    // PcCropAndSampleGpuOp: resize > 0 (tf_cropping.cpp:108), P > 0, C > 0 (:121-123)
    if (resize <= 0 || batch <= 0 || npts <= 0 || channel <= 0 || intensity_channel <= 0 || num_boxes < 0 || !pts ||
        !fts || !intensities || !mask || !boxes || !box_ind || !crop_pts || !crop_fts || !crop_intensities ||
        !crop_mask || !crop_ind || !non_empty_box)
        return HF_EINVAL;
    if (num_boxes == 0) return HF_OK;
    const size_t lds = sizeof(int) * static_cast<size_t>(resize) * (kCropWaves + 1);
    if (lds > 150 * 1024) return HF_EINVAL;  // resize > ~7600: no config comes close (R = 512)
    const bool al16 = (reinterpret_cast<uintptr_t>(fts) % 16 == 0) && (reinterpret_cast<uintptr_t>(crop_fts) % 16 == 0);
    if ((channel & 3) == 0 && !al16) return HF_EINVAL;
    if (const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(&crop_kernel), lds); lrc != HF_OK) return lrc;
    const int grid = num_boxes < kNumCU * 8 ? num_boxes : kNumCU * 8;
    hipLaunchKernelGGL(crop_kernel, dim3(grid), dim3(kCropThreads), lds, as_stream(stream), pts, fts, intensities, mask,
                       boxes, box_ind, num_boxes, npts, resize, channel, intensity_channel, crop_pts, crop_fts,
                       crop_intensities, crop_mask, crop_ind, non_empty_box);
    return launch_status();
}

HF_API int hf_pc_crop_and_sample_grad_fts(const int *box_ind, const int *crop_ind, const float *grad_crop_fts,
                                          int num_boxes, int batch, int npts, int resize, int channel,
                                          float *grad_fts, hf_stream_t stream)
{
    if (num_boxes < 0 || batch <= 0 || npts <= 0 || resize <= 0 || channel <= 0 || !box_ind || !crop_ind ||
        !grad_crop_fts || !grad_fts)
        return HF_EINVAL;
    hipStream_t st = as_stream(stream);
    int rc = hip_status(hipMemsetAsync(grad_fts, 0, sizeof(float) * static_cast<size_t>(batch) * npts * channel, st));
    if (rc != HF_OK) return rc;
    const long long total = static_cast<long long>(num_boxes) * resize * channel;
    if (total == 0) return HF_OK;
    long long g = (total + 255) / 256;
    if (g > kNumCU * 8) g = kNumCU * 8;
    hipLaunchKernelGGL(crop_grad_fts_kernel, dim3(static_cast<unsigned>(g)), dim3(256), 0, st, box_ind, crop_ind,
                       grad_crop_fts, total, npts, resize, channel, grad_fts);
    return launch_status();
}
