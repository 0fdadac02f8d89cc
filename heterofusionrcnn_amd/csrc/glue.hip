// glue.hip -- the element-wise / gather steps either side of the custom ops for gfx950.
//
//   project_gather   the LiDAR -> image fusion step: hf/core/projection.py:5-32 (tf_rect_to_image) +
//                    hf/core/models/rpn_model.py:227-235 (int cast, tf.gather_nd of the image feature map).  One
//                    pass: the (B,P,2) pixel tensor, its int cast, the (B,P,3) index tensor and the gather become
//                    one kernel that reads 12 B of coordinates and C floats of features per point.
//   bin_box_decode / bin_box_encode   hf/core/bin_based_box3d_encoder.py:9-269: ~20 TensorFlow ops (tile, stack,
//                    2x2 matmul, mod, where, clip, floor ...) per call as one element-wise kernel each.
// Arithmetic: fp32, no contraction, expressions in the order the oracle (oracle/hf_oracle.c) writes them.
#include <math.h>

#include "hf_common.h"

namespace hf {

struct Pix { int u, v; };

__device__ __forceinline__ Pix project_point(const float *__restrict__ P, float x, float y, float z)
{
    const float uh = P[0] * x + P[1] * y + P[2] * z + P[3];
    const float vh = P[4] * x + P[5] * y + P[6] * z + P[7];
    const float d = P[8] * x + P[9] * y + P[10] * z + P[11];
    const float uf = uh / d, vf = vh / d;
    const bool ok = uf > -2147483648.0f && uf < 2147483648.0f && vf > -2147483648.0f && vf < 2147483648.0f;
    Pix r;
    r.u = ok ? static_cast<int>(uf) : -1;  // truncation toward zero, as tf.cast
    r.v = ok ? static_cast<int>(vf) : -1;
    return r;
}

// c/VEC lanes per point; every lane recomputes the 12-flop projection of its point
template <int VEC>
__global__ void project_gather_kernel(int p, int h, int w, int c, long long nrows, const float *__restrict__ pts,
                                      const float *__restrict__ calib, const float *__restrict__ img,
                                      float *__restrict__ out, int *__restrict__ pix)
{
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int cv = c / VEC;
    const long long total = nrows * cv;
    for (long long e = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long row = e / cv;
        const int l = static_cast<int>(e - row * cv);
        const long long bb = row / p;
        const float *x = pts + row * 3;
        const Pix q = project_point(calib + bb * 12, x[0], x[1], x[2]);
        if (pix && l == 0) { pix[row * 2] = q.u; pix[row * 2 + 1] = q.v; }
        vec_t v;
        if (q.u >= 0 && q.u < w && q.v >= 0 && q.v < h) {
            v = *reinterpret_cast<const vec_t *>(img + ((bb * h + q.v) * w + q.u) * c + l * VEC);
        } else {
            if constexpr (VEC == 1) v = 0.0f; else v = vec_t{ 0.f, 0.f, 0.f, 0.f };
        }
        *reinterpret_cast<vec_t *>(out + row * c + l * VEC) = v;
    }
}

__global__ void project_gather_grad_kernel(int p, int h, int w, int c, long long nrows,
                                           const float *__restrict__ grad_out, const int *__restrict__ pix,
                                           float *__restrict__ grad_img)
{
    const long long total = nrows * c;
    for (long long e = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long row = e / c;
        const int l = static_cast<int>(e - row * c);
        const long long bb = row / p;
        const int u = pix[row * 2], v = pix[row * 2 + 1];
        if (u < 0 || u >= w || v < 0 || v >= h) continue;
        atomicAdd(grad_img + ((bb * h + v) * w + u) * c + l, grad_out[e]);
    }
}

__global__ void bin_box_decode_kernel(long long rows, int k, const float *__restrict__ ref_pts,
                                      const float *__restrict__ ref_theta, const int *__restrict__ bin_x,
                                      const float *__restrict__ res_x_norm, const int *__restrict__ bin_z,
                                      const float *__restrict__ res_z_norm, const int *__restrict__ bin_theta,
                                      const float *__restrict__ res_theta_norm, const float *__restrict__ res_y,
                                      const float *__restrict__ res_size_norm, const float *__restrict__ mean_sizes,
                                      const float *__restrict__ ss, const float *__restrict__ deltas, float r,
                                      float delta_theta, float *__restrict__ boxes)
{
    const long long total = rows * k;
    for (long long e = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long i = e / k;
        const int j = static_cast<int>(e - i * k);
        const float th0 = ref_theta ? ref_theta[i] : 0.0f;
        float dx = (static_cast<float>(bin_x[e]) + 0.5f) * deltas[j] - ss[j] + res_x_norm[e] * deltas[j];
        float dz = (static_cast<float>(bin_z[e]) + 0.5f) * deltas[j] - ss[j] + res_z_norm[e] * deltas[j];
        if (ref_theta) {
            const float sn = sinf(th0), cs = cosf(th0);
            const float rx = cs * dx + sn * dz, rz = -sn * dx + cs * dz;
            dx = rx; dz = rz;
        }
        float *o = boxes + e * 7;
        o[0] = dx + ref_pts[i * 3 + 0];
        o[1] = res_y[e] + ref_pts[i * 3 + 1];
        o[2] = dz + ref_pts[i * 3 + 2];
#pragma unroll
        for (int d = 0; d < 3; ++d) o[3 + d] = mean_sizes[e * 3 + d] + res_size_norm[e * 3 + d] * mean_sizes[e * 3 + d];
        o[6] = th0 + (static_cast<float>(bin_theta[e]) + 0.5f) * delta_theta - r + res_theta_norm[e] * 0.5f * delta_theta;
    }
}

// The decoding block of the RPN / RCNN heads in one pass: slice the head vector (rpn_model.py:870-935), arg-max each
// logit slice (first maximum, as tf.argmax), pick the residual of the winning bin (:248-290), decode (tf_decode) with the
// per-class mean size, optionally keep only the row's predicted class (:237-246).  One thread per (row, class).
__global__ void bin_head_decode_kernel(long long rows, int k, int nbx, int nbz, int nbt, const float *__restrict__ head,
                                       const float *__restrict__ ref_pts, const float *__restrict__ ref_theta,
                                       const float *__restrict__ mean_sizes_k, const float *__restrict__ ss,
                                       const float *__restrict__ deltas, float r, float delta_theta,
                                       const int *__restrict__ cls, float *__restrict__ boxes)
{
    const int d = 2 * nbx + 2 * nbz + 2 * nbt + 4;
    const long long total = rows * k;
    for (long long e = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long i = e / k;
        const int j = static_cast<int>(e - i * k);
        if (cls && cls[i] != j) continue;
        const float *v = head + e * d;
        int bx = 0, bz = 0, bt = 0;
        float best = v[0];
        for (int q = 1; q < nbx; ++q) { const float x = v[q]; if (x > best) { best = x; bx = q; } }
        const float *vz = v + 2 * nbx;
        best = vz[0];
        for (int q = 1; q < nbz; ++q) { const float x = vz[q]; if (x > best) { best = x; bz = q; } }
        const float *vt = v + 2 * nbx + 2 * nbz;
        best = vt[0];
        for (int q = 1; q < nbt; ++q) { const float x = vt[q]; if (x > best) { best = x; bt = q; } }
        const float rx = v[nbx + bx], rz = vz[nbz + bz], rt = vt[nbt + bt];
        const float *tail = v + 2 * nbx + 2 * nbz + 2 * nbt;
        const float th0 = ref_theta ? ref_theta[i] : 0.0f;
        float dx = (static_cast<float>(bx) + 0.5f) * deltas[j] - ss[j] + rx * deltas[j];
        float dz = (static_cast<float>(bz) + 0.5f) * deltas[j] - ss[j] + rz * deltas[j];
        if (ref_theta) {
            const float sn = sinf(th0), cs = cosf(th0);
            const float ax = cs * dx + sn * dz, az = -sn * dx + cs * dz;
            dx = ax; dz = az;
        }
        float *o = cls ? boxes + i * 7 : boxes + e * 7;
        o[0] = dx + ref_pts[i * 3 + 0];
        o[1] = tail[0] + ref_pts[i * 3 + 1];
        o[2] = dz + ref_pts[i * 3 + 2];
#pragma unroll
        for (int c = 0; c < 3; ++c) o[3 + c] = mean_sizes_k[j * 3 + c] + tail[1 + c] * mean_sizes_k[j * 3 + c];
        o[6] = th0 + (static_cast<float>(bt) + 0.5f) * delta_theta - r + rt * 0.5f * delta_theta;
    }
}

__device__ __forceinline__ float floormodf(float x, float y)
{
    float r = fmodf(x, y);
    if (r != 0.0f && ((y < 0.0f) != (r < 0.0f))) r += y;
    return r;
}

__global__ void bin_box_encode_kernel(long long rows, int k, int rcnn, const float *__restrict__ ref_pts,
                                      const float *__restrict__ ref_theta, const float *__restrict__ boxes,
                                      const float *__restrict__ mean_sizes, const float *__restrict__ ss,
                                      const float *__restrict__ deltas, const float *__restrict__ hi_xz, float r,
                                      float hi_theta, float delta_theta, float half_delta_theta,
                                      int *__restrict__ bin_x, float *__restrict__ res_x_norm, int *__restrict__ bin_z,
                                      float *__restrict__ res_z_norm, int *__restrict__ bin_theta,
                                      float *__restrict__ res_theta_norm, float *__restrict__ res_y,
                                      float *__restrict__ res_size_norm)
{
    const float two_pi = static_cast<float>(2.0 * 3.141592653589793), pi = static_cast<float>(3.141592653589793);
    const float half_pi = static_cast<float>(0.5 * 3.141592653589793);
    const float three_half_pi = static_cast<float>(1.5 * 3.141592653589793);
    for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < rows;
         i += static_cast<long long>(gridDim.x) * blockDim.x) {
        const float *bx = boxes + i * 7;
        float dx = bx[0] - ref_pts[i * 3 + 0];
        const float dy = bx[1] - ref_pts[i * 3 + 1];
        float dz = bx[2] - ref_pts[i * 3 + 2];
        const float th0 = ref_theta ? ref_theta[i] : 0.0f;
        if (ref_theta) {
            const float a = th0 * -1.0f;
            const float sn = sinf(a), cs = cosf(a);
            const float rx = cs * dx + sn * dz, rz = -sn * dx + cs * dz;
            dx = rx; dz = rz;
        }
        float dshift;
        if (!rcnn) {
            const float dtheta = bx[6] - th0;
            dshift = fminf(fmaxf(dtheta + r, 0.0f), hi_theta);
        } else {
            float dtheta = bx[6] - floormodf(th0, two_pi);
            dtheta = floormodf(dtheta, two_pi);
            if (dtheta > half_pi && dtheta < three_half_pi) dtheta = floormodf(dtheta + pi, two_pi);
            dshift = floormodf(dtheta + half_pi, two_pi);
            dshift = fminf(fmaxf(dshift - r, 1e-3f), hi_theta);
        }
        for (int j = 0; j < k; ++j) {
            const long long e = i * k + j;
            const float xs = fminf(fmaxf(dx + ss[j], 0.0f), hi_xz[j]);
            const float fx = floorf(xs / deltas[j]);
            bin_x[e] = static_cast<int>(fx);
            res_x_norm[e] = (xs - (fx + 0.5f) * deltas[j]) / deltas[j];
            const float zs = fminf(fmaxf(dz + ss[j], 0.0f), hi_xz[j]);
            const float fz = floorf(zs / deltas[j]);
            bin_z[e] = static_cast<int>(fz);
            res_z_norm[e] = (zs - (fz + 0.5f) * deltas[j]) / deltas[j];
        }
        const float ft = floorf(dshift / delta_theta);
        bin_theta[i] = static_cast<int>(ft);
        res_theta_norm[i] = (dshift - (ft + 0.5f) * delta_theta) / half_delta_theta;
        res_y[i] = dy;
#pragma unroll
        for (int d = 0; d < 3; ++d) res_size_norm[i * 3 + d] = (bx[3 + d] - mean_sizes[i * 3 + d]) / mean_sizes[i * 3 + d];
    }
}


// 'concat' fusion with path drop (rpn_model.py:515-546): out[row] = [a[row] * m0 | b[row] * m1], one pass instead of two
// scaled copies and a concat.  masks = two floats on the device (the path-drop decision of this step), NULL = (1, 1).
template <int VEC>
__global__ void fuse_concat_kernel(long long rows, int c1, int c2, const float *__restrict__ a, const float *__restrict__ b,
                                   const float *__restrict__ masks, float *__restrict__ out)
{
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int cv = (c1 + c2) / VEC, c1v = c1 / VEC;
    const float m0 = masks ? masks[0] : 1.0f, m1 = masks ? masks[1] : 1.0f;
    const long long total = rows * cv;
    for (long long e = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long row = e / cv;
        const int l = static_cast<int>(e - row * cv);
        vec_t v;
        if (l < c1v) v = *reinterpret_cast<const vec_t *>(a + row * c1 + l * VEC) * m0;
        else v = *reinterpret_cast<const vec_t *>(b + row * c2 + (l - c1v) * VEC) * m1;
        *reinterpret_cast<vec_t *>(out + e * VEC) = v;
    }
}

template <int VEC>
__global__ void fuse_concat_grad_kernel(long long rows, int c1, int c2, const float *__restrict__ g,
                                        const float *__restrict__ masks, float *__restrict__ ga, float *__restrict__ gb)
{
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int cv = (c1 + c2) / VEC, c1v = c1 / VEC;
    const float m0 = masks ? masks[0] : 1.0f, m1 = masks ? masks[1] : 1.0f;
    const long long total = rows * cv;
    for (long long e = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long row = e / cv;
        const int l = static_cast<int>(e - row * cv);
        const vec_t v = *reinterpret_cast<const vec_t *>(g + e * VEC);
        if (l < c1v) { if (ga) *reinterpret_cast<vec_t *>(ga + row * c1 + l * VEC) = v * m0; }
        else if (gb) *reinterpret_cast<vec_t *>(gb + row * c2 + (l - c1v) * VEC) = v * m1;
    }
}


// ------------------------------------------------------------------------------------------
// The RPN loss (rpn_model.py:1040-1128 + hf/core/losses.py:131-226) in two passes instead of ~80 framework kernels:
//   segmentation   focal loss on the softmax of the K+1 logits: alpha (1 - p_t)^2 (-log p_t), p_t clipped to [1e-7, 1 - 1e-7],
//                  summed over all points, x seg_weight / rows;
//   foreground points (label > 0), on the head row of the labelled class: softmax cross-entropy of the x / z / theta bin
//                  logits (x cls_weight) and smooth-L1 of the residual of the TRUE bin for x / z / theta, of y and of the three
//                  sizes (x reg_weight), both / max(number of foreground points, 1).
// Targets come straight from hf_bin_box_encode (per class for x / z): the kernel picks the labelled class's entries
// (the per-class gathers of rpn_model.py:733-776).  Forward: per-block partial sums (seg, cls, reg, #fg).  Backward: gradients
// w.r.t. the segmentation logits and the head, scaled by the upstream gradient read from the device; grad_head is zero-filled by
// the caller's memset and only foreground rows are written.
// ------------------------------------------------------------------------------------------
constexpr int kLossMaxK1 = 8, kLossMaxBins = 32;
struct LossArgs {
    long long rows;
    int k, nbx, nbt;   // classes, x/z bins, theta bins; head row = [bx nbx | rx nbx | bz nbx | rz nbx | bt nbt | rt nbt | ry | size 3]
    const float *seg_logits, *head;
    const int *label;                       // 0 = background, 1..k
    const int *bin_x, *bin_z, *bin_t;       // (rows, k), (rows, k), (rows)
    const float *res_x, *res_z, *res_t, *res_y, *res_size;   // (rows, k), (rows, k), (rows), (rows), (rows, 3)
    float seg_w, cls_w, reg_w;
};

__device__ __forceinline__ float smooth_l1(float d) { const float a = fabsf(d); return a < 1.0f ? 0.5f * a * a : a - 0.5f; }
__device__ __forceinline__ float smooth_l1_grad(float d) { return fabsf(d) < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f); }

template <bool BWD>
__global__ __launch_bounds__(256) void rpn_loss_kernel(LossArgs a, float *__restrict__ partial, const float *__restrict__ nfg_total,
                                                      const float *__restrict__ upstream, float *__restrict__ grad_seg,
                                                      float *__restrict__ grad_head)
{
    __shared__ float red[4][256];
    const int d = 4 * a.nbx + 2 * a.nbt + 4;
    const int k1 = a.k + 1;
    float s_seg = 0.f, s_cls = 0.f, s_reg = 0.f, s_fg = 0.f;
    float gscale = 0.f, inv_den = 0.f;
    if (BWD) {
        gscale = upstream[0];
        inv_den = 1.0f / fmaxf(nfg_total[0], 1.0f);
    }
    const float seg_scale = a.seg_w / static_cast<float>(a.rows);
    for (long long r = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; r < a.rows;
         r += static_cast<long long>(gridDim.x) * blockDim.x) {
        const int lab = a.label[r];
        // ---- segmentation: softmax over k1 logits, focal term of the true class
        float lg[kLossMaxK1], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < kLossMaxK1; ++j) { lg[j] = j < k1 ? a.seg_logits[r * k1 + j] : -INFINITY; mx = fmaxf(mx, lg[j]); }
        float se = 0.f;
#pragma unroll
        for (int j = 0; j < kLossMaxK1; ++j) { lg[j] = j < k1 ? expf(lg[j] - mx) : 0.f; se += lg[j]; }
        const float inv = 1.0f / se;
        float pt_raw = 0.f;
#pragma unroll
        for (int j = 0; j < kLossMaxK1; ++j) { lg[j] *= inv; if (j == lab) pt_raw = lg[j]; }
        const float pt = fminf(fmaxf(pt_raw, 1e-7f), 1.0f - 1e-7f);
        const float om = 1.0f - pt;
        if (!BWD) {
            s_seg += 0.25f * om * om * (-logf(pt));
        } else {
            // d/dpt [alpha (1-pt)^2 (-log pt)] = alpha (2 (1-pt) log pt - (1-pt)^2 / pt); zero where the clip is active
            const float inside = (pt_raw >= 1e-7f && pt_raw <= 1.0f - 1e-7f) ? 1.0f : 0.0f;
            const float dpt = inside * 0.25f * (2.0f * om * logf(pt) - om * om / pt) * seg_scale * gscale;
#pragma unroll
            for (int j = 0; j < kLossMaxK1; ++j)
                if (j < k1) grad_seg[r * k1 + j] = dpt * pt_raw * ((j == lab ? 1.0f : 0.0f) - lg[j]);
        }
        if (lab <= 0) continue;
        // ---- the labelled class's row of the head
        const int c = lab - 1;
        const float *h = a.head + (r * a.k + c) * d;
        float *gh = BWD ? grad_head + (r * a.k + c) * d : nullptr;
        s_fg += 1.0f;
        const int tb[3] = { a.bin_x[r * a.k + c], a.bin_z[r * a.k + c], a.bin_t[r] };
        const float tr[3] = { a.res_x[r * a.k + c], a.res_z[r * a.k + c], a.res_t[r] };
        int off = 0;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int nb = g < 2 ? a.nbx : a.nbt;
            float v[kLossMaxBins], m2 = -INFINITY;
#pragma unroll
            for (int j = 0; j < kLossMaxBins; ++j) { v[j] = j < nb ? h[off + j] : -INFINITY; m2 = fmaxf(m2, v[j]); }
            float s2 = 0.f, vt = 0.f;
#pragma unroll
            for (int j = 0; j < kLossMaxBins; ++j) { if (j == tb[g]) vt = v[j]; v[j] = j < nb ? expf(v[j] - m2) : 0.f; s2 += v[j]; }
            const float res = h[off + nb + tb[g]];          // the residual of the TRUE bin (rpn_model.py:778-786)
            if (!BWD) {
                s_cls += logf(s2) + m2 - vt;
                s_reg += smooth_l1(res - tr[g]);
            } else {
                const float cs = a.cls_w * inv_den * gscale, is2 = 1.0f / s2;
#pragma unroll
                for (int j = 0; j < kLossMaxBins; ++j)
                    if (j < nb) gh[off + j] = cs * (v[j] * is2 - (j == tb[g] ? 1.0f : 0.0f));
                for (int j = 0; j < nb; ++j) gh[off + nb + j] = 0.0f;
                gh[off + nb + tb[g]] = a.reg_w * inv_den * gscale * smooth_l1_grad(res - tr[g]);
            }
            off += 2 * nb;
        }
        const float dy = h[off] - a.res_y[r];
        if (!BWD) s_reg += smooth_l1(dy); else gh[off] = a.reg_w * inv_den * gscale * smooth_l1_grad(dy);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float ds = h[off + 1 + j] - a.res_size[r * 3 + j];
            if (!BWD) s_reg += smooth_l1(ds); else gh[off + 1 + j] = a.reg_w * inv_den * gscale * smooth_l1_grad(ds);
        }
    }
    if (BWD) return;
    const int t = threadIdx.x;
    red[0][t] = s_seg; red[1][t] = s_cls; red[2][t] = s_reg; red[3][t] = s_fg;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red[q][t] += red[q][t + w];
        }
        __syncthreads();
    }
    if (t < 4) partial[blockIdx.x * 4 + t] = red[t][0];
}

// one workgroup: the four column sums of the partials in fp64 -> out[0..3] = (seg, cls, reg, #fg) raw sums, out[4] = the loss
__global__ __launch_bounds__(256) void rpn_loss_finalize_kernel(int nblk, long long rows, float seg_w, float cls_w, float reg_w,
                                                               const float *__restrict__ partial, float *__restrict__ out)
{
    __shared__ double red[4][256];
    const int t = threadIdx.x;
    double s[4] = { 0.0, 0.0, 0.0, 0.0 };
    for (int i = t; i < nblk; i += 256)
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] += static_cast<double>(partial[i * 4 + q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) red[q][t] = s[q];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red[q][t] += red[q][t + w];
        }
        __syncthreads();
    }
    if (t == 0) {
        const double den = red[3][0] > 1.0 ? red[3][0] : 1.0;
        const double seg = red[0][0] * seg_w / static_cast<double>(rows), cls = red[1][0] * cls_w / den, reg = red[2][0] * reg_w / den;
        out[0] = static_cast<float>(seg); out[1] = static_cast<float>(cls); out[2] = static_cast<float>(reg);
        out[3] = static_cast<float>(red[3][0]); out[4] = static_cast<float>(seg + cls + reg);
    }
}

constexpr int kLossBlocks = 1024;

static int glue_grid(long long items, int block)
{
    long long g = (items + block - 1) / block;
    const long long cap = static_cast<long long>(kNumCU) * 16;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return static_cast<int>(g);
}

}  // namespace hf

using namespace hf;

HF_API int hf_project_gather(int b, int p, int h, int w, int c, const float *pts, const float *calib, const float *img,
                             float *out, int *pix, hf_stream_t stream)
{
    if (b < 0 || p < 0 || h <= 0 || w <= 0 || c <= 0) return HF_EINVAL;
    const long long nrows = static_cast<long long>(b) * p;
    if (nrows == 0) return HF_OK;  // empty tensors carry null pointers
    if (!pts || !calib || !img || !out) return HF_EINVAL;
    const bool vec4 = c % 4 == 0 && (reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(out)) % 16 == 0;
    if (vec4)
        hipLaunchKernelGGL((project_gather_kernel<4>), dim3(glue_grid(nrows * (c / 4), 256)), dim3(256), 0, as_stream(stream),
                           p, h, w, c, nrows, pts, calib, img, out, pix);
    else
        hipLaunchKernelGGL((project_gather_kernel<1>), dim3(glue_grid(nrows * c, 256)), dim3(256), 0, as_stream(stream), p,
                           h, w, c, nrows, pts, calib, img, out, pix);
    return launch_status();
}

HF_API int hf_project_gather_grad(int b, int p, int h, int w, int c, const float *grad_out, const int *pix,
                                  float *grad_img, hf_stream_t stream)
{
    if (b < 0 || p < 0 || h <= 0 || w <= 0 || c <= 0 || (b > 0 && !grad_img)) return HF_EINVAL;
    if (b == 0) return HF_OK;
    hipStream_t st = as_stream(stream);
    int rc = hip_status(hipMemsetAsync(grad_img, 0, sizeof(float) * static_cast<size_t>(b) * h * w * c, st));
    if (rc != HF_OK) return rc;
    const long long nrows = static_cast<long long>(b) * p;
    if (nrows == 0) return HF_OK;
    if (!grad_out || !pix) return HF_EINVAL;
    hipLaunchKernelGGL(project_gather_grad_kernel, dim3(glue_grid(nrows * c, 256)), dim3(256), 0, st, p, h, w, c, nrows,
                       grad_out, pix, grad_img);
    return launch_status();
}

HF_API int hf_fuse_concat(long long rows, int c1, int c2, const float *a, const float *b, const float *masks, float *out,
                          hf_stream_t stream)
{
    if (rows < 0 || c1 <= 0 || c2 <= 0) return HF_EINVAL;
    if (rows == 0) return HF_OK;
    if (!a || !b || !out) return HF_EINVAL;
    const bool vec4 = c1 % 4 == 0 && c2 % 4 == 0 &&
                      (reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) % 16 == 0;
    if (vec4)
        hipLaunchKernelGGL((fuse_concat_kernel<4>), dim3(glue_grid(rows * ((c1 + c2) / 4), 256)), dim3(256), 0, as_stream(stream),
                           rows, c1, c2, a, b, masks, out);
    else
        hipLaunchKernelGGL((fuse_concat_kernel<1>), dim3(glue_grid(rows * (c1 + c2), 256)), dim3(256), 0, as_stream(stream), rows,
                           c1, c2, a, b, masks, out);
    return launch_status();
}

HF_API int hf_fuse_concat_grad(long long rows, int c1, int c2, const float *grad_out, const float *masks, float *grad_a,
                               float *grad_b, hf_stream_t stream)
{
    if (rows < 0 || c1 <= 0 || c2 <= 0) return HF_EINVAL;
    if (rows == 0 || (!grad_a && !grad_b)) return HF_OK;
    if (!grad_out) return HF_EINVAL;
    const uintptr_t al = reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(grad_a) | reinterpret_cast<uintptr_t>(grad_b);
    if (c1 % 4 == 0 && c2 % 4 == 0 && al % 16 == 0)
        hipLaunchKernelGGL((fuse_concat_grad_kernel<4>), dim3(glue_grid(rows * ((c1 + c2) / 4), 256)), dim3(256), 0,
                           as_stream(stream), rows, c1, c2, grad_out, masks, grad_a, grad_b);
    else
        hipLaunchKernelGGL((fuse_concat_grad_kernel<1>), dim3(glue_grid(rows * (c1 + c2), 256)), dim3(256), 0, as_stream(stream),
                           rows, c1, c2, grad_out, masks, grad_a, grad_b);
    return launch_status();
}

static int loss_args_ok(long long rows, int k, int nbx, int nbt, const void *a, const void *b, const void *c)
{
    return rows >= 0 && k > 0 && k + 1 <= kLossMaxK1 && nbx > 0 && nbx <= kLossMaxBins && nbt > 0 && nbt <= kLossMaxBins && (rows == 0 || (a && b && c));
}

HF_API size_t hf_rpn_loss_workspace(void) { return sizeof(float) * 4 * kLossBlocks; }

HF_API int hf_rpn_loss_fwd(long long rows, int k, int nbx, int nbt, const float *seg_logits, const float *head, const int *label,
                           const int *bin_x, const float *res_x, const int *bin_z, const float *res_z, const int *bin_theta,
                           const float *res_theta, const float *res_y, const float *res_size, float seg_weight, float cls_weight,
                           float reg_weight, float *out5, void *workspace, size_t workspace_bytes, hf_stream_t stream)
{
    if (!loss_args_ok(rows, k, nbx, nbt, seg_logits, head, label) || !out5) return HF_EINVAL;
    if (rows > 0 && (!bin_x || !res_x || !bin_z || !res_z || !bin_theta || !res_theta || !res_y || !res_size)) return HF_EINVAL;
    if (!workspace || workspace_bytes < hf_rpn_loss_workspace()) return HF_EWORKSPACE;
    LossArgs a = { rows, k, nbx, nbt, seg_logits, head, label, bin_x, bin_z, bin_theta, res_x, res_z, res_theta, res_y, res_size,
                   seg_weight, cls_weight, reg_weight };
    int nblk = static_cast<int>((rows + 255) / 256);
    if (nblk > kLossBlocks) nblk = kLossBlocks;
    if (nblk < 1) nblk = 1;
    float *partial = static_cast<float *>(workspace);
    hipLaunchKernelGGL((rpn_loss_kernel<false>), dim3(nblk), dim3(256), 0, as_stream(stream), a, partial, nullptr, nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(rpn_loss_finalize_kernel, dim3(1), dim3(256), 0, as_stream(stream), nblk, rows > 0 ? rows : 1, seg_weight, cls_weight,
                       reg_weight, partial, out5);
    return launch_status();
}

HF_API int hf_rpn_loss_bwd(long long rows, int k, int nbx, int nbt, const float *seg_logits, const float *head, const int *label,
                           const int *bin_x, const float *res_x, const int *bin_z, const float *res_z, const int *bin_theta,
                           const float *res_theta, const float *res_y, const float *res_size, float seg_weight, float cls_weight,
                           float reg_weight, const float *out5, const float *upstream, float *grad_seg, float *grad_head,
                           hf_stream_t stream)
{
    if (!loss_args_ok(rows, k, nbx, nbt, seg_logits, head, label) || !out5 || !upstream || (rows > 0 && (!grad_seg || !grad_head)))
        return HF_EINVAL;
    if (rows == 0) return HF_OK;
    if (!bin_x || !res_x || !bin_z || !res_z || !bin_theta || !res_theta || !res_y || !res_size) return HF_EINVAL;
    hipStream_t st = as_stream(stream);
    const int d = 4 * nbx + 2 * nbt + 4;
    int rc = hip_status(hipMemsetAsync(grad_head, 0, sizeof(float) * static_cast<size_t>(rows) * k * d, st));
    if (rc != HF_OK) return rc;
    LossArgs a = { rows, k, nbx, nbt, seg_logits, head, label, bin_x, bin_z, bin_theta, res_x, res_z, res_theta, res_y, res_size,
                   seg_weight, cls_weight, reg_weight };
    int nblk = static_cast<int>((rows + 255) / 256);
    if (nblk > kLossBlocks * 4) nblk = kLossBlocks * 4;
    hipLaunchKernelGGL((rpn_loss_kernel<true>), dim3(nblk), dim3(256), 0, st, a, nullptr, out5 + 3, upstream, grad_seg, grad_head);
    return launch_status();
}

HF_API int hf_bin_box_decode(long long rows, int k, const float *ref_pts, const float *ref_theta, const int *bin_x,
                             const float *res_x_norm, const int *bin_z, const float *res_z_norm, const int *bin_theta,
                             const float *res_theta_norm, const float *res_y, const float *res_size_norm,
                             const float *mean_sizes, const float *ss, const float *deltas, float r, float delta_theta,
                             float *boxes, hf_stream_t stream)
{
    if (rows < 0 || k <= 0) return HF_EINVAL;
    if (rows == 0) return HF_OK;
    if (!ref_pts || !bin_x || !res_x_norm || !bin_z || !res_z_norm || !bin_theta || !res_theta_norm || !res_y ||
        !res_size_norm || !mean_sizes || !ss || !deltas || !boxes)
        return HF_EINVAL;
    hipLaunchKernelGGL(bin_box_decode_kernel, dim3(glue_grid(rows * k, 256)), dim3(256), 0, as_stream(stream), rows, k,
                       ref_pts, ref_theta, bin_x, res_x_norm, bin_z, res_z_norm, bin_theta, res_theta_norm, res_y,
                       res_size_norm, mean_sizes, ss, deltas, r, delta_theta, boxes);
    return launch_status();
}

HF_API int hf_bin_box_encode(long long rows, int k, int rcnn, const float *ref_pts, const float *ref_theta,
                             const float *boxes, const float *mean_sizes, const float *ss, const float *deltas,
                             const float *hi_xz, float r, float hi_theta, float delta_theta, float half_delta_theta,
                             int *bin_x, float *res_x_norm, int *bin_z, float *res_z_norm, int *bin_theta,
                             float *res_theta_norm, float *res_y, float *res_size_norm, hf_stream_t stream)
{
    if (rows < 0 || k <= 0) return HF_EINVAL;
    if (rows == 0) return HF_OK;
    if (!ref_pts || !boxes || !mean_sizes || !ss || !deltas || !hi_xz || !bin_x || !res_x_norm || !bin_z || !res_z_norm ||
        !bin_theta || !res_theta_norm || !res_y || !res_size_norm)
        return HF_EINVAL;
    hipLaunchKernelGGL(bin_box_encode_kernel, dim3(glue_grid(rows, 256)), dim3(256), 0, as_stream(stream), rows, k, rcnn,
                       ref_pts, ref_theta, boxes, mean_sizes, ss, deltas, hi_xz, r, hi_theta, delta_theta,
                       half_delta_theta, bin_x, res_x_norm, bin_z, res_z_norm, bin_theta, res_theta_norm, res_y,
                       res_size_norm);
    return launch_status();
}

HF_API int hf_bin_head_decode(long long rows, int k, int nbx, int nbz, int nbt, const float *head, const float *ref_pts,
                              const float *ref_theta, const float *mean_sizes_k, const float *ss, const float *deltas,
                              float r, float delta_theta, const int *cls, float *boxes, hf_stream_t stream)
{
    if (rows < 0 || k <= 0 || nbx <= 0 || nbz <= 0 || nbt <= 0) return HF_EINVAL;
    if (rows == 0) return HF_OK;
    if (!head || !ref_pts || !mean_sizes_k || !ss || !deltas || !boxes) return HF_EINVAL;
    hipLaunchKernelGGL(bin_head_decode_kernel, dim3(glue_grid(rows * k, 256)), dim3(256), 0, as_stream(stream), rows, k, nbx,
                       nbz, nbt, head, ref_pts, ref_theta, mean_sizes_k, ss, deltas, r, delta_theta, cls, boxes);
    return launch_status();
}
