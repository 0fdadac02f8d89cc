/*
 * hf_oracle.c -- CPU restatement of the HeteroFusionRCNN point-cloud hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / reported CPU baseline.  The product path
 * (heterofusionrcnn_amd + libhfops.so) never links or calls it.
 *
 * Every function restates, in plain single-threaded C, the algorithm of one
 * reference op and cites the reference file:line it follows (paths relative to
 * the reference checkout).  Canonical arithmetic: IEEE fp32, no FMA contraction
 * (build with -ffp-contract=off), expressions evaluated left to right exactly as
 * written in the reference.
 *
 * Pinning (see oracle/README.md): checked bit-for-bit against the reference's own
 * standalone CPU programs (oracle/_ref/libhfref_cpu.so, built from
 * grouping/test/query_ball_point.cpp, grouping/test/selection_sort.cpp,
 * interpolate/interpolate.cpp where they lie), against the reference's CUDA kernels
 * compiled unmodified by hipcc for gfx950 (oracle/_ref/libhfref_gpu.so; sampling,
 * grouping, bev_iou, cropping) on the GPU box, and against the known answers of
 * the reference demos (bev_iou/bev_iou.py:47-65).  three_nn / three_interpolate
 * kernels (interpolate/tf_interpolate_g.cu) need <cuda.h> and are unbuildable here:
 * three_nn is pinned by hand-derived cases only.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define HFO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ sampling */

/* farthest point sampling: sampling/tf_sampling_g.cu:105-170 (launch <<<32,512>>> :204).
 * The reference is a 512-thread block; thread t scans k = t, t+512, ... and keeps the
 * FIRST k with d2 > best (strict, best starts at -1); a pairwise tree then keeps the
 * left slot unless the right one is strictly greater (:153-164).  This is a sequential
 * but step-for-step emulation of that lock-step execution, so the tie rule
 * (smallest k mod 512, then smallest k) falls out of the same comparisons. */
HFO_API void hfo_farthest_point_sample(int b, int n, int m, const float *xyz, int *out)
{
    enum { T = 512 };
    if (m <= 0 || n <= 0) return;
    float *temp = (float *)malloc(sizeof(float) * (size_t)n);
    float slot_d[T];
    int slot_i[T];
    for (int i = 0; i < b; ++i) {
        const float *pts = xyz + (size_t)i * n * 3;
        int *o = out + (size_t)i * m;
        for (int k = 0; k < n; ++k) temp[k] = 1e38f;
        int old = 0;
        o[0] = 0;
        for (int j = 1; j < m; ++j) {
            const float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
            for (int t = 0; t < T; ++t) { slot_d[t] = -1.0f; slot_i[t] = 0; }
            for (int k = 0; k < n; ++k) {
                const int t = k % T;
                const float x2 = pts[k * 3 + 0], y2 = pts[k * 3 + 1], z2 = pts[k * 3 + 2];
                const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
                const float td = temp[k];
                const float d2 = fminf(d, td);
                if (d2 != td) temp[k] = d2;
                if (d2 > slot_d[t]) { slot_d[t] = d2; slot_i[t] = k; }
            }
            for (int u = 0; (1 << u) < T; ++u) {
                const int active = T >> (u + 1);
                for (int t = 0; t < active; ++t) {
                    const int i1 = (t * 2) << u, i2 = (t * 2 + 1) << u;
                    if (slot_d[i1] < slot_d[i2]) { slot_d[i1] = slot_d[i2]; slot_i[i1] = slot_i[i2]; }
                }
            }
            old = slot_i[0];
            o[j] = old;
        }
    }
    free(temp);
}

/* gather_point: sampling/tf_sampling_g.cu:172-181 */
HFO_API void hfo_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out)
{
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j) {
            const int a = idx[(size_t)i * m + j];
            for (int c = 0; c < 3; ++c)
                out[((size_t)i * m + j) * 3 + c] = inp[((size_t)i * n + a) * 3 + c];
        }
}

/* gather_point grad: zero fill (sampling/tf_sampling.cpp:174) + scatter add (tf_sampling_g.cu:183-192);
 * sequential order = ascending (batch, j). */
HFO_API void hfo_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx, float *inp_g)
{
    memset(inp_g, 0, sizeof(float) * (size_t)b * n * 3);
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j) {
            const int a = idx[(size_t)i * m + j];
            for (int c = 0; c < 3; ++c)
                inp_g[((size_t)i * n + a) * 3 + c] += out_g[((size_t)i * m + j) * 3 + c];
        }
}

/* ------------------------------------------------------------------ grouping */

/* query_ball_point: grouping/tf_grouping_g.cu:3-36 == grouping/test/query_ball_point.cpp:19-47.
 * Rows with no hit are never written by the reference (undefined there); defined as zeros here. */
HFO_API void hfo_query_ball_point(int b, int n, int m, float radius, int nsample,
                                  const float *xyz1, const float *xyz2, int *idx, int *pts_cnt)
{
    memset(idx, 0, sizeof(int) * (size_t)b * m * nsample);
    for (int i = 0; i < b; ++i) {
        const float *p1 = xyz1 + (size_t)i * n * 3;
        const float *p2 = xyz2 + (size_t)i * m * 3;
        int *row = idx + (size_t)i * m * nsample;
        for (int j = 0; j < m; ++j) {
            int cnt = 0;
            const float x2 = p2[j * 3 + 0], y2 = p2[j * 3 + 1], z2 = p2[j * 3 + 2];
            for (int k = 0; k < n; ++k) {
                if (cnt == nsample) break;
                const float x1 = p1[k * 3 + 0], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2];
                const float s = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
                const float d = fmaxf(sqrtf(s), 1e-20f);
                if (d < radius) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) row[(size_t)j * nsample + l] = k;
                    row[(size_t)j * nsample + cnt] = k;
                    cnt += 1;
                }
            }
            if (pts_cnt) pts_cnt[(size_t)i * m + j] = cnt;
        }
    }
}

/* group_point: grouping/tf_grouping_g.cu:40-57 == query_ball_point.cpp:52-66 */
HFO_API void hfo_group_point(int b, int n, int c, int m, int nsample,
                             const float *points, const int *idx, float *out)
{
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j)
            for (int k = 0; k < nsample; ++k) {
                const int ii = idx[((size_t)i * m + j) * nsample + k];
                memcpy(out + (((size_t)i * m + j) * nsample + k) * c,
                       points + ((size_t)i * n + ii) * c, sizeof(float) * (size_t)c);
            }
}

/* group_point grad: zero fill (grouping/tf_grouping.cpp:204) + scatter add
 * (tf_grouping_g.cu:61-78 == query_ball_point.cpp:70-84) */
HFO_API void hfo_group_point_grad(int b, int n, int c, int m, int nsample,
                                  const float *grad_out, const int *idx, float *grad_points)
{
    memset(grad_points, 0, sizeof(float) * (size_t)b * n * c);
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j)
            for (int k = 0; k < nsample; ++k) {
                const int ii = idx[((size_t)i * m + j) * nsample + k];
                const float *g = grad_out + (((size_t)i * m + j) * nsample + k) * c;
                float *dst = grad_points + ((size_t)i * n + ii) * c;
                for (int l = 0; l < c; ++l) dst[l] += g[l];
            }
}

/* select_top_k (SelectionSort op): grouping/tf_grouping_g.cu:83-123 == test/selection_sort.cpp */
HFO_API void hfo_select_top_k(int b, int n, int m, int k, const float *dist, int *outi, float *out)
{
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j) {
            const float *src = dist + ((size_t)i * m + j) * n;
            float *pd = out + ((size_t)i * m + j) * n;
            int *pi = outi + ((size_t)i * m + j) * n;
            for (int s = 0; s < n; ++s) { pd[s] = src[s]; pi[s] = s; }
            for (int s = 0; s < k && s < n; ++s) {
                int mn = s;
                for (int t = s + 1; t < n; ++t)
                    if (pd[t] < pd[mn]) mn = t;
                if (mn != s) {
                    const float tf = pd[mn]; pd[mn] = pd[s]; pd[s] = tf;
                    const int ti = pi[mn]; pi[mn] = pi[s]; pi[s] = ti;
                }
            }
        }
}

/* knn_point: grouping/tf_grouping.py:62-95 -- dist = |q|^2 - 2 q.p + |p|^2 then top_k(-dist).
 * tf.nn.top_k is third-party (TensorFlow, not vendored): restated as the k smallest
 * (dist, index) pairs, lower index first on ties (TF's documented tie rule).
 * dist here is the direct squared distance (qx-px)^2+...: the matmul form's rounding is
 * library-dependent -> "parity unpinned" for val; idx pinned only up to exact ties. */
HFO_API void hfo_knn_point(int b, int n, int m, int k, const float *xyz1, const float *xyz2,
                           float *val, int *idx)
{
    float *bd = (float *)malloc(sizeof(float) * (size_t)k);
    int *bi = (int *)malloc(sizeof(int) * (size_t)k);
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j) {
            const float *q = xyz2 + ((size_t)i * m + j) * 3;
            int have = 0;
            for (int p = 0; p < n; ++p) {
                const float *d1 = xyz1 + ((size_t)i * n + p) * 3;
                const float d = (q[0] - d1[0]) * (q[0] - d1[0]) + (q[1] - d1[1]) * (q[1] - d1[1]) +
                                (q[2] - d1[2]) * (q[2] - d1[2]);
                int pos;
                if (have < k) pos = have++;
                else if (d < bd[k - 1]) pos = k - 1;
                else continue;
                while (pos > 0 && d < bd[pos - 1]) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; --pos; }
                bd[pos] = d; bi[pos] = p;
            }
            for (int s = 0; s < k; ++s) {
                val[((size_t)i * m + j) * k + s] = s < have ? bd[s] : INFINITY;
                idx[((size_t)i * m + j) * k + s] = s < have ? bi[s] : 0;
            }
        }
    free(bd); free(bi);
}

/* --------------------------------------------------------------- interpolate */

/* three_nn: interpolate/tf_interpolate_g.cu:22-65.  Accumulators are double 1e40 (:43);
 * strict '<' cascade, earlier index wins ties; squared distances returned. */
HFO_API void hfo_three_nn(int b, int n, int m, const float *unknown, const float *known,
                          float *dist2, int *idx)
{
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < n; ++j) {
            const float *u = unknown + ((size_t)i * n + j) * 3;
            const float ux = u[0], uy = u[1], uz = u[2];
            double best1 = 1e40, best2 = 1e40, best3 = 1e40;
            int besti1 = 0, besti2 = 0, besti3 = 0;
            for (int k = 0; k < m; ++k) {
                const float *p = known + ((size_t)i * m + k) * 3;
                const float x = p[0], y = p[1], z = p[2];
                const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
                if (d < best1) {
                    best3 = best2; besti3 = besti2;
                    best2 = best1; besti2 = besti1;
                    best1 = d; besti1 = k;
                } else if (d < best2) {
                    best3 = best2; besti3 = besti2;
                    best2 = d; besti2 = k;
                } else if (d < best3) {
                    best3 = d; besti3 = k;
                }
            }
            float *od = dist2 + ((size_t)i * n + j) * 3;
            int *oi = idx + ((size_t)i * n + j) * 3;
            od[0] = (float)best1; od[1] = (float)best2; od[2] = (float)best3;
            oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
        }
}

/* three_interpolate, channel-first op layout: interpolate/tf_interpolate_g.cu:90-110
 * points (b,c,m), idx/weight (b,n,3) -> out (b,c,n);  w0*p0 + w1*p1 + w2*p2 left to right. */
HFO_API void hfo_three_interpolate(int b, int c, int m, int n, const float *points,
                                   const int *idx, const float *weight, float *out)
{
    for (int i = 0; i < b; ++i)
        for (int l = 0; l < c; ++l) {
            const float *p = points + ((size_t)i * c + l) * m;
            for (int j = 0; j < n; ++j) {
                const float *w = weight + ((size_t)i * n + j) * 3;
                const int *k = idx + ((size_t)i * n + j) * 3;
                out[((size_t)i * c + l) * n + j] = w[0] * p[k[0]] + w[1] * p[k[1]] + w[2] * p[k[2]];
            }
        }
}

/* three_interpolate grad, channel-first: zero fill (tf_interpolate.cpp:178) + tf_interpolate_g.cu:133-155 */
HFO_API void hfo_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                        const int *idx, const float *weight, float *grad_points)
{
    memset(grad_points, 0, sizeof(float) * (size_t)b * c * m);
    for (int i = 0; i < b; ++i)
        for (int l = 0; l < c; ++l) {
            float *g = grad_points + ((size_t)i * c + l) * m;
            for (int j = 0; j < n; ++j) {
                const float *w = weight + ((size_t)i * n + j) * 3;
                const int *k = idx + ((size_t)i * n + j) * 3;
                const float go = grad_out[((size_t)i * c + l) * n + j];
                g[k[0]] += go * w[0];
                g[k[1]] += go * w[1];
                g[k[2]] += go * w[2];
            }
        }
}

/* channel-last view of the same op = what interpolate/tf_interpolate.py:26-37 exposes
 * (points (b,m,c) -> out (b,n,c); it transposes around the channel-first op).  Values are
 * identical to hfo_three_interpolate on the transposed tensors; the reference's CPU twin
 * interpolate/interpolate.cpp:84-105 writes p*w instead of w*p (commutative, same bits). */
HFO_API void hfo_three_interpolate_cl(int b, int m, int c, int n, const float *points,
                                      const int *idx, const float *weight, float *out)
{
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < n; ++j) {
            const float *w = weight + ((size_t)i * n + j) * 3;
            const int *k = idx + ((size_t)i * n + j) * 3;
            const float *p0 = points + ((size_t)i * m + k[0]) * c;
            const float *p1 = points + ((size_t)i * m + k[1]) * c;
            const float *p2 = points + ((size_t)i * m + k[2]) * c;
            float *o = out + ((size_t)i * n + j) * c;
            for (int l = 0; l < c; ++l) o[l] = w[0] * p0[l] + w[1] * p1[l] + w[2] * p2[l];
        }
}

/* interpolate/interpolate.cpp:109-130 (channel-last grad), zero fill first */
HFO_API void hfo_three_interpolate_cl_grad(int b, int n, int c, int m, const float *grad_out,
                                           const int *idx, const float *weight, float *grad_points)
{
    memset(grad_points, 0, sizeof(float) * (size_t)b * m * c);
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < n; ++j) {
            const float *w = weight + ((size_t)i * n + j) * 3;
            const int *k = idx + ((size_t)i * n + j) * 3;
            const float *go = grad_out + ((size_t)i * n + j) * c;
            for (int l = 0; l < c; ++l) {
                grad_points[((size_t)i * m + k[0]) * c + l] += go[l] * w[0];
                grad_points[((size_t)i * m + k[1]) * c + l] += go[l] * w[1];
                grad_points[((size_t)i * m + k[2]) * c + l] += go[l] * w[2];
            }
        }
}

/* ------------------------------------------------------------------- bev_iou */

typedef struct { float x, y; } hfo_pt;

static const float HFO_EPS = 1e-8f; /* bev_iou/bev_iou_g.cu:7 */

/* bev_iou_g.cu:33-35 */
static float hfo_cross3(hfo_pt p1, hfo_pt p2, hfo_pt p0)
{
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

/* bev_iou_g.cu:37-43 */
static int hfo_rect_cross(hfo_pt p1, hfo_pt p2, hfo_pt q1, hfo_pt q2)
{
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

/* bev_iou_g.cu:45-60 */
static int hfo_in_box2d(const float *box, hfo_pt p)
{
    const float MARGIN = 1e-5f;
    const float cx = (box[0] + box[2]) / 2, cy = (box[1] + box[3]) / 2;
    const float ac = cosf(-box[4]), as = sinf(-box[4]);
    const float rx = (p.x - cx) * ac + (p.y - cy) * as + cx;
    const float ry = -(p.x - cx) * as + (p.y - cy) * ac + cy;
    return rx > box[0] - MARGIN && rx < box[2] + MARGIN && ry > box[1] - MARGIN && ry < box[3] + MARGIN;
}

/* bev_iou_g.cu:62-91 */
static int hfo_intersection(hfo_pt p1, hfo_pt p0, hfo_pt q1, hfo_pt q0, hfo_pt *ans)
{
    if (!hfo_rect_cross(p0, p1, q0, q1)) return 0;
    const float s1 = hfo_cross3(q0, p1, p0);
    const float s2 = hfo_cross3(p1, q1, p0);
    const float s3 = hfo_cross3(p0, q1, q0);
    const float s4 = hfo_cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    const float s5 = hfo_cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > HFO_EPS) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

/* bev_iou_g.cu:92-96 */
static hfo_pt hfo_rotate(hfo_pt c, float ac, float as, hfo_pt p)
{
    hfo_pt r;
    r.x = (p.x - c.x) * ac + (p.y - c.y) * as + c.x;
    r.y = -(p.x - c.x) * as + (p.y - c.y) * ac + c.y;
    return r;
}

/* bev_iou_g.cu:102-206 */
static float hfo_box_overlap(const float *a, const float *bx)
{
    const float a_x1 = a[0], a_y1 = a[1], a_x2 = a[2], a_y2 = a[3], a_angle = a[4];
    const float b_x1 = bx[0], b_y1 = bx[1], b_x2 = bx[2], b_y2 = bx[3], b_angle = bx[4];
    hfo_pt ca = { (a_x1 + a_x2) / 2, (a_y1 + a_y2) / 2 };
    hfo_pt cb = { (b_x1 + b_x2) / 2, (b_y1 + b_y2) / 2 };
    hfo_pt A[5] = { { a_x1, a_y1 }, { a_x2, a_y1 }, { a_x2, a_y2 }, { a_x1, a_y2 } };
    hfo_pt B[5] = { { b_x1, b_y1 }, { b_x2, b_y1 }, { b_x2, b_y2 }, { b_x1, b_y2 } };
    const float acs = cosf(a_angle), asn = sinf(a_angle);
    const float bcs = cosf(b_angle), bsn = sinf(b_angle);
    for (int k = 0; k < 4; ++k) {
        A[k] = hfo_rotate(ca, acs, asn, A[k]);
        B[k] = hfo_rotate(cb, bcs, bsn, B[k]);
    }
    A[4] = A[0];
    B[4] = B[0];

    hfo_pt cp[24]; /* reference declares 16 (:144); 16 crossings + 8 corners can reach 24 */
    hfo_pt ctr = { 0.0f, 0.0f };
    int cnt = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (hfo_intersection(A[i + 1], A[i], B[j + 1], B[j], &cp[cnt])) {
                ctr.x = ctr.x + cp[cnt].x; ctr.y = ctr.y + cp[cnt].y;
                ++cnt;
            }
    for (int k = 0; k < 4; ++k) {
        if (hfo_in_box2d(a, B[k])) {
            ctr.x = ctr.x + B[k].x; ctr.y = ctr.y + B[k].y;
            cp[cnt++] = B[k];
        }
        if (hfo_in_box2d(bx, A[k])) {
            ctr.x = ctr.x + A[k].x; ctr.y = ctr.y + A[k].y;
            cp[cnt++] = A[k];
        }
    }
    ctr.x /= cnt;
    ctr.y /= cnt;
    /* bubble sort by atan2 about the centroid, :181-190 (point_cmp :98-100) */
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (atan2f(cp[i].y - ctr.y, cp[i].x - ctr.x) > atan2f(cp[i + 1].y - ctr.y, cp[i + 1].x - ctr.x)) {
                const hfo_pt t = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = t;
            }
    float area = 0;
    for (int k = 0; k < cnt - 1; ++k) {
        const hfo_pt u = { cp[k].x - cp[0].x, cp[k].y - cp[0].y };
        const hfo_pt v = { cp[k + 1].x - cp[0].x, cp[k + 1].y - cp[0].y };
        area += u.x * v.y - u.y * v.x;
    }
    return fabsf(area) / 2.0f;
}

/* bev_iou_g.cu:208-215 */
static float hfo_iou_from_overlap(const float *a, const float *bx, float s)
{
    const float sa = (a[2] - a[0]) * (a[3] - a[1]);
    const float sb = (bx[2] - bx[0]) * (bx[3] - bx[1]);
    return s / fmaxf(sa + sb - s, HFO_EPS);
}

/* compute_bev_iou: bev_iou_g.cu:240-254 (one pair per thread) */
HFO_API void hfo_compute_bev_iou(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                                 float *ans_overlap, float *ans_iou)
{
    for (int i = 0; i < num_a; ++i)
        for (int j = 0; j < num_b; ++j) {
            const float s = hfo_box_overlap(boxes_a + (size_t)i * 5, boxes_b + (size_t)j * 5);
            if (ans_overlap) ans_overlap[(size_t)i * num_b + j] = s;
            if (ans_iou) ans_iou[(size_t)i * num_b + j] = hfo_iou_from_overlap(boxes_a + (size_t)i * 5, boxes_b + (size_t)j * 5, s);
        }
}

/* NMS suppression mask: bev_iou_g.cu:256-298.  mask is (n, ceil(n/64)) u64; every tile is
 * evaluated (the lower-triangle skip is commented out at :264). */
HFO_API void hfo_nms_mask(const float *boxes, uint64_t *mask, int n, float thresh)
{
    const int cb = (n + 63) / 64;
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < cb; ++c) {
            const int col_size = n - c * 64 < 64 ? n - c * 64 : 64;
            const int start = (i / 64 == c) ? (i % 64) + 1 : 0;
            uint64_t t = 0;
            for (int jj = start; jj < col_size; ++jj) {
                const float *bi = boxes + (size_t)i * 5, *bj = boxes + (size_t)(c * 64 + jj) * 5;
                const float s = hfo_box_overlap(bi, bj);
                if (hfo_iou_from_overlap(bi, bj, s) > thresh) t |= 1ULL << jj;
            }
            mask[(size_t)i * cb + c] = t;
        }
}

/* greedy sweep + pad with keep[0]: bev_iou/bev_iou.cpp:87-112 */
HFO_API int hfo_nms_sweep(const uint64_t *mask, int n, int *keep)
{
    const int cb = (n + 63) / 64;
    uint64_t *remv = (uint64_t *)calloc((size_t)cb, sizeof(uint64_t));
    int num = 0;
    for (int i = 0; i < n; ++i) {
        const int nb = i / 64, ib = i % 64;
        if (!(remv[nb] & (1ULL << ib))) {
            keep[num++] = i;
            const uint64_t *p = mask + (size_t)i * cb;
            for (int j = nb; j < cb; ++j) remv[j] |= p[j];
        }
    }
    const int kept = num;
    for (; num < n; ++num) keep[num] = keep[0];
    free(remv);
    return kept;
}

/* oriented_nms = mask kernel + host sweep (bev_iou.cpp:60-116); returns #kept before padding */
HFO_API int hfo_oriented_nms(const float *boxes, int n, float thresh, int *keep)
{
    const int cb = (n + 63) / 64;
    uint64_t *mask = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n * cb);
    hfo_nms_mask(boxes, mask, n, thresh);
    const int kept = hfo_nms_sweep(mask, n, keep);
    free(mask);
    return kept;
}

/* ------------------------------------------------------------------ cropping */

/* cropping/tf_cropping_g.cu:3-5 */
static float hfo_dot3(float x1, float y1, float z1, float x2, float y2, float z2)
{
    return x1 * x2 + y1 * y2 + z1 * z2;
}

/* cropping/tf_cropping_g.cu:7-41 */
static int hfo_point_inside(float px, float py, float pz, const float *p1, const float *p2,
                            const float *p4, const float *p5)
{
    const float ux = p2[0] - p1[0], uy = p2[1] - p1[1], uz = p2[2] - p1[2];
    const float vx = p4[0] - p1[0], vy = p4[1] - p1[1], vz = p4[2] - p1[2];
    const float wx = p5[0] - p1[0], wy = p5[1] - p1[1], wz = p5[2] - p1[2];
    const float u_x = hfo_dot3(ux, uy, uz, px, py, pz);
    const float u_1 = hfo_dot3(ux, uy, uz, p1[0], p1[1], p1[2]);
    const float u_2 = hfo_dot3(ux, uy, uz, p2[0], p2[1], p2[2]);
    const float v_x = hfo_dot3(vx, vy, vz, px, py, pz);
    const float v_1 = hfo_dot3(vx, vy, vz, p1[0], p1[1], p1[2]);
    const float v_4 = hfo_dot3(vx, vy, vz, p4[0], p4[1], p4[2]);
    const float w_x = hfo_dot3(wx, wy, wz, px, py, pz);
    const float w_1 = hfo_dot3(wx, wy, wz, p1[0], p1[1], p1[2]);
    const float w_5 = hfo_dot3(wx, wy, wz, p5[0], p5[1], p5[2]);
    return u_1 < u_x && u_x < u_2 && v_1 < v_x && v_x < v_4 && w_1 < w_x && w_x < w_5;
}

/* pc_crop_and_sample: cropping/tf_cropping_g.cu:43-132, outputs initialised as in
 * tf_cropping.cpp:171-176.  The reference appends in atomic arrival order; this is the
 * member of its output set that a sequential execution (blockDim.x == 1) yields:
 * ascending point index.  Padding :108-126: slot s <- slot (s - cnt) mod cnt. */
HFO_API void hfo_pc_crop_and_sample(const float *pts, const float *fts, const float *intens,
                                    const uint8_t *mask, const float *boxes, const int *box_ind,
                                    int num_boxes, int batch, int npts, int resize, int channel,
                                    int ichannel, float *crop_pts, float *crop_fts, float *crop_int,
                                    uint8_t *crop_mask, int *crop_ind, uint8_t *non_empty)
{
    (void)batch;
    memset(crop_pts, 0, sizeof(float) * (size_t)num_boxes * resize * 3);
    memset(crop_fts, 0, sizeof(float) * (size_t)num_boxes * resize * channel);
    memset(crop_int, 0, sizeof(float) * (size_t)num_boxes * resize * ichannel);
    memset(crop_mask, 0, (size_t)num_boxes * resize);
    memset(crop_ind, 0, sizeof(int) * (size_t)num_boxes * resize);
    memset(non_empty, 1, (size_t)num_boxes);
    for (int bx = 0; bx < num_boxes; ++bx) {
        const float *bb = boxes + (size_t)bx * 24;
        const float p1[3] = { bb[0], bb[8], bb[16] };
        const float p2[3] = { bb[1], bb[9], bb[17] };
        const float p4[3] = { bb[3], bb[11], bb[19] };
        const float p5[3] = { bb[4], bb[12], bb[20] };
        const int bch = box_ind[bx];
        const float *P = pts + (size_t)bch * npts * 3;
        const float *F = fts + (size_t)bch * npts * channel;
        const float *I = intens + (size_t)bch * npts * ichannel;
        const uint8_t *M = mask + (size_t)bch * npts;
        float *op = crop_pts + (size_t)bx * resize * 3;
        float *of = crop_fts + (size_t)bx * resize * channel;
        float *oi = crop_int + (size_t)bx * resize * ichannel;
        uint8_t *om = crop_mask + (size_t)bx * resize;
        int *on = crop_ind + (size_t)bx * resize;
        int cnt = 0;
        for (int p = 0; p < npts && cnt < resize; ++p) {
            const float px = P[p * 3], py = P[p * 3 + 1], pz = P[p * 3 + 2];
            if (!hfo_point_inside(px, py, pz, p1, p2, p4, p5)) continue;
            op[cnt * 3] = px; op[cnt * 3 + 1] = py; op[cnt * 3 + 2] = pz;
            memcpy(of + (size_t)cnt * channel, F + (size_t)p * channel, sizeof(float) * (size_t)channel);
            memcpy(oi + (size_t)cnt * ichannel, I + (size_t)p * ichannel, sizeof(float) * (size_t)ichannel);
            om[cnt] = M[p];
            on[cnt] = p;
            ++cnt;
        }
        if (cnt == 0) {
            non_empty[bx] = 0;
        } else if (cnt < resize) {
            const int init = cnt;
            int counter = 0;
            while (cnt < resize) {
                const int s = counter++ % init;
                op[cnt * 3] = op[s * 3]; op[cnt * 3 + 1] = op[s * 3 + 1]; op[cnt * 3 + 2] = op[s * 3 + 2];
                memcpy(of + (size_t)cnt * channel, of + (size_t)s * channel, sizeof(float) * (size_t)channel);
                memcpy(oi + (size_t)cnt * ichannel, oi + (size_t)s * ichannel, sizeof(float) * (size_t)ichannel);
                om[cnt] = om[s];
                on[cnt] = on[s];
                ++cnt;
            }
        }
    }
}

/* PcCropAndSampleGradFts: zero fill (tf_cropping.cpp:225) + tf_cropping_g.cu:134-150 */
HFO_API void hfo_pc_crop_and_sample_grad_fts(const int *box_ind, const int *crop_ind,
                                             const float *grad_crop_fts, int num_boxes, int batch,
                                             int npts, int resize, int channel, float *grad_fts)
{
    memset(grad_fts, 0, sizeof(float) * (size_t)batch * npts * channel);
    for (int bx = 0; bx < num_boxes; ++bx) {
        float *g = grad_fts + (size_t)box_ind[bx] * npts * channel;
        for (int p = 0; p < resize; ++p) {
            const int k = crop_ind[(size_t)bx * resize + p];
            const float *src = grad_crop_fts + ((size_t)bx * resize + p) * channel;
            for (int c = 0; c < channel; ++c) g[(size_t)k * channel + c] += src[c];
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * glue either side of the ops.
 *
 * hfo_project_gather -- the LiDAR->image fusion step, hf/core/projection.py:5-32 (tf_rect_to_image) followed by
 * hf/core/models/rpn_model.py:227-235: homogeneous point times the 3x4 P2 matrix, divide by depth, tf.cast to int32
 * (truncation toward zero), tf.gather_nd(img_fts, [b, v, u]).  The matmul's 4-term sums are evaluated left to
 * right (TensorFlow's order inside a 4-long dot product is not specified: parity unpinned there); an index outside
 * the image yields zeros, which is tf.gather_nd's documented behaviour on GPU.  pix receives [u, v] per point.
 * ------------------------------------------------------------------------------------------ */
HFO_API void hfo_project_gather(int b, int p, int h, int w, int c, const float *pts, const float *calib,
                                const float *img, float *out, int *pix)
{
    for (int i = 0; i < b; ++i) {
        const float *P = calib + (size_t)i * 12;
        for (int j = 0; j < p; ++j) {
            const float *x = pts + ((size_t)i * p + j) * 3;
            const float uh = P[0] * x[0] + P[1] * x[1] + P[2] * x[2] + P[3];
            const float vh = P[4] * x[0] + P[5] * x[1] + P[6] * x[2] + P[7];
            const float d = P[8] * x[0] + P[9] * x[1] + P[10] * x[2] + P[11];
            const float uf = uh / d, vf = vh / d;
            /* a float outside the int32 range (or NaN) has no defined cast: treated as outside the image */
            const int ok = uf > -2147483648.0f && uf < 2147483648.0f && vf > -2147483648.0f && vf < 2147483648.0f;
            const int u = ok ? (int)uf : -1, v = ok ? (int)vf : -1;
            if (pix) { pix[((size_t)i * p + j) * 2] = u; pix[((size_t)i * p + j) * 2 + 1] = v; }
            float *o = out + ((size_t)i * p + j) * c;
            if (u >= 0 && u < w && v >= 0 && v < h) {
                const float *s = img + (((size_t)i * h + v) * w + u) * c;
                for (int l = 0; l < c; ++l) o[l] = s[l];
            } else {
                for (int l = 0; l < c; ++l) o[l] = 0.0f;
            }
        }
    }
}

/* gradient w.r.t. the image features: rows scattered back in ascending point order */
HFO_API void hfo_project_gather_grad(int b, int p, int h, int w, int c, const float *grad_out, const int *pix,
                                     float *grad_img)
{
    memset(grad_img, 0, sizeof(float) * (size_t)b * h * w * c);
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < p; ++j) {
            const int u = pix[((size_t)i * p + j) * 2], v = pix[((size_t)i * p + j) * 2 + 1];
            if (u < 0 || u >= w || v < 0 || v >= h) continue;
            float *g = grad_img + (((size_t)i * h + v) * w + u) * c;
            const float *s = grad_out + ((size_t)i * p + j) * c;
            for (int l = 0; l < c; ++l) g[l] += s[l];
        }
}

/* tf.mod on floats = floored modulo (result takes the sign of the divisor) */
static float hfo_floormodf(float x, float y)
{
    float r = fmodf(x, y);
    if (r != 0.0f && ((y < 0.0f) != (r < 0.0f))) r += y;
    return r;
}

/* hfo_bin_box_decode -- hf/core/bin_based_box3d_encoder.py:9-139 (tf_decode), both ranks at once: `rows` = B*p (RPN) or
 * the number of RoIs (RCNN), k classes per row.  ref_theta may be NULL (the RPN passes the constant 0: no rotation).
 * ss / deltas are the per-class search range and bin length (float32 of the config's doubles), r / delta_theta the
 * orientation range and bin length.  rot = [[c, -s], [s, c]]; matmul(rot, [dx dz], transpose_a, transpose_b) gives
 * dx' = c*dx + s*dz, dz' = -s*dx + c*dz (:58-79).  boxes (rows, k, 7) = [x, y, z, l, w, h, ry]. */
HFO_API void hfo_bin_box_decode(long long rows, int k, const float *ref_pts, const float *ref_theta, const int *bin_x,
                                const float *res_x_norm, const int *bin_z, const float *res_z_norm,
                                const int *bin_theta, const float *res_theta_norm, const float *res_y,
                                const float *res_size_norm, const float *mean_sizes, const float *ss,
                                const float *deltas, float r, float delta_theta, float *boxes)
{
    for (long long i = 0; i < rows; ++i) {
        const float th0 = ref_theta ? ref_theta[i] : 0.0f;
        const float sn = ref_theta ? sinf(th0) : 0.0f, cs = ref_theta ? cosf(th0) : 1.0f;
        for (int j = 0; j < k; ++j) {
            const size_t e = (size_t)i * k + j;
            float dx = ((float)bin_x[e] + 0.5f) * deltas[j] - ss[j] + res_x_norm[e] * deltas[j];
            float dz = ((float)bin_z[e] + 0.5f) * deltas[j] - ss[j] + res_z_norm[e] * deltas[j];
            if (ref_theta) {
                const float rx = cs * dx + sn * dz, rz = -sn * dx + cs * dz;
                dx = rx; dz = rz;
            }
            float *o = boxes + e * 7;
            o[0] = dx + ref_pts[i * 3 + 0];
            o[1] = res_y[e] + ref_pts[i * 3 + 1];
            o[2] = dz + ref_pts[i * 3 + 2];
            for (int d = 0; d < 3; ++d) o[3 + d] = mean_sizes[e * 3 + d] + res_size_norm[e * 3 + d] * mean_sizes[e * 3 + d];
            o[6] = th0 + ((float)bin_theta[e] + 0.5f) * delta_theta - r + res_theta_norm[e] * 0.5f * delta_theta;
        }
    }
}

/* hfo_bin_box_encode -- bin_based_box3d_encoder.py:142-269 (tf_encode).  rcnn = 0: the RPN rank (dtheta = ry - ref_theta,
 * clipped to [0, 2R - 1e-3] after the +R shift); rcnn = 1: the RCNN rank (orientation folded into (-pi/2, pi/2] around
 * the proposal's, :235-246).  Outputs: bin_x/res_x_norm/bin_z/res_z_norm (rows, k); bin_theta/res_theta_norm/res_y
 * (rows); res_size_norm (rows, 3).  Python computes 2*Ss - 1e-3, 2*R - 1e-3, 0.5*DELTA_THETA in double and TF casts
 * the result to float32: the `hi_*` arguments carry those already-rounded constants. */
HFO_API void hfo_bin_box_encode(long long rows, int k, int rcnn, const float *ref_pts, const float *ref_theta,
                                const float *boxes, const float *mean_sizes, const float *ss, const float *deltas,
                                const float *hi_xz, float r, float hi_theta, float delta_theta, float half_delta_theta,
                                int *bin_x, float *res_x_norm, int *bin_z, float *res_z_norm, int *bin_theta,
                                float *res_theta_norm, float *res_y, float *res_size_norm)
{
    const float two_pi = (float)(2.0 * 3.141592653589793), pi = (float)3.141592653589793;
    const float half_pi = (float)(0.5 * 3.141592653589793), three_half_pi = (float)(1.5 * 3.141592653589793);
    for (long long i = 0; i < rows; ++i) {
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
            float dtheta = bx[6] - hfo_floormodf(th0, two_pi);
            dtheta = hfo_floormodf(dtheta, two_pi);
            if (dtheta > half_pi && dtheta < three_half_pi) dtheta = hfo_floormodf(dtheta + pi, two_pi);
            dshift = hfo_floormodf(dtheta + half_pi, two_pi);
            dshift = fminf(fmaxf(dshift - r, 1e-3f), hi_theta);
        }
        for (int j = 0; j < k; ++j) {
            const size_t e = (size_t)i * k + j;
            const float xs = fminf(fmaxf(dx + ss[j], 0.0f), hi_xz[j]);
            const float fx = floorf(xs / deltas[j]);
            bin_x[e] = (int)fx;
            res_x_norm[e] = (xs - (fx + 0.5f) * deltas[j]) / deltas[j];
            const float zs = fminf(fmaxf(dz + ss[j], 0.0f), hi_xz[j]);
            const float fz = floorf(zs / deltas[j]);
            bin_z[e] = (int)fz;
            res_z_norm[e] = (zs - (fz + 0.5f) * deltas[j]) / deltas[j];
        }
        const float ft = floorf(dshift / delta_theta);
        bin_theta[i] = (int)ft;
        res_theta_norm[i] = (dshift - (ft + 0.5f) * delta_theta) / half_delta_theta;
        res_y[i] = dy;
        for (int d = 0; d < 3; ++d) res_size_norm[i * 3 + d] = (bx[3 + d] - mean_sizes[i * 3 + d]) / mean_sizes[i * 3 + d];
    }
}

/* hfo_bin_head_decode -- the decoding block of the RPN / RCNN heads as one function: _parse_rpn_output
 * (hf/core/models/rpn_model.py:870-935: slices [bin_x logits | res_x_norms | bin_z logits | res_z_norms | bin_theta logits |
 * res_theta_norms | res_y | res_size_norm(3)] of each (row, class) vector of length d = 2*nbx + 2*nbz + 2*nbt + 4),
 * tf.argmax over each logit slice (first maximum), _gather_residuals (:248-290), per-class mean sizes tiled over the
 * rows (:609-621), tf_decode (:623-639) and, with cls != NULL, _gather_cls_proposals (:237-246, 640-642): row i keeps
 * only class cls[i] -> boxes (rows, 7); otherwise boxes (rows, k, 7). */
HFO_API void hfo_bin_head_decode(long long rows, int k, int nbx, int nbz, int nbt, const float *head,
                                 const float *ref_pts, const float *ref_theta, const float *mean_sizes_k,
                                 const float *ss, const float *deltas, float r, float delta_theta, const int *cls,
                                 float *boxes)
{
    const int d = 2 * nbx + 2 * nbz + 2 * nbt + 4;
    for (long long i = 0; i < rows; ++i) {
        const float th0 = ref_theta ? ref_theta[i] : 0.0f;
        const float sn = ref_theta ? sinf(th0) : 0.0f, cs = ref_theta ? cosf(th0) : 1.0f;
        for (int j = 0; j < k; ++j) {
            if (cls && cls[i] != j) continue;
            const float *v = head + ((size_t)i * k + j) * d;
            int bx = 0, bz = 0, bt = 0;
            for (int q = 1; q < nbx; ++q) if (v[q] > v[bx]) bx = q;
            const float *vz = v + 2 * nbx;
            for (int q = 1; q < nbz; ++q) if (vz[q] > vz[bz]) bz = q;
            const float *vt = v + 2 * nbx + 2 * nbz;
            for (int q = 1; q < nbt; ++q) if (vt[q] > vt[bt]) bt = q;
            const float rx = v[nbx + bx], rz = vz[nbz + bz], rt = vt[nbt + bt];
            const float *tail = v + 2 * nbx + 2 * nbz + 2 * nbt;
            float dx = ((float)bx + 0.5f) * deltas[j] - ss[j] + rx * deltas[j];
            float dz = ((float)bz + 0.5f) * deltas[j] - ss[j] + rz * deltas[j];
            if (ref_theta) {
                const float ax = cs * dx + sn * dz, az = -sn * dx + cs * dz;
                dx = ax; dz = az;
            }
            float *o = cls ? boxes + (size_t)i * 7 : boxes + ((size_t)i * k + j) * 7;
            o[0] = dx + ref_pts[i * 3 + 0];
            o[1] = tail[0] + ref_pts[i * 3 + 1];
            o[2] = dz + ref_pts[i * 3 + 2];
            for (int c = 0; c < 3; ++c) o[3 + c] = mean_sizes_k[j * 3 + c] + tail[1 + c] * mean_sizes_k[j * 3 + c];
            o[6] = th0 + ((float)bt + 0.5f) * delta_theta - r + rt * 0.5f * delta_theta;
        }
    }
}
