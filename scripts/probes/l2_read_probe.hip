// Probe: how fast can every CU read the same 196 KB cloud (8 clouds, 32 workgroups each) from its XCD's L2?
// Variants: 0 = dwordx3 per lane (AoS points, 12-byte stride), 1 = dwordx4 per lane (flat stream), 2 = 3 x dword per lane,
// 4 = three aligned dwordx4 per lane at a 48-byte lane stride (four points).  Build: hipcc --offload-arch=gfx950 -O3 l2_read_probe.hip -o l2_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
struct __attribute__((packed, aligned(4))) P3 { float x, y, z; };
template <int V, int NT>
__global__ __launch_bounds__(NT) void probe(const float *__restrict__ xyz, int n, float *out)
{
    const float *p = xyz + (size_t)blockIdx.x * n * 3;
    const int t = threadIdx.x;
    float acc = 0.f;
    constexpr int PPT = 16384 / NT;
    if (V == 0) {
        float x[PPT], y[PPT], z[PPT];
#pragma unroll
        for (int u = 0; u < PPT; ++u) { const P3 q = *(const P3 *)(p + (size_t)(u * NT + t) * 3); x[u] = q.x; y[u] = q.y; z[u] = q.z; }
#pragma unroll
        for (int u = 0; u < PPT; ++u) acc += x[u] + y[u] * z[u];
    } else if (V == 1) {
        constexpr int F4 = PPT * 3 / 4;
        float4 v[F4];
#pragma unroll
        for (int u = 0; u < F4; ++u) v[u] = *(const float4 *)(p + (size_t)(u * NT + t) * 4);
#pragma unroll
        for (int u = 0; u < F4; ++u) acc += v[u].x + v[u].y * v[u].z + v[u].w;
    } else if (V == 2) {
        float x[PPT * 3];
#pragma unroll
        for (int u = 0; u < PPT * 3; ++u) x[u] = p[u * NT + t];
#pragma unroll
        for (int u = 0; u < PPT * 3; ++u) acc += x[u];
    }
    if (V == 4) {
        // four consecutive points per lane = three aligned float4 at a 48-byte lane stride
        constexpr int Q = PPT / 4;
        float4 v[Q][3];
#pragma unroll
        for (int u = 0; u < Q; ++u)
#pragma unroll
            for (int i = 0; i < 3; ++i) v[u][i] = *(const float4 *)(p + (size_t)(u * NT + t) * 12 + 4 * i);
#pragma unroll
        for (int u = 0; u < Q; ++u)
#pragma unroll
            for (int i = 0; i < 3; ++i) acc += v[u][i].x + v[u][i].y * v[u][i].z + v[u][i].w;
    }
    if (acc == 123.456f) out[blockIdx.x * NT + t] = acc;
}
template <int V, int NT>
float run(const float *d, float *o, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<V, NT>), dim3(8, 32), dim3(NT), 0, 0, d, 16384, o);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((probe<V, NT>), dim3(8, 32), dim3(NT), 0, 0, d, 16384, o);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}
__global__ void empty_kernel(float *o) { if (o == nullptr) o[0] = 1; }
template <int NT> float run_empty(float *o, int iters, size_t lds)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void *)&empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(empty_kernel, dim3(8, 32), dim3(NT), lds, 0, o);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(empty_kernel, dim3(8, 32), dim3(NT), lds, 0, o);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}
int main()
{
    const size_t nfl = 8 * 16384 * 3;
    std::vector<float> h(nfl);
    for (size_t i = 0; i < nfl; ++i) h[i] = (float)(i % 977) * 0.01f;
    float *d, *o; hipMalloc(&d, nfl * 4); hipMalloc(&o, 256 * 1024 * 4);
    hipMemcpy(d, h.data(), nfl * 4, hipMemcpyHostToDevice);
    printf("burst average per launch (kernel + gap), 256 workgroups each reading one 196 KB cloud:\n");
    printf("empty 1024 thr, 0 LDS      %.2f us\n", run_empty<1024>(o, 200, 0));
    printf("empty 1024 thr, 90 KB LDS  %.2f us\n", run_empty<1024>(o, 200, 90 * 1024));
    printf("empty  512 thr, 90 KB LDS  %.2f us\n", run_empty<512>(o, 200, 90 * 1024));
    printf("empty  256 thr, 0 LDS      %.2f us\n", run_empty<256>(o, 200, 0));
    printf("dwordx3 1024 thr  %.2f us\n", run<0, 1024>(d, o, 200));
    printf("dwordx4 1024 thr  %.2f us\n", run<1, 1024>(d, o, 200));
    printf("3x dwordx4 @48B 1024 thr  %.2f us\n", run<4, 1024>(d, o, 200));
    printf("dword   1024 thr  %.2f us\n", run<2, 1024>(d, o, 200));
    printf("dwordx3  512 thr  %.2f us\n", run<0, 512>(d, o, 200));
    printf("dwordx4  512 thr  %.2f us\n", run<1, 512>(d, o, 200));
    return 0;
}
