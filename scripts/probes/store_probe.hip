// Probe: how fast can 256 workgroups of 1024 threads write 17 MB (the ball-query kernel's idx + grouped_xyz rows) with
// different store shapes?  Every wave owns 256 elements: 1 KB of idx + 3 KB of grouped_xyz, both contiguous.
//   0 = as the kernel did: 4 x (dword + b96) per lane, every instruction one contiguous range
//   1 = idx as one dwordx4 per lane; grouped as 3 dwordx4 per lane at a 48-byte lane stride (lane = 4 consecutive elements)
//   2 = idx as one dwordx4; grouped as 3 contiguous dwordx4 wave-stores (data would come through an LDS transpose)
//   3 = idx dwordx4; grouped 4 x b96
// Build: hipcc --offload-arch=gfx950 -O3 store_probe.hip -o store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u3v __attribute__((ext_vector_type(3)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
template <int V, int AUX>
__global__ __launch_bounds__(1024) void probe(int *idx, float *grouped, int rounds)
{
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const size_t e0 = ((size_t)blockIdx.x * 16 + wave) * 256 * rounds;
    for (int r = 0; r < rounds; ++r) {
        const size_t eb = e0 + (size_t)r * 256;
        __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(idx + eb, 0, 1024, 0x00020000);
        __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(grouped + eb * 3, 0, 3072, 0x00020000);
        const unsigned v = t + r;
        if (V == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_raw_buffer_store_b32(v + i, ri, (lane + 64 * i) * 4, 0, AUX);
                __builtin_amdgcn_raw_buffer_store_b96(u3v{ v, v + 1, v + 2 }, rg, (lane + 64 * i) * 12, 0, AUX);
            }
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(u4v{ v, v + 1, v + 2, v + 3 }, ri, lane * 16, 0, AUX);
            if (V == 1) {
#pragma unroll
                for (int i = 0; i < 3; ++i) __builtin_amdgcn_raw_buffer_store_b128(u4v{ v, v + 1, v + 2, v + i }, rg, lane * 48 + 16 * i, 0, AUX);
            } else if (V == 2) {
#pragma unroll
                for (int i = 0; i < 3; ++i) __builtin_amdgcn_raw_buffer_store_b128(u4v{ v, v + 1, v + 2, v + i }, rg, (lane + 64 * i) * 16, 0, AUX);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b96(u3v{ v, v + 1, v + i }, rg, (lane + 64 * i) * 12, 0, AUX);
            }
        }
    }
}
template <int V, int AUX>
float run(int *a, float *g, int iters)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<V, AUX>), dim3(256), dim3(1024), 0, 0, a, g, 1);
    (void)hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((probe<V, AUX>), dim3(256), dim3(1024), 0, 0, a, g, 1);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}
int main()
{
    int *a; float *g;
    const size_t ne = 256 * 16 * 256;   // 1 Mi elements = 4 MiB + 12 MiB
    (void)hipMalloc(&a, ne * 4); (void)hipMalloc(&g, ne * 12);
    printf("burst average per launch, 256 x 1024 threads writing 16.8 MB (plain / nontemporal):\n");
    printf("0: 4 x (dword + b96)            %.2f  %.2f us\n", run<0, 0>(a, g, 200), run<0, 2>(a, g, 200));
    printf("1: x4 + 3 x x4 @48B stride      %.2f  %.2f us\n", run<1, 0>(a, g, 200), run<1, 2>(a, g, 200));
    printf("2: x4 + 3 x x4 contiguous       %.2f  %.2f us\n", run<2, 0>(a, g, 200), run<2, 2>(a, g, 200));
    printf("3: x4 + 4 x b96                 %.2f  %.2f us\n", run<3, 0>(a, g, 200), run<3, 2>(a, g, 200));
    return 0;
}
