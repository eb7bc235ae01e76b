// Dev tool: how fast does the chip START workgroups?  Empty kernels (one store per workgroup) of G workgroups x T threads, with and without
// 64 KB of dynamic LDS, timed back to back.  Behind the finding that the latency-form GEMM (sixteen-wave workgroups) loses with more,
// smaller workgroups: time per launch grows linearly with G.
// usage (GPU box): hipcc --offload-arch=gfx950 -O3 tools/ubench_dispatch.hip -o /tmp/ubench_dispatch && /tmp/ubench_dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* p) {
    extern __shared__ float sm[];
    if (threadIdx.x == 0) p[blockIdx.x] = (float)blockIdx.x;
}
int main() {
    float* d; (void)hipMalloc(&d, 1 << 24);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int lds : {0, 64 * 1024}) for (int nt : {64, 256, 512, 1024}) {
        printf("LDS %2d KB, %4d threads:", lds >> 10, nt);
        for (int g : {64, 256, 1024, 4096, 16384}) {
            for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k, dim3(g), dim3(nt), lds, 0, d);
            (void)hipEventRecord(e0);
            for (int w = 0; w < 200; ++w) hipLaunchKernelGGL(k, dim3(g), dim3(nt), lds, 0, d);
            float ms; (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
            printf("  G %5d: %7.1f us", g, ms * 5.f);
        }
        printf("\n");
    }
    return 0;
}
