// Dev tool: does a kernel that uses scratch (private segment) start later than one that does not?  Two persistent-style kernels with the
// same body; B indexes a small private array dynamically (forced to scratch).  usage: hipcc --offload-arch=gfx950 -O3 tools/ubench_scratch.hip -o tools/ubench_scratch && tools/ubench_scratch
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void ka(float* p, int n) {
    float a = p[threadIdx.x];
    for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
    p[blockIdx.x * 256 + threadIdx.x] = a;
}
__global__ __launch_bounds__(256) void kb(float* p, int n, int sel) {
    volatile float t[12];                           // (volatile: stays in memory = scratch)
    for (int i = 0; i < 12; ++i) t[i] = p[threadIdx.x + i];
    float a = t[(sel + threadIdx.x) % 12];          // dynamic index: the array lives in scratch
    for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
    p[blockIdx.x * 256 + threadIdx.x] = a + t[(sel * 7 + 3) % 12];
}
int main() {
    float* d; (void)hipMalloc(&d, 1 << 24); (void)hipMemset(d, 0, 1 << 24);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int grid : {256, 512, 2048}) for (int n : {100, 20000}) {
        float ms[2];
        for (int v = 0; v < 2; ++v) {
            for (int w = 0; w < 20; ++w) { if (v) hipLaunchKernelGGL(kb, dim3(grid), dim3(256), 0, 0, d, n, w); else hipLaunchKernelGGL(ka, dim3(grid), dim3(256), 0, 0, d, n); }
            (void)hipEventRecord(e0);
            for (int w = 0; w < 500; ++w) { if (v) hipLaunchKernelGGL(kb, dim3(grid), dim3(256), 0, 0, d, n, w); else hipLaunchKernelGGL(ka, dim3(grid), dim3(256), 0, 0, d, n); }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[v], e0, e1);
        }
        printf("grid %5d body %6d: plain %.2f us/launch, scratch %.2f us/launch\n", grid, n, ms[0] * 2, ms[1] * 2);
    }
    return 0;
}
