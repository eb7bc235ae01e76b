// Dev tool: how fast does ONE launch read a few tens of MB that the chip has not touched since the previous frame?  The batch-1 decoder
// layers with 21-55 MB of weights (level-6 transposed conv, aerial descriptor conv, conv6.0) all sit at 1.2-1.5 TB/s whatever kernel runs
// them; this measures the plain read: every workgroup requests its whole share up front (UB x 16 bytes per lane), sums it and writes one
// float.  Rows: buffer size x workgroups x threads, warm (the same buffer every launch) and cold (a ring of buffers larger than the 256 MB
// Infinity Cache, so every launch reads from HBM).
// usage (GPU box): hipcc --offload-arch=gfx950 -O3 tools/ubench_stream_small.hip -o /tmp/ubench_stream_small && /tmp/ubench_stream_small
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int UB>
__global__ void rd(const f4* __restrict__ src, size_t n4, float* out) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = lo + threadIdx.x; i < hi; i += (size_t)blockDim.x * UB) {
        f4 v[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) v[u] = i + (size_t)u * blockDim.x < hi ? __builtin_nontemporal_load(src + i + (size_t)u * blockDim.x) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < UB; ++u) acc += v[u];
    }
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[blockIdx.x] = s;   // (never: keeps the loads alive without a store per thread)
}
int main() {
    const size_t ring_bytes = (size_t)768 << 20;
    char* ring; (void)hipMalloc(&ring, ring_bytes); (void)hipMemset(ring, 0, ring_bytes);
    float* out; (void)hipMalloc(&out, 1 << 20);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mb : {8, 21, 26, 55, 128}) for (int wgs : {256, 512, 1024}) for (int nt : {256, 1024}) {
        const size_t bytes = (size_t)mb << 20, n4 = bytes / 16;
        const int slots = (int)(ring_bytes / bytes);
        float ms[2];
        for (int cold = 0; cold < 2; ++cold) {
            const int reps = 200;
            for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(rd<8>, dim3(wgs), dim3(nt), 0, 0, (const f4*)(ring + (size_t)(cold ? w % slots : 0) * bytes), n4, out);
            (void)hipEventRecord(e0);
            for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(rd<8>, dim3(wgs), dim3(nt), 0, 0, (const f4*)(ring + (size_t)(cold ? w % slots : 0) * bytes), n4, out);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[cold], e0, e1);
            ms[cold] /= reps;
        }
        printf("%4d MB  %4d workgroups x %4d threads: warm %6.1f us (%5.2f TB/s)   cold %6.1f us (%5.2f TB/s)\n", mb, wgs, nt, ms[0] * 1e3, bytes / (ms[0] * 1e-3) * 1e-12,
               ms[1] * 1e3, bytes / (ms[1] * 1e-3) * 1e-12);
    }
    return 0;
}
