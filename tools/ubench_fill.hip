// Dev microbenchmark (gfx950): how many independent v_fma_f32 hide behind one MFMA of the SAME wave (one wave per SIMD)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fill.hip -o tools/ubench_fill
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int K>
__global__ __launch_bounds__(256) void k(int iters, float* out, long long* cyc) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[8];
    f32x16 big[2];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)(lane + i);
    const float a = (float)lane, b = 1.f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(float)lane; bb[i] = (__bf16)1.f; }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            if (KIND == 1) big[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[i & 1], 0, 0, 0);
            if (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < K; ++j) v[(i + j) & 7] = __builtin_fmaf(v[(i + j) & 7], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + v[i];
    s += big[0][0] + big[1][3];
    if (s == 12345.678f) out[lane] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, int K>
static void run(const char* name, float* out, long long* cyc) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, K>), dim3(256), dim3(256), 0, 0, iters, out, cyc);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND, K>), dim3(256), dim3(256), 0, 0, iters, out, cyc);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long c = 0;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    std::printf("%-22s + %d fma per MFMA: %7.2f ns per MFMA slot, s_memtime ticks per slot %6.1f\n", name, K, 1e6 * ms / (iters * 8.0), (double)c / (iters * 8.0));
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 8);
    run<0, 0>("f32 16x16x4", out, cyc); run<0, 2>("f32 16x16x4", out, cyc); run<0, 4>("f32 16x16x4", out, cyc); run<0, 6>("f32 16x16x4", out, cyc); run<0, 8>("f32 16x16x4", out, cyc); run<0, 12>("f32 16x16x4", out, cyc);
    run<1, 0>("f32 32x32x2", out, cyc); run<1, 4>("f32 32x32x2", out, cyc); run<1, 8>("f32 32x32x2", out, cyc); run<1, 12>("f32 32x32x2", out, cyc); run<1, 16>("f32 32x32x2", out, cyc);
    run<2, 0>("bf16 16x16x32", out, cyc); run<2, 2>("bf16 16x16x32", out, cyc); run<2, 4>("bf16 16x16x32", out, cyc);
    return 0;
}
