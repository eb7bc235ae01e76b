// Dev microbenchmark (gfx950): what overlaps with an fp32 MFMA stream of ANOTHER wave on the same SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_overlap.hip -o tools/ubench_overlap && tools/ubench_overlap
// One 512-thread workgroup per CU (8 waves, two per SIMD).  Waves 0-3 take role A, waves 4-7 role B; every role is a
// fixed amount of work, so  t(A|B) ~ max(t(A|idle), t(idle|B))  means the two streams overlap and  ~ sum  means they serialise.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { IDLE = 0, MFMA = 1, VALU = 2, LDSR = 3, VMEM = 4, MFMA_LDS = 5, MFMA_DEP = 6 };

__device__ __forceinline__ void role_mfma(int iters, float* out, int lane) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = (float)lane, b = 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[lane] = s;
}
__device__ __forceinline__ void role_valu(int iters, float* out, int lane) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)(lane + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[lane] = s;
}
__device__ __forceinline__ void role_lds(int iters, float* out, int lane, const float* lds) {
    f32x2 s = {0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s += *reinterpret_cast<const volatile f32x2*>(lds + ((lane * 2 + i * 128 + it * 2) & 4095));
    }
    if (s.x + s.y == 12345.678f) out[lane] = s.x;
}
__device__ __forceinline__ void role_vmem(int iters, float* out, int lane, const float* g) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s += *reinterpret_cast<const f32x4*>(g + (((it * 4 + i) * 256 + lane * 4) & 65535));
    }
    if (s.x + s.y == 12345.678f) out[lane] = s.x;
}
// the k-step of the Winograd kernel: one ds_read_b64 per two MFMAs
__device__ __forceinline__ void role_mfma_lds(int iters, float* out, int lane, const float* lds) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 a = *reinterpret_cast<const volatile f32x2*>(lds + ((lane * 2 + i * 128 + it * 2) & 4095));
            acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, 1.f, acc[2 * i], 0, 0, 0);
            acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, 1.f, acc[2 * i + 1], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.678f) out[lane] = s;
}

__global__ __launch_bounds__(512) void k(int roleA, int roleB, int itA, int itB, float* out, const float* g) {
    __shared__ float lds[4096 + 64];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int role = wave < 4 ? roleA : roleB;
    const int it = wave < 4 ? itA : itB;
    switch (role) {
        case MFMA: role_mfma(it, out, lane); break;
        case VALU: role_valu(it, out, lane); break;
        case LDSR: role_lds(it, out, lane, lds); break;
        case VMEM: role_vmem(it, out, lane, g); break;
        case MFMA_LDS: role_mfma_lds(it, out, lane, lds); break;
        default: break;
    }
}

static float run(int a, int b, int ia, int ib, float* out, const float* g) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, a, b, ia, ib, out, g);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, a, b, ia, ib, out, g);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 3.f;
}

int main() {
    float *out, *g;
    hipMalloc(&out, 4096);
    hipMalloc(&g, 65536 * 4 + 4096);
    hipMemset(g, 0, 65536 * 4 + 4096);
    const char* names[] = {"idle", "mfma", "valu", "lds-read", "vmem(L2)", "mfma+lds"};
    const int iters[] = {0, 4000, 32000, 16000, 8000, 8000};   // 8 MFMA / 8 FMA / 8 ds_read_b64 / 4 b128 loads / 4 reads + 8 MFMA per iteration
    std::printf("%-10s %-10s %10s %10s %10s   verdict\n", "A(0-3)", "B(4-7)", "A alone", "B alone", "together");
    const int pairs[][2] = {{MFMA, MFMA}, {MFMA, VALU}, {MFMA, LDSR}, {MFMA, VMEM}, {VALU, VALU}, {VALU, LDSR}, {MFMA_LDS, VALU}, {MFMA_LDS, MFMA_LDS}, {MFMA_LDS, LDSR}};
    for (auto& p : pairs) {
        const int a = p[0], b = p[1];
        const float ta = run(a, IDLE, iters[a], 0, out, g), tb = run(IDLE, b, 0, iters[b], out, g), tab = run(a, b, iters[a], iters[b], out, g);
        const float mx = ta > tb ? ta : tb;
        std::printf("%-10s %-10s %10.3f %10.3f %10.3f   overlap %.0f %%\n", names[a], names[b], ta, tb, tab, 100.f * (ta + tb - tab) / (ta + tb - mx));
    }
    return 0;
}
