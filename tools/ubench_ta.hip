// Dev microbenchmark (gfx950): rate of 16-byte-per-lane loads of an L2-resident [pixels x 192 channels] fp32 image by lane -> address
// pattern.  The MFMA operand layout of mbconv_image_kernel makes lane l read row (l & 15), 16-byte piece (l >> 4) of a 64-byte block:
// four consecutive lanes hit four different rows.  The coalesced pattern reads the same bytes with row (l >> 2), piece (l & 3).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_ta.hip -o tools/ubench_ta && tools/ubench_ta
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// KIND 0: MFMA-layout pattern into registers; 1: coalesced into registers; 2: coalesced through LDS-DMA (no registers)
template <int KIND>
__global__ __launch_bounds__(512) void k(const float* x, int iters, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int CIN = 192, PIX = 256, KCH = CIN / 16;
    const float* img = x + (size_t)(blockIdx.x % 32) * PIX * CIN;           // 196 KB per sample, 6.3 MB in all: L2 / MALL resident
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, PIX * CIN * 4, 0x00020000);
    const unsigned off = KIND == 0 ? (unsigned)(((lane & 15) * CIN + 4 * (lane >> 4)) * 4) : (unsigned)(((lane >> 2) * CIN + 4 * (lane & 3)) * 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float* my = smem + wave * (KCH * 256);                                    // 12 KB per wave (KIND 2)
    for (int it = 0; it < iters; ++it) {
        for (int mt = wave; mt < 16; mt += 8) {
            const unsigned base = off + (unsigned)(mt * 16 * CIN * 4);
            if (KIND == 2) {
#pragma unroll
                for (int kc = 0; kc < KCH; ++kc)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(my + kc * 256), 16, base, kc * 64, 0, 0);
            } else {
#pragma unroll
                for (int kc = 0; kc < KCH; ++kc) acc += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, base, kc * 64, 0));
            }
        }
    }
    if (KIND == 2) { __builtin_amdgcn_s_waitcnt(0); acc[0] = my[lane]; }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[threadIdx.x] = acc[0];
}

template <int KIND>
static void run(const char* name, const float* x, float* out) {
    const int iters = 400;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 98304, 0, x, iters, out);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 98304, 0, x, iters, out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * iters * 256 * 192 * 4;
    const double loads_per_cu = (double)iters * 16 * 12;
    std::printf("%-44s %7.3f ms  %6.2f TB/s over the chip  %6.1f B/clk/CU at 2.1 GHz  %5.1f ns per wave-load and CU\n", name, ms, bytes / ms * 1e-9, bytes / 256 / (ms * 1e-3 * 2.1e9),
                ms * 1e6 / loads_per_cu);
}

int main() {
    float *x, *out;
    (void)hipMalloc(&x, 32ull * 256 * 192 * 4); (void)hipMalloc(&out, 4096);
    (void)hipMemset(x, 0, 32ull * 256 * 192 * 4);
    run<0>("row = lane & 15, piece = lane >> 4 (MFMA)", x, out);
    run<1>("row = lane >> 2, piece = lane & 3 (coalesced)", x, out);
    run<2>("coalesced, LDS-DMA", x, out);
    return 0;
}
