// Minimal reproducer for the bf16x3 two-stream nondeterminism: a hardware hazard between packed fp32
// VALU instructions with operand-half selection (VOP3P op_sel) and v_mfma_f32_16x16x32_bf16 issued by ANOTHER wave.
//
// Observation in the full model: match_kernel (hipcc's SLP vectoriser turns its dot / norm accumulators into
// v_pk_fma_f32 ... op_sel:[0,1,0]) and conv_wino_kernel (hand-written v_pk_add_f32 with op_sel / neg) returned wrong
// values in lanes 48-63 while a DIFFERENT kernel issuing bf16 MFMAs shared their SIMDs through the second HIP stream;
// their inputs were verified identical in stream order, a scalar-FMA build of match_kernel was unaffected, and the same
// schedule with exact-fp32 MFMA neighbours is bit-reproducible.
//
// This program isolates the pair.  A "victim" kernel runs the same recurrence twice per lane: once with ONE packed
// instruction per step (explicit asm, a given op_sel / op_sel_hi / neg form) and once with two scalar fp32 instructions
// (bit-identical IEEE results expected); `iters` steps with every lane active, then a divergent tail (lane l runs
// 8 * (l >> 3) more steps) like match_kernel's loops.  An "aggressor" kernel spins on one MFMA type on a second stream.
// Output: per (aggressor, victim form) how many lane results disagreed and in which 16-lane group.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/repro_pk_mfma tools/repro_pk_mfma.hip && tools/repro_pk_mfma [reps]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                                         \
    do {                                                                                                 \
        hipError_t e_ = (x);                                                                             \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } \
    } while (0)

enum { A_BF16_16, A_BF16_32, A_F16_16, A_F16_32, A_I8_16, A_F32_16, A_F32_32, A_VALU, A_NONE, A_COUNT };
static const char* A_NAME[A_COUNT] = {"bf16 16x16x32", "bf16 32x32x16", "f16 16x16x32", "f16 32x32x16", "i8 16x16x64",
                                      "f32 16x16x4", "f32 32x32x2", "VALU only", "nothing"};

template <int TYPE>
__global__ __launch_bounds__(256) void aggressor(float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    f16x8 ha, hb;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i));
        ha[i] = (_Float16)(0.001f * (lane + i)); hb[i] = (_Float16)(0.002f * (lane - i));
    }
    const i32x4 ia = {lane, lane * 3, lane * 5, lane * 7}, ib = {lane * 11, lane * 13, 1, 2};
    f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
    i32x4 ci = {0, 0, 0, 0};
    f32x16 c16;
    for (int i = 0; i < 16; ++i) c16[i] = 0.f;
    float s = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (TYPE == A_BF16_16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4, 0, 0, 0);
        } else if (TYPE == A_BF16_32) {
#pragma unroll
            for (int k = 0; k < 4; ++k) c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16, 0, 0, 0);
        } else if (TYPE == A_F16_16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c4, 0, 0, 0);
        } else if (TYPE == A_F16_32) {
#pragma unroll
            for (int k = 0; k < 4; ++k) c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c16, 0, 0, 0);
        } else if (TYPE == A_I8_16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) ci = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, ci, 0, 0, 0);
        } else if (TYPE == A_F32_16) {
#pragma unroll
            for (int k = 0; k < 4; ++k) c4 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a[0], (float)b[0], c4, 0, 0, 0);
        } else if (TYPE == A_F32_32) {
#pragma unroll
            for (int k = 0; k < 2; ++k) c16 = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[0], (float)b[0], c16, 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 32; ++k) s = __builtin_fmaf(s, 0.999f, 0.001f * lane);
        }
    }
    sink[blockIdx.x * 256 + threadIdx.x] = c4[0] + c4[1] + c4[2] + c4[3] + c16[0] + c16[5] + s + (float)(ci[0] + ci[3]);
}

// ---- victims ---------------------------------------------------------------------------------------------------
// VOP3P semantics: result.lo = op(src0[op_sel[0]], src1[op_sel[1]], src2[op_sel[2]]), result.hi likewise with op_sel_hi
// (index 0 = low half, 1 = high half); neg_lo / neg_hi negate the operands of the low / high result.
// OPC 0: fma(p, m, a)   1: add(p, a)   2: mul(p, m)      (p is both src0 and the destination)
#define VICTIM(NAME, OPC, ASM, S0, S1, S2, H0, H1, H2, NL1, NH1, NL0, NH0)                                                  \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, int iters, const float* coef) {                    \
        const int gid = blockIdx.x * 256 + threadIdx.x;                                                            \
        const int lane = threadIdx.x & 63;                                                                         \
        const float m[2] = {coef[lane], coef[64 + lane]};                                                          \
        const float a[2] = {coef[128 + lane], coef[192 + lane]};                                                   \
        f32x2 p = {a[0], a[1]};                                                                                    \
        float s[2] = {a[0], a[1]};                                                                                 \
        const f32x2 pm = {m[0], m[1]}, pa = {a[0], a[1]};                                                          \
        const int total = iters + 8 * (lane >> 3);                                                                 \
        for (int it = 0; it < total; ++it) {                                                                       \
            if (OPC == 0) asm volatile(ASM : "+v"(p) : "v"(pm), "v"(pa));                                          \
            else if (OPC == 1) asm volatile(ASM : "+v"(p) : "v"(pa));                                              \
            else asm volatile(ASM : "+v"(p) : "v"(pm));                                                            \
            float lo, hi;                                                                                          \
            if (OPC == 0) { lo = __builtin_fmaf(s[S0], m[S1], a[S2]); hi = __builtin_fmaf(s[H0], m[H1], a[H2]); }  \
            else if (OPC == 1) { lo = (NL0 ? -s[S0] : s[S0]) + (NL1 ? -a[S1] : a[S1]); hi = (NH0 ? -s[H0] : s[H0]) + (NH1 ? -a[H1] : a[H1]); } \
            else { lo = s[S0] * m[S1]; hi = s[H0] * m[H1]; }                                                       \
            asm volatile("" : "+v"(lo), "+v"(hi));                                                                 \
            s[0] = lo; s[1] = hi;                                                                                  \
        }                                                                                                          \
        const float px = p[0], py = p[1];                                                                          \
        out[gid] = (__builtin_bit_cast(unsigned, px) != __builtin_bit_cast(unsigned, s[0]) ? 1u : 0u) |            \
                   (__builtin_bit_cast(unsigned, py) != __builtin_bit_cast(unsigned, s[1]) ? 2u : 0u);             \
    }

VICTIM(v_fma_plain, 0, "v_pk_fma_f32 %0, %0, %1, %2", 0, 0, 0, 1, 1, 1, 0, 0, 0, 0)
VICTIM(v_fma_s100, 0, "v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0]", 1, 0, 0, 1, 1, 1, 0, 0, 0, 0)
VICTIM(v_fma_s010, 0, "v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0]", 0, 1, 0, 1, 1, 1, 0, 0, 0, 0)
VICTIM(v_fma_s001, 0, "v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1]", 0, 0, 1, 1, 1, 1, 0, 0, 0, 0)
VICTIM(v_fma_h011, 0, "v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[0,1,1]", 0, 0, 0, 0, 1, 1, 0, 0, 0, 0)
VICTIM(v_fma_h101, 0, "v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]", 0, 0, 0, 1, 0, 1, 0, 0, 0, 0)
VICTIM(v_fma_h110, 0, "v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]", 0, 0, 0, 1, 1, 0, 0, 0, 0, 0)
VICTIM(v_fma_swap, 0, "v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,1] op_sel_hi:[0,0,0]", 1, 1, 1, 0, 0, 0, 0, 0, 0, 0)
VICTIM(v_mul_plain, 2, "v_pk_mul_f32 %0, %0, %1", 0, 0, 0, 1, 1, 1, 0, 0, 0, 0)
VICTIM(v_mul_s01, 2, "v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]", 0, 1, 0, 1, 1, 1, 0, 0, 0, 0)
VICTIM(v_mul_h10, 2, "v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]", 0, 0, 0, 1, 0, 1, 0, 0, 0, 0)
VICTIM(v_add_plain, 1, "v_pk_add_f32 %0, %0, %1", 0, 0, 0, 1, 1, 1, 0, 0, 0, 0)
VICTIM(v_add_s01, 1, "v_pk_add_f32 %0, %0, %1 op_sel:[0,1]", 0, 1, 0, 1, 1, 1, 0, 0, 0, 0)
VICTIM(v_add_h10, 1, "v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]", 0, 0, 0, 1, 0, 1, 0, 0, 0, 0)
VICTIM(v_add_neg, 1, "v_pk_add_f32 %0, %0, %1 neg_lo:[0,1]", 0, 0, 0, 1, 1, 1, 1, 0, 0, 0)
// the forms of conv_wino_kernel's column pass
VICTIM(v_add_wino1, 1, "v_pk_add_f32 %0, %0, %1 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]", 0, 0, 0, 1, 0, 1, 1, 0, 0, 0)
// rounds 1-2: lo = p.lo - a.hi (op_sel[1] = 1: the hazard encoding), hi = -p.hi + a.hi (neg_hi negates SRC0: NH0)
VICTIM(v_add_wino2, 1, "v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]", 0, 1, 0, 1, 1, 1, 1, 0, 0, 1)
// round 3: the crossed operand sits in src0 - lo = -p.hi + a.lo, hi = p.hi - a.hi
VICTIM(v_add_wino3, 1, "v_pk_add_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]", 1, 0, 0, 1, 1, 1, 0, 1, 1, 0)

typedef void (*victim_fn)(unsigned*, int, const float*);
struct Victim { const char* name; victim_fn fn; };
static const Victim VICTIMS[] = {
    {"pk_fma", v_fma_plain}, {"pk_fma op_sel:[1,0,0]", v_fma_s100}, {"pk_fma op_sel:[0,1,0]", v_fma_s010}, {"pk_fma op_sel:[0,0,1]", v_fma_s001},
    {"pk_fma op_sel_hi:[0,1,1]", v_fma_h011}, {"pk_fma op_sel_hi:[1,0,1]", v_fma_h101}, {"pk_fma op_sel_hi:[1,1,0]", v_fma_h110},
    {"pk_fma swapped halves", v_fma_swap}, {"pk_mul", v_mul_plain}, {"pk_mul op_sel:[0,1]", v_mul_s01}, {"pk_mul op_sel_hi:[1,0]", v_mul_h10},
    {"pk_add", v_add_plain}, {"pk_add op_sel:[0,1]", v_add_s01}, {"pk_add op_sel_hi:[1,0]", v_add_h10}, {"pk_add neg_lo:[0,1]", v_add_neg},
    {"pk_add wino form 1", v_add_wino1}, {"pk_add wino form 2 (r1-2)", v_add_wino2}, {"pk_add wino form 2 (r3)", v_add_wino3},
};

static void launch_aggr(int type, float* sink, int blocks, int iters, hipStream_t s) {
    switch (type) {
        case A_BF16_16: hipLaunchKernelGGL(aggressor<A_BF16_16>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
        case A_BF16_32: hipLaunchKernelGGL(aggressor<A_BF16_32>, dim3(blocks), dim3(256), 0, s, sink, iters / 2); break;
        case A_F16_16: hipLaunchKernelGGL(aggressor<A_F16_16>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
        case A_F16_32: hipLaunchKernelGGL(aggressor<A_F16_32>, dim3(blocks), dim3(256), 0, s, sink, iters / 2); break;
        case A_I8_16: hipLaunchKernelGGL(aggressor<A_I8_16>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
        case A_F32_16: hipLaunchKernelGGL(aggressor<A_F32_16>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
        case A_F32_32: hipLaunchKernelGGL(aggressor<A_F32_32>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
        case A_VALU: hipLaunchKernelGGL(aggressor<A_VALU>, dim3(blocks), dim3(256), 0, s, sink, iters / 4); break;
        default: break;
    }
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? std::atoi(argv[1]) : 3;
    const int ablocks = 1024, vblocks = 4096;   // aggressor: 4 workgroups per CU; the victim's waves fill the other wave slots
    float* sink; unsigned* out; float* coef;
    CHECK(hipMalloc((void**)&sink, ablocks * 256 * sizeof(float)));
    CHECK(hipMalloc((void**)&out, vblocks * 256 * sizeof(unsigned)));
    CHECK(hipMalloc((void**)&coef, 256 * sizeof(float)));
    std::vector<float> hc(256);
    for (int i = 0; i < 64; ++i) { hc[i] = 0.99990f + 1e-6f * i; hc[64 + i] = 0.99985f - 1e-6f * i; hc[128 + i] = 1e-3f * (i + 1); hc[192 + i] = -7e-4f * (i + 3); }
    CHECK(hipMemcpy(coef, hc.data(), 256 * sizeof(float), hipMemcpyHostToDevice));
    hipStream_t sa, sv;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    std::vector<unsigned> h(vblocks * 256);
    const int nv = (int)(sizeof(VICTIMS) / sizeof(VICTIMS[0]));
    int rc = 0;
    std::printf("%-26s", "victim \\ neighbour");
    for (int a = 0; a < A_COUNT; ++a) std::printf(" %13s", A_NAME[a]);
    std::printf("\n");
    for (int v = 0; v < nv; ++v) {
        std::printf("%-26s", VICTIMS[v].name);
        for (int a = 0; a < A_COUNT; ++a) {
            long bad = 0, g3 = 0;
            for (int rep = 0; rep < reps; ++rep) {
                CHECK(hipMemset(out, 0xff, vblocks * 256 * sizeof(unsigned)));
                CHECK(hipDeviceSynchronize());
                launch_aggr(a, sink, ablocks, 60000, sa);   // outlives the victim launch
                hipLaunchKernelGGL(VICTIMS[v].fn, dim3(vblocks), dim3(256), 0, sv, out, 20000, (const float*)coef);
                CHECK(hipGetLastError());
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(h.data(), out, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
                for (size_t i = 0; i < h.size(); ++i) {
                    if (h[i] > 3u) { std::fprintf(stderr, "victim kernel did not write lane %zu (0x%x)\n", i, h[i]); return 1; }
                    if (h[i]) { ++bad; if ((i & 63) >= 48) ++g3; }
                }
            }
            if (bad) { std::printf(" %8ld/%-4s", bad, g3 == bad ? "L48+" : "mix"); rc = 2; }
            else std::printf(" %13s", "0");
            std::fflush(stdout);
        }
        std::printf("\n");
    }
    std::printf("(cells: lane results that differ from the scalar chain out of %zu; L48+ = every one of them in lanes 48-63)\n", (size_t)reps * h.size());
    return rc;
}
