// Device helpers shared by the fp32 and the bf16x3 implicit-GEMM kernels.
#pragma once
#include "kernels.h"

namespace ccvpe {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

static constexpr int BK = 32;
static constexpr int LDK = 36;  // floats per LDS row: 32 + 4 pad -> 144 B, (144/16)=9 odd => b128 reads conflict-free

// XCD-aware M-tile order: workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2), so
// give each XCD a contiguous run of M tiles - neighbouring output rows re-read the same input rows (3x3 halo)
// and then hit the same L2.  Bijective for any grid size; placement only affects speed, never results.
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nb >> 3, r = nb & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// K order of the implicit GEMM (matches pack_conv on the host): channel-group major, tap minor -
//   for each group of 32 input channels: for each filter tap: the group's (up to 4) 8-channel chunks.
// Consecutive K tiles therefore re-read the SAME 128-byte lines of the activation at neighbouring pixels (the
// 3x3 halo), which keeps the 9x tap reuse in L1/L2 instead of the Infinity Cache; with tap-major order a
// workgroup only returned to a line after walking all channels of the tap (FETCH_SIZE 8-9x the input bytes).
// chunk g -> (tap, first channel c0); exact multiply-shift divisions (verified on the host per layer).
__device__ __forceinline__ void chunk_to_tap(const ConvParams& p, int g, int& tap, int& c0) {
    if (g < p.kfull_chunks) {
        const int cg = (int)(((unsigned)g * (unsigned)p.div_4t_mul) >> 20);   // g / (4*T)
        const int idx = g - cg * p.taps4;
        tap = idx >> 2;
        c0 = cg * 32 + (idx & 3) * 8;
    } else {
        const int idx = g - p.kfull_chunks;
        tap = (idx * p.div_nc_mul) >> 8;                                        // idx / nc, nc in {1,2,3}
        c0 = p.kfull_c0 + (idx - tap * p.knc) * 8;
    }
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_SWISH) return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));   // v_rcp_f32 (1 ulp), not a ~10-instruction IEEE division
    return v;
}

// n / d for 0 <= n < 2^31 with the host-made (mul, shift) pair of conv_igemm_prepare (runtime integer division
// costs ~25 VALU instructions; the pixel shuffle of a transposed conv needs four per 16-byte store)
__device__ __forceinline__ int fast_div(int n, unsigned mul, unsigned shift) {
    return (int)((__umulhi((unsigned)n, mul) + (unsigned)n) >> shift);
}

// pixel-shuffle address of GEMM row m, column group q = (dy, dx) of a k2 s2 transposed conv
__device__ __forceinline__ int deconv_pixel(const ConvParams& p, int m, int q) {
    const int t = fast_div(m, p.fdw_mul, p.fdw_shift);
    const int x = m - t * p.W;
    const int b = fast_div(t, p.fdh_mul, p.fdh_shift);
    const int y = t - b * p.H;
    return (b * 2 * p.H + 2 * y + (q >> 1)) * (2 * p.W) + 2 * x + (q & 1);
}

// Final placement of 4 consecutive output channels (n .. n+3) of GEMM row m: residual add, k2s2
// pixel-shuffle addressing for the transposed conv, up to 3 concat destinations; 16-byte stores when legal.
__device__ __forceinline__ void emit_out4(const ConvParams& p, int m, int n, f32x4 v) {
    int opix = m, o = n;
    if (p.mode == MODE_DECONV) {
        const int c = p.deconv_cout;
        const int q = (n >= c) + (n >= 2 * c) + (n >= 3 * c);   // n / deconv_cout, n < 4 * deconv_cout
        o = n - q * c;
        opix = deconv_pixel(p, m, q);
    }
    if (p.vec_epi) {
        if (p.resid) v += *reinterpret_cast<const f32x4*>(p.resid + (size_t)m * p.resid_ld + n);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (d >= p.ndst) continue;
            const size_t e = (size_t)opix * p.dst[d].ld + p.dst[d].coff + o;
            if (p.dst[d].split) {   // bf16x3 mode: hi = bf16(v), lo = bf16(v - hi), one plane each
                typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
                bf16x4_t h, l;
#pragma unroll
                for (int k = 0; k < 4; ++k) { h[k] = (__bf16)v[k]; l[k] = (__bf16)(v[k] - (float)h[k]); }
                __bf16* hp = reinterpret_cast<__bf16*>(p.dst[d].ptr);
                *reinterpret_cast<bf16x4_t*>(hp + e) = h;
                *reinterpret_cast<bf16x4_t*>(hp + p.dst[d].plane + e) = l;
            } else {
                *reinterpret_cast<f32x4*>(p.dst[d].ptr + e) = v;
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (n + e >= p.N) break;
            float ve = v[e];
            int oe = o + e, pe = opix;
            if (p.mode == MODE_DECONV) {   // a float4 may straddle two (dy,dx) groups when cout % 4 != 0
                const int c = p.deconv_cout;
                const int q = (n + e >= c) + (n + e >= 2 * c) + (n + e >= 3 * c);
                oe = (n + e) - q * c;
                pe = deconv_pixel(p, m, q);
            } else if (p.resid) {
                ve += p.resid[(size_t)m * p.resid_ld + n + e];
            }
#pragma unroll
            for (int d = 0; d < 3; ++d)
                if (d < p.ndst) p.dst[d].ptr[(size_t)pe * p.dst[d].ld + p.dst[d].coff + oe] = ve;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Self-reducing split-K (ticket.h).  Every K slice of an output region has stored its partial sums to its slab (slab z = blockIdx.z,
// layout [z][M][N], WRITE-THROUGH stores); the workgroup calls splitk_ticket() - all threads, after its last slab store - and, when it
// drew the last of the S tickets, splitk_finish(): the region's outputs = bias + sum over z = 0 .. S - 1 of the slabs (slice order: the
// same bits whoever arrives last, the same bits as splitk_reduce_kernel), activation, emit_out4 (residual, pixel shuffle, concat taps).
// Region = `rows` runs of `cols` consecutive GEMM rows, `wpitch` rows apart, starting at GEMM row pix0; columns [n0, n0 + nch).
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool splitk_ticket(const ConvParams& p, int region, unsigned* lds_flag) {
    return ticket_arrive(p.tickets + region, 1u, (unsigned)p.splitk, lds_flag);
}

template <int NT>
__device__ __forceinline__ void splitk_finish(const ConvParams& p, int pix0, int rows, int cols, int wpitch, int n0, int nch) {
    const unsigned zstride = (unsigned)p.M * (unsigned)p.N * 4u;
    const __amdgpu_buffer_rsrc_t slab = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, (unsigned)p.splitk * zstride, 0x00020000);
    const int S = p.splitk;
    if ((p.N & 3) == 0) {
        const int c4 = nch >> 2, total = rows * cols * c4;
        constexpr int PB = 4;                                      // positions per round; 4 slabs of each in flight
        for (int it0 = threadIdx.x; it0 < total; it0 += NT * PB) {
            unsigned off[PB];
            int mm[PB], nn[PB];
            f32x4 v[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int it = it0 + u * NT;
                const int q = it % c4, px = it / c4;
                const int y = px / cols, x = px - y * cols;
                mm[u] = pix0 + y * wpitch + x; nn[u] = n0 + 4 * q;
                const bool ok = it < total && mm[u] < p.M && nn[u] < p.N;
                off[u] = ok ? (unsigned)(mm[u] * p.N + nn[u]) * 4u : 0x80000000u;
                if (!ok) mm[u] = -1;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            for (int z0 = 0; z0 < S; z0 += 4) {
                f32x4 t[PB][4];
#pragma unroll
                for (int u = 0; u < PB; ++u)
#pragma unroll
                    for (int z = 0; z < 4; ++z) t[u][z] = ld_sc1_f4(slab, (z0 + z < S && mm[u] >= 0) ? off[u] + (unsigned)(z0 + z) * zstride : 0x80000000u);
#pragma unroll
                for (int u = 0; u < PB; ++u)
#pragma unroll
                    for (int z = 0; z < 4; ++z) v[u] += t[u][z];    // (slices past S: + 0)
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                if (mm[u] < 0) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[u][e] = apply_act(v[u][e] + p.bias[nn[u] + e], p.act);
                emit_out4(p, mm[u], nn[u], v[u]);
            }
        }
    } else {   // widths that are not a multiple of 4: element-wise (rare: only the narrow descriptor layers)
        const int total = rows * cols * nch;
        for (int it = threadIdx.x; it < total; it += NT) {
            const int c = it % nch, px = it / nch;
            const int y = px / cols, x = px - y * cols;
            const int m = pix0 + y * wpitch + x, n = n0 + c;
            if (m >= p.M || n >= p.N || (c & 3) != 0) continue;    // one thread per group of four columns (emit_out4 places a quad)
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            for (int z = 0; z < S; ++z)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < p.N) v[e] += ld_sc1(p.partial + ((size_t)z * p.M + m) * p.N + n + e);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e] + (n + e < p.N ? p.bias[n + e] : 0.f), p.act);
            emit_out4(p, m, n, v);
        }
    }
}

}  // namespace ccvpe
