// Rolling cross-view matching fused with the aerial L2-normalisation and the decoder concat.
//
// Reference: models.py:485-511 (and the five later copies :521-623; KITTI :789-915; Oxford :1088-1215):
//   for each roll r:  window_r = roll(x, -i_r*step, dim=1)[:, off:off+L]
//                     score_r  = sum_c g[c]*window_r[c] / (||window_r||_2 * ||g||_2)        (no epsilon)
//   ms   = stack_r score_r ;  max = max_r score_r
//   next = cat([max, F.normalize(x, p=2, dim=1)], dim=1)            (models.py:514, eps 1e-12)
// The reference materialises the broadcast descriptor map and one rolled copy of x per roll; here x is
// read from HBM exactly once per level: a block stages P pixels x C channels in LDS (row stride C+1,
// conflict-free across pixels), evaluates the (pixel, roll) dot products and window norms from LDS,
// and writes  (a) ms in NCHW (a forward output), (b) the localisation concat buffer
// [max | 7 zero pad | x/||x||] and optionally (c) the orientation concat buffer [scores | pad | x/||x||].
#include "kernels.h"

namespace ccvpe {

__global__ __launch_bounds__(256) void match_kernel(const MatchParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = p.P, C = p.C, L = p.L, R = p.R;
    const int ldx = C + 1;
    float* xs = smem;                    // [P][C+1]
    float* gs = xs + P * ldx;            // [L]
    float* sc = gs + ((L + 3) & ~3);     // [R+1][P]  (row R = squared norm of the full pixel vector)
    float* red = sc + (R + 1) * P;       // [4] block reduction scratch
    const int tid = threadIdx.x;
    const int blocks_per_sample = p.HW / P;
    const int b = blockIdx.x / blocks_per_sample;
    const int pix0 = (blockIdx.x - b * blocks_per_sample) * P;
    const float* xg = p.x + ((size_t)b * p.HW + pix0) * p.x_ld;

    // stage x tile (float4 global reads, scalar LDS writes because of the odd row stride)
    const int c4n = C >> 2;
    for (int i = tid; i < P * c4n; i += 256) {
        const int pp = i / c4n, c4 = i - pp * c4n;
        const float4 v = *reinterpret_cast<const float4*>(xg + (size_t)pp * p.x_ld + c4 * 4);
        float* d = xs + pp * ldx + c4 * 4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    // descriptor + its squared norm
    float gsq = 0.f;
    for (int i = tid; i < L; i += 256) {
        const float v = p.g[(size_t)b * p.g_ld + i];
        gs[i] = v;
        gsq = fmaf(v, v, gsq);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gsq += __shfl_xor(gsq, off);
    if ((tid & 63) == 0) red[tid >> 6] = gsq;
    __syncthreads();
    const float gnorm = sqrtf(red[0] + red[1] + red[2] + red[3]);

    // (pixel, roll) work items; item r == R is the full-vector squared norm
    for (int it = tid; it < P * (R + 1); it += 256) {
        const int pp = it & (P - 1);
        const int r = it / P;
        const float* xr = xs + pp * ldx;
        if (r == R) {
            float n2 = 0.f;
            for (int c = 0; c < C; ++c) n2 = fmaf(xr[c], xr[c], n2);
            sc[R * P + pp] = n2;
        } else {
            const int s = p.shift[r];
            float dot = 0.f, n2 = 0.f;
            const int first = min(L, C - s);
            for (int c = 0; c < first; ++c) {
                const float v = xr[c + s];
                dot = fmaf(gs[c], v, dot);
                n2 = fmaf(v, v, n2);
            }
            for (int c = first; c < L; ++c) {
                const float v = xr[c + s - C];
                dot = fmaf(gs[c], v, dot);
                n2 = fmaf(v, v, n2);
            }
            sc[r * P + pp] = dot / (sqrtf(n2) * gnorm);
        }
    }
    __syncthreads();

    // ms output, NCHW
    if (p.ms) {
        for (int it = tid; it < P * R; it += 256) {
            const int pp = it & (P - 1);
            const int r = it / P;
            p.ms[((size_t)b * R + r) * p.HW + pix0 + pp] = sc[r * P + pp];
        }
    }
    // score channels of the concat buffers
    if (p.cat_max) {
        for (int it = tid; it < P * 8; it += 256) {
            const int pp = it >> 3, ch = it & 7;
            float v = 0.f;
            if (ch == 0) {
                v = -INFINITY;
                for (int r = 0; r < R; ++r)
                    if ((p.inmax >> r) & 1u) v = fmaxf(v, sc[r * P + pp]);
            }
            p.cat_max[((size_t)b * p.HW + pix0 + pp) * p.cat_max_ld + ch] = v;
        }
    }
    if (p.cat_all) {
        for (int it = tid; it < P * p.rpad; it += 256) {
            const int pp = it / p.rpad, ch = it - pp * p.rpad;
            p.cat_all[((size_t)b * p.HW + pix0 + pp) * p.cat_all_ld + ch] = ch < R ? sc[ch * P + pp] : 0.f;
        }
    }
    // normalised x (F.normalize: x / max(||x||, 1e-12))
    for (int i = tid; i < P * c4n; i += 256) {
        const int pp = i / c4n, c4 = i - pp * c4n;
        const float inv = 1.f / fmaxf(sqrtf(sc[R * P + pp]), 1e-12f);
        const float* s = xs + pp * ldx + c4 * 4;
        const float4 v = make_float4(s[0] * inv, s[1] * inv, s[2] * inv, s[3] * inv);
        const size_t pix = (size_t)b * p.HW + pix0 + pp;
        if (p.cat_max) *reinterpret_cast<float4*>(p.cat_max + pix * p.cat_max_ld + 8 + c4 * 4) = v;
        if (p.cat_all) *reinterpret_cast<float4*>(p.cat_all + pix * p.cat_all_ld + p.rpad + c4 * 4) = v;
    }
}

int match_pixels_per_block(int HW, int C) {
    int P = 256;
    while (P > 8 && (size_t)P * (C + 1) * 4 > 40 * 1024) P >>= 1;
    while (P > HW) P >>= 1;
    return P;
}

void launch_match(const MatchParams& p, hipStream_t s) {
    size_t lds = ((size_t)p.P * (p.C + 1) + ((p.L + 3) & ~3) + (size_t)(p.R + 1) * p.P + 4) * sizeof(float);
    static size_t max_set = 0;
    if (lds > 64 * 1024 && lds > max_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(match_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        max_set = lds;
    }
    hipLaunchKernelGGL(match_kernel, dim3(p.B * (p.HW / p.P)), dim3(256), lds, s, p);
}

}  // namespace ccvpe
