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

#include <cstdlib>

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

// ------------------------------------------------------------------------------------------------
// Small-C form (levels 5-6: C = 32, 40, 80): one thread owns one pixel and keeps its C channels in
// registers; the descriptor is rolled instead of x:  score_r = sum_j x[j] * gpad[(j - s_r) mod C], with gpad
// (g zero padded to C) stored twice back to back so that gg[C - s_r + j], j = 0..C-1, is one contiguous,
// wave-uniform run (scalar / broadcast loads).  When L < C the window norm uses the doubled 0/1 mask the same
// way; when L == C it is the roll-invariant full norm.  x is read once, no LDS, no per-pixel modulo.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void match_prep_body(const MatchParams& p, float* gg, float* red) {
    // gg[b] = [ gpad | gpad | mask | mask | ||g|| ]  (4C + 1 floats per sample)
    const int b = blockIdx.x, C = p.C, L = p.L;
    float* o = gg + (size_t)b * (4 * C + 4);
    float gsq = 0.f;
    for (int i = threadIdx.x; i < C; i += 256) {
        const float v = i < L ? p.g[(size_t)b * p.g_ld + i] : 0.f;
        o[i] = v; o[C + i] = v;
        const float m = i < L ? 1.f : 0.f;
        o[2 * C + i] = m; o[3 * C + i] = m;
        gsq = fmaf(v, v, gsq);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gsq += __shfl_xor(gsq, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gsq;
    __syncthreads();
    if (threadIdx.x == 0) o[4 * C] = sqrtf(red[0] + red[1] + red[2] + red[3]);
}
__global__ __launch_bounds__(256) void match_prep_kernel(const MatchParams p, float* gg) {
    __shared__ float red[4];
    match_prep_body(p, gg, red);
}

// RG = 1: a thread per pixel walks all rolls.  RG = 4 (latency plans, round 4): the four WAVES of a workgroup share 64 pixels and take
// every fourth roll each (the roll - and with it the address of the rolled descriptor - stays wave-uniform: scalar loads) - level 5 at
// batch 1 is 16384 pixels = one wave per CU walking 20 rolls x 80 channels (24 us); the maximum over the rolls meets in LDS (order-free),
// the descriptor copy is shared between the four waves.
template <int C, int RG>
__global__ __launch_bounds__(256) void match_small_kernel(const MatchParams p, const float* __restrict__ gg) {
    __shared__ float bests[RG > 1 ? 256 : 1];
    const int b = blockIdx.y;
    const int rg = RG == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int pix_raw = RG == 1 ? blockIdx.x * 256 + (int)threadIdx.x : blockIdx.x * 64 + ((int)threadIdx.x & 63);
    if (RG == 1 && pix_raw >= p.HW) return;
    const bool live = pix_raw < p.HW;          // (RG > 1: everybody reaches the barrier; a lane past the image works on the last pixel and stores nothing)
    const int pix = live ? pix_raw : p.HW - 1;
    const float* xg = p.x + ((size_t)b * p.HW + pix) * p.x_ld;
    float x[C];
#pragma unroll
    for (int c4 = 0; c4 < C / 4; ++c4) {
        const float4 v = *reinterpret_cast<const float4*>(xg + c4 * 4);
        x[c4 * 4 + 0] = v.x; x[c4 * 4 + 1] = v.y; x[c4 * 4 + 2] = v.z; x[c4 * 4 + 3] = v.w;
    }
    float n2full = 0.f;
#pragma unroll
    for (int j = 0; j < C; ++j) n2full = fmaf(x[j], x[j], n2full);
    const float* gb = gg + (size_t)b * (4 * C + 4);
    const float gnorm = gb[4 * C];
    const bool full = p.L == C;
    float best = -INFINITY;
    for (int r = rg; r < p.R; r += RG) {
        const int base = C - p.shift[r];          // wave-uniform (RG = 1)
        const float* gr = gb + base;
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) dot = fmaf(x[j], gr[j], dot);
        float n2 = n2full;
        if (!full) {
            const float* mr = gb + 2 * C + base;
            n2 = 0.f;
#pragma unroll
            for (int j = 0; j < C; ++j) n2 = fmaf(x[j] * x[j], mr[j], n2);
        }
        const float sc = dot / (sqrtf(n2) * gnorm);
        if (p.ms && live) p.ms[((size_t)b * p.R + r) * p.HW + pix] = sc;
        if ((p.inmax >> r) & 1u) best = fmaxf(best, sc);
    }
    if (RG > 1) {
        bests[threadIdx.x] = best;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < RG; ++w) best = fmaxf(best, bests[w * 64 + (threadIdx.x & 63)]);
        if (!live) return;
    }
    const float inv = 1.f / fmaxf(sqrtf(n2full), 1e-12f);
    float* o = p.cat_max + ((size_t)b * p.HW + pix) * p.cat_max_ld;
    if (rg == 0) {
        *reinterpret_cast<float4*>(o) = make_float4(best, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(o + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int c4 = 0; c4 < C / 4; ++c4)
        if (RG == 1 || (c4 & (RG - 1)) == rg)
            *reinterpret_cast<float4*>(o + 8 + c4 * 4) = make_float4(x[c4 * 4] * inv, x[c4 * 4 + 1] * inv, x[c4 * 4 + 2] * inv, x[c4 * 4 + 3] * inv);
}

// ------------------------------------------------------------------------------------------------
// MFMA form for the wide levels (C >= 128; models.py:485-511 with C = 1280 .. 160, KITTI 2048 .. 128): the (pixel, roll)
// dot products of match_kernel are one contraction  S[p][r] = sum_j x[p][j] * Gm[j][r],  Gm[j][r] = g[(j - s_r) mod C] inside the
// window (0 outside), and the window norms are  N2[p][r] = sum_j x[p][j]^2 * Mk[j][r]  with the 0/1 window mask; column 31 of Mk is
// all ones (the full-vector norm F.normalize needs).  match_kernel evaluates them with two LDS reads per FMA and is LDS-bound
// (1.2 TB/s on levels 1-4); here a workgroup stages 16 pixels x C channels once (x is still read from HBM exactly once), its four
// waves split C between them, each k-step of 16 channels is one ds_read_b128 of x feeding 8 v_mfma_f32_16x16x4_f32 (2 roll tiles x
// {x, x^2}... 16 with the norms), and the partial sums meet in LDS.  Gm / Mk are built per sample by match_mfma_prep_kernel in
// B-fragment order [16-channel chunk][roll tile][lane][4] (one 16-byte load per lane and chunk).
// ------------------------------------------------------------------------------------------------
typedef float f32x4m __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void match_mfma_prep_body(const MatchParams& p, float* gm, float* red) {
    // gm[b]: [C/16 chunks][2 tiles][64 lanes][4] for G, then the same for the mask, then ||g||
    const int b = blockIdx.x, C = p.C, L = p.L;
    const size_t per = (size_t)C * 32;
    float* G = gm + (size_t)b * (2 * per + 4);
    float* M = G + per;
    const float* g = p.g + (size_t)b * p.g_ld;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < C * 32; i += gridDim.y * 256) {
        const int e = i & 3, lane = (i >> 2) & 63, t = (i >> 8) & 1, chunk = i >> 9;
        const int j = chunk * 16 + 4 * (lane >> 4) + e;       // input channel (k index of the fragment)
        const int r = t * 16 + (lane & 15);                   // roll column
        float gv = 0.f, mv = 0.f;
        if (r < p.R) {
            int c = j - p.shift[r];
            if (c < 0) c += C;
            if (c < L) { gv = g[c]; mv = 1.f; }
        } else if (r == 31) {
            mv = 1.f;
        }
        G[i] = gv; M[i] = mv;
    }
    if (blockIdx.y != 0) return;
    float gsq = 0.f;
    for (int i = threadIdx.x; i < L; i += 256) gsq = fmaf(g[i], g[i], gsq);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gsq += __shfl_xor(gsq, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gsq;
    __syncthreads();
    if (threadIdx.x == 0) G[2 * per] = sqrtf(red[0] + red[1] + red[2] + red[3]);
}
__global__ __launch_bounds__(256) void match_mfma_prep_kernel(const MatchParams p, float* gm) {
    __shared__ float red[4];
    match_mfma_prep_body(p, gm, red);
}
// The preparation of all matching levels in ONE launch (round 4): they depend on the ground descriptor only, and as six launches of a
// few microseconds each they were 47 us of the ground stream's issue order at batch 1.  grid (B, 8, levels); form[z]: 0 nothing to
// prepare (LDS form), 1 rolled descriptor of the register form, 2 Gm / Mk of the MFMA form.
struct MatchPrepAll { MatchParams p[6]; int form[6]; };
__global__ __launch_bounds__(256) void match_prep_all_kernel(const MatchPrepAll a) {
    __shared__ float red[4];
    const int z = blockIdx.z;
    if (a.form[z] == 2) match_mfma_prep_body(a.p[z], a.p[z].gg_scratch, red);
    else if (a.form[z] == 1 && blockIdx.y == 0) match_prep_body(a.p[z], a.p[z].gg_scratch, red);
}

// NW = 4 waves per workgroup, or 16 in latency plans (round 4): level 1 at batch 1 is FOUR workgroups whose waves walked 20 chunks of
// 1280 channels with one trip to memory per chunk (28 us); sixteen waves with their <= 5 chunks' Gm / Mk fragments requested at once
// make it one trip.
template <int NW>
__global__ __launch_bounds__(64 * NW) void match_mfma_kernel(const MatchParams p, const float* __restrict__ gm) {
    constexpr int NT = 64 * NW;
    constexpr int UB = 5;                  // chunks whose fragments a wave requests at once
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int C = p.C, R = p.R;
    const int ldx = C + 4;                 // 16-byte rows, 4 mod 32 floats (C % 32 == 0): conflict-free b128 reads across the 16 pixels
    float* xs = smem;                      // [16][C + 4]
    float* red = xs + 16 * ldx;            // [NW waves][2 (S, N2)][2 tiles][64 lanes][4]
    float* sc = red + NW * 2 * 2 * 256;    // [32][16] final scores (row 31: squared full norm)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blocks_per_sample = p.HW >> 4;
    const int b = blockIdx.x / blocks_per_sample;
    const int pix0 = (blockIdx.x - b * blocks_per_sample) << 4;
    const float* xg = p.x + ((size_t)b * p.HW + pix0) * p.x_ld;

    const int c4n = C >> 2;
    for (int i = tid; i < 16 * c4n; i += NT) {
        const int pp = i / c4n, c4 = i - pp * c4n;
        *reinterpret_cast<f32x4m*>(xs + pp * ldx + c4 * 4) = *reinterpret_cast<const f32x4m*>(xg + (size_t)pp * p.x_ld + c4 * 4);
    }

    const size_t per = (size_t)C * 32;
    const float* G = gm + (size_t)b * (2 * per + 4);
    const float* M = G + per;
    const float gnorm = G[2 * per];
    // wave w owns chunks w, w + NW, ... of 16 channels; the Gm / Mk fragments of UB chunks are requested together - and, the first
    // time, before the barrier that ends the staging of x (a chunk past the last re-reads the last one and multiplies it with zeros)
    f32x4m accS[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, accN[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const float* ap = xs + (lane & 15) * ldx + 4 * (lane >> 4);
    const int nchunks = C >> 4;
    for (int ch0 = wave; ch0 < nchunks; ch0 += NW * UB) {
        f32x4m g0[UB], g1[UB], m0[UB], m1[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int ch = min(ch0 + u * NW, nchunks - 1);
            g0[u] = *reinterpret_cast<const f32x4m*>(G + ((size_t)(ch * 2 + 0) * 64 + lane) * 4);
            g1[u] = *reinterpret_cast<const f32x4m*>(G + ((size_t)(ch * 2 + 1) * 64 + lane) * 4);
            m0[u] = *reinterpret_cast<const f32x4m*>(M + ((size_t)(ch * 2 + 0) * 64 + lane) * 4);
            m1[u] = *reinterpret_cast<const f32x4m*>(M + ((size_t)(ch * 2 + 1) * 64 + lane) * 4);
        }
        if (ch0 == wave) __syncthreads();   // (uniform: every wave's first batch) x is staged
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int ch = ch0 + u * NW;
            f32x4m a = *reinterpret_cast<const f32x4m*>(ap + min(ch, nchunks - 1) * 16);
            if (ch >= nchunks) a = f32x4m{0.f, 0.f, 0.f, 0.f};
            const f32x4m a2 = a * a;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                accS[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], g0[u][e], accS[0], 0, 0, 0);
                accS[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], g1[u][e], accS[1], 0, 0, 0);
                accN[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[e], m0[u][e], accN[0], 0, 0, 0);
                accN[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[e], m1[u][e], accN[1], 0, 0, 0);
            }
        }
    }
    if (wave >= nchunks) __syncthreads();   // (a wave without chunks never entered the loop: its share of the staging barrier)
    // partial sums -> LDS; accumulator element i of lane l is (pixel 4*(l>>4) + i, roll 16*t + (l&15))
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        *reinterpret_cast<f32x4m*>(red + (((wave * 2 + 0) * 2 + t) * 64 + lane) * 4) = accS[t];
        *reinterpret_cast<f32x4m*>(red + (((wave * 2 + 1) * 2 + t) * 64 + lane) * 4) = accN[t];
    }
    __syncthreads();
    {   // thread = (tile t, lane l, element i) of the 2 x 64 x 4 = 512 (pixel, roll) sums: two per thread
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = tid + k * 256;
            if (NW > 4 && tid >= 256) break;
            const int t = idx >> 8, l = (idx >> 2) & 63, i = idx & 3;
            float s = 0.f, n2 = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                s += red[(((w * 2 + 0) * 2 + t) * 64 + l) * 4 + i];
                n2 += red[(((w * 2 + 1) * 2 + t) * 64 + l) * 4 + i];
            }
            const int pp = 4 * (l >> 4) + i, r = 16 * t + (l & 15);
            sc[r * 16 + pp] = r == 31 ? n2 : s / (sqrtf(n2) * gnorm);     // no epsilon, as models.py:494
        }
    }
    __syncthreads();

    if (p.ms) {
        for (int it = tid; it < 16 * R; it += NT) {
            const int pp = it & 15, r = it >> 4;
            p.ms[((size_t)b * R + r) * p.HW + pix0 + pp] = sc[r * 16 + pp];
        }
    }
    if (p.cat_max) {
        for (int it = tid; it < 16 * 8; it += NT) {
            const int pp = it >> 3, ch = it & 7;
            float v = 0.f;
            if (ch == 0) {
                v = -INFINITY;
                for (int r = 0; r < R; ++r)
                    if ((p.inmax >> r) & 1u) v = fmaxf(v, sc[r * 16 + pp]);
            }
            p.cat_max[((size_t)b * p.HW + pix0 + pp) * p.cat_max_ld + ch] = v;
        }
    }
    if (p.cat_all) {
        for (int it = tid; it < 16 * p.rpad; it += NT) {
            const int pp = it / p.rpad, ch = it - pp * p.rpad;
            p.cat_all[((size_t)b * p.HW + pix0 + pp) * p.cat_all_ld + ch] = ch < R ? sc[ch * 16 + pp] : 0.f;
        }
    }
    for (int i = tid; i < 16 * c4n; i += NT) {
        const int pp = i / c4n, c4 = i - pp * c4n;
        const float inv = 1.f / fmaxf(sqrtf(sc[31 * 16 + pp]), 1e-12f);
        const f32x4m v = *reinterpret_cast<const f32x4m*>(xs + pp * ldx + c4 * 4) * inv;
        const size_t pix = (size_t)b * p.HW + pix0 + pp;
        if (p.cat_max) *reinterpret_cast<f32x4m*>(p.cat_max + pix * p.cat_max_ld + 8 + c4 * 4) = v;
        if (p.cat_all) *reinterpret_cast<f32x4m*>(p.cat_all + pix * p.cat_all_ld + p.rpad + c4 * 4) = v;
    }
}

bool match_mfma_supported(const MatchParams& p) {
    return p.gg_scratch != nullptr && p.C >= 128 && p.C % 32 == 0 && p.C <= 2048 && p.R <= 31 && p.HW % 16 == 0 && p.x_ld % 4 == 0 &&
           (p.cat_max == nullptr || p.cat_max_ld % 4 == 0) && (p.cat_all == nullptr || (p.cat_all_ld % 4 == 0 && p.rpad % 4 == 0));
}
size_t match_scratch_floats(int C) { return C >= 128 ? (size_t)2 * C * 32 + 4 : (size_t)4 * C + 4; }

bool match_small_supported(const MatchParams& p) {
    return (p.C == 32 || p.C == 40 || p.C == 80) && p.cat_all == nullptr && p.cat_max != nullptr && p.x_ld % 4 == 0;
}

int match_pixels_per_block(int HW, int C) {
    int P = 256;
    while (P > 8 && (size_t)P * (C + 1) * 4 > 40 * 1024) P >>= 1;
    while (P > HW) P >>= 1;
    return P;
}

static bool match_use_mfma(const MatchParams& p) {
    static const bool no_mfma = getenv("CCVPE_MATCH_MFMA") && std::atoi(getenv("CCVPE_MATCH_MFMA")) == 0;
    return !no_mfma && match_mfma_supported(p);
}

void launch_match_prep(const MatchParams& p, hipStream_t s) {
    if (match_use_mfma(p)) CCVPE_LAUNCH(match_mfma_prep_kernel, dim3(p.B, 8), dim3(256), 0, s, p, p.gg_scratch);
    else if (p.gg_scratch && match_small_supported(p)) CCVPE_LAUNCH(match_prep_kernel, dim3(p.B), dim3(256), 0, s, p, p.gg_scratch);
}

void launch_match_prep_all(const MatchParams* ps, int n, hipStream_t s) {
    MatchPrepAll a{};
    bool any = false;
    for (int i = 0; i < n && i < 6; ++i) {
        a.p[i] = ps[i];
        a.form[i] = match_use_mfma(ps[i]) ? 2 : (ps[i].gg_scratch && match_small_supported(ps[i])) ? 1 : 0;
        any = any || a.form[i] != 0;
    }
    if (any) CCVPE_LAUNCH(match_prep_all_kernel, dim3(ps[0].B, 8, std::min(n, 6)), dim3(256), 0, s, a);
}

void launch_match(const MatchParams& p, hipStream_t s) {
    if (match_use_mfma(p)) {
        if (!p.prep_done) CCVPE_LAUNCH(match_mfma_prep_kernel, dim3(p.B, 8), dim3(256), 0, s, p, p.gg_scratch);
        const size_t lds = ((size_t)16 * (p.C + 4) + 4 * 2 * 2 * 256 + 32 * 16) * sizeof(float);
        const size_t lds16 = ((size_t)16 * (p.C + 4) + 16 * 2 * 2 * 256 + 32 * 16) * sizeof(float);
        if (!p.no_wide && p.C >= 640 && p.B * (p.HW >> 4) <= 128 && lds16 <= 158 * 1024) {   // latency plans, >= 10 chunks per wave of four: sixteen waves per workgroup (C = 320: 11.7 -> 14.4 us, left with four)
            static LdsAttr attr16;
            ensure_dynamic_lds(attr16, reinterpret_cast<const void*>(match_mfma_kernel<16>), lds16);
            CCVPE_LAUNCH(match_mfma_kernel<16>, dim3(p.B * (p.HW >> 4)), dim3(1024), lds16, s, p, (const float*)p.gg_scratch);
            return;
        }
        static LdsAttr attr;
        if (lds > 64 * 1024) ensure_dynamic_lds(attr, reinterpret_cast<const void*>(match_mfma_kernel<4>), lds);
        CCVPE_LAUNCH(match_mfma_kernel<4>, dim3(p.B * (p.HW >> 4)), dim3(256), lds, s, p, (const float*)p.gg_scratch);
        return;
    }
    if (p.gg_scratch && match_small_supported(p)) {
        if (!p.prep_done) CCVPE_LAUNCH(match_prep_kernel, dim3(p.B), dim3(256), 0, s, p, p.gg_scratch);
        if (!p.no_wide && (long long)p.B * ((p.HW + 255) / 256) <= 256) {   // latency plans: four lanes per pixel
            dim3 grid((p.HW + 63) / 64, p.B);
            if (p.C == 32) CCVPE_LAUNCH((match_small_kernel<32, 4>), grid, dim3(256), 0, s, p, (const float*)p.gg_scratch);
            else if (p.C == 40) CCVPE_LAUNCH((match_small_kernel<40, 4>), grid, dim3(256), 0, s, p, (const float*)p.gg_scratch);
            else CCVPE_LAUNCH((match_small_kernel<80, 4>), grid, dim3(256), 0, s, p, (const float*)p.gg_scratch);
            return;
        }
        dim3 grid((p.HW + 255) / 256, p.B);
        if (p.C == 32) CCVPE_LAUNCH((match_small_kernel<32, 1>), grid, dim3(256), 0, s, p, (const float*)p.gg_scratch);
        else if (p.C == 40) CCVPE_LAUNCH((match_small_kernel<40, 1>), grid, dim3(256), 0, s, p, (const float*)p.gg_scratch);
        else CCVPE_LAUNCH((match_small_kernel<80, 1>), grid, dim3(256), 0, s, p, (const float*)p.gg_scratch);
        return;
    }
    size_t lds = ((size_t)p.P * (p.C + 1) + ((p.L + 3) & ~3) + (size_t)(p.R + 1) * p.P + 4) * sizeof(float);
    static LdsAttr attr;
    if (lds > 64 * 1024) ensure_dynamic_lds(attr, reinterpret_cast<const void*>(match_kernel), lds);
    CCVPE_LAUNCH(match_kernel, dim3(p.B * (p.HW / p.P)), dim3(256), lds, s, p);
}

}  // namespace ccvpe
