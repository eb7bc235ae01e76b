// Device-side input pipeline in front of forward(): resize + ToTensor + Normalize + panorama roll + FoV crop
// (SURVEY 8f row 2).
//
// Reference call sites: transforms.Resize([320, 640]) / Resize([512, 512]) + ToTensor + Normalize (train_VIGOR.py:57-70,
// applied at datasets.py:106), torch.roll(grd, shift, dims=2) (datasets.py:118), grd[..., :int(W*FoV/360)]
// (train_VIGOR.py:272-273).  torchvision's Resize on a PIL image is PIL.Image.resize(BILINEAR): Pillow's 8-bit
// resampler (Resample.c) - a triangle filter whose support grows with the down-scaling factor, 22-bit fixed-point
// taps, horizontal pass first, uint8 between the passes.  The kernels restate exactly that arithmetic (taps are
// recomputed per thread in double precision, bit for bit what precompute_coeffs / normalize_coeffs_8bpc produce),
// so the result equals PIL's byte for byte and the float stage is the same divide / subtract / divide as torchvision.
// HBM-bound: a 1024 x 2048 panorama is read once (6.3 MB), the 1024 x 640 intermediate is 2 MB, the output 2.5 MB.
#include "kernels.h"

namespace ccvpe {

static constexpr int RS_MAXK = 2 * 8 + 1;        // taps for down-scaling factors up to 8
static constexpr int RS_BITS = 32 - 8 - 2;       // Pillow PRECISION_BITS

// taps of output index xx for an axis resampled in_size -> out_size (Resample.c precompute_coeffs + normalize_coeffs_8bpc)
__device__ __forceinline__ void resize_taps(int in_size, int out_size, int xx, int& xmin, int& n, int* k) {
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double center = ((double)xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    n = xmax - xmin;
    double w[RS_MAXK];
    double ww = 0.0;
#pragma unroll
    for (int x = 0; x < RS_MAXK; ++x) {
        double t = ((double)(x + xmin) - center + 0.5) * ss;
        if (t < 0.0) t = -t;
        const double v = (x < n && t < 1.0) ? 1.0 - t : 0.0;
        w[x] = v;
        if (x < n) ww += v;
    }
#pragma unroll
    for (int x = 0; x < RS_MAXK; ++x) {
        double v = w[x];
        if (x < n && ww != 0.0) v = v / ww;
        k[x] = v < 0.0 ? (int)(-0.5 + v * (double)(1 << RS_BITS)) : (int)(0.5 + v * (double)(1 << RS_BITS));
    }
}

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= RS_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// pass 1: [B, IH, IW, 3] -> [B, IH, OW, 3], one thread per output pixel
__global__ __launch_bounds__(256) void resize_h_kernel(const ResizeParams p) {
    const long long total = (long long)p.B * p.IH * p.OW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int xo = (int)(i % p.OW);
        const long long row = i / p.OW;                      // b * IH + y
        int xmin, n, k[RS_MAXK];
        resize_taps(p.IW, p.OW, xo, xmin, n, k);
        const unsigned char* src = p.in + ((size_t)row * p.IW + xmin) * 3;
        int a0 = 1 << (RS_BITS - 1), a1 = a0, a2 = a0;
#pragma unroll
        for (int x = 0; x < RS_MAXK; ++x)
            if (x < n) { a0 += src[x * 3] * k[x]; a1 += src[x * 3 + 1] * k[x]; a2 += src[x * 3 + 2] * k[x]; }
        unsigned char* dst = p.tmp + (size_t)i * 3;
        dst[0] = clip8(a0); dst[1] = clip8(a1); dst[2] = clip8(a2);
    }
}

// pass 2 + ToTensor + Normalize + roll + crop: [B, IH, OW, 3] -> fp32 NCHW [B, 3, OH, crop_w]
__global__ __launch_bounds__(256) void resize_v_kernel(const ResizeParams p) {
    const long long total = (long long)p.B * p.OH * p.crop_w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % p.crop_w);
        long long t = i / p.crop_w;
        const int yo = (int)(t % p.OH);
        const int b = (int)(t / p.OH);
        int sx = x - (p.shift ? p.shift[b] : 0);             // torch.roll(grd, shift, dims=2): out[x] = in[(x - shift) mod W]
        sx %= p.OW;
        if (sx < 0) sx += p.OW;
        int ymin, n, k[RS_MAXK];
        resize_taps(p.IH, p.OH, yo, ymin, n, k);
        int a0, a1, a2;
        if (p.IH == p.OH) {                                  // Pillow skips a pass whose size does not change
            const unsigned char* s = p.tmp + (((size_t)b * p.IH + yo) * p.OW + sx) * 3;
            a0 = s[0]; a1 = s[1]; a2 = s[2];
        } else {
            const unsigned char* src = p.tmp + (((size_t)b * p.IH + ymin) * p.OW + sx) * 3;
            const size_t rs = (size_t)p.OW * 3;
            a0 = a1 = a2 = 1 << (RS_BITS - 1);
#pragma unroll
            for (int y = 0; y < RS_MAXK; ++y)
                if (y < n) { a0 += src[y * rs] * k[y]; a1 += src[y * rs + 1] * k[y]; a2 += src[y * rs + 2] * k[y]; }
            a0 = clip8(a0); a1 = clip8(a1); a2 = clip8(a2);
        }
        const size_t plane = (size_t)p.OH * p.crop_w;
        float* o = p.out + (size_t)b * 3 * plane + (size_t)yo * p.crop_w + x;
        o[0] = ((float)a0 / 255.0f - p.mean[0]) / p.stdv[0];
        o[plane] = ((float)a1 / 255.0f - p.mean[1]) / p.stdv[1];
        o[2 * plane] = ((float)a2 / 255.0f - p.mean[2]) / p.stdv[2];
    }
}

// width pass skipped (IW == OW): pass 2 reads the input directly
int launch_resize(const ResizeParams& p_in, hipStream_t s) {
    ResizeParams p = p_in;
    if (p.IW > 8 * p.OW || p.IH > 8 * p.OH) return -1;      // more taps than RS_MAXK
    auto blocks = [](long long total) { long long b = (total + 255) / 256; return (int)(b > 256 * 32 ? 256 * 32 : b); };
    if (p.IW != p.OW) CCVPE_LAUNCH(resize_h_kernel, dim3(blocks((long long)p.B * p.IH * p.OW)), dim3(256), 0, s, p);
    else p.tmp = const_cast<unsigned char*>(p.in);
    CCVPE_LAUNCH(resize_v_kernel, dim3(blocks((long long)p.B * p.OH * p.crop_w)), dim3(256), 0, s, p);
    return 0;
}

}  // namespace ccvpe
