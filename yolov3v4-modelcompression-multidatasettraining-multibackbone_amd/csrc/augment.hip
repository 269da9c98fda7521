// Device-side input pipeline, second slice: the training loader's 4-image mosaic, random affine warp, HSV
// augmentation, left-right flip, HWC -> CHW and the uint8 -> float scaling, as ONE gather kernel per output image.
// Replaces, on the host, utils/datasets.py load_mosaic (reference datasets.py:553-608: paste four images into a 2s x 2s canvas),
// random_affine (datasets.py:649-715: warpAffine, bilinear, border 114), augment_hsv (datasets.py:534-550), the flip and
// transpose of __getitem__ (datasets.py:470-505) and `imgs.to(device).float() / 256.0` (train.py:345) - five passes over a 4 s^2
// canvas and an s^2 image on one host core per item.  Random numbers, label geometry and file decoding stay on the host
// (engine/preprocess.py draws them in the host loader's order, so both paths see the same stream).
//
// Arithmetic is the host loader's, operation for operation, so the result is bit-identical to it:
//  * warp: Pillow's ImagingGenericTransform with affine_transform + bilinear_filter8 in double precision:
//      xin = a0 (x + .5) + a1 (y + .5) + a2 (same for yin); outside [0, W) x [0, H) -> fill; else x' = xin - .5, x0 = floor(x'),
//      v = row(y0)[x0] + (row(y0)[x0+1] - row(y0)[x0]) dx  (columns clamped), second row only if y0 + 1 is inside, (UINT8) truncation;
//    the canvas is virtual: a canvas pixel is looked up in the placement rectangle that covers it (they do not overlap) or is the
//    pad value.
//  * HSV: rgb -> hsv in float32 (numpy float32 arrays), gains and hsv -> rgb in float64 (numpy promotes float32 array * float64
//    scalar), `(rgb * 255 + 0.5)` clipped and truncated; numpy's remainder semantics for the two `%`.
//
// arith = YH_ARITH_CV2_LINEAR evaluates the REFERENCE's library calls instead (OpenCV, restated from its sources and checked
// against oracle/cv2_restated.py only - "third-party restated"):
//  * warp: cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT 114) of OpenCV 3.x - 4.10 (reference datasets.py:677): `inv` is OpenCV's own
//    inversion of M (engine/imgtables.py); X0 = round((inv1 y + inv2) 1024) + 16, adelta = round(inv0 x 1024) (round half even),
//    X = (X0 + adelta) >> 5: integer part X >> 5, 5-bit fraction X & 31 (same for Y); weights (32-fy)(32-fx) 32 ... (BilinearTab_i,
//    they sum to 2^15); v = (sum of the four neighbours, pad value where outside, times weights + 2^14) >> 15;
//  * the four source images were resized on the device by yh_letterbox_fwd (arith cv2, uint8 output) before this kernel runs;
//  * HSV: cv2.cvtColor(BGR2HSV) in 12-bit fixed point (RGB2HSV_b, hue range 180), the three 256-entry tables of augment_hsv
//    (datasets.py:539-542, built on the host from the random gains), cv2.cvtColor(HSV2BGR) in float32 (HSV2RGB_native).
// No fused multiply-adds anywhere: contraction is switched off for this file.
#pragma clang fp contract(off)
#include "common.h"

namespace yh {

__device__ __forceinline__ float np_remainder_f(float a, float b) {   // numpy remainder for b > 0
    float r = fmodf(a, b);
    if (r != 0.f) { if (r < 0.f) r += b; } else r = copysignf(0.f, b);
    return r;
}
__device__ __forceinline__ double np_remainder_d(double a, double b) {
    double r = fmod(a, b);
    if (r != 0.0) { if (r < 0.0) r += b; } else r = copysign(0.0, b);
    return r;
}

__device__ __forceinline__ int canvas_px(const yh_mosaic_desc& d, int cy, int cx, int ch) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (cx >= d.x1a[k] && cx < d.x2a[k] && cy >= d.y1a[k] && cy < d.y2a[k]) {
            const int sy = d.y1b[k] + (cy - d.y1a[k]), sx = d.x1b[k] + (cx - d.x1a[k]);
            return d.src[k][(long)sy * d.src_pitch[k] + (long)sx * d.c + ch];
        }
    }
    return d.pad_value;
}

__device__ __forceinline__ void hsv_augment(const yh_mosaic_desc& d, int (&px)[3]) {
    const float r = (float)px[0] / 255.0f, g = (float)px[1] / 255.0f, b = (float)px[2] / 255.0f;
    const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
    const float dl = mx - mn;
    float h = 0.f;
    if (dl > 0.f) {
        if (mx == r) h = np_remainder_f((g - b) / dl, 6.f);
        else if (mx == g) h = (b - r) / dl + 2.f;
        else h = (r - g) / dl + 4.f;
    }
    const float s = mx > 0.f ? dl / fmaxf(mx, 1e-12f) : 0.f;
    h = h / 6.0f;
    const double H = np_remainder_d((double)h * d.hsv_gain[0], 1.0);
    const double S = fmin(fmax((double)s * d.hsv_gain[1], 0.0), 1.0);
    const double V = fmin(fmax((double)mx * d.hsv_gain[2], 0.0), 1.0);
    const double h6 = H * 6.0;
    const double fi = floor(h6);
    const double f = h6 - fi;
    const double p = V * (1.0 - S), q = V * (1.0 - S * f), t = V * (1.0 - S * (1.0 - f));
    int i = (int)fi % 6;
    if (i < 0) i += 6;
    double R, G, B;
    switch (i) {
        case 0: R = V; G = t; B = p; break;
        case 1: R = q; G = V; B = p; break;
        case 2: R = p; G = V; B = t; break;
        case 3: R = p; G = q; B = V; break;
        case 4: R = t; G = p; B = V; break;
        default: R = V; G = p; B = q; break;
    }
    const double o[3] = {R, G, B};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double v = o[c] * 255.0 + 0.5;
        v = fmin(fmax(v, 0.0), 255.0);
        px[c] = (int)v;
    }
}

__device__ __forceinline__ int u8sat(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// cv2.cvtColor(BGR2HSV) -> cv2.LUT x 3 -> cv2.cvtColor(HSV2BGR) on one pixel held as (R, G, B)
__device__ __forceinline__ void hsv_augment_cv2(const uint8_t* __restrict__ lut, int (&px)[3]) {
    const int r = px[0], g = px[1], b = px[2];
    const int v = max(max(b, g), r), vmin = min(min(b, g), r);
    const int diff = v - vmin;
    const int sdiv = v ? (int)rint((double)(255 << 12) / (1.0 * (double)v)) : 0;
    const int hdiv = diff ? (int)rint((double)(180 << 12) / (6.0 * (double)diff)) : 0;
    const int s = (diff * sdiv + (1 << 11)) >> 12;
    int h = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
    h = (h * hdiv + (1 << 11)) >> 12;
    if (h < 0) h += 180;
    const int H = lut[u8sat(h)], S = lut[256 + s], V = lut[512 + v];
    const float fs = (float)S * (1.0f / 255.0f), fv = (float)V * (1.0f / 255.0f);
    float ob = fv, og = fv, orr = fv;
    if (S != 0) {
        float fh = (float)H * (6.f / 180.f);
        fh = fmodf(fh, 6.f);
        int sector = (int)floorf(fh);
        fh -= (float)sector;
        if ((unsigned)sector >= 6u) { sector = 0; fh = 0.f; }
        const float tab[4] = {fv, fv * (1.f - fs), fv * (1.f - fs * fh), fv * (1.f - fs * (1.f - fh))};
        // tab index of (b, g, r) per sector: {1,3,0} {1,0,2} {3,0,1} {0,2,1} {0,1,3} {2,1,0}
        const int ib = (0x200311 >> (4 * sector)) & 15, ig = (0x112003 >> (4 * sector)) & 15, ir = (0x031120 >> (4 * sector)) & 15;
        ob = tab[ib]; og = tab[ig]; orr = tab[ir];
    }
    px[0] = u8sat((int)rintf(orr * 255.f));
    px[1] = u8sat((int)rintf(og * 255.f));
    px[2] = u8sat((int)rintf(ob * 255.f));
}

template <typename OutT>
__device__ __forceinline__ void store_px(const yh_mosaic_desc& d, int Y, int X, const int (&px)[3]) {
    OutT* const dst = reinterpret_cast<OutT*>(d.dst);
    for (int ch = 0; ch < d.c; ++ch) {
        const long o = ((long)ch * d.out_h + Y) * d.out_w + X;
        if constexpr (sizeof(OutT) == 1) dst[o] = (OutT)px[ch];
        else dst[o] = (OutT)((float)px[ch] / d.divisor);
    }
}

template <typename OutT>
__global__ __launch_bounds__(256) void mosaic_affine_hsv_cv2_kernel(const yh_mosaic_desc d) {
    const long total = (long)d.out_h * d.out_w;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int X = (int)(i % d.out_w), Y = (int)(i / d.out_w);
    const int xo = d.flip_lr ? d.out_w - 1 - X : X;
    const int adelta = (int)rint(d.inv[0] * (double)xo * 1024.0), bdelta = (int)rint(d.inv[3] * (double)xo * 1024.0);
    const int X0 = (int)rint((d.inv[1] * (double)Y + d.inv[2]) * 1024.0) + 16;
    const int Y0 = (int)rint((d.inv[4] * (double)Y + d.inv[5]) * 1024.0) + 16;
    const int XX = (X0 + adelta) >> 5, YY = (Y0 + bdelta) >> 5;
    int sx = XX >> 5, sy = YY >> 5;
    sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
    sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
    const int fx = XX & 31, fy = YY & 31;
    const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    int px[3] = {d.pad_value, d.pad_value, d.pad_value};
    for (int ch = 0; ch < d.c; ++ch) {
        const int acc = canvas_px(d, sy, sx, ch) * w00 + canvas_px(d, sy, sx + 1, ch) * w01 + canvas_px(d, sy + 1, sx, ch) * w10 +
                        canvas_px(d, sy + 1, sx + 1, ch) * w11;
        px[ch] = u8sat((acc + (1 << 14)) >> 15);
    }
    if (d.hsv && d.c == 3) hsv_augment_cv2(d.lut, px);
    store_px<OutT>(d, Y, X, px);
}

template <typename OutT>
__global__ __launch_bounds__(256) void mosaic_affine_hsv_kernel(const yh_mosaic_desc d) {
    const long total = (long)d.out_h * d.out_w;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int X = (int)(i % d.out_w), Y = (int)(i / d.out_w);
    const int xo = d.flip_lr ? d.out_w - 1 - X : X;     // the flip mirrors the finished image: output X shows source column W-1-X
    int px[3] = {d.pad_value, d.pad_value, d.pad_value};
    const double xs = (double)xo + 0.5, ys = (double)Y + 0.5;
    double xin = d.inv[0] * xs + d.inv[1] * ys + d.inv[2];
    double yin = d.inv[3] * xs + d.inv[4] * ys + d.inv[5];
    const int W = d.canvas_w, H = d.canvas_h;
    if (!(xin < 0.0 || yin < 0.0 || xin >= (double)W || yin >= (double)H)) {
        xin -= 0.5;
        yin -= 0.5;
        const int x = (int)floor(xin), y = (int)floor(yin);
        const double dx = xin - (double)x, dy = yin - (double)y;
        const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
        const int x0 = x < 0 ? 0 : (x >= W ? W - 1 : x);
        const int x1 = x + 1 < 0 ? 0 : (x + 1 >= W ? W - 1 : x + 1);
        const bool second = y + 1 >= 0 && y + 1 < H;
        for (int ch = 0; ch < d.c; ++ch) {
            const double a = (double)canvas_px(d, yc, x0, ch), b = (double)canvas_px(d, yc, x1, ch);
            double v1 = a + (b - a) * dx;
            double v2 = v1;
            if (second) {
                const double a2 = (double)canvas_px(d, y + 1, x0, ch), b2 = (double)canvas_px(d, y + 1, x1, ch);
                v2 = a2 + (b2 - a2) * dx;
            }
            v1 = v1 + (v2 - v1) * dy;
            px[ch] = (int)(uint8_t)v1;
        }
    }
    if (d.hsv && d.c == 3) hsv_augment(d, px);
    store_px<OutT>(d, Y, X, px);
}

}  // namespace yh

extern "C" int yh_mosaic_affine_hsv(const yh_mosaic_desc* d, void* stream) {
    using namespace yh;
    if (!d || !d->dst) return YH_EINVAL;
    if (d->c != 1 && d->c != 3) return YH_EINVAL;
    if (d->out_h <= 0 || d->out_w <= 0 || d->canvas_h <= 0 || d->canvas_w <= 0 || !(d->divisor > 0.f)) return YH_EINVAL;
    for (int k = 0; k < 4; ++k) {
        const bool empty = d->x2a[k] <= d->x1a[k] || d->y2a[k] <= d->y1a[k];
        if (empty) continue;
        if (!d->src[k] || d->x1a[k] < 0 || d->y1a[k] < 0 || d->x2a[k] > d->canvas_w || d->y2a[k] > d->canvas_h) return YH_EINVAL;
        if (d->x1b[k] < 0 || d->y1b[k] < 0 || d->x1b[k] + (d->x2a[k] - d->x1a[k]) > d->src_w[k] ||
            d->y1b[k] + (d->y2a[k] - d->y1a[k]) > d->src_h[k] || d->src_pitch[k] < d->src_w[k] * d->c) return YH_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)d->out_h * d->out_w;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (d->arith != YH_ARITH_PILLOW && d->arith != YH_ARITH_CV2_LINEAR) return YH_EINVAL;
    if (d->arith == YH_ARITH_CV2_LINEAR) {
        if (d->hsv && d->c == 3 && !d->lut) return YH_EINVAL;
        switch (d->out_dtype) {
            case YH_MOSAIC_U8: hipLaunchKernelGGL(mosaic_affine_hsv_cv2_kernel<uint8_t>, dim3(blocks), dim3(256), 0, s, *d); break;
            case YH_MOSAIC_F32: hipLaunchKernelGGL(mosaic_affine_hsv_cv2_kernel<float>, dim3(blocks), dim3(256), 0, s, *d); break;
            case YH_MOSAIC_F16: hipLaunchKernelGGL(mosaic_affine_hsv_cv2_kernel<f16>, dim3(blocks), dim3(256), 0, s, *d); break;
            default: return YH_EINVAL;
        }
        return check_launch();
    }
    switch (d->out_dtype) {
        case YH_MOSAIC_U8: hipLaunchKernelGGL(mosaic_affine_hsv_kernel<uint8_t>, dim3(blocks), dim3(256), 0, s, *d); break;
        case YH_MOSAIC_F32: hipLaunchKernelGGL(mosaic_affine_hsv_kernel<float>, dim3(blocks), dim3(256), 0, s, *d); break;
        case YH_MOSAIC_F16: hipLaunchKernelGGL(mosaic_affine_hsv_kernel<f16>, dim3(blocks), dim3(256), 0, s, *d); break;
        default: return YH_EINVAL;
    }
    return check_launch();
}
