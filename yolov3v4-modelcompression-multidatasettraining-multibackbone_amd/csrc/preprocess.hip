// Device-side input pipeline, first slice: letterbox = aspect-preserving bilinear resize + constant border, the
// uint8 -> float scaling of detect.py:101 (x / 256, optionally * 2 - 1) and the HWC -> planar CHW layout change, straight into the
// fp32 NCHW batch the first-layer kernel reads.  Replaces, on the host, utils/datasets.py letterbox (reference datasets.py:611-646:
// cv2.resize INTER_LINEAR + cv2.copyMakeBorder) + `img[:, :, ::-1].transpose(2, 0, 1)` + `torch.from_numpy(img).to(device).float()
// / 256.0` - three host passes over the frame and a 3x larger upload than the original image needs.
//
// Resampling arithmetic is Pillow's 8-bit two-pass convolution (ImagingResample: horizontal pass into a uint8 image, then vertical),
// the one this package's host loader runs (OpenCV is not installed in this image; Pillow's BILINEAR widens its triangle filter when it
// shrinks an image).  Filter bounds and the 22-bit fixed-point coefficients are computed on the host exactly as Pillow's
// precompute_coeffs / normalize_coeffs_8bpc do (engine/preprocess.py) and passed in as tables, so the device result is bit-identical
// to the host loader's uint8 image and the float conversion is exact.
//
// arith != 0 selects OpenCV's uint8 arithmetic instead - the library the REFERENCE's loaders call (cv2.resize, reference
// datasets.py:519-526, :637) - as one fused gather kernel per frame, again driven by host tables (engine/imgtables.py):
//   1  INTER_LINEAR: r_j = S[j][i0] a0 + S[j][i1] a1 (11-bit weights), v = (((b0 (r_0 >> 4)) >> 16) + ((b1 (r_1 >> 4)) >> 16) + 2) >> 2
//      (HResizeLinear / VResizeLinear<uchar>; an exact 2x decimation, which resize() reroutes to the 2x2 area average, is the same
//      expression with all weights 1024);
//   2  INTER_AREA, non-integer shrink: float32 accumulation in table order, along x first, then sum (+)= beta * buf over the rows
//      (resizeArea_<uchar, float>), saturate_cast<uchar> = round half to even;
//   3  INTER_AREA, integer factors (resizeAreaFast_): integer cell sums, (s + 2) >> 2 for 2x2, s * (1.f / area) otherwise, ragged
//      edge cells divided by their sample count.
// OpenCV is not in this image: the formulas are restated from its sources and checked against oracle/cv2_restated.py only
// ("third-party restated").  The output is either the scaled fp32 CHW slot (detect) or a uint8 HWC image (the resized
// mosaic sources of csrc/augment.hip).  No fused multiply-adds: contraction is off for this file (mode 2 adds products in float).
#pragma clang fp contract(off)
#include "common.h"

namespace yh {

__device__ __forceinline__ int clip8(int v) {   // Pillow's clip8: (v >> PRECISION_BITS) clamped to 0..255
    v >>= 22;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// tmp[y][x][c] = clip8(2^21 + sum_i src[y][xmin + i][c] * k[x][i]),  y < h0, x < new_w
__global__ void resample_h_kernel(const uint8_t* __restrict__ src, int h0, int w0, int c, long pitch, const int32_t* __restrict__ bounds,
                                  const int32_t* __restrict__ kk, int ksize, uint8_t* __restrict__ tmp, int new_w) {
    const long total = (long)h0 * new_w;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % new_w);
        const long y = i / new_w;
        const int xmin = bounds[2 * x], n = bounds[2 * x + 1];
        const int32_t* k = kk + (long)x * ksize;
        const uint8_t* row = src + y * pitch + (long)xmin * c;
        for (int ch = 0; ch < c; ++ch) {
            int ss = 1 << 21;
            for (int t = 0; t < n; ++t) ss += (int)row[t * c + ch] * k[t];
            tmp[(y * new_w + x) * c + ch] = (uint8_t)clip8(ss);
        }
    }
}

// dst[ch'][Y][X] = scale * v + shift with v = the vertically resampled pixel inside the image area, the border value outside
__global__ void resample_v_pad_kernel(const uint8_t* __restrict__ tmp, int new_h, int new_w, int c, const int32_t* __restrict__ bounds,
                                      const int32_t* __restrict__ kk, int ksize, float* __restrict__ dst, int out_h, int out_w, int top,
                                      int left, int pad_value, int swap_rb, float scale, float shift) {
    const long total = (long)out_h * out_w;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int X = (int)(i % out_w), Y = (int)(i / out_w);
        const int x = X - left, y = Y - top;
        const bool inside = (unsigned)x < (unsigned)new_w && (unsigned)y < (unsigned)new_h;
        int ymin = 0, n = 0;
        const int32_t* k = kk;
        if (inside) {
            ymin = bounds[2 * y];
            n = bounds[2 * y + 1];
            k = kk + (long)y * ksize;
        }
        for (int ch = 0; ch < c; ++ch) {
            int v = pad_value;
            if (inside) {
                int ss = 1 << 21;
                for (int t = 0; t < n; ++t) ss += (int)tmp[((long)(ymin + t) * new_w + x) * c + ch] * k[t];
                v = clip8(ss);
            }
            const int oc = (swap_rb && c == 3) ? 2 - ch : ch;
            dst[((long)oc * out_h + Y) * out_w + X] = (float)v * scale + shift;
        }
    }
}

__device__ __forceinline__ int sat_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// one thread per output pixel of the letterboxed frame; the source is read straight from the decoded frame (L2 serves the overlap)
template <int ARITH, bool U8>
__global__ __launch_bounds__(256) void letterbox_cv2_kernel(const yh_letterbox_desc d) {
    const long total = (long)d.out_h * d.out_w;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int X = (int)(i % d.out_w), Y = (int)(i / d.out_w);
    const int x = X - d.left, y = Y - d.top;
    const bool inside = (unsigned)x < (unsigned)d.new_w && (unsigned)y < (unsigned)d.new_h;
    const int c = d.c;
    int px[4] = {d.pad_value, d.pad_value, d.pad_value, d.pad_value};
    if (inside) {
        const uint8_t* S = d.src;
        const long pitch = d.src_pitch;
        if constexpr (ARITH == 1) {
            const int i0 = d.hbounds[2 * x], i1 = d.hbounds[2 * x + 1], a0 = d.hk[2 * x], a1 = d.hk[2 * x + 1];
            const int j0 = d.vbounds[2 * y], j1 = d.vbounds[2 * y + 1], b0 = d.vk[2 * y], b1 = d.vk[2 * y + 1];
            const uint8_t *r0 = S + j0 * pitch, *r1 = S + j1 * pitch;
            for (int ch = 0; ch < c; ++ch) {
                const int s0 = (int)r0[i0 * c + ch] * a0 + (int)r0[i1 * c + ch] * a1;
                const int s1 = (int)r1[i0 * c + ch] * a0 + (int)r1[i1 * c + ch] * a1;
                px[ch] = sat_u8((((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2);
            }
        } else if constexpr (ARITH == 2) {
            const int x0 = d.hbounds[2 * x], nx = d.hbounds[2 * x + 1], y0 = d.vbounds[2 * y], ny = d.vbounds[2 * y + 1];
            const float* ka = reinterpret_cast<const float*>(d.hk) + (long)x * d.hksize;
            const float* kb = reinterpret_cast<const float*>(d.vk) + (long)y * d.vksize;
            for (int ch = 0; ch < c; ++ch) {
                float sum = 0.f;
                for (int t = 0; t < ny; ++t) {
                    const uint8_t* row = S + (long)(y0 + t) * pitch + (long)x0 * c + ch;
                    float buf = 0.f;
                    for (int u = 0; u < nx; ++u) buf = buf + (float)row[u * c] * ka[u];
                    sum = t == 0 ? kb[t] * buf : sum + kb[t] * buf;
                }
                px[ch] = sat_u8((int)rintf(sum));
            }
        } else {
            const int isx = d.hksize, isy = d.vksize;
            const int sx0 = x * isx, sy0 = y * isy;
            const bool full = sy0 + isy <= d.h0 && x < d.w0 / isx;
            const int ny = min(isy, d.h0 - sy0), nx = min(isx, d.w0 - sx0);
            for (int ch = 0; ch < c; ++ch) {
                int v = 0;
                if (ny > 0 && nx > 0) {
                    int sum = 0;
                    for (int t = 0; t < ny; ++t)
                        for (int u = 0; u < nx; ++u) sum += (int)S[(long)(sy0 + t) * pitch + (long)(sx0 + u) * c + ch];
                    if (full) v = (isx == 2 && isy == 2) ? (sum + 2) >> 2 : sat_u8((int)rintf((float)sum * (1.f / (float)(isx * isy))));
                    else v = sat_u8((int)rintf((float)sum / (float)(nx * ny)));
                }
                px[ch] = v;
            }
        }
    }
    if constexpr (U8) {
        uint8_t* dst = reinterpret_cast<uint8_t*>(d.dst) + ((long)Y * d.out_w + X) * c;
        for (int ch = 0; ch < c; ++ch) dst[ch] = (uint8_t)px[ch];
    } else {
        for (int ch = 0; ch < c; ++ch) {
            const int oc = (d.swap_rb && c == 3) ? 2 - ch : ch;
            d.dst[((long)oc * d.out_h + Y) * d.out_w + X] = (float)px[ch] * d.scale + d.shift;
        }
    }
}

template <int ARITH>
static void launch_cv2(const yh_letterbox_desc* d, hipStream_t s) {
    const long total = (long)d->out_h * d->out_w;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (d->out_u8) hipLaunchKernelGGL((letterbox_cv2_kernel<ARITH, true>), dim3(blocks), dim3(256), 0, s, *d);
    else hipLaunchKernelGGL((letterbox_cv2_kernel<ARITH, false>), dim3(blocks), dim3(256), 0, s, *d);
}

}  // namespace yh

extern "C" int yh_letterbox_fwd(const yh_letterbox_desc* d, void* stream) {
    using namespace yh;
    if (!d || !d->src || !d->dst) return YH_EINVAL;
    if (d->h0 <= 0 || d->w0 <= 0 || d->new_h <= 0 || d->new_w <= 0 || d->out_h <= 0 || d->out_w <= 0) return YH_EINVAL;
    if (d->c != 1 && d->c != 3 && d->c != 4) return YH_EINVAL;
    if (d->src_pitch < d->w0 * d->c || d->hksize <= 0 || d->vksize <= 0) return YH_EINVAL;
    if (d->top < 0 || d->left < 0 || d->top + d->new_h > d->out_h || d->left + d->new_w > d->out_w) return YH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (d->arith != YH_ARITH_PILLOW) {
        const bool tables = d->hbounds && d->hk && d->vbounds && d->vk;
        switch (d->arith) {
            case YH_ARITH_CV2_LINEAR:
                if (!tables || d->hksize != 2 || d->vksize != 2) return YH_EINVAL;
                launch_cv2<1>(d, s);
                break;
            case YH_ARITH_CV2_AREA:
                if (!tables) return YH_EINVAL;
                launch_cv2<2>(d, s);
                break;
            case YH_ARITH_CV2_AREA_FAST:
                if ((long)d->new_w * d->hksize > d->w0 + d->hksize - 1 || (long)d->new_h * d->vksize > d->h0 + d->vksize - 1) return YH_EINVAL;
                launch_cv2<3>(d, s);
                break;
            default: return YH_EINVAL;
        }
        return check_launch();
    }
    if (d->out_u8 || !d->tmp || !d->hbounds || !d->hk || !d->vbounds || !d->vk) return YH_EINVAL;
    const long n1 = (long)d->h0 * d->new_w, n2 = (long)d->out_h * d->out_w;
    long g1 = (n1 + 255) / 256, g2 = (n2 + 255) / 256;
    if (g1 > 8192) g1 = 8192;
    if (g2 > 8192) g2 = 8192;
    hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)g1), dim3(256), 0, s, d->src, d->h0, d->w0, d->c, (long)d->src_pitch, d->hbounds,
                       d->hk, d->hksize, d->tmp, d->new_w);
    hipLaunchKernelGGL(resample_v_pad_kernel, dim3((unsigned)g2), dim3(256), 0, s, d->tmp, d->new_h, d->new_w, d->c, d->vbounds, d->vk,
                       d->vksize, d->dst, d->out_h, d->out_w, d->top, d->left, d->pad_value, d->swap_rb, d->scale, d->shift);
    return check_launch();
}
