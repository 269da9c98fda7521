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

}  // namespace yh

extern "C" int yh_letterbox_fwd(const yh_letterbox_desc* d, void* stream) {
    using namespace yh;
    if (!d || !d->src || !d->tmp || !d->dst || !d->hbounds || !d->hk || !d->vbounds || !d->vk) return YH_EINVAL;
    if (d->h0 <= 0 || d->w0 <= 0 || d->new_h <= 0 || d->new_w <= 0 || d->out_h <= 0 || d->out_w <= 0) return YH_EINVAL;
    if (d->c != 1 && d->c != 3 && d->c != 4) return YH_EINVAL;
    if (d->src_pitch < d->w0 * d->c || d->hksize <= 0 || d->vksize <= 0) return YH_EINVAL;
    if (d->top < 0 || d->left < 0 || d->top + d->new_h > d->out_h || d->left + d->new_w > d->out_w) return YH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
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
