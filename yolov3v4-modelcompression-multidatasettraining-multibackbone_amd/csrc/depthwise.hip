// Depthwise convolution block and squeeze-excite for the MobilenetV3 / GhostNet backbones.
// Both are HBM-bound: NHWC, one thread = one pixel x one 16-byte channel vector, fp32 arithmetic.
#include "common.h"

namespace yh {

template <typename T> struct V16;
template <> struct V16<f16> { typedef f16x8 type; static constexpr int N = 8; };
template <> struct V16<float> { typedef f32x4 type; static constexpr int N = 4; };

template <typename T>
__global__ void pack_dw_weights_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                       const float* __restrict__ var, float eps, const int32_t* __restrict__ ch_map, int c,
                                       int taps, int c_phys, T* __restrict__ packed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c * taps) return;
    const int tap = i % taps, ch = i / taps;
    const float scale = gamma ? gamma[ch] / sqrtf(var[ch] + eps) : 1.f;
    packed[tap * c_phys + (ch_map ? ch_map[ch] : ch)] = (T)(w[i] * scale);
}

__global__ void pack_dw_bias_kernel(const float* __restrict__ conv_bias, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ mean,
                                    const float* __restrict__ var, float eps, const int32_t* __restrict__ ch_map, int c,
                                    float* __restrict__ out) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    float b = conv_bias ? conv_bias[ch] : 0.f;
    if (gamma) {
        const float sd = sqrtf(var[ch] + eps);
        b = (beta[ch] - gamma[ch] * mean[ch] / sd) + b * (gamma[ch] / sd);
    }
    out[ch_map ? ch_map[ch] : ch] = b;
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv_kernel(const yh_dw_desc d) {
    typedef typename V16<T>::type V;
    constexpr int VN = V16<T>::N;
    const int cg = d.c / VN;
    const long total = (long)d.n * d.ho * d.wo * cg;
    const T* x = reinterpret_cast<const T*>(d.x);
    const T* w = reinterpret_cast<const T*>(d.w);
    T* y = reinterpret_cast<T*>(d.y);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        long r = i / cg;
        const int wo = (int)(r % d.wo);
        r /= d.wo;
        const int ho = (int)(r % d.ho);
        const int n = (int)(r / d.ho);
        float acc[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[e] = d.bias[g * VN + e];
        for (int ky = 0; ky < d.k; ++ky) {
            const int hi = ho * d.stride - d.pad + ky;
            if ((unsigned)hi >= (unsigned)d.h) continue;
            for (int kx = 0; kx < d.k; ++kx) {
                const int wi = wo * d.stride - d.pad + kx;
                if ((unsigned)wi >= (unsigned)d.w_in) continue;
                const V xv = *reinterpret_cast<const V*>(x + (((long)n * d.h + hi) * d.w_in + wi) * d.ldx + g * VN);
                const V wv = *reinterpret_cast<const V*>(w + (long)(ky * d.k + kx) * d.c + g * VN);
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[e] = fmaf((float)xv[e], (float)wv[e], acc[e]);
            }
        }
        V o;
#pragma unroll
        for (int e = 0; e < VN; ++e) o[e] = (T)activate(acc[e], d.act, d.slope);
        *reinterpret_cast<V*>(y + (((long)n * d.ho + ho) * d.wo + wo) * d.ldy + g * VN) = o;
    }
}

// ---- squeeze-excite ---------------------------------------------------------------------------------
// grid (c_phys / VN, n): per-channel mean over the H*W pixels of one image
template <typename T>
__global__ __launch_bounds__(256) void se_pool_kernel(const yh_se_desc d) {
    typedef typename V16<T>::type V;
    constexpr int VN = V16<T>::N;
    __shared__ float red[256 * VN];
    const int g = blockIdx.x, n = blockIdx.y;
    const long hw = (long)d.h * d.w_in;
    const T* x = reinterpret_cast<const T*>(d.x) + (long)n * hw * d.ldx + g * VN;
    float acc[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) acc[e] = 0.f;
    for (long p = threadIdx.x; p < hw; p += blockDim.x) {
        const V v = *reinterpret_cast<const V*>(x + p * d.ldx);
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[e] += (float)v[e];
    }
#pragma unroll
    for (int e = 0; e < VN; ++e) red[e * 256 + threadIdx.x] = acc[e];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int e = 0; e < VN; ++e) red[e * 256 + threadIdx.x] += red[e * 256 + threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x < VN) d.pooled[(long)n * d.c_phys + g * VN + threadIdx.x] = red[threadIdx.x * 256] / (float)hw;
}

// grid n: hidden = relu(W1 pooled), gate = hsigmoid(W2 hidden); pad channels get gate 0
__global__ __launch_bounds__(256) void se_fc_kernel(const yh_se_desc d) {
    extern __shared__ float hidden[];
    const int n = blockIdx.x;
    const float* pooled = d.pooled + (long)n * d.c_phys;
    float* gate = d.gate + (long)n * d.c_phys;
    for (int c = threadIdx.x; c < d.c_phys; c += blockDim.x) gate[c] = 0.f;
    // (round 6: one output per WAVE with lanes striding over the row - coalesced loads, a wave reduction - measured 2.7 x SLOWER, se 1.12 ->
    // 3.02 ms on YOLOv3-Mobilenetv3: a wave then walks its outputs one after the other, a memory latency each, where 256 threads keep
    // 256 independent chains in flight on weights that sit in L2 anyway)
    for (int j = threadIdx.x; j < d.cr; j += blockDim.x) {
        const float* wr = d.w1 + (long)j * d.c;
        float s = 0.f;
        for (int i = 0; i < d.c; ++i) s = fmaf(wr[i], pooled[d.ch_map ? d.ch_map[i] : i], s);
        hidden[j] = fmaxf(s, 0.f);
    }
    __syncthreads();
    for (int co = threadIdx.x; co < d.c; co += blockDim.x) {
        const float* wr = d.w2 + (long)co * d.cr;
        float s = 0.f;
        for (int j = 0; j < d.cr; ++j) s = fmaf(wr[j], hidden[j], s);
        gate[d.ch_map ? d.ch_map[co] : co] = fminf(fmaxf(s + 3.f, 0.f), 6.f) / 6.f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void se_scale_kernel(const yh_se_desc d) {
    typedef typename V16<T>::type V;
    constexpr int VN = V16<T>::N;
    const int cg = d.c_phys / VN;
    const long hw = (long)d.h * d.w_in;
    const long total = (long)d.n * hw * cg;
    const T* x = reinterpret_cast<const T*>(d.x);
    T* y = reinterpret_cast<T*>(d.y);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const long pix = i / cg;
        const int n = (int)(pix / hw);
        const V v = *reinterpret_cast<const V*>(x + pix * d.ldx + g * VN);
        const float* gt = d.gate + (long)n * d.c_phys + g * VN;
        V o;
#pragma unroll
        for (int e = 0; e < VN; ++e) o[e] = (T)((float)v[e] * gt[e]);
        *reinterpret_cast<V*>(y + pix * d.ldy + g * VN) = o;
    }
}

static inline unsigned grid_cap(long total, long cap = 256L * 16) {
    long g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace yh

using namespace yh;

extern "C" int yh_dw_pack_weights(int dtype, const float* w, const float* conv_bias, const float* bn_gamma,
                                  const float* bn_beta, const float* bn_mean, const float* bn_var, float bn_eps,
                                  const int32_t* ch_map, int c, int k, int c_phys, void* packed, float* bias_out,
                                  void* stream) {
    if (!w || !packed || !bias_out || c <= 0 || k <= 0 || c_phys < c) return YH_EINVAL;
    if (dtype != YH_F16 && dtype != YH_F32) return YH_EINVAL;
    if (bn_gamma && (!bn_beta || !bn_mean || !bn_var)) return YH_EINVAL;
    if (c_phys % 8) return YH_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int taps = k * k;
    hipError_t e = hipMemsetAsync(packed, 0, (size_t)taps * c_phys * (dtype == YH_F16 ? 2 : 4), s);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(bias_out, 0, (size_t)c_phys * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    const int total = c * taps;
    if (dtype == YH_F16)
        hipLaunchKernelGGL(pack_dw_weights_kernel<f16>, dim3((total + 255) / 256), dim3(256), 0, s, w, bn_gamma, bn_var, bn_eps,
                           ch_map, c, taps, c_phys, (f16*)packed);
    else
        hipLaunchKernelGGL(pack_dw_weights_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, s, w, bn_gamma, bn_var, bn_eps,
                           ch_map, c, taps, c_phys, (float*)packed);
    hipLaunchKernelGGL(pack_dw_bias_kernel, dim3((c + 255) / 256), dim3(256), 0, s, conv_bias, bn_gamma, bn_beta, bn_mean, bn_var,
                       bn_eps, ch_map, c, bias_out);
    return check_launch();
}

extern "C" int yh_dwconv2d_fwd(const yh_dw_desc* d, void* stream) {
    if (!d || !d->x || !d->w || !d->bias || !d->y || d->n <= 0 || d->c <= 0 || d->k <= 0 || d->stride <= 0) return YH_EINVAL;
    if (d->dtype != YH_F16 && d->dtype != YH_F32) return YH_EINVAL;
    const int v = d->dtype == YH_F16 ? 8 : 4;
    if (d->c % v || d->ldx % v || d->ldy % v || !aligned16(d->x) || !aligned16(d->y) || !aligned16(d->w)) return YH_EALIGN;
    if (d->ho != (d->h + 2 * d->pad - d->k) / d->stride + 1 || d->wo != (d->w_in + 2 * d->pad - d->k) / d->stride + 1) return YH_EINVAL;
    const long total = (long)d->n * d->ho * d->wo * (d->c / v);
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == YH_F16) hipLaunchKernelGGL(dwconv_kernel<f16>, dim3(grid_cap(total)), dim3(256), 0, s, *d);
    else hipLaunchKernelGGL(dwconv_kernel<float>, dim3(grid_cap(total)), dim3(256), 0, s, *d);
    return check_launch();
}

extern "C" int yh_se_fwd(const yh_se_desc* d, void* stream) {
    if (!d || !d->x || !d->y || !d->w1 || !d->w2 || !d->pooled || !d->gate) return YH_EINVAL;
    if (d->n <= 0 || d->c <= 0 || d->cr <= 0 || d->c_phys < d->c || d->n > 65535) return YH_EINVAL;
    if (d->dtype != YH_F16 && d->dtype != YH_F32) return YH_EINVAL;
    const int v = d->dtype == YH_F16 ? 8 : 4;
    if (d->c_phys % 8 || d->ldx % v || d->ldy % v || !aligned16(d->x) || !aligned16(d->y)) return YH_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const dim3 pg(d->c_phys / v, d->n);
    if (d->dtype == YH_F16) hipLaunchKernelGGL(se_pool_kernel<f16>, pg, dim3(256), 0, s, *d);
    else hipLaunchKernelGGL(se_pool_kernel<float>, pg, dim3(256), 0, s, *d);
    hipLaunchKernelGGL(se_fc_kernel, dim3(d->n), dim3(256), (size_t)d->cr * sizeof(float), s, *d);
    const long total = (long)d->n * d->h * d->w_in * (d->c_phys / v);
    if (d->dtype == YH_F16) hipLaunchKernelGGL(se_scale_kernel<f16>, dim3(grid_cap(total)), dim3(256), 0, s, *d);
    else hipLaunchKernelGGL(se_scale_kernel<float>, dim3(grid_cap(total)), dim3(256), 0, s, *d);
    return check_launch();
}
