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

// Round 6: the same convolution with a FIXED channel group per thread and an incremental walk over the pixels (the form of the BatchNorm
// passes): no division per output (the kernel above decodes every output vector with three 64-bit divisions), the kernel size and the
// activation as template parameters (the run-time switch of activate() is expanded per element), the 3x3 weights of the thread's channel
// group in registers, padding taps through a zero page instead of a branch around the load.  Same multiply-add order per output (taps in
// row-major order onto the bias; a padding tap adds x = 0): bit-identical results.  K: 3 / 5, or 0 = run-time size (weights re-read per tap).
__device__ __attribute__((aligned(16))) unsigned int dwf_zero_page[8] = {0, 0, 0, 0, 0, 0, 0, 0};

template <typename T, int K, int ACT>
__global__ __launch_bounds__(256) void dwconv_walk_kernel(const yh_dw_desc d, const int cgb, const int rows, const int ppb) {
    typedef typename V16<T>::type V;
    constexpr bool WREG = K == 3;                       // 3x3: the nine weight vectors of the channel group stay in registers (5x5 would take 100)
    constexpr int VN = V16<T>::N, KK = WREG ? K * K : 1;
    const int cgs = d.c / VN;
    const int cgl = threadIdx.x % cgb, prow = threadIdx.x / cgb;
    const int g = blockIdx.x * cgb + cgl;
    const int pixels = d.n * d.ho * d.wo;
    const int p0 = blockIdx.y * ppb, p1 = min(p0 + ppb, pixels);
    int p = p0 + prow;
    if (prow >= rows || g >= cgs || p >= p1) return;
    const int k = K ? K : d.k;
    const T* const x = reinterpret_cast<const T*>(d.x) + g * VN;
    const T* const w = reinterpret_cast<const T*>(d.w) + g * VN;
    T* const y = reinterpret_cast<T*>(d.y) + g * VN;
    const T* const zero = reinterpret_cast<const T*>(dwf_zero_page);
    V wv[KK];
    if constexpr (WREG) {
#pragma unroll
        for (int t = 0; t < KK; ++t) wv[t] = *reinterpret_cast<const V*>(w + (long)t * d.c);
    }
    float bias[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) bias[e] = d.bias[g * VN + e];
    const int howo = d.ho * d.wo;
    int n = p / howo;
    const int rem = p - n * howo;
    int ho = rem / d.wo, wo = rem - ho * d.wo;
    const int rowpitch = d.w_in * d.ldx;
    for (; p < p1; p += rows) {
        float acc[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[e] = bias[e];
        const int hi0 = ho * d.stride - d.pad, wi0 = wo * d.stride - d.pad;
        const T* const xb = x + ((long)(n * d.h + hi0) * d.w_in + wi0) * d.ldx;      // only dereferenced at in-bounds taps
        if constexpr (K != 0) {
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const bool in = (unsigned)(hi0 + ky) < (unsigned)d.h && (unsigned)(wi0 + kx) < (unsigned)d.w_in;
                    const T* ad = in ? xb + ky * rowpitch + kx * d.ldx : zero;
                    asm volatile("" : "+v"(ad));              // keep the select: no branch around the load
                    const V xv = *reinterpret_cast<const V*>(ad);
                    const V w1 = WREG ? wv[WREG ? ky * K + kx : 0] : *reinterpret_cast<const V*>(w + (long)(ky * K + kx) * d.c);
#pragma unroll
                    for (int e = 0; e < VN; ++e) acc[e] = fmaf((float)xv[e], (float)w1[e], acc[e]);
                }
        } else {
            for (int ky = 0; ky < k; ++ky) {
                if ((unsigned)(hi0 + ky) >= (unsigned)d.h) continue;
                for (int kx = 0; kx < k; ++kx) {
                    if ((unsigned)(wi0 + kx) >= (unsigned)d.w_in) continue;
                    const V xv = *reinterpret_cast<const V*>(xb + ky * rowpitch + kx * d.ldx);
                    const V w1 = *reinterpret_cast<const V*>(w + (long)(ky * k + kx) * d.c);
#pragma unroll
                    for (int e = 0; e < VN; ++e) acc[e] = fmaf((float)xv[e], (float)w1[e], acc[e]);
                }
            }
        }
        V o;
#pragma unroll
        for (int e = 0; e < VN; ++e) o[e] = (T)activate(acc[e], ACT, d.slope);
        *reinterpret_cast<V*>(y + (long)p * d.ldy) = o;
        wo += rows;
        while (wo >= d.wo) {
            wo -= d.wo;
            if (++ho == d.ho) {
                ho = 0;
                ++n;
            }
        }
    }
}

// ---- squeeze-excite ---------------------------------------------------------------------------------
// per-channel mean over the H*W pixels of one image: common.h image_channel_sums_kernel

// grid n: hidden = relu(W1 pooled), gate = hsigmoid(W2 hidden); pad channels get gate 0.  LDS: hidden[cr] | pooled in logical order [c].
// Round 6: both products through common.h rows_dot16 with sixteen waves (thread-per-output read its row at one cache line per lane and
// load: ~0.2 ms per 960-channel layer, texture-addresser bound; one output per wave was slower still, see rows_dot16).
__global__ __launch_bounds__(1024) void se_fc_kernel(const yh_se_desc d) {
    extern __shared__ float hidden[];
    float* const pl = hidden + d.cr;
    const int n = blockIdx.x;
    const float* pooled = d.pooled + (long)n * d.c_phys;
    float* gate = d.gate + (long)n * d.c_phys;
    for (int c = threadIdx.x; c < d.c_phys; c += blockDim.x) gate[c] = 0.f;
    for (int c = threadIdx.x; c < d.c; c += blockDim.x) pl[c] = pooled[d.ch_map ? d.ch_map[c] : c];
    __syncthreads();
    rows_dot16(d.w1, d.cr, d.c, pl, [&](int j, float s) { hidden[j] = fmaxf(s, 0.f); });
    __syncthreads();
    rows_dot16(d.w2, d.c, d.cr, hidden,
               [&](int co, float s) { gate[d.ch_map ? d.ch_map[co] : co] = fminf(fmaxf(s + 3.f, 0.f), 6.f) / 6.f; });
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
    const long pixels = (long)d->n * d->ho * d->wo;
    static const bool old_form = getenv("YH_DW_WALK") && atoi(getenv("YH_DW_WALK")) == 0;      // A/B knob: 0 = the round-1 kernel
    if (!old_form && pixels < (1L << 31) - 65536 && (long)d->n * d->h * d->w_in * d->ldx < (1L << 40)) {
        // channel groups x pixel lanes per workgroup as the BatchNorm passes; >= 4 pixels per thread (the weights are loaded once), ~8192 workgroups
        const int cgs = d->c / v, cgb = cgs < 256 ? cgs : 256, rows = 256 / cgb, gx = (cgs + cgb - 1) / cgb;
        long ppb = (pixels * gx + 8191) / 8192;
        if (ppb < 4L * rows) ppb = 4L * rows;
        ppb = (ppb + rows - 1) / rows * rows;
        const dim3 grid(gx, (unsigned)((pixels + ppb - 1) / ppb));
#define YH_DWF_A(T, K, A) hipLaunchKernelGGL((dwconv_walk_kernel<T, K, A>), grid, dim3(256), 0, s, *d, cgb, rows, (int)ppb)
#define YH_DWF_K(T, K)                                                                   \
        switch (d->act) {                                                                    \
            case YH_ACT_LEAKY: YH_DWF_A(T, K, YH_ACT_LEAKY); break;                          \
            case YH_ACT_RELU: YH_DWF_A(T, K, YH_ACT_RELU); break;                            \
            case YH_ACT_RELU6: YH_DWF_A(T, K, YH_ACT_RELU6); break;                          \
            case YH_ACT_HSWISH: YH_DWF_A(T, K, YH_ACT_HSWISH); break;                        \
            case YH_ACT_MISH: YH_DWF_A(T, K, YH_ACT_MISH); break;                            \
            default: YH_DWF_A(T, K, YH_ACT_LINEAR); break;                                   \
        }
#define YH_DWF_T(T)                                                                      \
        if (d->k == 3) { YH_DWF_K(T, 3) } else if (d->k == 5) { YH_DWF_K(T, 5) } else { YH_DWF_K(T, 0) }
        if (d->dtype == YH_F16) { YH_DWF_T(f16) } else { YH_DWF_T(float) }
#undef YH_DWF_T
#undef YH_DWF_K
#undef YH_DWF_A
        return check_launch();
    }
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
    const dim3 pg((d->c_phys / v + 7) / 8, d->n);
    const int hw = d->h * d->w_in;
    if (d->dtype == YH_F16)
        hipLaunchKernelGGL((image_channel_sums_kernel<f16, f16x8, 8, false>), pg, dim3(1024), 0, s, (const f16*)d->x, (long)d->ldx, nullptr, 0L,
                           d->c_phys, hw, (float)hw, d->pooled, (long)d->c_phys);
    else
        hipLaunchKernelGGL((image_channel_sums_kernel<float, f32x4, 4, false>), pg, dim3(1024), 0, s, (const float*)d->x, (long)d->ldx, nullptr,
                           0L, d->c_phys, hw, (float)hw, d->pooled, (long)d->c_phys);
    hipLaunchKernelGGL(se_fc_kernel, dim3(d->n), dim3(1024), (size_t)(d->cr + d->c) * sizeof(float), s, *d);
    const long total = (long)d->n * d->h * d->w_in * (d->c_phys / v);
    if (d->dtype == YH_F16) hipLaunchKernelGGL(se_scale_kernel<f16>, dim3(grid_cap(total)), dim3(256), 0, s, *d);
    else hipLaunchKernelGGL(se_scale_kernel<float>, dim3(grid_cap(total)), dim3(256), 0, s, *d);
    return check_launch();
}
