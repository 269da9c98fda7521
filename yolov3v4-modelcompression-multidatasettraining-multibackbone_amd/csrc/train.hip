// Training-path kernels: train-mode BatchNorm + activation forward, and their backward.
// NHWC, one thread = one pixel x one 16-byte channel vector; per-channel reductions go through a per-block LDS
// tree and one fp32 atomic per channel per block (Guideline 12).  Statistics and gradients of gamma/beta are fp32.
#include "common.h"

namespace yh {

template <typename T> struct TV;
template <> struct TV<f16> { typedef f16x8 type; static constexpr int N = 8; };
template <> struct TV<float> { typedef f32x4 type; static constexpr int N = 4; };

// derivative of the activation w.r.t. its pre-activation u
__device__ __forceinline__ float act_grad(float u, int act, float slope) {
    switch (act) {
        case YH_ACT_LEAKY: return u > 0.f ? 1.f : slope;
        case YH_ACT_RELU: return u > 0.f ? 1.f : 0.f;
        case YH_ACT_RELU6: return (u > 0.f && u < 6.f) ? 1.f : 0.f;
        case YH_ACT_HSWISH: return u <= -3.f ? 0.f : (u >= 3.f ? 1.f : (2.f * u + 3.f) / 6.f);
        case YH_ACT_MISH: {
            const float e = expf(fminf(u, 20.f));
            const float n = e * (e + 2.f);
            const float t = u > 20.f ? 1.f : n / (n + 2.f);          // tanh(softplus(u))
            const float sg = 1.f / (1.f + expf(-u));
            return t + u * sg * (1.f - t * t);
        }
        default: return 1.f;
    }
}

// Block-level per-channel reduction of NV partial sums per thread for the channel group this block owns.
// Layout: grid.x = channel groups, grid.y = pixel chunks; threads stride over the chunk's pixels.
template <typename T, int NQ>
__device__ __forceinline__ void block_reduce_atomic(float (&acc)[NQ][TV<T>::N], float* const (&dst)[NQ], int cbase) {
    constexpr int VN = TV<T>::N;
    __shared__ float red[NQ * VN * 4];  // one slot per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            float v = acc[q][e];
            for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
            if (lane == 0) red[(q * VN + e) * 4 + wave] = v;
        }
    __syncthreads();
    if (threadIdx.x < NQ * VN) {
        const int q = threadIdx.x / VN, e = threadIdx.x % VN;
        const float v = red[threadIdx.x * 4] + red[threadIdx.x * 4 + 1] + red[threadIdx.x * 4 + 2] + red[threadIdx.x * 4 + 3];
        atomicAdd(dst[q] + cbase + e, v);
    }
}

constexpr int PIX_PER_BLOCK = 4096;

template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const yh_bn_desc d) {
    typedef typename TV<T>::type V;
    constexpr int VN = TV<T>::N;
    const int g = blockIdx.x;
    const long p0 = (long)blockIdx.y * PIX_PER_BLOCK;
    const long p1 = min(p0 + PIX_PER_BLOCK, (long)d.pixels);
    const T* z = reinterpret_cast<const T*>(d.z) + g * VN;
    float acc[2][VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) acc[0][e] = acc[1][e] = 0.f;
    for (long p = p0 + threadIdx.x; p < p1; p += 256) {
        const V v = *reinterpret_cast<const V*>(z + p * d.ldz);
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const float f = (float)v[e];
            acc[0][e] += f;
            acc[1][e] = fmaf(f, f, acc[1][e]);
        }
    }
    float* const dst[2] = {d.sum, d.sumsq};
    block_reduce_atomic<T, 2>(acc, dst, g * VN);
}

__global__ void bn_finalize_kernel(const yh_bn_desc d) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.c) return;
    const float P = (float)d.pixels;
    const float mean = d.sum[c] / P;
    const float var = fmaxf(d.sumsq[c] / P - mean * mean, 0.f);
    d.mean[c] = mean;
    d.invstd[c] = 1.f / sqrtf(var + d.eps);
    if (d.running_mean) d.running_mean[c] = (1.f - d.momentum) * d.running_mean[c] + d.momentum * mean;
    if (d.running_var) {
        const float unbiased = d.pixels > 1 ? var * P / (P - 1.f) : var;
        d.running_var[c] = (1.f - d.momentum) * d.running_var[c] + d.momentum * unbiased;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const yh_bn_desc d) {
    typedef typename TV<T>::type V;
    constexpr int VN = TV<T>::N;
    const int cg = d.c / VN;
    const long total = d.pixels * cg;
    const T* z = reinterpret_cast<const T*>(d.z);
    const T* res = reinterpret_cast<const T*>(d.res);
    T* y = reinterpret_cast<T*>(d.out);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const long pix = i / cg;
        const V v = *reinterpret_cast<const V*>(z + pix * d.ldz + g * VN);
        float o[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const int c = g * VN + e;
            float u;
            if (d.gamma) u = d.gamma[c] * (((float)v[e] - d.mean[c]) * d.invstd[c]) + d.beta[c];
            else u = (float)v[e] + (d.beta ? d.beta[c] : 0.f);
            o[e] = activate(u, d.act, d.slope);
        }
        if (res) {
            const V r = *reinterpret_cast<const V*>(res + pix * d.ldr + g * VN);
#pragma unroll
            for (int e = 0; e < VN; ++e) o[e] += (float)r[e];
        }
        V ov;
#pragma unroll
        for (int e = 0; e < VN; ++e) ov[e] = (T)o[e];
        if (d.ups == 2) {
            const int wi = (int)(pix % d.w_in);
            const long r = pix / d.w_in;
            const int hi = (int)(r % d.h);
            const long n = r / d.h;
            const long wo2 = 2L * d.w_in;
            T* dst = y + ((n * 2 * d.h + 2L * hi) * wo2 + 2L * wi) * d.ldo + g * VN;
            *reinterpret_cast<V*>(dst) = ov;
            *reinterpret_cast<V*>(dst + d.ldo) = ov;
            *reinterpret_cast<V*>(dst + wo2 * d.ldo) = ov;
            *reinterpret_cast<V*>(dst + (wo2 + 1) * d.ldo) = ov;
        } else {
            *reinterpret_cast<V*>(y + pix * d.ldo + g * VN) = ov;
        }
    }
}

// g = dy * act'(u); accumulates sum g (-> d.sum) and sum g*xhat (-> d.sumsq)
template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const yh_bn_desc d) {
    typedef typename TV<T>::type V;
    constexpr int VN = TV<T>::N;
    const int g = blockIdx.x;
    const long p0 = (long)blockIdx.y * PIX_PER_BLOCK;
    const long p1 = min(p0 + PIX_PER_BLOCK, (long)d.pixels);
    const T* z = reinterpret_cast<const T*>(d.z) + g * VN;
    const T* dy = reinterpret_cast<const T*>(d.dy) + g * VN;
    float ga[VN], be[VN], mu[VN], is[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) {
        const int c = g * VN + e;
        ga[e] = d.gamma ? d.gamma[c] : 1.f;
        be[e] = d.beta ? d.beta[c] : 0.f;
        mu[e] = d.gamma ? d.mean[c] : 0.f;
        is[e] = d.gamma ? d.invstd[c] : 1.f;
    }
    float acc[2][VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) acc[0][e] = acc[1][e] = 0.f;
    for (long p = p0 + threadIdx.x; p < p1; p += 256) {
        const V zv = *reinterpret_cast<const V*>(z + p * d.ldz);
        const V gv = *reinterpret_cast<const V*>(dy + p * d.lddy);
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const float xh = ((float)zv[e] - mu[e]) * is[e];
            const float u = ga[e] * xh + be[e];
            const float gg = (float)gv[e] * act_grad(u, d.act, d.slope);
            acc[0][e] += gg;
            acc[1][e] = fmaf(gg, xh, acc[1][e]);
        }
    }
    float* const dst[2] = {d.sum, d.sumsq};
    block_reduce_atomic<T, 2>(acc, dst, g * VN);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const yh_bn_desc d) {
    typedef typename TV<T>::type V;
    constexpr int VN = TV<T>::N;
    const int cg = d.c / VN;
    const long total = d.pixels * cg;
    const T* z = reinterpret_cast<const T*>(d.z);
    const T* dy = reinterpret_cast<const T*>(d.dy);
    T* dz = reinterpret_cast<T*>(d.out);
    const float invP = 1.f / (float)d.pixels;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const long pix = i / cg;
        const V zv = *reinterpret_cast<const V*>(z + pix * d.ldz + g * VN);
        const V gv = *reinterpret_cast<const V*>(dy + pix * d.lddy + g * VN);
        V ov;
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const int c = g * VN + e;
            if (d.gamma) {
                const float xh = ((float)zv[e] - d.mean[c]) * d.invstd[c];
                const float u = d.gamma[c] * xh + d.beta[c];
                const float gg = (float)gv[e] * act_grad(u, d.act, d.slope);
                ov[e] = (T)(d.gamma[c] * d.invstd[c] * (gg - d.sum[c] * invP - xh * d.sumsq[c] * invP));
            } else {
                const float u = (float)zv[e] + (d.beta ? d.beta[c] : 0.f);
                ov[e] = (T)((float)gv[e] * act_grad(u, d.act, d.slope));
            }
        }
        *reinterpret_cast<V*>(dz + pix * d.ldo + g * VN) = ov;
    }
}

static inline unsigned grid1(long total) {
    long g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

static int check_bn(const yh_bn_desc* d, bool need_dy, bool need_out) {
    if (!d || !d->z || d->pixels <= 0 || d->c <= 0) return YH_EINVAL;
    if (d->dtype != YH_F16 && d->dtype != YH_F32) return YH_EINVAL;
    const int v = d->dtype == YH_F16 ? 8 : 4;
    if (d->c % v || d->ldz % v || !aligned16(d->z)) return YH_EALIGN;
    if (need_dy && (!d->dy || d->lddy % v || !aligned16(d->dy))) return YH_EALIGN;
    if (need_out && (!d->out || d->ldo % v || !aligned16(d->out))) return YH_EALIGN;
    if (d->res && (d->ldr % v || !aligned16(d->res))) return YH_EALIGN;
    if (d->gamma && (!d->beta || !d->mean || !d->invstd)) return YH_EINVAL;
    return YH_OK;
}

}  // namespace yh

using namespace yh;

extern "C" int yh_bn_stats(const yh_bn_desc* d, void* stream) {
    int rc = check_bn(d, false, false);
    if (rc) return rc;
    if (!d->sum || !d->sumsq) return YH_EINVAL;
    const int v = d->dtype == YH_F16 ? 8 : 4;
    const dim3 grid(d->c / v, (unsigned)((d->pixels + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK));
    if (d->dtype == YH_F16) hipLaunchKernelGGL(bn_stats_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, *d);
    else hipLaunchKernelGGL(bn_stats_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, *d);
    return check_launch();
}

extern "C" int yh_bn_finalize(const yh_bn_desc* d, void* stream) {
    if (!d || !d->sum || !d->sumsq || !d->mean || !d->invstd || d->c <= 0 || d->pixels <= 0) return YH_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((d->c + 255) / 256), dim3(256), 0, (hipStream_t)stream, *d);
    return check_launch();
}

extern "C" int yh_bn_act_fwd(const yh_bn_desc* d, void* stream) {
    int rc = check_bn(d, false, true);
    if (rc) return rc;
    if (d->ups != 1 && d->ups != 2) return YH_EINVAL;
    if (d->ups == 2 && (long)d->n * d->h * d->w_in != d->pixels) return YH_EINVAL;
    const int v = d->dtype == YH_F16 ? 8 : 4;
    const long total = d->pixels * (d->c / v);
    if (d->dtype == YH_F16) hipLaunchKernelGGL(bn_act_fwd_kernel<f16>, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, *d);
    else hipLaunchKernelGGL(bn_act_fwd_kernel<float>, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, *d);
    return check_launch();
}

extern "C" int yh_bn_act_bwd_reduce(const yh_bn_desc* d, void* stream) {
    int rc = check_bn(d, true, false);
    if (rc) return rc;
    if (!d->sum || !d->sumsq) return YH_EINVAL;
    const int v = d->dtype == YH_F16 ? 8 : 4;
    const dim3 grid(d->c / v, (unsigned)((d->pixels + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK));
    if (d->dtype == YH_F16) hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, *d);
    else hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, *d);
    return check_launch();
}

extern "C" int yh_bn_act_bwd_apply(const yh_bn_desc* d, void* stream) {
    int rc = check_bn(d, true, true);
    if (rc) return rc;
    if (d->gamma && (!d->sum || !d->sumsq)) return YH_EINVAL;
    const int v = d->dtype == YH_F16 ? 8 : 4;
    const long total = d->pixels * (d->c / v);
    if (d->dtype == YH_F16) hipLaunchKernelGGL(bn_act_bwd_apply_kernel<f16>, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, *d);
    else hipLaunchKernelGGL(bn_act_bwd_apply_kernel<float>, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, *d);
    return check_launch();
}
