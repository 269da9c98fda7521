// int8 (PTQ eval) forms of the data-movement blocks: re-quantising copy / upsample, max-pool, quantised shortcut.
// Reference arithmetic: utils/quantized/quantized_ptq_cos.py (Round :14-20, shortcut :877-912,1029, concat :1540-1545).
// One thread = one pixel x 16 channels (16 bytes).
#include "common.h"

namespace yh {

__device__ __forceinline__ float rnd_away(float t) { return copysignf(floorf(fabsf(t) + 0.5f), t); }
__device__ __forceinline__ float clamp_i8(float t) { return fminf(fmaxf(t, -128.f), 127.f); }

struct B16 {
    int8_t v[16];
};
__device__ __forceinline__ B16 ld16(const int8_t* p) {
    B16 r;
    *reinterpret_cast<uint4*>(r.v) = *reinterpret_cast<const uint4*>(p);
    return r;
}
__device__ __forceinline__ void st16(int8_t* p, const B16& r) { *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(r.v); }

__global__ __launch_bounds__(256) void qcopy_kernel(const yh_qcopy_desc d) {
    const int cg = d.c / 16;
    const long total = (long)d.n * d.h * d.w_in * cg;
    const int8_t* x = reinterpret_cast<const int8_t*>(d.x);
    int8_t* y = reinterpret_cast<int8_t*>(d.y);
    const int u = d.ups;
    const bool requant = d.ratio != 1.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const long pix = i / cg;
        B16 v = ld16(x + pix * d.ldx + g * 16);
        if (requant) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v.v[e] = (int8_t)(int)clamp_i8(rnd_away((float)v.v[e] * d.ratio));
        }
        if (u == 1) {
            st16(y + pix * d.ldy + g * 16, v);
        } else {
            const int wi = (int)(pix % d.w_in);
            const long r = pix / d.w_in;
            const int hi = (int)(r % d.h);
            const long n = r / d.h;
            const long wo_n = (long)d.w_in * u;
            for (int dy = 0; dy < u; ++dy)
                for (int dx = 0; dx < u; ++dx)
                    st16(y + ((n * d.h * u + (long)hi * u + dy) * wo_n + (long)wi * u + dx) * d.ldy + g * 16, v);
        }
    }
}

__global__ __launch_bounds__(256) void qpool_kernel(const yh_pool_desc d) {
    const int cg = d.c / 16;
    const long total = (long)d.n * d.ho * d.wo * cg;
    const int8_t* x = reinterpret_cast<const int8_t*>(d.x);
    int8_t* y = reinterpret_cast<int8_t*>(d.y);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        long r = i / cg;
        const int wo = (int)(r % d.wo);
        r /= d.wo;
        const int ho = (int)(r % d.ho);
        const int n = (int)(r / d.ho);
        int m[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) m[e] = -128;
        for (int dy = 0; dy < d.k; ++dy) {
            const int hi = ho * d.stride - d.pad_lo + dy;
            for (int dx = 0; dx < d.k; ++dx) {
                const int wi = wo * d.stride - d.pad_lo + dx;
                if ((unsigned)hi < (unsigned)d.h && (unsigned)wi < (unsigned)d.w_in) {
                    const B16 v = ld16(x + (((long)n * d.h + hi) * d.w_in + wi) * d.ldx + g * 16);
#pragma unroll
                    for (int e = 0; e < 16; ++e) m[e] = max(m[e], (int)v.v[e]);
                } else if (d.edge_zero && hi >= 0 && wi >= 0) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) m[e] = max(m[e], 0);
                }
            }
        }
        B16 o;
#pragma unroll
        for (int e = 0; e < 16; ++e) o.v[e] = (int8_t)m[e];
        st16(y + (((long)n * d.ho + ho) * d.wo + wo) * d.ldy + g * 16, o);
    }
}

__global__ __launch_bounds__(256) void qadd_kernel(const yh_qadd_desc d) {
    const int cg = d.c / 16;
    const long total = d.pixels * cg;
    const int8_t* x = reinterpret_cast<const int8_t*>(d.x);
    const int8_t* a = reinterpret_cast<const int8_t*>(d.a);
    int8_t* y = reinterpret_cast<int8_t*>(d.y);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const long pix = i / cg;
        const B16 vx = ld16(x + pix * d.ldx + g * 16);
        const B16 va = ld16(a + pix * d.lda + g * 16);
        B16 o;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float xq = rnd_away((float)vx.v[e] * d.rx) * d.scale_x;
            const float aq = rnd_away((float)va.v[e] * d.ra) * d.scale_a;
            o.v[e] = (int8_t)(int)clamp_i8(rnd_away((xq + aq) * d.inv_scale_sum));
        }
        st16(y + pix * d.ldy + g * 16, o);
    }
}

static inline unsigned grid_q(long total) {
    long g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace yh

using namespace yh;

extern "C" int yh_qcopy(const yh_qcopy_desc* d, void* stream) {
    if (!d || !d->x || !d->y || d->n <= 0 || d->c <= 0 || (d->ups != 1 && d->ups != 2) || !(d->ratio > 0.f)) return YH_EINVAL;
    if (d->c % 16 || d->ldx % 16 || d->ldy % 16 || !aligned16(d->x) || !aligned16(d->y)) return YH_EALIGN;
    const long total = (long)d->n * d->h * d->w_in * (d->c / 16);
    hipLaunchKernelGGL(qcopy_kernel, dim3(grid_q(total)), dim3(256), 0, (hipStream_t)stream, *d);
    return check_launch();
}

extern "C" int yh_qpool(const yh_pool_desc* d, void* stream) {
    if (!d || !d->x || !d->y || d->n <= 0 || d->c <= 0 || d->k <= 0 || d->stride <= 0 || d->dtype != YH_I8) return YH_EINVAL;
    if (d->c % 16 || d->ldx % 16 || d->ldy % 16 || !aligned16(d->x) || !aligned16(d->y)) return YH_EALIGN;
    const long total = (long)d->n * d->ho * d->wo * (d->c / 16);
    hipLaunchKernelGGL(qpool_kernel, dim3(grid_q(total)), dim3(256), 0, (hipStream_t)stream, *d);
    return check_launch();
}

// Self-test of the int8 epilogues' Mish (common.h mish_for_grid) over float bit patterns [bits0, bits1): out[0] = values whose grid
// index round_clamp(mish * inv_s) differs between the exact form (common.h mish_f64: activate()'s unless -DYH_QMISH_TIE_F64) and mish_for_grid, out[1] = max relative difference (as float bits,
// in units of 1e-9) between the exact form and mish_fast over the values of at least a quarter grid step, out[2] = values on which the exact form was consulted.
__global__ __launch_bounds__(256) void qmish_selftest_kernel(unsigned bits0, unsigned bits1, float inv_s, unsigned long long* out) {
    unsigned long long bad = 0, slow = 0;
    unsigned worst = 0;
    const unsigned long long n = (unsigned long long)bits1 - bits0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float v = __uint_as_float(bits0 + (unsigned)i);
        if (!(fabsf(v) <= 64.f)) continue;
        const float ye = mish_f64(v), yf = mish_fast(v), yg = mish_for_grid(v, inv_s);
        const float qe = fminf(fmaxf(copysignf(floorf(fabsf(ye * inv_s) + 0.5f), ye * inv_s), -128.f), 127.f);
        const float qg = fminf(fmaxf(copysignf(floorf(fabsf(yg * inv_s) + 0.5f), yg * inv_s), -128.f), 127.f);
        float qq[1];
        const float vv[1] = {v};
        mish_quantize_n<1>(vv, inv_s, qq);      // the fused tie test + rounding of the conv epilogues (round 6)
        bad += qe != qg;
        bad += qe != qq[0];
        slow += __float_as_uint(yg) != __float_as_uint(yf);
        if (fabsf(ye) * inv_s >= 0.25f) {      // where the value can reach a rounding tie at all (far below zero Mish is ~1e-20 and rounds to 0)
            const float rel = fabsf(yf - ye) / fabsf(ye) * 1e9f;
            worst = max(worst, (unsigned)fminf(rel, 4e9f));
        }
    }
    atomicAdd(out, bad);
    atomicMax(out + 1, (unsigned long long)worst);
    atomicAdd(out + 2, slow);
}

extern "C" int yh_qmish_selftest(uint32_t bits0, uint32_t bits1, float inv_s, uint64_t* out, void* stream) {
    if (!out || bits1 < bits0 || !(inv_s > 0.f)) return YH_EINVAL;
    hipLaunchKernelGGL(qmish_selftest_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, bits0, bits1, inv_s,
                       reinterpret_cast<unsigned long long*>(out));
    return check_launch();
}

extern "C" int yh_qadd(const yh_qadd_desc* d, void* stream) {
    if (!d || !d->x || !d->a || !d->y || d->pixels <= 0 || d->c <= 0) return YH_EINVAL;
    if (!(d->rx > 0.f) || !(d->ra > 0.f) || !(d->scale_x > 0.f) || !(d->scale_a > 0.f) || !(d->inv_scale_sum > 0.f)) return YH_EINVAL;
    if (d->c % 16 || d->ldx % 16 || d->lda % 16 || d->ldy % 16 || !aligned16(d->x) || !aligned16(d->a) || !aligned16(d->y)) return YH_EALIGN;
    const long total = d->pixels * (d->c / 16);
    hipLaunchKernelGGL(qadd_kernel, dim3(grid_q(total)), dim3(256), 0, (hipStream_t)stream, *d);
    return check_launch();
}
