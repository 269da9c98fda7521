// Training-path kernels: train-mode BatchNorm + activation forward, and their backward.
// NHWC, one thread = one pixel x one 16-byte channel vector; per-channel reductions go through a per-block LDS
// tree and one fp32 atomic per channel per block (Guideline 12).  Statistics and gradients of gamma/beta are fp32.
#include <stdio.h>

#include "common.h"

#ifndef YH_BN_UNROLL_FWD
#define YH_BN_UNROLL_FWD 4
#define YH_BN_UNROLL_RED 2
#define YH_BN_UNROLL_APP 4
#endif
#ifndef YH_BN_REVERSE_DEFAULT
#define YH_BN_REVERSE_DEFAULT 4
#endif

namespace yh {

template <typename T> struct TV;
template <> struct TV<f16> { typedef f16x8 type; static constexpr int N = 8; };
template <> struct TV<float> { typedef f32x4 type; static constexpr int N = 4; };

// Thread mapping shared by all five kernels: a workgroup covers `cgb` channel groups (16 bytes each) x `rows` pixels
// per pass, consecutive lanes on consecutive channel groups of the same pixel, so a wave reads whole contiguous rows
// (1 KB per instruction when the tensor is dense).  Each thread keeps ONE channel group for its whole pixel range:
// the per-channel parameters live in registers, loaded once.
struct BnGeom {
    int cgs;      // channel groups in the tensor (c / VN)
    int cgb;      // channel groups per workgroup (<= 256)
    int rows;     // pixels per pass = 256 / cgb
    int ppb;      // pixels per workgroup
    int two_stage;  // reductions: partial sums to the workspace + a summing launch (no atomics)
    int reverse;    // walk the pixel chunks from the END of the tensor (workgroup y -> chunk gridDim.y - 1 - y): the producer wrote / read the
                    // tensor front to back, so its tail is what the 256 MiB Infinity Cache still holds when this pass starts (YH_BN_REVERSE)
};

template <int VN>
__device__ __forceinline__ void load8(const float* p, int c0, float (&dst)[VN], float fill) {
#pragma unroll
    for (int e = 0; e < VN; ++e) dst[e] = p ? p[c0 + e] : fill;
}

// Sum acc[NQ][VN] over the `rows` threads that share a channel group, then one atomic per channel per workgroup.
template <int NQ, int VN>
__device__ __forceinline__ void rows_reduce_atomic(float (&acc)[NQ][VN], float* const (&dst)[NQ], const BnGeom& gm, int cgl,
                                                   int prow, int c0, bool active, float* part, int c_total) {
    __shared__ float red[256 * NQ * VN];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < VN; ++e) red[(q * VN + e) * 256 + threadIdx.x] = active ? acc[q][e] : 0.f;
    __syncthreads();
    // the `rows` threads of one channel group share the NQ*VN columns; each sums its columns over the rows
    if (active)
        for (int it = prow; it < NQ * VN; it += gm.rows) {
            float v = 0.f;
            for (int r = 0; r < gm.rows; ++r) v += red[it * 256 + r * gm.cgb + cgl];
            if (part) part[((long)blockIdx.y * NQ + it / VN) * c_total + c0 + it % VN] = v;  // summed by bn_partials_kernel
            else atomicAdd(dst[it / VN] + c0 + it % VN, v);
        }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const yh_bn_desc d, const BnGeom gm) {
    typedef typename TV<T>::type V;
    constexpr int VN = TV<T>::N;
    const int cgl = threadIdx.x % gm.cgb, prow = threadIdx.x / gm.cgb;
    const int g = blockIdx.x * gm.cgb + cgl;
    const bool lane_ok = prow < gm.rows && g < gm.cgs;
    const long p0 = (long)(gm.reverse ? gridDim.y - 1 - blockIdx.y : blockIdx.y) * gm.ppb;
    const long p1 = min(p0 + gm.ppb, (long)d.pixels);
    const T* z = reinterpret_cast<const T*>(d.z) + g * VN;
    float acc[2][VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) acc[0][e] = acc[1][e] = 0.f;
    if (lane_ok)
        for (long p = p0 + prow; p < p1; p += gm.rows) {
            const V v = *reinterpret_cast<const V*>(z + p * d.ldz);
#pragma unroll
            for (int e = 0; e < VN; ++e) {
                const float f = (float)v[e];
                acc[0][e] += f;
                acc[1][e] = fmaf(f, f, acc[1][e]);
            }
        }
    float* const dst[2] = {d.sum, d.sumsq};
    rows_reduce_atomic<2, VN>(acc, dst, gm, cgl, prow, g * VN, lane_ok, gm.two_stage ? d.ws : nullptr, d.c);
}

// second stage of the reductions: sum[c] += sum over workgroups of part[wg][0][c], sumsq likewise.
// 16 columns x 16 partial-lanes per workgroup (a wave reads 64-byte row pieces of 4 partial rows at a time), row groups
// in grid.y meeting in a few atomics per channel.
__global__ __launch_bounds__(1024) void bn_partials_kernel(const float* part, int nparts, int per_group, int c, float* s0, float* s1) {
    // blockDim.x = 16 columns x L row lanes (L = 16: several row groups that meet in atomics; L = 64 and ONE group: the deterministic
    // form - every channel's sum is taken by one thread in a fixed order, common.h deterministic())
    __shared__ float red[1024];
    const int L = blockDim.x >> 4;
    const int col = blockIdx.x * 16 + (threadIdx.x & 15), lane = threadIdx.x >> 4;
    const int r0 = blockIdx.y * per_group, r1 = min(r0 + per_group, nparts);
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (col < 2 * c) {
        const int q = col / c, ch = col - q * c;
        const float* src = part + (long)q * c + ch;
        const long pitch = 2L * c;
        int k = r0 + lane;
        for (; k + 3 * L < r1; k += 4 * L) {      // four independent chains: the loads pipeline
            v0 += src[k * pitch];
            v1 += src[(k + L) * pitch];
            v2 += src[(k + 2 * L) * pitch];
            v3 += src[(k + 3 * L) * pitch];
        }
        for (; k < r1; k += L) v0 += src[k * pitch];
    }
    red[threadIdx.x] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (threadIdx.x < 16 && col < 2 * c) {
        float t = 0.f;
        for (int r = 0; r < L; ++r) t += red[r * 16 + threadIdx.x];
        const int q = col / c, ch = col - q * c;
        float* dst = (q ? s1 : s0) + ch;
        if (gridDim.y == 1) *dst += t;            // the only writer of this channel: plain accumulate
        else atomicAdd(dst, t);
    }
}

// groups > 0 (deterministic form): bn_conv_partials_kernel left row group g's column sums in row g * per_group of the workspace; they
// are added here in group order - no atomics, and no launch more than the round-5 form had
__global__ void bn_finalize_kernel(const yh_bn_desc d, const int groups, const int per_group) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.c) return;
    const float P = (float)d.pixels;
    if (groups > 0) {
        float s0 = d.sum[c], s1 = d.sumsq[c];
        const float* row = d.ws + c;
        const long pitch = (long)per_group * 2 * d.c;
        for (int g = 0; g < groups; ++g) {
            s0 += row[g * pitch];
            s1 += row[g * pitch + d.c];
        }
        d.sum[c] = s0;
        d.sumsq[c] = s1;
    }
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

// U = pixels per thread whose loads are issued before the first of them is used (round 6).  With U = 1 (rounds 1 - 5) a thread has ONE
// 16-byte load in flight: 8 waves per SIMD x 64 lanes x 16 B = 32 KB per CU, and at ~2 us of loaded memory latency that is 4 - 5 TB/s -
// exactly where these passes sat while a plain torch elementwise kernel moved the same bytes at 6.0 (tools/probe/run_bn_probe.py,
// profiles/r06_bn_probe.txt).  ACT: the activation as a COMPILE-TIME constant (round 6) - with the run-time `d.act` the switch of
// activate() / act_grad() was expanded per element inside the loop: five scalar branches per value, ~100 per trip, in a kernel that
// should be nothing but loads, ~12 VALU operations per value and stores.  Same arithmetic per element in every form: bit-identical.
template <typename T, int U, int ACT>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const yh_bn_desc d, const BnGeom gm) {
    typedef typename TV<T>::type V;
    constexpr int VN = TV<T>::N;
    const int cgl = threadIdx.x % gm.cgb, prow = threadIdx.x / gm.cgb;
    const int g = blockIdx.x * gm.cgb + cgl;
    if (prow >= gm.rows || g >= gm.cgs) return;
    const long p0 = (long)(gm.reverse ? gridDim.y - 1 - blockIdx.y : blockIdx.y) * gm.ppb;
    const long p1 = min(p0 + gm.ppb, (long)d.pixels);
    const int c0 = g * VN;
    const T* z = reinterpret_cast<const T*>(d.z) + c0;
    const bool has_res = d.res != nullptr;      // wave-uniform (a test of the per-thread pointer below is compiled as divergent)
    const T* res = reinterpret_cast<const T*>(d.res) + c0;
    T* y = reinterpret_cast<T*>(d.out) + c0;
    float ga[VN], be[VN], mu[VN], is[VN];
    load8<VN>(d.gamma, c0, ga, 1.f);
    load8<VN>(d.beta, c0, be, 0.f);
    load8<VN>(d.gamma ? d.mean : nullptr, c0, mu, 0.f);
    load8<VN>(d.gamma ? d.invstd : nullptr, c0, is, 1.f);
    auto finish = [&](const long p, const V& v, const V& r) {
        float o[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) o[e] = activate_train_t<T>(ga[e] * (((float)v[e] - mu[e]) * is[e]) + be[e], ACT, d.slope);
        if (has_res) {
#pragma unroll
            for (int e = 0; e < VN; ++e) o[e] += (float)r[e];
        }
        V ov;
#pragma unroll
        for (int e = 0; e < VN; ++e) ov[e] = (T)o[e];
        if (d.ups == 2) {
            const int wi = (int)(p % d.w_in);
            const long r = p / d.w_in;
            const int hi = (int)(r % d.h);
            const long n = r / d.h;
            const long wo2 = 2L * d.w_in;
            T* dst = y + ((n * 2 * d.h + 2L * hi) * wo2 + 2L * wi) * d.ldo;
            *reinterpret_cast<V*>(dst) = ov;
            *reinterpret_cast<V*>(dst + d.ldo) = ov;
            *reinterpret_cast<V*>(dst + wo2 * d.ldo) = ov;
            *reinterpret_cast<V*>(dst + (wo2 + 1) * d.ldo) = ov;
        } else {
            *reinterpret_cast<V*>(y + p * d.ldo) = ov;
        }
    };
    long p = p0 + prow;
    const long step = gm.rows;
    if constexpr (U > 1) {
        for (; p + (U - 1) * step < p1; p += U * step) {
            V v[U], r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const V*>(z + (p + u * step) * d.ldz);
            if (has_res) {
#pragma unroll
                for (int u = 0; u < U; ++u) r[u] = *reinterpret_cast<const V*>(res + (p + u * step) * d.ldr);
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) r[u] = v[u];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) finish(p + u * step, v[u], r[u]);
        }
    }
    for (; p < p1; p += step) {
        const V v = *reinterpret_cast<const V*>(z + p * d.ldz);
        const V r = has_res ? *reinterpret_cast<const V*>(res + p * d.ldr) : v;
        finish(p, v, r);
    }
}

// g = dy * act'(u); accumulates sum g (-> d.sum) and sum g*xhat (-> d.sumsq)
template <typename T, int U, int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const yh_bn_desc d, const BnGeom gm) {
    typedef typename TV<T>::type V;
    constexpr int VN = TV<T>::N;
    const int cgl = threadIdx.x % gm.cgb, prow = threadIdx.x / gm.cgb;
    const int g = blockIdx.x * gm.cgb + cgl;
    const bool lane_ok = prow < gm.rows && g < gm.cgs;
    const long p0 = (long)(gm.reverse ? gridDim.y - 1 - blockIdx.y : blockIdx.y) * gm.ppb;
    const long p1 = min(p0 + gm.ppb, (long)d.pixels);
    const int c0 = g * VN;
    const T* z = reinterpret_cast<const T*>(d.z) + c0;
    const T* dy = reinterpret_cast<const T*>(d.dy) + c0;
    float ga[VN], be[VN], mu[VN], is[VN];
    float acc[2][VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) acc[0][e] = acc[1][e] = 0.f;
    if (lane_ok) {
        load8<VN>(d.gamma, c0, ga, 1.f);
        load8<VN>(d.beta, c0, be, 0.f);
        load8<VN>(d.gamma ? d.mean : nullptr, c0, mu, 0.f);
        load8<VN>(d.gamma ? d.invstd : nullptr, c0, is, 1.f);
        auto add = [&](const V& zv, const V& gv) {
#pragma unroll
            for (int e = 0; e < VN; ++e) {
                const float xh = ((float)zv[e] - mu[e]) * is[e];
                const float u = ga[e] * xh + be[e];
                const float gg = (float)gv[e] * act_grad_t<T>(u, ACT, d.slope);
                acc[0][e] += gg;
                acc[1][e] = fmaf(gg, xh, acc[1][e]);
            }
        };
        long p = p0 + prow;
        const long step = gm.rows;
        if constexpr (U > 1) {      // same pixel order per thread as U = 1: the sums are bit-identical
            for (; p + (U - 1) * step < p1; p += U * step) {
                V zv[U], gv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    zv[u] = *reinterpret_cast<const V*>(z + (p + u * step) * d.ldz);
                    gv[u] = *reinterpret_cast<const V*>(dy + (p + u * step) * d.lddy);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) add(zv[u], gv[u]);
            }
        }
        for (; p < p1; p += step) {
            const V zv = *reinterpret_cast<const V*>(z + p * d.ldz);
            const V gv = *reinterpret_cast<const V*>(dy + p * d.lddy);
            add(zv, gv);
        }
    }
    float* const dst[2] = {d.sum, d.sumsq};
    rows_reduce_atomic<2, VN>(acc, dst, gm, cgl, prow, c0, lane_ok, gm.two_stage ? d.ws : nullptr, d.c);
}

template <typename T, int U, int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const yh_bn_desc d, const BnGeom gm) {
    typedef typename TV<T>::type V;
    constexpr int VN = TV<T>::N;
    const int cgl = threadIdx.x % gm.cgb, prow = threadIdx.x / gm.cgb;
    const int g = blockIdx.x * gm.cgb + cgl;
    if (prow >= gm.rows || g >= gm.cgs) return;
    const long p0 = (long)(gm.reverse ? gridDim.y - 1 - blockIdx.y : blockIdx.y) * gm.ppb;
    const long p1 = min(p0 + gm.ppb, (long)d.pixels);
    const int c0 = g * VN;
    const T* z = reinterpret_cast<const T*>(d.z) + c0;
    const T* dy = reinterpret_cast<const T*>(d.dy) + c0;
    T* dz = reinterpret_cast<T*>(d.out) + c0;
    const float invP = 1.f / (float)d.pixels;
    const bool bn = d.gamma != nullptr;
    float ga[VN], be[VN], mu[VN], is[VN], m1[VN], m2[VN];
    load8<VN>(d.gamma, c0, ga, 1.f);
    load8<VN>(d.beta, c0, be, 0.f);
    load8<VN>(bn ? d.mean : nullptr, c0, mu, 0.f);
    load8<VN>(bn ? d.invstd : nullptr, c0, is, 1.f);
    load8<VN>(bn ? d.sum : nullptr, c0, m1, 0.f);
    load8<VN>(bn ? d.sumsq : nullptr, c0, m2, 0.f);
#pragma unroll
    for (int e = 0; e < VN; ++e) { m1[e] *= invP; m2[e] *= invP; }
    auto finish = [&](const long p, const V& zv, const V& gv) {
        V ov;
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const float xh = ((float)zv[e] - mu[e]) * is[e];
            const float u = ga[e] * xh + be[e];
            const float gg = (float)gv[e] * act_grad_t<T>(u, ACT, d.slope);
            ov[e] = bn ? (T)(ga[e] * is[e] * (gg - m1[e] - xh * m2[e])) : (T)gg;
        }
        *reinterpret_cast<V*>(dz + p * d.ldo) = ov;
    };
    long p = p0 + prow;
    const long step = gm.rows;
    if constexpr (U > 1) {
        for (; p + (U - 1) * step < p1; p += U * step) {
            V zv[U], gv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                zv[u] = *reinterpret_cast<const V*>(z + (p + u * step) * d.ldz);
                gv[u] = *reinterpret_cast<const V*>(dy + (p + u * step) * d.lddy);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) finish(p + u * step, zv[u], gv[u]);
        }
    }
    for (; p < p1; p += step) {
        const V zv = *reinterpret_cast<const V*>(z + p * d.ldz);
        const V gv = *reinterpret_cast<const V*>(dy + p * d.lddy);
        finish(p, zv, gv);
    }
}

// pixels in flight per thread of the three streaming passes (0: forward, 1: backward reduce, 2: backward apply); A/B knob
// YH_BN_UNROLL = "f,r,a" or one number for all three
static int bn_unroll(int which) {
    static int v[3] = {-1, -1, -1};
    if (v[0] < 0) {
        int t[3] = {YH_BN_UNROLL_FWD, YH_BN_UNROLL_RED, YH_BN_UNROLL_APP};
        const char* e = getenv("YH_BN_UNROLL");
        if (e) {
            int a = 0, b = 0, c = 0;
            const int n = sscanf(e, "%d,%d,%d", &a, &b, &c);
            if (n == 1) t[0] = t[1] = t[2] = a;
            else if (n == 3) { t[0] = a; t[1] = b; t[2] = c; }
        }
        v[2] = t[2]; v[1] = t[1]; v[0] = t[0];
    }
    return v[which];
}

// bit 0: forward pass, bit 1: backward reduce, bit 2: backward apply walk their tensor back to front (A/B knob YH_BN_REVERSE)
static int bn_reverse() {
    static const int v = [] { const char* e = getenv("YH_BN_REVERSE"); return e ? atoi(e) : YH_BN_REVERSE_DEFAULT; }();
    return v;
}

static BnGeom bn_geom(const yh_bn_desc* d, int vn, dim3* grid, int target = 4096) {
    BnGeom gm;
    gm.cgs = d->c / vn;
    gm.cgb = gm.cgs < 256 ? gm.cgs : 256;
    gm.rows = 256 / gm.cgb;
    const int gx = (gm.cgs + gm.cgb - 1) / gm.cgb;
    // workgroups aimed for.  Streaming passes: 16384 since round 6 (4096 before: 152^2 x 128 forward 0.149 -> 0.134 ms, 304^2 x 64 backward
    // apply 0.45 -> 0.42 - short-lived workgroups dispatched in address order keep the chip on a compact window of memory, long-lived
    // ones spread 2048 slow streams over the whole tensor; profiles/r06_bn_probe.txt).  Reductions: every workgroup leaves a row of
    // partial sums for the second stage, ~1024.  A/B knobs: YH_BN_TARGET, YH_BN_RTARGET
    static const int env_target = [] { const char* e = getenv("YH_BN_TARGET"); return e ? atoi(e) : 16384; }();
    static const int env_rtarget = [] { const char* e = getenv("YH_BN_RTARGET"); return e ? atoi(e) : 1024; }();
    target = target == 1024 ? env_rtarget : env_target;
    long ppb = (d->pixels * gx + target - 1) / target;
    const long min_ppb = (long)gm.rows * 8;             // at least 8 passes: amortise the parameter loads / atomics
    if (ppb < min_ppb) ppb = min_ppb;
    ppb = (ppb + gm.rows - 1) / gm.rows * gm.rows;
    gm.ppb = (int)ppb;
    *grid = dim3(gx, (unsigned)((d->pixels + ppb - 1) / ppb));
    gm.two_stage = d->ws && d->ws_floats >= (int64_t)grid->y * 2 * d->c;
    gm.reverse = 0;
    return gm;
}

static void sum_partials(const yh_bn_desc* d, const dim3& grid, void* stream) {
    const int nparts = (int)grid.y;
    if (deterministic()) {      // one owner per channel: 64 row lanes x four load chains each, summed in a fixed order
        hipLaunchKernelGGL(bn_partials_kernel, dim3((2 * d->c + 15) / 16, 1), dim3(1024), 0, (hipStream_t)stream, d->ws, nparts, nparts,
                           d->c, d->sum, d->sumsq);
        return;
    }
    int groups = (nparts + 63) / 64;                        // >= 64 rows per group
    const int per_group = (nparts + groups - 1) / groups;
    groups = (nparts + per_group - 1) / per_group;
    hipLaunchKernelGGL(bn_partials_kernel, dim3((2 * d->c + 15) / 16, groups), dim3(256), 0, (hipStream_t)stream, d->ws, nparts,
                       per_group, d->c, d->sum, d->sumsq);
}

// ------------------------------------------------------------------------------------------ maxpool backward
__device__ __forceinline__ void atomic_add_elem(float* base, long idx, float v) { atomicAdd(base + idx, v); }
__device__ __forceinline__ void atomic_add_elem(f16* base, long idx, float v) {
    // packed fp16 atomic on the aligned pair that holds the element; the other half adds zero
    typedef f16 __attribute__((ext_vector_type(2))) h2;
    const h2 val = (idx & 1) ? h2{(f16)0.f, (f16)v} : h2{(f16)v, (f16)0.f};
    __builtin_amdgcn_global_atomic_fadd_v2f16(reinterpret_cast<h2 __attribute__((address_space(1)))*>((uintptr_t)(base + (idx & ~1L))), val);
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const yh_pool_bwd_desc d) {
    typedef typename TV<T>::type V;
    constexpr int VN = TV<T>::N;
    const int cg = d.c / VN;
    const long total = (long)d.n * d.ho * d.wo * cg;
    const T* x = reinterpret_cast<const T*>(d.x);
    const T* dy = reinterpret_cast<const T*>(d.dy);
    T* dx = reinterpret_cast<T*>(d.dx);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        long r = i / cg;
        const int wo = (int)(r % d.wo);
        r /= d.wo;
        const int ho = (int)(r % d.ho);
        const int n = (int)(r / d.ho);
        float m[VN];
        long arg[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) { m[e] = -INFINITY; arg[e] = -1; }
        for (int ky = 0; ky < d.k; ++ky) {
            const int hi = ho * d.stride - d.pad_lo + ky;
            for (int kx = 0; kx < d.k; ++kx) {
                const int wi = wo * d.stride - d.pad_lo + kx;
                if ((unsigned)hi < (unsigned)d.h && (unsigned)wi < (unsigned)d.w_in) {
                    const long pix = ((long)n * d.h + hi) * d.w_in + wi;
                    const V v = *reinterpret_cast<const V*>(x + pix * d.ldx + g * VN);
#pragma unroll
                    for (int e = 0; e < VN; ++e)
                        if ((float)v[e] > m[e]) { m[e] = (float)v[e]; arg[e] = pix; }
                } else if (d.edge_zero && hi >= 0 && wi >= 0) {
#pragma unroll
                    for (int e = 0; e < VN; ++e)
                        if (0.f > m[e]) { m[e] = 0.f; arg[e] = -1; }   // the zero padding wins: no gradient
                }
            }
        }
        const V gy = *reinterpret_cast<const V*>(dy + (((long)n * d.ho + ho) * d.wo + wo) * d.lddy + g * VN);
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            if (arg[e] < 0) continue;
            const long at = arg[e] * d.lddx + g * VN + e;
            if (d.stride >= d.k) dx[at] = (T)((float)dx[at] + (float)gy[e]);   // windows do not overlap: one writer per element
            else atomic_add_elem(dx, at, (float)gy[e]);
        }
    }
}

// Round 6: overlapping windows (SPP: 5 / 9 / 13 at stride 1; the tiny nets' 2 at stride 1) on planes that fit in LDS, WITHOUT atomics.
// One workgroup = one image x one channel group.  The window maximum is found separably - per row the first maximum of the k columns, then
// over the k rows the first row that holds the maximum: the same "first maximum in row-major order" as the scan above, in 2 k instead of k^2
// steps - and every input pixel then GATHERS the dy of the windows that chose it, in window order, in fp32, rounded once.  The scatter form
// met in packed-fp16 atomics (a rounding per contribution, order-dependent: the one non-reproducible kernel of YOLOv4's step) and spent
// 1.0 ms per YOLOv4-608 step on three 19 x 19 planes.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_plane_kernel(const yh_pool_bwd_desc d) {
    typedef typename TV<T>::type V;
    constexpr int VN = TV<T>::N;
    typedef short SV __attribute__((ext_vector_type(VN)));
    extern __shared__ __attribute__((aligned(16))) char pool_sm[];
    const int hw = d.h * d.w_in, hwo = d.h * d.wo, howo = d.ho * d.wo;
    V* const xs = reinterpret_cast<V*>(pool_sm);               // [h][w]
    V* const rmax = xs + hw;                                    // [h][wo]   first maximum of the row's k columns
    V* const dys = rmax + hwo;                                  // [ho][wo]
    SV* const rarg = reinterpret_cast<SV*>(dys + howo);         // [h][wo]   its column (-1: the implicit zero padding, -2: nothing)
    SV* const arg = rarg + hwo;                                 // [ho][wo]  the window's pixel h * w + w (-1: none)
    const int cg = d.c / VN;
    const int g = blockIdx.x % cg, n = blockIdx.x / cg;
    const T* const x = reinterpret_cast<const T*>(d.x) + (long)n * hw * d.ldx + g * VN;
    const T* const dy = reinterpret_cast<const T*>(d.dy) + (long)n * howo * d.lddy + g * VN;
    T* const dx = reinterpret_cast<T*>(d.dx) + (long)n * hw * d.lddx + g * VN;
    for (int p = threadIdx.x; p < hw; p += blockDim.x) xs[p] = *reinterpret_cast<const V*>(x + (long)p * d.ldx);
    for (int p = threadIdx.x; p < howo; p += blockDim.x) dys[p] = *reinterpret_cast<const V*>(dy + (long)p * d.lddy);
    __syncthreads();
    for (int i = threadIdx.x; i < hwo; i += blockDim.x) {
        const int hi = i / d.wo, wo = i - hi * d.wo;
        float m[VN];
        SV a;
#pragma unroll
        for (int e = 0; e < VN; ++e) { m[e] = -INFINITY; a[e] = -2; }
        for (int kx = 0; kx < d.k; ++kx) {
            const int wi = wo * d.stride - d.pad_lo + kx;
            if ((unsigned)wi < (unsigned)d.w_in) {
                const V v = xs[hi * d.w_in + wi];
#pragma unroll
                for (int e = 0; e < VN; ++e)
                    if ((float)v[e] > m[e]) { m[e] = (float)v[e]; a[e] = (short)wi; }
            } else if (d.edge_zero && wi >= 0) {
#pragma unroll
                for (int e = 0; e < VN; ++e)
                    if (0.f > m[e]) { m[e] = 0.f; a[e] = -1; }
            }
        }
        V mv;
#pragma unroll
        for (int e = 0; e < VN; ++e) mv[e] = (T)m[e];      // exact: a value of the tensor, 0 or -inf
        rmax[i] = mv;
        rarg[i] = a;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < howo; i += blockDim.x) {
        const int ho = i / d.wo, wo = i - ho * d.wo;
        float m[VN];
        SV a;
#pragma unroll
        for (int e = 0; e < VN; ++e) { m[e] = -INFINITY; a[e] = -1; }
        for (int ky = 0; ky < d.k; ++ky) {
            const int hi = ho * d.stride - d.pad_lo + ky;
            if ((unsigned)hi < (unsigned)d.h) {
                const V v = rmax[hi * d.wo + wo];
                const SV c = rarg[hi * d.wo + wo];
#pragma unroll
                for (int e = 0; e < VN; ++e)
                    if ((float)v[e] > m[e]) { m[e] = (float)v[e]; a[e] = c[e] < 0 ? (short)-1 : (short)(hi * d.w_in + c[e]); }
            } else if (d.edge_zero && hi >= 0) {      // a row of the implicit zero padding (its columns wi >= 0: every window has one)
#pragma unroll
                for (int e = 0; e < VN; ++e)
                    if (0.f > m[e]) { m[e] = 0.f; a[e] = -1; }
            }
        }
        arg[i] = a;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < hw; p += blockDim.x) {
        const int hi = p / d.w_in, wi = p - hi * d.w_in;
        // windows that contain the pixel: ho * stride - pad_lo <= hi <= ho * stride - pad_lo + k - 1
        const int th = hi + d.pad_lo, tw = wi + d.pad_lo;
        const int ho0 = max(0, (th - d.k + d.stride) / d.stride), ho1 = min(d.ho - 1, th / d.stride);
        const int wo0 = max(0, (tw - d.k + d.stride) / d.stride), wo1 = min(d.wo - 1, tw / d.stride);
        float acc[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[e] = 0.f;
        for (int ho = ho0; ho <= ho1; ++ho)
            for (int wo = wo0; wo <= wo1; ++wo) {
                const SV a = arg[ho * d.wo + wo];
                const V gv = dys[ho * d.wo + wo];
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[e] += a[e] == (short)p ? (float)gv[e] : 0.f;
            }
        T* const dst = dx + (long)p * d.lddx;
        const V old = *reinterpret_cast<const V*>(dst);
        V o;
#pragma unroll
        for (int e = 0; e < VN; ++e) o[e] = (T)((float)old[e] + acc[e]);
        *reinterpret_cast<V*>(dst) = o;
    }
}

// ------------------------------------------------------------------------------------------ depthwise backward
// Up to 9 taps per pass (grid.z = tap groups): a thread owns one 16-byte channel vector and 9 x VN accumulators, so dz is
// read once per group (once for 3x3, three times for 5x5) instead of once per tap, with full-width loads.
// Round 6: the pixel walk is incremental 32-bit arithmetic and the taps' offsets are per-thread constants - the round-1 form decoded every
// pixel with two 64-bit divisions and every tap with two more by the run-time kernel size (~500 instructions of index arithmetic per pixel
// around 72 multiply-adds: YOLOv3-Mobilenetv3's 15 depthwise weight gradients took 8.4 ms of its 30.7 ms training step).  Padding taps read a
// zero page instead of branching (a branch around a load makes the compiler wait for every load at once).  K: 3 / 5 compile-time, 0 run-time.
__device__ __attribute__((aligned(16))) unsigned int dw_zero_page[8] = {0, 0, 0, 0, 0, 0, 0, 0};

template <typename T, int K>
__global__ __launch_bounds__(256) void dw_wgrad_taps_kernel(const yh_dw_bwd_desc d, const BnGeom gm) {
    typedef typename TV<T>::type V;
    constexpr int VN = TV<T>::N, TG = 9;
    const int cgl = threadIdx.x % gm.cgb, prow = threadIdx.x / gm.cgb;
    const int g = blockIdx.x * gm.cgb + cgl;
    const bool ok = prow < gm.rows && g < gm.cgs;
    const int k = K ? K : d.k;
    const int t0 = blockIdx.z * TG, taps = k * k;
    const int pixels = d.n * d.ho * d.wo;
    const int p0 = blockIdx.y * gm.ppb, p1 = min(p0 + gm.ppb, pixels);
    const T* const x = reinterpret_cast<const T*>(d.x) + g * VN;
    const T* const dz = reinterpret_cast<const T*>(d.dz) + g * VN;
    const T* const zero = reinterpret_cast<const T*>(dw_zero_page);
    float acc[TG][VN];
    int tky[TG], tkx[TG], toff[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        const int tap = t0 + t;
        const int ky = tap / k, kx = tap - ky * k;
        tky[t] = tap < taps ? ky : -(1 << 20);               // a tap beyond the kernel is out of bounds for every pixel
        tkx[t] = kx;
        toff[t] = (ky * d.w_in + kx) * d.ldx;
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[t][e] = 0.f;
    }
    if (ok && p0 + prow < p1) {
        int p = p0 + prow;
        const int howo = d.ho * d.wo;
        int n = p / howo;
        const int rem = p - n * howo;
        int ho = rem / d.wo, wo = rem - ho * d.wo;
        for (; p < p1; p += gm.rows) {
            const V gv = *reinterpret_cast<const V*>(dz + (long)p * d.lddz);
            const int hi0 = ho * d.stride - d.pad, wi0 = wo * d.stride - d.pad;
            const T* const xb = x + ((long)(n * d.h + hi0) * d.w_in + wi0) * d.ldx;      // only dereferenced at in-bounds taps
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const bool in = (unsigned)(hi0 + tky[t]) < (unsigned)d.h && (unsigned)(wi0 + tkx[t]) < (unsigned)d.w_in;
                const T* ad = in ? xb + toff[t] : zero;
                asm volatile("" : "+v"(ad));                  // keep the select (the optimiser would turn it back into a branch around the load)
                const V xv = *reinterpret_cast<const V*>(ad);
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[t][e] = fmaf((float)gv[e], (float)xv[e], acc[t][e]);
            }
            wo += gm.rows;
            while (wo >= d.wo) {
                wo -= d.wo;
                if (++ho == d.ho) {
                    ho = 0;
                    ++n;
                }
            }
        }
    }
    __shared__ float red[256 * VN];
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        const bool live = t0 + t < taps;      // block-uniform
#pragma unroll
        for (int e = 0; e < VN; ++e) red[e * 256 + threadIdx.x] = ok ? acc[t][e] : 0.f;
        __syncthreads();
        if (ok && live)
            for (int it = prow; it < VN; it += gm.rows) {
                float v = 0.f;
                for (int r = 0; r < gm.rows; ++r) v += red[it * 256 + r * gm.cgb + cgl];
                // with a workspace: this workgroup's row [taps][c] of partial sums, added up by dw_partials_kernel (the contended
                // same-address atomics of ~1500 workgroups were 90 % of this kernel's time)
                if (d.ws) d.ws[((long)blockIdx.y * taps + t0 + t) * d.c + g * VN + it] = v;
                else atomicAdd(d.dw + (long)(g * VN + it) * taps + t0 + t, v);
            }
        __syncthreads();
    }
}

// dw[ch][tap] += sum over the pixel chunks' rows in a fixed order (deterministic).  256 threads = 16 columns (i = tap * c + channel: a wave
// load covers four 64-byte runs) x 16 row segments, four load chains per thread, segments added up in LDS (round 6, second form: one thread
// per column with all ~512 rows behind it left ~24 workgroups walking 64 dependent iterations - as long as the taps kernel itself)
__global__ __launch_bounds__(256) void dw_partials_kernel(const float* __restrict__ ws, int rows, int taps, int c, float* dw) {
    __shared__ float red[256];
    const int col = threadIdx.x & 15, seg = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + col, total = taps * c;
    const long pitch = total;
    const int per = (rows + 15) / 16, r0 = seg * per, r1 = min(rows, r0 + per);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < total) {
        int r = r0;
        for (; r + 3 < r1; r += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] += ws[(r + u) * pitch + i];
        }
        for (int u = 0; r < r1; ++r, ++u) v[u] += ws[r * pitch + i];
    }
    red[seg * 16 + col] = (v[0] + v[1]) + (v[2] + v[3]);
    __syncthreads();
    if (seg == 0 && i < total) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q * 16 + col];
        const int tap = i / c, ch = i - tap * c;
        dw[(long)ch * taps + tap] += t;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const yh_dw_bwd_desc d) {
    typedef typename TV<T>::type V;
    constexpr int VN = TV<T>::N;
    const int cg = d.c / VN;
    const long total = (long)d.n * d.h * d.w_in * cg;
    const T* dz = reinterpret_cast<const T*>(d.dz);
    const T* w = reinterpret_cast<const T*>(d.w);
    T* dx = reinterpret_cast<T*>(d.dx);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        long r = i / cg;
        const int wi = (int)(r % d.w_in);
        r /= d.w_in;
        const int hi = (int)(r % d.h);
        const long n = r / d.h;
        float acc[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[e] = 0.f;
        for (int kr = 0; kr < d.k; ++kr) {
            const int th = hi + d.pad - kr;
            if (th < 0 || th % d.stride) continue;
            const int ho = th / d.stride;
            if (ho >= d.ho) continue;
            for (int ks = 0; ks < d.k; ++ks) {
                const int tw = wi + d.pad - ks;
                if (tw < 0 || tw % d.stride) continue;
                const int wo = tw / d.stride;
                if (wo >= d.wo) continue;
                const V gv = *reinterpret_cast<const V*>(dz + ((n * d.ho + ho) * d.wo + wo) * d.lddz + g * VN);
                const V wv = *reinterpret_cast<const V*>(w + (long)(kr * d.k + ks) * d.c + g * VN);
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[e] = fmaf((float)gv[e], (float)wv[e], acc[e]);
            }
        }
        T* dst = dx + ((n * d.h + hi) * d.w_in + wi) * d.lddx + g * VN;
        V o;
        if (d.accumulate) {
            const V old = *reinterpret_cast<const V*>(dst);
#pragma unroll
            for (int e = 0; e < VN; ++e) o[e] = (T)((float)old[e] + acc[e]);
        } else {
#pragma unroll
            for (int e = 0; e < VN; ++e) o[e] = (T)acc[e];
        }
        *reinterpret_cast<V*>(dst) = o;
    }
}

// Round 6: the depthwise data gradient with a fixed channel group per thread and an incremental walk over the INPUT pixels (no division per
// element, kernel size and stride as template parameters, absent taps through the zero page instead of branches); same multiply-add
// order per element as dw_dgrad_kernel: bit-identical.  K: 3 / 5 / 0 = run time, S: 1 / 2 / 0 = run time.
template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dw_dgrad_walk_kernel(const yh_dw_bwd_desc d, const int cgb, const int rows, const int ppb) {
    typedef typename TV<T>::type V;
    constexpr int VN = TV<T>::N;
    const int cgs = d.c / VN;
    const int cgl = threadIdx.x % cgb, prow = threadIdx.x / cgb;
    const int g = blockIdx.x * cgb + cgl;
    const int pixels = d.n * d.h * d.w_in;
    const int p0 = blockIdx.y * ppb, p1 = min(p0 + ppb, pixels);
    int p = p0 + prow;
    if (prow >= rows || g >= cgs || p >= p1) return;
    const int k = K ? K : d.k, stride = S ? S : d.stride;
    const T* const dz = reinterpret_cast<const T*>(d.dz) + g * VN;
    const T* const w = reinterpret_cast<const T*>(d.w) + g * VN;
    T* const dx = reinterpret_cast<T*>(d.dx) + g * VN;
    const T* const zero = reinterpret_cast<const T*>(dw_zero_page);
    const int hw = d.h * d.w_in;
    int n = p / hw;
    const int rem = p - n * hw;
    int hi = rem / d.w_in, wi = rem - hi * d.w_in;
    for (; p < p1; p += rows) {
        float acc[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[e] = 0.f;
        const T* const zb = dz + (long)n * d.ho * d.wo * d.lddz;
        auto tap = [&](int kr, int ks, bool present) {
            const int th = hi + d.pad - kr, tw = wi + d.pad - ks;
            bool in = present && th >= 0 && tw >= 0;
            int ho, wo;
            if (stride == 1) { ho = th; wo = tw; }
            else if (stride == 2) { in = in && !((th | tw) & 1); ho = th >> 1; wo = tw >> 1; }
            else { in = in && th % stride == 0 && tw % stride == 0; ho = th / stride; wo = tw / stride; }
            in = in && ho < d.ho && wo < d.wo;
            const T* ad = in ? zb + ((long)ho * d.wo + wo) * d.lddz : zero;
            asm volatile("" : "+v"(ad));
            const V gv = *reinterpret_cast<const V*>(ad);
            const V wv = *reinterpret_cast<const V*>(w + (long)(present ? kr * k + ks : 0) * d.c);
#pragma unroll
            for (int e = 0; e < VN; ++e) acc[e] = fmaf((float)gv[e], (float)wv[e], acc[e]);
        };
        if constexpr (K != 0 && S == 2) {      // only the taps of this pixel's parity class reach an output: (K + 1) / 2 per axis, not K
            const int ph = (hi + d.pad) & 1, pw = (wi + d.pad) & 1;
#pragma unroll
            for (int a = 0; a < (K + 1) / 2; ++a)
#pragma unroll
                for (int b = 0; b < (K + 1) / 2; ++b) tap(ph + 2 * a, pw + 2 * b, ph + 2 * a < K && pw + 2 * b < K);
        } else if constexpr (K != 0) {
#pragma unroll
            for (int kr = 0; kr < K; ++kr)
#pragma unroll
                for (int ks = 0; ks < K; ++ks) tap(kr, ks, true);
        } else {
            for (int kr = 0; kr < k; ++kr)
                for (int ks = 0; ks < k; ++ks) tap(kr, ks, true);
        }
        T* const dst = dx + (long)p * d.lddx;
        V o;
        if (d.accumulate) {
            const V old = *reinterpret_cast<const V*>(dst);
#pragma unroll
            for (int e = 0; e < VN; ++e) o[e] = (T)((float)old[e] + acc[e]);
        } else {
#pragma unroll
            for (int e = 0; e < VN; ++e) o[e] = (T)acc[e];
        }
        *reinterpret_cast<V*>(dst) = o;
        wi += rows;
        while (wi >= d.w_in) {
            wi -= d.w_in;
            if (++hi == d.h) {
                hi = 0;
                ++n;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ squeeze-excite backward
// A: scratch[n][c] = sum over the image's pixels of dy * x: common.h image_channel_sums_kernel

// B: the two Linear layers backward for one image; scratch[n] turns from d(gate) into d(pooled).
// Round 6: the weight gradients are no longer 2 x c x cr fp32 atomics per image (YOLOv3-Mobilenetv3, c 672: 7.2 M contended atomics per
// layer): with scratch2 each image leaves its factors dA2 | relu(a1) | dA1 and se_bwd_wgrad_kernel sums the outer products over the images in
// order.  (One output per WAVE for the two products whose rows are strided across threads was measured slower, see depthwise.hip se_fc_kernel.)

__global__ __launch_bounds__(1024) void se_bwd_fc_kernel(const yh_se_bwd_desc d) {
    extern __shared__ float sh[];            // a1[cr] | dA1[cr] | dA2[c] | pooled[c] | relu(a1)[cr]
    float* a1 = sh;
    float* dA1 = sh + d.cr;
    float* dA2 = sh + 2 * d.cr;
    float* pl = dA2 + d.c;
    float* h1 = pl + d.c;
    const int n = blockIdx.x;
    const float* pooled = d.pooled + (long)n * d.c;
    float* sc = d.scratch + (long)n * d.c;
    // the forward again, the two row-contiguous products through rows_dot16 (as depthwise.hip se_fc_kernel)
    for (int c = threadIdx.x; c < d.c; c += blockDim.x) pl[c] = pooled[c];
    __syncthreads();
    rows_dot16(d.w1, d.cr, d.c, pl, [&](int j, float v) {
        a1[j] = v;                           // pre-ReLU
        h1[j] = fmaxf(v, 0.f);
    });
    __syncthreads();
    rows_dot16(d.w2, d.c, d.cr, h1, [&](int c, float a2) {
        const float dsig = (a2 > -3.f && a2 < 3.f) ? (1.f / 6.f) : 0.f;   // hsigmoid(x) = relu6(x + 3) / 6
        dA2[c] = sc[c] * dsig;
    });
    __syncthreads();
    for (int j = threadIdx.x; j < d.cr; j += blockDim.x) {
        float v = 0.f;
        for (int c = 0; c < d.c; ++c) v = fmaf(d.w2[(long)c * d.cr + j], dA2[c], v);
        dA1[j] = a1[j] > 0.f ? v : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d.c; c += blockDim.x) {
        float v = 0.f;
        for (int j = 0; j < d.cr; ++j) v = fmaf(d.w1[(long)j * d.c + c], dA1[j], v);
        sc[c] = v;                           // d(pooled)
    }
    if (d.scratch2) {
        float* f = d.scratch2 + (long)n * (d.c + 2 * d.cr);
        for (int c = threadIdx.x; c < d.c; c += blockDim.x) f[c] = dA2[c];
        for (int j = threadIdx.x; j < d.cr; j += blockDim.x) {
            f[d.c + j] = fmaxf(a1[j], 0.f);
            f[d.c + d.cr + j] = dA1[j];
        }
        return;
    }
    // weight gradients: outer products, summed over the images by atomics
    for (int i = threadIdx.x; i < d.c * d.cr; i += blockDim.x) {
        const int c = i / d.cr, j = i - c * d.cr;
        atomicAdd(d.dw2 + i, dA2[c] * fmaxf(a1[j], 0.f));            // dw2[c][j]
        const int j1 = i / d.c, c1 = i - j1 * d.c;
        atomicAdd(d.dw1 + i, dA1[j1] * pooled[c1]);                  // dw1[j][c]
    }
}

// B2: dw2[c][j] += sum_n dA2[n][c] relu(a1)[n][j], dw1[j][c] += sum_n dA1[n][j] pooled[n][c] - one thread per weight, images in order
__global__ __launch_bounds__(256) void se_bwd_wgrad_kernel(const yh_se_bwd_desc d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.c * d.cr) return;
    const int c = i / d.cr, j = i - c * d.cr;
    const int j1 = i / d.c, c1 = i - j1 * d.c;
    const int fp = d.c + 2 * d.cr;
    float s2 = 0.f, s1 = 0.f;
    for (int n = 0; n < d.n; ++n) {
        const float* f = d.scratch2 + (long)n * fp;
        s2 = fmaf(f[c], f[d.c + j], s2);
        s1 = fmaf(f[d.c + d.cr + j1], d.pooled[(long)n * d.c + c1], s1);
    }
    d.dw2[i] += s2;
    d.dw1[i] += s1;
}

// C: dx (+)= dy * gate + d(pooled) / HW.  Grid (channel-group blocks, pixel chunks, n): a thread keeps one channel group of one image - its
// gate and d(pooled) values in registers - and walks the chunk's pixels (round 6; before: two 64-bit divisions and 16 scalar loads per element)
template <typename T>
__global__ __launch_bounds__(256) void se_bwd_apply_kernel(const yh_se_bwd_desc d, const int cgb, const int rows, const int ppb) {
    typedef typename TV<T>::type V;
    constexpr int VN = TV<T>::N;
    const int cgs = d.c / VN, hw = d.h * d.w_in;
    const int cgl = threadIdx.x % cgb, prow = threadIdx.x / cgb;
    const int g = blockIdx.x * cgb + cgl, n = blockIdx.z;
    const int p0 = blockIdx.y * ppb, p1 = min(p0 + ppb, hw);
    if (prow >= rows || g >= cgs) return;
    const float inv = 1.f / (float)hw;
    float gt[VN], ad[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) {
        gt[e] = d.gate[(long)n * d.c + g * VN + e];
        ad[e] = d.scratch[(long)n * d.c + g * VN + e] * inv;
    }
    const T* const dy = reinterpret_cast<const T*>(d.dy) + (long)n * hw * d.lddy + g * VN;
    T* const dx = reinterpret_cast<T*>(d.dx) + (long)n * hw * d.lddx + g * VN;
    for (int p = p0 + prow; p < p1; p += rows) {
        const V gv = *reinterpret_cast<const V*>(dy + (long)p * d.lddy);
        T* const dst = dx + (long)p * d.lddx;
        V o;
        if (d.accumulate) {
            const V old = *reinterpret_cast<const V*>(dst);
#pragma unroll
            for (int e = 0; e < VN; ++e) o[e] = (T)(fmaf((float)gv[e], gt[e], ad[e]) + (float)old[e]);
        } else {
#pragma unroll
            for (int e = 0; e < VN; ++e) o[e] = (T)fmaf((float)gv[e], gt[e], ad[e]);
        }
        *reinterpret_cast<V*>(dst) = o;
    }
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

static int check_dw(const yh_dw_bwd_desc* d) {
    if (!d || !d->dz || d->n <= 0 || d->c <= 0 || d->k <= 0 || d->stride <= 0 || d->pad < 0) return YH_EINVAL;
    if (d->dtype != YH_F16 && d->dtype != YH_F32) return YH_EINVAL;
    const int v = d->dtype == YH_F16 ? 8 : 4;
    if (d->c % v || d->lddz % v || !aligned16(d->dz)) return YH_EALIGN;
    return YH_OK;
}

static BnGeom dw_wgrad_geom(const yh_dw_bwd_desc* d, dim3* grid) {
    yh_bn_desc geo = {};
    geo.c = d->c;
    geo.pixels = (long)d->n * d->ho * d->wo;
    geo.dtype = d->dtype;
    const BnGeom gm = bn_geom(&geo, d->dtype == YH_F16 ? 8 : 4, grid, 512);
    grid->z = (d->k * d->k + 8) / 9;                    // tap groups of 9
    return gm;
}

extern "C" int64_t yh_dw_wgrad_workspace(const yh_dw_bwd_desc* d) {
    if (check_dw(d)) return 0;
    dim3 grid;
    dw_wgrad_geom(d, &grid);
    return (int64_t)grid.y * d->k * d->k * d->c;
}

extern "C" int yh_dw_wgrad(const yh_dw_bwd_desc* d0, void* stream) {
    int rc = check_dw(d0);
    if (rc) return rc;
    if (!d0->x || !d0->dw) return YH_EINVAL;
    const int v = d0->dtype == YH_F16 ? 8 : 4;
    if (d0->ldx % v || !aligned16(d0->x)) return YH_EALIGN;
    dim3 grid;
    const BnGeom gm = dw_wgrad_geom(d0, &grid);
    yh_dw_bwd_desc dd = *d0;
    if (dd.ws && dd.ws_floats < (int64_t)grid.y * dd.k * dd.k * dd.c) dd.ws = nullptr;       // too small: the atomics
    const yh_dw_bwd_desc* const d = &dd;
    if ((long)d->n * d->ho * d->wo >= (1L << 31) - 65536 || (long)d->n * d->h * d->w_in * d->ldx >= (1L << 40)) return YH_EINVAL;
#define YH_DWW(K)                                                                                                          \
    do {                                                                                                                    \
        if (d->dtype == YH_F16) hipLaunchKernelGGL((dw_wgrad_taps_kernel<f16, K>), grid, dim3(256), 0, (hipStream_t)stream, *d, gm);   \
        else hipLaunchKernelGGL((dw_wgrad_taps_kernel<float, K>), grid, dim3(256), 0, (hipStream_t)stream, *d, gm);          \
    } while (0)
    if (d->k == 3) YH_DWW(3);
    else if (d->k == 5) YH_DWW(5);
    else YH_DWW(0);
#undef YH_DWW
    if (d->ws) {
        const int total = d->k * d->k * d->c;
        hipLaunchKernelGGL(dw_partials_kernel, dim3((total + 15) / 16), dim3(256), 0, (hipStream_t)stream, d->ws, (int)grid.y, d->k * d->k,
                           d->c, d->dw);
    }
    return check_launch();
}

extern "C" int yh_dw_dgrad(const yh_dw_bwd_desc* d, void* stream) {
    int rc = check_dw(d);
    if (rc) return rc;
    if (!d->w || !d->dx) return YH_EINVAL;
    const int v = d->dtype == YH_F16 ? 8 : 4;
    if (d->lddx % v || !aligned16(d->dx) || !aligned16(d->w)) return YH_EALIGN;
    const long total = (long)d->n * d->h * d->w_in * (d->c / v);
    const long pixels = (long)d->n * d->h * d->w_in;
    static const bool old_form = getenv("YH_DW_WALK") && atoi(getenv("YH_DW_WALK")) == 0;      // A/B knob: 0 = the round-1 kernels
    if (!old_form && pixels < (1L << 31) - 65536 && (long)d->n * d->ho * d->wo * d->lddz < (1L << 40)) {
        const int cgs = d->c / v, cgb = cgs < 256 ? cgs : 256, rows = 256 / cgb, gx = (cgs + cgb - 1) / cgb;
        long ppb = (pixels * gx + 8191) / 8192;
        if (ppb < 2L * rows) ppb = 2L * rows;
        ppb = (ppb + rows - 1) / rows * rows;
        const dim3 wgrid(gx, (unsigned)((pixels + ppb - 1) / ppb));
#define YH_DWD(T, K, S) hipLaunchKernelGGL((dw_dgrad_walk_kernel<T, K, S>), wgrid, dim3(256), 0, (hipStream_t)stream, *d, cgb, rows, (int)ppb)
#define YH_DWD_T(T)                                                                          \
        if (d->k == 3 && d->stride == 1) YH_DWD(T, 3, 1);                                        \
        else if (d->k == 3 && d->stride == 2) YH_DWD(T, 3, 2);                                   \
        else if (d->k == 5 && d->stride == 1) YH_DWD(T, 5, 1);                                   \
        else if (d->k == 5 && d->stride == 2) YH_DWD(T, 5, 2);                                   \
        else YH_DWD(T, 0, 0)
        if (d->dtype == YH_F16) { YH_DWD_T(f16); } else { YH_DWD_T(float); }
#undef YH_DWD_T
#undef YH_DWD
        return check_launch();
    }
    long gsz = (total + 255) / 256;
    const dim3 grid((unsigned)(gsz < 1 ? 1 : (gsz > 16384 ? 16384 : gsz)));
    if (d->dtype == YH_F16) hipLaunchKernelGGL(dw_dgrad_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, *d);
    else hipLaunchKernelGGL(dw_dgrad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, *d);
    return check_launch();
}

extern "C" int yh_se_bwd(const yh_se_bwd_desc* d, void* stream) {
    if (!d || !d->x || !d->dy || !d->dx || !d->w1 || !d->w2 || !d->pooled || !d->gate || !d->dw1 || !d->dw2 || !d->scratch) return YH_EINVAL;
    if (d->n <= 0 || d->c <= 0 || d->cr <= 0 || (d->dtype != YH_F16 && d->dtype != YH_F32)) return YH_EINVAL;
    const int v = d->dtype == YH_F16 ? 8 : 4;
    if (d->c % v || d->ldx % v || d->lddy % v || d->lddx % v || !aligned16(d->x) || !aligned16(d->dy) || !aligned16(d->dx)) return YH_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int cgs = d->c / v, cgb = cgs < 256 ? cgs : 256, hw = d->h * d->w_in;
    if (d->n > 65535) return YH_EINVAL;
    const dim3 rgrid((cgs + 7) / 8, d->n);
    if (d->dtype == YH_F16)
        hipLaunchKernelGGL((image_channel_sums_kernel<f16, f16x8, 8, true>), rgrid, dim3(1024), 0, s, (const f16*)d->x, (long)d->ldx,
                           (const f16*)d->dy, (long)d->lddy, d->c, hw, 1.f, d->scratch, (long)d->c);
    else
        hipLaunchKernelGGL((image_channel_sums_kernel<float, f32x4, 4, true>), rgrid, dim3(1024), 0, s, (const float*)d->x, (long)d->ldx,
                           (const float*)d->dy, (long)d->lddy, d->c, hw, 1.f, d->scratch, (long)d->c);
    yh_se_bwd_desc dd = *d;
    if (dd.scratch2 && dd.scratch2_floats < (int64_t)dd.n * (dd.c + 2 * dd.cr)) dd.scratch2 = nullptr;
    hipLaunchKernelGGL(se_bwd_fc_kernel, dim3(d->n), dim3(1024), (size_t)(3 * d->cr + 2 * d->c) * sizeof(float), s, dd);
    if (dd.scratch2) hipLaunchKernelGGL(se_bwd_wgrad_kernel, dim3((d->c * d->cr + 255) / 256), dim3(256), 0, s, dd);
    const int rows = 256 / cgb, gx = (cgs + cgb - 1) / cgb;
    long ppb = ((long)hw * gx * d->n + 8191) / 8192;      // ~8192 workgroups, at least four pixels per thread
    if (ppb < 4L * rows) ppb = 4L * rows;
    ppb = (ppb + rows - 1) / rows * rows;
    const dim3 agrid(gx, (unsigned)((hw + ppb - 1) / ppb), d->n);
    if (d->dtype == YH_F16) hipLaunchKernelGGL(se_bwd_apply_kernel<f16>, agrid, dim3(256), 0, s, *d, cgb, rows, (int)ppb);
    else hipLaunchKernelGGL(se_bwd_apply_kernel<float>, agrid, dim3(256), 0, s, *d, cgb, rows, (int)ppb);
    return check_launch();
}

extern "C" int yh_maxpool2d_bwd(const yh_pool_bwd_desc* d, void* stream) {
    if (!d || !d->x || !d->dy || !d->dx || d->n <= 0 || d->c <= 0 || d->k <= 0 || d->stride <= 0) return YH_EINVAL;
    if (d->dtype != YH_F16 && d->dtype != YH_F32) return YH_EINVAL;
    const int v = d->dtype == YH_F16 ? 8 : 4;
    if (d->c % v || d->ldx % v || d->lddy % v || d->lddx % 2 || !aligned16(d->x) || !aligned16(d->dy) || (((uintptr_t)d->dx) & 3u)) return YH_EALIGN;
    const long total = (long)d->n * d->ho * d->wo * (d->c / v);
    {   // overlapping windows on a plane that fits in LDS: the gather form (no atomics, deterministic)
        const long hw = (long)d->h * d->w_in, hwo = (long)d->h * d->wo, howo = (long)d->ho * d->wo;
        const size_t shmem = (size_t)(hw + hwo + howo) * 16 + (size_t)(hwo + howo) * v * 2;
        static const bool scatter = getenv("YH_POOL_BWD_SCATTER") && atoi(getenv("YH_POOL_BWD_SCATTER")) != 0;      // A/B: the atomics
        if (!scatter && d->stride < d->k && hw < 32768 && shmem <= 96 * 1024 && (long)d->n * (d->c / v) < (1L << 31) && d->lddx % v == 0 &&
            aligned16(d->dx)) {
            const dim3 pgrid((unsigned)((long)d->n * (d->c / v)));
            hipError_t e;
            if (d->dtype == YH_F16) {
                e = ensure_dynamic_lds(reinterpret_cast<const void*>(maxpool_bwd_plane_kernel<f16>), shmem);
                if (e != hipSuccess) return (int)e;
                hipLaunchKernelGGL(maxpool_bwd_plane_kernel<f16>, pgrid, dim3(256), shmem, (hipStream_t)stream, *d);
            } else {
                e = ensure_dynamic_lds(reinterpret_cast<const void*>(maxpool_bwd_plane_kernel<float>), shmem);
                if (e != hipSuccess) return (int)e;
                hipLaunchKernelGGL(maxpool_bwd_plane_kernel<float>, pgrid, dim3(256), shmem, (hipStream_t)stream, *d);
            }
            return check_launch();
        }
    }
    long gsz = (total + 255) / 256;
    const dim3 grid((unsigned)(gsz < 1 ? 1 : (gsz > 16384 ? 16384 : gsz)));
    if (d->dtype == YH_F16) hipLaunchKernelGGL(maxpool_bwd_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, *d);
    else hipLaunchKernelGGL(maxpool_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, *d);
    return check_launch();
}

extern "C" int64_t yh_bn_reduce_workspace(const yh_bn_desc* d) {
    if (!d || d->c <= 0 || d->pixels <= 0 || (d->dtype != YH_F16 && d->dtype != YH_F32)) return 0;
    if (d->nparts > 0) return (int64_t)d->nparts * 2 * d->c;      // the rows a data-gradient launch left (yh_conv_desc.bwd_z)
    dim3 grid;
    yh_bn_desc tmp = *d;
    tmp.ws = nullptr;
    bn_geom(&tmp, d->dtype == YH_F16 ? 8 : 4, &grid, 1024);
    return (int64_t)grid.y * 2 * d->c;
}

extern "C" int yh_bn_stats(const yh_bn_desc* d, void* stream) {
    int rc = check_bn(d, false, false);
    if (rc) return rc;
    if (!d->sum || !d->sumsq) return YH_EINVAL;
    dim3 grid;
    const BnGeom gm = bn_geom(d, d->dtype == YH_F16 ? 8 : 4, &grid, 1024);
    if (d->dtype == YH_F16) hipLaunchKernelGGL(bn_stats_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, *d, gm);
    else hipLaunchKernelGGL(bn_stats_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, *d, gm);
    if (gm.two_stage) sum_partials(d, grid, stream);
    return check_launch();
}

// rows of [2][c] partials (conv epilogue) -> sum / sumsq: 16 columns x 16 row lanes per workgroup, row groups in grid.y
__global__ __launch_bounds__(256) void bn_conv_partials_kernel(float* part, int nparts, int per_group, int c, float* s0,
                                                               float* s1, int inplace) {
    __shared__ float red[256];
    const int col = blockIdx.x * 16 + (threadIdx.x & 15), lane = threadIdx.x >> 4;
    const int r0 = blockIdx.y * per_group, r1 = min(r0 + per_group, nparts);
    float v = 0.f;
    if (col < 2 * c)
        for (int k = r0 + lane; k < r1; k += 16) v += part[(long)k * 2 * c + col];
    red[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x < 16 && col < 2 * c) {
        float t = 0.f;
        for (int r = 0; r < 16; ++r) t += red[r * 16 + threadIdx.x];
        // inplace (deterministic form): the group's sums replace its first row (this workgroup alone reads and writes these 16 columns
        // of rows r0 .. r1 - 1, and every read is behind the barrier above); bn_finalize_kernel adds the groups in order
        if (inplace) part[(long)r0 * 2 * c + col] = t;
        else atomicAdd((col < c ? s0 : s1) + (col < c ? col : col - c), t);
    }
}

extern "C" int yh_bn_finalize(const yh_bn_desc* d, void* stream) {
    if (!d || !d->sum || !d->sumsq || !d->mean || !d->invstd || d->c <= 0 || d->pixels <= 0) return YH_EINVAL;
    if (d->nparts > 0) {
        if (!d->ws || d->ws_floats < (int64_t)d->nparts * 2 * d->c) return YH_EINVAL;
        int groups = (d->nparts + 255) / 256;               // >= 256 rows per group, <= 64 groups (64 atomics per channel)
        if (groups > 64) groups = 64;
        const int per_group = (d->nparts + groups - 1) / groups;
        groups = (d->nparts + per_group - 1) / per_group;
        const int det = deterministic() ? 1 : 0;
        hipLaunchKernelGGL(bn_conv_partials_kernel, dim3((2 * d->c + 15) / 16, groups), dim3(256), 0, (hipStream_t)stream, d->ws,
                           d->nparts, per_group, d->c, d->sum, d->sumsq, det);
        if (det) {
            hipLaunchKernelGGL(bn_finalize_kernel, dim3((d->c + 255) / 256), dim3(256), 0, (hipStream_t)stream, *d, groups, per_group);
            return check_launch();
        }
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((d->c + 255) / 256), dim3(256), 0, (hipStream_t)stream, *d, 0, 0);
    return check_launch();
}

extern "C" int yh_bn_act_fwd(const yh_bn_desc* d, void* stream) {
    int rc = check_bn(d, false, true);
    if (rc) return rc;
    if (d->ups != 1 && d->ups != 2) return YH_EINVAL;
    if (d->ups == 2 && (long)d->n * d->h * d->w_in != d->pixels) return YH_EINVAL;
    dim3 grid;
    BnGeom gm = bn_geom(d, d->dtype == YH_F16 ? 8 : 4, &grid);
    gm.reverse = bn_reverse() & 1;
    const int u = bn_unroll(0);
#define YH_BN_LAUNCH_A(K, A)                                                                                       \
    if (d->dtype == YH_F16) {                                                                                      \
        if (u >= 4) hipLaunchKernelGGL((K<f16, 4, A>), grid, dim3(256), 0, (hipStream_t)stream, *d, gm);            \
        else if (u >= 2) hipLaunchKernelGGL((K<f16, 2, A>), grid, dim3(256), 0, (hipStream_t)stream, *d, gm);       \
        else hipLaunchKernelGGL((K<f16, 1, A>), grid, dim3(256), 0, (hipStream_t)stream, *d, gm);                   \
    } else {                                                                                                       \
        if (u >= 4) hipLaunchKernelGGL((K<float, 4, A>), grid, dim3(256), 0, (hipStream_t)stream, *d, gm);          \
        else if (u >= 2) hipLaunchKernelGGL((K<float, 2, A>), grid, dim3(256), 0, (hipStream_t)stream, *d, gm);     \
        else hipLaunchKernelGGL((K<float, 1, A>), grid, dim3(256), 0, (hipStream_t)stream, *d, gm);                 \
    }
#define YH_BN_LAUNCH(K)                                                                                            \
    switch (d->act) {                                                                                              \
        case YH_ACT_LEAKY: YH_BN_LAUNCH_A(K, YH_ACT_LEAKY) break;                                                  \
        case YH_ACT_RELU: YH_BN_LAUNCH_A(K, YH_ACT_RELU) break;                                                    \
        case YH_ACT_RELU6: YH_BN_LAUNCH_A(K, YH_ACT_RELU6) break;                                                  \
        case YH_ACT_HSWISH: YH_BN_LAUNCH_A(K, YH_ACT_HSWISH) break;                                                \
        case YH_ACT_MISH: YH_BN_LAUNCH_A(K, YH_ACT_MISH) break;                                                    \
        default: YH_BN_LAUNCH_A(K, YH_ACT_LINEAR) break;                                                           \
    }
    YH_BN_LAUNCH(bn_act_fwd_kernel)
    return check_launch();
}

extern "C" int yh_bn_act_bwd_reduce(const yh_bn_desc* d, void* stream) {
    if (d && d->nparts > 0) {
        // the sums were taken by the launch that completed dy (yh_conv_desc.bwd_z, conv_pw_lds.hip): d->ws holds nparts rows of
        // [sum g | sum g xhat][c]; add them into dbeta / dgamma in a fixed order
        if (!d->sum || !d->sumsq || !d->ws || d->c <= 0 || d->ws_floats < (int64_t)d->nparts * 2 * d->c) return YH_EINVAL;
        hipLaunchKernelGGL(bn_partials_kernel, dim3((2 * d->c + 15) / 16, 1), dim3(1024), 0, (hipStream_t)stream, d->ws, d->nparts,
                           d->nparts, d->c, d->sum, d->sumsq);
        return check_launch();
    }
    int rc = check_bn(d, true, false);
    if (rc) return rc;
    if (!d->sum || !d->sumsq) return YH_EINVAL;
    dim3 grid;
    BnGeom gm = bn_geom(d, d->dtype == YH_F16 ? 8 : 4, &grid, 1024);
    gm.reverse = (bn_reverse() >> 1) & 1;
    const int u = bn_unroll(1);
    YH_BN_LAUNCH(bn_act_bwd_reduce_kernel)
    if (gm.two_stage) sum_partials(d, grid, stream);
    return check_launch();
}

extern "C" int yh_bn_act_bwd_apply(const yh_bn_desc* d, void* stream) {
    int rc = check_bn(d, true, true);
    if (rc) return rc;
    if (d->gamma && (!d->sum || !d->sumsq)) return YH_EINVAL;
    dim3 grid;
    BnGeom gm = bn_geom(d, d->dtype == YH_F16 ? 8 : 4, &grid);
    gm.reverse = (bn_reverse() >> 2) & 1;
    const int u = bn_unroll(2);
    YH_BN_LAUNCH(bn_act_bwd_apply_kernel)
#undef YH_BN_LAUNCH
#undef YH_BN_LAUNCH_A
    return check_launch();
}
