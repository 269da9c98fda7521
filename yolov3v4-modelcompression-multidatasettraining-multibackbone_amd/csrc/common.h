// Shared device helpers for libyolo_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <set>
#include <utility>

#include "../../include/yolo_hip.h"

namespace yh {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Activation of the conv epilogue (fp32 in, fp32 out).  `act` is wave-uniform.
// mish(v) = v * tanh(softplus(v)) = v * n / (n + 2) with n = e^v (e^v + 2): one exp, one divide.
__device__ __forceinline__ float activate(float v, int act, float slope) {
    switch (act) {
        case YH_ACT_LEAKY: return v > 0.f ? v : v * slope;
        case YH_ACT_RELU: return fmaxf(v, 0.f);
        case YH_ACT_RELU6: return fminf(fmaxf(v, 0.f), 6.f);
        case YH_ACT_HSWISH: return v * (fminf(fmaxf(v + 3.f, 0.f), 6.f) / 6.f);
        case YH_ACT_MISH: {
            float e = expf(fminf(v, 20.f));
            float n = e * (e + 2.f);
            return v > 20.f ? v : v * (n / (n + 2.f));
        }
        default: return v;
    }
}

// e^x to ~1 ulp on the hardware exp2: x log2(e) in two pieces (product and its rounding error + the low word of log2 e), the second
// folded back in as a first-order correction.  ocml's expf costs ~25 VALU slots per value, this 7; v_exp_f32 saturates to 0 / inf
// at the ends of the range like expf.
__device__ __forceinline__ float exp_fast(float x) {
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.92596299e-8f;
    const float t = x * L2E_HI;
    const float r = fmaf(x, L2E_LO, fmaf(x, L2E_HI, -t));
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, r * 0.693147180559945f, e);
}
// 1 / d for d >= 1 (finite or +inf): the hardware reciprocal (1 ulp) refined by one Newton step
__device__ __forceinline__ float rcp_fast(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return d < 3.0e38f ? fmaf(fmaf(-d, r, 1.f), r, r) : r;
}

// Mish for the int8 (PTQ) epilogues, whose result is rounded onto the activation grid right away.  mish_fast costs ~20 VALU slots
// where activate()'s form (ocml expf + IEEE divide) costs ~45, and agrees with it to < 1e-6 relative for every float (exhaustive
// device self-test yh_qmish_selftest, tests/test_gpu_kernels.py).  mish_for_grid returns a value whose ROUNDED grid index is that
// of activate()'s for every input: where the fast value, scaled by 1 / s_a, lies within 4e-6 relative of a rounding tie (k + 0.5)
// - a few values in 10^5 - the exact form decides (mish_f64 below: activate()'s float form unless built with -DYH_QMISH_TIE_F64).  The int8 heads stay bit-equal to the reference on exact frames
// (tests/test_ptq_large.py), which a plain substitution of the fast form did not (measured: a handful of values per tensor flip).
// Round 6: nine VALU slots instead of ~21.  e^v straight off the hardware exp2 (argument v log2(e) rounded once: relative error
// <= |v| 6e-8 + 1 ulp), the hardware reciprocal without its Newton step (1 ulp), no selects: above 20 the clamped form gives
// n / (n + 2) = 1 - 2^-57, i.e. v itself to within an ulp; far below zero e^v underflows to 0 and the result is -0 like the exact
// form's.  The errors enter through n / (n + 2), whose sensitivity to e^v is at most 2 and vanishes where Mish is the identity:
// measured over every float with |v| <= 64 (yh_qmish_selftest): wherever the result is at least a quarter of a grid step the form stays
// within 1e-6 relative of activate()'s (far below zero, where Mish is ~1e-20, it reaches |v| 6e-8), inside the 4e-6 band in which
// mish_for_grid consults the exact form - and the grid values are unchanged for all 2.2e9 inputs per scale.
__device__ __forceinline__ float mish_fast(float v) {
    // No operation of this body may be contracted into an fma with a neighbour - in particular not the last product with a residual
    // add in the caller's epilogue (it was, in one of two kernels serving the same layer: 7 of 55 680 fp16 values a ulp apart).  HIP's
    // __fmul_rn is a plain product, so the pragma is what pins it (hipcc's default is -ffp-contract=fast-honor-pragmas).
#pragma clang fp contract(off)
    const float u = __builtin_amdgcn_exp2f(fminf(v, 20.f) * 1.44269504088896340736f);
    const float n = u * (u + 2.f);
    const float q = n * __builtin_amdgcn_rcpf(n + 2.f);
    return v * q;
}

// derivative of the activation w.r.t. its pre-activation u (training backward: train.hip, conv_pw_lds.hip)
__device__ __forceinline__ float act_grad(float u, int act, float slope) {
    switch (act) {
        case YH_ACT_LEAKY: return u > 0.f ? 1.f : slope;
        case YH_ACT_RELU: return u > 0.f ? 1.f : 0.f;
        case YH_ACT_RELU6: return (u > 0.f && u < 6.f) ? 1.f : 0.f;
        case YH_ACT_HSWISH: return u <= -3.f ? 0.f : (u >= 3.f ? 1.f : (2.f * u + 3.f) / 6.f);
        case YH_ACT_MISH: {
            // one exp and two hardware reciprocals (1 ulp) per element: the reduce and apply kernels both evaluate this,
            // and with two exps and two IEEE divides they were ALU bound on the mish networks (YOLOv4)
            const float e = expf(fminf(u, 20.f));
            const float n = e * (e + 2.f);
            const float t = u > 20.f ? 1.f : n * __builtin_amdgcn_rcpf(n + 2.f);          // tanh(softplus(u))
            const float sg = u > 20.f ? 1.f : e * __builtin_amdgcn_rcpf(e + 1.f);         // sigmoid(u)
            return t + u * sg * (1.f - t * t);
        }
        default: return 1.f;
    }
}
// d mish / du on the hardware exp2 and ONE reciprocal (round 6; the fp16 training kernels).  With e = e^u, n = e (e + 2), t = tanh(softplus(u)) =
// n / (n + 2): 1 - t^2 = 4 (n + 1) / (n + 2)^2 and n + 1 = (e + 1)^2, so t + u sigmoid(u) (1 - t^2) = [n (n + 2) + 4 u e (e + 1)] / (n + 2)^2.
// ~16 VALU slots against the ~45 of act_grad's ocml expf + two reciprocals (YOLOv4's BatchNorm backward passes were VALU-bound on it, not
// byte-bound); within ~1e-6 of it relative to the derivative's scale.  u is clamped to [-60, 20]: 1 above, u e^u (< 1e-24) below, no overflow
// (n (n + 2) <= 5.5e34).  No contraction across the body, so that every kernel evaluates the same bits (see mish_fast).
__device__ __forceinline__ float mish_grad_fast(float u) {
#pragma clang fp contract(off)
    const float c = __builtin_amdgcn_fmed3f(u, -60.f, 20.f);
    const float e = __builtin_amdgcn_exp2f(c * 1.44269504088896340736f);
    const float n = e * (e + 2.f);
    const float d = n + 2.f;
    const float r = __builtin_amdgcn_rcpf(d);
    const float num = fmaf((4.f * c) * e, e + 1.f, n * d);
    return (num * r) * r;
}
// the training kernels' choice by storage type: fp16 tensors take the fast forms (their results are rounded to 11 bits anyway), fp32 keeps
// the exact ones - it is the side of the fp32 comparisons against the reference
template <typename T> __device__ __forceinline__ float act_grad_t(float u, int act, float slope) {
    if (sizeof(T) == 2 && act == YH_ACT_MISH) return mish_grad_fast(u);
    return act_grad(u, act, slope);
}
template <typename T> __device__ __forceinline__ float activate_train_t(float v, int act, float slope) {
    if (sizeof(T) == 2 && act == YH_ACT_MISH) return mish_fast(v);
    return activate(v, act, slope);
}

// Mish to one rounding: v tanh(softplus(v)) = v n / (n + 2), n = e^v (e^v + 2), evaluated in double and rounded once to float - what
// the reference's fp32 `x * torch.tanh(F.softplus(x))` (utils/layers.py:148; softplus passes x through above 20) approximates to ~3
// ulp.  Round 5 tried it as the form that decides a value next to a rounding tie of the activation grid (0.6 - 1.4 %
// of YOLOv4-640's head values sit one grid step from the CPU modules on the calibrated state) - and measured that it is NOT the lever:
// the share of differing head values stayed 0.6 % / 1.4 % / 1.2 % to the digit (the flips come from torch's own libm error crossing a
// tie, which no GPU formula reproduces), while the double-precision branch cost every int8 Mish kernel 17 - 40 registers (pointwise
// <i8, i8, 2, 1>: 112 -> 152, conv1x1_lds <i8, 4, 1>: 72 -> 96) and YOLOv4-640 int8 3 % of its throughput.  The tie band therefore keeps
// activate()'s float form (rounds 1 - 4); -DYH_QMISH_TIE_F64 builds the one-rounding form for the record.
__device__ __forceinline__ float mish_f64(float v) {
#ifndef YH_QMISH_TIE_F64
    return activate(v, YH_ACT_MISH, 0.f);
#else
    if (v > 20.f) return v;
    const double e = exp((double)v);
    const double n = e * (e + 2.0);
    return (float)((double)v * (n / (n + 2.0)));
#endif
}
__device__ __forceinline__ float mish_for_grid(float v, float inv_s) {
    float y = mish_fast(v);
    const float t = fabsf(y * inv_s);
    const float f = t - floorf(t);
    if (fabsf(f - 0.5f) <= 4e-6f * t) y = mish_f64(v);
    return y;
}
// the same for the N values of a fragment with ONE branch: a per-value branch costs the fast form its advantage (measured: no gain,
// every value a basic block of its own); if any of the N lies next to a tie, all N take the exact form (same grid values either way)
template <int N> __device__ __forceinline__ void mish_for_grid_n(float (&v)[N], float inv_s) {
    float y[N];
    bool near = false;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        y[e] = mish_fast(v[e]);
        const float t = fabsf(y[e] * inv_s);
        near = near || fabsf((t - floorf(t)) - 0.5f) <= 4e-6f * t;
    }
    if (near) {
#pragma unroll
        for (int e = 0; e < N; ++e) y[e] = mish_f64(v[e]);
    }
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] = y[e];
}

// The whole int8 Mish epilogue of N values: q[e] = round_half_away(mish(v[e]) / s_a) clamped to int8, as a float.  One scaled value
// t = y / s_a serves the tie test (v_fract) and the rounding; the exact form replaces all N when any of them lies next to a tie
// (same grid values either way, mish_for_grid_n); the clamp is one v_med3.
template <int N> __device__ __forceinline__ void mish_quantize_n(const float (&v)[N], float inv_s, float (&q)[N]) {
    float t[N];
    float m = 1.f;      // min over the N values of |fract |t| - 0.5| - 4e-6 |t|: <= 0 when any of them lies next to a tie.  (As a chain of
                        // v_fma / v_min3 it is ~3 instructions per value; `near = near || ...` was compiled into a bit mask: compare,
                        // select, shift, or - 6.5 per value on kernels that are VALU-bound on this epilogue.)
#pragma unroll
    for (int e = 0; e < N; ++e) {
        t[e] = mish_fast(v[e]) * inv_s;
        const float a = fabsf(t[e]);
        m = fminf(m, fmaf(-4e-6f, a, fabsf(__builtin_amdgcn_fractf(a) - 0.5f)));
    }
    if (m <= 0.f) {
#pragma unroll
        for (int e = 0; e < N; ++e) t[e] = mish_f64(v[e]) * inv_s;
    }
#pragma unroll
    for (int e = 0; e < N; ++e) q[e] = __builtin_amdgcn_fmed3f(copysignf(floorf(fabsf(t[e]) + 0.5f), t[e]), -128.f, 127.f);
}

// out(r) = sum_k W[r * len + k] * v[k] for r < rows - the squeeze-excite matrix-vector products whose rows are contiguous.  Sixteen
// lanes per row (a wave load touches four 64-byte runs, not 64 lines - thread-per-row is bound by the texture addresser at ~64 cycles per
// load), eight rows of a wave in flight at once (one row per wave at a time was measured 2.7 x slower than thread-per-row: a memory latency
// per row); v in LDS.  Fixed order: deterministic.
template <typename F>
__device__ __forceinline__ void rows_dot16(const float* __restrict__ W, const int rows, const int len, const float* v, F&& emit) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int q = lane >> 4, l = lane & 15;
    for (int r0 = wave * 8; r0 < rows; r0 += nw * 8) {
        const int ra = r0 + q, rb = r0 + 4 + q;
        const float* const wa = W + (long)(ra < rows ? ra : rows - 1) * len;
        const float* const wb = W + (long)(rb < rows ? rb : rows - 1) * len;
        float sa = 0.f, sb = 0.f;
        for (int k = l; k < len; k += 16) {
            const float x = v[k];
            sa = fmaf(wa[k], x, sa);
            sb = fmaf(wb[k], x, sb);
        }
#pragma unroll
        for (int m = 8; m; m >>= 1) {
            sa += __shfl_xor(sa, m, 16);
            sb += __shfl_xor(sb, m, 16);
        }
        if (l == 0) {
            if (ra < rows) emit(ra, sa);
            if (rb < rows) emit(rb, sb);
        }
    }
}

// Per image and channel: the sum over the image's pixels of x (MUL false: squeeze-excite pooling) or of dy * x (MUL true: its backward),
// divided by `div`.  Grid (ceil(channel groups / 8), n), 1024 threads = 8 channel groups (16 bytes each: a wave load covers 128-byte runs of
// 8 pixels) x 128 pixel rows, two load chains per thread, rows added up in LDS in a fixed order.  Round 6: replaces one 256-thread workgroup
// per image (64 workgroups on 256 CUs, ~90 dependent iterations) and a 256-thread tree per 8 channels of an image.
template <typename T, typename V, int VN, bool MUL>
__global__ __launch_bounds__(1024) void image_channel_sums_kernel(const T* __restrict__ x, const long ldx, const T* __restrict__ dy, const long lddy,
                                                                   const int c, const int hw, const float div, float* __restrict__ out,
                                                                   const long ldo) {
    __shared__ float red[1024 * VN];
    const int cgs = c / VN;
    const int cgl = threadIdx.x & 7, prow = threadIdx.x >> 3;
    const int g = blockIdx.x * 8 + cgl, n = blockIdx.y;
    float a0[VN], a1[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) a0[e] = a1[e] = 0.f;
    if (g < cgs) {
        const T* const xp = x + (long)n * hw * ldx + g * VN;
        const T* const gp = MUL ? dy + (long)n * hw * lddy + g * VN : nullptr;
        int p = prow;
        for (; p + 128 < hw; p += 256) {
            const V x0 = *reinterpret_cast<const V*>(xp + (long)p * ldx), x1 = *reinterpret_cast<const V*>(xp + (long)(p + 128) * ldx);
            if constexpr (MUL) {
                const V g0 = *reinterpret_cast<const V*>(gp + (long)p * lddy), g1 = *reinterpret_cast<const V*>(gp + (long)(p + 128) * lddy);
#pragma unroll
                for (int e = 0; e < VN; ++e) {
                    a0[e] = fmaf((float)x0[e], (float)g0[e], a0[e]);
                    a1[e] = fmaf((float)x1[e], (float)g1[e], a1[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < VN; ++e) {
                    a0[e] += (float)x0[e];
                    a1[e] += (float)x1[e];
                }
            }
        }
        if (p < hw) {
            const V x0 = *reinterpret_cast<const V*>(xp + (long)p * ldx);
            if constexpr (MUL) {
                const V g0 = *reinterpret_cast<const V*>(gp + (long)p * lddy);
#pragma unroll
                for (int e = 0; e < VN; ++e) a0[e] = fmaf((float)x0[e], (float)g0[e], a0[e]);
            } else {
#pragma unroll
                for (int e = 0; e < VN; ++e) a0[e] += (float)x0[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VN; ++e) red[e * 1024 + threadIdx.x] = a0[e] + a1[e];       // [e][pixel row][channel group]
    __syncthreads();
    const int o = threadIdx.x >> 4, part = threadIdx.x & 15;      // output (e, channel group) x 16 parts of 8 pixel rows
    if (o < 8 * VN) {      // (whole 16-lane groups: the shuffle below stays inside them)
        const int ocg = o & 7, oe = o >> 3;
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) v += red[oe * 1024 + (part * 8 + r) * 8 + ocg];
#pragma unroll
        for (int m = 8; m; m >>= 1) v += __shfl_xor(v, m, 16);
        const int og = blockIdx.x * 8 + ocg;
        if (part == 0 && og < cgs) out[(long)n * ldo + og * VN + oe] = v / div;
    }
}

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }

inline int check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? YH_OK : (int)e;
}

// Dynamic LDS above 64 KB needs the function attribute raised - per DEVICE (a single process may drive several GPUs:
// DataParallel, `--device 0,1`), so the "already raised" state is keyed by (kernel, current device).
inline hipError_t ensure_dynamic_lds(const void* kern, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> raised;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (raised.count({kern, dev})) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) raised.insert({kern, dev});
    return e;
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// Weight-gradient reduce launches off the critical path (round 6, yh_plan_set_async_reduce).  A weight gradient is a split-K GEMM: the main
// kernel leaves per-split partial tiles in the workspace and small HBM-bound launches add them into dW - which nothing reads before the
// optimizer.  Inside a plan whose weight gradients own their workspace, those launches go to the plan's reduce stream: they run in the idle
// CUs of the next kernels' last rounds instead of in front of them.  The executor sets the context around a weight-gradient op; the next
// weight gradient's main kernel waits for the previous reduce (same workspace), every plan range joins before it returns.
struct AsyncReduce {
    hipStream_t side = nullptr;
    hipEvent_t main_done = nullptr, red_done = nullptr;
    bool pending = false;      // a reduce is in flight on `side`
};
extern thread_local AsyncReduce* g_async_reduce;      // plan.hip; nullptr outside an op that may use it
// before the main kernel overwrites the workspace: the previous reduce must have read it
inline void reduce_guard_workspace(hipStream_t st) {
    AsyncReduce* c = g_async_reduce;
    if (c && c->pending) {
        (void)hipStreamWaitEvent(st, c->red_done, 0);
        c->pending = false;
    }
}
// the stream the reduce launches of this weight gradient go to (after the main kernel on `st`)
inline hipStream_t reduce_begin(hipStream_t st) {
    AsyncReduce* c = g_async_reduce;
    if (!c) return st;
    if (hipEventRecord(c->main_done, st) != hipSuccess || hipStreamWaitEvent(c->side, c->main_done, 0) != hipSuccess) return st;
    return c->side;
}
inline void reduce_end(hipStream_t st, hipStream_t rs) {
    AsyncReduce* c = g_async_reduce;
    if (c && rs != st && hipEventRecord(c->red_done, rs) == hipSuccess) c->pending = true;
}

// Run-to-run deterministic reductions (round 6; the reference's CPU training step is bit-reproducible).  Every sum over workgroups of
// the training path - BatchNorm statistics, BatchNorm backward sums, weight-gradient pixel splits, the first block's backward - is
// taken in a FIXED order: partial results in a workspace, summed by one owner per output element, never by fp32 atomics whose
// arrival order varies.  YH_DETERMINISTIC=0 brings back the round-5 forms (split groups meeting in atomics) for A/B runs.
// What stays order-dependent is listed in DESIGN.md section 8 (max-pool / depthwise / SE backward scatter, >= 3 labels in one cell).
inline bool deterministic() {
    static const int v = [] { const char* e = getenv("YH_DETERMINISTIC"); return e ? (atoi(e) != 0 ? 1 : 0) : 1; }();
    return v != 0;
}

// conv_stem_mfma.hip: the 3 x 3 x 3-plane first layer on the matrix cores (YH_EUNSUPPORTED: use the kernel in elementwise.hip)
int launch_stem_mfma(const yh_stem_desc& d, hipStream_t stream);
long stem_mfma_stats_rows(const yh_stem_desc& d);

// conv_wgrad_roll.hip: the rolling-halo 3x3 weight gradient (YH_EUNSUPPORTED: layer does not qualify / workspace too small)
#ifndef YH_WGRAD_HALO_DEFAULT
#define YH_WGRAD_HALO_DEFAULT 2
#endif
int launch_wgrad_roll(const yh_wgrad_desc* d, hipStream_t stream);
int64_t wgrad_roll_workspace(const yh_wgrad_desc* d);

}  // namespace yh
