// Pieces shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_igemm_k64.hip).
#pragma once
#include <type_traits>

#include "common.h"

namespace yh {

struct ConvArgs {
    const void* x;
    const void* w;
    const float* bias;
    const void* res;
    void* y;
    int N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad;
    int ldx, ldr, ldy;
    int cin_k;       // padded per-tap K (multiple of BK)
    int ktot;        // R*S*cin_k
    long P;          // N*Ho*Wo
    int m_tiles, p_tiles;
    int m_pad;
    float acc_scale, out_scale, inv_out_scale;  // int8 path: s_w*s_x, s_a, 1/s_a (all powers of two)
    int act;
    float slope;
    int ups;
    float* stats_part;   // training forward: per (pixel tile, wave column) partial sums of y and y^2 per channel, or NULL
    int y_h, y_w, y_off_h, y_off_w;  // ups == 3: output pixel (n, ho, wo) is stored at (n, 2 ho + y_off_h, 2 wo + y_off_w) of a y_h x y_w tensor
    float q_rx, q_ra, q_scale_x, q_scale_a, q_inv_scale_sum;   // int8 + res: fused quantised shortcut (yh_qadd arithmetic)
    int no_lds_store;    // conv_pointwise.hip A/B switch: direct 4-channel stores instead of row stores through LDS
    int hpp_stagger;     // conv_halo_pp.hip A/B switch (YH_HPP_STAGGER): cycles over which the first round's workgroups are delayed; 0 = off
    // training backward (yh_conv_desc.bwd_z; conv_pw_lds.hip modes 3 / 4): the block whose gradient this launch completes
    const void* bz;
    const float *bgamma, *bbeta, *bmean, *binvstd;
    int ldbz, bact;
    float bslope;
};

template <int CTRL> __device__ __forceinline__ float dpp_shr_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, t);
}
// inclusive scan step pattern over a 16-lane DPP row; lane 15 of the row holds the sum of all 16 lanes
__device__ __forceinline__ float row16_sum(float v) {
    v = dpp_shr_add<0x111>(v);   // row_shr:1
    v = dpp_shr_add<0x112>(v);   // row_shr:2
    v = dpp_shr_add<0x114>(v);   // row_shr:4
    v = dpp_shr_add<0x118>(v);   // row_shr:8
    return v;
}

// compile-time unrolled loop: every array index below is a constant, so staging registers never
// fall back to scratch (runtime-indexed private arrays do on this compiler)
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // one 16-byte LDS cell / global vector

template <typename T> struct Prec;
template <> struct Prec<f16> {
    static constexpr int VEC = 8;   // elements per 16-byte unit
};
template <> struct Prec<float> {
    static constexpr int VEC = 4;
};
template <> struct Prec<int8_t> {
    static constexpr int VEC = 16;  // int8 PTQ path: 64 channels per K step on v_mfma_i32_16x16x64_i8
};

typedef int i32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct AccOf { typedef f32x4 type; };
template <> struct AccOf<int8_t> { typedef i32x4 type; };

// PTQ rounding (utils/quantized/quantized_ptq_cos.py:14-20): half away from zero, then clamp to int8
__device__ __forceinline__ float round_clamp_i8(float t) {
    const float r = copysignf(floorf(fabsf(t) + 0.5f), t);
    return __builtin_amdgcn_fmed3f(r, -128.f, 127.f);
}

template <typename OutT> __device__ __forceinline__ void store4(OutT* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<f16>(f16* p, float a, float b, float c, float d) {
    f16x4 v = {(f16)a, (f16)b, (f16)c, (f16)d};
    *reinterpret_cast<f16x4*>(p) = v;
}
template <> __device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
    f32x4 v = {a, b, c, d};
    *reinterpret_cast<f32x4*>(p) = v;
}

template <> __device__ __forceinline__ void store4<int8_t>(int8_t* p, float a, float b, float c, float d) {
    const unsigned v = ((unsigned)(int)a & 0xffu) | (((unsigned)(int)b & 0xffu) << 8) | (((unsigned)(int)c & 0xffu) << 16) |
                       (((unsigned)(int)d & 0xffu) << 24);
    *reinterpret_cast<unsigned*>(p) = v;
}

template <typename T> __device__ __forceinline__ void load4(const T* p, float (&o)[4]);
template <> __device__ __forceinline__ void load4<f16>(const f16* p, float (&o)[4]) {
    f16x4 v = *reinterpret_cast<const f16x4*>(p);
    o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
}
template <> __device__ __forceinline__ void load4<int8_t>(const int8_t* p, float (&o)[4]) {
    const unsigned v = *reinterpret_cast<const unsigned*>(p);
    o[0] = (float)(int8_t)(v & 0xff); o[1] = (float)(int8_t)((v >> 8) & 0xff);
    o[2] = (float)(int8_t)((v >> 16) & 0xff); o[3] = (float)(int8_t)(v >> 24);
}
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&o)[4]) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}

static __device__ __attribute__((aligned(16))) unsigned int g_zero_page[4];   // one copy per translation unit (no -fgpu-rdc)

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    // counted wait on this wave's vector-memory queue (LDS-DMA included); literal operand, compiler barrier
#define YH_VMCNT_CASE(K) else if constexpr (N == K) asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory")
    if constexpr (N < 0) {}
    YH_VMCNT_CASE(0); YH_VMCNT_CASE(1); YH_VMCNT_CASE(2); YH_VMCNT_CASE(3); YH_VMCNT_CASE(4); YH_VMCNT_CASE(5);
    YH_VMCNT_CASE(6); YH_VMCNT_CASE(7); YH_VMCNT_CASE(8); YH_VMCNT_CASE(9); YH_VMCNT_CASE(10); YH_VMCNT_CASE(11);
    YH_VMCNT_CASE(12); YH_VMCNT_CASE(13); YH_VMCNT_CASE(14); YH_VMCNT_CASE(15); YH_VMCNT_CASE(16); YH_VMCNT_CASE(18);
    YH_VMCNT_CASE(20); YH_VMCNT_CASE(24);
    else static_assert(N < 0, "add the literal");
#undef YH_VMCNT_CASE
}

__device__ __forceinline__ int swz_f(int g) { return (0x78 >> (2 * (g & 3))) & 3; }  // {0,2,3,1}

// Epilogue shared by the LDS-DMA kernels: lane (mq = 4 * (lane >> 4), pc = lane & 15) holds channels m .. m+3 of pixel p for each
// 16 x 16 fragment (i, j) of its wave tile: bias + activation (+ residual) -> one 8-byte (f16) / 16-byte (f32) / 4-byte (i8)
// NHWC store; optional 2x nearest store, stride-2 data-gradient phase scatter (ups 3 / 4), int8 requantisation, and the
// per-tile BatchNorm partial sums of the training forward.
// Quantised shortcut on a conv output grid value q and a routed int8 value r (csrc/quant.hip qadd_kernel, operation for
// operation; __fmul_rn / __fadd_rn keep the products and the sum from being contracted into an fma)
__device__ __forceinline__ float qadd_value(float q, float r, const ConvArgs& a) {
    const float xq = __fmul_rn(copysignf(floorf(fabsf(__fmul_rn(q, a.q_rx)) + 0.5f), __fmul_rn(q, a.q_rx)), a.q_scale_x);
    const float aq = __fmul_rn(copysignf(floorf(fabsf(__fmul_rn(r, a.q_ra)) + 0.5f), __fmul_rn(r, a.q_ra)), a.q_scale_a);
    return round_clamp_i8(__fmul_rn(__fadd_rn(xq, aq), a.q_inv_scale_sum));
}

// Round 5: the same shortcut when every factor is a power of two - which is every calibrated COS-PTQ graph: float ranges are 2^step
// (quantized_ptq_cos.py:84, :109, :292), so rx, ra are 2^k, and the fused `_min` form has k >= 0.  Then q rx and r ra are integers (the
// two roundings are the identity), their scaled sum spans <= 24 bits and the final factor is a power of two: every intermediate of
// qadd_value() is EXACT, and so is q A + r B with A = rx scale_x / scale_sum, B = ra scale_a / scale_sum - the same real number rounded
// once by round_clamp_i8: bit-identical by construction, 7 instead of 16 VALU slots per value.  The launch code (conv_igemm.hip,
// qadd_pow2_args) checks the conditions on the five floats and passes A, B in q_rx, q_ra with q_scale_x = 0 as the marker.
template <int N>
__device__ __forceinline__ void qadd_n(float (&v)[N], const float (&r)[N], const ConvArgs& a) {
    if (a.q_scale_x == 0.f) {      // uniform
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = round_clamp_i8(__fmaf_rn(v[e], a.q_rx, __fmul_rn(r[e], a.q_ra)));
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = qadd_value(v[e], r[e], a);
    }
}

template <typename T> struct ResVec { typedef f16x4 type; };
template <> struct ResVec<float> { typedef f32x4 type; };
template <> struct ResVec<int8_t> { typedef unsigned type; };

// residual columns loaded ahead of their stores: about 16 registers' worth (the 4-waves-per-SIMD kernels have 128 in all)
template <typename T, int TM, int TN> struct ResChunk {
    static constexpr int value = sizeof(T) == 2 ? (8 / TM >= TN ? TN : (8 / TM >= 1 ? 8 / TM : 1)) : 1;
};

template <int ACT> __device__ __forceinline__ float activate_c(float v, float slope) {
    if constexpr (ACT == YH_ACT_LEAKY) return v > 0.f ? v : v * slope;
    else if constexpr (ACT == YH_ACT_RELU) return fmaxf(v, 0.f);
    else if constexpr (ACT == YH_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
    else if constexpr (ACT == YH_ACT_HSWISH) return v * (fminf(fmaxf(v + 3.f, 0.f), 6.f) / 6.f);
    else if constexpr (ACT == YH_ACT_MISH) return activate(v, YH_ACT_MISH, slope);
    else return v;
}

// the fp16 epilogues' activation (round 5; DESIGN.md 3, round 4 item c''): Mish through the hardware exp2 / rcp form of common.h (~20
// VALU slots per value instead of the ~45 of ocml expf + an IEEE divide; within 1e-6 relative of it for every float: the rounded
// fp16 value differs by at most one ulp on ~0.1 % of the values) - YOLOv4's 1x1 layers were VALU-bound on this, not byte-bound.  The
// fp32 kernels keep the exact form (they are the side of the fp32 box / conf comparisons against the oracle).
template <int ACT, typename T> __device__ __forceinline__ float activate_t(float v, float slope) {
    if constexpr (ACT == YH_ACT_MISH && sizeof(T) == 2) return mish_fast(v);
    else return activate_c<ACT>(v, slope);
}

// the same choice for the epilogues that take the activation at run time (residual / 2x-store / fp32-out forms): every fp16 kernel
// evaluates Mish by ONE formula, so two kernels that serve the same layer shape stay bit-identical to each other
template <typename T> __device__ __forceinline__ float activate_rt(float v, int act, float slope) {
    if (sizeof(T) == 2 && act == YH_ACT_MISH) return mish_fast(v);
    return activate(v, act, slope);
}

// the int8 epilogues' activation: activate_c, with mish through common.h's mish_for_grid (same grid value, ~half the instructions)
template <int ACT> __device__ __forceinline__ float activate_q(float v, float slope, float inv_out_scale) {
    if constexpr (ACT == YH_ACT_MISH) return mish_for_grid(v, inv_out_scale);
    else return activate_c<ACT>(v, slope);
}

// int8 epilogue of one fragment: q[e] = round_clamp_i8(act(acc[e] * s_w s_x + bias[e]) / s_a), the four values of a lane together
// (mish takes its fast / exact decision once per fragment, common.h mish_for_grid_n)
// (CHUNK 2: the decision per pair - same grid values, fewer live registers; conv_stream3.hip's 128-channel residual form needs it)
template <int ACT, typename AccV, int CHUNK = 4>
__device__ __forceinline__ void quantize4(const AccV& acc, const f32x4& bias, const ConvArgs& a, float (&q)[4]) {
    float y[4];
    if constexpr (ACT == YH_ACT_MISH && CHUNK == 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float y2[2], q2[2];
            y2[0] = (float)acc[2 * h] * a.acc_scale + bias[2 * h];
            y2[1] = (float)acc[2 * h + 1] * a.acc_scale + bias[2 * h + 1];
            mish_quantize_n<2>(y2, a.inv_out_scale, q2);
            q[2 * h] = q2[0];
            q[2 * h + 1] = q2[1];
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = (float)acc[e] * a.acc_scale + bias[e];
    if constexpr (ACT == YH_ACT_MISH) {
        mish_quantize_n<4>(y, a.inv_out_scale, q);      // tie test and rounding share one scaled value (common.h)
        return;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = activate_c<ACT>(y[e], a.slope);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) q[e] = round_clamp_i8(y[e] * a.inv_out_scale);
}

// The common case of the epilogue - plain dense NHWC store (ups == 1) - with the activation fixed at compile time: the code a
// wave executes for its 32 fragments is then ~1.5 k instructions instead of a 40 k-instruction body with every activation and
// every store form unrolled per fragment (instruction-cache misses dominated the generic form on the 128 x 64 wave tiles).
template <typename T, typename OutT, int TM, int TN, int BN, int WN, int ACT, typename AccT>
__device__ __forceinline__ void conv_epilogue_plain(const ConvArgs& a, AccT (&acc)[TM][TN], const f32x4 (&bvs)[TM], const int m0,
                                                    const long p0, const int wm, const int wn, const int lane) {
    const int mq = (lane >> 4) << 2, pc = lane & 15;
    OutT* const yg = reinterpret_cast<OutT*>(a.y);
    const T* const rg = reinterpret_cast<const T*>(a.res);
    const int mbase = m0 + wm * TM * 16 + mq;
    const long pbase = p0 + wn * TN * 16 + pc;
    auto value = [&](auto ic, auto jc, int e) {
        constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
        if constexpr (sizeof(T) == 1) {
            const float y = activate_q<ACT>((float)acc[i][j][e] * a.acc_scale + bvs[i][e], a.slope, a.inv_out_scale);
            const float q = round_clamp_i8(y * a.inv_out_scale);
            return sizeof(OutT) == 1 ? q : q * a.out_scale;
        } else {
            return activate_t<ACT, T>((float)acc[i][j][e] + bvs[i][e], a.slope);
        }
    };
    if constexpr (sizeof(T) != 1) {
        if (a.stats_part != nullptr) {   // BatchNorm partial sums (see conv_epilogue): one channel quad at a time
            float* const row = a.stats_part + ((long)(p0 / BN) * WN + wn) * 2 * a.Cout;
            static_for<TM>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
                static_for<TN>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if (pbase + j * 16 < a.P) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float q = (float)(OutT)value(ic, jc, e);
                            s1[e] += q;
                            s2[e] = fmaf(q, q, s2[e]);
                        }
                    }
                });
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t1 = row16_sum(s1[e]), t2 = row16_sum(s2[e]);
                    const int m = mbase + i * 16 + e;
                    if (pc == 15 && m < a.Cout) {
                        row[m] = t1;
                        row[a.Cout + m] = t2;
                    }
                }
            });
        }
    }
    typedef typename ResVec<T>::type res_t;
    constexpr int JCH = ResChunk<T, TM, TN>::value;
    static_assert(TN % JCH == 0, "column chunks must tile the wave tile");
    static_for<TN / JCH>([&](auto cc) {
        constexpr int j0 = decltype(cc)::value * JCH;
        res_t rv[JCH][TM];
        bool have_res = false;
        {
            have_res = rg != nullptr;
            if (have_res) {
                static_for<JCH>([&](auto jc) {
                    constexpr int jj = decltype(jc)::value;
                    const long p = pbase + (j0 + jj) * 16;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        rv[jj][i] = res_t{};
                        if (p < a.P && mbase + i * 16 < a.Cout) rv[jj][i] = *reinterpret_cast<const res_t*>(rg + p * a.ldr + mbase + i * 16);
                    }
                });
                static_for<JCH>([&](auto jc) {   // one wait for the whole chunk, before its first store
                    constexpr int jj = decltype(jc)::value;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        res_t t = rv[jj][i];
                        asm volatile("" : "+v"(t));
                        rv[jj][i] = t;
                    }
                });
            }
        }
        static_for<JCH>([&](auto jc) {
            constexpr int jj = decltype(jc)::value;
            typedef std::integral_constant<int, j0 + jj> J;
            const long p = pbase + (j0 + jj) * 16;
            if (p >= a.P) return;
            OutT* const prow = yg + p * a.ldy;
            static_for<TM>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const int m = mbase + i * 16;
                if (m >= a.Cout) return;
                float v[4];
                if constexpr (sizeof(T) == 1) {
                    quantize4<ACT>(acc[i][J::value], bvs[i], a, v);
                    if constexpr (sizeof(OutT) != 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= a.out_scale;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = value(ic, J{}, e);
                }
                if constexpr (sizeof(T) != 1) {
                    if (have_res) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)rv[jj][i][e];
                    }
                } else if constexpr (sizeof(OutT) == 1) {
                    if (have_res) {   // fused quantised shortcut: v holds the grid values the conv would have stored
                        float r4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) r4[e] = (float)(int8_t)((rv[jj][i] >> (8 * e)) & 0xff);
                        qadd_n<4>(v, r4, a);
                    }
                }
                store4<OutT>(prow + m, v[0], v[1], v[2], v[3]);
            });
        });
    });
}

template <typename T, typename OutT, int TM, int TN, int BN, int WN, typename AccT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, AccT (&acc)[TM][TN], const int m0, const long p0, const int wm,
                                              const int wn, const int lane) {
    const int HoWo = a.Ho * a.Wo;
    const int mq = (lane >> 4) << 2, pc = lane & 15;
    OutT* const yg = reinterpret_cast<OutT*>(a.y);
    const T* const rg = reinterpret_cast<const T*>(a.res);
    // Every LOAD of the epilogue is issued before the first STORE: loads and stores share one in-order counter (vmcnt), so a
    // bias or residual load placed after a store has to wait for that store's write acknowledgement (~2000 cycles under load;
    // measured with s_memtime stamps: 70 000 cycles for the 32 fragments of a 128 x 64 wave tile, as long as 21 K tiles).
    // Bias (TM x 4 floats) up front; residual fragments in chunks of pixel columns (~32 registers), loads of a chunk before its stores.
    f32x4 bvs[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * TM * 16 + i * 16 + mq;
        bvs[i] = m < a.Cout ? *reinterpret_cast<const f32x4*>(a.bias + m) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // consume the loaded registers HERE, on the straight-line path: the compiler then waits for the loads once, now; left to
    // their first uses inside the branchy code below, it re-waits (vmcnt(0), i.e. also for every store issued so far) in each
    // conditional block because some path around the previous wait exists
#pragma unroll
    for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(bvs[i]));
    if (a.ups == 1) {
        switch (a.act) {
#define YH_PLAIN(A) case A: conv_epilogue_plain<T, OutT, TM, TN, BN, WN, A>(a, acc, bvs, m0, p0, wm, wn, lane); return
            // the activations of the YOLOv3 / v4 graphs; the Mobilenet ones (relu, relu6, h-swish) take the general form below
            YH_PLAIN(YH_ACT_LINEAR); YH_PLAIN(YH_ACT_LEAKY); YH_PLAIN(YH_ACT_MISH);
#undef YH_PLAIN
            default: break;
        }
    }
    // ---- the other store forms (2x nearest store, stride-2 data-gradient scatters), run-time activation
    struct Col { bool ok; long p, opix; int wo2, ho4, wo4; };
    auto column = [&](int j) {   // destination geometry of pixel column j of the wave tile
        Col c;
        c.p = p0 + wn * TN * 16 + j * 16 + pc;
        c.ok = c.p < a.P;
        c.opix = c.p;
        c.wo2 = c.ho4 = c.wo4 = 0;
        if (!c.ok) return c;
        if (a.ups == 2) {
            const int n = (int)(c.p / HoWo);
            const int rem = (int)(c.p - (long)n * HoWo);
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            c.wo2 = 2 * a.Wo;
            c.opix = ((long)n * 2 * a.Ho + 2 * ho) * c.wo2 + 2 * wo;
        } else if (a.ups == 3) {  // phase scatter (stride-2 data gradient): every other pixel of a y_h x y_w tensor
            const int n = (int)(c.p / HoWo);
            const int rem = (int)(c.p - (long)n * HoWo);
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            c.opix = ((long)n * a.y_h + 2 * ho + a.y_off_h) * a.y_w + 2 * wo + a.y_off_w;
        } else if (a.ups == 4) {  // all four phases: the 2x2 output block of (n, ho, wo); the row group picks the corner
            const int n = (int)(c.p / HoWo);
            const int rem = (int)(c.p - (long)n * HoWo);
            c.ho4 = rem / a.Wo;
            c.wo4 = rem - c.ho4 * a.Wo;
            c.opix = ((long)n * a.y_h + 2 * c.ho4) * a.y_w + 2 * c.wo4;
        }
        return c;
    };
    struct Row { bool ok; int mc; long pix; };
    auto channel = [&](const Col& c, int i) {   // destination of channel quad i in column c
        Row r;
        const int m = m0 + wm * TM * 16 + i * 16 + mq;
        r.ok = c.ok && m < a.Cout;
        r.mc = m;
        r.pix = c.opix;
        if (a.ups == 4 && r.ok) {
            const int cpp = a.Cout >> 2, ph = m / cpp;
            r.mc = m - ph * cpp;
            if (2 * c.ho4 + (ph >> 1) >= a.y_h || 2 * c.wo4 + (ph & 1) >= a.y_w) r.ok = false;
            r.pix = c.opix + (long)(ph >> 1) * a.y_w + (ph & 1);
        }
        return r;
    };

    if constexpr (sizeof(T) != 1) {
        if (a.stats_part != nullptr) {
            // BatchNorm batch statistics of this wave's outputs (training forward: plain dense store, no residual), one channel
            // quad at a time so that only 8 sums are live: values as stored (rounded to the output type), summed over the wave's
            // pixel columns in column order, then over the 16 pixel lanes on the DPP path (row_shr 1, 2, 4, 8 with zero fill:
            // lane 15 of each row ends with the total); one row of partials per (pixel tile, wave column)
            float* const row = a.stats_part + ((long)(p0 / BN) * WN + wn) * 2 * a.Cout;
            static_for<TM>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
                static_for<TN>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if (p0 + wn * TN * 16 + j * 16 + pc < a.P) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float q = (float)(OutT)activate_rt<T>((float)acc[i][j][e] + bvs[i][e], a.act, a.slope);
                            s1[e] += q;
                            s2[e] = fmaf(q, q, s2[e]);
                        }
                    }
                });
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t1 = row16_sum(s1[e]), t2 = row16_sum(s2[e]);
                    const int m = m0 + wm * TM * 16 + i * 16 + mq + e;
                    if (pc == 15 && m < a.Cout) {
                        row[m] = t1;
                        row[a.Cout + m] = t2;
                    }
                }
            });
        }
    }

    typedef typename ResVec<T>::type res_t;
    constexpr int JCH = ResChunk<T, TM, TN>::value;
    static_assert(TN % JCH == 0, "column chunks must tile the wave tile");
    static_for<TN / JCH>([&](auto cc) {
        constexpr int j0 = decltype(cc)::value * JCH;
        res_t rv[JCH][TM];
        if constexpr (sizeof(T) != 1) {
            if (rg != nullptr) {
                static_for<JCH>([&](auto jc) {
                    constexpr int jj = decltype(jc)::value;
                    const Col c = column(j0 + jj);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const Row r = channel(c, i);
                        rv[jj][i] = res_t{};
                        if (r.ok) rv[jj][i] = *reinterpret_cast<const res_t*>(rg + (a.ups >= 3 ? r.pix : c.p) * a.ldr + r.mc);
                    }
                });
                static_for<JCH>([&](auto jc) {   // one wait for the whole chunk, before its first store (see the bias above)
                    constexpr int jj = decltype(jc)::value;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        res_t t = rv[jj][i];
                        asm volatile("" : "+v"(t));
                        rv[jj][i] = t;
                    }
                });
            }
        }
        // compile-time j (template recursion): with 8 x 4 fragments per wave a plain `#pragma unroll` is left rolled and the
        // accumulators would be indexed at run time, i.e. live in scratch
        static_for<JCH>([&](auto jc) {
            constexpr int jj = decltype(jc)::value, j = j0 + jj;
            const Col c = column(j);
            if (!c.ok) return;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const Row r = channel(c, i);
                if (!r.ok) continue;
                const f32x4 bv = bvs[i];
                float v[4];
                if constexpr (sizeof(T) == 1) {
                    // PTQ eval arithmetic (quantized_ptq_cos.py:288-296,543-567,717): dequantised conv + quantised bias,
                    // activation in fp32, then round-half-away/clamp onto the activation grid
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = (float)acc[i][j][e] * a.acc_scale + bv[e];
                        const float y = a.act == YH_ACT_MISH ? mish_for_grid(t, a.inv_out_scale) : activate(t, a.act, a.slope);
                        const float q = round_clamp_i8(y * a.inv_out_scale);
                        v[e] = sizeof(OutT) == 1 ? q : q * a.out_scale;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = activate_rt<T>((float)acc[i][j][e] + bv[e], a.act, a.slope);
                    if (rg != nullptr) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)rv[jj][i][e];
                    }
                }
                OutT* dst = yg + r.pix * a.ldy + r.mc;
                store4<OutT>(dst, v[0], v[1], v[2], v[3]);
                if (a.ups == 2) {
                    store4<OutT>(dst + a.ldy, v[0], v[1], v[2], v[3]);
                    store4<OutT>(dst + (long)c.wo2 * a.ldy, v[0], v[1], v[2], v[3]);
                    store4<OutT>(dst + (long)(c.wo2 + 1) * a.ldy, v[0], v[1], v[2], v[3]);
                }
            }
        });
    });
}

// conv_igemm_k64.hip: full-line K step kernels (tile codes 61 - 63); f16 and int8, OutT by out_f32
int launch_k64_tile(const ConvArgs& a, int tile, int dtype, int out_f32, hipStream_t stream);
int launch_pointwise_tile(const ConvArgs& a, int dtype, int out_f32, hipStream_t stream);   // conv_pointwise.hip, tile code 71
int launch_stream3_tile(const ConvArgs& a, int dtype, hipStream_t stream);                    // conv_stream3.hip, tile code 72
long stream3_stats_rows(long P, int cout);                                                       // its statistics rows: one per wave
// conv_pw_lds.hip, tile code 73: 1x1 / s1 with the whole weight matrix (<= 64 KB, Cout <= 256, Cin <= 256) resident in LDS; f16,
// plain / residual / statistics forms
int launch_pwl_tile(const ConvArgs& a, int dtype, hipStream_t stream);
bool pwl_supported(int dtype, int out_f32, int cin, int cin_k, int cout, long P, int ldx, int ldy, int ldr, const void* x, const void* y,
                   const void* res, bool stats, bool bwd);
long pwl_stats_rows(long P, int cout, bool bwd = false);                                                          // one row per pixel stream
// conv_halo_pp.hip, tile code 43: 3x3 / s1 / p1, 128 channels x 512 virtual pixels; f16 and int8, output type = input type
int launch_hpp_tile(const ConvArgs& a, int dtype, hipStream_t stream);
bool hpp_geometry(int W, int cin_k, int bk, int* rows_hp, int* lb, int* hbufs, size_t* lds);

}  // namespace yh
