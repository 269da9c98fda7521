// 1x1 convolutions over few channels on large grids (YOLOv4's CSP stages at 320^2 / 160^2, Darknet-53's 304^2 / 152^2 bottlenecks:
// Cin 32 .. 128, Cout <= 128): a streaming kernel without LDS.
//
// On these layers a 64 x 128 ring-kernel tile lives for ONE or two K steps: 25 600 workgroups that each set up an LDS ring, wait for
// one tile, and drain - int8 takes as long as fp16 (YOLOv4-640 batch 32: 14 launches, 2.5 ms, 4x off the HBM roofline in int8).  Here
// the whole weight matrix of a wave's output rows sits in registers (Cout x Cin <= 128 x 128: 16 - 64 VGPRs as MFMA A fragments),
// a wave walks over blocks of 64 pixels, and the B fragments are loaded straight from global memory in MFMA layout: lane
// (pc = lane & 15, kq = lane >> 4) of a 16 x 16 x 64-i8 (32-f16) MFMA needs 16 consecutive bytes of pixel pc at byte offset 16 kq of
// its channel row - one global_load_dwordx4 per lane, and the 16 pixels x 64 .. 256 bytes a fragment covers are contiguous NHWC rows.
// The next block's fragments are in flight while the current block's MFMAs and stores run.  Same epilogue arithmetic, operation for
// operation, as the LDS-DMA kernels (conv_igemm.h): results are bit-identical to them.
#include "conv_igemm.h"

namespace yh {

template <typename T> struct PwFrag;
template <> struct PwFrag<f16> { typedef f16x8 type; static constexpr int K = 32; };
template <> struct PwFrag<int8_t> { typedef u32x4 type; static constexpr int K = 64; };

template <typename T> __device__ __forceinline__ typename AccOf<T>::type pw_mfma(const typename PwFrag<T>::type& a,
                                                                                 const typename PwFrag<T>::type& b,
                                                                                 typename AccOf<T>::type c);
template <> __device__ __forceinline__ f32x4 pw_mfma<f16>(const f16x8& a, const f16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ i32x4 pw_mfma<int8_t>(const u32x4& a, const u32x4& b, i32x4 c) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), c, 0, 0, 0);
}

// MT: 16-row groups of output channels held by a wave (Cout <= 16 MT); KS: MFMA K steps (Cin = KS * K); TN = 4 pixel groups of 16
template <typename T, typename OutT, int MT, int KS, int ACT>
__global__ __launch_bounds__(256) void conv_pointwise_kernel(const ConvArgs a, const long nblocks) {
    typedef typename PwFrag<T>::type frag_t;
    typedef typename AccOf<T>::type acc_t;
    constexpr int K = PwFrag<T>::K, TN = 4, UNIT = 16 / (int)sizeof(T);   // elements per 16-byte unit
    const int lane = threadIdx.x & 63;
    const int pc = lane & 15, kq = lane >> 4, mq = kq << 2;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
    const T* const xg = reinterpret_cast<const T*>(a.x);
    OutT* const yg = reinterpret_cast<OutT*>(a.y);
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // whole 16-byte units per pixel row, 16-byte aligned rows: otherwise the direct 4-channel stores
    const bool coalesced = (a.Cout * (int)sizeof(OutT)) % 16 == 0 && (a.ldy * (int)sizeof(OutT)) % 16 == 0 &&
                           (reinterpret_cast<uintptr_t>(a.y) & 15u) == 0 && !a.no_lds_store;

    frag_t wa[MT][KS];
    f32x4 bvs[MT];
    {
        const T* const wg = reinterpret_cast<const T*>(a.w);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int k = 0; k < KS; ++k)
                wa[i][k] = *reinterpret_cast<const frag_t*>(wg + (long)(i * 16 + pc) * a.cin_k + k * K + kq * UNIT);
            const int m = i * 16 + mq;
            bvs[i] = m < a.Cout ? *reinterpret_cast<const f32x4*>(a.bias + m) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    auto load_block = [&](long blk, frag_t (&fb)[TN][KS]) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            long p = blk * (TN * 16) + j * 16 + pc;
            if (p >= a.P) p = a.P - 1;                       // tail: a valid address, the result is not stored
            const T* row = xg + p * a.ldx + kq * UNIT;
#pragma unroll
            for (int k = 0; k < KS; ++k) fb[j][k] = *reinterpret_cast<const frag_t*>(row + k * K);
        }
    };
    // two fragment buffers, the second copied into the first after each block: the next block's loads are in flight while this block
    // computes and stores.  (Rotating buffers without the copy, or a third buffer, cost 60 - 130 more VGPRs and were 25 - 40 % slower.)
    frag_t cur[TN][KS], nxt[TN][KS];
    auto compute = [&](const long blk) {
        acc_t acc[MT][TN];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc_t c = {0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < KS; ++k) c = pw_mfma<T>(wa[i][k], cur[j][k], c);
                acc[i][j] = c;
            }
        auto value = [&](int i, int j, int e) {
            if constexpr (sizeof(T) == 1) {
                const float y = activate_q<ACT>((float)acc[i][j][e] * a.acc_scale + bvs[i][e], a.slope, a.inv_out_scale);
                const float q = round_clamp_i8(y * a.inv_out_scale);
                return sizeof(OutT) == 1 ? q : q * a.out_scale;
            } else {
                return activate_t<ACT, T>((float)acc[i][j][e] + bvs[i][e], a.slope);
            }
        };
        auto value4 = [&](int i, int j, float (&o)[4]) {
            if constexpr (sizeof(T) == 1) {
                quantize4<ACT>(acc[i][j], bvs[i], a, o);
                if constexpr (sizeof(OutT) != 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] *= a.out_scale;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = value(i, j, e);
            }
        };
        if (coalesced) {
            // A lane owns 4 channels of one pixel per fragment, so a direct store instruction touches 16 pixel rows with 4 - 16
            // bytes each.  Through a per-wave LDS tile [64 pixels][16 MT channels] the wave instead writes whole rows: 16 bytes per
            // lane, consecutive lanes on consecutive 16-byte units of a pixel row.
            constexpr int ROWB = MT * 16 * (int)sizeof(OutT), PITCH = ROWB + 16, UNITS = ROWB / 16;
            char* const tile = lds + (threadIdx.x >> 6) * (64 * PITCH);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    float o[4];
                    value4(i, j, o);
                    store4<OutT>(reinterpret_cast<OutT*>(tile + (j * 16 + pc) * PITCH) + i * 16 + mq, o[0], o[1], o[2], o[3]);
                }
            __builtin_amdgcn_wave_barrier();
            const long p0 = blk * (TN * 16);
            const int valid_units = (a.Cout * (int)sizeof(OutT) + 15) / 16;
#pragma unroll
            for (int u0 = 0; u0 < 64 * UNITS; u0 += 64) {
                const int u = u0 + lane, row = u / UNITS, col = u - row * UNITS;
                const u32x4 v = *reinterpret_cast<const u32x4*>(tile + row * PITCH + col * 16);
                if (p0 + row < a.P && col < valid_units)
                    *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(yg + (p0 + row) * a.ldy) + col * 16) = v;
            }
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const long p = blk * (TN * 16) + j * 16 + pc;
                if (p < a.P) {
                    OutT* const prow = yg + p * a.ldy;
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const int m = i * 16 + mq;
                        if (m < a.Cout) {
                            float o[4];
                            value4(i, j, o);
                            store4<OutT>(prow + m, o[0], o[1], o[2], o[3]);
                        }
                    }
                }
            }
        }
    };
    long blk = wave;
    if (blk < nblocks) load_block(blk, cur);
    for (; blk < nblocks; blk += nwaves) {
        const long nb = blk + nwaves;
        if (nb < nblocks) load_block(nb, nxt);
        compute(blk);
        if (nb < nblocks) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int k = 0; k < KS; ++k) cur[j][k] = nxt[j][k];
        }
    }
}

template <typename T, typename OutT, int MT, int KS> static int launch_pw_act(const ConvArgs& a, hipStream_t s) {
    const long nblocks = (a.P + 63) / 64;
    long grid = (nblocks + 3) / 4;                       // 4 waves per workgroup
    const long cap = 256L * 8;                           // 8 workgroups per CU: 32 waves, ~half the VGPR budget each
    if (grid > cap) grid = cap;
    const size_t shmem = 4 * 64 * (size_t)(MT * 16 * sizeof(OutT) + 16);
    switch (a.act) {
#define YH_PW(A) case A: hipLaunchKernelGGL((conv_pointwise_kernel<T, OutT, MT, KS, A>), dim3((unsigned)grid), dim3(256), shmem, s, a, nblocks); break
        YH_PW(YH_ACT_LINEAR); YH_PW(YH_ACT_LEAKY); YH_PW(YH_ACT_MISH);
#undef YH_PW
        default: return YH_EUNSUPPORTED;
    }
    return check_launch();
}

template <typename T, typename OutT> static int launch_pw(const ConvArgs& a, hipStream_t s) {
    const int mt = (a.Cout + 15) / 16, ks = a.cin_k / PwFrag<T>::K;
    // register budget (A fragments 4 MT KS, accumulators 16 MT, two B buffers 32 KS): (MT, KS) in (2,1) (2,2) (2,4) (4,1) (4,2) (8,1)
    if (mt <= 2) {
        if (ks == 1) return launch_pw_act<T, OutT, 2, 1>(a, s);
        if (ks == 2) return launch_pw_act<T, OutT, 2, 2>(a, s);
        if (ks == 4) return launch_pw_act<T, OutT, 2, 4>(a, s);
    } else if (mt <= 4) {
        if (ks == 1) return launch_pw_act<T, OutT, 4, 1>(a, s);
        if (ks == 2) return launch_pw_act<T, OutT, 4, 2>(a, s);
    } else if (mt <= 8) {
        if (ks == 1) return launch_pw_act<T, OutT, 8, 1>(a, s);
    }
    return YH_EUNSUPPORTED;
}

// tile code 71 (see conv_igemm.hip yh_conv2d_tile): dtype / output type resolved here
int launch_pointwise_tile(const ConvArgs& a, int dtype, int out_f32, hipStream_t s) {
    if (dtype == YH_F16) return out_f32 ? launch_pw<f16, float>(a, s) : launch_pw<f16, f16>(a, s);
    if (dtype == YH_I8) return out_f32 ? launch_pw<int8_t, float>(a, s) : launch_pw<int8_t, int8_t>(a, s);
    return YH_EUNSUPPORTED;
}

}  // namespace yh
