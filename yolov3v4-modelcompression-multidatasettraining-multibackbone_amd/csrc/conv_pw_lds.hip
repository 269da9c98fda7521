// 1x1 convolutions whose whole weight matrix fits in LDS, as a PERSISTENT streaming kernel (tile code 73; round 4).
//
// Why: the 1x1 layers of Darknet-53 at 304^2 / 152^2 / 76^2 (64 <-> 32, 128 <-> 64, 256 <-> 128 channels; forward AND data gradient, the
// latter with its residual accumulate) are byte-bound - 85 .. 170 FLOP per byte - and ran at 3.1 - 3.9 TB/s on the ring kernels: a
// 256 x 128 tile lives for 4 - 8 K steps, so ring fill, first-tile latency and the epilogue are ~40 % of a workgroup's life, and two
// workgroups per CU only partly cover each other (profiles/r03_train_layers_final.txt: 9.8 ms of the
// 56 ms step).  conv_pointwise.hip removed that for Cout x Cin <= 128 x 32 .. 32 x 128 (weights in registers).  Here the weights
// (<= 64 KB) sit in LDS as ready-made MFMA A fragments, one workgroup per CU stays resident, and every wave streams over 32-pixel
// blocks:
//   * B fragments straight from global memory in MFMA layout (16 bytes of one pixel row per lane), refilled with the NEXT block's
//     bytes as soon as a K step's MFMAs have consumed them - one block of distance between a load and its use, straight-line code
//     (tails re-load the last block: a branch around a load would reset the compiler's counted waits, DESIGN.md "Toolchain facts");
//   * A fragments by one conflict-free ds_read_b128 each (MT per K step for 2 MT MFMAs);
//   * epilogue through a per-wave LDS tile so that loads and stores are whole pixel rows (16 bytes per lane, consecutive lanes on
//     consecutive units): the NEXT block's residual rows are loaded before this block's stores are issued (one in-order vmcnt queue),
//     parked in the tile, added in fp32 to the un-rounded accumulator and rounded ONCE - bit-identical to the ring kernels' epilogue
//     (conv_igemm.h conv_epilogue_plain), bias + activation included;
//   * training forward: BatchNorm partial sums of the values as stored, per lane over all of a wave's blocks (a lane always owns the
//     same 8 channels in the row-major pass), one partial row per pixel stream.
// Cout up to 256: MS = 2 waves share a pixel stream, each computing 128 of the output channels from the same B rows (the second
// wave's loads hit L1 / L2).  fp16, and int8 (PTQ inference: v_mfma_i32_16x16x64_i8, 64 channels per K step, the requantising
// epilogue of conv_igemm.h; plain mode only - the quantised shortcut follows the 3x3 layers, never a 1x1).
#include "conv_igemm.h"

namespace yh {

template <typename T> struct PwlMma;
template <> struct PwlMma<f16> {
    typedef f16x8 frag_t;
    typedef f32x4 acc_t;
    static __device__ __forceinline__ acc_t mma(const frag_t& a, const frag_t& b, const acc_t& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct PwlMma<int8_t> {
    typedef i32x4 frag_t;
    typedef i32x4 acc_t;
    static __device__ __forceinline__ acc_t mma(const frag_t& a, const frag_t& b, const acc_t& c) {
        return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    }
};

// MT: 16-channel row groups per wave (<= 8); KS: K steps of 32 (fp16) / 64 (int8) input channels; MS: waves per pixel stream
// (channel split); MODE 0 plain, 1 residual, 2 statistics (fp16 only).
// MODE 3 / 4 (round 6; fp16 data gradients, yh_conv_desc.bwd_z): the tensor this launch stores - with (3) or without (4) its residual
// accumulate - COMPLETES the gradient dy of a BatchNorm + activation block, so the block's backward sums are taken here, from the
// rows as stored: the row-major pass already holds 8 channels of one pixel per lane; it loads the matching 16 bytes of the block's
// pre-BatchNorm output z (issued before the K loop, consumed after it), forms g = dy act'(gamma xhat + beta), xhat = (z - mean) invstd
// with train.hip's arithmetic and keeps sum g / sum g xhat per lane over all of the wave's blocks - one partial row per pixel stream,
// as mode 2.  ACT is then the BLOCK's activation (the data gradient itself is linear).  What it saves: the separate reduction pass
// read dy and z (4 bytes per element); here z alone is read (2), in a kernel that streams 10 bytes per element anyway.
template <typename T, int MT, int KS, int MS, int ACT, int MODE>
__global__ __launch_bounds__(512, 2) void conv1x1_lds_kernel(const ConvArgs a, const int nblocks) {
    typedef typename PwlMma<T>::frag_t frag_t;
    typedef typename PwlMma<T>::acc_t acc_t;
    static_assert(sizeof(T) == 2 || MODE == 0, "int8: plain mode only");
    constexpr bool RES = MODE == 1 || MODE == 3, BWD = MODE == 3 || MODE == 4;
    constexpr int CACT = BWD ? YH_ACT_LINEAR : ACT;           // the convolution's own activation
    constexpr int ES = sizeof(T), VEC = 16 / ES, KB = 4 * VEC; // bytes per element, elements per 16-byte unit, channels per K step
    constexpr int NW = 8, TN = 2, BP = TN * 16;               // 8 waves; 32 pixels per block
    constexpr int ROWB = MT * 16 * ES;                        // bytes of this wave's channels in one pixel row
    constexpr int PITCH = ROWB + 16;                          // staging tile pitch (bank spread)
    constexpr int UNITS = ROWB / 16;                          // 16-byte units per pixel row of the wave's channel range
    constexpr int RPI = 64 / UNITS;                           // pixel rows per row-major pass instruction
    constexpr int NPASS = BP / RPI;                           // row-major instructions per block
    constexpr int W_BYTES = MS * MT * KS * 1024;              // A fragments: [MS * MT][KS][64 lanes][16 B]
    constexpr int BIAS_BYTES = MS * MT * 16 * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char pl_smem[];
    unsigned char* const wl = pl_smem;
    float* const bl = reinterpret_cast<float*>(pl_smem + W_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const tile = pl_smem + W_BYTES + BIAS_BYTES + wave * (BP * PITCH);
    const int pc = lane & 15, kq = lane >> 4, mq = kq << 2;
    const T* const xg = reinterpret_cast<const T*>(a.x);
    const T* const rg = reinterpret_cast<const T*>(a.res);
    T* const yg = reinterpret_cast<T*>(a.y);

    // ---- weights -> LDS as A fragments (fragment f = (row group, k step): lane (pc, kq) holds W[16 rg + pc][KB k + VEC kq .. +VEC-1])
    {
        const T* const wg = reinterpret_cast<const T*>(a.w);
        for (int f = wave; f < MS * MT * KS; f += NW) {
            const int rgp = f / KS, k = f - rgp * KS;
            const u32x4 v = *reinterpret_cast<const u32x4*>(wg + (long)(rgp * 16 + pc) * a.cin_k + k * KB + kq * VEC);
            *reinterpret_cast<u32x4*>(wl + (f * 64 + lane) * 16) = v;
        }
        for (int m = tid; m < MS * MT * 16; m += NW * 64) bl[m] = m < a.Cout ? a.bias[m] : 0.f;
    }
    __syncthreads();

    const int half = wave % MS;                               // which MT x 16 channels of the output this wave computes
    const int stream = (blockIdx.x * NW + wave) / MS, nstreams = gridDim.x * NW / MS;
    const int ch0 = half * MT * 16;
    const unsigned char* const wl_h = wl + half * (MT * KS * 1024) + lane * 16;
    const float* const bl_h = bl + ch0 + mq;

    // row-major pass geometry: lane -> (pixel row rr + RPI * pass, 16-byte unit cu) of the wave's channel range
    const int cu = lane % UNITS, rr = lane / UNITS;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    float pg[8], pb[8], pm[8], pi[8];                         // BWD: this lane's 8 channels of the block's BatchNorm
    const T* const zg = reinterpret_cast<const T*>(a.bz);
    if constexpr (BWD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = ch0 + cu * 8 + e;
            pg[e] = a.bgamma[c];
            pb[e] = a.bbeta[c];
            pm[e] = a.bmean[c];
            pi[e] = a.binvstd[c];
        }
    }

    frag_t fb[TN][KS];
    u32x4 rv[NPASS];
    u32x4 zv[BWD ? NPASS : 1];
    auto load_z = [&](int blk) {              // the block's z rows of THIS pixel block, row-major as the store pass
#pragma unroll
        for (int q = 0; q < (BWD ? NPASS : 0); ++q) {
            long p = (long)blk * BP + rr + q * RPI;
            p = p < a.P ? p : a.P - 1;
            zv[q] = *reinterpret_cast<const u32x4*>(zg + p * a.ldbz + ch0 + cu * VEC);
        }
    };
    auto load_b = [&](int blk, int k) {       // K step k of block blk (clamped: a tail re-loads valid bytes that are never stored)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            long p = (long)blk * BP + j * 16 + pc;
            p = p < a.P ? p : a.P - 1;
            fb[j][k] = *reinterpret_cast<const frag_t*>(xg + p * a.ldx + k * KB + kq * VEC);
        }
    };
    auto load_res = [&](int blk) {
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            long p = (long)blk * BP + rr + q * RPI;
            p = p < a.P ? p : a.P - 1;
            rv[q] = *reinterpret_cast<const u32x4*>(rg + p * a.ldr + ch0 + cu * VEC);
        }
    };

    int blk = stream;
    if (blk < nblocks) {
#pragma unroll
        for (int k = 0; k < KS; ++k) load_b(blk, k);
        if constexpr (RES) load_res(blk);
    }
    for (; blk < nblocks; blk += nstreams) {
        const int nb = blk + nstreams < nblocks ? blk + nstreams : blk;     // past the end: this block again (unused)
        if constexpr (BWD) load_z(blk);       // in flight over the K loop, used in the store pass
        if constexpr (RES) {
            // park this block's residual rows in the staging tile (row-major 16-byte units), then fetch the next block's
#pragma unroll
            for (int q = 0; q < NPASS; ++q) *reinterpret_cast<u32x4*>(tile + (rr + q * RPI) * PITCH + cu * 16) = rv[q];
        }
        acc_t acc[MT][TN];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            frag_t wa[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) wa[i] = *reinterpret_cast<const frag_t*>(wl_h + (i * KS + k) * 1024);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = PwlMma<T>::mma(wa[i], fb[j][k], acc[i][j]);
            load_b(nb, k);                     // refill: the next block's bytes for this K step
        }
        if constexpr (RES) load_res(nb);  // before this block's stores: loads and stores share one in-order queue
        __builtin_amdgcn_wave_barrier();
        // ---- phase 1: bias + activation (+ residual from the tile, fp32, one rounding) -> the tile, 4 channels of one pixel per lane
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bl_h + i * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                T* const cell = reinterpret_cast<T*>(tile + (j * 16 + pc) * PITCH) + i * 16 + mq;
                float v[4];
                if constexpr (sizeof(T) == 1) {      // the requantising epilogue of conv_igemm.h (conv_epilogue_plain)
                    quantize4<CACT>(acc[i][j], bv, a, v);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = activate_t<CACT, T>(acc[i][j][e] + bv[e], a.slope);
                }
                if constexpr (RES) {
                    const f16x4 r = *reinterpret_cast<const f16x4*>(cell);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
                }
                store4<T>(cell, v[0], v[1], v[2], v[3]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- phase 2: whole pixel rows to global memory
        const long p0 = (long)blk * BP;
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            const int row = rr + q * RPI;
            const u32x4 v = *reinterpret_cast<const u32x4*>(tile + row * PITCH + cu * 16);
            const bool ok = p0 + row < a.P;
            if (ok) *reinterpret_cast<u32x4*>(yg + (p0 + row) * a.ldy + ch0 + cu * VEC) = v;
            if constexpr (MODE == 2) {
                const f16x8 h = __builtin_bit_cast(f16x8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = ok ? (float)h[e] : 0.f;
                    s1[e] += f;
                    s2[e] = fmaf(f, f, s2[e]);
                }
            }
            if constexpr (BWD) {              // train.hip bn_act_bwd_reduce_kernel's arithmetic on the row as stored
                const f16x8 h = __builtin_bit_cast(f16x8, v), zz = __builtin_bit_cast(f16x8, zv[q]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = ((float)zz[e] - pm[e]) * pi[e];
                    const float u = pg[e] * xh + pb[e];
                    const float gg = (ok ? (float)h[e] : 0.f) * act_grad_t<f16>(u, ACT, a.bslope);
                    s1[e] += gg;
                    s2[e] = fmaf(gg, xh, s2[e]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (MODE == 2 || BWD) {
        // one partial row per pixel stream: [sum | sum of squares][Cout]; the RPI lanes that share a channel unit add up first
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int m = UNITS; m < 64; m <<= 1) {
                s1[e] += __shfl_xor(s1[e], m);
                s2[e] += __shfl_xor(s2[e], m);
            }
        if (lane < UNITS) {
            float* const row = a.stats_part + (long)stream * 2 * a.Cout + ch0 + cu * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                row[e] = s1[e];
                row[a.Cout + e] = s2[e];
            }
        }
    }
}

// (MT, KS, MS) of a layer, or false: Cout = 16 MT MS exactly, Cin = kb KS (kb = 32 fp16 / 64 int8), weights <= 64 KB, MT <= 8
// bwd (modes 3 / 4): 64 channels per wave (MT = 4) whatever Cout is - the z rows, the block's 32 BatchNorm parameters and the two sums
// per lane leave no room for 128-channel accumulators (256 VGPRs: the MT = 8 forms spilled 20 .. 108 bytes)
static bool pwl_shape(int cin_k, int cout, int kb, int* mt, int* ks, int* ms, bool bwd = false) {
    if (cin_k % kb || cout % 16 || cout < 32 || cout > 256 || cin_k < kb || cin_k > 8 * kb) return false;
    *ms = bwd ? cout / 64 : (cout > 128 ? 2 : 1);
    if (bwd && (cout % 64 || *ms < 1)) return false;
    if (cout % (16 * *ms)) return false;
    *mt = cout / (16 * *ms);
    *ks = cin_k / kb;
    if (*mt > 8 || (*mt != 2 && *mt != 4 && *mt != 8) || (*ks != 1 && *ks != 2 && *ks != 4 && *ks != 8)) return false;
    return (long)cout * cin_k * (kb == 32 ? 2 : 1) <= 64 * 1024;
}

static int pwl_ms(int cout, bool bwd) { return bwd ? cout / 64 : (cout > 128 ? 2 : 1); }

static int pwl_grid(long P, int ms, int* nblocks) {
    *nblocks = (int)((P + 31) / 32);
    long wgs = ((long)*nblocks * ms + 7) / 8;
    if (wgs > 256) wgs = 256;                    // one resident workgroup per CU
    return (int)(wgs < 1 ? 1 : wgs);
}

// int8 instantiations (launch_pwl_tile): 256 -> 128 (76^2 of Darknet-53), 128 -> 64, 128 -> 128, 64 -> 64, 256 -> 256 / 128 -> 256 (CSP stages)
static bool pwl_i8_case(int mt, int ks, int ms) {
    return (mt == 8 && ks == 4 && ms == 1) || (mt == 4 && ks == 2 && ms == 1) || (mt == 8 && ks == 2 && ms == 1) || (mt == 4 && ks == 1 && ms == 1) ||
           (mt == 8 && ks == 4 && ms == 2) || (mt == 8 && ks == 2 && ms == 2);
}

// fp16 instantiations (launch_pwl_tile's YH_PWL_CASE list, kept in step with it): every other (mt, ks, ms) stays on the ring kernels -
// pruned / custom cfgs reach shapes such as 256 -> 64 or 32 -> 256 (the auto-picked tile 73 returned YH_EUNSUPPORTED there)
static bool pwl_f16_case(int mt, int ks, int ms) {
    return (mt == 2 && ks == 2 && ms == 1) || (mt == 4 && ks == 1 && ms == 1) || (mt == 4 && ks == 4 && ms == 1) || (mt == 8 && ks == 2 && ms == 1) ||
           (mt == 8 && ks == 8 && ms == 1) || (mt == 8 && ks == 4 && ms == 2) || (mt == 8 && ks == 4 && ms == 1) || (mt == 4 && ks == 2 && ms == 1);
}

// the data-gradient shapes that carry a block's backward sums (modes 3 / 4): 32 -> 64 (304^2), 64 -> 128 (152^2), 128 -> 256 (76^2)
static bool pwl_bwd_case(int mt, int ks, int ms) {
    return mt == 4 && ((ks == 1 && ms == 1) || (ks == 2 && ms == 2) || (ks == 4 && ms == 4));
}

bool pwl_supported(int dtype, int out_f32, int cin, int cin_k, int cout, long P, int ldx, int ldy, int ldr, const void* x, const void* y,
                   const void* res, bool stats, bool bwd) {
    int mt, ks, ms;
    if ((dtype != YH_F16 && dtype != YH_I8) || out_f32 || cin != cin_k) return false;
    if (!pwl_shape(cin_k, cout, dtype == YH_I8 ? 64 : 32, &mt, &ks, &ms, bwd)) return false;
    if (bwd) return dtype == YH_F16 && !stats && pwl_bwd_case(mt, ks, ms) && ldx % 8 == 0 && ldy % 8 == 0 && (!res || ldr % 8 == 0) &&
                    aligned16(x) && aligned16(y) && (!res || aligned16(res)) && P > 0 && P < (1L << 31) - 64;
    if (stats && res) return false;
    if (dtype == YH_I8 && (stats || res || !pwl_i8_case(mt, ks, ms))) return false;      // plain mode, the shapes instantiated below
    if (dtype == YH_F16 && !pwl_f16_case(mt, ks, ms)) return false;
    const int vec = dtype == YH_I8 ? 16 : 8;
    if (ldx % vec || ldy % vec || (res && ldr % vec) || !aligned16(x) || !aligned16(y) || (res && !aligned16(res))) return false;
    return P > 0 && P < (1L << 31) - 64;
}

long pwl_stats_rows(long P, int cout, bool bwd) {
    int nblocks;
    const int ms = pwl_ms(cout, bwd);
    return (long)pwl_grid(P, ms, &nblocks) * 8 / ms;
}

template <typename T, int MT, int KS, int MS, int ACT> static int launch_pwl_mode(const ConvArgs& a, hipStream_t s) {
    int nblocks;
    const int grid = pwl_grid(a.P, MS, &nblocks);
    const size_t lds = (size_t)MS * MT * KS * 1024 + (size_t)MS * MT * 16 * 4 + (size_t)8 * 32 * (MT * 16 * sizeof(T) + 16);
#define YH_PWL_GO(MODE)                                                                                       \
    do {                                                                                                      \
        auto kern = conv1x1_lds_kernel<T, MT, KS, MS, ACT, MODE>;                                             \
        const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);                    \
        if (e != hipSuccess) return (int)e;                                                                   \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a, nblocks);                                  \
    } while (0)
    if constexpr (sizeof(T) == 1) {
        if (a.stats_part || a.res || a.bz) return YH_EUNSUPPORTED;
        YH_PWL_GO(0);
    } else {
        if (a.bz) {
            // backward sums of the block this data gradient completes: ACT is that block's activation
            if constexpr ((ACT == YH_ACT_LEAKY || ACT == YH_ACT_MISH) && MT == 4 && ((KS == 1 && MS == 1) || (KS == 2 && MS == 2) || (KS == 4 && MS == 4))) {
                if (!a.stats_part) return YH_EINVAL;
                if (a.res) YH_PWL_GO(3);
                else YH_PWL_GO(4);
            } else {
                return YH_EUNSUPPORTED;
            }
        } else if (a.stats_part) YH_PWL_GO(2);
        else if (a.res) YH_PWL_GO(1);
        else YH_PWL_GO(0);
    }
#undef YH_PWL_GO
    return check_launch();
}

template <typename T, int MT, int KS, int MS> static int launch_pwl_act(const ConvArgs& a, hipStream_t s) {
    switch (a.bz ? a.bact : a.act) {
        case YH_ACT_LINEAR: return launch_pwl_mode<T, MT, KS, MS, YH_ACT_LINEAR>(a, s);
        case YH_ACT_LEAKY: return launch_pwl_mode<T, MT, KS, MS, YH_ACT_LEAKY>(a, s);
        case YH_ACT_MISH: return launch_pwl_mode<T, MT, KS, MS, YH_ACT_MISH>(a, s);
        default: return YH_EUNSUPPORTED;
    }
}

// tile code 73 (conv_igemm.hip yh_conv2d_tile)
int launch_pwl_tile(const ConvArgs& a, int dtype, hipStream_t s) {
    int mt, ks, ms;
    if (!pwl_shape(a.cin_k, a.Cout, dtype == YH_I8 ? 64 : 32, &mt, &ks, &ms, a.bz != nullptr)) return YH_EUNSUPPORTED;
#define YH_PWL_CASE(T, M, K, S2) if (mt == M && ks == K && ms == S2) return launch_pwl_act<T, M, K, S2>(a, s)
    if (dtype == YH_I8) {
        YH_PWL_CASE(int8_t, 8, 4, 1);    // 256 -> 128   (76^2)
        YH_PWL_CASE(int8_t, 4, 2, 1);    // 128 ->  64   (152^2)
        YH_PWL_CASE(int8_t, 8, 2, 1);    // 128 -> 128   (YOLOv4 CSP stages)
        YH_PWL_CASE(int8_t, 4, 1, 1);    //  64 ->  64
        YH_PWL_CASE(int8_t, 8, 4, 2);    // 256 -> 256
        YH_PWL_CASE(int8_t, 8, 2, 2);    // 128 -> 256
        return YH_EUNSUPPORTED;
    }
    if (dtype != YH_F16) return YH_EUNSUPPORTED;
    YH_PWL_CASE(f16, 2, 2, 1);    //  64 ->  32   (304^2 forward)
    YH_PWL_CASE(f16, 4, 1, 1);    //  32 ->  64   (304^2 data gradient)
    YH_PWL_CASE(f16, 4, 4, 1);    // 128 ->  64   (152^2 forward)
    YH_PWL_CASE(f16, 8, 2, 1);    //  64 -> 128   (152^2 data gradient)
    YH_PWL_CASE(f16, 8, 8, 1);    // 256 -> 128   (76^2 forward)
    YH_PWL_CASE(f16, 8, 4, 2);    // 128 -> 256   (76^2 data gradient)
    YH_PWL_CASE(f16, 8, 4, 1);    // 128 -> 128   (YOLOv4 CSP stages)
    YH_PWL_CASE(f16, 4, 2, 1);    //  64 ->  64
    YH_PWL_CASE(f16, 4, 2, 2);    //  64 -> 128   (152^2 data gradient carrying a block's backward sums: 64 channels per wave)
    YH_PWL_CASE(f16, 4, 4, 4);    // 128 -> 256   (76^2, likewise)
#undef YH_PWL_CASE
    return YH_EUNSUPPORTED;
}

}  // namespace yh
