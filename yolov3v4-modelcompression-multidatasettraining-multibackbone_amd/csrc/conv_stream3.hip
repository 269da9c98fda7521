// 3x3 convolutions over ONE MFMA K step of input channels (fp16: Cin <= 32, int8: Cin <= 64) with at most 64 output channels, on
// large grids: Darknet-53's 608 -> 304 stem pair (3x3 / 2 32 -> 64, 3x3 32 -> 64 + shortcut), YOLOv4's 32 -> 64 / 64 -> 64 stages.
//
// On these layers an LDS-DMA ring tile (64 channels x 128 pixels) lives for nine K steps of 16 MFMAs: 46 000 workgroups per launch
// that each fill a three-stage ring, wait, and drain - 0.57 ms in fp16 and 0.60 ms in int8 at 304 x 304, batch 64, against 0.1 ms of
// HBM time and 0.1 - 0.2 ms of MFMA time (profiles/r03_layers_yolov3_608_*: `igemm_dma3_64x128`).  Here, like the 1x1 streaming
// kernel (conv_pointwise.hip), nothing is staged per tile:
//   * the whole weight tensor (<= 64 x 9 x 64 B = 36 KB) is put into LDS once per workgroup, already in MFMA A-fragment order
//     (one conflict-free ds_read_b128 per fragment);
//   * a wave walks over blocks of 32 output pixels; for each of the nine taps every lane loads its B fragment - 16 consecutive bytes
//     of one input pixel - straight from global memory (neighbouring taps hit L1 / L2), one whole block ahead of its use;
//     padding taps are zero fragments;
//   * the epilogue is the ring kernels' arithmetic operation for operation (bias, activation, fp16 residual, int8 requantisation and
//     fused quantised shortcut), so results are bit-identical to them; rows leave through a per-wave LDS tile as whole 16-byte units.
// Exactly 32 or 64 output channels on 16-byte aligned rows; BatchNorm statistics of the training forward as one partial row per wave
// (fp16, no residual); no upsample, no fp32 output.
#include "conv_igemm.h"

namespace yh {

template <typename T> struct S3Frag;
template <> struct S3Frag<f16> { typedef f16x8 type; static constexpr int K = 32; };
template <> struct S3Frag<int8_t> { typedef u32x4 type; static constexpr int K = 64; };

template <typename T> __device__ __forceinline__ typename AccOf<T>::type s3_mfma(const typename S3Frag<T>::type& a,
                                                                                 const typename S3Frag<T>::type& b,
                                                                                 typename AccOf<T>::type c);
template <> __device__ __forceinline__ f32x4 s3_mfma<f16>(const f16x8& a, const f16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ i32x4 s3_mfma<int8_t>(const u32x4& a, const u32x4& b, i32x4 c) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), c, 0, 0, 0);
}


// MT: 16-row groups of output channels (Cout == 16 MT exactly); blocks of TN = 2 pixel groups of 16; S3_WAVES waves per workgroup, two
// workgroups per CU, OCC waves per SIMD: 64 output channels need ~210 registers (72 for the nine taps' B fragments, 32 accumulators,
// 16 residual) -> 4-wave workgroups, 2 waves per SIMD; 32 output channels fit the 168 of 3 waves per SIMD (6-wave workgroups).
//
// The block loop is STRAIGHT-LINE code: no branch around any load or store (tails recompute and re-store the last pixel, a wave
// without a next block prefetches the zero page, the residual is a template parameter).  The compiler's s_waitcnt pass counts the
// in-order vmcnt queue exactly only through straight-line code; with `if (more)` / `if (p < P)` around the loads it fell back to the
// conservative count at every join and the last taps of a block waited for the NEXT block's prefetches (vmcnt(0) at tap 8: 0.50 ms
// instead of the numbers in profiles/r03_stream3_ab.txt).
// KS (round 4): MFMA K steps of input channels per tap - 2 for the fp16 data gradient of Darknet-53's conv3 (64 -> 32 at 304^2, which
// ran 0.76 ms on the register-staged 32 x 256 tile against a byte floor of 0.21): 18 B fragments per pixel group instead of 9.
template <typename T, int MT, int ACT, bool HAS_RES, bool STATS, int S3_WAVES, int OCC, int KS = 1>
__global__ __launch_bounds__(S3_WAVES * 64, OCC) void conv3x3_stream_kernel(const ConvArgs a, const long nblocks) {
    typedef T OutT;
    typedef typename S3Frag<T>::type frag_t;
    typedef typename AccOf<T>::type acc_t;
    typedef typename ResVec<T>::type res_t;
    constexpr int TN = 2, UNIT = 16 / (int)sizeof(T);
    constexpr int ROWB = MT * 16 * (int)sizeof(OutT), PITCH = ROWB + 16, UNITS = ROWB / 16;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pc = lane & 15, kq = lane >> 4, mq = kq << 2;
    const long wave = (long)blockIdx.x * S3_WAVES + wv, nwaves = (long)gridDim.x * S3_WAVES;
    const T* const xg = reinterpret_cast<const T*>(a.x);
    const T* const rg = reinterpret_cast<const T*>(a.res);
    OutT* const yg = reinterpret_cast<OutT*>(a.y);
    extern __shared__ __attribute__((aligned(16))) char lds[];
    frag_t* const wl = reinterpret_cast<frag_t*>(lds);                      // [MT * 9 * KS][64 lanes]
    char* const tile = lds + MT * 9 * KS * 64 * 16 + wv * (16 * PITCH);     // per wave: [16 pixels][PITCH]

    {
        const T* const wg = reinterpret_cast<const T*>(a.w);
        for (int idx = threadIdx.x; idx < MT * 9 * KS * 64; idx += S3_WAVES * 64) {
            const int l = idx & 63, itk = idx >> 6, it = itk / KS, k = itk - it * KS, i = it / 9, t = it - i * 9;
            wl[idx] = *reinterpret_cast<const frag_t*>(wg + (long)(i * 16 + (l & 15)) * a.ktot + t * a.cin_k + k * S3Frag<T>::K + (l >> 4) * UNIT);
        }
    }
    f32x4 bvs[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) bvs[i] = *reinterpret_cast<const f32x4*>(a.bias + i * 16 + mq);
    __syncthreads();

    bool cok[KS];                                    // lanes past the channel count (int8 Cin 32 in a 64-byte step) hold zeros
#pragma unroll
    for (int k = 0; k < KS; ++k) cok[k] = k * S3Frag<T>::K + kq * UNIT < a.Cin;
    const int HoWo = a.Ho * a.Wo;
    const char* const zpage = reinterpret_cast<const char*>(g_zero_page);
    // per pixel group: byte offset of input pixel (hi0, wi0) = the top-left tap, 32 bits (the picker keeps x below 2 GB)
    struct Geo { int off[TN]; int hi0[TN], wi0[TN]; };
    const char* const xb = reinterpret_cast<const char*>(xg);
    const int pix_bytes = a.ldx * (int)sizeof(T);
    auto geometry = [&](long blk) {
        Geo g;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            // 32-bit pixel index (the picker keeps P below 2^31): a 64-bit division is a branchy routine, and a branch inside the
            // block loop costs the exact vmcnt counts
            const int p = (int)min(blk * (TN * 16) + j * 16 + pc, a.P - 1);      // tail (and the block after the last): the last pixel again
            const int n = p / HoWo;
            const int rem = p - n * HoWo;
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            g.hi0[j] = ho * a.stride - 1;
            g.wi0[j] = wo * a.stride - 1;
            g.off[j] = ((n * a.H + g.hi0[j]) * a.W + g.wi0[j]) * pix_bytes + kq * 16;
        }
        return g;
    };
    auto load_tap = [&](const Geo& g, const int t, const int k, const bool live, frag_t (&f)[TN]) {
        const int r = t / 3, s = t - 3 * r;                  // compile-time after unrolling
        const int tap = (r * a.W + s) * pix_bytes + k * 64;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const bool ok = live && cok[k] && (unsigned)(g.hi0[j] + r) < (unsigned)a.H && (unsigned)(g.wi0[j] + s) < (unsigned)a.W;
            // a padding tap (or a lane past the channel count, or a wave without a next block) reads the zero page: the select sits
            // on the ADDRESS, so the load is unconditional and nothing waits for its data before the MFMAs that use it
            // (the address goes through an opaque register move: left visible, the select of two pointers under a load is turned
            // back into a branch with one load on each side)
            unsigned long long ad = ok ? reinterpret_cast<unsigned long long>(xb + (unsigned)(g.off[j] + tap))
                                       : reinterpret_cast<unsigned long long>(zpage);
            asm volatile("" : "+v"(ad));
            f[j] = *reinterpret_cast<const __attribute__((address_space(1))) frag_t*>(ad);
        }
    };

    // Software pipeline over blocks: the B fragments of all nine taps of a block are in registers / in flight at once (fr), and as
    // soon as a tap's MFMAs have consumed fr[t], the NEXT block's tap t is loaded into it - a whole block (72 MFMAs + epilogue) of
    // distance between a load and its use, with no second set of registers.  (One tap of prefetch measured 1.7 us PER TAP: every tap
    // paid a memory latency.)  The residual of a block is loaded at its start for the same reason.
    // training forward: this wave's sums of y and y * y (as stored) per channel over every pixel it produces; one row per wave
    float st1[MT][4], st2[MT][4];
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) st1[i][e] = st2[i][e] = 0.f;
    }
    long blk = wave;
    Geo g = geometry(blk);
    frag_t fr[9 * KS][TN];
    static_for<9 * KS>([&](auto tc) { load_tap(g, decltype(tc)::value / KS, decltype(tc)::value % KS, blk < nblocks, fr[decltype(tc)::value]); });
    for (; blk < nblocks; blk += nwaves) {
        const long p0 = blk * (TN * 16);
        res_t rv[TN][MT];
        if constexpr (HAS_RES) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const long p = min(p0 + j * 16 + pc, a.P - 1);
#pragma unroll
                for (int i = 0; i < MT; ++i) rv[j][i] = *reinterpret_cast<const res_t*>(rg + p * a.ldr + i * 16 + mq);
            }
        }
        const long nb = blk + nwaves;
        const bool more = nb < nblocks;
        g = geometry(nb);
        acc_t acc[MT][TN];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
        static_for<9 * KS>([&](auto tc) {
            constexpr int tk = decltype(tc)::value, t = tk / KS, k = tk % KS;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const frag_t wa = wl[((i * 9 + t) * KS + k) * 64 + lane];
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = s3_mfma<T>(wa, fr[tk][j], acc[i][j]);
            }
            load_tap(g, t, k, more, fr[tk]);
        });
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                float v[4];
                if constexpr (sizeof(T) == 1) {
                    quantize4<ACT, acc_t, (MT == 8 && HAS_RES) ? 2 : 4>(acc[i][j], bvs[i], a, v);     // (256 registers: no room for four)
                    if constexpr (HAS_RES) {
                        float r4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) r4[e] = (float)(int8_t)((rv[j][i] >> (8 * e)) & 0xff);
                        qadd_n<4>(v, r4, a);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (sizeof(T) == 1) {
                    } else {
                        float y = activate_t<ACT, T>((float)acc[i][j][e] + bvs[i][e], a.slope);
                        if constexpr (HAS_RES) y += (float)rv[j][i][e];
                        v[e] = y;
                    }
                }
                if constexpr (STATS) {
                    const bool counted = p0 + j * 16 + pc < a.P;      // the tail's repeated last pixel counts once
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float q = counted ? (float)(OutT)v[e] : 0.f;
                        st1[i][e] += q;
                        st2[i][e] = fmaf(q, q, st2[i][e]);
                    }
                }
                store4<OutT>(reinterpret_cast<OutT*>(tile + pc * PITCH) + i * 16 + mq, v[0], v[1], v[2], v[3]);
            }
            // rows leave as whole 16-byte units: lane -> (pixel row, unit) of the wave's [16 pixels][MT * 16 channels] tile
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u0 = 0; u0 < 16 * UNITS; u0 += 64) {
                const int u = (u0 + lane) % (16 * UNITS), row = u / UNITS, col = u - row * UNITS;   // (a 32-unit tile is written twice)
                const u32x4 w16 = *reinterpret_cast<const u32x4*>(tile + row * PITCH + col * 16);
                const long p = min(p0 + j * 16 + row, a.P - 1);
                *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(yg + p * a.ldy) + col * 16) = w16;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if constexpr (STATS) {      // every wave writes its row (zeros when it had no block): yh_bn_finalize sums all of them
        float* const row = a.stats_part + wave * 2 * a.Cout;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t1 = row16_sum(st1[i][e]), t2 = row16_sum(st2[i][e]);
                if (pc == 15) {
                    row[i * 16 + mq + e] = t1;
                    row[a.Cout + i * 16 + mq + e] = t2;
                }
            }
    }
}

// workgroups / waves of a launch: also the number of statistics rows (one per wave)
template <int WAVES> static long s3_grid(long P) {
    const long nblocks = (P + 31) / 32;
    const long grid = (nblocks + WAVES - 1) / WAVES;
    const long cap = WAVES == 8 ? 256 : 512;             // two workgroups per CU (one of eight waves for the 128-channel int8 form)
    return grid > cap ? cap : grid;
}
long stream3_stats_rows(long P, int cout) { return cout == 64 ? s3_grid<4>(P) * 4 : s3_grid<6>(P) * 6; }     // fp16 forms only

template <typename T, int MT, int WAVES, int OCC, int KS = 1> static int launch_s3_act(const ConvArgs& a, hipStream_t s) {
    const long nblocks = (a.P + 31) / 32;
    const long grid = s3_grid<WAVES>(a.P);
    const size_t shmem = (size_t)MT * 9 * KS * 64 * 16 + (size_t)WAVES * 16 * (MT * 16 * sizeof(T) + 16);
    const bool res = a.res != nullptr, stats = a.stats_part != nullptr;
    if (stats && (res || sizeof(T) == 1 || KS != 1)) return YH_EINVAL;      // (the two-step form is a data-gradient form: no statistics rows)
#define YH_S3_GO(A, R, S)                                                                                                      \
    do {                                                                                                                       \
        auto kern = conv3x3_stream_kernel<T, MT, A, R, S, WAVES, OCC, KS>;                                                     \
        const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), shmem);   /* int8 Cout 128: ~92 KB */    \
        if (e != hipSuccess) return (int)e;                                                                                    \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVES * 64), shmem, s, a, nblocks);                                \
    } while (0)
    switch (a.act) {
#define YH_S3(A)                                                         \
    case A:                                                               \
        if constexpr (sizeof(T) == 2 && KS == 1) {                        \
            if (stats) { YH_S3_GO(A, false, true); break; }               \
        }                                                                 \
        if (res) YH_S3_GO(A, true, false);                                \
        else YH_S3_GO(A, false, false);                                   \
        break
        YH_S3(YH_ACT_LINEAR); YH_S3(YH_ACT_LEAKY); YH_S3(YH_ACT_MISH);
#undef YH_S3
#undef YH_S3_GO
        default: return YH_EUNSUPPORTED;
    }
    return check_launch();
}

// tile code 72 (conv_igemm.hip yh_conv2d_tile): Cout 32 or 64 exactly; int8 also 128 (64 -> 128 at 304^2 / 152^2 is one K step there)
int launch_stream3_tile(const ConvArgs& a, int dtype, hipStream_t s) {
    if (dtype == YH_F16) {
        if (a.Cout == 32 && a.cin_k == 64) return launch_s3_act<f16, 2, 4, 2, 2>(a, s);     // two K steps per tap: 144 fragment registers
        if (a.cin_k != 32) return YH_EUNSUPPORTED;
        if (a.Cout == 32) return launch_s3_act<f16, 2, 6, 3>(a, s);
        if (a.Cout == 64) return launch_s3_act<f16, 4, 4, 2>(a, s);
    } else if (dtype == YH_I8) {
        if (a.Cout == 32) return launch_s3_act<int8_t, 2, 6, 3>(a, s);
        if (a.Cout == 64) return launch_s3_act<int8_t, 4, 4, 2>(a, s);
        if (a.Cout == 128) return launch_s3_act<int8_t, 8, 8, 2>(a, s);     // 74 KB of weights: ONE 8-wave workgroup per CU
    }
    return YH_EUNSUPPORTED;
}

}  // namespace yh
