// First-layer convolution on the matrix cores (round 4): 3 x 3 taps of the caller's 3-plane NCHW fp32 frames, stride 1 or 2, pad 1
// (reference models.py:88-113 with in_channels = 3; the frames arrive NCHW fp32 from utils/datasets.py).
//
// The VALU form in elementwise.hip issues 27 x cout / 2 packed FMAs per pixel and gathers its 27 samples from global memory: 0.70 ms
// at 608 x 608, batch 64, where the bytes (12 in + 2 cout out per pixel) need 0.3 ms.  Here the image tile is staged ONCE through LDS
// with coalesced row loads (the stage of the next tile is in flight while this one is multiplied), and the contraction is seven
// v_mfma_f32_16x16x4_f32 per 16 pixels x 16 channels: K = 27 taps*planes (padded to 28 with a zero weight row).  The fp32 MFMA multiplies
// exactly and accumulates in fp32 - the same arithmetic class as the fused multiply-add chain it replaces (summation order differs:
// results agree to ~1e-7 relative, and bit for bit wherever every partial sum is representable, e.g. the dyadic frames of the PTQ
// parity tests).  The matrix pipe runs at the packed-FMA rate (256 FLOP / clk / CU), but the VALU is now free for staging, the
// epilogue and - in training - the BatchNorm partial sums of the stored values (stats_ws), which removes yh_bn_stats' pass over z.
//
// Workgroup = 4 waves = one 16 x 32 output tile at a time (608 / 32 = 19, 416 / 32 = 13, 640 / 32 = 20: no ragged tiles at the
// training / detection sizes), persistent over tiles.  Wave w owns rows 4w .. 4w+3: eight 16-pixel groups.  Fragments:
//   A (weights)  a[t][j] = W[k = (lane >> 4) + 4 j][channel(t, lane & 15)]     resident, 7 x NT registers
//   B (samples)  b[j]    = patch[pixel lane & 15][k = (lane >> 4) + 4 j]      one ds_read_b32 each
//   D            lane holds channels 4 (lane >> 4) + r, r < 4, of pixel lane & 15
// With two channel tiles the rows of tile t are permuted (row m <-> channel 8 (m / 4) + 4 t + m % 4) so that a lane ends up with
// eight CONSECUTIVE channels of its pixel: one 16-byte store per pixel and lane (fp16), a pixel's 32 channels by four lanes of one
// instruction.
#include "common.h"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

namespace yh {

template <int S> struct StemTile {
    static constexpr int TH = 16, TW = 32;
    static constexpr int PR = (TH - 1) * S + 3, PC = (TW - 1) * S + 3;
    static constexpr int PCP = PC + 1;                 // 35 / 66 floats per row
    static constexpr int PLANE = PR * PCP;
    static constexpr int ELEMS = 3 * PR * PC;          // samples staged per tile
    static constexpr int PER_THREAD = (ELEMS + 255) / 256;
};

// Output values of a lane's CPL consecutive channels, packed as they are stored (the epilogue keeps all eight groups of a tile in
// registers and stores them after the next tile's samples were committed to LDS, see the kernel).
template <typename T, int N> struct StemPack;
template <int N> struct StemPack<f16, N> {
    typedef f16 __attribute__((ext_vector_type(N))) V;
    static __device__ __forceinline__ V pack(const float (&v)[N]) {
        V o;
#pragma unroll
        for (int e = 0; e < N; ++e) o[e] = (f16)v[e];
        return o;
    }
    static __device__ __forceinline__ float stored(float v) { return (float)(f16)v; }
};
template <int N> struct StemPack<float, N> {
    typedef float __attribute__((ext_vector_type(N))) V;
    static __device__ __forceinline__ V pack(const float (&v)[N]) {
        V o;
#pragma unroll
        for (int e = 0; e < N; ++e) o[e] = v[e];
        return o;
    }
    static __device__ __forceinline__ float stored(float v) { return v; }
};
template <int N> struct StemPack<int8_t, N> {
    typedef unsigned __attribute__((ext_vector_type(N / 4))) V;
    static __device__ __forceinline__ V pack(const float (&v)[N]) {
        V o;
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            unsigned u = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) u |= ((unsigned)(int)v[q * 4 + e] & 0xffu) << (8 * e);
            o[q] = u;
        }
        return o;
    }
    static __device__ __forceinline__ float stored(float v) { return v; }
};
template <> struct StemPack<int8_t, 4> {
    typedef unsigned V;
    static __device__ __forceinline__ V pack(const float (&v)[4]) {
        unsigned u = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) u |= ((unsigned)(int)v[e] & 0xffu) << (8 * e);
        return u;
    }
    static __device__ __forceinline__ float stored(float v) { return v; }
};

// grid.x = persistent workgroups over the n * tiles_y * tiles_x tiles, grid.y = blocks of 16 NT channels.
// ACT: YH_ACT_LINEAR / YH_ACT_LEAKY compiled in, -1 = the run-time switch of activate().  STATS: the BatchNorm partial sums.
// The VALU budget matters as much as the MFMAs here (a 16-pixel group is 7 NT MFMAs = 224 NT cycles of the matrix pipe; a wave issues
// one VALU instruction per 4 cycles): every per-element index is computed ONCE per thread (sample offsets in global memory and in
// LDS, the lane's store offset), tiles that lie inside the frame take a path without bounds arithmetic, and per-group offsets are
// compile-time constants that fold into the instructions' immediate fields.
// SPLIT (round 5; fp16 / int8 outputs): the contraction on v_mfma_f32_16x16x32_f16 instead of the fp32 MFMA, which runs at the
// vector rate (0.52 ms where the bytes need 0.30).  K = 27 padded to 32 = ONE MFMA of 16 cycles where the
// fp32 form needs seven of 32.  fp32 operands are split into two fp16 halves, v = hi + lo / 2048 with hi = fp16(v) and lo = fp16((v -
// hi) * 2048) (the scale keeps the residual out of fp16's subnormal range; 2048 is exact): w x = w_hi x_hi + (w_lo x_hi + w_hi x_lo) /
// 2048 to 2^-21 relative (the lo x lo term is dropped), three MFMAs with fp32 accumulation, the two cross terms in an accumulator of
// their own that is folded in with one fma.  Every partial product is exact in fp32; where BOTH operands are fp16 numbers - the dyadic
// frames and power-of-two weight grids of the int8 parity tests - lo is zero and the result equals the fp32 form bit for bit
// (whenever every partial sum is representable, the same condition as before).  The fp32 engine keeps the fp32 MFMA (YH_STEM_F32=1
// selects it for every precision).
template <typename T, int NT, int S, int ACT, bool STATS, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void conv_stem_mfma_kernel(const yh_stem_desc d, const int tiles_x, const int tiles_y, const int total_tiles) {
    typedef StemTile<S> G;
    constexpr int CO = 16 * NT, CPL = 4 * NT;          // channels per workgroup pass / per lane
    constexpr int NU = G::PER_THREAD;
    __shared__ float patch[2][3 * G::PLANE];
    __shared__ float red[4][2][CO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, px = lane & 15;
    const int co0 = blockIdx.y * CO;

    // resident weight fragments and the lane's bias / first channel
    float a[NT][7];
    f16x8 ah[NT], al[NT];            // SPLIT: k = 8 g + e, hi and scaled lo halves
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ch = co0 + (NT == 2 ? 8 * (px >> 2) + 4 * t + (px & 3) : px);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int k = g + 4 * j;
            a[t][j] = (!SPLIT && k < 27) ? d.w[k * d.cout_pad + ch] : 0.f;
        }
        if constexpr (SPLIT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 8 * g + e;
                const float w = k < 27 ? d.w[k * d.cout_pad + ch] : 0.f;
                const f16 hi = (f16)w;
                ah[t][e] = hi;
                al[t][e] = (f16)((w - (float)hi) * 2048.f);
            }
        }
    }
    const int ch_lane = co0 + CPL * g;                 // the lane's CPL consecutive channels (tile t: ch_lane + 4 t + r)
    f32x4 bias[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[t][r] = d.bias[ch_lane + 4 * t + r];
    // LDS offsets of the lane's seven samples of group 0 of its wave (k = 27: any valid word, its weight is zero)
    int laneoff[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int k = g + 4 * j;
        const int tap = k / 3, ci = k - tap * 3;
        laneoff[j] = (k < 27 ? ci * G::PLANE + (tap / 3) * G::PCP + tap % 3 : 0) + wave * 4 * S * G::PCP + px * S;
    }
    // SPLIT: the lane's eight samples k = 8 g + e (k >= 27: any valid word, its weight is zero); two 16-bit offsets per register in the
    // form with statistics, which is two registers short otherwise (tools/check_spills.py)
    constexpr bool PACK8 = SPLIT && STATS;
    int laneoff8[PACK8 ? 4 : 8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * g + e;
        const int tap = k / 3, ci = k - tap * 3;
        const int off = (k < 27 ? ci * G::PLANE + (tap / 3) * G::PCP + tap % 3 : 0) + wave * 4 * S * G::PCP + px * S;
        if constexpr (PACK8) laneoff8[e >> 1] = (e & 1) ? (laneoff8[e >> 1] | (off << 16)) : off;
        else laneoff8[e] = off;
    }
    const float inv_q = sizeof(T) == 1 ? 1.f / d.out_scale : 1.f;
    const bool lane_ch_ok = ch_lane < d.cout;          // cout % 8 == 0: whole lanes
    const int store_lane = (wave * 4 * d.wo + px) * d.ldy + ch_lane;      // element offset of the lane's group 0 inside a tile
    float ssum[NT][4], ssq[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) ssum[t][r] = ssq[t][r] = 0.f;

    // this thread's staged samples: offsets in LDS and relative to the tile's first sample in global memory.  Held in registers for
    // stride 1 (8 samples per thread); recomputed per tile for stride 2 (26 per thread: the registers are worth more than the VALU)
    constexpr bool KEEP = S == 1 && !(SPLIT && STATS);      // (the split form with statistics spilled 17 registers with the 16 offsets resident)
    constexpr int NK = KEEP ? NU : 1;
    int loff[NK], goff[NK];
    auto sample = [&](int u, int& lo, int& go, int tid) {
        const int e = min(tid + u * 256, G::ELEMS - 1);
        const int ci = e / (G::PR * G::PC), rem = e - ci * (G::PR * G::PC);
        const int pr = rem / G::PC, pc = rem - pr * G::PC;
        lo = ci * G::PLANE + pr * G::PCP + pc;
        go = (ci * d.h + pr) * d.w_in + pc;
    };
    if constexpr (KEEP) {
#pragma unroll
        for (int u = 0; u < NU; ++u) sample(u, loff[u], goff[u], tid);
    }
    const bool last_ok = tid + (NU - 1) * 256 < G::ELEMS;

    auto tile_origin = [&](int tile, int& n, int& oy0, int& ox0) {
        const int tx = tile % tiles_x, r = tile / tiles_x;
        const int ty = r % tiles_y;
        n = r / tiles_y;
        oy0 = ty * G::TH;
        ox0 = tx * G::TW;
    };
    auto interior = [&](int oy0, int ox0) {          // every sample inside the frame, every output inside the map
        return oy0 > 0 && ox0 > 0 && (oy0 + G::TH - 1) * S + 1 < d.h && (ox0 + G::TW - 1) * S + 1 < d.w_in && oy0 + G::TH <= d.ho &&
               ox0 + G::TW <= d.wo;
    };
    // A tile's samples go from global memory into registers (fetch) and, a tile of arithmetic later, into LDS (commit).  fetch must
    // not touch the loaded values - a use would wait for the loads at the top of the tile: the loads of tiles inside the frame and
    // of tiles that overlap its border are the SAME instructions (offsets clamped into the image, which is harmless inside), and
    // the zeroing of out-of-frame samples happens at commit through a bit mask computed from indices alone.
    float stage[NU];
    unsigned emask = ~0u;          // bit u: sample u of the tile being fetched lies inside the frame
    auto fetch = [&](int n, int oy0, int ox0) {
        const float* xin = d.x + (long)n * 3 * d.h * d.w_in;
        const int gy0 = oy0 * S - 1, gx0 = ox0 * S - 1;
        const int base = gy0 * d.w_in + gx0, lim = 3 * d.h * d.w_in - 1;
        int tl = tid;
        if constexpr (!KEEP) asm volatile("" : "+v"(tl));      // recompute per tile: keeps the offsets out of loop-invariant registers
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            int lo, go;
            if constexpr (KEEP) go = goff[u];
            else sample(u, lo, go, tl);
            stage[u] = xin[min(max(base + go, 0), lim)];
        }
        emask = ~0u;
        if (!interior(oy0, ox0)) {
            int te = tid;
            asm volatile("" : "+v"(te));
            unsigned m = 0;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int e = min(te + u * 256, G::ELEMS - 1);
                const int ci = e / (G::PR * G::PC), rem = e - ci * (G::PR * G::PC);
                const int pr = rem / G::PC, pc = rem - pr * G::PC;
                const bool ok = (unsigned)(gy0 + pr) < (unsigned)d.h && (unsigned)(gx0 + pc) < (unsigned)d.w_in;
                m |= ok ? 1u << u : 0u;
            }
            emask = m;
        }
    };
    // SPLIT: a sample is converted ONCE, when it is written to LDS, into the word fp16(v) | fp16((v - fp16(v)) * 2048) << 16 - a tile's
    // 1836 samples are read ~18 times each by the fragment gathers, which then only shuffle halves (first version: 40 conversion
    // instructions per 16-pixel group, 0.49 -> 0.42 ms; the byte floor is 0.30)
    auto split_word = [&](float v) {
        if constexpr (SPLIT) {
            const f16 hi = (f16)v;
            const f16 lo = (f16)((v - (float)hi) * 2048.f);
            const unsigned u = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
            return __uint_as_float(u);
        } else {
            return v;
        }
    };
    auto commit = [&](int buf, bool inside) {
        float* pw = patch[buf];
        int tl = tid;
        if constexpr (!KEEP) asm volatile("" : "+v"(tl));
        if (inside) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                int lo, go;
                if constexpr (KEEP) lo = loff[u];
                else sample(u, lo, go, tl);
                if (u < NU - 1 || last_ok) pw[lo] = split_word(stage[u]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                int lo, go;
                if constexpr (KEEP) lo = loff[u];
                else sample(u, lo, go, tl);
                if (u < NU - 1 || last_ok) pw[lo] = (emask >> u) & 1u ? split_word(stage[u]) : 0.f;
            }
        }
    };
    auto finish = [&](float v) {
        float y = ACT == YH_ACT_LINEAR ? v : ACT == YH_ACT_LEAKY ? (v > 0.f ? v : v * d.slope)
                  : ACT == YH_ACT_MISH ? (sizeof(T) == 1 ? mish_for_grid(v, inv_q) : sizeof(T) == 2 ? mish_fast(v) : activate(v, YH_ACT_MISH, 0.f))
                  : (sizeof(T) == 1 && d.act == YH_ACT_MISH) ? mish_for_grid(v, inv_q) : activate(v, d.act, d.slope);
        if constexpr (sizeof(T) == 1) {      // PTQ: onto the activation grid (round half away, clamp)
            const float s = y * inv_q;
            y = fminf(fmaxf(copysignf(floorf(fabsf(s) + 0.5f), s), -128.f), 127.f);
        }
        return y;
    };
    // The eight 16-pixel groups of this wave: results packed into registers (and into the statistics), stored later.  `vmcnt` counts
    // loads and stores in ONE in-order queue: stores issued here, behind the next tile's sample loads, would have to be acknowledged
    // by memory before those samples could be written to LDS - a full store round trip per tile (measured: 0.55 ms against 0.35).
    // With the order  [next tile's loads] - [MFMAs, results to registers] - [samples to LDS] - [stores]  the wait for the samples
    // covers only the previous tile's stores, which had a whole tile of arithmetic to drain.
    typedef StemPack<T, CPL> PK;
    typename PK::V outv[8];
    auto groups = [&](const float* pb, bool whole, int oy0, int ox0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int qlds = (q >> 1) * S * G::PCP + (q & 1) * 16 * S;      // compile-time after unrolling
            f32x4 acc[NT];
            if constexpr (SPLIT) {
                unsigned wd[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int lo8 = PACK8 ? ((e & 1) ? (int)((unsigned)laneoff8[e >> 1] >> 16) : (laneoff8[e >> 1] & 0xffff)) : laneoff8[PACK8 ? 0 : e];
                    wd[e] = __float_as_uint(pb[lo8 + qlds]);
                }
                typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
                u32x4s uh, ul;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uh[e] = __builtin_amdgcn_perm(wd[2 * e + 1], wd[2 * e], 0x05040100u);      // the two hi halves
                    ul[e] = __builtin_amdgcn_perm(wd[2 * e + 1], wd[2 * e], 0x07060302u);      // the two lo halves
                }
                const f16x8 bh = __builtin_bit_cast(f16x8, uh), bl = __builtin_bit_cast(f16x8, ul);
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4 hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh, bias[t], 0, 0, 0);
                    f32x4 cr = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh, zero, 0, 0, 0);
                    cr = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl, cr, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][r] = fmaf(cr[r], 1.f / 2048.f, hh[r]);
                }
            } else {
                float b[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) b[j] = pb[laneoff[j] + qlds];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][0], b[0], bias[t], 0, 0, 0);
#pragma unroll
                for (int j = 1; j < 7; ++j)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][j], b[j], acc[t], 0, 0, 0);
            }
            float v[CPL];
            if constexpr (sizeof(T) == 1 && ACT == YH_ACT_MISH) {      // four values per tie decision, one scaled value for test and rounding
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float in4[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
                    float q4[4];
                    mish_quantize_n<4>(in4, inv_q, q4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[t * 4 + r] = q4[r];
                }
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[t * 4 + r] = finish(acc[t][r]);
            }
            outv[q] = PK::pack(v);
            if constexpr (STATS) {
                const bool ok = whole || (lane_ch_ok && oy0 + wave * 4 + (q >> 1) < d.ho && ox0 + (q & 1) * 16 + px < d.wo);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float s = ok ? PK::stored(v[t * 4 + r]) : 0.f;
                        ssum[t][r] += s;
                        ssq[t][r] = fmaf(s, s, ssq[t][r]);
                    }
            }
            if (STATS || SPLIT || (q & 1)) __builtin_amdgcn_sched_barrier(0);      // interleave pairs of groups at most (registers)
        }
    };
    auto stores = [&](T* ytile, bool whole, int oy0, int ox0) {
        if (whole) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                *reinterpret_cast<typename PK::V*>(ytile + (store_lane + ((q >> 1) * d.wo + (q & 1) * 16) * d.ldy)) = outv[q];
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (lane_ch_ok && oy0 + wave * 4 + (q >> 1) < d.ho && ox0 + (q & 1) * 16 + px < d.wo)
                    *reinterpret_cast<typename PK::V*>(ytile + (store_lane + ((q >> 1) * d.wo + (q & 1) * 16) * d.ldy)) = outv[q];
        }
    };

    // every load issued so far is consumed HERE: a load still pending at the loop header would turn the first use of its value inside
    // the loop into a counted wait that - the queue being in order - also waits for the previous tile's stores on every later trip
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int j = 0; j < 7; ++j) asm volatile("" ::"v"(a[t][j]));
        if constexpr (SPLIT) {
            asm volatile("" ::"v"(ah[t]));
            asm volatile("" ::"v"(al[t]));
        }
        asm volatile("" ::"v"(bias[t]));
    }
    const bool all_channels = co0 + CO <= d.cout;
    int tile = blockIdx.x, buf = 0;
    int n, oy0, ox0;
    tile_origin(tile, n, oy0, ox0);
    fetch(n, oy0, ox0);
    commit(0, interior(oy0, ox0));
    __syncthreads();
    while (true) {
        const int next = tile + (int)gridDim.x;
        const bool more = next < total_tiles;
        int nn = 0, noy = 0, nox = 0;
        if (more) {
            tile_origin(next, nn, noy, nox);
            fetch(nn, noy, nox);
        }
        __builtin_amdgcn_sched_barrier(0);
        T* const ytile = reinterpret_cast<T*>(d.y) + (((long)n * d.ho + oy0) * d.wo + ox0) * d.ldy;
        const bool whole = all_channels && oy0 + G::TH <= d.ho && ox0 + G::TW <= d.wo;
        groups(patch[buf], whole, oy0, ox0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) commit(buf ^ 1, interior(noy, nox));
        __builtin_amdgcn_sched_barrier(0);
        stores(ytile, whole, oy0, ox0);
        if (!more) break;
        __syncthreads();
        buf ^= 1;
        tile = next;
        n = nn;
        oy0 = noy;
        ox0 = nox;
    }

    if constexpr (STATS) {      // one row of [2][cout] partial sums per persistent workgroup (fixed tile assignment: deterministic)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s0 = ssum[t][r], s1 = ssq[t][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    s0 += __shfl_xor(s0, m, 64);
                    s1 += __shfl_xor(s1, m, 64);
                }
                if (px == 0) {
                    red[wave][0][CPL * g + 4 * t + r] = s0;
                    red[wave][1][CPL * g + 4 * t + r] = s1;
                }
            }
        __syncthreads();
        if (tid < 2 * CO) {
            const int q = tid / CO, c = tid - q * CO;
            const float s = (red[0][q][c] + red[1][q][c]) + (red[2][q][c] + red[3][q][c]);
            if (co0 + c < d.cout) d.stats_ws[((long)blockIdx.x * 2 + q) * d.cout + co0 + c] = s;
        }
    }
}

static bool stem_mfma_shape(const yh_stem_desc& d) {
    if (d.cin != 3 || d.kh != 3 || d.kw != 3 || d.pad != 1 || (d.stride != 1 && d.stride != 2)) return false;
    if (d.cout_pad % 16 || d.cout % 8) return false;
    if ((long)d.n * 3 * d.h * d.w_in >= 0x7fffffffL || (long)d.h * d.w_in * 12 >= 0x7fffffffL) return false;
    if ((long)(d.ho + 16) * (d.wo + 32) * d.ldy >= 0x7fffffffL) return false;      // 32-bit offsets inside a frame / an output map
    return true;
}

typedef void (*stem_kern_t)(const yh_stem_desc, const int, const int, const int);

// the instantiation a descriptor runs on (nullptr: none)
static stem_kern_t stem_pick(const yh_stem_desc& d) {
    const bool wide = d.cout_pad % 32 == 0, s1 = d.stride == 1;
    const char* f32_env = getenv("YH_STEM_F32");      // A/B knob: the fp32 MFMA form for every precision
    const bool split = d.dtype != YH_F32 && !(f32_env && atoi(f32_env));
#define YH_STEMK1(T, ACT, STATS, SP)                                                                                       \
    (wide ? (s1 ? (stem_kern_t)conv_stem_mfma_kernel<T, 2, 1, ACT, STATS, SP> : (stem_kern_t)conv_stem_mfma_kernel<T, 2, 2, ACT, STATS, SP>) \
          : (s1 ? (stem_kern_t)conv_stem_mfma_kernel<T, 1, 1, ACT, STATS, SP> : (stem_kern_t)conv_stem_mfma_kernel<T, 1, 2, ACT, STATS, SP>))
    // (stride 2 keeps the fp32 form: its 26 staged samples per thread leave no room for the split operands - 14 .. 53 spilled registers)
#define YH_STEMK2(T, ACT, STATS, SP)                                                                                       \
    (wide ? (s1 ? (stem_kern_t)conv_stem_mfma_kernel<T, 2, 1, ACT, STATS, SP> : (stem_kern_t)conv_stem_mfma_kernel<T, 2, 2, ACT, STATS, false>) \
          : (s1 ? (stem_kern_t)conv_stem_mfma_kernel<T, 1, 1, ACT, STATS, SP> : (stem_kern_t)conv_stem_mfma_kernel<T, 1, 2, ACT, STATS, false>))
#define YH_STEMK(T, ACT, STATS) (split ? YH_STEMK2(T, ACT, STATS, !(std::is_same<T, float>::value)) : YH_STEMK1(T, ACT, STATS, false))
#define YH_STEMM(T)                                                                                                       \
    do {                                                                                                                   \
        if (d.stats_ws_floats > 0) return d.act == YH_ACT_LINEAR ? YH_STEMK(T, YH_ACT_LINEAR, true) : nullptr;             \
        if (d.act == YH_ACT_LINEAR) return YH_STEMK(T, YH_ACT_LINEAR, false);                                              \
        if (d.act == YH_ACT_LEAKY) return YH_STEMK(T, YH_ACT_LEAKY, false);                                                \
        if (d.act == YH_ACT_MISH) return YH_STEMK(T, YH_ACT_MISH, false);      /* YOLOv4's first block, round 6 */            \
        return YH_STEMK(T, -1, false);                                                                                     \
    } while (0)
    if (d.dtype == YH_F16) YH_STEMM(f16);
    if (d.dtype == YH_F32) YH_STEMM(float);
    if (d.dtype == YH_I8) {
        if (d.stats_ws_floats > 0) return nullptr;
        if (d.act == YH_ACT_LEAKY) return YH_STEMK(int8_t, YH_ACT_LEAKY, false);
        if (d.act == YH_ACT_MISH) return YH_STEMK(int8_t, YH_ACT_MISH, false);
        return YH_STEMK(int8_t, -1, false);
    }
#undef YH_STEMM
#undef YH_STEMK
#undef YH_STEMK1
#undef YH_STEMK2
    return nullptr;
}

// One round of resident workgroups (the kernel is persistent over tiles; more workgroups than fit would run in rounds of unequal
// length): occupancy x CUs, per kernel and device.  Also the number of statistics rows.
static unsigned stem_resident(stem_kern_t k) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, unsigned> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair((const void*)k, dev);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k), 256, 0) != hipSuccess || per_cu <= 0) per_cu = 2;
    if (const char* e = getenv("YH_STEM_PER_CU")) {      // A/B: persistent workgroups per CU
        if (atoi(e) > 0) per_cu = atoi(e);
        else fprintf(stderr, "stem_mfma: occupancy %d workgroups per CU\n", per_cu);
    }
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const unsigned r = (unsigned)per_cu * (unsigned)cus;
    cache[key] = r;
    return r;
}

static bool stem_mfma_geometry(const yh_stem_desc& d, stem_kern_t* kern, int* tiles_x, int* tiles_y, long* total, unsigned* grid) {
    *kern = stem_mfma_shape(d) ? stem_pick(d) : nullptr;
    if (!*kern) return false;
    *tiles_x = (d.wo + 31) / 32;
    *tiles_y = (d.ho + 15) / 16;
    *total = (long)d.n * *tiles_x * *tiles_y;
    if (*total <= 0 || *total > 0x7fffffffL) return false;
    const unsigned res = stem_resident(*kern);
    *grid = (unsigned)(*total < (long)res ? *total : (long)res);
    return true;
}

long stem_mfma_stats_rows(const yh_stem_desc& d0) {
    if (d0.dtype == YH_I8 || d0.act != YH_ACT_LINEAR) return 0;
    yh_stem_desc d = d0;
    d.stats_ws_floats = 1;      // select the instantiation with the statistics epilogue
    stem_kern_t kern;
    int tx, ty;
    long total;
    unsigned grid;
    return stem_mfma_geometry(d, &kern, &tx, &ty, &total, &grid) ? (long)grid : 0;
}

int launch_stem_mfma(const yh_stem_desc& d0, hipStream_t s) {
    yh_stem_desc d = d0;
    if (!d.stats_ws) d.stats_ws_floats = 0;
    else if (d.stats_ws_floats <= 0) return YH_EINVAL;
    stem_kern_t kern;
    int tx, ty;
    long total;
    unsigned gx;
    if (!stem_mfma_geometry(d, &kern, &tx, &ty, &total, &gx)) return YH_EUNSUPPORTED;
    if (d.stats_ws && d.stats_ws_floats < (int64_t)gx * 2 * d.cout) return YH_EINVAL;
    const bool wide = d.cout_pad % 32 == 0;
    const dim3 grid(gx, (unsigned)(d.cout_pad / (wide ? 32 : 16)));
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, d, tx, ty, (int)total);
    return check_launch();
}

}  // namespace yh
