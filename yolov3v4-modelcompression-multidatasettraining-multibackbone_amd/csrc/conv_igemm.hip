// Fused dense convolution block as an implicit GEMM on the CDNA4 matrix cores.
//
//   D[m][p] = sum_k  W'[m][k] * X[k][p]        m = output channel, p = (n, ho, wo) output pixel,
//                                              k = (tap r,s ; input channel c)
//   y = act(D + bias) (+ residual), stored NHWC (optionally 2x nearest-upsampled, optionally fp32)
//
// Operand roles are chosen for the NHWC epilogue: weights are the MFMA "A" operand (rows = channels),
// activations the "B" operand (columns = pixels), so a lane's 4 accumulator registers are 4
// *consecutive channels of one pixel* = one 8-byte (f16) or 16-byte (f32) store.
//
// Tiling: one workgroup = 4 waves computes a BM(channels) x BN(pixels) tile, K step BK = 4 "units" of
// 16 bytes per row (32 f16 / 16 f32 channels of one filter tap).  Both operand tiles are staged
// global -> VGPR -> LDS with the next K step's global loads issued before the current step's MFMAs
// (double-buffered LDS, one barrier per step).  LDS image is [unit][row] in 16-byte cells:
//   * a 16-row x 1-unit MFMA fragment read (ds_read_b128, lane = row) touches 16 consecutive cells:
//     conflict free for every lane group of the instruction;
//   * the staging store is arranged so every 8-lane group writes 8 consecutive cells (ds_write_b128).
// The im2col gather is done by the loader: each thread owns fixed pixel rows and walks (r, s, c0) with
// scalar counters; out-of-image taps and channel tails load zeros.
//
// Workgroup -> tile mapping is XCD-aware: the 8 XCDs each take a contiguous range of tiles, ordered so
// that the channel tiles of one pixel tile are adjacent (the activation tile is fetched from HBM once
// per XCD L2, weights are small and stay resident).
#include "conv_igemm.h"

namespace yh {

// One K step (KU units of 16 bytes per row) of MFMAs for a wave: TM x TN fragments of 16x16.
template <typename T, int TM, int TN, int KU> struct MmaStep;

template <int TM, int TN, int KU> struct MmaStep<f16, TM, TN, KU> {
    static __device__ __forceinline__ void run(const u32x4* As, const u32x4* Bs, int BM, int BN, int arow, int brow,
                                               int lane, f32x4 (&acc)[TM][TN]) {
        const int r = lane & 15;
#pragma unroll
        for (int h = 0; h < KU / 4; ++h) {  // 4 units x 8 halfs = K 32 per 16x16x32 MFMA
            const int u = h * 4 + (lane >> 4);
            f16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                u32x4 v = As[u * BM + arow + i * 16 + r];
                a[i] = *reinterpret_cast<f16x8*>(&v);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                u32x4 v = Bs[u * BN + brow + j * 16 + r];
                b[j] = *reinterpret_cast<f16x8*>(&v);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
};

template <int TM, int TN, int KU> struct MmaStep<float, TM, TN, KU> {
    static __device__ __forceinline__ void run(const u32x4* As, const u32x4* Bs, int BM, int BN, int arow, int brow,
                                               int lane, f32x4 (&acc)[TM][TN]) {
        const int kq = lane >> 4, r = lane & 15;
        const float* Af = reinterpret_cast<const float*>(As);
        const float* Bf = reinterpret_cast<const float*>(Bs);
#pragma unroll
        for (int u = 0; u < KU; ++u) {  // one unit = 4 floats = one 16x16x4 MFMA
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = Af[(u * BM + arow + i * 16 + r) * 4 + kq];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bf[(u * BN + brow + j * 16 + r) * 4 + kq];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
};

template <typename T, typename OutT, int BM, int BN, int WM, int WN, int KU>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int VEC = Prec<T>::VEC, BK = VEC * KU, UNITS = KU;
    constexpr int NW = WM * WN;              // waves per workgroup
    constexpr int RPW = 64 / KU;             // tile rows one wave stages per pass (64 lanes = RPW rows x KU units)
    constexpr int RPP = NW * RPW;            // rows per pass of the whole workgroup
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int APASS = (BM + RPP - 1) / RPP, BPASS = (BN + RPP - 1) / RPP;
    static_assert(KU == 4 || KU == 8, "K step is 4 or 8 units");
    static_assert(TM >= 1 && TN >= 1, "tile too small");

    __shared__ u32x4 smem[2 * UNITS * (BM + BN)];
    u32x4* const As = smem;                   // [2][UNITS][BM]
    u32x4* const Bs = smem + 2 * UNITS * BM;  // [2][UNITS][BN]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // ---- workgroup -> (pixel tile, channel tile); each XCD (blockIdx % 8) gets a contiguous range
    int m_tile, p_tile;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, rr = nb & 7, xcd = bid & 7;
        const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
        p_tile = logical / a.m_tiles;
        m_tile = logical - p_tile * a.m_tiles;
    }
    const int m0 = m_tile * BM;
    const long p0 = (long)p_tile * BN;

    // ---- loader coordinates: every 8-lane group stages 8 consecutive rows of one unit
    const int lu = (lane >> 3) % KU;
    const int lrow = wave * RPW + ((lane >> 3) / KU) * 8 + (lane & 7);

    const T* const xg = reinterpret_cast<const T*>(a.x);
    const T* wsrc[APASS];
    static_for<APASS>([&](auto c) {
        constexpr int ps = decltype(c)::value;
        int row = lrow + ps * RPP;
        if (row >= BM) row = BM - 1;  // inactive lanes still hold a valid address
        row = min(m0 + row, a.m_pad - 1);  // tiles taller than the packed image's 128-row padding
        wsrc[ps] = reinterpret_cast<const T*>(a.w) + (long)row * a.ktot + lu * VEC;
    });
    long bbase[BPASS];
    int bhi[BPASS], bwi[BPASS];
    const int HoWo = a.Ho * a.Wo;
    static_for<BPASS>([&](auto c) {
        constexpr int ps = decltype(c)::value;
        const int row = lrow + ps * RPP;
        const long p = p0 + row;
        if (row < BN && p < a.P) {
            const int n = (int)(p / HoWo);
            const int rem = (int)(p - (long)n * HoWo);
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            bhi[ps] = ho * a.stride - a.pad;
            bwi[ps] = wo * a.stride - a.pad;
            bbase[ps] = (((long)n * a.H + bhi[ps]) * a.W + bwi[ps]) * a.ldx + lu * VEC;
        } else {
            bhi[ps] = -(1 << 28);
            bwi[ps] = -(1 << 28);
            bbase[ps] = 0;
        }
    });

    u32x4 ra[APASS], rb[BPASS];
    int kr = 0, ks = 0, kc = 0;  // filter tap (r, s) and channel offset of the K step being loaded
    int kofs = 0;                // element offset of that K step inside a packed weight row

    auto load_step = [&]() {
        static_for<APASS>([&](auto c) {
            constexpr int ps = decltype(c)::value;
            if (BM % RPP == 0 || lrow + ps * RPP < BM) ra[ps] = *reinterpret_cast<const u32x4*>(wsrc[ps] + kofs);
        });
        const long tap = ((long)kr * a.W + ks) * a.ldx + kc;
        const bool cok = kc + lu * VEC < a.Cin;
        static_for<BPASS>([&](auto c) {
            constexpr int ps = decltype(c)::value;
            const bool ok = cok && (unsigned)(bhi[ps] + kr) < (unsigned)a.H && (unsigned)(bwi[ps] + ks) < (unsigned)a.W;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *reinterpret_cast<const u32x4*>(xg + bbase[ps] + tap);
            rb[ps] = v;
        });
    };
    auto advance = [&]() {
        kofs += BK;
        kc += BK;
        if (kc >= a.cin_k) {
            kc = 0;
            if (++ks == a.S) { ks = 0; ++kr; }
        }
    };
    auto stash = [&](int buf) {
        static_for<APASS>([&](auto c) {
            constexpr int ps = decltype(c)::value;
            const int row = lrow + ps * RPP;
            if (BM % RPP == 0 || row < BM) As[(buf * UNITS + lu) * BM + row] = ra[ps];
        });
        static_for<BPASS>([&](auto c) {
            constexpr int ps = decltype(c)::value;
            const int row = lrow + ps * RPP;
            if (BN % RPP == 0 || row < BN) Bs[(buf * UNITS + lu) * BN + row] = rb[ps];
        });
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = a.ktot / BK;
    load_step();
    stash(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            advance();
            load_step();
        }
        MmaStep<T, TM, TN, KU>::run(As + cur * UNITS * BM, Bs + cur * UNITS * BN, BM, BN, wm * TM * 16, wn * TN * 16, lane,
                                acc);
        if (more) stash(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds channels m..m+3 of pixel p for each (i, j) fragment
    const int mq = (lane >> 4) << 2, pc = lane & 15;
    OutT* const yg = reinterpret_cast<OutT*>(a.y);
    const T* const rg = reinterpret_cast<const T*>(a.res);
    float st1[TM][4], st2[TM][4];   // BatchNorm batch statistics of this wave's outputs (training forward only)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) st1[i][e] = st2[i][e] = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const long p = p0 + wn * TN * 16 + j * 16 + pc;
        if (p >= a.P) continue;
        long opix = p;  // output pixel index in the (possibly upsampled) destination
        int wo2 = 0;
        if (a.ups == 2) {
            const int n = (int)(p / HoWo);
            const int rem = (int)(p - (long)n * HoWo);
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            wo2 = 2 * a.Wo;
            opix = ((long)n * 2 * a.Ho + 2 * ho) * wo2 + 2 * wo;
        } else if (a.ups == 3) {  // phase scatter (stride-2 data gradient): every other pixel of a y_h x y_w tensor
            const int n = (int)(p / HoWo);
            const int rem = (int)(p - (long)n * HoWo);
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            opix = ((long)n * a.y_h + 2 * ho + a.y_off_h) * a.y_w + 2 * wo + a.y_off_w;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * TM * 16 + i * 16 + mq;
            if (m >= a.Cout) continue;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + m);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = activate_rt<T>(acc[i][j][e] + bv[e], a.act, a.slope);
            if (a.stats_part != nullptr) {   // statistics of the values as stored (rounded to the output type)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float q = (float)(OutT)v[e];
                    st1[i][e] += q;
                    st2[i][e] = fmaf(q, q, st2[i][e]);
                }
            }
            if (rg != nullptr) {
                float r4[4];
                load4<T>(rg + (a.ups == 3 ? opix : p) * a.ldr + m, r4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += r4[e];
            }
            OutT* dst = yg + opix * a.ldy + m;
            store4<OutT>(dst, v[0], v[1], v[2], v[3]);
            if (a.ups == 2) {
                store4<OutT>(dst + a.ldy, v[0], v[1], v[2], v[3]);
                store4<OutT>(dst + (long)wo2 * a.ldy, v[0], v[1], v[2], v[3]);
                store4<OutT>(dst + (long)(wo2 + 1) * a.ldy, v[0], v[1], v[2], v[3]);
            }
        }
    }
    if (a.stats_part != nullptr) {
        // sum over the 16 pixel lanes (lane & 15) of every channel quad, then one row of partials per (pixel tile, wn)
        float* const row = a.stats_part + ((long)(p0 / BN) * WN + wn) * 2 * a.Cout;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // 16-lane row sums on the DPP path (row_shr 1, 2, 4, 8 with zero fill): lane 15 of each row ends with the total
                const float s1 = row16_sum(st1[i][e]), s2 = row16_sum(st2[i][e]);
                const int m = m0 + wm * TM * 16 + i * 16 + mq + e;
                if (pc == 15 && m < a.Cout) {
                    row[m] = s1;
                    row[a.Cout + m] = s2;
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// LDS-DMA variant: operand tiles go global -> LDS directly (global_load_lds_dwordx4), STAGES K steps deep.
//
// Why: the register-staged kernel above is latency bound (rocprofv3: waves parked 64 % of their cycles,
// MFMA pipe 17 % busy) — its prefetch distance is one K step and every staged byte costs VGPRs and a
// ds_write.  LDS-DMA needs neither, so the ring can be 3-4 stages deep at higher occupancy.
//
// An LDS-DMA wave instruction writes 64 lanes x 16 B to *consecutive* LDS cells (M0 base + lane*16), so
// the tile image is row-major here: row = 4 cells (one 64-byte K step of f16), 16 rows per instruction,
// which keeps the global side at 64 contiguous bytes per row.  Bank conflicts on the fragment reads are
// removed by permuting the 4 cells of a row with f[(row >> 2) & 3], f = {0,2,3,1}: the *source* unit a
// lane fetches is (lane & 3) ^ f, the reader looks unit u up at cell u ^ f (both sides, same involution).
// Out-of-image taps and channel tails read a 16-byte zero page instead of branching.
template <typename T, int TM, int TN> struct MmaStepRM;  // row-major swizzled image, 4 units per row
template <int TM, int TN> struct MmaStepRM<f16, TM, TN> {
    static __device__ __forceinline__ void run(const u32x4* As, const u32x4* Bs, int arow, int brow, int lane,
                                               f32x4 (&acc)[TM][TN]) {
        const int r = lane & 15;
        const int off = r * 4 + ((lane >> 4) ^ swz_f(r >> 2));
        f16x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            u32x4 v = As[(arow + i * 16) * 4 + off];
            a[i] = *reinterpret_cast<f16x8*>(&v);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            u32x4 v = Bs[(brow + j * 16) * 4 + off];
            b[j] = *reinterpret_cast<f16x8*>(&v);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
};
template <int TM, int TN> struct MmaStepRM<float, TM, TN> {
    static __device__ __forceinline__ void run(const u32x4* As, const u32x4* Bs, int arow, int brow, int lane,
                                               f32x4 (&acc)[TM][TN]) {
        const int r = lane & 15, kq = lane >> 4, fr = swz_f(r >> 2);
        const float* Af = reinterpret_cast<const float*>(As);
        const float* Bf = reinterpret_cast<const float*>(Bs);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int off = (r * 4 + (u ^ fr)) * 4 + kq;
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = Af[(arow + i * 16) * 16 + off];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bf[(brow + j * 16) * 16 + off];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
};

// ABL: ablation switch for profiling only (0 = normal, 1 = skip the LDS-DMA loads, 2 = skip the MFMAs)
template <int TM, int TN> struct MmaStepRM<int8_t, TM, TN> {
    static __device__ __forceinline__ void run(const u32x4* As, const u32x4* Bs, int arow, int brow, int lane,
                                               i32x4 (&acc)[TM][TN]) {
        const int r = lane & 15;
        const int off = r * 4 + ((lane >> 4) ^ swz_f(r >> 2));
        i32x4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            u32x4 v = As[(arow + i * 16) * 4 + off];
            a[i] = *reinterpret_cast<i32x4*>(&v);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            u32x4 v = Bs[(brow + j * 16) * 4 + off];
            b[j] = *reinterpret_cast<i32x4*>(&v);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i][j], 0, 0, 0);
    }
};

// two workgroups per CU are the design point of these kernels (73.7 KB of LDS for the 8-wave tiles): 8 waves -> 4 per SIMD, i.e. at
// most 128 registers; 4 waves -> 2 per SIMD
template <typename T, typename OutT, int BM, int BN, int WM, int WN, int STAGES, int ABL = 0>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN * 2) / 4) void conv_igemm_glds_kernel(const ConvArgs a) {
    constexpr int VEC = Prec<T>::VEC, BK = VEC * 4;
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int GA = BM / 16, GB = BN / 16;        // 16-row groups (one LDS-DMA instruction each) per tile
    constexpr int GPW = (GA + GB) / NW;               // groups per wave per K step
    constexpr int GAW = GA / NW;                      // of which weight groups
    static_assert(GA % NW == 0 && GB % NW == 0, "tile rows must split evenly over the waves");
    static_assert(STAGES == 3 || STAGES == 4, "3 or 4 stage ring");

    __shared__ u32x4 smem[STAGES * 4 * (BM + BN)];    // one array only: [stage][A rows*4 | B rows*4]
    constexpr int STAGE_CELLS = 4 * (BM + BN);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;

    int m_tile, p_tile;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, rr = nb & 7, xcd = bid & 7;
        const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
        p_tile = logical / a.m_tiles;
        m_tile = logical - p_tile * a.m_tiles;
    }
    const int m0 = m_tile * BM;
    const long p0 = (long)p_tile * BN;

    // loader: lane -> (row within its 16-row group, source unit)
    const int lrow = lane >> 2;
    const int lu = (lane & 3) ^ swz_f(lane >> 4);
    const T* const xg = reinterpret_cast<const T*>(a.x);
    const T* const zero = reinterpret_cast<const T*>(g_zero_page);

    const T* wsrc[GAW > 0 ? GAW : 1];
    static_for<GAW>([&](auto c) {
        constexpr int i = decltype(c)::value;
        const int row = min(m0 + (wave + i * NW) * 16 + lrow, a.m_pad - 1);  // tiles taller than the 128-row padding
        wsrc[i] = reinterpret_cast<const T*>(a.w) + (long)row * a.ktot + lu * VEC;
    });
    constexpr int GBW = GPW - GAW;
    long bbase[GBW];
    int bhi[GBW], bwi[GBW];
    const int HoWo = a.Ho * a.Wo;
    static_for<GBW>([&](auto c) {
        constexpr int i = decltype(c)::value;
        const int row = (wave + i * NW) * 16 + lrow;
        const long p = p0 + row;
        if (p < a.P) {
            const int n = (int)(p / HoWo);
            const int rem = (int)(p - (long)n * HoWo);
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            bhi[i] = ho * a.stride - a.pad;
            bwi[i] = wo * a.stride - a.pad;
            bbase[i] = (((long)n * a.H + bhi[i]) * a.W + bwi[i]) * a.ldx + lu * VEC;
        } else {
            bhi[i] = -(1 << 28);
            bwi[i] = -(1 << 28);
            bbase[i] = 0;
        }
    });

    int kr = 0, ks = 0, kc = 0, kofs = 0;
    auto issue = [&](int st) {  // LDS-DMA of the K step (kr, ks, kc) into ring slot st, then advance the step
        u32x4* const base = smem + st * STAGE_CELLS;
        if constexpr (ABL != 1) static_for<GAW>([&](auto c) {
            constexpr int i = decltype(c)::value;
            u32x4* dst = base + (wave + i * NW) * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + kofs),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        });
        const long tap = ((long)kr * a.W + ks) * a.ldx + kc;
        const bool cok = kc + lu * VEC < a.Cin;
        if constexpr (ABL != 1) static_for<GBW>([&](auto c) {
            constexpr int i = decltype(c)::value;
            const bool ok = cok && (unsigned)(bhi[i] + kr) < (unsigned)a.H && (unsigned)(bwi[i] + ks) < (unsigned)a.W;
            const T* src = ok ? xg + bbase[i] + tap : zero;
            u32x4* dst = base + 4 * BM + (wave + i * NW) * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        });
        kofs += BK;
        kc += BK;
        if (kc >= a.cin_k) {
            kc = 0;
            if (++ks == a.S) { ks = 0; ++kr; }
        }
    };

    typedef typename AccOf<T>::type acc_t;
    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

    const int nk = a.ktot / BK;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) issue(s);
    int st_read = 0, st_write = STAGES - 1;
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's share of step kt has landed once at most `ahead` later steps are still in flight
        const int ahead = min(STAGES - 2, nk - 1 - kt);
        if (ahead >= 2) wait_vmcnt<2 * GPW>();
        else if (ahead == 1) wait_vmcnt<GPW>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();  // everyone's share landed; everyone is done reading slot st_write
        if (kt + STAGES - 1 < nk) issue(st_write);
        const u32x4* As = smem + st_read * STAGE_CELLS;
        if constexpr (ABL != 2) MmaStepRM<T, TM, TN>::run(As, As + 4 * BM, wm * TM * 16, wn * TN * 16, lane, acc);
        st_read = st_read + 1 == STAGES ? 0 : st_read + 1;
        st_write = st_write + 1 == STAGES ? 0 : st_write + 1;
    }

    conv_epilogue<T, OutT, TM, TN, BN, WN>(a, acc, m0, p0, wm, wn, lane);
}

template <typename T, typename OutT, int BM, int BN, int WM, int WN, int STAGES, int ABL = 0>
static int launch_glds(const ConvArgs& a0, hipStream_t stream) {
    ConvArgs a = a0;
    a.m_tiles = (a.Cout + BM - 1) / BM;
    a.p_tiles = (int)((a.P + BN - 1) / BN);
    const long blocks = (long)a.m_tiles * a.p_tiles;
    if (blocks <= 0 || blocks > 0x7fffffffL) return YH_EINVAL;
    hipLaunchKernelGGL((conv_igemm_glds_kernel<T, OutT, BM, BN, WM, WN, STAGES, ABL>), dim3((unsigned)blocks),
                       dim3(WM * WN * 64), 0, stream, a);
    return check_launch();
}


// ---------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 "halo" kernel: the im2col gather is done from LDS instead of from L2.
//
// rocprofv3 on the LDS-DMA im2col kernels: L2 (TCC) busy 80 %, MFMA pipe ~30 % — every activation row is
// fetched from L2 once per filter tap (9x).  Here a workgroup stages, once per 32-channel chunk, the rows of
// all input pixels its 256 output pixels touch (the tile plus one image row and one pixel on either side)
// and produces the nine shifted operand tiles as *views* of that LDS image: global traffic per K step drops
// from BM+256 rows to BM + ~(256 + 2W)/9 rows.
//
// To make every tap a constant row offset, output pixels are indexed in a virtual space with padding built
// in: image n, row y, column x  <->  v = n*(H+1)*(W+2) + (y+1)*(W+2) + (x+1); one zero row is shared between
// consecutive images, two zero columns end every row.  Output tile = 256 consecutive v; its inputs are the
// consecutive range [v0 - (W+2) - 1, v0 + 256 + (W+2) + 1); tap (r, s) of output v sits (r*(W+2) + s) rows
// further into the staged image.  Virtual positions that are padding are loaded from the zero page and their
// (garbage) outputs are not stored: (H+1)(W+2)/(HW) - 1 of the MFMA work is wasted (4 % at 76x76, 16 % at 19x19).
//
// Pipeline per K step (one tap of one chunk): weights stream through a 3-deep LDS-DMA ring exactly like the
// im2col kernel; the next chunk's halo image is fetched into the second halo buffer right after tap 0.
template <typename T, typename OutT, int BM, int LB>
__global__ __launch_bounds__(512, (BM <= 128 ? 4 : 2)) void conv3x3_halo_kernel(const ConvArgs a, const int rows_h) {
    constexpr int VEC = Prec<T>::VEC, BK = VEC * 4, BN = 256, NW = 8, WN = 4, STAGES = 3;
    constexpr int WM = NW / WN;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int GAW = BM / 16 / NW;  // weight groups (16 rows) per wave per K step
    static_assert(BM % (16 * NW) == 0, "weight tile must split over 8 waves");
    constexpr int LA = GAW;

    extern __shared__ __attribute__((aligned(16))) u32x4 dsm[];
    u32x4* const Aring = dsm;                                  // [STAGES][BM*4]
    u32x4* const Bbuf0 = dsm + STAGES * BM * 4;                 // [2][rows_h*4]
    u32x4* const dummy = Bbuf0 + 2 * rows_h * 4;                // [64] sink for padding loads

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;

    int m_tile, p_tile;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, rr = nb & 7, xcd = bid & 7;
        const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
        p_tile = logical / a.m_tiles;
        m_tile = logical - p_tile * a.m_tiles;
    }
    const int m0 = m_tile * BM;
    const long q0 = (long)p_tile * BN;  // first virtual output position of this tile
    const int Wp = a.W + 2;
    const long IMG = (long)(a.H + 1) * Wp;

    const int lrow = lane >> 2;
    const int lu = (lane & 3) ^ (((lane >> 4) & 1) << 1);  // source unit of this lane (swizzle f = 2*((row>>2)&1))
    const T* const xg = reinterpret_cast<const T*>(a.x);
    const T* const zero = reinterpret_cast<const T*>(g_zero_page);

    const T* wsrc[GAW];
    static_for<GAW>([&](auto c) {
        constexpr int i = decltype(c)::value;
        const int row = min(m0 + (wave + i * NW) * 16 + lrow, a.m_pad - 1);
        wsrc[i] = reinterpret_cast<const T*>(a.w) + (long)row * a.ktot + lu * VEC;
    });
    // halo rows owned by this lane: group g = wave + i*NW, row j = g*16 + lrow, input position v = v0 + j
    const long v0 = q0 - Wp - 1;
    const T* bsrc[LB];
    static_for<LB>([&](auto c) {
        constexpr int i = decltype(c)::value;
        const int j = (wave + i * NW) * 16 + lrow;
        const long v = v0 + j;
        const T* ptr = nullptr;
        if (j < rows_h && v >= 0) {
            const long n = v / IMG;
            const int rem = (int)(v - n * IMG);
            const int yy = rem / Wp - 1, xx = rem - (rem / Wp) * Wp - 1;
            if (n < a.N && yy >= 0 && xx >= 0 && xx < a.W)  // yy < H always: IMG has H+1 rows, row 0 is the pad row
                ptr = xg + ((n * a.H + yy) * (long)a.W + xx) * a.ldx + lu * VEC;
        }
        bsrc[i] = ptr;
    });

    auto issue_a = [&](int st, int tap, int kc) {
        static_for<GAW>([&](auto c) {
            constexpr int i = decltype(c)::value;
            u32x4* dst = Aring + st * (BM * 4) + (wave + i * NW) * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + tap * a.cin_k + kc),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        });
    };
    auto issue_b = [&](int buf, int kc) {
        const bool cok = kc + lu * VEC < a.Cin;
        static_for<LB>([&](auto c) {
            constexpr int i = decltype(c)::value;
            const int g = wave + i * NW;  // wave-uniform
            const T* src = (cok && bsrc[i] != nullptr) ? bsrc[i] + kc : zero;
            u32x4* dst = (g * 16 < rows_h) ? Bbuf0 + buf * (rows_h * 4) + g * 64 : dummy;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        });
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.cin_k / BK;
    const int nk = 9 * nchunks;
    // prologue: halo image of chunk 0, then weights of steps 0 and 1
    issue_b(0, 0);
    issue_a(0, 0, 0);
    issue_a(1, 1, 0);  // nk >= 9 always
    int tap = 0, chunk = 0;        // the step being computed
    int ptap = 2, pkc = 0;         // tap / channel offset of the next weight tile to prefetch (step k+2)
    int st_read = 0, st_write = 2;
    const int r16 = lane & 15;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more_a = kt + 1 < nk;                          // A(k+1) still in flight behind A(k)
        const bool b_behind = (tap == 1 || tap == 2) && chunk + 1 < nchunks;  // next halo image issued after A(k)
        if (more_a) { if (b_behind) wait_vmcnt<LA + LB>(); else wait_vmcnt<LA>(); }
        else        { if (b_behind) wait_vmcnt<LB>(); else wait_vmcnt<0>(); }
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue_a(st_write, ptap, pkc);
        if (tap == 0 && chunk + 1 < nchunks) issue_b((chunk + 1) & 1, (chunk + 1) * BK);
        if (++ptap == 9) { ptap = 0; pkc += BK; }

        // ---- MFMAs of this tap: weights from the ring slot, activations = halo image shifted by the tap
        {
            const u32x4* As = Aring + st_read * (BM * 4);
            const u32x4* Bs = Bbuf0 + (chunk & 1) * (rows_h * 4);
            const int tapoff = (tap / 3) * Wp + (tap % 3);
            if constexpr (sizeof(T) == 2) {
                const int u = lane >> 4;
                f16x8 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm * TM * 16 + i * 16 + r16;
                    u32x4 v = As[row * 4 + (u ^ (((row >> 2) & 1) << 1))];
                    fa[i] = *reinterpret_cast<f16x8*>(&v);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = wn * TN * 16 + j * 16 + r16 + tapoff;
                    u32x4 v = Bs[row * 4 + (u ^ (((row >> 2) & 1) << 1))];
                    fb[j] = *reinterpret_cast<f16x8*>(&v);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            } else {
                const int kq = lane >> 4;
                const float* Af = reinterpret_cast<const float*>(As);
                const float* Bf = reinterpret_cast<const float*>(Bs);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float fa[TM], fb[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int row = wm * TM * 16 + i * 16 + r16;
                        fa[i] = Af[(row * 4 + (u ^ (((row >> 2) & 1) << 1))) * 4 + kq];
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int row = wn * TN * 16 + j * 16 + r16 + tapoff;
                        fb[j] = Bf[(row * 4 + (u ^ (((row >> 2) & 1) << 1))) * 4 + kq];
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        st_read = st_read + 1 == STAGES ? 0 : st_read + 1;
        st_write = st_write + 1 == STAGES ? 0 : st_write + 1;
        if (++tap == 9) { tap = 0; ++chunk; }
    }

    // ---- epilogue: virtual position -> real pixel (padding positions are dropped).  Bias and the residual column are loaded
    // BEFORE the stores they precede (one in-order memory counter: a load behind a store waits for the store's acknowledgement)
    const int mq = (lane >> 4) << 2;
    OutT* const yg = reinterpret_cast<OutT*>(a.y);
    const T* const rg = reinterpret_cast<const T*>(a.res);
    f32x4 bvs[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * TM * 16 + i * 16 + mq;
        bvs[i] = m < a.Cout ? *reinterpret_cast<const f32x4*>(a.bias + m) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(bvs[i]));
    typedef typename ResVec<T>::type res_t;
    static_for<TN>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const long q = q0 + wn * TN * 16 + j * 16 + r16;
        const long n = q / IMG;
        const int rem = (int)(q - n * IMG);
        const int yy = rem / Wp - 1, xx = rem - (rem / Wp) * Wp - 1;
        const bool ok = !(n >= a.N || yy < 0 || xx < 0 || xx >= a.W);
        const long p = (n * a.H + yy) * (long)a.W + xx;
        long opix = p;
        int wo2 = 0;
        if (a.ups == 2) {
            wo2 = 2 * a.W;
            opix = (n * 2 * a.H + 2 * yy) * (long)wo2 + 2 * xx;
        }
        res_t rv[TM];
        const bool have_res = rg != nullptr;
        if (have_res) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm * TM * 16 + i * 16 + mq;
                rv[i] = res_t{};
                if (ok && m < a.Cout) rv[i] = *reinterpret_cast<const res_t*>(rg + p * a.ldr + m);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                res_t t = rv[i];
                asm volatile("" : "+v"(t));
                rv[i] = t;
            }
        }
        if (!ok) return;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * TM * 16 + i * 16 + mq;
            if (m >= a.Cout) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = activate_rt<T>(acc[i][j][e] + bvs[i][e], a.act, a.slope);
            if (have_res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rv[i][e];
            }
            OutT* dst = yg + opix * a.ldy + m;
            store4<OutT>(dst, v[0], v[1], v[2], v[3]);
            if (a.ups == 2) {
                store4<OutT>(dst + a.ldy, v[0], v[1], v[2], v[3]);
                store4<OutT>(dst + (long)wo2 * a.ldy, v[0], v[1], v[2], v[3]);
                store4<OutT>(dst + (long)(wo2 + 1) * a.ldy, v[0], v[1], v[2], v[3]);
            }
        }
    });
}

template <typename T, typename OutT, int BM> static int launch_halo(const ConvArgs& a0, hipStream_t stream) {
    ConvArgs a = a0;
    if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1) return YH_EUNSUPPORTED;
    const int Wp = a.W + 2;
    const int rows_h = ((256 + 2 * Wp + 2 + 15) / 16) * 16;
    const int groups = rows_h / 16;
    const int lb = (groups + 7) / 8;
    if (lb > 5) return YH_EUNSUPPORTED;  // halo image too large for LDS (W > ~190)
    a.m_tiles = (a.Cout + BM - 1) / BM;
    const long Q = (long)a.N * (a.H + 1) * Wp;
    a.p_tiles = (int)((Q + 255) / 256);
    const long blocks = (long)a.m_tiles * a.p_tiles;
    if (blocks <= 0 || blocks > 0x7fffffffL) return YH_EINVAL;
    const size_t lds = ((size_t)3 * BM * 4 + (size_t)2 * rows_h * 4 + 64) * 16;
    if (lds > 160 * 1024) return YH_EUNSUPPORTED;
#define YH_HALO_CASE(LBV)                                                                                             \
    case LBV: {                                                                                                       \
        auto kern = conv3x3_halo_kernel<T, OutT, BM, LBV>;                                                            \
        {                                                                                                             \
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);                                  \
            if (e != hipSuccess) return (int)e;                                                                       \
        }                                                                                                             \
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, stream, a, rows_h);                          \
        break;                                                                                                        \
    }
    switch (lb) {
        YH_HALO_CASE(1) YH_HALO_CASE(2) YH_HALO_CASE(3) YH_HALO_CASE(4) YH_HALO_CASE(5)
        default: return YH_EUNSUPPORTED;
    }
#undef YH_HALO_CASE
    return check_launch();
}

template <typename T, typename OutT, int BM, int BN, int WM, int WN, int KU>
static int launch(const ConvArgs& a0, hipStream_t stream) {
    ConvArgs a = a0;
    if (a.cin_k % (Prec<T>::VEC * KU)) return YH_EALIGN;
    a.m_tiles = (a.Cout + BM - 1) / BM;
    a.p_tiles = (int)((a.P + BN - 1) / BN);
    const long blocks = (long)a.m_tiles * a.p_tiles;
    if (blocks <= 0 || blocks > 0x7fffffffL) return YH_EINVAL;
    hipLaunchKernelGGL((conv_igemm_kernel<T, OutT, BM, BN, WM, WN, KU>), dim3((unsigned)blocks), dim3(WM * WN * 64), 0, stream,
                       a);
    return check_launch();
}

// Tile codes (channels x pixels).  Register-staged kernels: 1 = 128x128, 2 = 64x256, 3 = 32x256, 4 = 64x128,
// 5 = 128x64, 6 = 256x128 (8 waves); +10 = 8-unit K step.  LDS-DMA ring kernels: 21 = 128x128, 22 = 64x256,
// 24 = 64x128, 25 = 128x64, 26 = 256x128 (8 waves), 27 = 128x256 (8 waves) with 3 stages; 3x = 4 stages.
// The automatic choice below was fitted to per-layer measurements on MI355X (profiles/r01_tile_ab.txt):
// the kernels are bound by global->LDS delivery, so the tile with the most FLOP per staged byte wins as
// long as it still yields >= ~1 workgroup wave over the 256 CUs.
static int pick_tile(int cout, long P, int cin_k, int vec, int taps = 1) {
    (void)vec;
    const int c = cout;
    const long w256 = ((c + 255) / 256) * 256, w128 = ((c + 127) / 128) * 128, w64 = ((c + 63) / 64) * 64;
    auto blocks = [&](int bm, int bn) { return (long)((c + bm - 1) / bm) * ((P + bn - 1) / bn); };
    // Round 6 (profiles/r06_ring_tile_sweep.txt, sweeps on YOLOv3 / YOLOv4 / YOLOv3-Mobilenetv3 with the final kernels):
    //  * the 32-row register-staged tile only for <= 32 outputs over a K of one step.  The round-1 rule also sent every width that pads least on
    //    32 rows (72, 80, 160, 184, 200, 240, 480, 672, 960 ... : Mobilenetv3) there, where ANY ring tile is 1.3 - 2 x faster (26^2 112 -> 672:
    //    0.052 -> 0.028 ms), and <= 32 outputs over 64 - 72 channels (104^2 64 -> 24: 0.046 -> 0.033 on the 64 x 128 ring tile)
    //  * 256 x 128 / 128 x 256 only from two workgroups per CU on; below, 128 x 128 (38^2 256 -> 256 at batch 32: 0.0253 -> 0.0232 ms on nine
    //    layers, the data gradients 0.0284 -> 0.0255; 19^2 1024 -> 512 at batch 64: 0.046 -> 0.044)
    if (c <= 32) return (long)cin_k * taps >= 64 ? 24 : 3;
    if (w128 <= w64) {
        if (w256 == w128 && blocks(256, 128) >= 512) return 26;
        // (one or two K steps - YOLOv4's 80^2 128 -> 128 1x1 layers - are all prologue and epilogue: the narrower tile's extra workgroups overlap
        // them better, 0.037 -> 0.033 ms on eleven layers)
        if (blocks(128, 256) >= 512) return (long)cin_k * taps <= 128 ? 21 : 27;
        if (blocks(128, 128) >= 256) return 21;
        return 25;
    }
    return 24;      // 64-row tiles pad less
}

template <typename T, typename OutT> static int dispatch_tile(const ConvArgs& a, int tile, hipStream_t s) {
    if (tile == 0) tile = pick_tile(a.Cout, a.P, a.cin_k, Prec<T>::VEC, a.R * a.S);
    switch (tile) {
        case 1: return launch<T, OutT, 128, 128, 2, 2, 4>(a, s);
        case 2: return launch<T, OutT, 64, 256, 1, 4, 4>(a, s);
        case 3: return launch<T, OutT, 32, 256, 1, 4, 4>(a, s);
        case 4: return launch<T, OutT, 64, 128, 2, 2, 4>(a, s);
        case 5: return launch<T, OutT, 128, 64, 2, 2, 4>(a, s);
        case 6: return launch<T, OutT, 256, 128, 4, 2, 4>(a, s);
        case 11: return launch<T, OutT, 128, 128, 2, 2, 8>(a, s);
        case 12: return launch<T, OutT, 64, 256, 1, 4, 8>(a, s);
        case 14: return launch<T, OutT, 64, 128, 2, 2, 8>(a, s);
        case 15: return launch<T, OutT, 128, 64, 2, 2, 8>(a, s);
        case 16: return launch<T, OutT, 256, 128, 4, 2, 8>(a, s);
        // LDS-DMA ring kernels: 2x = 3 stages, 3x = 4 stages
        case 21: return launch_glds<T, OutT, 128, 128, 2, 2, 3>(a, s);
        case 22: return launch_glds<T, OutT, 64, 256, 1, 4, 3>(a, s);
        case 24: return launch_glds<T, OutT, 64, 128, 2, 2, 3>(a, s);
        case 25: return launch_glds<T, OutT, 128, 64, 2, 2, 3>(a, s);
        case 26: return launch_glds<T, OutT, 256, 128, 4, 2, 3>(a, s);
        case 27: return launch_glds<T, OutT, 128, 256, 2, 4, 3>(a, s);
        // (the same tiles on 4 waves, 128 x 64 per wave, measured 10-20 % slower here - 8 waves per CU do not hide the
        // per-step barrier + LDS latency without intra-wave pipelining - although that shape wins in the wgrad kernel)
        case 51: return launch_glds<T, OutT, 256, 128, 4, 2, 3, 1>(a, s);  // ablation: no loads (results are garbage)
        case 52: return launch_glds<T, OutT, 256, 128, 4, 2, 3, 2>(a, s);  // ablation: no MFMAs (results are garbage)
        case 41: return launch_halo<T, OutT, 128>(a, s);   // 3x3 s1 halo kernel, 128 channels x 256 virtual pixels
        case 42: return launch_halo<T, OutT, 256>(a, s);
        case 43:   // halo ping-pong kernel (conv_halo_pp.hip): f16 in, f16 out
            if constexpr (sizeof(T) == 2 && std::is_same<OutT, T>::value) return launch_hpp_tile(a, YH_F16, s);
            else return YH_EUNSUPPORTED;
        case 31: return launch_glds<T, OutT, 128, 128, 2, 2, 4>(a, s);
        // full-line K step (64 f16 channels), 128 x 64 per wave: 61 = 256 x 256, 62 = 128 x 512 (8 waves, 2 stages),
        // 63 = 256 x 128 (4 waves, 3 stages)
        case 61: case 62: case 63: case 64: case 65: case 66: case 67: case 68: case 69:
            if constexpr (sizeof(T) == 2) return launch_k64_tile(a, tile, YH_F16, std::is_same<OutT, float>::value ? 1 : 0, s);
            else return YH_EINVAL;
        case 32: return launch_glds<T, OutT, 64, 256, 1, 4, 4>(a, s);
        case 34: return launch_glds<T, OutT, 64, 128, 2, 2, 4>(a, s);
        case 35: return launch_glds<T, OutT, 128, 64, 2, 2, 4>(a, s);
        default: return YH_EINVAL;
    }
}

template <typename OutT> static int dispatch_tile_i8(const ConvArgs& a, int tile, hipStream_t s) {
    if (tile == 0) {
        tile = pick_tile(a.Cout, a.P, a.cin_k, 16);
        if (tile == 3) tile = 24;  // no 32-channel LDS-DMA tile: use 64x128
    }
    switch (tile) {
        case 21: return launch_glds<int8_t, OutT, 128, 128, 2, 2, 3>(a, s);
        case 24: return launch_glds<int8_t, OutT, 64, 128, 2, 2, 3>(a, s);
        case 25: return launch_glds<int8_t, OutT, 128, 64, 2, 2, 3>(a, s);
        case 26: return launch_glds<int8_t, OutT, 256, 128, 4, 2, 3>(a, s);
        case 27: return launch_glds<int8_t, OutT, 128, 256, 2, 4, 3>(a, s);
        case 64: case 65: case 66: return launch_k64_tile(a, tile, YH_I8, std::is_same<OutT, float>::value ? 1 : 0, s);
        case 43:
            if constexpr (std::is_same<OutT, int8_t>::value) return launch_hpp_tile(a, YH_I8, s);
            else return YH_EUNSUPPORTED;
        default: return YH_EINVAL;
    }
}

__global__ void pack_qconv_weights_kernel(const float* __restrict__ qw, float inv_scale, const int32_t* __restrict__ cin_map,
                                          int cout, int cin, int taps, int cin_k, int8_t* __restrict__ packed) {
    const long total = (long)cout * cin * taps;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const long r = i / taps;
        const int ci = (int)(r % cin), co = (int)(r / cin);
        const int pc = cin_map ? cin_map[ci] : ci;
        packed[((long)co * taps + tap) * cin_k + pc] = (int8_t)(int)round_clamp_i8(qw[i] * inv_scale);
    }
}

}  // namespace yh

extern "C" int yh_qconv_pack_weights(const float* q_weight, float w_scale, const int32_t* cin_map, int cout, int cin, int kh,
                                     int kw, int cin_k, int m_pad, void* packed, void* stream) {
    using namespace yh;
    if (!q_weight || !packed || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 || !(w_scale > 0.f)) return YH_EINVAL;
    if (cin_k % 64 || m_pad % 128 || m_pad < cout) return YH_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(packed, 0, (size_t)m_pad * kh * kw * cin_k, s);
    if (e != hipSuccess) return (int)e;
    const long total = (long)cout * cin * kh * kw;
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_qconv_weights_kernel, dim3((unsigned)g), dim3(256), 0, s, q_weight, 1.f / w_scale, cin_map, cout, cin,
                       kh * kw, cin_k, (int8_t*)packed);
    return check_launch();
}

// The ping-pong kernels (conv_igemm_k64.hip) against the ring kernels, by a cost model fitted to per-layer A/B runs on MI355X
// (profiles/r02_conv_tile_ab.txt, in-kernel stamps profiles/r02_pp_timing.txt):
//   ping-pong: one 8-wave workgroup per CU; per workgroup ~20 000 cycles of set-up, first-tile latency and epilogue plus 3 300 per
//              64-channel K tile (the global -> LDS delivery of a CU saturates at ~20 B/clk: 64 KB per K tile of a 256 x 256 tile);
//   ring:      ~420 kFLOP per cycle chip-wide on 3x3 layers (880 TFLOP/s at 2.1 GHz), two workgroups per CU hiding each other's
//              prologue and epilogue.
// Long K (3x3 over >= 256 channels) with a workgroup count that fills whole rounds of 256 CUs goes to ping-pong; 1x1 layers
// (4 - 16 K tiles: the fixed cost dominates) and everything else stay on the ring kernels.
static int pick_pp_tile(const yh_conv_desc* d) {
    const int line = d->dtype == YH_I8 ? 128 : 64;          // channels per 128-byte line
    if ((d->dtype != YH_F16 && d->dtype != YH_I8) || d->ups == 4 || d->cin != d->cin_k || d->cin_k % line) return 0;
    const int taps = d->kh * d->kw;
    if (taps < 4 || d->cout < 256) return 0;
    const long P = (long)d->n * d->ho * d->wo;
    // 32-bit element offsets inside the kernel
    if ((long)d->n * d->h * d->w_in * d->ldx + (long)(d->kh + 1) * d->w_in * d->ldx >= 0x7fffffffL) return 0;
    if ((long)d->m_pad * taps * d->cin_k >= 0x7fffffffL || d->h >= 32768 || d->w_in >= 32768) return 0;
    const double flops = 2.0 * (double)P * d->cout * taps * d->cin;
    const double ring = flops / (d->dtype == YH_I8 ? 760e3 : 420e3);     // int8 ring kernels: ~1.6 POP/s on 3x3 layers
    const long nk = (long)taps * d->cin_k / line;
    auto pp = [&](int bm, int bn) {
        const long blocks = (long)((d->cout + bm - 1) / bm) * ((P + bn - 1) / bn);
        return (double)((blocks + 255) / 256) * (20000.0 + 3300.0 * nk);
    };
    const double c64 = pp(256, 256), c66 = d->cout >= 512 ? pp(512, 128) : 1e30;
    const double best = c64 < c66 ? c64 : c66;
    if (best > 0.95 * ring) return 0;
    return c66 <= c64 ? 66 : 64;
}

// The 3x3 halo kernel (activations fetched once per 32-channel chunk instead of once per tap) against the ring / ping-pong kernels
// after the epilogue fix (profiles/r02_conv_tile_ab.txt): ahead only on the 76 x 76 layers with 128- or 256-row weight tiles -
// data gradient 256 -> 128: 936 vs 809 TFLOP/s, forward 128 -> 256: 894 vs 850 - behind on the 152, 38 and 19 grids.  It has no
// statistics epilogue, so the training forward never takes it.
static bool pick_halo_tile(const yh_conv_desc* d) {
    if (d->dtype != YH_F16 || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->stats_ws || d->ups > 2) return false;
    if (d->cout != 128 && d->cout != 256) return false;
    return d->cin >= 128 && d->cin % 32 == 0 && d->w_in >= 48 && d->w_in <= 96 && (long)d->n * d->h * d->w_in >= 131072;
}

// The halo ping-pong kernel (conv_halo_pp.hip, tile 43): 3x3 / s1 / p1 with one of the compile-time activations, plain dense store
// (residual and fused statistics included), input and output of the same type.  128-row weight tiles: the weight tile is re-read
// by every pixel tile, so it takes the layers whose pixel axis is long enough to fill the chip with 512-pixel tiles.
static bool pick_hpp_tile(const yh_conv_desc* d) {
    static const bool off = getenv("YH_NO_HPP") != nullptr;
    if (off || (d->dtype != YH_F16 && d->dtype != YH_I8) || d->out_f32) return false;
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->ups != 1) return false;
    if (d->act != YH_ACT_LINEAR && d->act != YH_ACT_LEAKY && d->act != YH_ACT_MISH) return false;
    const int bk = d->dtype == YH_I8 ? 64 : 32;
    // measured (profiles/r03_hpp_sweep.txt, batch 64): ahead by 15 - 35 % on the 76 x 76 / 38 x 38 / 19 x 19 layers, by 5 - 13 % on
    // the 152 x 152 layers (64 -> 128, and the data gradient 128 -> 64 with a half-empty weight tile); behind only with a single
    // channel chunk (304 x 304 32 -> 64: 9 K steps do not amortise the halo prologue)
    if (d->cin_k % bk || d->cout < 64 || d->cin_k / bk < 2) return false;
    // fp16, a half-empty 128-row weight tile AND only two channel chunks (YOLOv4's 152^2 / 160^2 64 -> 64 layers): the 64 x 128 ring tile is ahead,
    // 0.204 -> 0.147 ms in detection, 0.170 -> 0.119 in the training forward (128 -> 64, four chunks, stays: 5 - 13 % ahead here, round 3)
    if (d->dtype == YH_F16 && d->cout <= 64 && d->cin_k / bk == 2) return false;
    // int8 with exactly two channel chunks (128 input channels: 18 K steps): the 128 x 256 ring tile is ahead when the layer carries the fused
    // quantised shortcut (YOLOv3-608 b64 76^2 128 -> 256 +res: 0.188 against 0.170 ms on eight layers; without the shortcut 0.138 on this kernel) or is
    // small (YOLOv4-640 b32 80^2: 0.075 / 0.092 against 0.068 / 0.086) - profiles/r06_ring_tile_sweep.txt
    if (d->dtype == YH_I8 && d->cin_k / bk == 2 && (d->res || (long)d->n * d->ho * d->wo < 262144)) return false;
    const unsigned amask = d->dtype == YH_F16 ? 15u : 7u;     // 8-channel stores / residual loads
    if (d->cout % 8 || d->ldy % 8 || (((uintptr_t)d->y) & amask) || (d->res && (d->ldr % 8 || (((uintptr_t)d->res) & amask)))) return false;
    if ((long)(d->n + 1) * (d->h + 1) * (d->w_in + 1) + 4096 >= 0x7fffffffL) return false;
    int rows_hp, lb, hbufs;
    size_t lds;
    if (!yh::hpp_geometry(d->w_in, d->cin_k, bk, &rows_hp, &lb, &hbufs, &lds)) return false;
    if ((long)d->n * d->h * d->w_in * d->ldx + d->cin_k >= 0x7fffffffL || (long)d->m_pad * 9 * d->cin_k >= 0x7fffffffL) return false;
    const long blocks = (long)((d->cout + 127) / 128) * (((long)d->n * (d->h + 1) * (d->w_in + 1) + 511) / 512);
    // one workgroup per CU: a partly filled last round of the 256 CUs is paid in full
    const long rounds = (blocks + 255) / 256;
    // (38 x 38 512 -> 256 at batch 64 fills 75 % of two rounds and is still 30 % ahead of the ring / ping-pong kernels)
    return blocks >= 192 && (rounds >= 4 || blocks * 100 >= rounds * 256 * 60);
}

// The LDS-free streaming kernel (conv_pointwise.hip) for 1x1 convolutions over few channels on large grids, where a ring-kernel
// tile lives for one or two K steps: whole-K weights in registers, (MT, KS) = (16-row channel groups, MFMA K steps) within its
// register budget, plain dense store without residual / statistics / upsample.
static bool pick_pointwise_tile(const yh_conv_desc* d) {
    static const bool off = getenv("YH_NO_POINTWISE") != nullptr;
    if (off || (d->dtype != YH_F16 && d->dtype != YH_I8)) return false;
    if (d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->ups != 1 || d->res || d->stats_ws) return false;
    if (d->act != YH_ACT_LINEAR && d->act != YH_ACT_LEAKY && d->act != YH_ACT_MISH) return false;
    const int k = d->dtype == YH_I8 ? 64 : 32;
    if (d->cin != d->cin_k || d->cin_k % k || d->cout > 128) return false;
    const int mt = d->cout <= 32 ? 2 : (d->cout <= 64 ? 4 : 8), ks = d->cin_k / k;
    if (!((mt == 2 && (ks == 1 || ks == 2 || ks == 4)) || (mt == 4 && (ks == 1 || ks == 2)) || (mt == 8 && ks == 1))) return false;
    return (long)d->n * d->ho * d->wo >= 262144;     // the high-resolution stages; smaller grids keep the ring kernels
}

// The streaming 3x3 kernel (conv_stream3.hip): one MFMA K step of input channels, at most 64 output channels, large grids; plain
// dense fp16 / int8 store with or without residual (or fused quantised shortcut), or with BatchNorm statistics; no upsample / fp32 output.
static bool stream3_supported(const yh_conv_desc* d) {
    if (d->dtype != YH_F16 && d->dtype != YH_I8) return false;
    if (d->kh != 3 || d->kw != 3 || d->pad != 1 || (d->stride != 1 && d->stride != 2) || d->ups != 1 || d->out_f32) return false;
    if (d->stats_ws && (d->dtype != YH_F16 || d->res)) return false;
    if (d->act != YH_ACT_LINEAR && d->act != YH_ACT_LEAKY && d->act != YH_ACT_MISH) return false;
    const bool two_steps = d->dtype == YH_F16 && d->cin_k == 64 && d->cout == 32 && !d->stats_ws;     // 64 -> 32: conv3's data gradient
    if (d->cin_k != (d->dtype == YH_I8 ? 64 : 32) && !two_steps) return false;
    if (d->cout != 32 && d->cout != 64 && !(d->cout == 128 && d->dtype == YH_I8)) return false;
    const int esz = d->dtype == YH_I8 ? 1 : 2;
    if ((d->ldy * esz) % 16 || (reinterpret_cast<uintptr_t>(d->y) & 15u)) return false;                    // whole 16-byte units per row
    if ((long)d->n * d->ho * d->wo >= (1L << 31)) return false;                                              // 32-bit pixel index
    return (long)d->n * d->h * d->w_in * d->ldx * esz < (1L << 31);                                          // 32-bit tap offsets
}
static bool pick_stream3_tile(const yh_conv_desc* d) {
    static const bool off = getenv("YH_NO_STREAM3") != nullptr;
    // The two-K-step form (fp16 64 -> 32, round 4) is an explicit-tile A/B form only: 0.77 ms on conv3's data gradient against 0.72 ms
    // for the 64 x 128 ring tile - with 18 fragment loads per pixel group the nine-fold tap re-reads through L1 / TA (9 x 757 MB) bound
    // it, not HBM (profiles/r04_stream3_two_steps.txt)
    if (d->dtype == YH_F16 && d->cin_k != 32) return false;
    return !off && stream3_supported(d) && (long)d->n * d->ho * d->wo >= 262144;
}

// The persistent LDS-resident-weights 1x1 kernel (conv_pw_lds.hip): byte-bound 1x1 layers on large grids, forward (with BatchNorm
// statistics) and data gradient (with the residual accumulate) alike.
static bool pwl_desc_supported(const yh_conv_desc* d) {
    if (d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->ups != 1) return false;
    if (d->act != YH_ACT_LINEAR && d->act != YH_ACT_LEAKY && d->act != YH_ACT_MISH) return false;
    const bool bwd = d->bwd_z != nullptr;      // backward sums of the block this data gradient completes: stats_ws then holds THEIR rows
    if (bwd && (d->act != YH_ACT_LINEAR || (d->bwd_act != YH_ACT_LEAKY && d->bwd_act != YH_ACT_MISH))) return false;
    return yh::pwl_supported(d->dtype, d->out_f32, d->cin, d->cin_k, d->cout, (long)d->n * d->ho * d->wo, d->ldx, d->ldy, d->ldr, d->x, d->y,
                             d->res, !bwd && d->stats_ws != nullptr, bwd);
}
static bool pick_pwl_tile(const yh_conv_desc* d) {
    static const bool off = getenv("YH_NO_PWL") != nullptr;
    static const long min_px = [] { const char* e = getenv("YH_PWL_MIN_PIXELS"); return e ? atol(e) : 262144L; }();      // A/B knob
    return !off && pwl_desc_supported(d) && (long)d->n * d->ho * d->wo >= min_px;
}

extern "C" int yh_conv2d_tile(const yh_conv_desc* d) {
    if (!d) return YH_EINVAL;
    if (d->tile != 0) return d->tile;
    if (pick_pointwise_tile(d)) return 71;
    if (pick_pwl_tile(d)) return 73;
    if (pick_stream3_tile(d)) return 72;
    if (pick_hpp_tile(d)) return 43;
    if (pick_halo_tile(d)) return 41;
    if (const int pp = pick_pp_tile(d)) return pp;
    const int t = yh::pick_tile(d->cout, (long)d->n * d->ho * d->wo, d->cin_k, d->dtype == YH_F16 ? 8 : 4, d->kh * d->kw);
    // a layer that stores its 2x upsampling (four stores per value) wants the small tile: more workgroups to overlap the store phase
    // (round 6, profiles/r06_ring_tile_sweep.txt: 19^2 512 -> 256 0.040 -> 0.029 ms, 38^2 256 -> 128 0.059 -> 0.048, YOLOv4's 40^2 0.040 -> 0.029)
    // (int8: 128 x 64 - 0.035 -> 0.024 and 0.045 -> 0.034 ms)
    if (d->ups == 2 && (t == 21 || t == 26 || t == 27 || t == 25)) {
        if (d->dtype == YH_F16) return 24;
        if (d->dtype == YH_I8) return 25;
    }
    // int8 layers with at most four K steps (1x1 from <= 256 channels: YOLOv4's 80^2 128 -> 128 and 256 -> 128): 128 x 64 (0.035 / 0.040 -> 0.029 / 0.032 ms)
    if (d->dtype == YH_I8 && (t == 21 || t == 27) && (long)d->cin_k * d->kh * d->kw <= 256) return 25;
    return (d->dtype == YH_I8 && t == 3) ? 24 : t;
}

// (pixel-tile width, wave columns) of a tile code: the geometry of the fused-statistics partial rows
static bool tile_geometry(int tile, int* bn, int* wn) {
    switch (tile) {
        case 1: case 11: case 21: case 31: *bn = 128; *wn = 2; return true;
        case 2: case 12: case 22: case 32: case 3: *bn = 256; *wn = 4; return true;
        case 4: case 14: case 24: case 34: *bn = 128; *wn = 2; return true;
        case 5: case 15: case 25: case 35: *bn = 64; *wn = 2; return true;
        case 6: case 16: case 26: case 51: case 52: *bn = 128; *wn = 2; return true;
        case 27: case 61: case 64: case 67: *bn = 256; *wn = 4; return true;
        case 62: case 65: case 68: *bn = 512; *wn = 8; return true;
        case 63: case 66: case 69: *bn = 128; *wn = 2; return true;
        default: return false;
    }
}

// rows of [sum g | sum g xhat][cout] a launch with bwd_z leaves in stats_ws; 0: this launch cannot carry the block's backward sums
// (only the persistent 1x1 kernel can: fp16, the data-gradient shapes of conv_pw_lds.hip, >= 262 144 pixels)
extern "C" int64_t yh_conv2d_bwd_stats_rows(const yh_conv_desc* d) {
    if (!d || !d->bwd_z || !d->bwd_gamma || !d->bwd_beta || !d->bwd_mean || !d->bwd_invstd || d->n <= 0 || d->ho <= 0 || d->wo <= 0) return 0;
    if (d->dtype != YH_F16 || d->bwd_ldz % 8 || !yh::aligned16(d->bwd_z) || (d->tile != 0 && d->tile != 73)) return 0;
    static const bool off = getenv("YH_NO_BWD_SUMS") != nullptr;      // A/B knob
    if (off || !pick_pwl_tile(d)) return 0;
    return (int64_t)yh::pwl_stats_rows((long)d->n * d->ho * d->wo, d->cout, true);
}

extern "C" int64_t yh_conv2d_stats_rows(const yh_conv_desc* d) {
    if (!d || d->n <= 0 || d->ho <= 0 || d->wo <= 0 || d->dtype == YH_I8) return 0;
    if (d->bwd_z) return 0;
    int bn, wn;
    // the geometry of the launch that WILL carry the statistics: the workspace is attached after this query, and kernels without
    // a statistics epilogue (halo) must not be chosen on its account
    yh_conv_desc q = *d;
    if (!q.stats_ws) q.stats_ws = reinterpret_cast<float*>(sizeof(float));
    const int tile = yh_conv2d_tile(&q);
    if (tile == 72) return (int64_t)yh::stream3_stats_rows((long)d->n * d->ho * d->wo, d->cout);       // one row per wave of the launch
    if (tile == 73) return (int64_t)yh::pwl_stats_rows((long)d->n * d->ho * d->wo, d->cout);            // one row per pixel stream
    if (tile == 43)   // halo ping-pong kernel: 512 VIRTUAL pixels (one shared pad row / column) per tile, one row per wave
        return (int64_t)(((long)d->n * (d->h + 1) * (d->w_in + 1) + 511) / 512) * 8;
    if (!tile_geometry(tile, &bn, &wn)) return 0;
    const long P = (long)d->n * d->ho * d->wo;
    return (int64_t)((P + bn - 1) / bn) * wn;
}

// The fused quantised shortcut with power-of-two factors (conv_igemm.h qadd_n): exactness conditions on the five floats.  rx, ra must be
// integers (2^k, 0 <= k <= 15: |q rx| < 2^23, so |t| + 0.5 and its floor are exact), the two scaled terms may differ by at most 15 binary
// places (8-bit integers: their sum fits 24 bits), and no product leaves the normal range.  YH_QADD_POW2=0 keeps the general arithmetic (A/B).
static void qadd_pow2_args(yh::ConvArgs& a) {
    const char* env = getenv("YH_QADD_POW2");
    if (env && env[0] == '0') return;
    int e[5];
    const float f[5] = {a.q_rx, a.q_ra, a.q_scale_x, a.q_scale_a, a.q_inv_scale_sum};
    for (int i = 0; i < 5; ++i) {
        int ex;
        if (!(f[i] > 0.f) || frexpf(f[i], &ex) != 0.5f) return;
        e[i] = ex - 1;
        if (e[i] < -40 || e[i] > 40) return;
    }
    if (e[0] < 0 || e[0] > 15 || e[1] < 0 || e[1] > 15) return;
    const int ex = e[0] + e[2], ea = e[1] + e[3];
    if (ex - ea > 15 || ea - ex > 15) return;
    a.q_rx = ldexpf(1.f, ex + e[4]);
    a.q_ra = ldexpf(1.f, ea + e[4]);
    a.q_scale_x = 0.f;      // the marker qadd_n tests
}

extern "C" int yh_conv2d_fwd(const yh_conv_desc* d, void* stream) {
    using namespace yh;
    if (!d || !d->x || !d->w || !d->bias || !d->y) return YH_EINVAL;
    if (d->n <= 0 || d->h <= 0 || d->w_in <= 0 || d->cin <= 0 || d->ho <= 0 || d->wo <= 0 || d->cout <= 0) return YH_EINVAL;
    if (d->kh <= 0 || d->kw <= 0 || d->stride <= 0 || d->pad < 0) return YH_EINVAL;
    if (d->dtype != YH_F16 && d->dtype != YH_F32 && d->dtype != YH_I8) return YH_EINVAL;
    const int bk = d->dtype == YH_F16 ? 32 : (d->dtype == YH_I8 ? 64 : 16), vec = d->dtype == YH_F16 ? 8 : (d->dtype == YH_I8 ? 16 : 4);
    if (d->dtype == YH_I8 && (!(d->acc_scale > 0.f) || !(d->out_scale > 0.f))) return YH_EINVAL;
    if (d->dtype == YH_I8 && d->res) {
        // fused quantised shortcut: plain int8 store with one of the compile-time activations, positive scales, int8 residual rows
        if (d->ups != 1 || d->out_f32 || !(d->q_rx > 0.f) || !(d->q_ra > 0.f) || !(d->q_scale_x > 0.f) || !(d->q_scale_a > 0.f) ||
            !(d->q_inv_scale_sum > 0.f) || d->ldr % 16 || (((uintptr_t)d->res) & 3u)) return YH_EINVAL;
        if (d->act != YH_ACT_LINEAR && d->act != YH_ACT_LEAKY && d->act != YH_ACT_MISH) return YH_EUNSUPPORTED;
    }
    if (d->cin % vec || d->ldx % vec || d->cin_k % bk || d->cin_k < d->cin || d->m_pad % 128 || d->m_pad < d->cout) return YH_EALIGN;
    if (d->cout % 4 || d->ldy % 4 || (d->res && d->ldr % 4)) return YH_EALIGN;
    if (!aligned16(d->x) || !aligned16(d->w) || !aligned16(d->bias) || (((uintptr_t)d->y) & 7u) || (((uintptr_t)d->res) & 7u)) return YH_EALIGN;
    if (d->ups < 1 || d->ups > 4) return YH_EINVAL;
    if (d->ups == 4) {
        // the four phases of a stride-2 data gradient in one pass: LDS-DMA tiles only (>= 64 rows), 2x2 window
        if (d->dtype == YH_I8 || d->cout % 16 || d->cout < 64 || d->kh != 2 || d->kw != 2 || d->stride != 1 || d->pad != 0) return YH_EINVAL;
        if (d->tile != 0 && !(d->tile >= 21 && d->tile <= 35)) return YH_EINVAL;
        if (d->y_h <= 0 || d->y_w <= 0 || 2 * d->ho < d->y_h || 2 * d->wo < d->y_w || 2 * (d->ho - 1) >= d->y_h || 2 * (d->wo - 1) >= d->y_w) return YH_EINVAL;
        if (d->stats_ws) return YH_EINVAL;
    } else
    if (d->ups == 3) {
        // phase scatter: free window geometry (taps beyond the input read zeros), destination must hold every pixel
        if (d->dtype == YH_I8 || (d->tile >= 40 && d->tile < 50)) return YH_EINVAL;
        if (d->y_off_h < 0 || d->y_off_w < 0 || 2 * (d->ho - 1) + d->y_off_h >= d->y_h || 2 * (d->wo - 1) + d->y_off_w >= d->y_w) return YH_EINVAL;
    } else if (d->ho != (d->h + 2 * d->pad - d->kh) / d->stride + 1 || d->wo != (d->w_in + 2 * d->pad - d->kw) / d->stride + 1) {
        return YH_EINVAL;
    }

    ConvArgs a;
    a.x = d->x; a.w = d->w; a.bias = d->bias; a.res = d->res; a.y = d->y;
    a.N = d->n; a.H = d->h; a.W = d->w_in; a.Cin = d->cin; a.Ho = d->ho; a.Wo = d->wo; a.Cout = d->cout;
    a.R = d->kh; a.S = d->kw; a.stride = d->stride; a.pad = d->pad;
    a.ldx = d->ldx; a.ldr = d->ldr; a.ldy = d->ldy;
    a.cin_k = d->cin_k; a.ktot = d->kh * d->kw * d->cin_k;
    a.P = (long)d->n * d->ho * d->wo;
    a.m_tiles = a.p_tiles = 0;
    a.m_pad = d->m_pad;
    a.acc_scale = d->acc_scale; a.out_scale = d->out_scale; a.inv_out_scale = d->out_scale > 0.f ? 1.f / d->out_scale : 0.f;
    a.act = d->act; a.slope = d->slope; a.ups = d->ups;
    a.y_h = d->y_h; a.y_w = d->y_w; a.y_off_h = d->y_off_h; a.y_off_w = d->y_off_w;
    a.q_rx = d->q_rx; a.q_ra = d->q_ra; a.q_scale_x = d->q_scale_x; a.q_scale_a = d->q_scale_a; a.q_inv_scale_sum = d->q_inv_scale_sum;
    if (d->dtype == YH_I8 && d->res) qadd_pow2_args(a);
    a.no_lds_store = getenv("YH_PW_DIRECT") != nullptr;
    a.hpp_stagger = 0;
    a.stats_part = nullptr;
    a.bz = nullptr;
    a.bgamma = a.bbeta = a.bmean = a.binvstd = nullptr;
    a.ldbz = a.bact = 0;
    a.bslope = 0.f;
    if (d->bwd_z) {
        // the block's backward sums ride in this data gradient (conv_pw_lds.hip modes 3 / 4): rows into stats_ws
        const int64_t rows = yh_conv2d_bwd_stats_rows(d);
        if (rows <= 0) return YH_EUNSUPPORTED;
        if (!d->stats_ws || d->stats_ws_floats < rows * 2 * d->cout) return YH_EINVAL;
        a.stats_part = d->stats_ws;
        a.bz = d->bwd_z; a.ldbz = d->bwd_ldz; a.bact = d->bwd_act; a.bslope = d->bwd_slope;
        a.bgamma = d->bwd_gamma; a.bbeta = d->bwd_beta; a.bmean = d->bwd_mean; a.binvstd = d->bwd_invstd;
        return launch_pwl_tile(a, d->dtype, (hipStream_t)stream);
    }
    if (d->stats_ws) {
        // fused BatchNorm statistics: plain dense output only, and never on the halo kernels
        if (d->ups != 1 || d->res || d->dtype == YH_I8 || (d->tile >= 40 && d->tile < 50 && d->tile != 43)) return YH_EINVAL;
        const int64_t rows = yh_conv2d_stats_rows(d);
        if (rows <= 0 || d->stats_ws_floats < rows * 2 * d->cout) return YH_EINVAL;
        a.stats_part = d->stats_ws;
    }
    hipStream_t s = (hipStream_t)stream;
    int tile = d->tile;
    if (tile == 0) {
        tile = yh_conv2d_tile(d);
        if (d->ups == 4 && !(tile >= 21 && tile <= 35)) tile = 24;   // only the LDS-DMA ring kernels are used for the four-phase scatter
    }
    if (tile == 71) {
        // explicit requests are validated like the automatic choice (the kernel has no residual / statistics / upsample forms)
        if (d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->ups != 1 || d->res || d->stats_ws || d->cin != d->cin_k ||
            d->cout > 128 || d->dtype == YH_F32) return YH_EUNSUPPORTED;
        return launch_pointwise_tile(a, d->dtype, d->out_f32, s);
    }
    if (tile == 72) {
        if (!stream3_supported(d)) return YH_EUNSUPPORTED;
        return launch_stream3_tile(a, d->dtype, s);
    }
    if (tile == 73) {
        if (!pwl_desc_supported(d)) return YH_EUNSUPPORTED;
        return launch_pwl_tile(a, d->dtype, s);
    }
    if (d->dtype == YH_F16) {
        return d->out_f32 ? dispatch_tile<f16, float>(a, tile, s) : dispatch_tile<f16, f16>(a, tile, s);
    }
    if (d->dtype == YH_I8) {
        return d->out_f32 ? dispatch_tile_i8<float>(a, tile, s) : dispatch_tile_i8<int8_t>(a, tile, s);
    }
    return dispatch_tile<float, float>(a, tile, s);
}
