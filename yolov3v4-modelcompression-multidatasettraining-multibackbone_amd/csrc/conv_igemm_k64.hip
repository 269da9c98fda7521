// Full-line K step implicit-GEMM convolution kernels (tile codes 61 - 63) - see the comment block below.
#include "conv_igemm.h"

// Profiling build only (`make timing` -> libyolo_hip_timing.so, loaded through YOLO_HIP_LIB): thread 0 of every workgroup of the
// ping-pong kernel stores s_memtime at five points (start, loader set up, first K tile landed, K loop done, epilogue done).
#ifdef YH_PP_TIMING
__device__ unsigned long long* g_pp_stamps;
extern "C" int yh_debug_set_pp_stamps(void* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pp_stamps), &p, sizeof(p));
}
#define YH_STAMP(k)                                                                                   \
    do {                                                                                              \
        if (threadIdx.x == 0 && g_pp_stamps) g_pp_stamps[(size_t)blockIdx.x * 64 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
// per-segment stamps of one K tile (kt == 4) for wave 0 (group 0) and wave 4 (group 1): slot 8 + 24 * group + 6 * phase + point
#define YH_FINE(kt, phase, point)                                                                     \
    do {                                                                                              \
        if ((kt) == 4 && (threadIdx.x & 255) == 0 && g_pp_stamps)                                     \
            g_pp_stamps[(size_t)blockIdx.x * 64 + 8 + 24 * (threadIdx.x >> 8) + 6 * (phase) + (point)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define YH_STAMP(k) do {} while (0)
#define YH_FINE(kt, phase, point) do {} while (0)
#endif

namespace yh {

// ---------------------------------------------------------------------------------------------------
// Full-line LDS-DMA variant ("k64"): K step = one 128-byte line per tile row (64 f16 / 128 int8 channels of one tap).
//
// Why (rocprofv3 on the ring kernel above, profiles/r01_*): with a 64-byte K step every tile row is half a cache line per
// L2 request (TCC busy 80 % at 11 TB/s of useful bytes), there is a barrier every 32 channels, and with 64 x 64 per wave the
// fragment reads plus the DMA writes keep the LDS 87 % busy against the MFMA time of a step - nothing is left to overlap.
// Here: whole lines per request, half the barriers, and 128 x 64 outputs per wave (12 fragment reads per 32 MFMAs instead
// of 8 per 16), i.e. a 256 x 256 (or 128 x 512) block tile on 8 waves: LDS traffic per MFMA cycle drops from 0.87 to ~0.62 and
// L2 bytes per FLOP by 1.5 - 2x.  Two LDS stages of (BM + BN) x 128 B; the DMA of step t+1 is issued right after the one
// barrier of step t and lands while step t's 64 MFMAs per wave run (guide: "glds, 2 LDS buffers, BK = 64" form).
//
// LDS image: row-major, 8 cells of 16 B per row; an LDS-DMA instruction fills 8 consecutive rows (lane -> row lane >> 3,
// position lane & 7).  Position c' of row r holds source cell c' ^ ((r >> 1) & 7): a ds_read_b128 fragment read (lane = row
// r of 16, cell 4 h + (lane >> 4)) then touches 16 distinct 16-byte slots of the 256-byte bank row in every 16-lane group of
// the instruction - conflict free (the swizzle is applied to the per-lane SOURCE address and to the read, never to the DMA
// destination, which is lane-linear by construction).
template <typename T, int TM, int TN> struct MmaStepK64;
template <int TM, int TN> struct MmaStepK64<f16, TM, TN> {
    static __device__ __forceinline__ void run(const u32x4* As, const u32x4* Bs, int arow, int brow, int lane, f32x4 (&acc)[TM][TN]) {
        const int r = lane & 15, kq = lane >> 4, f = (r >> 1) & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // two 32-channel MFMA steps per line, in channel order (same summation order as the ring kernel)
            const int off = r * 8 + ((h * 4 + kq) ^ f);
            f16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                u32x4 v = As[(arow + i * 16) * 8 + off];
                af[i] = *reinterpret_cast<f16x8*>(&v);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                u32x4 v = Bs[(brow + j * 16) * 8 + off];
                bf[j] = *reinterpret_cast<f16x8*>(&v);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
};
template <int TM, int TN> struct MmaStepK64<int8_t, TM, TN> {
    static __device__ __forceinline__ void run(const u32x4* As, const u32x4* Bs, int arow, int brow, int lane, i32x4 (&acc)[TM][TN]) {
        const int r = lane & 15, kq = lane >> 4, f = (r >> 1) & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // two 64-channel v_mfma_i32_16x16x64_i8 steps per 128-channel line
            const int off = r * 8 + ((h * 4 + kq) ^ f);
            i32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                u32x4 v = As[(arow + i * 16) * 8 + off];
                af[i] = *reinterpret_cast<i32x4*>(&v);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                u32x4 v = Bs[(brow + j * 16) * 8 + off];
                bf[j] = *reinterpret_cast<i32x4*>(&v);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
};

template <typename T, typename OutT, int BM, int BN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN) / 4) void conv_igemm_k64_kernel(const ConvArgs a) {
    constexpr int VEC = Prec<T>::VEC, BK = VEC * 8;   // one 128-byte line per row per step
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int GA = BM / 8, GB = BN / 8;           // 8-row groups: one LDS-DMA instruction each
    constexpr int GAW = GA / NW, GBW = GB / NW;       // per wave per K step
    static_assert(GA % NW == 0 && GB % NW == 0, "tile rows must split evenly over the waves");
    static_assert(sizeof(T) <= 2, "f16 / int8 only");
    constexpr int STAGE_CELLS = 8 * (BM + BN);
    __shared__ u32x4 smem[STAGES * STAGE_CELLS];      // the only LDS object of the kernel

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

    // loader: lane -> row lane >> 3 of its 8-row group, LDS position lane & 7, which holds source cell (lane & 7) ^ f(row)
    const int lrow = lane >> 3;
    const T* const xg = reinterpret_cast<const T*>(a.x);
    const T* const zero = reinterpret_cast<const T*>(g_zero_page);
    const T* wsrc[GAW];
    static_for<GAW>([&](auto c) {
        constexpr int i = decltype(c)::value;
        const int trow = (wave + i * NW) * 8 + lrow;                       // row inside the tile
        const int lu = (lane & 7) ^ ((trow >> 1) & 7);
        const int row = min(m0 + trow, a.m_pad - 1);                        // tiles taller than the packed image's padding
        wsrc[i] = reinterpret_cast<const T*>(a.w) + (long)row * a.ktot + lu * VEC;
    });
    long bbase[GBW];
    int bhi[GBW], bwi[GBW], bcell[GBW];
    const int HoWo = a.Ho * a.Wo;
    static_for<GBW>([&](auto c) {
        constexpr int i = decltype(c)::value;
        const int trow = (wave + i * NW) * 8 + lrow;
        const int lu = (lane & 7) ^ ((trow >> 1) & 7);
        bcell[i] = lu * VEC;
        const long p = p0 + trow;
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
    auto issue = [&](int st) {  // LDS-DMA of the K step (kr, ks, kc) into stage st, then advance the step
        u32x4* const base = smem + st * STAGE_CELLS;
        static_for<GAW>([&](auto c) {
            constexpr int i = decltype(c)::value;
            u32x4* dst = base + (wave + i * NW) * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + kofs),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        });
        const long tap = ((long)kr * a.W + ks) * a.ldx + kc;
        const bool inr = (unsigned)kr < (unsigned)a.R;   // always true; keeps the tap test below branch-free
        static_for<GBW>([&](auto c) {
            constexpr int i = decltype(c)::value;
            const bool ok = inr && kc + bcell[i] < a.Cin && (unsigned)(bhi[i] + kr) < (unsigned)a.H &&
                            (unsigned)(bwi[i] + ks) < (unsigned)a.W;
            const T* src = ok ? xg + bbase[i] + tap : zero;
            u32x4* dst = base + 8 * BM + (wave + i * NW) * 64;
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

    constexpr int GPW = GAW + GBW;
    const int nk = a.ktot / BK;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) issue(s);
    int st_read = 0, st_write = STAGES - 1;
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's share of step kt has landed once at most `ahead` later steps are still in flight
        const int ahead = min(STAGES - 2, nk - 1 - kt);
        if (STAGES >= 3 && ahead >= 1) wait_vmcnt<(STAGES >= 3 ? GPW : 0)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();  // everyone's share of step kt landed; everyone is done reading stage st_write (step kt-1)
        if (kt + STAGES - 1 < nk) issue(st_write);
        const u32x4* As = smem + st_read * STAGE_CELLS;
        MmaStepK64<T, TM, TN>::run(As, As + 8 * BM, wm * TM * 16, wn * TN * 16, lane, acc);
        st_read = st_read + 1 == STAGES ? 0 : st_read + 1;
        st_write = st_write + 1 == STAGES ? 0 : st_write + 1;
    }

    conv_epilogue<T, OutT, TM, TN, BN, WN>(a, acc, m0, p0, wm, wn, lane);
}

template <typename T, typename OutT, int BM, int BN, int WM, int WN, int STAGES>
static int launch_k64(const ConvArgs& a0, hipStream_t stream) {
    ConvArgs a = a0;
    if (a.cin_k % (Prec<T>::VEC * 8)) return YH_EALIGN;   // whole 128-byte lines per tap
    a.m_tiles = (a.Cout + BM - 1) / BM;
    a.p_tiles = (int)((a.P + BN - 1) / BN);
    const long blocks = (long)a.m_tiles * a.p_tiles;
    if (blocks <= 0 || blocks > 0x7fffffffL) return YH_EINVAL;
    hipLaunchKernelGGL((conv_igemm_k64_kernel<T, OutT, BM, BN, WM, WN, STAGES>), dim3((unsigned)blocks), dim3(WM * WN * 64), 0,
                       stream, a);
    return check_launch();
}


// ---------------------------------------------------------------------------------------------------
// "Ping-pong" form of the full-line kernel: the two waves that share a SIMD alternate between a LOAD segment (fragment
// reads from LDS + LDS-DMA issue for the next K tile) and an MFMA segment (16 MFMAs on one 64 x 32 quadrant of the wave's
// 128 x 64 outputs), so each SIMD's matrix pipe always has one wave in its MFMA segment while the other wave's LDS / VMEM
// issue runs beside it (MI355X_MICROARCH.md "Two waves per SIMD"; the 8-phase GEMM schedule of the programming guide).
//
//   waves 0-3 (group 0) and 4-7 (group 1) sit pairwise on the 4 SIMDs; group 1 runs one barrier interval behind group 0:
//       interval:   0     1     2     3     4     5     6     7     8 ...
//       group 0:    L1    M1    L2    M2    L3    M3    L4    M4    L1' ...
//       group 1:    -     L1    M1    L2    M2    L3    M3    L4    M4 ...
//   every interval ends with one s_barrier executed by all 8 waves (group 1 executes one extra at the start, group 0 one
//   extra at the end, so the counts match).
//
// K tile = 64 channels of one tap = 4 phases (quadrants (a0,b0) (a0,b1) (a1,b1) (a1,b0) of the wave tile; a = 64-row half of
// the wave's A rows, b = 32-pixel half of its B rows).  Phase 1 reads A[a0] + B[b0] (12 ds_read_b128), phase 2 B[b1] (4),
// phase 3 A[a1] (8), phase 4 nothing (B[b0] is still in registers).  The LDS-DMA of K tile t+1 is issued during tile t in
// first-use order - A[a0] rows of all waves in phase 1, B[b0] rows in phase 2, B[b1] in phase 3, A[a1] in phase 4 - into the
// other LDS stage, so every piece has three phase intervals (~1500 cycles) to land: two stages of (BM + BN) x 128 B suffice.
// A wave waits for ITS pieces with a counted vmcnt at the end of the load segment that precedes the consumer phase; the
// barrier that ends that interval (and, for the other group, the next one) makes them visible to the readers.
// Write-after-read: a region of the other stage was last read two or more phases before its refill is issued.
template <typename T, typename OutT, int WM, int WN>
__global__ __launch_bounds__(512, 2) void conv_igemm_pp_kernel(const ConvArgs a) {
    static_assert(WM * WN == 8 && sizeof(T) <= 2, "8 waves; f16 (64 channels per line) or int8 (128 channels per line)");
    constexpr int VEC = Prec<T>::VEC, BK = VEC * 8;
    constexpr int BM = WM * 128, BN = WN * 64;
    constexpr int TM = 8, TN = 4;
    constexpr int NA = WM, NB = WN / 2;               // LDS-DMA instructions per wave per piece set (A half / B half)
    static_assert(WN % 2 == 0, "B piece sets must split over 8 waves");
    constexpr int STAGE_CELLS = 8 * (BM + BN);
    __shared__ u32x4 smem[2 * STAGE_CELLS];           // the only LDS object of the kernel

    YH_STAMP(0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int group = wave >> 2;

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

    // ---- loader.  Piece set X in {A0, A1, B0, B1} = the rows every wave needs first in phase 1 / 3 / 1 / 2:
    //      A_h: tile rows (q >> 6) * 128 + h * 64 + (q & 63), q < 64 WM;   B_h: (q >> 5) * 64 + h * 32 + (q & 31), q < 32 WN.
    //      This wave issues the 8-row groups q0 = (wave * N + i) * 8 of each set; lane -> row q0 + (lane >> 3), position lane & 7.
    const int lrow = lane >> 3;
    const T* const xg = reinterpret_cast<const T*>(a.x);
    const T* const zero = reinterpret_cast<const T*>(g_zero_page);
    const T* wsrc[2][NA];
    int adst[2][NA];
    static_for<2 * NA>([&](auto c) {
        constexpr int h = decltype(c)::value / NA, i = decltype(c)::value % NA;
        const int q0 = (wave * NA + i) * 8;
        const int trow0 = (q0 >> 6) * 128 + h * 64 + (q0 & 63);
        const int trow = trow0 + lrow;
        const int lu = (lane & 7) ^ ((trow >> 1) & 7);
        const int row = min(m0 + trow, a.m_pad - 1);
        wsrc[h][i] = reinterpret_cast<const T*>(a.w) + (long)row * a.ktot + lu * VEC;
        adst[h][i] = trow0 * 8;
    });
    long bbase[2][NB];
    int bhw[2][NB], bcell[2][NB], bdst[2][NB];
    const int HoWo = a.Ho * a.Wo;
    static_for<2 * NB>([&](auto c) {
        constexpr int h = decltype(c)::value / NB, i = decltype(c)::value % NB;
        const int q0 = (wave * NB + i) * 8;
        const int trow0 = (q0 >> 5) * 64 + h * 32 + (q0 & 31);
        const int trow = trow0 + lrow;
        const int lu = (lane & 7) ^ ((trow >> 1) & 7);
        bcell[h][i] = lu * VEC;
        bdst[h][i] = (BM + trow0) * 8;
        const long p = p0 + trow;
        if (p < a.P) {
            const int n = (int)(p / HoWo);
            const int rem = (int)(p - (long)n * HoWo);
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            const int hi = ho * a.stride - a.pad, wi = wo * a.stride - a.pad;
            bhw[h][i] = (hi << 16) | (wi & 0xffff);     // both fit 16 signed bits (images up to 32k pixels a side)
            bbase[h][i] = (((long)n * a.H + hi) * a.W + wi) * a.ldx + lu * VEC;
        } else {
            bhw[h][i] = (int)0x80008000;                // far outside every image: every tap reads the zero page
            bbase[h][i] = 0;
        }
    });

    int kr = 0, ks = 0, kc = 0, kofs = 0;   // tap / channel offset of the K tile being FETCHED
    long tap = 0;
    auto advance = [&]() {
        kofs += BK;
        kc += BK;
        if (kc >= a.cin_k) {
            kc = 0;
            if (++ks == a.S) { ks = 0; ++kr; }
        }
        tap = ((long)kr * a.W + ks) * a.ldx + kc;
    };
    auto issue_a = [&](u32x4* base, auto hc) {
        constexpr int h = decltype(hc)::value;
        static_for<NA>([&](auto c) {
            constexpr int i = decltype(c)::value;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[h][i] + kofs),
                                             (__attribute__((address_space(3))) void*)(base + adst[h][i]), 16, 0, 0);
        });
    };
    auto issue_b = [&](u32x4* base, auto hc) {
        constexpr int h = decltype(hc)::value;
        static_for<NB>([&](auto c) {
            constexpr int i = decltype(c)::value;
            const int hi = bhw[h][i] >> 16, wi = (int)(short)(bhw[h][i] & 0xffff);
            const bool ok = kc + bcell[h][i] < a.Cin && (unsigned)(hi + kr) < (unsigned)a.H && (unsigned)(wi + ks) < (unsigned)a.W;
            const T* src = ok ? xg + bbase[h][i] + tap : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + bdst[h][i]), 16, 0, 0);
        });
    };
    typedef std::integral_constant<int, 0> H0;
    typedef std::integral_constant<int, 1> H1;

    typedef typename AccOf<T>::type acc_t;
    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

    // fragment registers: A half (4 fragments x 2 K halves), both B halves (2 fragments x 2 K halves each); a fragment is one
    // 16-byte cell per lane: 8 f16 (v_mfma_f32_16x16x32_f16) or 16 int8 (v_mfma_i32_16x16x64_i8)
    typedef typename std::conditional<sizeof(T) == 2, f16x8, i32x4>::type frag_t;
    frag_t fa[2][4], fb[2][2][2];
    const int r16 = lane & 15, kq = lane >> 4, fsw = (r16 >> 1) & 7;
    const int off0 = r16 * 8 + ((0 + kq) ^ fsw), off1 = r16 * 8 + ((4 + kq) ^ fsw);
    const int arow = wm * 128, brow = BM + wn * 64;
    auto read_a = [&](const u32x4* st, int half) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x4 v0 = st[(arow + half * 64 + i * 16) * 8 + off0];
            u32x4 v1 = st[(arow + half * 64 + i * 16) * 8 + off1];
            fa[0][i] = *reinterpret_cast<frag_t*>(&v0);
            fa[1][i] = *reinterpret_cast<frag_t*>(&v1);
        }
    };
    auto read_b = [&](const u32x4* st, auto hc) {
        constexpr int h = decltype(hc)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u32x4 v0 = st[(brow + h * 32 + j * 16) * 8 + off0];
            u32x4 v1 = st[(brow + h * 32 + j * 16) * 8 + off1];
            fb[h][0][j] = *reinterpret_cast<frag_t*>(&v0);
            fb[h][1][j] = *reinterpret_cast<frag_t*>(&v1);
        }
    };
    auto mma = [&](auto ac, auto bc) {   // quadrant (a, b): 4 x 2 fragments x 2 K halves, channel order within the line
        constexpr int ah = decltype(ac)::value, bh = decltype(bc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if constexpr (sizeof(T) == 2)
                        acc[ah * 4 + i][bh * 2 + j] =
                            __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[k][i], fb[bh][k][j], acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
                    else
                        acc[ah * 4 + i][bh * 2 + j] =
                            __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[k][i], fb[bh][k][j], acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // end of an interval: all of this wave's LDS reads have returned (they feed the MFMAs after the barrier), then the barrier
#define YH_PP_SYNC()                                        \
    do {                                                    \
        __builtin_amdgcn_sched_barrier(0);                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                       \
        __builtin_amdgcn_sched_barrier(0);                  \
    } while (0)
#define YH_PP_BARRIER()                      \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)

    const int nk = a.ktot / BK;
    // prologue: K tile 0 into stage 0 (all four piece sets), everyone waits for everything
    YH_STAMP(1);
    issue_a(smem, H0{});
    issue_b(smem, H0{});
    issue_b(smem, H1{});
    issue_a(smem, H1{});
    wait_vmcnt<0>();
    YH_PP_BARRIER();
    YH_STAMP(2);
    if (group == 1) YH_PP_BARRIER();   // stagger: group 1 starts one interval late

    for (int kt = 0; kt < nk; ++kt) {
        const u32x4* cur = smem + (kt & 1) * STAGE_CELLS;
        u32x4* nxt = smem + ((kt + 1) & 1) * STAGE_CELLS;
        const bool more = kt + 1 < nk;
        // one phase = LOAD segment, barrier, MFMA segment, barrier (YH_FINE: profiling-build timestamps, no code otherwise)
#define YH_PP_PHASE(P, LOADS, MMA)                                  \
        YH_FINE(kt, P, 0);                                          \
        LOADS;                                                      \
        __builtin_amdgcn_sched_barrier(0);                          \
        YH_FINE(kt, P, 1);                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
        YH_FINE(kt, P, 2);                                          \
        __builtin_amdgcn_s_barrier();                               \
        __builtin_amdgcn_sched_barrier(0);                          \
        YH_FINE(kt, P, 3);                                          \
        MMA;                                                        \
        __builtin_amdgcn_sched_barrier(0);                          \
        YH_FINE(kt, P, 4);                                          \
        __builtin_amdgcn_s_barrier();                               \
        __builtin_amdgcn_sched_barrier(0);                          \
        YH_FINE(kt, P, 5)
        // ---- phase 1: quadrant (a0, b0).  B[b1] of THIS tile must have landed before phase 2 reads it: at most A[a1](kt) and
        // A[a0](kt+1) stay in flight
        YH_PP_PHASE(0, { read_a(cur, 0); read_b(cur, H0{}); if (more) { advance(); issue_a(nxt, H0{}); }
                         if (more) wait_vmcnt<2 * NA>(); else wait_vmcnt<NA>(); }, mma(H0{}, H0{}));
        // ---- phase 2: quadrant (a0, b1).  A[a1] of this tile before phase 3: A[a0], B[b0] of the next tile stay in flight
        YH_PP_PHASE(1, { read_b(cur, H1{}); if (more) issue_b(nxt, H0{});
                         if (more) wait_vmcnt<NA + NB>(); else wait_vmcnt<0>(); }, mma(H0{}, H1{}));
        // ---- phase 3: quadrant (a1, b1)
        YH_PP_PHASE(2, { read_a(cur, 1); if (more) issue_b(nxt, H1{}); }, mma(H1{}, H1{}));
        // ---- phase 4: quadrant (a1, b0); no reads.  A[a0], B[b0] of the next tile before its phase 1: B[b1], A[a1] stay in flight
        YH_PP_PHASE(3, { if (more) { issue_a(nxt, H1{}); wait_vmcnt<NA + NB>(); } }, mma(H1{}, H0{}));
#undef YH_PP_PHASE
    }
    if (group == 0) YH_PP_BARRIER();   // matches group 1's extra barrier at the start
#undef YH_PP_SYNC
#undef YH_PP_BARRIER
    YH_STAMP(3);

    conv_epilogue<T, OutT, TM, TN, BN, WN>(a, acc, m0, p0, wm, wn, lane);
#ifdef YH_PP_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wave's stores have left
    YH_STAMP(4);
#endif
}

// The ping-pong kernels pack (hi, wi) into 16 signed bits each and (pp2) keep 32-bit element offsets: images below 32k pixels a
// side, both operands below 2^31 elements (the far corner of the last tap of the last pixel included).  Checked by every launcher,
// so an explicit tile code cannot reach the kernels with a tensor the automatic choice (pick_pp_tile) would have refused.
static bool pp_offsets_fit(const ConvArgs& a) {
    if (a.H >= 32768 || a.W >= 32768) return false;
    return (long)a.N * a.H * a.W * a.ldx + (long)(a.R + 1) * a.W * a.ldx < 0x7fffffffL && (long)a.m_pad * a.ktot < 0x7fffffffL;
}

template <typename T, typename OutT, int WM, int WN>
static int launch_pp(const ConvArgs& a0, hipStream_t stream) {
    constexpr int BM = WM * 128, BN = WN * 64;
    ConvArgs a = a0;
    if (a.cin_k % (Prec<T>::VEC * 8)) return YH_EALIGN;
    if (!pp_offsets_fit(a)) return YH_EUNSUPPORTED;
    a.m_tiles = (a.Cout + BM - 1) / BM;
    a.p_tiles = (int)((a.P + BN - 1) / BN);
    const long blocks = (long)a.m_tiles * a.p_tiles;
    if (blocks <= 0 || blocks > 0x7fffffffL) return YH_EINVAL;
    hipLaunchKernelGGL((conv_igemm_pp_kernel<T, OutT, WM, WN>), dim3((unsigned)blocks), dim3(512), 0, stream, a);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------------
// Ping-pong kernel, second schedule ("pp2"): balanced load segments and a two-tile-deep LDS-DMA pipeline in the same two stages.
//
// Measured on the first schedule (s_memtime stamps, profiles/r02_pp_timing.txt): 3300 cycles per K tile against 2048 of MFMA
// work - every barrier interval lasts max(load segment, 256-cycle MFMA segment) and the load segments were 12 / 4 / 8 / 0
// fragment reads.  Here the A fragments are double buffered in registers (a0 / a1 halves of the wave's rows), only one B half
// is resident, and the reads are spread 8 / 8 / 4 / 8 over the four phases:
//     L1: B[b0], A[a1] lo        M1: (a0, b0)
//     L2: B[b1], A[a1] hi        M2: (a0, b1)
//     L3: next tile's A[a0] lo   M3: (a1, b1)
//     L4: B[b0], next A[a0] hi   M4: (a1, b0)
// A region of an LDS stage is refilled as soon as its last reader phase is over, for the tile that will use that stage next:
//     L1(t): A[a0] of tile t+2 -> this stage      L2(t): B[b0] of tile t+1 -> other stage
//     L3(t): A[a1] of tile t+2 -> this stage      L4(t): B[b1] of tile t+2 -> this stage
// so every piece has 4 - 6 phase intervals to land, and one counted vmcnt per K tile (end of L4: B[b0] of the next tile and,
// being older in the in-order queue, everything else the next tile's first three phases read) is the only wait on memory.
template <typename T, typename OutT, int WM, int WN>
__global__ __launch_bounds__(512, 2) void conv_igemm_pp2_kernel(const ConvArgs a) {
    static_assert(WM * WN == 8 && sizeof(T) == 2, "8 waves, f16");
    constexpr int VEC = Prec<T>::VEC, BK = VEC * 8;
    constexpr int BM = WM * 128, BN = WN * 64;
    constexpr int TM = 8, TN = 4;
    constexpr int NA = WM, NB = WN / 2;               // LDS-DMA instructions per wave per piece set (A half / B half)
    static_assert(WN % 2 == 0, "B piece sets must split over 8 waves");
    constexpr int STAGE_CELLS = 8 * (BM + BN);
    __shared__ u32x4 smem[2 * STAGE_CELLS];           // the only LDS object of the kernel

    YH_STAMP(0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int group = wave >> 2;

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

    // ---- loader (same piece sets as the first schedule)
    const int lrow = lane >> 3;
    const T* const xg = reinterpret_cast<const T*>(a.x);
    const T* const wg = reinterpret_cast<const T*>(a.w);
    const T* const zero = reinterpret_cast<const T*>(g_zero_page);
    int wsrc[2][NA];     // 32-bit element offsets (launch_pp2 checks that both operands stay below 2^31 elements)
    int adst[2][NA];
    static_for<2 * NA>([&](auto c) {
        constexpr int h = decltype(c)::value / NA, i = decltype(c)::value % NA;
        const int q0 = (wave * NA + i) * 8;
        const int trow0 = (q0 >> 6) * 128 + h * 64 + (q0 & 63);
        const int trow = trow0 + lrow;
        const int lu = (lane & 7) ^ ((trow >> 1) & 7);
        const int row = min(m0 + trow, a.m_pad - 1);
        wsrc[h][i] = row * a.ktot + lu * VEC;
        adst[h][i] = trow0 * 8;
    });
    // the source cell of a lane is ((lane & 7) ^ (lane >> 4)) ^ 4 u with u = bit 3 of the (wave-uniform) first row of its
    // 8-row group: one register for all pieces plus a scalar per piece
    int bbase[2][NB];
    int bhw[2][NB], bflip[2][NB], bdst[2][NB];
    const int cell0 = ((lane & 7) ^ (lane >> 4)) * VEC;
    const int HoWo = a.Ho * a.Wo;
    static_for<2 * NB>([&](auto c) {
        constexpr int h = decltype(c)::value / NB, i = decltype(c)::value % NB;
        const int q0 = (wave * NB + i) * 8;
        const int trow0 = (q0 >> 5) * 64 + h * 32 + (q0 & 31);
        const int trow = trow0 + lrow;
        const int lu = (lane & 7) ^ ((trow >> 1) & 7);
        bflip[h][i] = ((trow0 >> 3) & 1) * 4 * VEC;
        bdst[h][i] = (BM + trow0) * 8;
        const long p = p0 + trow;
        if (p < a.P) {
            const int n = (int)(p / HoWo);
            const int rem = (int)(p - (long)n * HoWo);
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            const int hi = ho * a.stride - a.pad, wi = wo * a.stride - a.pad;
            bhw[h][i] = (hi << 16) | (wi & 0xffff);
            bbase[h][i] = (int)((((long)n * a.H + hi) * a.W + wi) * a.ldx) + lu * VEC;
        } else {
            bhw[h][i] = (int)0x80008000;
            bbase[h][i] = 0;
        }
    });

    struct Tap { int kr, ks, kc, kofs, tap; };   // filter tap / channel offset of a K tile (wave-uniform)
    auto next_tap = [&](Tap t) {
        t.kofs += BK;
        t.kc += BK;
        if (t.kc >= a.cin_k) {
            t.kc = 0;
            if (++t.ks == a.S) { t.ks = 0; ++t.kr; }
        }
        t.tap = (t.kr * a.W + t.ks) * a.ldx + t.kc;
        return t;
    };
    auto issue_a = [&](u32x4* base, const Tap& t, auto hc) {
        constexpr int h = decltype(hc)::value;
        static_for<NA>([&](auto c) {
            constexpr int i = decltype(c)::value;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg + (wsrc[h][i] + t.kofs)),
                                             (__attribute__((address_space(3))) void*)(base + adst[h][i]), 16, 0, 0);
        });
    };
    auto issue_b = [&](u32x4* base, const Tap& t, auto hc) {
        constexpr int h = decltype(hc)::value;
        static_for<NB>([&](auto c) {
            constexpr int i = decltype(c)::value;
            const int hi = bhw[h][i] >> 16, wi = (int)(short)(bhw[h][i] & 0xffff);
            const bool ok = t.kc + (cell0 ^ bflip[h][i]) < a.Cin && (unsigned)(hi + t.kr) < (unsigned)a.H &&
                            (unsigned)(wi + t.ks) < (unsigned)a.W;
            const T* src = ok ? xg + (bbase[h][i] + t.tap) : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + bdst[h][i]), 16, 0, 0);
        });
    };
    typedef std::integral_constant<int, 0> H0;
    typedef std::integral_constant<int, 1> H1;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment registers: both A halves (4 fragments x 2 K halves each), one B half (2 fragments x 2 K halves)
    f16x8 fa[2][2][4], fb[2][2];
    const int r16 = lane & 15, kq = lane >> 4, fsw = (r16 >> 1) & 7;
    const int off0 = r16 * 8 + ((0 + kq) ^ fsw), off1 = r16 * 8 + ((4 + kq) ^ fsw);
    const int arow = wm * 128, brow = BM + wn * 64;
    auto read_a = [&](const u32x4* st, auto hc, auto pc_) {   // A half hc, fragments 2 part .. 2 part + 1
        constexpr int h = decltype(hc)::value, part = decltype(pc_)::value;
#pragma unroll
        for (int i = 2 * part; i < 2 * part + 2; ++i) {
            u32x4 v0 = st[(arow + h * 64 + i * 16) * 8 + off0];
            u32x4 v1 = st[(arow + h * 64 + i * 16) * 8 + off1];
            fa[h][0][i] = *reinterpret_cast<f16x8*>(&v0);
            fa[h][1][i] = *reinterpret_cast<f16x8*>(&v1);
        }
    };
    auto read_b = [&](const u32x4* st, auto hc) {
        constexpr int h = decltype(hc)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u32x4 v0 = st[(brow + h * 32 + j * 16) * 8 + off0];
            u32x4 v1 = st[(brow + h * 32 + j * 16) * 8 + off1];
            fb[0][j] = *reinterpret_cast<f16x8*>(&v0);
            fb[1][j] = *reinterpret_cast<f16x8*>(&v1);
        }
    };
    auto mma = [&](auto ac, auto bc) {   // quadrant (a, b): 4 x 2 fragments x 2 K halves, channel order within the line
        constexpr int ah = decltype(ac)::value, bh = decltype(bc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ah * 4 + i][bh * 2 + j] =
                        __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ah][k][i], fb[k][j], acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
#define YH_PP_SYNC()                                        \
    do {                                                    \
        __builtin_amdgcn_sched_barrier(0);                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                       \
        __builtin_amdgcn_sched_barrier(0);                  \
    } while (0)
#define YH_PP_BARRIER()                      \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)

    const int nk = a.ktot / BK;
    u32x4* const stage0 = smem;
    u32x4* const stage1 = smem + STAGE_CELLS;
    // prologue: all of K tile 0 into stage 0; of tile 1 what the steady state would already have issued (A[a0], A[a1], B[b1])
    Tap t1{0, 0, 0, 0, 0};          // tile 0 for now
    YH_STAMP(1);
    issue_a(stage0, t1, H0{});
    issue_b(stage0, t1, H0{});
    issue_b(stage0, t1, H1{});
    issue_a(stage0, t1, H1{});
    t1 = next_tap(t1);              // tile 1
    if (nk > 1) {
        issue_a(stage1, t1, H0{});
        issue_a(stage1, t1, H1{});
        issue_b(stage1, t1, H1{});
    }
    Tap t2 = next_tap(t1);          // tile 2
    wait_vmcnt<0>();
    YH_PP_BARRIER();
    read_a(stage0, H0{}, H0{});     // A[a0] of tile 0 (later tiles get it in L3 / L4 of their predecessor)
    read_a(stage0, H0{}, H1{});
    YH_PP_SYNC();                   // everyone holds A[a0](0): its LDS region may be refilled from L1(0) on
    YH_STAMP(2);
    if (group == 1) YH_PP_BARRIER();   // stagger: group 1 starts one interval late

    for (int kt = 0; kt < nk; ++kt) {
        u32x4* const cur = (kt & 1) ? stage1 : stage0;
        u32x4* const nxt = (kt & 1) ? stage0 : stage1;
        const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
        // ---- phase 1: quadrant (a0, b0)
        read_b(cur, H0{});
        read_a(cur, H1{}, H0{});
        if (more2) issue_a(cur, t2, H0{});
        YH_PP_SYNC();
        mma(H0{}, H0{});
        YH_PP_BARRIER();
        // ---- phase 2: quadrant (a0, b1)
        read_b(cur, H1{});
        read_a(cur, H1{}, H1{});
        if (more1) issue_b(nxt, t1, H0{});
        YH_PP_SYNC();
        mma(H0{}, H1{});
        YH_PP_BARRIER();
        // ---- phase 3: quadrant (a1, b1); the next tile's A[a0] starts to move into the registers M1 / M2 are done with
        if (more1) read_a(nxt, H0{}, H0{});
        if (more2) issue_a(cur, t2, H1{});
        YH_PP_SYNC();
        mma(H1{}, H1{});
        YH_PP_BARRIER();
        // ---- phase 4: quadrant (a1, b0)
        read_b(cur, H0{});
        if (more1) read_a(nxt, H0{}, H1{});
        if (more2) issue_b(cur, t2, H1{});
        // B[b0] of the next tile (issued in L2) must have landed before its L1; A[a1] / B[b1] of tile t+2 stay in flight
        if (more2) wait_vmcnt<NA + NB>(); else wait_vmcnt<0>();
        YH_PP_SYNC();
        mma(H1{}, H0{});
        YH_PP_BARRIER();
        t1 = t2;
        t2 = next_tap(t2);
    }
    if (group == 0) YH_PP_BARRIER();   // matches group 1's extra barrier at the start
#undef YH_PP_SYNC
#undef YH_PP_BARRIER
    YH_STAMP(3);

    conv_epilogue<T, OutT, TM, TN, BN, WN>(a, acc, m0, p0, wm, wn, lane);
#ifdef YH_PP_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    YH_STAMP(4);
#endif
}

template <typename T, typename OutT, int WM, int WN>
static int launch_pp2(const ConvArgs& a0, hipStream_t stream) {
    constexpr int BM = WM * 128, BN = WN * 64;
    ConvArgs a = a0;
    if (a.cin_k % (Prec<T>::VEC * 8)) return YH_EALIGN;
    if (!pp_offsets_fit(a)) return YH_EUNSUPPORTED;
    a.m_tiles = (a.Cout + BM - 1) / BM;
    a.p_tiles = (int)((a.P + BN - 1) / BN);
    const long blocks = (long)a.m_tiles * a.p_tiles;
    if (blocks <= 0 || blocks > 0x7fffffffL) return YH_EINVAL;
    hipLaunchKernelGGL((conv_igemm_pp2_kernel<T, OutT, WM, WN>), dim3((unsigned)blocks), dim3(512), 0, stream, a);
    return check_launch();
}

template <typename T, typename OutT> static int dispatch_k64(const ConvArgs& a, int tile, hipStream_t s) {
    switch (tile) {
        case 61: return launch_k64<T, OutT, 256, 256, 2, 4, 2>(a, s);   // 8 waves, 128 KB of LDS
        case 62: return launch_k64<T, OutT, 128, 512, 1, 8, 2>(a, s);   // 8 waves, 160 KB
        case 63: return launch_k64<T, OutT, 256, 128, 2, 2, 3>(a, s);   // 4 waves (one per SIMD), 144 KB, 3 stages
        case 64: return launch_pp<T, OutT, 2, 4>(a, s);                 // ping-pong 256 x 256
        case 65: return launch_pp<T, OutT, 1, 8>(a, s);                 // ping-pong 128 x 512
        case 66: return launch_pp<T, OutT, 4, 2>(a, s);                 // ping-pong 512 x 128
        case 67: return launch_pp2<T, OutT, 2, 4>(a, s);                // second schedule, 256 x 256
        case 68: return launch_pp2<T, OutT, 1, 8>(a, s);                // 128 x 512
        case 69: return launch_pp2<T, OutT, 4, 2>(a, s);                // 512 x 128
        default: return YH_EINVAL;
    }
}

template <typename OutT> static int dispatch_pp_i8(const ConvArgs& a, int tile, hipStream_t s) {
    switch (tile) {   // int8: one line = 128 channels of a tap
        case 64: return launch_pp<int8_t, OutT, 2, 4>(a, s);
        case 65: return launch_pp<int8_t, OutT, 1, 8>(a, s);
        case 66: return launch_pp<int8_t, OutT, 4, 2>(a, s);
        default: return YH_EINVAL;
    }
}

int launch_k64_tile(const ConvArgs& a, int tile, int dtype, int out_f32, hipStream_t stream) {
    if (dtype == YH_F16) return out_f32 ? dispatch_k64<f16, float>(a, tile, stream) : dispatch_k64<f16, f16>(a, tile, stream);
    if (dtype == YH_I8) return out_f32 ? dispatch_pp_i8<float>(a, tile, stream) : dispatch_pp_i8<int8_t>(a, tile, stream);
    return YH_EINVAL;
}

}  // namespace yh
