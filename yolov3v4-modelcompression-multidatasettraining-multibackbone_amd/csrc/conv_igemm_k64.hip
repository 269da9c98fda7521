// Full-line K step implicit-GEMM convolution kernels (tile codes 61 - 63) - see the comment block below.
#include "conv_igemm.h"

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


template <typename T, typename OutT> static int dispatch_k64(const ConvArgs& a, int tile, hipStream_t s) {
    switch (tile) {
        case 61: return launch_k64<T, OutT, 256, 256, 2, 4, 2>(a, s);   // 8 waves, 128 KB of LDS
        case 62: return launch_k64<T, OutT, 128, 512, 1, 8, 2>(a, s);   // 8 waves, 160 KB
        case 63: return launch_k64<T, OutT, 256, 128, 2, 2, 3>(a, s);   // 4 waves (one per SIMD), 144 KB, 3 stages
        default: return YH_EINVAL;
    }
}

int launch_k64_tile(const ConvArgs& a, int tile, int dtype, int out_f32, hipStream_t stream) {
    if (dtype == YH_F16) return out_f32 ? dispatch_k64<f16, float>(a, tile, stream) : dispatch_k64<f16, f16>(a, tile, stream);
    return YH_EINVAL;
}

}  // namespace yh
