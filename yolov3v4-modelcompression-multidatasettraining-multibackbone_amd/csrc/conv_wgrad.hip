// Weight gradient of the NHWC convolution on MFMA, plus the small backward helpers around it.
//
//   dw[co][ci][r][s] = sum over output pixels p of dz[p][co] * x[pixel p shifted by tap (r,s)][ci]
//
// is a GEMM whose contraction index is the PIXEL, while both operands are stored pixel-major (NHWC).  MFMA wants,
// per lane, 8 (fp16) consecutive K values of one row, so the kernel transposes on the way into LDS: every thread
// loads two 16-byte channel vectors (pixels 2q, 2q+1) and stores them as [channel][pixel] pairs; fragment reads are
// then plain 16-byte rows.  Row pitch 80 bytes makes the fragment reads bank-conflict free (20 dwords * i mod 64 hits
// every 4-bank group once for i = 0..15).  One workgroup owns a 128 (co) x 128 (ci) tile of ONE tap and a contiguous
// range of pixels; partial sums are added to dw with fp32 atomics (a few thousand per workgroup).
#include "common.h"
#include <type_traits>
#include <stdlib.h>

namespace yh {

template <typename T> struct WG;
template <> struct WG<f16> {
    static constexpr int VEC = 8, BK = 32;
    typedef f16x8 vec;
};
template <> struct WG<float> {
    static constexpr int VEC = 4, BK = 16;
    typedef f32x4 vec;
};

__device__ __attribute__((aligned(16))) const uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};  // source of every padded / out-of-range 16-byte load

template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

constexpr int WG_TILE = 128;     // co and ci tile
constexpr int WG_PITCH_DW = 20;  // LDS row pitch in dwords (80 B)

struct WgradArgs {
    yh_wgrad_desc d;
    int tiles_m, tiles_n, ksteps, ksteps_per_split, ncols;  // ncols = kh*kw*cin: the flattened (tap, ci) axis
    int two_stage, dma, bn, bm, cin_w;   // dma: the fp16 LDS-DMA kernel; bn / bm: its column / row tile
    int xcd_splits;                  // > 0: 1-D launch, consecutive pixel splits share an XCD (and its L2)
    int pp;                          // 256 x 256 tile: ping-pong schedule (YH_WGRAD_BIG = 2)
    int halo, h_rows, h_lbw, h_chunks, h_cps;   // 3x3 halo form: staged halo rows, pieces per wave, 512-pixel chunks (per split)
    int rw, rh, qh;                  // bk = qw * wo + rw, qw = qh * ho + rh: per-step pixel advance without divisions
    long pixels;
};

// TM = 16-row MFMA tiles per wave along co: TM 4 -> 128 x 128 tile, TM 2 -> 64 x 128 (layers with <= 64 outputs).
// The N axis of the GEMM is the flattened (tap, ci) index, so a 3x3 layer with 32 inputs fills 128-wide tiles with
// 4 taps each instead of running 9 quarter-empty tiles, and dz is read once per 128 columns, not once per tap.
template <typename T, int TM>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
    typedef typename WG<T>::vec V;
    constexpr int VEC = WG<T>::VEC, BK = WG<T>::BK, PP = BK / 2, BM = TM * 32;
    const yh_wgrad_desc& d = a.d;
    __shared__ uint32_t lds_a2[2][BM * WG_PITCH_DW];      // double-buffered: one barrier per K step
    __shared__ uint32_t lds_b2[2][WG_TILE * WG_PITCH_DW];

    const int tm = blockIdx.x % a.tiles_m, tn = blockIdx.x / a.tiles_m;
    const int co0 = tm * BM, n0 = tn * WG_TILE;
    const int ks0 = blockIdx.y * a.ksteps_per_split;
    const int ks1 = min(ks0 + a.ksteps_per_split, a.ksteps);
    if (ks0 >= ks1) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int cg = tid / PP, pp = tid % PP;       // loader role: channel group, pixel pair
    const int ca = co0 + cg * VEC;
    const bool a_ok = cg * VEC < BM && ca < d.cout;   // cout / cin are the PHYSICAL channel counts of dz / x
    const int nb = n0 + cg * VEC;                     // this thread's column group: (tap, ci..ci+VEC)
    const bool b_ok = nb < a.ncols;
    const int tap_b = b_ok ? nb / d.cin : 0;
    const int cb = nb - tap_b * d.cin;
    const int tr = tap_b / d.kw - d.pad, ts = tap_b % d.kw - d.pad;
    const T* dz = reinterpret_cast<const T*>(d.dz);
    const T* x = reinterpret_cast<const T*>(d.x);

    f32x4 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (n, ho, wo) of this thread's two pixels, advanced by BK pixels per K step without divisions or branches:
    // BK / wo row wraps at most (wrap_w conditional subtracts), one image wrap per row wrap.
    int pn[2], ph[2], pw[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const long p = (long)ks0 * BK + 2 * pp + u;
        const int hw_o = d.ho * d.wo;
        pn[u] = (int)(p / hw_o);
        const int rem = (int)(p - (long)pn[u] * hw_o);
        ph[u] = rem / d.wo;
        pw[u] = rem - ph[u] * d.wo;
    }
    typedef const V __attribute__((address_space(1))) * gvec_ptr;   // plain global loads (a generic pointer would go flat)
    const gvec_ptr zero = (gvec_ptr)(uintptr_t)g_zero16;
    const T* dz_t = dz + ca;              // this thread's channel group
    const T* x_t = x + cb;
    const unsigned npix = (unsigned)a.pixels;

    // Register staging in two sets: the loads of tile ks+2 are issued while tile ks is multiplied, and consumed (stashed
    // to LDS, transposed) one iteration later -- two K steps of latency cover instead of one.  Out-of-range pixels,
    // padding taps and channel tails load from a 16-byte zero page: every load is unconditional.  Element offsets
    // fit 32 bits (checked by the launcher).
    V ra[2][2], rb[2][2];
    auto fetch = [&](auto set, int ks) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned p = (unsigned)ks * BK + 2 * pp + u;
            const bool in = p < npix;
            const int hi = ph[u] * d.stride + tr, wi = pw[u] * d.stride + ts;
            const bool tap_ok = in && b_ok && (unsigned)hi < (unsigned)d.h && (unsigned)wi < (unsigned)d.w_in;
            const unsigned offa = p * (unsigned)d.lddz;
            const unsigned offb = ((unsigned)(pn[u] * d.h + hi) * (unsigned)d.w_in + (unsigned)wi) * (unsigned)d.ldx;
            const gvec_ptr pa = (in && a_ok) ? (gvec_ptr)(uintptr_t)(dz_t + offa) : zero;
            const gvec_ptr pb = tap_ok ? (gvec_ptr)(uintptr_t)(x_t + offb) : zero;
            ra[S][u] = *pa;
            rb[S][u] = *pb;
            // advance by BK pixels: BK = qw * wo + rw and qw = qh * ho + rh (launcher constants), so one compare-and-
            // carry per level replaces the divisions
            int w = pw[u] + a.rw, h = ph[u] + a.rh, n = pn[u] + a.qh;
            const bool cw = w >= d.wo;
            w = cw ? w - d.wo : w;
            h = cw ? h + 1 : h;
            const bool ch = h >= d.ho;
            h = ch ? h - d.ho : h;
            n = ch ? n + 1 : n;
            pw[u] = w; ph[u] = h; pn[u] = n;
        }
    };
    auto stash = [&](auto set, int buf) {
        constexpr int S = decltype(set)::value;
        uint32_t* lds_a = lds_a2[buf];
        uint32_t* lds_b = lds_b2[buf];
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const f16x2 wa = {ra[S][0][e], ra[S][1][e]};
                const f16x2 wb = {rb[S][0][e], rb[S][1][e]};
                if (cg * VEC < BM) lds_a[(cg * VEC + e) * WG_PITCH_DW + pp] = __builtin_bit_cast(uint32_t, wa);
                lds_b[(cg * VEC + e) * WG_PITCH_DW + pp] = __builtin_bit_cast(uint32_t, wb);
            }
        } else {
            typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
            const u32x4_t a0 = __builtin_bit_cast(u32x4_t, ra[S][0]), a1 = __builtin_bit_cast(u32x4_t, ra[S][1]);
            const u32x4_t b0 = __builtin_bit_cast(u32x4_t, rb[S][0]), b1 = __builtin_bit_cast(u32x4_t, rb[S][1]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (cg * 4 < BM) {
                    lds_a[(cg * 4 + e) * WG_PITCH_DW + 2 * pp] = a0[e];
                    lds_a[(cg * 4 + e) * WG_PITCH_DW + 2 * pp + 1] = a1[e];
                }
                lds_b[(cg * 4 + e) * WG_PITCH_DW + 2 * pp] = b0[e];
                lds_b[(cg * 4 + e) * WG_PITCH_DW + 2 * pp + 1] = b1[e];
            }
        }
    };
    auto multiply = [&](int buf) {
        const uint32_t* lds_a = lds_a2[buf];
        const uint32_t* lds_b = lds_b2[buf];
        const int ri = lane & 15, kq = lane >> 4;
        if constexpr (sizeof(T) == 2) {
            f16x8 fa[TM], fb[4];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const f16x8*>(&lds_a[(wm * TM * 16 + i * 16 + ri) * WG_PITCH_DW + kq * 4]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fb[j] = *reinterpret_cast<const f16x8*>(&lds_b[(wn * 64 + j * 16 + ri) * WG_PITCH_DW + kq * 4]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float fa[TM], fb[4];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i] = __builtin_bit_cast(float, lds_a[(wm * TM * 16 + i * 16 + ri) * WG_PITCH_DW + kk * 4 + kq]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    fb[j] = __builtin_bit_cast(float, lds_b[(wn * 64 + j * 16 + ri) * WG_PITCH_DW + kk * 4 + kq]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;

    // prologue: tile ks0 -> LDS buffer 0, tile ks0+1 in flight in set 1
    fetch(S0{}, ks0);
    fetch(S1{}, ks0 + 1);
    stash(S0{}, 0);
    __syncthreads();
    // Two K steps per trip, no conditionals: prefetch tile ks+2 into the set tile ks came from, stash tile ks+1 (loaded
    // one step ago) into the other LDS buffer, multiply tile ks, one barrier.  The step count is even by construction
    // (ksteps_per_split is even; the last split may run one step past the end, which reads the zero page), and tiles
    // fetched or stashed beyond ks1 are simply never multiplied.
    const int ks1e = ks0 + ((ks1 - ks0 + 1) & ~1);
    for (int ks = ks0; ks < ks1e; ks += 2) {
        fetch(S0{}, ks + 2);
        stash(S1{}, 1);
        multiply(0);
        __syncthreads();
        fetch(S1{}, ks + 3);
        stash(S0{}, 0);
        multiply(1);
        __syncthreads();
    }

    if (a.two_stage) {
        // partial tile in MFMA-native order, 16 bytes per lane, fully coalesced; wgrad_reduce_kernel sums the splits
        f32x4* part = reinterpret_cast<f32x4*>(d.ws) + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (TM * 4 * 256);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(i * 4 + j) * 256 + tid] = acc[i][j];
        return;
    }
    // D layout: column (lane & 15) = flattened (tap, ci), row 4*(lane>>4)+reg = co
    const int taps = d.kh * d.kw;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + (lane & 15);
        if (n >= a.ncols) continue;
        const int tap = n / d.cin, ci = n - tap * d.cin;
        if (ci >= a.cin_w) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * TM * 16 + i * 16 + 4 * (lane >> 4) + r;
                if (co < d.cout) atomicAdd(d.dw + ((long)co * a.cin_w + ci) * taps + tap, acc[i][j][r]);
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// fp16 production kernel: same GEMM, but the pixel-major tiles go global -> LDS by LDS-DMA exactly as they lie in memory
// (row = one pixel, 256 B of channels; no VGPR staging, no packing, no ds_write), a 3-stage ring with counted vmcnt and
// one raw barrier per K step like the forward kernel, and the transpose happens on the READ side: a lane's MFMA fragment
// (8 consecutive pixels of one channel) is eight ds_read_u16 at a fixed 256 B stride (immediate offsets, no address math).
// 16 lanes of a fragment read 32 contiguous bytes of one row; the four 8-pixel groups of a wave read rows 2 KB apart, which
// would alias to the same banks, so the 16-byte units of a row are XOR-permuted by 2 * (row >> 3 & 3) on the way in (the
// DMA lane fetches source unit u ^ f for destination cell u) and un-permuted by the reader: conflict free.
// Unit permutation of pixel row r in a row of UNITS 16-byte units.  Fragments are fetched with ds_read_b64_tr_b16: a
// 16-lane group reads a [4 pixels][16 channels] block (32 bytes of 4 consecutive rows) and receives it transposed.  Rows
// are 256 B apart (= all 64 banks), so without a permutation the 4 rows of a block, and the blocks of the two groups that
// share an LDS cycle (rows 8 apart), would all sit on the same 8 banks: the 32-byte column of row r is XORed with
// (r & 3) | ((r >> 3) & 1) << 2, which spreads those 8 row pieces over all 64 banks.  64-channel rows (8 units) only have
// room for the (r & 3) part.
template <int UNITS> __device__ __forceinline__ int wg_swz(int r) {
    return UNITS >= 16 ? (((r & 3) | (((r >> 3) & 1) << 2)) << 1) : ((r & 3) << 1);
}

typedef int wg_v2i __attribute__((ext_vector_type(2)));
template <int OFF> __device__ __forceinline__ wg_v2i ds_read_tr16(unsigned lds_byte_addr) {
    wg_v2i r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_byte_addr), "n"(OFF));
    return r;
}

template <int N> __device__ __forceinline__ void wg_wait_vmcnt() {
#define YH_WG_VMCNT(K) else if constexpr (N == K) asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory")
    if constexpr (N < 0) {}
    YH_WG_VMCNT(0); YH_WG_VMCNT(1); YH_WG_VMCNT(2); YH_WG_VMCNT(3); YH_WG_VMCNT(4); YH_WG_VMCNT(5); YH_WG_VMCNT(6); YH_WG_VMCNT(7); YH_WG_VMCNT(8);
    YH_WG_VMCNT(9); YH_WG_VMCNT(10);
    else static_assert(N < 0, "add the literal");
#undef YH_WG_VMCNT
}

// WNW = waves along the (tap, ci) axis: 2 -> 128 columns on 4 waves, 4 -> 256 columns on 8 waves (24 KB per K step for
// twice the MFMA work of the 16 KB 128 x 128 step: the L2 -> LDS stream is what bounds this kernel).
// PP (256 x 256 tile on 8 waves only): the ping-pong schedule of the forward kernels (conv_igemm_k64.hip) - a K step is two
// phases of 16 MFMAs per wave (channel halves of the wave's 128 rows), each phase = LOAD segment (transposed fragment reads,
// LDS-DMA issue, counted vmcnt), raw barrier, MFMA segment, raw barrier; waves 4 .. 7 run one barrier interval behind waves
// 0 .. 3, so that every SIMD has one wave in its MFMA segment while its partner reads.  Same ring, same fragment layout, same
// summation order as the plain form: results are bit-identical.
template <int TM, int WNW, bool PP = false>
__global__ __launch_bounds__(128 * WNW) void conv_wgrad_dma_kernel(const WgradArgs a) {
    constexpr int BK = 32, BM = TM * 32, BN = WNW * 64, NWAVES = 2 * WNW, NT = 64 * NWAVES, STAGES = 3;
    constexpr int A_UNITS = BM / 8;                    // 16-byte units per A row (16 or 8)
    constexpr int A_ROWS_PER_INSTR = 64 / A_UNITS;     // 4 or 8 pixel rows per LDS-DMA instruction
    constexpr int A_INSTR = BK / A_ROWS_PER_INSTR;     // 8 or 4 per K step
    constexpr int A_PER_WAVE = (A_INSTR + NWAVES - 1) / NWAVES;  // 2, 1 (or 1 on the first A_INSTR waves only)
    constexpr int B_UNITS = BN / 8;                    // 16 or 32 units per B row
    constexpr int B_ROWS_PER_INSTR = 64 / B_UNITS;     // 4 or 2
    constexpr int B_INSTR = BK / B_ROWS_PER_INSTR;     // 8 or 16
    constexpr int B_PER_WAVE = B_INSTR / NWAVES;       // 2
    static_assert(B_INSTR % NWAVES == 0, "B instructions must split evenly over the waves");
    constexpr int GPW = A_PER_WAVE + B_PER_WAVE;
    constexpr int A_BYTES = BK * BM * 2, B_BYTES = BK * BN * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    const yh_wgrad_desc& d = a.d;
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGES * STAGE_BYTES];

    // Workgroups go to the 8 XCDs round-robin by linear id.  All tiles of one pixel split read the same dz / x rows, so
    // each XCD is given a contiguous run of the split-major order (its private L2 then fetches those rows once instead
    // of all 8 L2s fetching them): the same chunked mapping as the forward kernel.
    int tile_id = blockIdx.x, split_id = blockIdx.y;
    if (a.xcd_splits > 0) {
        const int tiles = a.tiles_m * a.tiles_n, nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, rr = nb & 7, xcd = bid & 7;
        const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
        split_id = logical / tiles;
        tile_id = logical - split_id * tiles;
    }
    const int tm = tile_id % a.tiles_m, tn = tile_id / a.tiles_m;
    const int co0 = tm * BM, n0 = tn * BN;
    const int ks0 = split_id * a.ksteps_per_split;
    const int ks1 = min(ks0 + a.ksteps_per_split, a.ksteps);
    if (ks0 >= ks1) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNW, wn = wave % WNW;
    typedef const void __attribute__((address_space(1))) * gptr_t;
    typedef void __attribute__((address_space(3))) * lptr_t;
    const f16* dz = reinterpret_cast<const f16*>(d.dz);
    const f16* x = reinterpret_cast<const f16*>(d.x);
    const f16* zero = reinterpret_cast<const f16*>(g_zero16);
    const unsigned npix = (unsigned)a.pixels;

    // ---- loader roles.  A instruction q covers pixel rows q*A_ROWS_PER_INSTR.., B instruction q rows 4q..4q+3.
    int a_row[A_PER_WAVE];          // pixel row within the K step
    int a_coff[A_PER_WAVE];         // channel offset of the source unit, or -1 when beyond cout
    static_for<A_PER_WAVE>([&](auto c) {
        constexpr int i = decltype(c)::value;
        const int q = wave + NWAVES * i;                 // waves beyond A_INSTR have no A share (a_live below)
        const int row = q * A_ROWS_PER_INSTR + lane / A_UNITS;
        const int us = (lane % A_UNITS) ^ wg_swz<A_UNITS>(row);
        a_row[i] = row;
        a_coff[i] = (co0 + us * 8 < d.cout) ? co0 + us * 8 : -1;
    });
    int b_row[B_PER_WAVE], b_coff[B_PER_WAVE], b_tr[B_PER_WAVE], b_ts[B_PER_WAVE];
    static_for<B_PER_WAVE>([&](auto c) {
        constexpr int i = decltype(c)::value;
        const int q = wave + NWAVES * i;
        const int row = q * B_ROWS_PER_INSTR + lane / B_UNITS;
        const int us = (lane % B_UNITS) ^ wg_swz<16>(row);
        const int nb = n0 + us * 8;
        b_row[i] = row;
        if (nb < a.ncols) {
            const int tap = nb / d.cin;
            b_coff[i] = nb - tap * d.cin;
            b_tr[i] = tap / d.kw - d.pad;
            b_ts[i] = tap % d.kw - d.pad;
        } else {
            b_coff[i] = -1; b_tr[i] = 0; b_ts[i] = 0;
        }
    });
    // pixel coordinates of this lane's rows (A and B rows coincide when A_PER_WAVE == 2; tracked separately otherwise)
    constexpr int NP = A_PER_WAVE + B_PER_WAVE;
    int pn[NP], ph[NP], pw[NP];
    static_for<NP>([&](auto c) {
        constexpr int i = decltype(c)::value;
        const int row = i < A_PER_WAVE ? a_row[i < A_PER_WAVE ? i : 0] : b_row[i >= A_PER_WAVE ? i - A_PER_WAVE : 0];
        const long p = (long)ks0 * BK + row;
        const int hw_o = d.ho * d.wo;
        pn[i] = (int)(p / hw_o);
        const int rem = (int)(p - (long)pn[i] * hw_o);
        ph[i] = rem / d.wo;
        pw[i] = rem - ph[i] * d.wo;
    });

    const bool a_live = wave < A_INSTR || A_PER_WAVE > 1;   // wave-uniform: this wave issues A loads
    int ks_issue = ks0;
    auto issue = [&](int st) {
        unsigned char* const base = smem + st * STAGE_BYTES;
        static_for<A_PER_WAVE>([&](auto c) {
            constexpr int i = decltype(c)::value;
            const unsigned p = (unsigned)ks_issue * BK + a_row[i];
            const bool ok = p < npix && a_coff[i] >= 0;
            const f16* src = ok ? dz + p * (unsigned)d.lddz + a_coff[i] : zero;
            unsigned char* dst = base + (wave + NWAVES * i) * 1024;
            if (a_live) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        });
        static_for<B_PER_WAVE>([&](auto c) {
            constexpr int i = decltype(c)::value;
            constexpr int k = A_PER_WAVE + i;
            const unsigned p = (unsigned)ks_issue * BK + b_row[i];
            const int hi = ph[k] * d.stride + b_tr[i], wi = pw[k] * d.stride + b_ts[i];
            const bool ok = p < npix && b_coff[i] >= 0 && (unsigned)hi < (unsigned)d.h && (unsigned)wi < (unsigned)d.w_in;
            const unsigned off = ((unsigned)(pn[k] * d.h + hi) * (unsigned)d.w_in + (unsigned)wi) * (unsigned)d.ldx + b_coff[i];
            const f16* src = ok ? x + off : zero;
            unsigned char* dst = base + A_BYTES + (wave + NWAVES * i) * 1024;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        });
        // advance every tracked pixel by BK (only the B rows need coordinates; A rows use the flat index)
        static_for<B_PER_WAVE>([&](auto c) {
            constexpr int k = A_PER_WAVE + decltype(c)::value;
            int w = pw[k] + a.rw, h = ph[k] + a.rh, n = pn[k] + a.qh;
            const bool cw = w >= d.wo;
            w = cw ? w - d.wo : w;
            h = cw ? h + 1 : h;
            const bool ch = h >= d.ho;
            h = ch ? h - d.ho : h;
            n = ch ? n + 1 : n;
            pw[k] = w; ph[k] = h; pn[k] = n;
        });
        ++ks_issue;
    };

    f32x4 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment addressing for ds_read_b64_tr_b16.  Lane (q = lane & 15, g = lane >> 4) of a 16-lane group supplies the
    // address of 4 consecutive channels (8 bytes) of pixel row 8g + 4h + q/4 (h = which half of the 8-pixel K group) and
    // receives 4 consecutive pixels of channel (tile base + q): two reads build one MFMA fragment, no packing.
    const int q = lane & 15, g = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
    unsigned a_addr[2], b_addr[2];   // per half h: byte address inside a stage for tile 0; tiles step 32 bytes (16 channels)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 8 * g + 4 * h + (q >> 2);
        const int cha = wm * TM * 16 + 4 * (q & 3), chb = wn * 64 + 4 * (q & 3);
        a_addr[h] = row * (BM * 2) + ((((cha >> 3) ^ wg_swz<A_UNITS>(row)) << 4) | ((cha & 7) * 2));
        b_addr[h] = A_BYTES + row * (BN * 2) + ((((chb >> 3) ^ wg_swz<16>(row)) << 4) | ((chb & 7) * 2));
    }

    const int nk = ks1 - ks0;
    if constexpr (PP) {
        static_assert(!PP || (TM == 8 && WNW == 4), "ping-pong form: 256 x 256 on 8 waves");
        const int group = wave >> 2;
#define YH_WPP_BARRIER()                     \
        do {                                 \
            __builtin_amdgcn_sched_barrier(0);   \
            __builtin_amdgcn_s_barrier();        \
            __builtin_amdgcn_sched_barrier(0);   \
        } while (0)
        issue(0);
        if (nk > 1) { issue(1); wg_wait_vmcnt<GPW>(); } else wg_wait_vmcnt<0>();
        YH_WPP_BARRIER();
        if (group == 1) YH_WPP_BARRIER();   // stagger
        int st_r = 0, st_w = 2;
        for (int kt = 0; kt < nk; ++kt) {
            const unsigned stage = lds0 + st_r * STAGE_BYTES;
            wg_v2i ra[4][2], rb[4][2];
            typedef int v4i __attribute__((ext_vector_type(4)));
            f16x8 fa[4], fb[4];
            // ---- phase X: channel rows 0 .. 63 of the wave tile; the tiles of step kt + 2 go into the stage step kt - 1 was read from
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ra[i][h] = ds_read_tr16<0>(stage + (a_addr[h] ^ (i << 5)));
#pragma unroll
                for (int j = 0; j < 4; ++j) rb[j][h] = ds_read_tr16<0>(stage + (b_addr[h] ^ (j << 5)));
            }
            if (kt + 2 < nk) issue(st_w);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(ra[2][0]), "+v"(ra[2][1]),
                           "+v"(ra[3][0]), "+v"(ra[3][1]), "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[1][0]), "+v"(rb[1][1]),
                           "+v"(rb[2][0]), "+v"(rb[2][1]), "+v"(rb[3][0]), "+v"(rb[3][1])
                         :
                         : "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const v4i t = {ra[i][0][0], ra[i][0][1], ra[i][1][0], ra[i][1][1]};
                fa[i] = __builtin_bit_cast(f16x8, t);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v4i t = {rb[j][0][0], rb[j][0][1], rb[j][1][0], rb[j][1][1]};
                fb[j] = __builtin_bit_cast(f16x8, t);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            YH_WPP_BARRIER();
            // ---- phase Y: channel rows 64 .. 127; this wave's share of step kt + 1 must have landed before the barrier
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) ra[i][h] = ds_read_tr16<0>(stage + (a_addr[h] ^ ((i + 4) << 5)));
            if (kt + 2 < nk) wg_wait_vmcnt<GPW>(); else wg_wait_vmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(ra[2][0]), "+v"(ra[2][1]),
                           "+v"(ra[3][0]), "+v"(ra[3][1])
                         :
                         : "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const v4i t = {ra[i][0][0], ra[i][0][1], ra[i][1][0], ra[i][1][1]};
                fa[i] = __builtin_bit_cast(f16x8, t);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[4 + i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            YH_WPP_BARRIER();
            st_r = st_r + 1 == STAGES ? 0 : st_r + 1;
            st_w = st_w + 1 == STAGES ? 0 : st_w + 1;
        }
        if (group == 0) YH_WPP_BARRIER();
#undef YH_WPP_BARRIER
    } else {
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) issue(s);
    int st_read = 0, st_write = STAGES - 1;
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = min(STAGES - 2, nk - 1 - kt);
        if (ahead >= 1) {
            if (a_live) wg_wait_vmcnt<GPW>(); else wg_wait_vmcnt<B_PER_WAVE>();
        } else {
            wg_wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        if (kt + STAGES - 1 < nk) issue(st_write);
        const unsigned stage = lds0 + st_read * STAGE_BYTES;
        // tile i of the wave sits 16 channels = 32 bytes further; the XOR permutation acts on bits >= 5 of the in-row
        // offset for a fixed row, so stepping tiles is an XOR with i << 5 (folded into the address, offset field 0)
        wg_v2i ra[TM][2], rb[4][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < TM; ++i) ra[i][h] = ds_read_tr16<0>(stage + (a_addr[h] ^ (i << 5)));
#pragma unroll
            for (int j = 0; j < 4; ++j) rb[j][h] = ds_read_tr16<0>(stage + (b_addr[h] ^ (j << 5)));
        }
        // the reads are asynchronous and invisible to the compiler's waitcnt insertion: wait here, and thread every
        // fragment through the asm so that no consumer can be scheduled above it
        if constexpr (TM == 8) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(ra[2][0]), "+v"(ra[2][1]),
                           "+v"(ra[3][0]), "+v"(ra[3][1]), "+v"(ra[4][0]), "+v"(ra[4][1]), "+v"(ra[5][0]), "+v"(ra[5][1]),
                           "+v"(ra[6][0]), "+v"(ra[6][1]), "+v"(ra[7][0]), "+v"(ra[7][1]), "+v"(rb[0][0]), "+v"(rb[0][1]),
                           "+v"(rb[1][0]), "+v"(rb[1][1]), "+v"(rb[2][0]), "+v"(rb[2][1]), "+v"(rb[3][0]), "+v"(rb[3][1])
                         :
                         : "memory");
        } else if constexpr (TM == 4) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(ra[2][0]), "+v"(ra[2][1]),
                           "+v"(ra[3][0]), "+v"(ra[3][1]), "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[1][0]), "+v"(rb[1][1]),
                           "+v"(rb[2][0]), "+v"(rb[2][1]), "+v"(rb[3][0]), "+v"(rb[3][1])
                         :
                         : "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(rb[0][0]), "+v"(rb[0][1]),
                           "+v"(rb[1][0]), "+v"(rb[1][1]), "+v"(rb[2][0]), "+v"(rb[2][1]), "+v"(rb[3][0]), "+v"(rb[3][1])
                         :
                         : "memory");
        }
        f16x8 fa[TM], fb[4];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            typedef int v4i __attribute__((ext_vector_type(4)));
            const v4i t = {ra[i][0][0], ra[i][0][1], ra[i][1][0], ra[i][1][1]};
            fa[i] = __builtin_bit_cast(f16x8, t);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            typedef int v4i __attribute__((ext_vector_type(4)));
            const v4i t = {rb[j][0][0], rb[j][0][1], rb[j][1][0], rb[j][1][1]};
            fb[j] = __builtin_bit_cast(f16x8, t);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        st_read = st_read + 1 == STAGES ? 0 : st_read + 1;
        st_write = st_write + 1 == STAGES ? 0 : st_write + 1;
    }
    }   // !PP

    if (a.two_stage) {
        f32x4* part = reinterpret_cast<f32x4*>(d.ws) + ((long)split_id * (a.tiles_m * a.tiles_n) + tile_id) * (TM * 4 * NT);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(i * 4 + j) * NT + tid] = acc[i][j];
        return;
    }
    const int taps = d.kh * d.kw;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + (lane & 15);
        if (n >= a.ncols) continue;
        const int tap = n / d.cin, ci = n - tap * d.cin;
        if (ci >= a.cin_w) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * TM * 16 + i * 16 + 4 * (lane >> 4) + r;
                if (co < d.cout) atomicAdd(d.dw + ((long)co * a.cin_w + ci) * taps + tap, acc[i][j][r]);
            }
    }
}

// Second stage: one thread per (tile, i, j, lane slot) and split group: sum the group's splits, add into dw[co][ci][tap]
// (a handful of groups per element, so these atomics are uncontended).
template <int TM, int WNW>
__global__ __launch_bounds__(128 * WNW) void wgrad_reduce_kernel(const WgradArgs a, int splits, int per_group, int sstep, int native) {
    constexpr int NT = 128 * WNW, BN = 64 * WNW;
    const yh_wgrad_desc& d = a.d;
    const int tiles = a.tiles_m * a.tiles_n;
    const int tile = blockIdx.x / (TM * 4), ij = blockIdx.x % (TM * 4);
    const int i = ij / 4, j = ij % 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WNW, wn = wave % WNW;
    f32x4* part = reinterpret_cast<f32x4*>(d.ws) + ((long)tile * (TM * 4) + ij) * NT + tid;
    const long stride = (long)tiles * (TM * 4) * NT * sstep;      // sstep > 1: the second pass, over the groups' in-place sums
    const int s0 = blockIdx.y * per_group, s1 = min(s0 + per_group, splits);
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0, v2 = v0, v3 = v0;
    int sp = s0;
    for (; sp + 3 < s1; sp += 4) {   // four independent chains: the loads pipeline
        v0 += part[sp * stride];
        v1 += part[(sp + 1) * stride];
        v2 += part[(sp + 2) * stride];
        v3 += part[(sp + 3) * stride];
    }
    for (; sp < s1; ++sp) v0 += part[sp * stride];
    const f32x4 v = (v0 + v1) + (v2 + v3);
    if (native) {      // deterministic form, first pass: the group's sum replaces its first partial tile (same element order)
        part[s0 * stride] = v;
        return;
    }
    const int tm = tile % a.tiles_m, tn = tile / a.tiles_m;
    const int n = tn * BN + wn * 64 + j * 16 + (lane & 15);
    if (n >= a.ncols) return;
    const int taps = d.kh * d.kw;
    const int tap = n / d.cin, ci = n - tap * d.cin;
    if (ci >= a.cin_w) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = tm * (TM * 32) + wm * TM * 16 + i * 16 + 4 * (lane >> 4) + r;
        if (co >= d.cout) continue;
        float* dst = d.dw + ((long)co * a.cin_w + ci) * taps + tap;
        if (gridDim.y == 1) *dst += v[r];     // a single split group owns the element: plain accumulate
        else atomicAdd(dst, v[r]);
    }
}


// ------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 weight gradient, "halo" form (fp16, cout % 256 == 0, cin % 32 == 0).
//
// The LDS-DMA kernel above is bound by global -> LDS delivery (round 2: 48 KB per 1024 MFMA cycles of a CU against the ~20 B/clk
// a CU is handed): every (tap, ci) column tile fetches its own shifted copy of the x rows, nine times per layer, and dz once per
// column tile.  Here ONE workgroup owns a 256 (co) x [9 taps x 32 ci] tile, so that
//   * the x rows of a 256-pixel chunk (plus one image row and one pixel either side) are staged ONCE as a halo image in the
//     virtual pixel space of conv_halo_pp.hip (one shared pad row / column: tap (r, s) = row offset r (W+1) + s) and all nine
//     taps read it as shifted views - through ds_read_b64_tr_b16, which takes any row offset;
//   * dz streams through the 3-stage ring in 32-pixel steps, in virtual pixel order (pad positions read the zero page): 16 KB
//     per step for 256 x 288 x 32 MACs = 16 B/clk at the MFMA rate, below the delivery wall (5-stage ring: four steps in flight).
// 12 waves (3 per SIMD): wave (wm, wn) owns channels 64 wm .. +63 x the three taps of filter row wn x 32 ci = 4 x 6 fragments
// (96 accumulator registers), 20 transposed fragment reads per 24 MFMAs.  Pixel splits as above: partial tiles in the
// workspace in MFMA-native order, summed by wgrad_halo_reduce_kernel.  LDS rows of the halo image are 64 B (4 cells); the two
// 32-byte halves of row r are swapped when bit 3 of r is set, so that the 4-row blocks of the two lane groups that share an LDS
// cycle (rows 8 apart) never meet in a bank, for any tap offset.
// ABL (profiling only, results are garbage): 1 = no MFMAs, 2 = no fragment reads, 3 = no LDS-DMA
template <int LBW, int ABL = 0>
__global__ __launch_bounds__(768, 3) void conv_wgrad_halo_kernel(const WgradArgs a) {
    constexpr int BM = 256, KP = 256, BK = 32, NWAVES = 12, NT = 64 * NWAVES, STAGES = 5, SUBS = KP / BK;
    constexpr int A_BYTES = BK * BM * 2;       // one dz step: 32 pixel rows x 512 B
    const yh_wgrad_desc& d = a.d;
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];   // the only LDS object
    const int hbytes = a.h_rows * 64;
    typedef const void __attribute__((address_space(1))) * gptr_t;
    typedef void __attribute__((address_space(3))) * lptr_t;

    const int tiles = a.tiles_m * a.tiles_n;
    int tile_id, split_id;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, rr = nb & 7, xcd = bid & 7;
        const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
        split_id = logical / tiles;
        tile_id = logical - split_id * tiles;
    }
    const int tm = tile_id % a.tiles_m, tn = tile_id / a.tiles_m;
    const int co0 = tm * BM, ci0 = tn * 32;
    const int c0 = split_id * a.h_cps, c1 = min(c0 + a.h_cps, a.h_chunks);
    if (c0 >= c1) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / 3, wn = wave - wm * 3;
    const f16* dz = reinterpret_cast<const f16*>(d.dz);
    const f16* x = reinterpret_cast<const f16*>(d.x);
    // the zero page's address as two SCALARS: a pointer select against it then costs no vector registers (the kernel sits at the
    // 168-register cap; a spilled pointer would be reloaded by a scratch load, which drains the LDS-DMA queue - same counter)
    const unsigned zlo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)g_zero16);
    const unsigned zhi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)g_zero16 >> 32));
    auto pick = [&](bool ok, const f16* p) {
        const unsigned long long u = (unsigned long long)(uintptr_t)p;
        const unsigned lo = ok ? (unsigned)u : zlo, hi = ok ? (unsigned)(u >> 32) : zhi;
        return ((unsigned long long)hi << 32) | lo;      // (an integer: a lambda returning an address-space pointer loses the host stub)
    };
    const int Wp = d.w_in + 1, IMG = (d.h + 1) * Wp;

    // ---- pixel table.  Every virtual position this workgroup touches - its 256-pixel chunks plus a halo margin of W + 2 either
    // side - is decoded ONCE into a 16-bit entry in LDS: real pixel index minus p_base, or 0xffff for a pad position / beyond the
    // batch.  The LDS-DMA bookkeeping of a step is then a table read and a multiply per instruction instead of ~40 instructions
    // of wrap arithmetic: a wave issues an instruction every ~4-5 cycles, and 150 instructions of bookkeeping per step in ONE
    // group's interval cost 3 x 1200 cycles per step where the MFMAs need 3 x 384 (measured, profiles/r03_wgrad_halo.txt).
    unsigned short* const tbl = reinterpret_cast<unsigned short*>(hsm + STAGES * A_BYTES + 2 * hbytes + 1024);
    const int vstart = c0 * KP - Wp - 1;                       // virtual position of table entry 0
    const int tbl_n = (c1 - c0) * KP + 2 * Wp + 2 + 16;        // + the rows that round the last halo image up to 16
    int p_base;                                                // wave-uniform: first pixel of the image row vstart lies in
    {
        const int v = max(vstart, 0);
        const int n = v / IMG;
        const int yy = (v - n * IMG) / Wp;
        p_base = (n * d.h + max(yy - 1, 0)) * d.w_in;
    }
    for (int t = tid; t < tbl_n; t += NT) {
        const int v = vstart + t;
        unsigned short e = 0xffffu;
        if (v >= 0 && t < tbl_n - 16) {
            const int n = v / IMG;
            const int rem = v - n * IMG;
            const int yy = rem / Wp, xx = rem - yy * Wp;
            if (n < d.n && yy >= 1 && xx >= 1) e = (unsigned short)((n * d.h + yy - 1) * d.w_in + xx - 1 - p_base);
        }
        tbl[t] = e;
    }
    __syncthreads();       // no LDS-DMA is in flight yet: a plain barrier
    const unsigned tbl_lds = (unsigned)(uintptr_t)(lptr_t)tbl;
    auto lookup = [&](int idx) {       // table entry idx of this lane (inline asm: the compiler must not order it against the DMA queue)
        unsigned e;
        asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(tbl_lds + 2u * (unsigned)idx) : "memory");
        return e;
    };

    // ---- dz loader: instruction q of a step covers pixel rows 2q, 2q+1 (512 B each); wave w issues q = w and, w < 4, q = w + 12
    // Roles: waves 0 .. 7 stream dz (two instructions each per step), waves 8 .. 11 fetch the halo images - the counter that
    // orders a wave's LDS-DMA returns is in-order, and a 16-row halo piece (16 scattered half lines) in front of the dz tiles
    // held their counted wait up
    const bool dz_wave = wave < 8;                                                   // wave-uniform
    int dz_idx = Wp + 1 + 2 * wave + (lane >> 5);       // table index of this lane's row in the step being issued
    auto issue_dz = [&](int st) {
        unsigned char* const base = hsm + st * A_BYTES;
        const int hi = lane >> 5;
        {
            const unsigned e = lookup(dz_idx);
            const unsigned off = (unsigned)(p_base + (int)e) * (unsigned)d.lddz + (unsigned)(co0 + (((lane & 31) ^ wg_swz<32>(2 * wave + hi)) << 3));
            __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)pick(e != 0xffffu, dz + off), (lptr_t)(base + wave * 1024), 16, 0, 0);
        }
        {
            const unsigned e = lookup(dz_idx + 16);
            const unsigned off = (unsigned)(p_base + (int)e) * (unsigned)d.lddz + (unsigned)(co0 + (((lane & 31) ^ wg_swz<32>(2 * (wave + 8) + hi)) << 3));
            __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)pick(e != 0xffffu, dz + off), (lptr_t)(base + (wave + 8) * 1024), 16, 0, 0);
        }
        dz_idx += BK;
    };
    // ---- halo loader: piece g = 16 rows x 64 B, lane -> row g 16 + (lane >> 2), LDS cell lane & 3 holding source cell
    // (lane & 3) ^ 2 ((row >> 3) & 1); wave 8 + w issues pieces w, w + 4, ...; halo row j of chunk c is table entry (c - c0) KP + j
    auto issue_halo = [&](int chunk, int i) {
        const int h_cell = ((lane & 3) ^ (((lane >> 5) & 1) << 1)) * 8;
        const int g = (wave - 8) + 4 * i;                     // wave-uniform
        const int j = g * 16 + (lane >> 2);
        const bool in = g * 16 < a.h_rows;                    // wave-uniform: h_rows is a multiple of 16
        const unsigned e = in ? lookup((chunk - c0) * KP + j) : 0xffffu;
        const unsigned off = (unsigned)(p_base + (int)e) * (unsigned)d.ldx + (unsigned)(ci0 + h_cell);
        unsigned char* dst = in ? hsm + STAGES * A_BYTES + (chunk & 1) * hbytes + g * 1024 : hsm + STAGES * A_BYTES + 2 * hbytes;
        __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)pick(e != 0xffffu, x + off), (lptr_t)dst, 16, 0, 0);
    };

    f32x4 acc[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment addressing (ds_read_b64_tr_b16: lane (q, g) supplies 4 consecutive channels of pixel row 8g + 4h + q/4 and
    // receives 4 consecutive pixels of channel q of the 16-channel block)
    const int q = lane & 15, g = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)hsm;
    unsigned a_addr0 = 0, b_addr[3][2];     // dz fragments: half 1 = half 0 + 4 rows (same permutation): immediate offset 2048
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 8 * g + 4 * h + (q >> 2);
        const int cha = wm * 64 + 4 * (q & 3);
        if (h == 0) a_addr0 = row * (BM * 2) + ((((cha >> 3) ^ wg_swz<32>(row)) << 4) | ((cha & 7) * 2));
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int hrow = row + wn * Wp + s;                 // tap (wn, s) of this wave
            const int bit = (hrow >> 3) & 1;
            // channel block 0 of the row; block 1 is the other 32-byte half (address ^ 32)
            b_addr[s][h] = STAGES * A_BYTES + hrow * 64 + (((q & 3) >> 1) << 4) + ((q & 3) & 1) * 8 + 32 * bit;
        }
    }

    // ---- K loop.  The 12 waves are three groups of one wave per SIMD (waves 0-3, 4-7, 8-11) that rotate through three roles,
    // one barrier interval each: LOAD (20 transposed fragment reads + this wave's LDS-DMA issues + the counted wait), MFMA (24
    // MFMAs), idle.  Group g runs g intervals behind group 0, so in every interval one group feeds the matrix pipes while
    // another one reads: measured on the lock-step form (one barrier per step, everyone reads, then everyone multiplies), the
    // read phase and the MFMA phase simply added up (MFMA busy 38 %, profiles/r03_wgrad_halo_ablation.txt).
    const int nsteps = (c1 - c0) * SUBS;
    const int nA = 2;
    const int grp = a.pp ? 0 : (wave >> 2);      // a.pp (profiling knob YH_WGRAD_HALO_NOSTAGGER): all groups in phase
    auto wait_keep = [&](int keep) {      // counted wait with a run-time count: literal operands only
        switch (keep) {
            case 10: wg_wait_vmcnt<10>(); break;
            case 9: wg_wait_vmcnt<9>(); break;
            case 8: wg_wait_vmcnt<8>(); break;
            case 7: wg_wait_vmcnt<7>(); break;
            case 6: wg_wait_vmcnt<6>(); break;
            case 5: wg_wait_vmcnt<5>(); break;
            case 4: wg_wait_vmcnt<4>(); break;
            case 3: wg_wait_vmcnt<3>(); break;
            case 2: wg_wait_vmcnt<2>(); break;
            case 1: wg_wait_vmcnt<1>(); break;
            default: wg_wait_vmcnt<0>(); break;
        }
    };
#define YH_WH_BARRIER()                      \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)
    if constexpr (ABL != 3) {
        if (dz_wave) {
            for (int k = 0; k < STAGES - 1 && k < nsteps; ++k) issue_dz(k);
            // steps 0 and 1 have landed when only the tiles of steps 2, 3 are in flight
            wait_keep((nsteps > STAGES - 1 ? STAGES - 3 : max(0, nsteps - 2)) * nA);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) issue_halo(c0, i);
            wait_keep(0);
        }
    }
    YH_WH_BARRIER();
    for (int k = 0; k < grp; ++k) YH_WH_BARRIER();      // stagger
    int st_read = 0, st_write = STAGES - 1;
    for (int s = 0; s < nsteps; ++s) {
        const int sub = s % SUBS, chunk = c0 + s / SUBS;
        // ---- LOAD
        const unsigned stage = lds0 + st_read * A_BYTES;
        const unsigned hb = lds0 + (chunk & 1) * hbytes + sub * (BK * 64);
        wg_v2i ra[4][2], rb[6][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i][0] = ABL == 2 ? wg_v2i{(int)(stage + i), 0} : ds_read_tr16<0>(stage + (a_addr0 ^ (i << 5)));
            ra[i][1] = ABL == 2 ? wg_v2i{(int)(stage + i), 1} : ds_read_tr16<2048>(stage + (a_addr0 ^ (i << 5)));
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                rb[2 * sx][h] = ABL == 2 ? wg_v2i{(int)hb, sx} : ds_read_tr16<0>(hb + b_addr[sx][h]);
                rb[2 * sx + 1][h] = ABL == 2 ? wg_v2i{(int)hb + 1, sx} : ds_read_tr16<0>(hb + (b_addr[sx][h] ^ 32));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(ra[2][0]), "+v"(ra[2][1]),
                       "+v"(ra[3][0]), "+v"(ra[3][1]), "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[1][0]), "+v"(rb[1][1]),
                       "+v"(rb[2][0]), "+v"(rb[2][1]), "+v"(rb[3][0]), "+v"(rb[3][1]), "+v"(rb[4][0]), "+v"(rb[4][1]),
                       "+v"(rb[5][0]), "+v"(rb[5][1])
                     :
                     : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA
        typedef int v4i __attribute__((ext_vector_type(4)));
        f16x8 fa[4], fb[6];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const v4i t = {ra[i][0][0], ra[i][0][1], ra[i][1][0], ra[i][1][1]};
            fa[i] = __builtin_bit_cast(f16x8, t);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const v4i t = {rb[j][0][0], rb[j][0][1], rb[j][1][0], rb[j][1][1]};
            fb[j] = __builtin_bit_cast(f16x8, t);
        }
        if constexpr (ABL == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
            for (int j = 0; j < 6; ++j) asm volatile("" ::"v"(fb[j]));
        } else {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        YH_WH_BARRIER();
        // ---- third interval (the other two groups read / multiply): this wave's LDS-DMA bookkeeping
        if constexpr (ABL != 3) {
            // tiles of step s + 5 into the stage step s - 1 was read from (its last reader, group 2, finished three intervals ago).
            // Four steps (64 KB) stay in flight per CU: dz streams from HBM, and 16 B/clk at ~2 us of latency needs about that much
            if (dz_wave) {
                // tiles of step s + 4 into the stage step s - 1 was read from (its last reader, group 2, finished three intervals
                // ago).  (Issuing them in the LOAD segment instead - two more intervals to land - measured slower: the segment
                // that the other groups' MFMAs have to cover grows by ~25 instructions.)
                if (s + STAGES - 1 < nsteps) issue_dz(st_write);
                // this wave's share of step s + 2 (and everything older) has landed when only the tiles of steps s + 3, s + 4 are
                // still in flight: group 1 waits here two intervals before group 0 reads step s + 2, the barriers in between
                // publish it.  (Letting group 0 wait for step s + 1 only - it reads it in the next interval - measured 6 % slower.)
                wait_keep(min(STAGES - 3, max(0, nsteps - 3 - s)) * nA);
            } else if (chunk + 1 < c1) {
                // the next chunk's halo image: two pieces per wave in each of the chunk's first four steps, complete (and published
                // by the following barriers) well before the chunk's last step ends
                if (sub < 4) { issue_halo(chunk + 1, 2 * sub); issue_halo(chunk + 1, 2 * sub + 1); }
                if (sub == 6) wait_keep(0);
            }
        }
        YH_WH_BARRIER();
        st_read = st_read + 1 == STAGES ? 0 : st_read + 1;
        st_write = st_write + 1 == STAGES ? 0 : st_write + 1;
    }
    for (int k = grp; k < 2; ++k) YH_WH_BARRIER();      // every wave has executed the same number of barriers
#undef YH_WH_BARRIER

    f32x4* part = reinterpret_cast<f32x4*>(d.ws) + ((long)split_id * tiles + tile_id) * (24 * NT);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) part[(i * 6 + j) * NT + tid] = acc[i][j];
}

// one thread per (tile, fragment, lane slot) and split group, as wgrad_reduce_kernel
__global__ __launch_bounds__(768) void wgrad_halo_reduce_kernel(const WgradArgs a, int splits, int per_group, int sstep, int native) {
    constexpr int NT = 768;
    const yh_wgrad_desc& d = a.d;
    const int tiles = a.tiles_m * a.tiles_n;
    const int tile = blockIdx.x / 24, ij = blockIdx.x % 24;
    const int i = ij / 6, j = ij % 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / 3, wn = wave - wm * 3;
    f32x4* part = reinterpret_cast<f32x4*>(d.ws) + ((long)tile * 24 + ij) * NT + tid;
    const long stride = (long)tiles * 24 * NT * sstep;
    const int s0 = blockIdx.y * per_group, s1 = min(s0 + per_group, splits);
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0, v2 = v0, v3 = v0;
    int sp = s0;
    for (; sp + 3 < s1; sp += 4) {
        v0 += part[sp * stride];
        v1 += part[(sp + 1) * stride];
        v2 += part[(sp + 2) * stride];
        v3 += part[(sp + 3) * stride];
    }
    for (; sp < s1; ++sp) v0 += part[sp * stride];
    const f32x4 v = (v0 + v1) + (v2 + v3);
    if (native) {      // deterministic form, first pass: the group's sum replaces its first partial tile (same element order)
        part[s0 * stride] = v;
        return;
    }
    const int tm = tile % a.tiles_m, tn = tile / a.tiles_m;
    const int tap = wn * 3 + (j >> 1);
    const int ci = tn * 32 + (j & 1) * 16 + (lane & 15);
    if (ci >= a.cin_w) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = tm * 256 + wm * 64 + i * 16 + 4 * (lane >> 4) + r;
        if (co >= d.cout) continue;
        float* dst = d.dw + ((long)co * a.cin_w + ci) * 9 + tap;
        if (gridDim.y == 1) *dst += v[r];
        else atomicAdd(dst, v[r]);
    }
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* x, T* y, int n, int c, int h, int w, int c_pad, int ldy) {
    const long total = (long)n * h * w;
    const long plane = (long)h * w;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const long img = p / plane, off = p - img * plane;
        for (int ch = 0; ch < c_pad; ++ch) y[p * ldy + ch] = ch < c ? (T)x[(img * c + ch) * plane + off] : (T)0;
    }
}

// First layer: x is the fp32 NCHW image with 3 channels, 3x3 taps.  Thread = (co, pixel lane); 27 accumulators.
template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const yh_wgrad_desc d, long pixels, int chunk) {
    constexpr int CIN = 3, KK = 3;
    const int co = threadIdx.x % d.cout;
    const int lanes = 256 / d.cout;
    const int pl = threadIdx.x / d.cout;
    const T* dz = reinterpret_cast<const T*>(d.dz);
    const float* x = reinterpret_cast<const float*>(d.x);
    float acc[CIN * KK * KK];
#pragma unroll
    for (int i = 0; i < CIN * KK * KK; ++i) acc[i] = 0.f;
    const long p0 = (long)blockIdx.x * chunk, p1 = min(p0 + chunk, pixels);
    const int hw_o = d.ho * d.wo;
    const long plane = (long)d.h * d.w_in;
    if (pl < lanes)
        for (long p = p0 + pl; p < p1; p += lanes) {
            const float g = (float)dz[p * d.lddz + co];
            const int n = (int)(p / hw_o);
            const int rem = (int)(p - (long)n * hw_o);
            const int ho = rem / d.wo, wo = rem - ho * d.wo;
            const float* xn = x + (long)n * CIN * plane;
#pragma unroll
            for (int c = 0; c < CIN; ++c)
#pragma unroll
                for (int r = 0; r < KK; ++r)
#pragma unroll
                    for (int s = 0; s < KK; ++s) {
                        const int hi = ho * d.stride + r - d.pad, wi = wo * d.stride + s - d.pad;
                        const bool ok = (unsigned)hi < (unsigned)d.h && (unsigned)wi < (unsigned)d.w_in;
                        const float xv = ok ? xn[c * plane + (long)hi * d.w_in + wi] : 0.f;
                        acc[(c * KK + r) * KK + s] = fmaf(g, xv, acc[(c * KK + r) * KK + s]);
                    }
        }
    // reduce the pixel lanes through LDS, then one atomic per (co, ci, r, s) per workgroup
    __shared__ float red[256];
#pragma unroll
    for (int i = 0; i < CIN * KK * KK; ++i) {
        red[threadIdx.x] = (pl < lanes) ? acc[i] : 0.f;
        __syncthreads();
        if (threadIdx.x < d.cout) {
            float v = 0.f;
            for (int q = 0; q < lanes; ++q) v += red[q * d.cout + threadIdx.x];
            atomicAdd(d.dw + (long)threadIdx.x * (CIN * KK * KK) + i, v);
        }
        __syncthreads();
    }
}

template <typename T>
__global__ void pack_dgrad_kernel(const float* w, int cout, int cin, int kh, int kw, int cout_k, int m_pad, T* packed) {
    const long total = (long)m_pad * kh * kw * cout_k;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout_k);
        long r = i / cout_k;
        const int tap = (int)(r % (kh * kw));
        const int ci = (int)(r / (kh * kw));
        float v = 0.f;
        if (ci < cin && co < cout) {
            const int fr = kh - 1 - tap / kw, fs = kw - 1 - tap % kw;
            v = w[(((long)co * cin + ci) * kh + fr) * kw + fs];
        }
        packed[i] = (T)v;
    }
}

// one phase of the stride-2 data gradient: packed[ci][t*kwp + u][co] = w[co][ci][a + pad - 2t][b + pad - 2u]
template <typename T>
__global__ void pack_dgrad_phase_kernel(const float* w, int cout, int cin, int kh, int kw, int pad, int pa, int pb, int khp,
                                        int kwp, int cout_k, int m_pad, T* packed) {
    const long total = (long)m_pad * khp * kwp * cout_k;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout_k);
        long r = i / cout_k;
        const int tap = (int)(r % (khp * kwp));
        const int ci = (int)(r / (khp * kwp));
        const int fr = pa + pad - 2 * (tap / kwp), fs = pb + pad - 2 * (tap % kwp);
        float v = 0.f;
        if (ci < cin && co < cout && fr >= 0 && fr < kh && fs >= 0 && fs < kw) v = w[(((long)co * cin + ci) * kh + fr) * kw + fs];
        packed[i] = (T)v;
    }
}

// every weight image of the step: blockIdx.y = item, blockIdx.x strides over the destination elements (gather form: padding
// rows / columns are written as zeros, no memset)
template <typename T>
__device__ __forceinline__ void pack_item_elems(const yh_pack_item& it) {
    T* out = reinterpret_cast<T*>(it.packed);
    const int taps = it.kh * it.kw;
    if (it.mode == 0) {
        const long total = (long)it.m_pad * taps * it.k_pad;
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const int c = (int)(i % it.k_pad);
            const long r = i / it.k_pad;
            const int tap = (int)(r % taps), m = (int)(r / taps);
            out[i] = (m < it.cout && c < it.cin) ? (T)it.w[((long)m * it.cin + c) * taps + tap] : (T)0.f;
        }
    } else if (it.mode == 5) {
        // all four parity phases of a stride-2 data gradient: row = phase * cout_pad + ci, taps (t, u) of a 2x2 window
        const long total = (long)it.m_pad * 4 * it.k_pad;
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const int co = (int)(i % it.k_pad);
            const long r = i / it.k_pad;
            const int tap = (int)(r & 3), m = (int)(r >> 2);
            const int ph = m / it.cout_pad, ci = m - ph * it.cout_pad;
            const int fr = (ph >> 1) + it.pad - 2 * (tap >> 1), fs = (ph & 1) + it.pad - 2 * (tap & 1);
            float v = 0.f;
            if (ph < 4 && ci < it.cin && co < it.cout && fr >= 0 && fr < it.kh && fs >= 0 && fs < it.kw)
                v = it.w[(((long)co * it.cin + ci) * it.kh + fr) * it.kw + fs];
            out[i] = (T)v;
        }
    } else {
        const int khp = it.mode == 2 ? (it.pa + it.pad) / 2 + 1 : it.kh, kwp = it.mode == 2 ? (it.pb + it.pad) / 2 + 1 : it.kw;
        const long total = (long)it.m_pad * khp * kwp * it.k_pad;
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const int co = (int)(i % it.k_pad);
            const long r = i / it.k_pad;
            const int tap = (int)(r % (khp * kwp)), ci = (int)(r / (khp * kwp));
            int fr, fs;
            if (it.mode == 2) { fr = it.pa + it.pad - 2 * (tap / kwp); fs = it.pb + it.pad - 2 * (tap % kwp); }
            else { fr = it.kh - 1 - tap / it.kw; fs = it.kw - 1 - tap % it.kw; }
            float v = 0.f;
            if (ci < it.cin && co < it.cout && fr >= 0 && fr < it.kh && fs >= 0 && fs < it.kw)
                v = it.w[(((long)co * it.cin + ci) * it.kh + fr) * it.kw + fs];
            out[i] = (T)v;
        }
    }
}

// Tiled form of the dense images (modes 0, 1, 2; 1x1 and 3x3 filters; round 5).  The gather form above reads the fp32 parameter with
// the stride of the DESTINATION order: consecutive threads of a data-gradient image are consecutive output channels, 4 bytes from
// 64 different lines (cin * taps floats apart) per wave instruction - 0.97 - 1.9 GB fetched to repack 248 MB of weights
// (profiles/r04_rocprof_pmc_train.txt), 0.445 ms per step.  Here a workgroup moves a tile [64 rows of the SOURCE's slow axis][E = CI x
// taps contiguous floats] through LDS: rows are read as they lie (288- / 256-byte runs), the image is written in 128-byte runs.
//   mode 0: source row = output channel m, run = 8 (1x1: 64) input channels x taps; image row (m, tap) gets 8 / 64 consecutive c
//           -> tile [8 m][64 c x taps] instead: for the forward image the contiguous axis of BOTH sides is c, so rows = m, E = 64 taps
//   mode 1 / 2: source row = output channel co, run = CI input channels x taps; image row (ci, tap') gets 64 consecutive co
constexpr int PK_PITCH = 73;      // floats per LDS row of the [64][72] tile (odd: the transposed reads walk rows conflict-free)
template <typename T>
__device__ __forceinline__ void pack_item_tiled(const yh_pack_item& it, float* lds) {
    T* out = reinterpret_cast<T*>(it.packed);
    const int taps = it.kh * it.kw, tid = threadIdx.x;
    if (it.mode == 0) {
        // tile: 8 output channels x 64 input channels (x taps): lds[m_l][c_l * taps + tap], pitch 64 * taps + 1
        const int pitch = 64 * taps + 1;
        const int ct = (it.k_pad + 63) / 64, mt = (it.m_pad + 7) / 8;
        for (int tile = blockIdx.x; tile < ct * mt; tile += gridDim.x) {
            const int m0 = (tile / ct) * 8, c0 = (tile % ct) * 64;
            const int run = min(64, it.cin - c0) * taps;         // contiguous floats of a source row inside the tile (<= 0: padding only)
            __syncthreads();
            for (int idx = tid; idx < 8 * 64 * taps; idx += 256) {
                const int m_l = idx / (64 * taps), e = idx - m_l * (64 * taps);
                const int m = m0 + m_l;
                lds[m_l * pitch + e] = (m < it.cout && e < run) ? it.w[((long)m * it.cin + c0) * taps + e] : 0.f;
            }
            __syncthreads();
            for (int idx = tid; idx < 8 * taps * 64; idx += 256) {
                const int c_l = idx & 63, r = idx >> 6;          // r = m_l * taps + tap
                const int m_l = r / taps, tap = r - m_l * taps;
                const int m = m0 + m_l, c = c0 + c_l;
                if (m < it.m_pad && c < it.k_pad) out[((long)m * taps + tap) * it.k_pad + c] = (T)lds[m_l * pitch + c_l * taps + tap];
            }
        }
    } else {
        // modes 1 / 2: tile 64 output channels (image columns) x CI input channels (image rows) x source taps
        const int CI = taps == 1 ? 64 : 8, E = CI * taps;
        const int khp = it.mode == 2 ? (it.pa + it.pad) / 2 + 1 : it.kh, kwp = it.mode == 2 ? (it.pb + it.pad) / 2 + 1 : it.kw;
        const int tp = khp * kwp;                                 // taps of the image
        const int cot = (it.k_pad + 63) / 64, cit = (it.m_pad + CI - 1) / CI;
        for (int tile = blockIdx.x; tile < cot * cit; tile += gridDim.x) {
            const int ci0 = (tile / cot) * CI, co0 = (tile % cot) * 64;
            const int run = min(CI, it.cin - ci0) * taps;
            __syncthreads();
            for (int idx = tid; idx < 64 * E; idx += 256) {
                const int co_l = idx / E, e = idx - co_l * E;
                const int co = co0 + co_l;
                lds[co_l * PK_PITCH + e] = (co < it.cout && e < run) ? it.w[((long)co * it.cin + ci0) * taps + e] : 0.f;
            }
            __syncthreads();
            for (int idx = tid; idx < CI * tp * 64; idx += 256) {
                const int co_l = idx & 63, r = idx >> 6;         // r = ci_l * tp + tap'
                const int ci_l = r / tp, t2 = r - ci_l * tp;
                int fr, fs;
                if (it.mode == 2) { fr = it.pa + it.pad - 2 * (t2 / kwp); fs = it.pb + it.pad - 2 * (t2 % kwp); }
                else { fr = it.kh - 1 - t2 / it.kw; fs = it.kw - 1 - t2 % it.kw; }
                const int ci = ci0 + ci_l, co = co0 + co_l;
                float v = 0.f;
                if (fr >= 0 && fr < it.kh && fs >= 0 && fs < it.kw) v = lds[co_l * PK_PITCH + ci_l * taps + fr * it.kw + fs];
                if (ci < it.m_pad && co < it.k_pad) out[((long)ci * tp + t2) * it.k_pad + co] = (T)v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void pack_batch_kernel(const yh_pack_item* items, const int pack_tiled_enabled) {
    __shared__ float pk_lds[8 * (64 * 9 + 1) > 64 * PK_PITCH ? 8 * (64 * 9 + 1) : 64 * PK_PITCH];
    const yh_pack_item it = items[blockIdx.y];
    const int taps = it.kh * it.kw;
    if ((it.mode == 0 || it.mode == 1 || it.mode == 2) && it.dtype == YH_F16 && (taps == 1 || (it.kh == 3 && it.kw == 3)) &&
        pack_tiled_enabled) {
        pack_item_tiled<f16>(it, pk_lds);
    } else
    if (it.mode == 4) {   // depthwise: packed[tap][c] = w[c][tap], zero for padded channels
        const long total = (long)taps * it.k_pad;
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const int c = (int)(i % it.k_pad), tap = (int)(i / it.k_pad);
            const float v = c < it.cout ? it.w[(long)c * taps + tap] : 0.f;
            if (it.dtype == YH_F16) reinterpret_cast<f16*>(it.packed)[i] = (f16)v;
            else reinterpret_cast<float*>(it.packed)[i] = v;
        }
    } else if (it.mode == 3) {
        float* out = reinterpret_cast<float*>(it.packed);
        const long total = (long)taps * it.cin * it.cout_pad;
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const int co = (int)(i % it.cout_pad);
            const long r = i / it.cout_pad;
            const int ci = (int)(r % it.cin), tap = (int)(r / it.cin);
            out[i] = co < it.cout ? it.w[((long)co * it.cin + ci) * taps + tap] : 0.f;
        }
    } else if (it.dtype == YH_F16) {
        pack_item_elems<f16>(it);
    } else {
        pack_item_elems<float>(it);
    }
    if (it.bias_out) {
        const int nb = it.mode == 3 ? it.cout_pad : (it.mode == 4 ? it.k_pad : it.m_pad);
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x)
            it.bias_out[i] = (it.bias && i < it.cout) ? it.bias[i] : 0.f;
    }
}

template <typename T, bool SCATTER>
__global__ void resample2_kernel(const yh_resample_desc d) {
    typedef typename WG<T>::vec V;
    constexpr int VEC = WG<T>::VEC;
    const int cg = d.c / VEC;
    const long total = (long)d.n * d.h * d.w_in * cg;
    const T* x = reinterpret_cast<const T*>(d.x);
    T* y = reinterpret_cast<T*>(d.y);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        long pix = i / cg;
        const int wi = (int)(pix % d.w_in);
        const long r = pix / d.w_in;
        const int hi = (int)(r % d.h);
        const long n = r / d.h;
        const long w2 = d.big_w;
        const long big = ((n * d.big_h + 2L * hi) * w2 + 2L * wi);
        if constexpr (SCATTER) {  // dilate: small x -> even positions of big y
            *reinterpret_cast<V*>(y + big * d.ldy + g * VEC) = *reinterpret_cast<const V*>(x + pix * d.ldx + g * VEC);
        } else {                   // upsample backward: big x summed into small y
            const T* s = x + big * d.ldx + g * VEC;
            const V v0 = *reinterpret_cast<const V*>(s), v1 = *reinterpret_cast<const V*>(s + d.ldx);
            const V v2 = *reinterpret_cast<const V*>(s + w2 * d.ldx), v3 = *reinterpret_cast<const V*>(s + (w2 + 1) * d.ldx);
            V o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = (T)(((float)v0[e] + (float)v1[e]) + ((float)v2[e] + (float)v3[e]));
            *reinterpret_cast<V*>(y + pix * d.ldy + g * VEC) = o;
        }
    }
}

template <typename T>
__global__ void cast_f32_kernel(const yh_cast_desc d) {
    typedef typename WG<T>::vec V;
    constexpr int VEC = WG<T>::VEC;
    const int cg = d.c / VEC;
    const long total = d.pixels * cg;
    T* y = reinterpret_cast<T*>(d.y);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const long pix = i / cg;
        V o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = (T)d.x[pix * d.ldx + g * VEC + e];
        *reinterpret_cast<V*>(y + pix * d.ldy + g * VEC) = o;
    }
}

static inline unsigned grid_for(long total) {
    long g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace yh

using namespace yh;

extern "C" int yh_conv_pack_weights_dgrad(int dtype, const float* w, int cout, int cin, int kh, int kw, int cout_k,
                                          int m_pad, void* packed, void* stream) {
    if (!w || !packed || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 || cout_k < cout || m_pad < cin) return YH_EINVAL;
    const long total = (long)m_pad * kh * kw * cout_k;
    if (dtype == YH_F16)
        hipLaunchKernelGGL(pack_dgrad_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, cout, cin, kh,
                           kw, cout_k, m_pad, (f16*)packed);
    else if (dtype == YH_F32)
        hipLaunchKernelGGL(pack_dgrad_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, cout, cin,
                           kh, kw, cout_k, m_pad, (float*)packed);
    else return YH_EINVAL;
    return check_launch();
}

// YH_WGRAD_HALO: 0 = im2col kernels only, 1 = conv_wgrad_halo_kernel (round 3), 2 (default) and 3 = conv_wgrad_roll_kernel (conv_wgrad_roll.hip,
// round 5) on every layer it qualifies for (with the fragment-refresh order it is the faster form on every shape, profiles/r05_wgrad_roll_order3_ab.txt,
// so the per-layer choice mode 2 once made is gone and the two values are synonyms - ADVICE r5).  Read at every call (the tests switch it per case).
static int wgrad_halo_mode() {
    const char* e = getenv("YH_WGRAD_HALO");
    return e ? atoi(e) : YH_WGRAD_HALO_DEFAULT;
}

static bool wgrad_xcd_mapping() {
    const char* e = getenv("YH_WGRAD_XCD");   // A/B knob: 0 = plain (tile, split) grid
    return !e || atoi(e) != 0;
}

static bool wgrad_big_default(const yh_wgrad_desc* d, int ncols) {
    (void)d; (void)ncols;
    return false;   // see the measurement above
}

// geometry of the 3x3 halo form (conv_wgrad_halo_kernel); false when the layer does not qualify
static bool wgrad_halo_geometry(const yh_wgrad_desc* d, WgradArgs* pa, int* psplits, size_t* plds) {
    WgradArgs& a = *pa;
    // On by default (YH_WGRAD_HALO=0 disables it): 807 / 916 / 856 vs 759 / 861 / 740 TFLOP/s for the im2col kernel on the
    // 76 / 38 / 19 grids at batch 64 (profiles/r03_wgrad_halo.txt); its compute side alone (LDS-DMA ablated) runs at 1550
    const int mode = wgrad_halo_mode();
    if (!mode || d->dtype != YH_F16 || d->splits == -1) return false;
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->ho != d->h || d->wo != d->w_in) return false;
    if (d->cout % 256 || d->cin % 32 || d->w_in < 16) return false;
    const int Wp = d->w_in + 1;
    const int rows_hp = ((256 + 2 * Wp + 2 + 15) / 16) * 16;
    const int lbw = (rows_hp / 16 + 11) / 12;
    if (lbw < 2 || lbw > 4 || rows_hp > 512) return false;     // the four halo waves fetch 32 pieces per chunk
    const long Q = (long)d->n * (d->h + 1) * Wp;
    if (Q + 4096 >= 0x7fffffffL) return false;
    a.halo = 1;
    { const char* e = getenv("YH_WGRAD_HALO_NOSTAGGER"); a.pp = e && atoi(e) ? 1 : 0; }
    a.h_rows = rows_hp;
    a.h_lbw = lbw;
    a.h_chunks = (int)((Q + 255) / 256);
    a.bm = 256;
    a.bn = 288;
    a.tiles_m = d->cout / 256;
    a.tiles_n = d->cin / 32;
    const int tiles = a.tiles_m * a.tiles_n;
    int splits = d->splits > 0 ? d->splits : 256 / tiles;      // one workgroup per CU (134 KB of LDS, 12 waves)
    { const char* e = getenv("YH_WGRAD_HALO_WGS"); if (e && d->splits <= 0) splits = atoi(e) / tiles; }
    if (splits < 1) splits = 1;
    if (splits > a.h_chunks) splits = a.h_chunks;
    a.h_cps = (a.h_chunks + splits - 1) / splits;
    *psplits = (a.h_chunks + a.h_cps - 1) / a.h_cps;
    // 5-stage dz ring + two halo images + sink + the 16-bit pixel table of one split (entries are relative to its first image row)
    const size_t tbl = (((size_t)a.h_cps * 256 + 2 * Wp + 2 + 16) * 2 + 15) & ~(size_t)15;
    const size_t lds = (size_t)5 * 16384 + (size_t)2 * rows_hp * 64 + 1024 + tbl;
    if (lds > 160 * 1024 || (long)a.h_cps * 256 + 4L * Wp + 8 >= 65535) { a.halo = 0; return false; }
    *plds = lds;
    return true;
}

static void wgrad_geometry(const yh_wgrad_desc* d, WgradArgs* pa, int* psplits, bool allow_halo = true) {
    WgradArgs& a = *pa;
    const int bk = d->dtype == YH_F16 ? 32 : 16;
    a.d = *d;
    a.pp = 0;
    a.halo = 0;
    a.cin_w = d->cin_w > 0 ? d->cin_w : d->cin;
    a.two_stage = 0;
    a.dma = d->dtype == YH_F16 && d->splits != -1;
    a.ncols = d->kh * d->kw * d->cin;
    a.pixels = (long)d->n * d->ho * d->wo;
    a.xcd_splits = 0;
    { size_t lds; if (allow_halo && wgrad_halo_geometry(d, pa, psplits, &lds)) return; }
    a.pixels = (long)d->n * d->ho * d->wo;
    int bm = d->cout <= 64 ? 64 : WG_TILE;        // 64-row tiles for the early, wide-resolution layers
    a.ncols = d->kh * d->kw * d->cin;
    a.dma = d->dtype == YH_F16 && d->splits != -1;           // the LDS-DMA kernel
    // 256-column tiles (8 waves) when they stay >= 85 % full; the register-staged kernels are 128 wide
    a.bn = WG_TILE;   // 256-column (8-wave) tiles measured slower (VGPR-limited to one workgroup per CU); YH_WGRAD_BN=256 selects them
    { const char* e = getenv("YH_WGRAD_BN"); if (e && a.dma && bm == 128) a.bn = atoi(e); }
    // 256-row tiles on 4 waves (128 x 64 per wave: 25 % fewer LDS bytes and 33 % fewer L2 bytes per MFMA; 72 KB of LDS, two
    // workgroups per CU) for the K-heavy layers that fill them: +20 % on the 3x3 layers at batch 64 (76x76 128 -> 256:
    // 633 -> 765 TFLOP/s, 38x38 256 -> 512: 775 -> 917), neutral on 1x1 layers and when a workgroup gets < 64 K steps.
    if (a.dma && a.bn == 128 && bm == 128 && d->cout % 256 == 0) {
        const char* e = getenv("YH_WGRAD_BM");              // A/B knob: 128 / 256 force the tile
        const int force = e ? atoi(e) : 0;
        const long ksteps = ((long)d->n * d->ho * d->wo + bk - 1) / bk;
        const int tiles256 = (d->cout / 256) * ((a.ncols + a.bn - 1) / a.bn);
        const int splits256 = tiles256 >= 512 ? 1 : 512 / tiles256;
        if (force == 256 || (force != 128 && d->kh * d->kw > 1 && ksteps / splits256 >= 64)) bm = 256;
    }
    // 256 x 256 tiles on 8 waves (128 x 64 per wave, 96 KB of LDS, one workgroup per CU): 32 KB per K step for 256 MFMAs instead
    // of 2 x 24 KB - the bytes-per-FLOP argument of the forward ping-pong kernel.  Measured (profiles/r02_wgrad_tile_ab.txt):
    // not faster - 76x76 128 -> 256: 678 vs 752 TFLOP/s, 38x38 / 19x19: equal within noise; one workgroup per CU loses the overlap
    // of two.  Kept behind YH_WGRAD_BIG=1 as the A/B baseline; never chosen automatically.
    if (a.dma && bm == 256 && a.bn == 128) {
        const char* e = getenv("YH_WGRAD_BIG");
        const int big = e ? atoi(e) : -1;
        if (big >= 1 || (big == -1 && wgrad_big_default(d, a.ncols))) a.bn = 256;
        a.pp = big == 2 ? 1 : 0;    // YH_WGRAD_BIG = 2: the ping-pong schedule of the 256 x 256 tile
    }
    a.bm = bm;
    a.tiles_m = (d->cout + bm - 1) / bm;
    a.tiles_n = (a.ncols + a.bn - 1) / a.bn;
    a.ksteps = (int)((a.pixels + bk - 1) / bk);
    a.two_stage = 0;
    a.cin_w = d->cin_w > 0 ? d->cin_w : d->cin;
    { const int qw = bk / d->wo; a.rw = bk - qw * d->wo; a.qh = qw / d->ho; a.rh = qw - a.qh * d->ho; }
    const int tiles = a.tiles_m * a.tiles_n;
    int splits = d->splits;
    if (splits <= 0) {
        // one resident wave of workgroups (3 per CU at 48 KB of LDS, 4 for the 64-row tiles; 256 CUs), rounded DOWN: a
        // partly filled second wave costs 10-20 % (measured: 1024 -> 768 workgroups, 76x76 128 -> 256: 0.324 -> 0.287 ms),
        // and every extra split costs a partial tile of traffic
        // (round 6, profiles/r06_wgrad_split_sweep.txt: the 1x1 layers - 8 .. 32 tiles, a partial tile of traffic per split - are best at two
        // workgroups per CU: 38^2 512 -> 256 0.064 -> 0.057 ms, 19^2 1024 -> 512 0.059 -> 0.053, 76^2 256 -> 128 0.083 -> 0.080; every other target
        // between -384 and -1536 is slower on them, and -512 is slower on the 64-row 3x3 layers)
        // (a single tiny tile - Mobilenetv3's 208^2 16 -> 16 - keeps -1024: 0.088 against 0.113 ms)
        const bool one_small_tile = d->cout <= 32 && a.ncols <= 32;
        int target = d->kh * d->kw == 1 && !one_small_tile ? -512 : (bm == 64 ? -1024 : (bm == 256 ? (a.bn == 256 ? -256 : -512) : -768));
        { const char* e = getenv("YH_WGRAD_TARGET"); if (e) target = atoi(e); }
        splits = target > 0 ? (target + tiles - 1) / tiles : (-target) / tiles;   // negative: round down (one wave of workgroups)
        if (splits < 1) splits = 1;
        const int max_splits = (a.ksteps + 7) / 8;         // at least 8 K steps per workgroup
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    a.ksteps_per_split = ((a.ksteps + splits - 1) / splits + 1) & ~1;   // even: the kernel runs two steps per trip
    *psplits = (a.ksteps + a.ksteps_per_split - 1) / a.ksteps_per_split;
}

extern "C" int yh_conv_pack_weights_dgrad_phase(int dtype, const float* w, int cout, int cin, int kh, int kw, int pad, int pa,
                                                int pb, int cout_k, int m_pad, void* packed, int* kh_p, int* kw_p, void* stream) {
    if (!w || !packed || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 || pad < 0 || cout_k < cout || m_pad < cin) return YH_EINVAL;
    if (pa < 0 || pa > 1 || pb < 0 || pb > 1) return YH_EINVAL;
    // taps r = pa + pad - 2t with 0 <= r < kh: t from max(0, ceil((pa+pad-kh+1)/2)) .. floor((pa+pad)/2); the window
    // is anchored at t = 0, so leading taps that fall outside the kernel are packed as zeros
    const int khp = (pa + pad) / 2 + 1, kwp = (pb + pad) / 2 + 1;
    if (kh_p) *kh_p = khp;
    if (kw_p) *kw_p = kwp;
    const long total = (long)m_pad * khp * kwp * cout_k;
    if (dtype == YH_F16)
        hipLaunchKernelGGL(pack_dgrad_phase_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, cout, cin, kh,
                           kw, pad, pa, pb, khp, kwp, cout_k, m_pad, (f16*)packed);
    else if (dtype == YH_F32)
        hipLaunchKernelGGL(pack_dgrad_phase_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, cout, cin,
                           kh, kw, pad, pa, pb, khp, kwp, cout_k, m_pad, (float*)packed);
    else return YH_EINVAL;
    return check_launch();
}

extern "C" int yh_pack_batch(const yh_pack_item* items, int n_items, void* stream) {
    if (!items || n_items <= 0 || n_items > 65535) return YH_EINVAL;
    const char* e = getenv("YH_PACK_GATHER");        // A/B knob: 1 = the gather form for every image
    hipLaunchKernelGGL(pack_batch_kernel, dim3(512, n_items), dim3(256), 0, (hipStream_t)stream, items, (e && atoi(e)) ? 0 : 1);
    return check_launch();
}

extern "C" int yh_conv2d_wgrad(const yh_wgrad_desc* d, void* stream) {
    if (!d || !d->x || !d->dz || !d->dw || d->n <= 0 || d->cin <= 0 || d->cout <= 0) return YH_EINVAL;
    if (d->dtype != YH_F16 && d->dtype != YH_F32) return YH_EINVAL;
    const int vec = d->dtype == YH_F16 ? 8 : 4;
    if (d->ldx % vec || d->lddz % vec || !aligned16(d->x) || !aligned16(d->dz)) return YH_EALIGN;
    if (d->cin % vec || d->cin_w > d->cin) return YH_EALIGN;
    if ((long)d->n * d->h * d->w_in * d->ldx >= (1L << 31) || (long)d->n * d->ho * d->wo * d->lddz >= (1L << 31)) return YH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (wgrad_halo_mode() >= 2) {
        const int rc = launch_wgrad_roll(d, st);
        if (rc != YH_EUNSUPPORTED) return rc;      // does not qualify / no workspace: the forms below
    }
    WgradArgs a;
    int splits;
    wgrad_geometry(d, &a, &splits);
    if (a.halo) {
        const int htiles = a.tiles_m * a.tiles_n;
        if (d->ws && d->ws_floats >= (int64_t)splits * htiles * a.bm * a.bn) {
            size_t lds;
            int s2;
            wgrad_halo_geometry(d, &a, &s2, &lds);
            a.two_stage = 1;
#define YH_WH_CASE(L)                                                                                                          \
            case L: {                                                                                                          \
                auto kern = conv_wgrad_halo_kernel<L>;                                                                         \
                {                                                                                                              \
                    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);                                   \
                    if (e != hipSuccess) return (int)e;                                                                        \
                }                                                                                                              \
                reduce_guard_workspace(st);                                                                                    \
                hipLaunchKernelGGL(kern, dim3((unsigned)(htiles * splits)), dim3(768), lds, st, a);                            \
                break;                                                                                                         \
            }
            switch (a.h_lbw) {
                YH_WH_CASE(2) YH_WH_CASE(3) YH_WH_CASE(4)
                default: return YH_EUNSUPPORTED;
            }
#undef YH_WH_CASE
            int groups = (splits + 15) / 16;
            if (groups > 32) groups = 32;
            const int per_group = (splits + groups - 1) / groups;
            groups = (splits + per_group - 1) / per_group;
            const hipStream_t rs = reduce_begin(st);
            if (deterministic() && groups > 1) {      // groups in place, then one owner per element adds them in order (no atomics)
                hipLaunchKernelGGL(wgrad_halo_reduce_kernel, dim3(htiles * 24, groups), dim3(768), 0, rs, a, splits, per_group, 1, 1);
                hipLaunchKernelGGL(wgrad_halo_reduce_kernel, dim3(htiles * 24, 1), dim3(768), 0, rs, a, groups, groups, per_group, 0);
            } else {
                hipLaunchKernelGGL(wgrad_halo_reduce_kernel, dim3(htiles * 24, groups), dim3(768), 0, rs, a, splits, per_group, 1, 0);
            }
            reduce_end(st, rs);
            return check_launch();
        }
        wgrad_geometry(d, &a, &splits, false);   // no workspace for the partial tiles: the im2col form
    }
    const int tiles = a.tiles_m * a.tiles_n;
    const bool narrow = d->cout <= 64;
    a.two_stage = d->ws && d->ws_floats >= (int64_t)splits * tiles * a.bm * a.bn;
    dim3 grid(tiles, splits);
    a.xcd_splits = 0;
    if (a.dma && splits >= 2 && wgrad_xcd_mapping()) {
        a.xcd_splits = splits;
        grid = dim3((unsigned)(tiles * splits), 1);
    }
    if (a.two_stage) reduce_guard_workspace(st);      // a previous weight gradient's reduce may still be reading the workspace
    if (a.dma) {
        if (narrow) hipLaunchKernelGGL((conv_wgrad_dma_kernel<2, 2>), grid, dim3(256), 0, st, a);
        else if (a.bm == 256 && a.bn == 256 && a.pp) hipLaunchKernelGGL((conv_wgrad_dma_kernel<8, 4, true>), grid, dim3(512), 0, st, a);
        else if (a.bm == 256 && a.bn == 256) hipLaunchKernelGGL((conv_wgrad_dma_kernel<8, 4>), grid, dim3(512), 0, st, a);
        else if (a.bm == 256) hipLaunchKernelGGL((conv_wgrad_dma_kernel<8, 2>), grid, dim3(256), 0, st, a);
        else if (a.bn == 256) hipLaunchKernelGGL((conv_wgrad_dma_kernel<4, 4>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((conv_wgrad_dma_kernel<4, 2>), grid, dim3(256), 0, st, a);
    } else if (d->dtype == YH_F16) {   // splits == -1: the register-staged kernel (kept as the A/B baseline)
        if (narrow) hipLaunchKernelGGL((conv_wgrad_kernel<f16, 2>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv_wgrad_kernel<f16, 4>), grid, dim3(256), 0, st, a);
    } else {
        if (narrow) hipLaunchKernelGGL((conv_wgrad_kernel<float, 2>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv_wgrad_kernel<float, 4>), grid, dim3(256), 0, st, a);
    }
    if (a.two_stage) {
        int groups = (splits + 15) / 16;                    // >= 16 splits per group, <= 32 groups: the atomics that join the
        if (groups > 32) groups = 32;                       // groups are the expensive part (4x more groups measured 30 % slower)
        const int per_group = (splits + groups - 1) / groups;
        groups = (splits + per_group - 1) / per_group;
        const hipStream_t ms = st;
        st = reduce_begin(ms);
        auto reduce = [&](int ngroups, int nsplits, int per, int sstep, int native) {
            if (narrow) hipLaunchKernelGGL((wgrad_reduce_kernel<2, 2>), dim3(tiles * 8, ngroups), dim3(256), 0, st, a, nsplits, per, sstep, native);
            else if (a.bm == 256 && a.bn == 256) hipLaunchKernelGGL((wgrad_reduce_kernel<8, 4>), dim3(tiles * 32, ngroups), dim3(512), 0, st, a, nsplits, per, sstep, native);
            else if (a.bm == 256) hipLaunchKernelGGL((wgrad_reduce_kernel<8, 2>), dim3(tiles * 32, ngroups), dim3(256), 0, st, a, nsplits, per, sstep, native);
            else if (a.bn == 256) hipLaunchKernelGGL((wgrad_reduce_kernel<4, 4>), dim3(tiles * 16, ngroups), dim3(512), 0, st, a, nsplits, per, sstep, native);
            else hipLaunchKernelGGL((wgrad_reduce_kernel<4, 2>), dim3(tiles * 16, ngroups), dim3(256), 0, st, a, nsplits, per, sstep, native);
        };
        if (deterministic() && groups > 1) {      // the groups' sums in place, then one owner per element adds them in order (no atomics)
            reduce(groups, splits, per_group, 1, 1);
            reduce(1, groups, groups, per_group, 0);
        } else {
            reduce(groups, splits, per_group, 1, 0);
        }
        reduce_end(ms, st);
    }
    return check_launch();
}

extern "C" int yh_conv2d_wgrad_kernel(const yh_wgrad_desc* d) {
    if (!d || d->n <= 0 || d->cin <= 0 || d->cout <= 0 || (d->dtype != YH_F16 && d->dtype != YH_F32)) return YH_EINVAL;
    if (wgrad_halo_mode() >= 2 && wgrad_roll_workspace(d) > 0) return 91;
    WgradArgs a;
    int splits;
    wgrad_geometry(d, &a, &splits);
    if (a.halo) return 90;
    if (!a.dma) return 1;
    if (d->cout <= 64) return 22;
    return (a.bm == 256 ? 80 : 40) + (a.bn == 256 ? 4 : 2);
}

extern "C" int64_t yh_conv2d_wgrad_workspace(const yh_wgrad_desc* d) {
    if (!d || d->n <= 0 || d->cin <= 0 || d->cout <= 0 || (d->dtype != YH_F16 && d->dtype != YH_F32)) return 0;
    WgradArgs a;
    int splits;
    wgrad_geometry(d, &a, &splits);
    const int64_t need = (int64_t)splits * a.tiles_m * a.tiles_n * a.bm * a.bn;
    const int64_t roll = wgrad_halo_mode() >= 2 ? wgrad_roll_workspace(d) : 0;
    return roll > need ? roll : need;
}

extern "C" int yh_stem_wgrad(const yh_wgrad_desc* d, void* stream) {
    if (!d || !d->x || !d->dz || !d->dw) return YH_EINVAL;
    if (d->cin != 3 || d->kh != 3 || d->kw != 3 || d->cout <= 0 || d->cout > 256) return YH_EINVAL;
    if (d->dtype != YH_F16 && d->dtype != YH_F32) return YH_EINVAL;
    const long pixels = (long)d->n * d->ho * d->wo;
    const int chunk = 4096;
    const dim3 grid((unsigned)((pixels + chunk - 1) / chunk));
    if (d->dtype == YH_F16) hipLaunchKernelGGL(stem_wgrad_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, *d, pixels, chunk);
    else hipLaunchKernelGGL(stem_wgrad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, *d, pixels, chunk);
    return check_launch();
}

static int check_resample(const yh_resample_desc* d) {
    if (!d || !d->x || !d->y || d->n <= 0 || d->h <= 0 || d->w_in <= 0 || d->c <= 0) return YH_EINVAL;
    if (d->dtype != YH_F16 && d->dtype != YH_F32) return YH_EINVAL;
    if (d->big_h < 2 * d->h - 1 || d->big_h > 2 * d->h || d->big_w < 2 * d->w_in - 1 || d->big_w > 2 * d->w_in) return YH_EINVAL;
    const int vec = d->dtype == YH_F16 ? 8 : 4;
    if (d->c % vec || d->ldx % vec || d->ldy % vec || !aligned16(d->x) || !aligned16(d->y)) return YH_EALIGN;
    return YH_OK;
}

extern "C" int yh_dilate2(const yh_resample_desc* d, void* stream) {
    int rc = check_resample(d);
    if (rc) return rc;
    const int vec = d->dtype == YH_F16 ? 8 : 4;
    const long total = (long)d->n * d->h * d->w_in * (d->c / vec);
    if (d->dtype == YH_F16) hipLaunchKernelGGL((resample2_kernel<f16, true>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, *d);
    else hipLaunchKernelGGL((resample2_kernel<float, true>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, *d);
    return check_launch();
}

extern "C" int yh_upsample2_bwd(const yh_resample_desc* d, void* stream) {
    int rc = check_resample(d);
    if (rc) return rc;
    if (d->big_h != 2 * d->h || d->big_w != 2 * d->w_in) return YH_EINVAL;
    const int vec = d->dtype == YH_F16 ? 8 : 4;
    const long total = (long)d->n * d->h * d->w_in * (d->c / vec);
    if (d->dtype == YH_F16) hipLaunchKernelGGL((resample2_kernel<f16, false>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, *d);
    else hipLaunchKernelGGL((resample2_kernel<float, false>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, *d);
    return check_launch();
}

extern "C" int yh_nchw_to_nhwc(const float* x, void* y, int n, int c, int h, int w, int c_pad, int ldy, int dtype, void* stream) {
    if (!x || !y || n <= 0 || c <= 0 || h <= 0 || w <= 0 || c_pad < c || ldy < c_pad) return YH_EINVAL;
    const long total = (long)n * h * w;
    if (dtype == YH_F16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, (f16*)y, n, c, h, w, c_pad, ldy);
    else if (dtype == YH_F32)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, (float*)y, n, c, h, w, c_pad, ldy);
    else return YH_EINVAL;
    return check_launch();
}

extern "C" int yh_cast_f32(const yh_cast_desc* d, void* stream) {
    if (!d || !d->x || !d->y || d->pixels <= 0 || d->c <= 0) return YH_EINVAL;
    if (d->dtype != YH_F16 && d->dtype != YH_F32) return YH_EINVAL;
    const int vec = d->dtype == YH_F16 ? 8 : 4;
    if (d->c % vec || d->ldy % vec || !aligned16(d->y)) return YH_EALIGN;
    const long total = d->pixels * (d->c / vec);
    if (d->dtype == YH_F16) hipLaunchKernelGGL(cast_f32_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, *d);
    else hipLaunchKernelGGL(cast_f32_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, *d);
    return check_launch();
}
