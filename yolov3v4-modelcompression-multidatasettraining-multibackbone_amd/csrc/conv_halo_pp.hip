// 3x3 / stride 1 / pad 1 convolution, "halo ping-pong" kernel (tile code 43): 128 channels x 512 virtual pixels per workgroup.
//
// Why this shape (round-2 measurements): every implicit-GEMM kernel of this library sits on the same wall, ~20 B/clk
// per CU of global -> LDS delivery.  An im2col tile of BM x BN outputs has to be handed (BM + BN) rows per K step, whatever the
// schedule does; the 3x3 halo form (conv_igemm.hip, tile 41) fetches the activation rows once per channel chunk instead of once
// per tap, which leaves the WEIGHT rows as the delivery: at 128 x 256 that is still 11 KB per 512 MFMA cycles = 21 B/clk.  Here
//   * the pixel tile is 512 wide, so a K step (one tap of one 64-byte channel chunk) needs 8 KB of weights + 1/9 of a
//     ~43 KB halo image per 1024 MFMA cycles = 12.5 B/clk: below the wall with margin;
//   * each of the 8 waves owns 128 channels x 64 pixels (12 fragment reads per 32 MFMAs, the ratio of the ping-pong kernels),
//     every wave reads the whole weight tile;
//   * the two waves of a SIMD alternate LOAD segments (fragment reads, LDS-DMA issue, counted vmcnt) and 16-MFMA segments
//     between raw barriers, group 1 one interval behind group 0 - the schedule of conv_igemm_pp_kernel, two phases per K step.
//
// Virtual pixel space (one pad column and one pad row are SHARED between neighbours, so only (H+1)(W+1)/(HW) - 1 of the MFMA work
// is padding: 2.7 % at 76 x 76, 10.8 % at 19 x 19): image n, row y, column x  <->  v = n (H+1)(W+1) + (y+1)(W+1) + (x+1).
// Column 0 of a row is the zero to the left of x = 0 AND the zero to the right of x = W-1 of the row above; row 0 of an image
// is its top padding AND the bottom padding of the image before.  Tap (r, s) of output v is input v + (r-1)(W+1) + (s-1): a
// constant row offset into the staged image, which holds the consecutive inputs [v0 - (W+1) - 1, v0 + 512 + (W+1) + 1).
//
// LDS: weights in a 4-stage ring of 128 rows x 64 B (LDS-DMA of step s+3 issued in step s), halo image double-buffered per
// channel chunk (the next chunk's image is fetched one 16-row piece per wave per tap during taps 0 .. LB-1), rows of 4 cells of
// 16 B, cell c of row r stored at c ^ (2 * ((r >> 2) & 1)): ds_read_b128 fragment reads are conflict free for ANY row offset
// (the rows r and r + 4, r + 12 that share an LDS cycle always differ in that bit).
#include "conv_igemm.h"

namespace yh {

template <typename T> struct HppMma;
template <> struct HppMma<f16> {
    typedef f16x8 frag_t;
    static __device__ __forceinline__ f32x4 mma(const frag_t& a, const frag_t& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct HppMma<int8_t> {
    typedef i32x4 frag_t;
    static __device__ __forceinline__ i32x4 mma(const frag_t& a, const frag_t& b, const i32x4& c) {
        return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    }
};

// Epilogue of the 128 x 64 wave tile over virtual pixels: pix[j] = real NHWC pixel index of column j of this lane, or -1 for
// a padding position.  Same arithmetic and the same load-before-store ordering as conv_epilogue_plain (conv_igemm.h).
template <typename T, int ACT, typename AccT>
__device__ __forceinline__ void hpp_epilogue(const ConvArgs& a, AccT (&acc)[8][4], const f32x4 (&bvs)[8], const long (&pix)[4],
                                             const int m0, const long tile_row, const int lane, float* const stage) {
    constexpr int TM = 8, TN = 4;
    const int mq = (lane >> 4) << 2, pc = lane & 15;
    T* const yg = reinterpret_cast<T*>(a.y);
    const T* const rg = reinterpret_cast<const T*>(a.res);
    const int mbase = m0 + mq;
    auto value = [&](auto ic, auto jc, int e) {
        constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
        if constexpr (sizeof(T) == 1) {
            const float y = activate_q<ACT>((float)acc[i][j][e] * a.acc_scale + bvs[i][e], a.slope, a.inv_out_scale);
            return round_clamp_i8(y * a.inv_out_scale);
        } else {
            return activate_t<ACT, T>((float)acc[i][j][e] + bvs[i][e], a.slope);
        }
    };
    if constexpr (sizeof(T) != 1) {
        if (a.stats_part != nullptr) {   // BatchNorm partial sums of the values as stored; one row per (pixel tile, wave)
            float* const row = a.stats_part + tile_row * 2 * a.Cout;
            static_for<TM>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
                static_for<TN>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if (pix[j] >= 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float q = (float)(T)value(ic, jc, e);
                            s1[e] += q;
                            s2[e] = fmaf(q, q, s2[e]);
                        }
                    }
                });
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t1 = row16_sum(s1[e]), t2 = row16_sum(s2[e]);
                    // round 6: the wave's 2 x 128 sums are collected in its LDS row and leave as ONE 1 KB store - written straight from
                    // the four lanes that hold them they were 64 four-lane store instructions per wave, four times the 16 of the output
                    // itself, in front of it in the same queue (a layer with statistics ran 9 - 15 % slower than without)
                    // (no staging row - the largest geometries fill the LDS, the persistent A/B form has none - : the direct stores)
                    const int m = mbase + i * 16 + e;
                    if (pc == 15 && stage != nullptr) {
                        stage[i * 16 + mq + e] = t1;
                        stage[128 + i * 16 + mq + e] = t2;
                    } else if (pc == 15 && m < a.Cout) {
                        row[m] = t1;
                        row[a.Cout + m] = t2;
                    }
                }
            });
            __builtin_amdgcn_wave_barrier();
            if (stage != nullptr) {
                const int half = lane >> 5, c4 = (lane & 31) * 4;
                const f32x4 sv = *reinterpret_cast<const f32x4*>(stage + half * 128 + c4);
                if (m0 + c4 < a.Cout) *reinterpret_cast<f32x4*>(row + half * a.Cout + m0 + c4) = sv;
            }
        }
    }
    // ---- stores, 8 consecutive channels per lane.  A 16 x 16 fragment leaves a lane (k = lane >> 4, pixel lane & 15) with channels
    // 4k .. 4k+3; v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of its second, so for a
    // fragment pair (i, i+1), swapping the pair's values dword by dword leaves row k with channels 8 (k >> 1) .. +7 of fragment
    // i + (k & 1): one 16-byte (f16) / 8-byte (int8) store per pair instead of two half-width ones (the store tail of a 128 x 64 wave
    // tile is issue-bound: 32 -> 16 store instructions).  The swap moves the un-rounded fp32 values, the residual is added to them
    // in the new layout, and the result is rounded once: bit-identical to the narrow form.
    const int k4 = lane >> 4;
    const int chan = ((k4 & 1) << 4) + ((k4 >> 1) << 3);      // channel offset of this lane's 8-group inside a fragment pair
    const bool have_res = rg != nullptr;
    static_for<TN>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const long p = pix[j];
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        typedef typename std::conditional<sizeof(T) == 2, u32x4, u32x2>::type res8_t;   // 8 residual channels
        res8_t rv[TM / 2];
        if (have_res) {
#pragma unroll
            for (int i2 = 0; i2 < TM / 2; ++i2) {
                rv[i2] = res8_t{};
                const int m = m0 + i2 * 32 + chan;
                if (p >= 0 && m < a.Cout) rv[i2] = *reinterpret_cast<const res8_t*>(rg + p * a.ldr + m);
            }
#pragma unroll
            for (int i2 = 0; i2 < TM / 2; ++i2) {   // one wait for the column's residuals, before its first store
                res8_t t = rv[i2];
                asm volatile("" : "+v"(t));
                rv[i2] = t;
            }
        }
        T* const prow = yg + (p >= 0 ? p : 0) * a.ldy;
        static_for<TM / 2>([&](auto ic) {
            constexpr int i2 = decltype(ic)::value;
            typedef std::integral_constant<int, 2 * i2> I0;
            typedef std::integral_constant<int, 2 * i2 + 1> I1;
            float v[8], qa[4], qb[4];
            if constexpr (sizeof(T) == 1) {
                quantize4<ACT>(acc[I0::value][decltype(jc)::value], bvs[I0::value], a, qa);
                quantize4<ACT>(acc[I1::value][decltype(jc)::value], bvs[I1::value], a, qb);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    qa[e] = value(I0{}, jc, e);
                    qb[e] = value(I1{}, jc, e);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned lo = __builtin_bit_cast(unsigned, qa[e]);
                const unsigned hi = __builtin_bit_cast(unsigned, qb[e]);
                const auto r = __builtin_amdgcn_permlane16_swap(lo, hi, false, false);
                v[e] = __builtin_bit_cast(float, (unsigned)r[0]);
                v[4 + e] = __builtin_bit_cast(float, (unsigned)r[1]);
            }
            const int m = m0 + i2 * 32 + chan;
            if (p < 0 || m >= a.Cout) return;
            if constexpr (sizeof(T) == 2) {
                if (have_res) {
                    const f16x8 r8 = __builtin_bit_cast(f16x8, rv[i2]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
                }
                const f16x8 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3], (f16)v[4], (f16)v[5], (f16)v[6], (f16)v[7]};
                *reinterpret_cast<f16x8*>(prow + m) = o;
            } else {
                if (have_res) {   // fused quantised shortcut: v holds the grid values the conv would have stored
                    float r8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) r8[e] = (float)(int8_t)((rv[i2][e >> 2] >> (8 * (e & 3))) & 0xff);
                    qadd_n<8>(v, r8, a);
                }
                u32x2 o;
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    o[h] = ((unsigned)(int)v[4 * h] & 0xffu) | (((unsigned)(int)v[4 * h + 1] & 0xffu) << 8) |
                           (((unsigned)(int)v[4 * h + 2] & 0xffu) << 16) | (((unsigned)(int)v[4 * h + 3] & 0xffu) << 24);
                *reinterpret_cast<u32x2*>(prow + m) = o;
            }
        });
    });
}

// vmcnt allowance at the end of tap t's second load segment while the next chunk's halo image is being fetched: the two weight
// tiles issued after the one the next step reads, plus the halo pieces issued after it (taps t-2 .. t that have a piece).  The
// last tap allows the weight tiles only: every halo piece must have landed before the next chunk's first fragment read.
template <int T9, int LB> struct HppAllow {
    static constexpr int in(int t) { return (t >= 0 && t < LB) ? 1 : 0; }
    static constexpr int value = T9 == 8 ? 2 : 2 + in(T9 - 2) + in(T9 - 1) + in(T9);
};

// The same for the REFRESH form of the K loop (KFORM 2), whose waits cover the weight tile of step s + 2: group 0 waits at the end of
// tap t (behind it in the queue: the halo piece of tap t - 1, the weight tile of step s + 3, the halo piece of tap t), group 1 before
// its mid-step barrier (behind it: the halo piece of tap t - 1 and the tile of step s + 3).  From tap 7 on every halo piece must have
// landed (the next chunk's image is first read in tap 8's second phase).
template <int T9, int LB> struct HppAllow2 {
    static constexpr int in(int t) { return (t >= 0 && t < LB) ? 1 : 0; }
    static constexpr int end = T9 >= 7 ? 1 : 1 + in(T9 - 1) + in(T9);
    static constexpr int mid = T9 >= 7 ? 1 : 1 + in(T9 - 1);
};

// ROLE = how the LDS-DMA issue is shared out.  0: every wave issues one 16-row weight piece per K step and one halo piece per tap
// (taps 0 .. LB-1).  1: the waves of group 0 issue the weight tiles (two pieces each per step), the waves of group 1 the halo images
// (two pieces each per tap in taps 0 .. 5, offsets from a table in LDS).  A wave's LDS-DMA returns are counted IN ORDER, so with
// ROLE 0 a 16-row halo piece (16 scattered half lines, slow) in front of the weight pieces holds every counted wait for the next
// weight tile up - the weight-gradient halo kernel gained 20 % from separating the two streams (profiles/r03_wgrad_halo.txt).
// Division of a 31-bit index by a run-time constant as multiply-high + shift (host: hpp_magic): the ~35-instruction software divide
// sat 2 (LB + 4) times per lane on the critical path of every workgroup - the halo offsets must exist before the first LDS-DMA
// piece can be issued, the pixel indices before the first store - ~600 instructions of prologue before the first load.
struct HppDiv { unsigned m_img, s_img, m_wp, s_wp; };
__device__ __forceinline__ int hpp_div(int n, unsigned m, unsigned s) { return (int)(__umulhi((unsigned)n, m) >> s); }

template <typename T, int LB, int ROLE, int KFORM>
__global__ __launch_bounds__(512, 2) void conv3x3_hpp_kernel(const ConvArgs a, const int rows_hp, const int hbufs_arg, const HppDiv dv) {
    const int hbufs = hbufs_arg & 0xff;                  // bit 8: the launch reserved the statistics staging rows behind the halo table
    constexpr bool ONEBAR = KFORM == 1, REFRESH = KFORM == 2;
    // KFORM 1 (ONEBAR): ONE barrier per K step instead of four (round 5).  The instruction stream stays what it is - X loads, X MFMAs, Y loads, Y
    // MFMAs - and so does the stagger of the two wave groups, but it is no longer enforced segment by segment: group 1's barrier of a
    // step falls after its Y loads, group 0's after its Y MFMAs, so between two barriers group 0 runs XL XM YL YM of step s while group
    // 1 runs YM of step s - 1 and XL XM YL of step s.  Every barrier is a round trip of 8 waves during which a SIMD's matrix pipe
    // drains; the same change on the weight-gradient kernel (conv_wgrad_roll.hip) gained 3 - 5 %.  Hazards: the weight piece of step
    // s + 3 is issued in XL(s), after barrier s - 1, into the stage last read in YL(s - 1), before that barrier, by either group; the
    // counted wait of YL(s) covers step s + 1 and precedes barrier s for both groups, the first read of step s + 1 follows it.
    constexpr int VEC = Prec<T>::VEC, BK = VEC * 4;       // one 64-byte chunk per row per K step: 32 f16 / 64 int8 channels
    constexpr int BM = 128, BN = 512, TM = 8, TN = 4, NW = 8, SA = 4;
    constexpr int A_CELLS = BM * 4;
    constexpr int LBR = ROLE ? 1 : LB;                    // per-lane halo offsets held in registers (ROLE 1: none, see the table)
    static_assert(sizeof(T) <= 2, "f16 / int8");
    typedef typename HppMma<T>::frag_t frag_t;
    typedef typename AccOf<T>::type acc_t;
    typedef const void __attribute__((address_space(1))) * gptr_t;
    typedef void __attribute__((address_space(3))) * lptr_t;

    extern __shared__ __attribute__((aligned(16))) u32x4 hsm[];   // the only LDS object of the kernel
    u32x4* const Aring = hsm;                                     // [SA][128 rows x 4 cells]
    u32x4* const Hbuf = hsm + SA * A_CELLS;                       // [hbufs][rows_hp x 4 cells]
    u32x4* const dummy = Hbuf + hbufs * rows_hp * 4;              // [64] sink of the pieces beyond the image
    int* const htab = reinterpret_cast<int*>(dummy + 64);         // ROLE 1: [rows_hp] element offset of halo row j, or -1

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int group = wave >> 2;

    // STAGGER (round 6 experiment, YH_HPP_STAGGER = cycles; 0 = off; measured +0.3 % over 13 layer shapes, profiles/r06_hpp_stagger_ab.txt: not the lever): one workgroup per CU and tiles of equal length keep all 256 CUs in
    // phase - they all compute, then all store 128 KB each (33 MB in one burst, at the HBM write rate: the 12 600 cycles an epilogue
    // takes are exactly 128 KB x 256 CUs at 6.3 TB/s), then all compute again.  Delaying the workgroups of the FIRST round by
    // different fractions of that burst time puts the CUs out of phase for the whole launch.
    if (a.hpp_stagger > 0 && blockIdx.x < 256) {
        const long long t_end = clock64() + (long long)((blockIdx.x * 37u) & 255u) * a.hpp_stagger / 256;
        while (clock64() < t_end) __builtin_amdgcn_s_sleep(16);
    }

    int m_tile, p_tile;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, rr = nb & 7, xcd = bid & 7;
        const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
        p_tile = logical / a.m_tiles;
        m_tile = logical - p_tile * a.m_tiles;
    }
    const int m0 = m_tile * BM;
    const long q0 = (long)p_tile * BN;          // first virtual output position of the tile
    const int Wp = a.W + 1;
    const int IMG = (a.H + 1) * Wp;

    // ---- loader: an LDS-DMA instruction fills 16 rows x 4 cells; lane -> row lane >> 2, destination cell lane & 3, which holds
    // source cell (lane & 3) ^ 2 ((row >> 2) & 1) - the same for weight rows and halo rows (16-row pieces start on multiples of 16)
    const int lrow = lane >> 2;
    const int lu = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    const T* const xg = reinterpret_cast<const T*>(a.x);
    // weight piece of this wave: ROLE 0 rows 16 wave .., ROLE 1 (group 0 only) rows 32 wave .. and 16 further
    const int wrow = (ROLE ? 2 * (wave & 3) : wave) * 16 + lrow;
    const T* const wsrc = reinterpret_cast<const T*>(a.w) + (long)min(m0 + wrow, a.m_pad - 1) * a.ktot + lu * VEC;
    const int wsrc2 = (min(m0 + wrow + 16, a.m_pad - 1) - min(m0 + wrow, a.m_pad - 1)) * a.ktot;   // ROLE 1: second piece (elements)
    auto halo_offset = [&](int j) {      // element offset of halo row j (channel 0), or -1: padding / beyond the batch
        const int v = (int)q0 - Wp - 1 + j;      // the launcher keeps the whole virtual space below 2^31
        int off = -1;
        if (j < rows_hp && v >= 0) {
            const int n = hpp_div(v, dv.m_img, dv.s_img);
            const int rem = v - n * IMG;
            const int yy = hpp_div(rem, dv.m_wp, dv.s_wp), xx = rem - yy * Wp;
            if (n < a.N && yy >= 1 && xx >= 1) off = ((n * a.H + yy - 1) * a.W + xx - 1) * a.ldx;
        }
        return off;
    };
    int boff[LBR];
    if constexpr (ROLE == 0) {
        static_for<LB>([&](auto c) {
            constexpr int i = decltype(c)::value;
            boff[i] = halo_offset((wave + i * NW) * 16 + lrow);
        });
    } else {
        boff[0] = 0;
        for (int j = threadIdx.x; j < rows_hp; j += 512) htab[j] = halo_offset(j);
        __syncthreads();       // no LDS-DMA in flight yet: a plain barrier
    }
    const unsigned htab_lds = (unsigned)(uintptr_t)(lptr_t)htab;
    auto issue_a = [&](int stage, int tap, int kc) {
        if constexpr (ROLE == 0) {
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (tap * a.cin_k + kc)), (lptr_t)(Aring + stage * A_CELLS + wave * 64), 16, 0, 0);
        } else {
            const T* const src = wsrc + (tap * a.cin_k + kc);
            u32x4* const dst = Aring + stage * A_CELLS + (wave & 3) * 128;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(src + wsrc2), (lptr_t)(dst + 64), 16, 0, 0);
        }
    };
    // the zero page's address as two scalars: the pointer select then holds no vector registers (a spilled value inside the K
    // loop is reloaded by a scratch load, which drains the LDS-DMA queue: same counter)
    const unsigned zlo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)g_zero_page);
    const unsigned zhi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)g_zero_page >> 32));
    auto issue_piece = [&](int buf, int kc, int g, int off) {     // halo piece g (wave-uniform), this lane's row offset `off`
        // the piece's LDS destination is re-derived at every issue (four scalar instructions): with g visible as a loop invariant the
        // compiler keeps the LB destinations and their in-image flags in scalar registers over the whole K loop, and the int8 LB = 7
        // form ran out of them (11 SGPR spills and a scratch frame: `make check-spills`)
        asm volatile("" : "+s"(g));
        const bool ok = off >= 0 && kc + lu * VEC < a.Cin;
        const unsigned long long u = (unsigned long long)(uintptr_t)(xg + (off + kc + lu * VEC));
        const unsigned lo = ok ? (unsigned)u : zlo, hi = ok ? (unsigned)(u >> 32) : zhi;
        u32x4* dst = (g * 16 < rows_hp) ? Hbuf + buf * (rows_hp * 4) + g * 64 : dummy;
        __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(((unsigned long long)hi << 32) | lo), (lptr_t)dst, 16, 0, 0);
    };
    auto issue_h = [&](int buf, int kc, auto ic) {      // ROLE 0: piece i of this wave for the chunk at channel offset kc
        constexpr int i = decltype(ic)::value;
        issue_piece(buf, kc, wave + i * NW, boff[i < LBR ? i : 0]);
    };
    auto issue_h2 = [&](int buf, int kc, int i) {       // ROLE 1 (group 1): pieces 2i, 2i+1 of this wave: g = (wave - 4) + 4 (2i + k)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int g = (wave & 3) + 4 * (2 * i + k);
            const int j = min(g * 16 + lrow, rows_hp - 1);
            int off;   // inline asm: an ordinary LDS load would be ordered against the LDS-DMA queue by the compiler
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(off) : "v"(htab_lds + 4u * (unsigned)j) : "memory");
            issue_piece(buf, kc, g, off);
        }
    };

    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

    // ---- fragment addressing.  Weights: row i 16 + r16 of the stage (rows 64 .. 127 in the second phase), cell kq ^ f(row); the
    // permutation bit of a row only depends on r16 there.  Halo: row wave 64 + j 16 + r16 + tapoff, whose bit depends on r16 + tapoff.
    const int r16 = lane & 15, kq = lane >> 4;
    const int a_off = r16 * 4 + (kq ^ (((r16 >> 2) & 1) << 1));
    frag_t fa[4], fb[4];
    auto read_a = [&](const u32x4* st, int half) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x4 v = st[(half * 64 + i * 16) * 4 + a_off];
            fa[i] = *reinterpret_cast<frag_t*>(&v);
        }
    };
    auto read_b = [&](const u32x4* hb, int tapoff) {
        const int rt = r16 + tapoff;
        const int off = (wave * 64 + rt) * 4 + (kq ^ (((rt >> 2) & 1) << 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4 v = hb[off + j * 64];
            fb[j] = *reinterpret_cast<frag_t*>(&v);
        }
    };
    auto mma = [&](auto hc) {
        constexpr int h = decltype(hc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h * 4 + i][j] = HppMma<T>::mma(fa[i], fb[j], acc[h * 4 + i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    typedef std::integral_constant<int, 0> H0;
    typedef std::integral_constant<int, 1> H1;
#define YH_HPP_BARRIER()                     \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)
    // one phase = LOAD segment (ends with: all of this wave's LDS reads returned), barrier, 16 MFMAs, barrier
#define YH_HPP_PHASE(LOADS, MMA, BAR_L, BAR_M)                \
    do {                                                      \
        LOADS;                                                \
        __builtin_amdgcn_sched_barrier(0);                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
        if (BAR_L) __builtin_amdgcn_s_barrier();              \
        __builtin_amdgcn_sched_barrier(0);                    \
        MMA;                                                  \
        __builtin_amdgcn_sched_barrier(0);                    \
        if (BAR_M) __builtin_amdgcn_s_barrier();              \
        __builtin_amdgcn_sched_barrier(0);                    \
    } while (0)

    const int nchunks = a.cin_k / BK;
    const int nk = 9 * nchunks;
    // ---- prologue: halo image of chunk 0, weight tiles of steps 0 .. 2; steps 1 and 2 stay in flight
    if constexpr (ROLE == 0) {
        static_for<LB>([&](auto ic) { issue_h(0, 0, ic); });
        issue_a(0, 0, 0);
        issue_a(1, 1, 0);
        issue_a(2, 2, 0);
        wait_vmcnt<2>();
    } else if (group == 0) {
        issue_a(0, 0, 0);
        issue_a(1, 1, 0);
        issue_a(2, 2, 0);
        wait_vmcnt<4>();
    } else {
        for (int i = 0; i < 6; ++i) issue_h2(0, 0, i);
        wait_vmcnt<0>();
    }
    if constexpr (REFRESH) {
        // ---- REFRESH form (round 5).  The weight-gradient kernel gained 8 - 16 % when its fragment reads stopped being a segment of
        // their own (conv_wgrad_roll.hip ORDER 3; tools/probe/run_lds_probe.py): a wave holds ONE set of fragments and re-reads each
        // right after its last MFMA.  Here: fa = the four weight fragments of a 64-channel half, fb = the four pixel fragments of the tap.
        //   X (channels 0 .. 63), row-major: fa[i] is dead after its four MFMAs -> refreshed with the SAME step's second half;
        //   Y (channels 64 .. 127), column-major: fb[j] is dead after column j -> refreshed with the next tap's (at tap 8: the next chunk's
        //   image) rows; fa[i] after MFMA (i, 3) -> the next step's first half.
        // 12 reads between 32 MFMAs, no LOAD segment, ONE barrier per step: group 0's at the end of the step, group 1's after X, each
        // preceded by that group's counted wait for the weight tile of step s + 2 (HppAllow2).  Hazards: the tile of step s + 3 is issued at
        // the start of step s into the stage last read in X(s - 1), which precedes barrier s - 1 for both groups; stage s + 1 is read from
        // Y(s) on, after barrier s - 1, which every wave reaches with its share of tile s + 1 landed.
        static_assert(!REFRESH || ROLE == 0, "the refresh form shares the LDS-DMA issue among all waves");
        wait_vmcnt<1>();       // (the shared prologue above left the tiles of steps 1 and 2 in flight) step 1's is read from Y(0) on
        YH_HPP_BARRIER();
        read_a(Aring, 0);
        read_b(Hbuf, 0);
        __builtin_amdgcn_sched_barrier(0);
        int s = 0, st_r = 0, st_w = 3, ptap = 3, pkc = 0;
        for (int c = 0; c < nchunks; ++c) {
            const bool more_h = LB <= 7 ? c + 1 < nchunks : false;
            const u32x4* const hb = Hbuf + (c & (hbufs - 1)) * (rows_hp * 4);
            const u32x4* const hb_next = Hbuf + ((c + 1) & (hbufs - 1)) * (rows_hp * 4);
            const int nbuf = (c + 1) & 1, nkc = (c + 1) * BK;
            static_for<9>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                const u32x4* const st = Aring + st_r * A_CELLS;
                const u32x4* const stn = Aring + ((st_r + 1) & (SA - 1)) * A_CELLS;
                const u32x4* const hbn = t < 8 ? hb : hb_next;
                const int rt_next = r16 + (t < 8 ? ((t + 1) / 3) * Wp + ((t + 1) % 3) : 0);
                const int b_next = (wave * 64 + rt_next) * 4 + (kq ^ (((rt_next >> 2) & 1) << 1));
                if (s + 3 < nk) issue_a(st_w, ptap, pkc);
                __builtin_amdgcn_sched_barrier(0);
                // ---- X
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = HppMma<T>::mma(fa[i], fb[j], acc[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                    u32x4 v = st[(64 + i * 16) * 4 + a_off];
                    fa[i] = *reinterpret_cast<frag_t*>(&v);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (group == 1) {
                    if (more_h) wait_vmcnt<HppAllow2<t, LB>::mid>();
                    else if (s + 3 < nk) wait_vmcnt<1>();
                    else wait_vmcnt<0>();
                    YH_HPP_BARRIER();
                }
                if constexpr (t < LB) { if (more_h) issue_h(nbuf, nkc, tc); }
                __builtin_amdgcn_sched_barrier(0);
                // ---- Y
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[4 + i][j] = HppMma<T>::mma(fa[i], fb[j], acc[4 + i][j]);
                        if (j == 3) {
                            __builtin_amdgcn_sched_barrier(0);
                            u32x4 v = stn[(i * 16) * 4 + a_off];
                            fa[i] = *reinterpret_cast<frag_t*>(&v);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    u32x4 v = hbn[b_next + j * 64];
                    fb[j] = *reinterpret_cast<frag_t*>(&v);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (group == 0) {
                    if (more_h) wait_vmcnt<HppAllow2<t, LB>::end>();
                    else if (s + 3 < nk) wait_vmcnt<1>();
                    else wait_vmcnt<0>();
                    YH_HPP_BARRIER();
                }
                ++s;
                st_r = (st_r + 1) & (SA - 1);
                st_w = (st_w + 1) & (SA - 1);
                if (++ptap == 9) { ptap = 0; pkc += BK; }
            });
        }
    } else {
    YH_HPP_BARRIER();
    if (!ONEBAR && group == 1) YH_HPP_BARRIER();   // stagger: group 1 runs one barrier interval behind group 0
    int s = 0;                   // K step being computed
    int st_r = 0, st_w = 3;      // ring stage of step s / of step s + 3
    int ptap = 3, pkc = 0;       // tap / channel offset of step s + 3
    for (int c = 0; c < nchunks; ++c) {
        // the next chunk's halo image is fetched during this chunk; LB 9 (W >= 262) is a single-chunk form (hpp_geometry), so its
        // in-loop prefetch compiles out - with it the 9 per-lane piece offsets stayed live through the K loop and spilled
        const bool more_h = LB <= 7 ? c + 1 < nchunks : false;
        const u32x4* const hb = Hbuf + (c & (hbufs - 1)) * (rows_hp * 4);
        const int nbuf = (c + 1) & 1, nkc = (c + 1) * BK;
        static_for<9>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const u32x4* const st = Aring + st_r * A_CELLS;
            const int tapoff = (t / 3) * Wp + (t % 3);
            constexpr int allow = HppAllow<t, LB>::value;   // (a comma inside <> would split the macro argument)
            // ---- phase X: channels 0 .. 63 of the tile x this wave's 64 pixels; weight tile of step s + 3 into the stage that
            // step s - 1 was read from (both groups are past its last read: the barrier that ended the previous interval)
            YH_HPP_PHASE({ read_a(st, 0); read_b(hb, tapoff); if ((ROLE == 0 || group == 0) && s + 3 < nk) issue_a(st_w, ptap, pkc); }, mma(H0{}), !ONEBAR, !ONEBAR);
            // ---- phase Y: channels 64 .. 127; halo pieces of the next chunk; then this wave's share of step s + 1 (and of
            // everything older) must have landed: the barrier after this segment, and for the other group the next one, publish it
            YH_HPP_PHASE({
                read_a(st, 1);
                if constexpr (ROLE == 0) {
                    if constexpr (t < LB) { if (more_h) issue_h(nbuf, nkc, tc); }
                    if (more_h) wait_vmcnt<allow>();
                    else {
                        const int rem = nk - 2 - s;
                        if (rem >= 2) wait_vmcnt<2>(); else if (rem == 1) wait_vmcnt<1>(); else wait_vmcnt<0>();
                    }
                } else if (group == 0) {      // weight waves: two pieces per step, steps s + 2 and s + 3 stay in flight
                    const int rem = nk - 2 - s;
                    if (rem >= 2) wait_vmcnt<4>(); else if (rem == 1) wait_vmcnt<2>(); else wait_vmcnt<0>();
                } else if (more_h) {          // halo waves: the next chunk's image in taps 0 .. 5, complete two taps before its first read
                    if constexpr (t < 6) issue_h2(nbuf, nkc, t);
                    if constexpr (t == 7) wait_vmcnt<0>();
                }
            }, mma(H1{}), !ONEBAR || group == 1, !ONEBAR || group == 0);
            ++s;
            st_r = (st_r + 1) & (SA - 1);
            st_w = (st_w + 1) & (SA - 1);
            if (++ptap == 9) { ptap = 0; pkc += BK; }
        });
    }
    if (!ONEBAR && group == 0) YH_HPP_BARRIER();   // matches group 1's extra barrier at the start
    }
#undef YH_HPP_PHASE
#undef YH_HPP_BARRIER

    // ---- epilogue.  Every load before the first store (bias here; residual columns inside hpp_epilogue)
    const int mq = (lane >> 4) << 2;
    f32x4 bvs[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + i * 16 + mq;
        bvs[i] = m < a.Cout ? *reinterpret_cast<const f32x4*>(a.bias + m) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(bvs[i]));
    long pix[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int q = (int)q0 + wave * 64 + j * 16 + r16;
        const int n = hpp_div(q, dv.m_img, dv.s_img);
        const int rem = q - n * IMG;
        const int yy = hpp_div(rem, dv.m_wp, dv.s_wp), xx = rem - yy * Wp;
        pix[j] = (n < a.N && yy >= 1 && xx >= 1) ? ((long)n * a.H + yy - 1) * a.W + xx - 1 : -1;
    }
    const long tile_row = (long)p_tile * NW + wave;
    // statistics staging row of this wave (1 KB) behind the halo offset table: launch_hpp adds the 8 KB when statistics are asked for
    float* const stage = (hbufs_arg >> 8) ? reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(htab) + ((rows_hp * 4 + 15) & ~15)) + wave * 256 : nullptr;
    switch (a.act) {
        case YH_ACT_LEAKY: hpp_epilogue<T, YH_ACT_LEAKY>(a, acc, bvs, pix, m0, tile_row, lane, stage); break;
        case YH_ACT_MISH: hpp_epilogue<T, YH_ACT_MISH>(a, acc, bvs, pix, m0, tile_row, lane, stage); break;
        default: hpp_epilogue<T, YH_ACT_LINEAR>(a, acc, bvs, pix, m0, tile_row, lane, stage); break;
    }
}

#ifdef YH_HPP_PERSIST_AB
// ---- PERSISTENT form (tile code 44; round 4).  The kernel above pays ~29 000 cycles per 128 x 512 tile outside its K loop - workgroup
// launch, index set-up, the first halo image and weight stages arriving into an idle CU, the epilogue's stores draining before the CU
// takes its next workgroup - as much as 22 K steps; a 76 x 76 128 -> 256 layer has 18.  Here one
// workgroup per CU walks over its tiles (the same XCD-chunked order) and the stream never stops at a tile boundary:
//   * the weight ring keeps running: in the last three K steps of a tile the stages of the NEXT tile's steps 0 .. 2 are issued;
//   * the next tile's first halo image is fetched during the current tile's last channel chunk, exactly like a next chunk's image;
//   * the two wave groups re-align for the epilogue and stagger again for the next tile (one barrier interval per tile);
//   * halo piece offsets are computed when a piece is issued (multiply-high divisions), not held in registers: the next tile's
//     offsets need no state, and the kernel has LB registers fewer than the one above.
// The epilogue's stores sit on the same in-order queue as the next tile's LDS-DMA: the first counted waits of a tile also wait for the
// previous tile's last stores (~2000 cycles once per tile).  Needs >= 2 channel chunks (two halo buffers).  Same arithmetic, same
// summation order, same statistics rows as the kernel above: bit-identical results.
template <typename T, int LB>
__global__ __launch_bounds__(512, 2) void conv3x3_hpp_persist_kernel(const ConvArgs a, const int rows_hp, const HppDiv dv, const int total_tiles) {
    constexpr int VEC = Prec<T>::VEC, BK = VEC * 4;
    constexpr int BM = 128, BN = 512, TM = 8, TN = 4, NW = 8, SA = 4;
    constexpr int A_CELLS = BM * 4;
    static_assert(sizeof(T) <= 2, "f16 / int8");
    typedef typename HppMma<T>::frag_t frag_t;
    typedef typename AccOf<T>::type acc_t;
    typedef const void __attribute__((address_space(1))) * gptr_t;
    typedef void __attribute__((address_space(3))) * lptr_t;

    extern __shared__ __attribute__((aligned(16))) u32x4 hsm[];
    u32x4* const Aring = hsm;                                     // [SA][128 rows x 4 cells]
    u32x4* const Hbuf = hsm + SA * A_CELLS;                       // [2][rows_hp x 4 cells]
    u32x4* const dummy = Hbuf + 2 * rows_hp * 4;                  // [64] sink of the pieces beyond the image

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int group = wave >> 2;
    const int Wp = a.W + 1;
    const int IMG = (a.H + 1) * Wp;
    const int lrow = lane >> 2;
    const int lu = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    const T* const xg = reinterpret_cast<const T*>(a.x);
    const int wrow = wave * 16 + lrow;

    // tile k of this workgroup = virtual block blockIdx.x + k gridDim.x of a total_tiles-block launch of the kernel above
    auto tile_of = [&](int v, int& m_tile, int& p_tile) {
        const int nb = total_tiles;
        const int q = nb >> 3, rr = nb & 7, xcd = v & 7;
        const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (v >> 3);
        p_tile = logical / a.m_tiles;
        m_tile = logical - p_tile * a.m_tiles;
    };
    // this lane's element offset into the packed weights for the tile rows at m0 (the launcher keeps the image below 2^31 elements):
    // one register for the current tile; the next tile's is recomputed at its three issue sites
    const T* const wbase = reinterpret_cast<const T*>(a.w);
    auto weight_off = [&](int m0) { return min(m0 + wrow, a.m_pad - 1) * a.ktot + lu * VEC; };
    auto halo_offset = [&](int q0, int j) {      // element offset of halo row j of the tile at q0 (channel 0), or -1
        const int v = q0 - Wp - 1 + j;
        int off = -1;
        if (j < rows_hp && v >= 0) {
            const int n = hpp_div(v, dv.m_img, dv.s_img);
            const int rem = v - n * IMG;
            const int yy = hpp_div(rem, dv.m_wp, dv.s_wp), xx = rem - yy * Wp;
            if (n < a.N && yy >= 1 && xx >= 1) off = ((n * a.H + yy - 1) * a.W + xx - 1) * a.ldx;
        }
        return off;
    };
    const unsigned zlo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)g_zero_page);
    const unsigned zhi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)g_zero_page >> 32));
    auto issue_a = [&](int woff, int stage, int tap, int kc) {
        __builtin_amdgcn_global_load_lds((gptr_t)(wbase + (woff + tap * a.cin_k + kc)), (lptr_t)(Aring + stage * A_CELLS + wave * 64), 16, 0, 0);
    };
    int boff[LB];      // this lane's row offsets of the image being fetched: the current tile's; from the last chunk on the next tile's
    auto set_offsets = [&](int q0) {
        static_for<LB>([&](auto ic) { boff[decltype(ic)::value] = halo_offset(q0, (wave + decltype(ic)::value * NW) * 16 + lrow); });
    };
    auto issue_h = [&](int buf, int kc, auto ic) {      // piece i of this wave (group g = wave + 8 i)
        constexpr int i = decltype(ic)::value;
        const int g = wave + i * NW;
        const int off = boff[i];
        const bool ok = off >= 0 && kc + lu * VEC < a.Cin;
        const unsigned long long u = (unsigned long long)(uintptr_t)(xg + (off + kc + lu * VEC));
        const unsigned lo = ok ? (unsigned)u : zlo, hi = ok ? (unsigned)(u >> 32) : zhi;
        u32x4* dst = (g * 16 < rows_hp) ? Hbuf + buf * (rows_hp * 4) + g * 64 : dummy;
        __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(((unsigned long long)hi << 32) | lo), (lptr_t)dst, 16, 0, 0);
    };

    const int r16 = lane & 15, kq = lane >> 4;
    const int a_off = r16 * 4 + (kq ^ (((r16 >> 2) & 1) << 1));
    frag_t fa[4], fb[4];
    acc_t acc[TM][TN];
    auto read_a = [&](const u32x4* st, int half) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x4 v = st[(half * 64 + i * 16) * 4 + a_off];
            fa[i] = *reinterpret_cast<frag_t*>(&v);
        }
    };
    auto read_b = [&](const u32x4* hb, int tapoff) {
        const int rt = r16 + tapoff;
        const int off = (wave * 64 + rt) * 4 + (kq ^ (((rt >> 2) & 1) << 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4 v = hb[off + j * 64];
            fb[j] = *reinterpret_cast<frag_t*>(&v);
        }
    };
    auto mma = [&](auto hc) {
        constexpr int h = decltype(hc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h * 4 + i][j] = HppMma<T>::mma(fa[i], fb[j], acc[h * 4 + i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    typedef std::integral_constant<int, 0> H0;
    typedef std::integral_constant<int, 1> H1;
#define YH_HPP_BARRIER()                     \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)
#define YH_HPP_PHASE(LOADS, MMA)                              \
    do {                                                      \
        LOADS;                                                \
        __builtin_amdgcn_sched_barrier(0);                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
        __builtin_amdgcn_s_barrier();                         \
        __builtin_amdgcn_sched_barrier(0);                    \
        MMA;                                                  \
        YH_HPP_BARRIER();                                     \
    } while (0)

    const int nchunks = a.cin_k / BK;
    const int nk = 9 * nchunks;
    int vb = blockIdx.x;                         // virtual block of the current tile
    int m_tile, p_tile;
    tile_of(vb, m_tile, p_tile);
    int wsrc = weight_off(m_tile * BM);
    // ---- prologue of the FIRST tile: halo image of chunk 0, weight tiles of steps 0 .. 2; steps 1 and 2 stay in flight
    set_offsets(p_tile * BN);
    static_for<LB>([&](auto ic) { issue_h(0, 0, ic); });
    issue_a(wsrc, 0, 0, 0);
    issue_a(wsrc, 1, 1, 0);
    issue_a(wsrc, 2, 2, 0);
    wait_vmcnt<2>();
    YH_HPP_BARRIER();

    int st_r = 0, st_w = 3;      // ring stage of step s / of step s + 3 (the ring runs on across tiles)
    int cc = 0;                  // running chunk count: chunk cc lives in halo buffer cc & 1
    for (;;) {
        const int m0 = m_tile * BM;
        const int q0 = p_tile * BN;
        const int vnext = vb + (int)gridDim.x;
        const bool has_next = vnext < total_tiles;
        int m_next = 0, p_next = 0;
        if (has_next) tile_of(vnext, m_next, p_next);
        const int q0_next = p_next * BN;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
        if (group == 1) YH_HPP_BARRIER();   // stagger: group 1 runs one barrier interval behind group 0

        int s = 0;                   // K step being computed
        int ptap = 3, pkc = 0;       // tap / channel offset of step s + 3
        for (int c = 0; c < nchunks; ++c, ++cc) {
            const bool last_chunk = c + 1 == nchunks;
            const bool more_h = !last_chunk || has_next;       // an image is fetched during this chunk: the next chunk's, or the next tile's first
            const u32x4* const hb = Hbuf + (cc & 1) * (rows_hp * 4);
            const int nbuf = (cc + 1) & 1;
            const int hkc = last_chunk ? 0 : (c + 1) * BK;
            if (last_chunk && has_next) set_offsets(q0_next);      // the current tile's offsets were last used in the previous chunk
            static_for<9>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                const u32x4* const st = Aring + st_r * A_CELLS;
                const int tapoff = (t / 3) * Wp + (t % 3);
                constexpr int allow = HppAllow<t, LB>::value;
                // ---- phase X; weight tile of step s + 3: this tile's, or - in the last three steps - the next tile's steps 0 .. 2
                YH_HPP_PHASE({
                    read_a(st, 0);
                    read_b(hb, tapoff);
                    if (s + 3 < nk) issue_a(wsrc, st_w, ptap, pkc);
                    else if (has_next) issue_a(weight_off(m_next * BM), st_w, s + 3 - nk, 0);
                }, mma(H0{}));
                // ---- phase Y; halo pieces; then this wave's share of step s + 1 (and of everything older) must have landed
                YH_HPP_PHASE({
                    read_a(st, 1);
                    if constexpr (t < LB) { if (more_h) issue_h(nbuf, hkc, tc); }
                    if (more_h) wait_vmcnt<allow>();
                    else {
                        const int rem = nk - 2 - s;
                        if (rem >= 2) wait_vmcnt<2>(); else if (rem == 1) wait_vmcnt<1>(); else wait_vmcnt<0>();
                    }
                }, mma(H1{}));
                ++s;
                st_r = (st_r + 1) & (SA - 1);
                st_w = (st_w + 1) & (SA - 1);
                if (++ptap == 9) { ptap = 0; pkc += BK; }
            });
        }
        if (group == 0) YH_HPP_BARRIER();   // re-align: matches group 1's extra barrier at the start of the tile

        // ---- epilogue of this tile (every load before the first store); the next tile's first stages are already in LDS / in flight.
        // Its arguments (output / residual / statistics pointers, pitches, scales: ~30 scalars the K loop never touches) are re-read
        // from the kernel-argument segment HERE, through a pointer the compiler cannot see through: kept live across the tile loop
        // they overflow the scalar register file and the spills land in vector registers the K loop needs
        unsigned long long kp = (unsigned long long)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        typedef const ConvArgs __attribute__((address_space(4))) * kargs_t;     // constant address space: scalar loads
        const kargs_t ka = reinterpret_cast<kargs_t>((uintptr_t)kp);
        ConvArgs ae;
        ae.bias = ka->bias; ae.res = ka->res; ae.y = ka->y; ae.Cout = ka->Cout; ae.ldr = ka->ldr; ae.ldy = ka->ldy;
        ae.acc_scale = ka->acc_scale; ae.inv_out_scale = ka->inv_out_scale; ae.act = ka->act; ae.slope = ka->slope;
        ae.stats_part = ka->stats_part; ae.q_rx = ka->q_rx; ae.q_ra = ka->q_ra; ae.q_scale_x = ka->q_scale_x;
        ae.q_scale_a = ka->q_scale_a; ae.q_inv_scale_sum = ka->q_inv_scale_sum;
        const int mq = (lane >> 4) << 2;
        f32x4 bvs[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + i * 16 + mq;
            bvs[i] = m < ae.Cout ? *reinterpret_cast<const f32x4*>(ae.bias + m) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(bvs[i]));
        long pix[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int q = q0 + wave * 64 + j * 16 + r16;
            const int n = hpp_div(q, dv.m_img, dv.s_img);
            const int rem = q - n * IMG;
            const int yy = hpp_div(rem, dv.m_wp, dv.s_wp), xx = rem - yy * Wp;
            pix[j] = (n < a.N && yy >= 1 && xx >= 1) ? ((long)n * a.H + yy - 1) * a.W + xx - 1 : -1;
        }
        const long tile_row = (long)p_tile * NW + wave;
        switch (ae.act) {
            case YH_ACT_LEAKY: hpp_epilogue<T, YH_ACT_LEAKY>(ae, acc, bvs, pix, m0, tile_row, lane, nullptr); break;
            case YH_ACT_MISH: hpp_epilogue<T, YH_ACT_MISH>(ae, acc, bvs, pix, m0, tile_row, lane, nullptr); break;
            default: hpp_epilogue<T, YH_ACT_LINEAR>(ae, acc, bvs, pix, m0, tile_row, lane, nullptr); break;
        }
        if (!has_next) break;
        vb = vnext;
        m_tile = m_next;
        p_tile = p_next;
        wsrc = weight_off(m_next * BM);
    }
#undef YH_HPP_PHASE
#undef YH_HPP_BARRIER
}

#endif  // YH_HPP_PERSIST_AB

// n / d for 0 <= n < 2^31, d >= 2: q = mulhi(n, m) >> s with L = ceil(log2 d), m = ceil(2^(31 + L) / d) < 2^32, s = L - 1
// (n m / 2^(31 + L) = n / d + n e / (d 2^(31 + L)) with 0 <= e < d <= 2^L: the error term stays below 1 / d for n < 2^31)
static void hpp_magic(unsigned d, unsigned* m, unsigned* s) {
    unsigned L = 0;
    while ((1ull << L) < d) ++L;
    const unsigned long long num = 1ull << (31 + L);
    *m = (unsigned)((num + d - 1) / d);
    *s = L - 1;
}

// geometry shared by the launcher, the tile picker and the statistics-row query
bool hpp_geometry(int W, int cin_k, int bk, int* rows_hp, int* lb, int* hbufs, size_t* lds) {
    const int Wp = W + 1;
    const int rows_h = 512 + 2 * Wp + 2;
    *rows_hp = ((rows_h + 15) / 16) * 16;
    const int groups = *rows_hp / 16;
    int l = (groups + 7) / 8;
    const int nchunks = cin_k / bk;
    *hbufs = nchunks > 1 ? 2 : 1;
    if (l == 8) l = 9;
    if (l < 5 || l > 9 || (nchunks > 1 && l > 7)) return false;   // in-loop halo prefetch issues one piece per tap in taps 0 .. 6
    *lb = l;
    *lds = ((size_t)4 * 128 * 4 + (size_t)*hbufs * *rows_hp * 4 + 64) * 16 + (size_t)*rows_hp * 4;   // + the halo offset table
    return *lds <= 160 * 1024;
}

// KFORM 2 (fragments refreshed between the MFMAs, no LOAD segment - the form that gained the weight-gradient kernel 8 - 16 %) measured
// 0 - 8 % SLOWER here, bit-identical outputs (profiles/r05_hpp_refresh_ab.txt: sum over 13 layer shapes 2.76 against 2.64 ms): with a
// 4-stage weight ring the counted wait may leave ONE tile in flight instead of two, and the 16-byte fragment reads of this kernel were
// never the issue-bound 8-byte reads of the weight gradient.  Compiled only with -DYH_HPP_REFRESH_AB (YH_HPP_KFORM=2 selects it; its
// int8 instantiations spill 2 - 4 SGPRs).
// ONEBAR (one barrier per K step) measured 3 - 10 % SLOWER than the four-barrier form on every 3x3 layer of both models, bit-identical
// outputs (profiles/r05_hpp_one_barrier_ab.txt: sum over 13 layer shapes 2.96 against 2.78 ms) - unlike the weight-gradient kernel, whose
// three wave groups gained 3 - 5 % from it: here the enforced alternation of the two groups IS the overlap.  Its instantiations are only
// compiled into an A/B build (make CXXFLAGS+=-DYH_HPP_ONEBAR_AB; then YH_HPP_BARRIERS=1 selects them); the int8 LB = 7 form runs out of
// scalar registers with it (8 SGPR spills, tools/check_spills.py) and always keeps four barriers.
template <typename T, int LBV> static auto hpp_kern(int kform) -> void (*)(const ConvArgs, const int, const int, const HppDiv) {
#ifdef YH_HPP_ONEBAR_AB
    if constexpr (!(sizeof(T) == 1 && LBV == 7)) {
        if (kform == 1) return conv3x3_hpp_kernel<T, LBV, 0, 1>;
    }
#endif
#ifdef YH_HPP_REFRESH_AB
    if constexpr (LBV <= 7) {
        if (kform == 2) return conv3x3_hpp_kernel<T, LBV, 0, 2>;
    }
#endif
    (void)kform;
    return conv3x3_hpp_kernel<T, LBV, 0, 0>;
}

#ifndef YH_HPP_ONE_BARRIER_DEFAULT
#define YH_HPP_ONE_BARRIER_DEFAULT 0
#endif
#ifndef YH_HPP_KFORM_DEFAULT
#define YH_HPP_KFORM_DEFAULT 0
#endif
template <typename T> static int launch_hpp(const ConvArgs& a0, hipStream_t stream) {
    constexpr int BK = Prec<T>::VEC * 4;
    ConvArgs a = a0;
    if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.ups != 1) return YH_EUNSUPPORTED;
    if (a.act != YH_ACT_LINEAR && a.act != YH_ACT_LEAKY && a.act != YH_ACT_MISH) return YH_EUNSUPPORTED;
    if (a.cin_k % BK) return YH_EALIGN;
    int rows_hp, lb, hbufs;
    size_t lds;
    if (!hpp_geometry(a.W, a.cin_k, BK, &rows_hp, &lb, &hbufs, &lds)) return YH_EUNSUPPORTED;
    // 32-bit element offsets inside the kernel
    if ((long)a.N * a.H * a.W * a.ldx + a.cin_k >= 0x7fffffffL || (long)a.m_pad * a.ktot >= 0x7fffffffL) return YH_EUNSUPPORTED;
    if ((long)(a.N + 1) * (a.H + 1) * (a.W + 1) + 4096 >= 0x7fffffffL) return YH_EUNSUPPORTED;
    // 8-channel (16-byte f16 / 8-byte int8) stores and residual loads
    const unsigned amask = sizeof(T) == 2 ? 15u : 7u;
    if (a.Cout % 8 || a.ldy % 8 || (((uintptr_t)a.y) & amask) || (a.res && (a.ldr % 8 || (((uintptr_t)a.res) & amask)))) return YH_EALIGN;
    a.m_tiles = (a.Cout + 127) / 128;
    const long Q = (long)a.N * (a.H + 1) * (a.W + 1);
    a.p_tiles = (int)((Q + 511) / 512);
    const long blocks = (long)a.m_tiles * a.p_tiles;
    if (blocks <= 0 || blocks > 0x7fffffffL) return YH_EINVAL;
    // statistics: one 1 KB staging row per wave behind the halo table, if the LDS has room for the 8 KB (YH_HPP_STATS_STAGE=0: the direct stores)
    int stage_ok = 0;
    if (a.stats_part != nullptr && sizeof(T) == 2) {
        const char* e = getenv("YH_HPP_STATS_STAGE");
        const size_t base = (lds + 15) & ~(size_t)15;
        if (!(e && atoi(e) == 0) && base + 8192 <= 160 * 1024) {
            lds = base + 8192;
            stage_ok = 1;
        }
    }
    { const char* e = getenv("YH_HPP_STAGGER"); a.hpp_stagger = (e && blocks > 512) ? atoi(e) : 0; }   // first-round phase stagger in cycles (A/B)
    HppDiv dv;
    hpp_magic((unsigned)((a.H + 1) * (a.W + 1)), &dv.m_img, &dv.s_img);       // both divisors >= 4 (H, W >= 1)
    hpp_magic((unsigned)(a.W + 1), &dv.m_wp, &dv.s_wp);
    const char* bar_env = getenv("YH_HPP_BARRIERS");      // A/B knob: 4 = a barrier after every segment (rounds 3 - 4), 1 = one per K step
    const int one_bar = bar_env ? (atoi(bar_env) == 1 ? 1 : 0) : YH_HPP_ONE_BARRIER_DEFAULT;
    const char* form_env = getenv("YH_HPP_KFORM");        // A/B knob: 0 = LOAD / MFMA segments (rounds 3 - 4), 2 = fragments refreshed between the MFMAs
    const int kform = one_bar ? 1 : (form_env ? atoi(form_env) : YH_HPP_KFORM_DEFAULT);
    // ROLE 1 (separate weight / halo waves; needs its 48 halo pieces to cover the image) is an A/B form, measured 5 - 9 % SLOWER
    // than the shared form here (profiles/r03_hpp_role_ab.txt) - the weight tiles are L2-resident and short, unlike the dz stream
    // of the weight-gradient kernel where the same separation gained 20 %.  Its instantiations spill 3 registers at the 256 cap,
    // so they are only compiled into an A/B build (make CXXFLAGS+=-DYH_HPP_ROLE_AB; then YH_HPP_ROLE=1 selects them).
#ifdef YH_HPP_ROLE_AB
    static const int role_env = [] { const char* e = getenv("YH_HPP_ROLE"); return e ? atoi(e) : 0; }();
    const int role = (role_env && rows_hp <= 768) ? 1 : 0;
#define YH_HPP_KERN(LBV) (role ? conv3x3_hpp_kernel<T, LBV, 1, 0> : hpp_kern<T, LBV>(kform))
#else
#define YH_HPP_KERN(LBV) hpp_kern<T, LBV>(kform)
#endif
#define YH_HPP_CASE(LBV)                                                                                                       \
    case LBV: {                                                                                                                \
        auto kern = YH_HPP_KERN(LBV);                                                                                          \
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);   /* per (kernel, device) */                      \
        if (e != hipSuccess) return (int)e;                                                                                    \
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, stream, a, rows_hp, hbufs | (stage_ok << 8), dv);     \
        break;                                                                                                                 \
    }
    switch (lb) {
        YH_HPP_CASE(5) YH_HPP_CASE(6) YH_HPP_CASE(7) YH_HPP_CASE(9)
        default: return YH_EUNSUPPORTED;
    }
#undef YH_HPP_CASE
#undef YH_HPP_KERN
    return check_launch();
}

#ifdef YH_HPP_PERSIST_AB
template <typename T> static int launch_hpp_persist(const ConvArgs& a0, hipStream_t stream) {
    constexpr int BK = Prec<T>::VEC * 4;
    ConvArgs a = a0;
    if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.ups != 1) return YH_EUNSUPPORTED;
    if (a.act != YH_ACT_LINEAR && a.act != YH_ACT_LEAKY && a.act != YH_ACT_MISH) return YH_EUNSUPPORTED;
    if (a.cin_k % BK) return YH_EALIGN;
    int rows_hp, lb, hbufs;
    size_t lds;
    if (!hpp_geometry(a.W, a.cin_k, BK, &rows_hp, &lb, &hbufs, &lds) || hbufs != 2) return YH_EUNSUPPORTED;      // >= 2 channel chunks
    if ((long)a.N * a.H * a.W * a.ldx + a.cin_k >= 0x7fffffffL || (long)a.m_pad * a.ktot >= 0x7fffffffL) return YH_EUNSUPPORTED;
    if ((long)(a.N + 1) * (a.H + 1) * (a.W + 1) + 4096 >= 0x7fffffffL) return YH_EUNSUPPORTED;
    const unsigned amask = sizeof(T) == 2 ? 15u : 7u;
    if (a.Cout % 8 || a.ldy % 8 || (((uintptr_t)a.y) & amask) || (a.res && (a.ldr % 8 || (((uintptr_t)a.res) & amask)))) return YH_EALIGN;
    a.m_tiles = (a.Cout + 127) / 128;
    const long Q = (long)a.N * (a.H + 1) * (a.W + 1);
    a.p_tiles = (int)((Q + 511) / 512);
    const long blocks = (long)a.m_tiles * a.p_tiles;
    if (blocks <= 0 || blocks > 0x7fffffffL) return YH_EINVAL;
    HppDiv dv;
    hpp_magic((unsigned)((a.H + 1) * (a.W + 1)), &dv.m_img, &dv.s_img);
    hpp_magic((unsigned)(a.W + 1), &dv.m_wp, &dv.s_wp);
    static const long wgs = [] { const char* e = getenv("YH_HPP_PERSIST_GRID"); return e ? atol(e) : 256L; }();      // 256: one workgroup per CU; 512: two
    const unsigned grid = (unsigned)(blocks < wgs ? blocks : wgs);
#define YH_HPPP_CASE(LBV)                                                                                                      \
    case LBV: {                                                                                                                \
        auto kern = conv3x3_hpp_persist_kernel<T, LBV>;                                                                        \
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);                                           \
        if (e != hipSuccess) return (int)e;                                                                                    \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, a, rows_hp, dv, (int)blocks);                             \
        break;                                                                                                                 \
    }
    switch (lb) {
        YH_HPPP_CASE(5) YH_HPPP_CASE(6) YH_HPPP_CASE(7)
        default: return YH_EUNSUPPORTED;
    }
#undef YH_HPPP_CASE
    return check_launch();
}

int launch_hpp_persist_tile(const ConvArgs& a, int dtype, hipStream_t stream) {
    if (dtype == YH_F16) return launch_hpp_persist<f16>(a, stream);
    if (dtype == YH_I8) return launch_hpp_persist<int8_t>(a, stream);
    return YH_EINVAL;
}
#endif  // YH_HPP_PERSIST_AB

int launch_hpp_tile(const ConvArgs& a, int dtype, hipStream_t stream) {
    // The persistent form (one workgroup per CU walking over its tiles) measured 10 - 25 % SLOWER than one tile per workgroup on every
    // layer of the two models (profiles/r04_hpp_persist_ab.txt) and spills; it is only compiled into an A/B build
    // (make CXXFLAGS+=-DYH_HPP_PERSIST_AB), where YH_HPP_PERSIST=1 routes the launches it supports (>= 2 channel chunks, > 256 tiles).
#ifdef YH_HPP_PERSIST_AB
    static const bool persist = [] { const char* e = getenv("YH_HPP_PERSIST"); return e && atoi(e) != 0; }();
    if (persist && a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.ups == 1) {
        const int bk = dtype == YH_I8 ? 64 : 32;
        const long tiles = (long)((a.Cout + 127) / 128) * (((long)a.N * (a.H + 1) * (a.W + 1) + 511) / 512);
        if (a.cin_k % bk == 0 && a.cin_k / bk >= 2 && tiles > 256) {
            const int rc = launch_hpp_persist_tile(a, dtype, stream);
            if (rc != YH_EUNSUPPORTED) return rc;
        }
    }
#endif
    if (dtype == YH_F16) return launch_hpp<f16>(a, stream);
    if (dtype == YH_I8) return launch_hpp<int8_t>(a, stream);
    return YH_EINVAL;
}

}  // namespace yh
