// 3x3 / stride 1 / pad 1 weight gradient, "rolling halo" form (fp16; cout % 128 == 0, cin % 64 == 0, 16 <= W <= 190; round 5).
//
//   dw[co][ci][r][s] = sum over output pixels p of dz[p][co] * x[p shifted by tap (r, s)][ci]        (autograd of models.py:92-98)
//
// conv_wgrad_halo_kernel (conv_wgrad.hip, round 3) owns a 256 (co) x [9 taps x 32 ci] tile per workgroup and moves 16 KB of dz plus
// 1/8 of a 256-pixel halo image of x per 32-pixel K step: 18 - 19 KB for 4.7 MFLOP, and its compute side alone runs at 1550 TFLOP/s where
// the full kernel reaches 830 - 970 - the global -> LDS stream, not the matrix pipe, sets its step time.
// This kernel keeps the accumulator budget (73 728 outputs = 295 KB of registers, one workgroup per CU, 12 waves) and the three
// rotating wave groups of that kernel, and changes what the stream has to deliver:
//   * tile 128 (co) x [9 taps x 64 ci]: the squarest split of 73 728 outputs.  Per 32-pixel step 8 KB of dz + 4 KB of x = 12 KB;
//   * x is not staged per chunk but ROLLS: the virtual pixel space of conv_halo_pp.hip (one shared pad row / column per image:
//     v = n (H+1)(W+1) + (y+1)(W+1) + (x+1), tap (r, s) = row offset (r-1)(W+1) + (s-1)) is walked in order, row v of x lives in
//     slot v mod 512 of a 64 KB ring of 128-byte rows, and every step fetches exactly the 32 rows that enter the window
//     [v - (W+2), v + 31 + (W+2)] - no halo overlap is fetched twice inside a split, no second buffer, no per-split pixel table;
//   * every DMA lane decodes its own row: (image, y, x) advance by 32 positions per step with two compare-selects (32 = q (W+1) + r),
//     a pad position (or one beyond the batch) reads the zero page.  One LDS-DMA instruction per wave per step: waves 0 .. 7 the 4-row
//     pieces of dz (256-byte rows), waves 8 .. 11 the 8-row pieces of x - a uniform stream, every wave's in-order counter sees only
//     its own pieces;
//   * the LDS that the second halo image and the table took goes into the dz ring: S = min(10, 16 - JL) stages (JL = x steps the window
//     leads by), S - 1 steps = up to 108 KB in flight per CU where the round-3 kernel had 64 KB for 1.5 x the bytes per step.
// Fragment reads are ds_read_b64_tr_b16 on both operands (rows = pixels, transposed on the read side).  dz rows: the unit permutation
// of conv_wgrad_dma_kernel<4, .> (16 units, XOR ((r & 3) | ((r >> 3) & 1) << 2) << 1).  x rows (128 B = half of the 64 banks, any tap
// offset): 32-byte column k of row r is stored at k ^ (((r >> 1) & 1) | ((r >> 3) & 1) << 1) - the 8 row pieces that share an LDS cycle
// (rows e + {0,1,2,3,8,9,10,11}) split into two parities of four rows {e, e+2, e+8, e+10} whose XOR values are always distinct (bit 3
// flips between r and r + 8, bit 1 between r and r + 2, a carry out of bits 1-2 flips both of one pair): conflict-free for every e.
// Partial tiles per pixel split go to the workspace in MFMA-native order, summed by wgrad_roll_reduce_kernel in split order.
#include "common.h"
#include <stdlib.h>

namespace yh {

__device__ __attribute__((aligned(16))) const uint32_t g_zero16r[4] = {0u, 0u, 0u, 0u};  // source of every pad-position 16-byte load

struct RollArgs {
    yh_wgrad_desc d;
    int tiles_m, tiles_n;     // cout / 128, cin / 64
    int steps_total, sps;     // 32-position steps of the virtual pixel space, steps per split
    int S, JL;                // dz ring stages; x stream lead in steps: ceil((2 (W+1) + 2) / 32)
    int q32, r32;             // 32 = q32 (W+1) + r32
    int cin_w;
    int nostagger;            // profiling knob (YH_WGRAD_HALO_NOSTAGGER): the three wave groups in phase
    int prio;                 // 1: s_setprio 1 around the MFMA segments
    unsigned long long* timing;   // TIMING builds only: [3 groups][8] cycle sums of workgroup 0 (tools/wgrad_ab.py --timing)
};

typedef int wr_v2i __attribute__((ext_vector_type(2)));
template <int OFF> __device__ __forceinline__ wr_v2i wr_read_tr16(unsigned lds_byte_addr) {
    wr_v2i r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_byte_addr), "n"(OFF));
    return r;
}

__device__ __forceinline__ int wr_swz_dz(int r) { return ((r & 3) | (((r >> 3) & 1) << 2)) << 1; }   // 16-byte units of a 256-byte dz row
__device__ __forceinline__ int wr_swz_x(int r) { return ((r >> 1) & 1) | (((r >> 3) & 1) << 1); }    // 32-byte columns of a 128-byte x row

__device__ __forceinline__ void wr_wait_keep(int keep) {      // counted wait with a wave-uniform run-time count: literal operands only
    switch (keep) {
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

#ifndef YH_WGRAD_ROLL_MFMA32_DEFAULT
#define YH_WGRAD_ROLL_MFMA32_DEFAULT 0
#endif
constexpr int WR_XRING = 65536;     // 512 rows x 128 B
constexpr int WR_ABYTES = 8192;     // one dz step: 32 rows x 256 B

// ABL (profiling only, bit flags; results are garbage unless ABL & 7 == 0): 1 = no LDS-DMA (the compute side alone), 2 = no fragment
// reads and no MFMAs (the stream alone), 4 = every piece reads the zero page (the DMA path without memory traffic), 8 = s_memtime
// stamps around every segment of the K loop, summed per wave group of one workgroup into a.timing
#define YH_WR_STAMP(k)                                                                   \
    do {                                                                                 \
        if constexpr (TIMING) {                                                          \
            __builtin_amdgcn_sched_barrier(0);                                           \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                           \
            tsum[k] += now_ - tlast;                                                     \
            tlast = now_;                                                                \
            __builtin_amdgcn_sched_barrier(0);                                           \
        }                                                                                \
    } while (0)
// ORDER: 0 = every fragment read of a step in one barrier interval (LOAD / MFMA / stream), 1 = SPLIT (R1 / R2 / MFMA), 2 = FREE (one
// barrier per step, see the K loop)
template <int ABL, int ORDER, bool M32 = false>
__global__ __launch_bounds__(768, 3) void conv_wgrad_roll_kernel(const RollArgs a) {
    static_assert(!M32 || ORDER == 3, "the 32x32x16 form exists for the refresh order only");
    constexpr bool SPLIT = ORDER == 1, FREE = ORDER == 2, REFRESH = ORDER == 3;
    constexpr bool TIMING = (ABL & 8) != 0, NODMA = (ABL & 1) != 0, NOCOMPUTE = (ABL & 2) != 0, ZEROSRC = (ABL & 4) != 0;
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    constexpr int NT = 768;
    const yh_wgrad_desc& d = a.d;
    extern __shared__ __attribute__((aligned(128))) unsigned char rsm[];   // [x ring 64 KB][S dz stages of 8 KB]: the only LDS object
    typedef const void __attribute__((address_space(1))) * gptr_t;
    typedef void __attribute__((address_space(3))) * lptr_t;

    const int tiles = a.tiles_m * a.tiles_n;
    int tile_id, split_id;
    {   // consecutive (split, tile) ids share an XCD: all tiles of a split read the same dz / x rows through one L2
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, rr = nb & 7, xcd = bid & 7;
        const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
        split_id = logical / tiles;
        tile_id = logical - split_id * tiles;
    }
    const int tm = tile_id % a.tiles_m, tn = tile_id / a.tiles_m;
    const int co0 = tm * 128, ci0 = tn * 64;
    const int s0 = split_id * a.sps;
    const int nsteps = min(a.sps, a.steps_total - s0);
    if (nsteps <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / 6, wn = wave - wm * 6;      // wave tile: channels 64 wm .. +63 x the three taps of filter row wn >> 1 x 32 ci (half wn & 1)
    const int trow = wn >> 1, hh = wn & 1;
    const int S = a.S, D = FREE ? S - 2 : S - 1, JL = a.JL;
    constexpr int KW = REFRESH ? 3 : 2;        // the counted wait of a step covers the piece of KW steps on
    const int Wp = d.w_in + 1, Hp = d.h + 1, IMG = Hp * Wp;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)rsm;

    // the zero page's address as two SCALARS: a pointer select against it costs no vector registers (168-register cap; a spilled
    // pointer would be reloaded by a scratch load, which sits on the LDS-DMA queue's counter)
    const unsigned zlo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)g_zero16r);
    const unsigned zhi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)g_zero16r >> 32));

    // ---- this lane's row of the stream.  dz waves (0 .. 7): instruction = 4 rows x 16 units, row 4 wave + lane / 16 of the step;
    // x waves (8 .. 11): 8 rows x 8 units, row 8 (wave - 8) + lane / 8 of the x step, which starts W + 2 positions before the dz step
    const bool dz_wave = wave < 8;                                      // wave-uniform
    int n, yy, xx, cofs;
    {
        int rowl, p;
        if (dz_wave) {
            rowl = 4 * wave + (lane >> 4);
            cofs = co0 + (((lane & 15) ^ (M32 ? (rowl & 3) << 2 : wr_swz_dz(rowl))) << 3);
            p = 32 * s0 + rowl;
        } else {
            rowl = 8 * (wave - 8) + (lane >> 3);
            cofs = ci0 + (((lane & 7) ^ ((M32 ? ((rowl >> 1) & 1) << 1 : wr_swz_x(rowl)) << 1)) << 3);
            p = 32 * s0 - Wp - 1 + rowl;
        }
        const int pp = p + IMG;                      // p >= -(W + 2) > -IMG
        const int q = pp / IMG;
        const int rem = pp - q * IMG;
        n = q - 1;
        yy = rem / Wp;
        xx = rem - yy * Wp;
    }
    const f16* const srcbase = reinterpret_cast<const f16*>(dz_wave ? d.dz : d.x);
    const unsigned ld = (unsigned)(dz_wave ? d.lddz : d.ldx);
    unsigned slot;                                   // LDS byte offset (from rsm) of this wave's next piece
    int stage_i = 0;                                 // dz: ring stage of the next piece; x: x step of the next piece (mod 16)
    slot = dz_wave ? WR_XRING + wave * 1024 : (wave - 8) * 1024;
    auto issue = [&]() {
        const bool ok = !ZEROSRC && (unsigned)n < (unsigned)d.n && yy >= 1 && xx >= 1;
        const unsigned pix = (unsigned)((n * d.h + yy - 1) * d.w_in + xx - 1);
        const unsigned long long u = (unsigned long long)(uintptr_t)(srcbase + (pix * ld + (unsigned)cofs));
        const unsigned lo = ok ? (unsigned)u : zlo, hi = ok ? (unsigned)(u >> 32) : zhi;
        __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(((unsigned long long)hi << 32) | lo), (lptr_t)(rsm + slot), 16, 0, 0);
        // the next step's row: 32 positions on
        xx += a.r32;
        yy += a.q32;
        const bool cx = xx >= Wp;
        xx = cx ? xx - Wp : xx;
        yy = cx ? yy + 1 : yy;
        const bool cy = yy >= Hp;
        yy = cy ? yy - Hp : yy;
        n = cy ? n + 1 : n;
        ++stage_i;
        if (dz_wave) {
            const bool w = stage_i == S;
            stage_i = w ? 0 : stage_i;
            slot = w ? WR_XRING + wave * 1024 : slot + WR_ABYTES;
        } else {
            slot = (slot + 4096) & (WR_XRING - 1);
        }
    };

    f32x4 acc[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment addressing (ds_read_b64_tr_b16: lane (q, g) supplies 4 consecutive channels of pixel row 8g + 4h + q/4 and
    // receives 4 consecutive pixels of channel q of the 16-channel block)
    const int q = lane & 15, g = lane >> 4;
    unsigned a_addr0;                  // dz fragment 0, half 0, inside a stage; fragment i: ^ (i << 5); half 1: + 4 rows = offset 1024
    unsigned b_abs[3][2];              // x: tap column s, half h; channel block 0 (block 1 = ^ 32); absolute LDS address, rolls 4 KB per step
    {
        const int row = 8 * g + (q >> 2);
        const int cha = wm * 64 + 4 * (q & 3);
        a_addr0 = row * 256 + ((((cha >> 3) ^ wr_swz_dz(row)) << 4) | ((cha & 7) * 2));
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int xr = row + 4 * h + trow * Wp + s;      // slot row at step 0 (slot 0 = position 32 s0 - (W + 2))
                b_abs[s][h] = lds0 + (unsigned)((xr & 511) * 128 + (((2 * hh) ^ wr_swz_x(xr)) << 5) + (q & 3) * 8);
            }
    }
    const unsigned roll_add = 4096u - lds0;       // b = lds0 + ((b - lds0 + 4096) & 0xffff)

    // ---- K loop.  Three groups of one wave per SIMD (waves 0-3, 4-7, 8-11) rotate through LOAD (20 transposed fragment reads), MFMA
    // (24 MFMAs) and the stream interval (this wave's LDS-DMA piece for step s + D, the counted wait, the address roll), one barrier
    // interval each; group g runs g intervals behind group 0.  Interval t = 3 s + phase + g:
    //   * a piece of step s + D is issued at t >= 3 s + 2 into the stage / slot of step s - 1, whose last reader (group 2, LOAD of
    //     step s - 1) finished at t = 3 s - 1;
    //   * a wave's wait in step s's stream interval (t <= 3 s + 4) covers its piece of step s + 2; the first read of step s + 2 is at
    //     t = 3 s + 6: one full barrier in between, for every group.
    const int grp = a.nostagger ? 0 : (wave >> 2);
#define YH_WR_BARRIER()                      \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)
    if constexpr (!NODMA) {
        const int n_pro = dz_wave ? min(D, nsteps) : min(JL + D, nsteps + JL);
        for (int k = 0; k < n_pro; ++k) issue();
        wr_wait_keep(max(0, n_pro - (dz_wave ? KW : JL + KW)));       // steps 0 .. KW - 1 (x steps 0 .. JL + KW - 1) have landed
    }
    YH_WR_BARRIER();
    if constexpr (!FREE && !REFRESH) for (int k = 0; k < grp; ++k) YH_WR_BARRIER();      // stagger (ORDER 0 / 1: whole barrier intervals)
    int st_read = 0;
    if constexpr (TIMING) tlast = __builtin_amdgcn_s_memtime();
    // SPLIT form.  Measured on the form below (profiles/r05_wgrad_roll_segments.txt): with the LDS-DMA ablated its LOAD segment - 20
    // ds_read_b64_tr_b16 from ONE wave per SIMD - takes ~500 cycles where the 24 MFMAs of the group next to it take ~410: 8-byte LDS
    // reads reach their rate only with several waves per SIMD issuing them (MI355X_MICROARCH.md, LDS), so the interval is set by the
    // reads.  Here a step's reads are spread over BOTH intervals in which a wave does not multiply - R1: the 12 x fragments, R2: the
    // LDS-DMA piece, the 8 dz fragments, the counted wait and the address roll - so two waves per SIMD read in every interval.
    // Interval t = 3 s + phase + g (phase 0 = R1, 1 = R2, 2 = MFMA):
    //   * the piece of step s + D is issued in R2(s) (t >= 3 s + 1) into the stage of step s - 1, last read in R2(s - 1) (t <= 3 s),
    //     and into the x slot of x step s - 1, last read in R1(s - 1) (t <= 3 s - 1);
    //   * the wait in R2(s) (t <= 3 s + 3) covers the piece of step s + 2, first read in R1(s + 2) at t >= 3 s + 6.
    // FREE form (ORDER 2).  Measured on the two forms below (profiles/r05_wgrad_roll_segments.txt): a segment of 8, 12 or 20 fragment
    // reads takes 450 - 600 cycles whatever its length, the 24 MFMAs ~430, and every one of the three barriers of a step adds the
    // round trip of 12 waves plus the restart of the next segment to an interval in which the matrix pipe of each SIMD then idles: ~600
    // cycles per interval for ~410 of MFMA work.  Here the rotation is kept but no longer enforced segment by segment: ONE barrier per
    // step; between two barriers every wave runs its three segments back to back, the groups in rotated order
    //   group 0: MFMA(T)  R1(T+1)  R2(T+1)      group 1: R1(T)  R2(T)  MFMA(T)      group 2: R2(T)  MFMA(T)  R1(T+1)
    // so that, with equal segment lengths, one wave per SIMD multiplies at any time and nothing but the matrix pipe itself serialises
    // two waves that overlap.  Hazards (barrier T opens interval T):
    //   * the piece of step s + D (D = S - 2) is issued in R2(s) - group 0: interval s - 1 - into the stage of step s - 2 and the x slot
    //     of x step s - 2, whose last readers (groups 1, 2: R2(s - 2) / R1(s - 2)) ran in interval s - 2;
    //   * the wait in R2(s) - groups 1, 2: interval s - covers the piece of step s + 2 and is published by barrier s + 1; the first
    //     read of step s + 2 is group 0's R1(s + 2) in interval s + 1.
    // REFRESH form (ORDER 3).  tools/probe/run_lds_probe.py: twelve waves that each do 20 fragment reads THEN 24 MFMAs per trip need
    // 1530 cycles per trip, the same waves with every fragment re-read right after its last MFMA of the trip 1382 - a segment of 8-byte
    // LDS reads is issue-bound per wave (one per ~24 cycles) and its latency is dead time for that wave, spread between the MFMAs it
    // is not.  So a wave holds ONE set of fragments and refreshes it in place: x fragment pair j (tap column j) is dead after the four
    // MFMAs of column j and is re-read for step s + 1 at once, dz fragment i after MFMA (i, last column).  No LOAD segment is left:
    // a step is C_0 .. C_5 (4 MFMAs + refresh reads each) and ST (this wave's LDS-DMA piece, the counted wait).  One barrier per step, at
    // a group-dependent point of the stream (after C_1 / C_3 / C_5) so that the three waves of a SIMD stay a third of a step apart.
    // Hazards (b_s = barrier of step s; interval I_s ends with it): step s + 1's data is read in C(s), i.e. in I_s or I_(s+1), and
    // every such read has RETURNED at the lgkmcnt(0) that opens step s + 1, before b_(s+1); the wait in ST(s') (in I_(s'+1)) covers the
    // piece of step s' + 3, published by b_(s'+1), first read in C_0(s' + 2) in I_(s'+2); the piece of step s' + D (D = S - 1) issued in
    // ST(s') overwrites the stage of step s' - 1, whose reads returned before b_(s'-1).
    if constexpr (REFRESH && M32) {
        // ---- v_mfma_f32_32x32x16_f16 form of the refresh order (VERDICT r4 item 1 / 3 i: never tried in csrc/ before round 5).  Same wave
        // tile (64 co x [3 taps x 32 ci]), same 20 fragment reads per step, 12 MFMAs of 32 cycles instead of 24 of 16: half the MFMA
        // issues and the shape the pipe sustains best (MI355X_MICROARCH.md: 2382 against 2075 TFLOP/s in the micro-benchmark).  Operand
        // layout: lane l holds row / column l & 31 and the 8 K values 8 (l >> 5) ..; a K step of 32 pixels = two MFMAs (pixels 0 - 15,
        // 16 - 31).  With ds_read_b64_tr_b16 the 16-lane group g reads channel block g & 1 of pixel group g >> 1, so the lanes that share
        // an LDS cycle (g = 0, 1) read TWO adjacent 32-byte columns of FOUR rows: dz rows (256 B, all banks) are permuted by the 64-byte
        // column pair, (r & 3) << 1 on the 32-byte column index; x rows (128 B, half the banks; rows r and r + 2 share them) by bit 1 of r
        // on bit 1 of the column index - the LDS-DMA side of this instantiation permutes its source units the same way.
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        typedef int v4i __attribute__((ext_vector_type(4)));
        const int g3 = wave >> 2;
        const int cblk = g & 1, pg = g >> 1;
        const int row0 = 8 * pg + (q >> 2);
        unsigned a32, b32[3][2];
        {
            const int cha = wm * 64 + 16 * cblk + 4 * (q & 3);
            a32 = row0 * 256 + ((((cha >> 3) ^ ((row0 & 3) << 2)) << 4) | ((cha & 7) * 2));      // co block m: ^ (m << 6); pixel half: + 4096; h: + 1024
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    const int xr = row0 + 4 * h + trow * Wp + sx;
                    b32[sx][h] = lds0 + (unsigned)((xr & 511) * 128 + (((2 * hh + cblk) ^ (((xr >> 1) & 1) << 1)) << 5) + (q & 3) * 8);
                }
        }
        const unsigned half_add = 2048u - lds0;       // the second pixel half: 16 rows further in the x ring (wraps with it)
        f32x16 c32[2][3];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int sx = 0; sx < 3; ++sx)
#pragma unroll
                for (int e = 0; e < 16; ++e) c32[m][sx][e] = 0.f;
        wr_v2i pa[2][2][2], pb[3][2][2];       // [co block / tap column][pixel half][h]
        auto read_pa = [&](int m, unsigned stage) {
            const unsigned ad = stage + (a32 ^ (m << 6));
            pa[m][0][0] = wr_read_tr16<0>(ad);
            pa[m][0][1] = wr_read_tr16<1024>(ad);
            pa[m][1][0] = wr_read_tr16<4096>(ad);
            pa[m][1][1] = wr_read_tr16<5120>(ad);
        };
        auto read_pb = [&](int sx) {
            pb[sx][0][0] = wr_read_tr16<0>(b32[sx][0]);
            pb[sx][0][1] = wr_read_tr16<0>(b32[sx][1]);
            pb[sx][1][0] = wr_read_tr16<0>(lds0 + ((b32[sx][0] + half_add) & (WR_XRING - 1)));
            pb[sx][1][1] = wr_read_tr16<0>(lds0 + ((b32[sx][1] + half_add) & (WR_XRING - 1)));
        };
        read_pa(0, lds0 + WR_XRING);
        read_pa(1, lds0 + WR_XRING);
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) read_pb(sx);
        st_read = 1 == S ? 0 : 1;
        for (int s = 0; s < nsteps; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(pa[0][0][0]), "+v"(pa[0][0][1]), "+v"(pa[0][1][0]), "+v"(pa[0][1][1]), "+v"(pa[1][0][0]), "+v"(pa[1][0][1]),
                           "+v"(pa[1][1][0]), "+v"(pa[1][1][1]), "+v"(pb[0][0][0]), "+v"(pb[0][0][1]), "+v"(pb[0][1][0]), "+v"(pb[0][1][1]),
                           "+v"(pb[1][0][0]), "+v"(pb[1][0][1]), "+v"(pb[1][1][0]), "+v"(pb[1][1][1]), "+v"(pb[2][0][0]), "+v"(pb[2][0][1]),
                           "+v"(pb[2][1][0]), "+v"(pb[2][1][1])
                         :
                         : "memory");
            const unsigned stage = lds0 + WR_XRING + st_read * WR_ABYTES;      // dz of step s + 1
            f16x8 fa[2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    const v4i t = {pa[m][kh][0][0], pa[m][kh][0][1], pa[m][kh][1][0], pa[m][kh][1][1]};
                    fa[m][kh] = __builtin_bit_cast(f16x8, t);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                f16x8 fb[2];
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    const v4i t = {pb[sx][kh][0][0], pb[sx][kh][0][1], pb[sx][kh][1][0], pb[sx][kh][1][1]};
                    fb[kh] = __builtin_bit_cast(f16x8, t);
                }
                // the two MFMAs of an accumulator (pixel halves) are kept two issues apart: back to back they serialise on the 64-cycle
                // accumulator latency (first version: 9 - 13 % slower than the 16x16x32 form, SQ_WAIT_INST_ANY + 20 %)
                if constexpr (!NOCOMPUTE) {
                    c32[0][sx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][0], fb[0], c32[0][sx], 0, 0, 0);
                    c32[1][sx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][0], fb[0], c32[1][sx], 0, 0, 0);
                    c32[0][sx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][1], fb[1], c32[0][sx], 0, 0, 0);
                }
                if (sx == 2) {          // the dz fragments of co block 0 are dead: refresh them from the stage of step s + 1
                    __builtin_amdgcn_sched_barrier(0);
                    read_pa(0, stage);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (!NOCOMPUTE) c32[1][sx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][1], fb[1], c32[1][sx], 0, 0, 0);
                if (sx == 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    read_pa(1, stage);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
                b32[sx][0] = lds0 + ((b32[sx][0] + roll_add) & (WR_XRING - 1));
                b32[sx][1] = lds0 + ((b32[sx][1] + roll_add) & (WR_XRING - 1));
                read_pb(sx);
                __builtin_amdgcn_sched_barrier(0);
                if (g3 == sx) YH_WR_BARRIER();
                if constexpr (!NODMA) {
                    if (sx == 1 && s + D < nsteps) issue();
                }
            }
            st_read = st_read + 1 == S ? 0 : st_read + 1;
            if constexpr (!NODMA) wr_wait_keep(min(D - 3, max(0, nsteps - 4 - s)));
        }
        // partial tile: f32x4 number (m * 3 + tap column) * 4 + rq holds accumulator registers 4 rq .. 4 rq + 3 of that 32 x 32 block
        f32x4* part32 = reinterpret_cast<f32x4*>(d.ws) + ((long)split_id * tiles + tile_id) * (24 * NT);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int sx = 0; sx < 3; ++sx)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
                    part32[((m * 3 + sx) * 4 + rq) * NT + tid] = f32x4{c32[m][sx][4 * rq], c32[m][sx][4 * rq + 1], c32[m][sx][4 * rq + 2], c32[m][sx][4 * rq + 3]};
        return;
    } else if constexpr (REFRESH) {
        wr_v2i ra[4][2], rb[6][2];
        const int g3 = wave >> 2;
        // fragments of step 0 (landed and published by the prologue), addresses rolled to step 1
        {
            const unsigned stage = lds0 + WR_XRING;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i][0] = wr_read_tr16<0>(stage + (a_addr0 ^ (i << 5)));
                ra[i][1] = wr_read_tr16<1024>(stage + (a_addr0 ^ (i << 5)));
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    rb[2 * sx][h] = wr_read_tr16<0>(b_abs[sx][h]);
                    rb[2 * sx + 1][h] = wr_read_tr16<0>(b_abs[sx][h] ^ 32);
                }
        }
        st_read = 1 == S ? 0 : 1;
        for (int s = 0; s < nsteps; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(ra[2][0]), "+v"(ra[2][1]),
                           "+v"(ra[3][0]), "+v"(ra[3][1]), "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[1][0]), "+v"(rb[1][1]),
                           "+v"(rb[2][0]), "+v"(rb[2][1]), "+v"(rb[3][0]), "+v"(rb[3][1]), "+v"(rb[4][0]), "+v"(rb[4][1]),
                           "+v"(rb[5][0]), "+v"(rb[5][1])
                         :
                         : "memory");
            const unsigned stage = lds0 + WR_XRING + st_read * WR_ABYTES;      // dz of step s + 1
            typedef int v4i __attribute__((ext_vector_type(4)));
            f16x8 fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const v4i t = {ra[i][0][0], ra[i][0][1], ra[i][1][0], ra[i][1][1]};
                fa[i] = __builtin_bit_cast(f16x8, t);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
#pragma unroll
                for (int bq = 0; bq < 2; ++bq) {
                    const int j = 2 * sx + bq;
                    const v4i tb = {rb[j][0][0], rb[j][0][1], rb[j][1][0], rb[j][1][1]};
                    const f16x8 fb = __builtin_bit_cast(f16x8, tb);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if constexpr (!NOCOMPUTE) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb, acc[i][j], 0, 0, 0);
                        if (j == 5) {           // dz fragment i is dead: refresh it from the stage of step s + 1
                            __builtin_amdgcn_sched_barrier(0);
                            ra[i][0] = wr_read_tr16<0>(stage + (a_addr0 ^ (i << 5)));
                            ra[i][1] = wr_read_tr16<1024>(stage + (a_addr0 ^ (i << 5)));
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // x fragment j is dead: refresh it (step s + 1's rows; block bq of tap column sx; the pair's addresses roll
                    // here, between the MFMAs, not in a block of their own)
                    if (bq == 0) {
                        b_abs[sx][0] = lds0 + ((b_abs[sx][0] + roll_add) & (WR_XRING - 1));
                        b_abs[sx][1] = lds0 + ((b_abs[sx][1] + roll_add) & (WR_XRING - 1));
                    }
                    rb[j][0] = wr_read_tr16<0>(b_abs[sx][0] ^ (32 * bq));
                    rb[j][1] = wr_read_tr16<0>(b_abs[sx][1] ^ (32 * bq));
                    __builtin_amdgcn_sched_barrier(0);
                    if (bq == 1 && g3 == sx) YH_WR_BARRIER();       // this group's one barrier of the step: after C_1 / C_3 / C_5
                    if constexpr (!NODMA) {
                        // this wave's LDS-DMA piece of step s + D, behind column 2 (after group 0's barrier point, so that every
                        // group issues it after ITS barrier of the step ... no: groups 1, 2 issue it before theirs - the stage it
                        // overwrites was last read two steps ago either way, see the hazard note above)
                        if (j == 2 && s + D < nsteps) issue();
                    }
                }
            }
            st_read = st_read + 1 == S ? 0 : st_read + 1;
            // ---- this wave's piece of step s + 3 must have landed (its piece of step s + D was issued behind column 2)
            if constexpr (!NODMA) wr_wait_keep(min(D - 3, max(0, nsteps - 4 - s)));
        }
    } else if constexpr (FREE) {
        wr_v2i ra[4][2], rb[6][2];
        auto seg_r1 = [&]() {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    rb[2 * sx][h] = NOCOMPUTE ? wr_v2i{(int)b_abs[sx][h], h} : wr_read_tr16<0>(b_abs[sx][h]);
                    rb[2 * sx + 1][h] = NOCOMPUTE ? wr_v2i{(int)b_abs[sx][h], sx} : wr_read_tr16<0>(b_abs[sx][h] ^ 32);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[1][0]), "+v"(rb[1][1]), "+v"(rb[2][0]), "+v"(rb[2][1]),
                           "+v"(rb[3][0]), "+v"(rb[3][1]), "+v"(rb[4][0]), "+v"(rb[4][1]), "+v"(rb[5][0]), "+v"(rb[5][1])
                         :
                         : "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        auto seg_r2 = [&](int s) {
            if constexpr (!NODMA) {
                if (s + D < nsteps) issue();
            }
            const unsigned stage = lds0 + WR_XRING + st_read * WR_ABYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i][0] = NOCOMPUTE ? wr_v2i{(int)stage, i} : wr_read_tr16<0>(stage + (a_addr0 ^ (i << 5)));
                ra[i][1] = NOCOMPUTE ? wr_v2i{(int)stage, i} : wr_read_tr16<1024>(stage + (a_addr0 ^ (i << 5)));
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) b_abs[sx][h] = lds0 + ((b_abs[sx][h] + roll_add) & (WR_XRING - 1));
            st_read = st_read + 1 == S ? 0 : st_read + 1;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!NODMA) wr_wait_keep(min(D - 2, max(0, nsteps - 3 - s)));
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(ra[2][0]), "+v"(ra[2][1]),
                           "+v"(ra[3][0]), "+v"(ra[3][1])
                         :
                         : "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        auto seg_mm = [&]() {
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
            if constexpr (NOCOMPUTE) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
                for (int j = 0; j < 6; ++j) asm volatile("" ::"v"(fb[j]));
            } else {
                if (a.prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                if (a.prio) __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // one instruction stream for every wave (R1, R2, MFMA, R1, ..): a group differs only in WHERE its one barrier per step falls -
        // group 2 after R1, group 0 after R2, group 1 after the MFMAs (and once before its first segment instead of after its last)
        const int g3 = wave >> 2;
        if (g3 == 1) YH_WR_BARRIER();
        for (int s = 0; s < nsteps; ++s) {
            seg_r1();
            if (g3 == 2) YH_WR_BARRIER();
            seg_r2(s);
            if (g3 == 0) YH_WR_BARRIER();
            seg_mm();
            if (g3 == 1 && s + 1 < nsteps) YH_WR_BARRIER();
        }
    } else if constexpr (SPLIT) {
        for (int s = 0; s < nsteps; ++s) {
            wr_v2i ra[4][2], rb[6][2];
            // ---- R1: x fragments
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    rb[2 * sx][h] = NOCOMPUTE ? wr_v2i{(int)b_abs[sx][h], h} : wr_read_tr16<0>(b_abs[sx][h]);
                    rb[2 * sx + 1][h] = NOCOMPUTE ? wr_v2i{(int)b_abs[sx][h], sx} : wr_read_tr16<0>(b_abs[sx][h] ^ 32);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[1][0]), "+v"(rb[1][1]), "+v"(rb[2][0]), "+v"(rb[2][1]),
                           "+v"(rb[3][0]), "+v"(rb[3][1]), "+v"(rb[4][0]), "+v"(rb[4][1]), "+v"(rb[5][0]), "+v"(rb[5][1])
                         :
                         : "memory");
            YH_WR_STAMP(0);      // x fragments returned
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            YH_WR_STAMP(1);      // barrier 1
            // ---- R2: this wave's LDS-DMA piece, dz fragments, address roll, counted wait
            if constexpr (!NODMA) {
                if (s + D < nsteps) issue();
                YH_WR_STAMP(6);      // piece issued
            }
            const unsigned stage = lds0 + WR_XRING + st_read * WR_ABYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i][0] = NOCOMPUTE ? wr_v2i{(int)stage, i} : wr_read_tr16<0>(stage + (a_addr0 ^ (i << 5)));
                ra[i][1] = NOCOMPUTE ? wr_v2i{(int)stage, i} : wr_read_tr16<1024>(stage + (a_addr0 ^ (i << 5)));
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) b_abs[sx][h] = lds0 + ((b_abs[sx][h] + roll_add) & (WR_XRING - 1));
            st_read = st_read + 1 == S ? 0 : st_read + 1;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!NODMA) wr_wait_keep(min(D - 2, max(0, nsteps - 3 - s)));
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(ra[2][0]), "+v"(ra[2][1]),
                           "+v"(ra[3][0]), "+v"(ra[3][1])
                         :
                         : "memory");
            YH_WR_STAMP(4);      // R2's work
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            YH_WR_STAMP(5);      // barrier 2
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
            if constexpr (NOCOMPUTE) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
                for (int j = 0; j < 6; ++j) asm volatile("" ::"v"(fb[j]));
            } else {
                if (a.prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                if (a.prio) __builtin_amdgcn_s_setprio(0);
            }
            YH_WR_STAMP(2);      // 24 MFMAs issued
            YH_WR_BARRIER();
            YH_WR_STAMP(3);      // barrier 3
        }
    } else
    for (int s = 0; s < nsteps; ++s) {
        // ---- LOAD
        const unsigned stage = lds0 + WR_XRING + st_read * WR_ABYTES;
        wr_v2i ra[4][2], rb[6][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i][0] = NOCOMPUTE ? wr_v2i{(int)stage, i} : wr_read_tr16<0>(stage + (a_addr0 ^ (i << 5)));
            ra[i][1] = NOCOMPUTE ? wr_v2i{(int)stage, i} : wr_read_tr16<1024>(stage + (a_addr0 ^ (i << 5)));
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                rb[2 * sx][h] = NOCOMPUTE ? wr_v2i{(int)b_abs[sx][h], h} : wr_read_tr16<0>(b_abs[sx][h]);
                rb[2 * sx + 1][h] = NOCOMPUTE ? wr_v2i{(int)b_abs[sx][h], sx} : wr_read_tr16<0>(b_abs[sx][h] ^ 32);
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
        YH_WR_STAMP(0);      // fragment reads issued and returned
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        YH_WR_STAMP(1);      // barrier 1
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
        if constexpr (NOCOMPUTE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
            for (int j = 0; j < 6; ++j) asm volatile("" ::"v"(fb[j]));
        } else {
            if (a.prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            if (a.prio) __builtin_amdgcn_s_setprio(0);
        }
        YH_WR_STAMP(2);      // 24 MFMAs issued
        YH_WR_BARRIER();
        YH_WR_STAMP(3);      // barrier 2
        // ---- stream interval (the other two groups read / multiply)
        if constexpr (!NODMA) {
            if (s + D < nsteps) issue();
            YH_WR_STAMP(6);      // piece issued
            wr_wait_keep(min(D - 2, max(0, nsteps - 3 - s)));
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) b_abs[sx][h] = lds0 + ((b_abs[sx][h] + roll_add) & (WR_XRING - 1));
        st_read = st_read + 1 == S ? 0 : st_read + 1;
        YH_WR_STAMP(4);      // stream interval's work (issue, counted wait, address roll)
        YH_WR_BARRIER();
        YH_WR_STAMP(5);      // barrier 3
    }
    if constexpr (!FREE && !REFRESH) for (int k = grp; k < 2; ++k) YH_WR_BARRIER();      // every wave has executed the same number of barriers
#undef YH_WR_BARRIER

    if constexpr (TIMING) {
        if (blockIdx.x == 8 && (wave & 3) == 0 && lane == 0 && a.timing) {
#pragma unroll
            for (int k = 0; k < 7; ++k) a.timing[(wave >> 2) * 8 + k] = tsum[k];
            a.timing[(wave >> 2) * 8 + 7] = (unsigned long long)nsteps;
        }
    }
    f32x4* part = reinterpret_cast<f32x4*>(d.ws) + ((long)split_id * tiles + tile_id) * (24 * NT);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) part[(i * 6 + j) * NT + tid] = acc[i][j];
}

// Partial tiles -> dw.  One thread per (tile, fragment, lane slot); the splits of a group are added in index order (four running sums
// of every fourth split), groups meet in fp32 atomics (one group for <= 16 splits: plain accumulate, bit-reproducible).
__global__ __launch_bounds__(768) void wgrad_roll_reduce_kernel(const RollArgs a, int splits, int per_group) {
    constexpr int NT = 768;
    const yh_wgrad_desc& d = a.d;
    const int tiles = a.tiles_m * a.tiles_n;
    const int tile = blockIdx.x / 24, ij = blockIdx.x % 24;
    const int i = ij / 6, j = ij % 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / 6, wn = wave - wm * 6;
    const f32x4* part = reinterpret_cast<const f32x4*>(d.ws) + ((long)tile * 24 + ij) * NT + tid;
    const long stride = (long)tiles * 24 * NT;
    const int sA = blockIdx.y * per_group, sB = min(sA + per_group, splits);
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0, v2 = v0, v3 = v0;
    int sp = sA;
    for (; sp + 3 < sB; sp += 4) {
        v0 += part[sp * stride];
        v1 += part[(sp + 1) * stride];
        v2 += part[(sp + 2) * stride];
        v3 += part[(sp + 3) * stride];
    }
    for (; sp < sB; ++sp) v0 += part[sp * stride];
    const f32x4 v = (v0 + v1) + (v2 + v3);
    const int tm = tile % a.tiles_m, tn = tile / a.tiles_m;
    const int tap = (wn >> 1) * 3 + (j >> 1);
    const int ci = tn * 64 + (wn & 1) * 32 + (j & 1) * 16 + (lane & 15);
    if (ci >= a.cin_w) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = tm * 128 + wm * 64 + i * 16 + 4 * (lane >> 4) + r;
        if (co >= d.cout) continue;
        float* dst = d.dw + ((long)co * a.cin_w + ci) * 9 + tap;
        if (gridDim.y == 1) *dst += v[r];
        else atomicAdd(dst, v[r]);
    }
}

// Partial tiles -> dw, second form (the default).  The first form above scatters: a lane's four values are four output channels of one
// (ci, tap), 36 bytes from its neighbour's - 1.2 M four-byte read-modify-writes (or atomics) per 76 x 76 layer - and measured 41 us per
// launch for 75 MB (1.9 TB/s), a sixth of the whole weight gradient.  Here one workgroup owns a [16 co][16 ci][9 taps] block of dw:
// wave t (of 9) sums the fragment that holds tap t of the block over the splits (eight independent 16-byte loads in flight per lane,
// fixed order), the block is transposed through LDS and leaves as 16 rows of 576 contiguous bytes.  Workgroups = tiles x 32 x G; G > 1
// (few tiles: the 76 x 76 and 152 x 152 layers) splits the pixel splits over G workgroups that meet in row-contiguous atomics.
template <bool VEC, bool M32>
__global__ __launch_bounds__(576) void wgrad_roll_reduce2_kernel(const RollArgs a, int splits, int per_group, int sstep, int native) {
    constexpr int NT = 768;
    const yh_wgrad_desc& d = a.d;
    __shared__ __attribute__((aligned(16))) float blk[16][148];
    const int tiles = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    const int b = bid & 1, hh = (bid >> 1) & 1, i = (bid >> 2) & 3, wm = (bid >> 4) & 1, tile = bid >> 5;
    const int tid = threadIdx.x, lane = tid & 63;
    const int tap = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int trow = tap / 3, tcol = tap - trow * 3;
    const int w_src = wm * 6 + trow * 2 + hh;
    // 16x16x32 partial tiles: fragment (i, 2 tcol + b), lane = (co quad, ci); 32x32x16: block (i >> 1, tcol), register quad 2 (i & 1) +
    // (lane >> 5) of source lane 32 ((lane >> 4) & 1) + 16 b + (lane & 15), whose four values are rows e + 8 (lane >> 5) + 4 ((lane >> 4) & 1)
    const int ij = M32 ? ((i >> 1) * 3 + tcol) * 4 + 2 * (i & 1) + (lane >> 5) : i * 6 + tcol * 2 + b;
    const int src_lane = M32 ? 32 * ((lane >> 4) & 1) + 16 * b + (lane & 15) : lane;
    const int row_base = M32 ? 8 * (lane >> 5) + 4 * ((lane >> 4) & 1) : 4 * (lane >> 4);
    f32x4* part = reinterpret_cast<f32x4*>(d.ws) + ((long)tile * 24 + ij) * NT + w_src * 64 + src_lane;
    const long stride = (long)tiles * 24 * NT * sstep;      // sstep > 1: the second pass, over the groups' in-place sums
    const int sA = blockIdx.y * per_group, sB = min(sA + per_group, splits);
    f32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    int sp = sA;
    for (; sp + 7 < sB; sp += 8) {
        f32x4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = part[(sp + k) * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += t[k];
    }
    for (int k = 0; sp < sB; ++sp, ++k) v[k & 7] += part[sp * stride];
    const f32x4 sum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    if (native) {      // deterministic form (common.h), first pass: the group's sum replaces its first partial tile - every thread reads and
        part[sA * stride] = sum;      // writes its own element only - and a second launch with ONE group adds the groups in order
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) blk[row_base + r][(lane & 15) * 9 + tap] = sum[r];
    __syncthreads();
    const int row = tid / 36, c4 = tid - row * 36;
    const int tm = tile % a.tiles_m, tn = tile / a.tiles_m;
    const int co = tm * 128 + wm * 64 + i * 16 + row, ci0 = tn * 64 + hh * 32 + b * 16;
    float* dst = d.dw + ((long)co * a.cin_w + ci0) * 9 + c4 * 4;
    const f32x4 o = *reinterpret_cast<const f32x4*>(&blk[row][c4 * 4]);
    if (gridDim.y == 1) {       // this workgroup owns the rows: plain accumulate (dw is accumulated into; the caller zeroes it)
        if constexpr (VEC) {
            f32x4 cur = *reinterpret_cast<f32x4*>(dst);
            cur += o;
            *reinterpret_cast<f32x4*>(dst) = cur;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e] += o[e];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dst + e, o[e]);
    }
}

// geometry of the rolling form; false when the layer does not qualify (host code, no launch)
bool wgrad_roll_geometry(const yh_wgrad_desc* d, RollArgs* pa, int* psplits, size_t* plds) {
    RollArgs& a = *pa;
    const char* mode_env = getenv("YH_WGRAD_HALO");
    const int force = mode_env ? atoi(mode_env) : YH_WGRAD_HALO_DEFAULT;
    if (d->dtype != YH_F16 || d->splits == -1) return false;
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->ho != d->h || d->wo != d->w_in) return false;
    if (d->cout % 128 || d->cin % 64 || d->w_in < 16 || d->h < 2) return false;
    if (d->cin_w > 0 && d->cin_w != d->cin) return false;
    const int Wp = d->w_in + 1;
    const int JL = (2 * Wp + 2 + 31) / 32;
    // dz ring stages.  The x ring bounds them at 16 - JL, the LDS at 10; measured (profiles/r05_wgrad_roll_ab.txt): 6 stages = four
    // steps (48 KB) in flight are the fastest on every layer of YOLOv3-608 - 8 and 10 lose 1 - 5 %: a CU is served ~11 B / clk of LDS-DMA
    // however deep its queue is (tools/probe/run_probe.py), more pieces in flight only lengthen the in-order queue each wave waits on
    const int s_max = 16 - JL < 10 ? 16 - JL : 10;
    int S = s_max < 8 ? s_max : 8;
    { const char* e = getenv("YH_WGRAD_ROLL_STAGES"); if (e && atoi(e) >= 4 && atoi(e) <= s_max) S = atoi(e); }   // A/B knob
    if (S < 4) return false;
    const long Q = (long)d->n * (d->h + 1) * Wp;
    if (Q + 4096 >= 0x7fffffffL) return false;
    a.d = *d;
    a.cin_w = d->cin;
    a.tiles_m = d->cout / 128;
    a.tiles_n = d->cin / 64;
    a.steps_total = (int)((Q + 31) / 32);
    a.S = S;
    a.JL = JL;
    a.q32 = 32 / Wp;
    a.r32 = 32 - a.q32 * Wp;
    { const char* e = getenv("YH_WGRAD_HALO_NOSTAGGER"); a.nostagger = e && atoi(e) ? 1 : 0; }
    const int tiles = a.tiles_m * a.tiles_n;
    // Every layer that qualifies takes this form (YH_WGRAD_HALO = 2 and 3 alike): with the fragment-refresh order (ORDER 3) it beats the
    // round-3 kernel on all four stage shapes of YOLOv3-608 batch 64 - 76^2 0.201 / 0.241 ms, 38^2 0.198 / 0.216, 19^2 0.211 / 0.241, 152^2
    // 0.221 / 0.316 (profiles/r05_wgrad_roll_order3_ab.txt); the per-layer choice of the ORDER-2 days (few-tile layers only) is gone.
    // Split groups of the reduce launch: summed in place and added by a one-group launch (deterministic, common.h), or - YH_DETERMINISTIC=0 -
    // meeting in fp32 atomics on the few-tile layers.
    (void)force;
    int splits = d->splits > 0 ? d->splits : 256 / tiles;          // one workgroup per CU
    {
        const char* e = getenv("YH_WGRAD_HALO_WGS");       // A/B and test knob: total workgroups aimed for
        if (e && d->splits <= 0) splits = atoi(e) / tiles;
        else if (d->splits <= 0 && splits > a.steps_total / 8) splits = a.steps_total / 8;      // library's choice: >= 8 steps per workgroup
    }
    if (splits < 1) splits = 1;
    if (splits > a.steps_total) splits = a.steps_total;
    a.sps = (a.steps_total + splits - 1) / splits;
    *psplits = (a.steps_total + a.sps - 1) / a.sps;
    *plds = (size_t)WR_XRING + (size_t)S * WR_ABYTES;
    return true;
}

int64_t wgrad_roll_workspace(const yh_wgrad_desc* d) {
    RollArgs a;
    int splits;
    size_t lds;
    if (!wgrad_roll_geometry(d, &a, &splits, &lds)) return 0;
    return (int64_t)splits * a.tiles_m * a.tiles_n * 128 * 576;
}

// YH_EUNSUPPORTED: the layer does not qualify or the workspace is too small (the caller falls back to the other forms)
int launch_wgrad_roll(const yh_wgrad_desc* d, hipStream_t st) {
    RollArgs a;
    int splits;
    size_t lds;
    if (!wgrad_roll_geometry(d, &a, &splits, &lds)) return YH_EUNSUPPORTED;
    const int tiles = a.tiles_m * a.tiles_n;
    if (!d->ws || d->ws_floats < (int64_t)splits * tiles * 128 * 576) return YH_EUNSUPPORTED;
    bool m32 = false;
    const char* red_env0 = getenv("YH_WGRAD_ROLL_REDUCE");
    const char* abl_env = getenv("YH_WGRAD_ROLL_ABL");      // profiling only
    const int abl = abl_env ? atoi(abl_env) : 0;
    {
        const char* order_env = getenv("YH_WGRAD_ROLL_ORDER");     // A/B knob: 0 = one read interval, 1 = split reads, 2 = one barrier per step
        const int order = order_env ? atoi(order_env) : 3;      // 3 = fragments refreshed between the MFMAs (no LOAD segment)
        { const char* e = getenv("YH_WGRAD_ROLL_PRIO"); a.prio = e ? atoi(e) : 0; }   // s_setprio 1 around the MFMAs measured 1 - 3 % slower
        const char* m32_env = getenv("YH_WGRAD_ROLL_MFMA32");       // A/B knob: v_mfma_f32_32x32x16_f16 in the refresh order
        m32 = order == 3 && abl == 0 && (m32_env ? atoi(m32_env) != 0 : YH_WGRAD_ROLL_MFMA32_DEFAULT != 0) &&
              !(red_env0 && atoi(red_env0) == 1);
        auto kern = m32 ? conv_wgrad_roll_kernel<0, 3, true> : order == 3 ? conv_wgrad_roll_kernel<0, 3> : order == 2 ? conv_wgrad_roll_kernel<0, 2> : order == 1 ? conv_wgrad_roll_kernel<0, 1> : conv_wgrad_roll_kernel<0, 0>;
        if (abl == 1) kern = order == 3 ? conv_wgrad_roll_kernel<1, 3> : order == 2 ? conv_wgrad_roll_kernel<1, 2> : order == 1 ? conv_wgrad_roll_kernel<1, 1> : conv_wgrad_roll_kernel<1, 0>;
        if (abl == 2) kern = order == 2 ? conv_wgrad_roll_kernel<2, 2> : conv_wgrad_roll_kernel<2, 0>;
#define YH_WR_PICK(A) if (abl == A) kern = order == 1 ? conv_wgrad_roll_kernel<A, 1> : conv_wgrad_roll_kernel<A, 0>
        YH_WR_PICK(8); YH_WR_PICK(9); YH_WR_PICK(10); YH_WR_PICK(12);
#undef YH_WR_PICK
        a.timing = nullptr;
        if (abl & 8) {      // the stamps land behind the partial tiles (the caller's workspace holds 64 more floats: tools/wgrad_ab.py)
            if (d->ws_floats < (int64_t)splits * tiles * 128 * 576 + 64) return YH_EINVAL;
            a.timing = reinterpret_cast<unsigned long long*>(d->ws + (int64_t)splits * tiles * 128 * 576);
        }
        const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) return (int)e;
        reduce_guard_workspace(st);       // a previous weight gradient's reduce may still be reading the workspace (common.h AsyncReduce)
        hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * splits)), dim3(768), lds, st, a);
    }
    const hipStream_t ms = st;
    st = reduce_begin(ms);                // the reduce launches below: on the plan's reduce stream when it has one
    const char* red_env = getenv("YH_WGRAD_ROLL_REDUCE");      // A/B knob: 1 = the scattering first form
    if (red_env && atoi(red_env) == 1) {
        int groups = (splits + 15) / 16;
        if (groups > 32) groups = 32;
        const int per_group = (splits + groups - 1) / groups;
        groups = (splits + per_group - 1) / per_group;
        hipLaunchKernelGGL(wgrad_roll_reduce_kernel, dim3(tiles * 24, groups), dim3(768), 0, st, a, splits, per_group);
        reduce_end(ms, st);
    } else {
        int groups = (256 + tiles * 32 - 1) / (tiles * 32);       // >= one workgroup per CU where the splits allow it
        if (groups > splits / 4) groups = splits / 4;
        if (groups < 1) groups = 1;
        const int per_group = (splits + groups - 1) / groups;
        groups = (splits + per_group - 1) / per_group;
        auto reduce = [&](int ngroups, int nsplits, int per, int sstep, int native) {
            const dim3 rg(tiles * 32, ngroups);
            if (m32) {
                if (aligned16(d->dw)) hipLaunchKernelGGL((wgrad_roll_reduce2_kernel<true, true>), rg, dim3(576), 0, st, a, nsplits, per, sstep, native);
                else hipLaunchKernelGGL((wgrad_roll_reduce2_kernel<false, true>), rg, dim3(576), 0, st, a, nsplits, per, sstep, native);
            } else {
                if (aligned16(d->dw)) hipLaunchKernelGGL((wgrad_roll_reduce2_kernel<true, false>), rg, dim3(576), 0, st, a, nsplits, per, sstep, native);
                else hipLaunchKernelGGL((wgrad_roll_reduce2_kernel<false, false>), rg, dim3(576), 0, st, a, nsplits, per, sstep, native);
            }
        };
        if (deterministic() && groups > 1) {      // few-tile layers (76 x 76, 152 x 152): the split groups no longer meet in fp32 atomics
            reduce(groups, splits, per_group, 1, 1);
            reduce(1, groups, groups, per_group, 0);
        } else {
            reduce(groups, splits, per_group, 1, 0);
        }
        reduce_end(ms, st);
    }
    return check_launch();
}

}  // namespace yh
