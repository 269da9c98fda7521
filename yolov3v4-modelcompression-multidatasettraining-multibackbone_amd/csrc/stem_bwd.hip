// Backward of the FIRST conv block (Conv2d 3x3 / s1 / p1 on the fp32 NCHW image -> train-mode BatchNorm -> activation) in one pass
// over dy and z (include/yolo_hip.h yh_stem_bwd; reference: autograd through models.py:92-113 for block 0).
//
// Why a kernel of its own: the first block's tensors are the largest of the net (608 x 608 x 32 channels x batch 64 = 1.5 GB each)
// and its dz has exactly one reader, its own weight gradient - there is no data gradient into the image.  The generic backward
// reads dy and z twice (reduce, apply), writes dz, converts the image to NHWC and reads dz again: 13.5 GB of traffic, 2.34 ms of
// the 58 ms step (profiles/r03_train_layers_final.txt: dbn0 0.56 + dbnx0 0.93 + image0 0.28 + wgrad0 0.57).  Here every sum over
// pixels that the result needs is taken in ONE pass (3.3 GB):
//     S1 = sum g, S2 = sum g xhat                      (g = dy act'(u), xhat = (z - mean) invstd; = dbeta, dgamma)
//     Q = g X^T, R = xhat X^T, SX = sum X              (X = im2col of the image, 27 columns)
//     dW = gamma invstd (Q - S1/P SX - S2/P R)         (stem_bwd_final_kernel, in double)
// HBM-bound byte work: the three products are one small GEMM [2 cout + 1 rows] x [27 columns] with the pixel index as K, run on
// MFMA only because 2 x 32 x 27 fused multiply-adds per pixel would cost more VALU time than the bytes take to arrive.
//
// Work unit = one wave x one SEGMENT of 32 consecutive pixels of an image row.  Per segment the wave
//   * loads dy and z of the 32 pixels (16 bytes = 8 channels per lane per load, whole 64-byte pixel rows per 4 lanes), turns them
//     into g and xhat, adds them to its per-lane S1 / S2 and writes them as f16 rows [pixel][g | xhat | ones] of its private LDS
//     tile; the MFMA A fragments (row = channel, K = 8 consecutive pixels) are read back TRANSPOSED by ds_read_b64_tr_b16, the
//     idiom of conv_wgrad.hip;
//   * loads the 3 x cin image row pieces the segment's taps touch (34 floats each, coalesced) and writes them as f16 into the
//     LDS tile [column k = (r, s, ci)][pixel]: shifted copies per horizontal tap, so that a B fragment (column k, 8 consecutive
//     pixels) is ONE aligned ds_read_b128 - no transpose on this side; padding and the pixels beyond the row end are zeros;
//   * runs (2 cout / 16 + 1) x 2 MFMAs 16x16x32 into accumulators that live for all of the wave's segments.
// The next segment's global loads are issued before the current segment is processed.  A wave's LDS traffic is private: program
// order is the only synchronisation (no workgroup barrier inside the loop).
#include "common.h"

namespace yh {

typedef unsigned int sb_u32x4 __attribute__((ext_vector_type(4)));
typedef int sb_v2i __attribute__((ext_vector_type(2)));
typedef int sb_v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sb_act_grad(float u, int act, float slope) {
    switch (act) {
        case YH_ACT_LEAKY: return u > 0.f ? 1.f : slope;
        case YH_ACT_RELU: return u > 0.f ? 1.f : 0.f;
        case YH_ACT_RELU6: return (u > 0.f && u < 6.f) ? 1.f : 0.f;
        case YH_ACT_HSWISH: return u <= -3.f ? 0.f : (u >= 3.f ? 1.f : (2.f * u + 3.f) / 6.f);
        case YH_ACT_MISH: {      // as train.hip act_grad
            const float e = expf(fminf(u, 20.f));
            const float n = e * (e + 2.f);
            const float t = u > 20.f ? 1.f : n * __builtin_amdgcn_rcpf(n + 2.f);
            const float sg = u > 20.f ? 1.f : e * __builtin_amdgcn_rcpf(e + 1.f);
            return t + u * sg * (1.f - t * t);
        }
        default: return 1.f;
    }
}

__device__ __forceinline__ sb_v2i sb_read_tr16(unsigned lds_byte_addr) {
    sb_v2i r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(lds_byte_addr) : "memory");
    return r;
}

// partial-row layout (floats): [2 C][32] products (rows 0 .. C-1: Q, rows C .. 2C-1: R) | SX[32] | S1[C] | S2[C]
template <int C> struct SbRow { static constexpr int PROD = 2 * C * 32, SX = PROD, S1 = PROD + 32, S2 = S1 + C, SIZE = S2 + C; };

// ACT: YH_ACT_LEAKY / YH_ACT_MISH fixed at compile time (the YOLOv3 / v4 first blocks); -1 = the run-time switch (other activations)
template <int ACT> __device__ __forceinline__ float sb_dact(float u, int act, float slope) {
    if constexpr (ACT == YH_ACT_LEAKY) return u > 0.f ? 1.f : slope;
    else if constexpr (ACT == YH_ACT_MISH) return mish_grad_fast(u);      // fp16 tensors: as train.hip act_grad_t<f16> (common.h)
    else return act == YH_ACT_MISH ? mish_grad_fast(u) : sb_act_grad(u, act, slope);
}

template <int C, int CIN, int ACT>
__global__ __launch_bounds__(256, 2) void stem_bwd_partial_kernel(const yh_stem_bwd_desc d, const int segs_per_row, const int nseg) {
    constexpr int AROWS = 2 * C + 16;           // channels of an A row: g | xhat | ones block (first entry 1, rest 0)
    constexpr int PA = AROWS * 2;               // bytes per pixel row of the A tile (160 / 96: multiples of 16)
    constexpr int PB = 96;                      // bytes per column row of the X tile: pixels -2 .. 33 (pixel 0 at byte 16), so that the
                                                // three shifted copies of a loaded row piece are written without any per-lane condition
    constexpr int NI = AROWS / 16;              // MFMA row blocks
    constexpr int UPP = C / 8;                  // 16-byte units per pixel of dy / z
    constexpr int PXP = 64 / UPP;               // pixels covered by one load instruction of the wave
    constexpr int NP = 32 / PXP;                // load instructions per operand per segment
    constexpr int A_BYTES = 32 * PA, B_BYTES = 32 * PB, WAVE_BYTES = A_BYTES + B_BYTES;
    constexpr int NRC = 3 * CIN;                // image row pieces per segment: (r, ci); 9 CIN <= 27 columns
    typedef SbRow<C> Row;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * WAVE_BYTES];
    __shared__ float part[Row::SIZE];
    typedef void __attribute__((address_space(3))) * lptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const At = smem + wave * WAVE_BYTES;
    unsigned char* const Bt = At + A_BYTES;
    const unsigned at_lds = (unsigned)(uintptr_t)(lptr_t)At;

    // ---- one-time LDS set-up: zero the wave's tiles, then the ones column
    for (int i = lane; i < WAVE_BYTES / 16; i += 64) reinterpret_cast<sb_u32x4*>(At)[i] = sb_u32x4{0u, 0u, 0u, 0u};
    __builtin_amdgcn_wave_barrier();
    if (lane < 32) *reinterpret_cast<f16*>(At + lane * PA + 2 * C * 2) = (f16)1.f;
    for (int i = tid; i < Row::SIZE; i += 256) part[i] = 0.f;

    // ---- per-lane constants: this lane always handles channel unit `cu` (8 channels) of pixels (lane / UPP) + PXP * pass.
    // xhat = z * is - mu * is, u = (ga * is) z + (be - ga mu is): two fused multiply-adds per element, on packed fp32 pairs
    const int cu = lane % UPP, pl = lane / UPP;
    f32x2 isv[4], nmu[4], gis[4], bsh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int ch = cu * 8 + 2 * e + k;
            const float is = d.invstd[ch], mu = d.mean[ch], ga = d.gamma[ch], be = d.beta[ch];
            isv[e][k] = is;
            nmu[e][k] = -mu * is;
            gis[e][k] = ga;
            bsh[e][k] = be;
        }
    f32x2 s1[4], s2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) s1[e] = s2[e] = f32x2{0.f, 0.f};
    f32x4 acc[NI][2];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};

    const f16* const dyg = reinterpret_cast<const f16*>(d.dy) + cu * 8;
    const f16* const zg = reinterpret_cast<const f16*>(d.z) + cu * 8;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;      // segment indices fit 31 bits (launcher)
    // position of the current / next segment, advanced without divisions: image n, row h, segment sg of the row
    const int d_row = nw / segs_per_row, d_sg = nw - d_row * segs_per_row;
    struct Pos { int n, h, sg; };
    auto advance = [&](Pos& p) {
        p.sg += d_sg;
        p.h += d_row;
        if (p.sg >= segs_per_row) { p.sg -= segs_per_row; ++p.h; }
        while (p.h >= d.h) { p.h -= d.h; ++p.n; }
    };

    struct Seg { sb_u32x4 dy[NP], z[NP]; float img[9]; };
    const int HW = d.h * d.w_in;
    auto load_seg = [&](const Pos& ps, Seg& g) {
        const int w0 = ps.sg * 32;
        const long p0 = ((long)ps.n * d.h + ps.h) * d.w_in + w0;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int px = pl + q * PXP;
            const long p = w0 + px < d.w_in ? p0 + px : p0;      // a valid address; the values are masked in process()
            g.dy[q] = *reinterpret_cast<const sb_u32x4*>(dyg + p * d.lddy);
            g.z[q] = *reinterpret_cast<const sb_u32x4*>(zg + p * d.ldz);
        }
        const int wi = w0 - 1 + lane;                      // lanes 0 .. 33 cover the 34 image columns the three taps touch
        const int wc = min(max(wi, 0), d.w_in - 1);
        const bool wok = lane < 34 && wi >= 0 && wi < d.w_in;
        const float* const xin = d.x + (long)ps.n * CIN * HW + wc;
#pragma unroll
        for (int rc = 0; rc < 9; ++rc) {
            g.img[rc] = 0.f;
            if constexpr (true) {
                if (rc < NRC) {
                    const int r = rc / CIN, ci = rc - r * CIN;
                    const int hi = ps.h + r - 1;
                    const int hc = min(max(hi, 0), d.h - 1);
                    // always loaded (clamped address) and masked by a MULTIPLICATION: a select would let the compiler predicate the
                    // load, and a branch around a load in the pipelined loop makes its wait-count pass give up (vmcnt(0) at joins)
                    const float v = xin[ci * HW + hc * d.w_in];
                    g.img[rc] = v * ((wok && hi >= 0 && hi < d.h) ? 1.f : 0.f);
                }
            }
        }
    };

    const int q16 = lane & 15, g4 = lane >> 4;
    const unsigned a_rd0 = at_lds + (8 * g4 + (q16 >> 2)) * PA + 4 * (q16 & 3) * 2;      // half h adds 4 rows
    const unsigned char* const b_rd = Bt + q16 * PB + 16 + g4 * 16;
    unsigned char* const b_wr = Bt + 16 + lane * 2;
    unsigned char* const a_wr = At + pl * PA + cu * 16;
    auto process = [&](const Pos& ps, Seg& g) {
        const int w0 = ps.sg * 32;
        const int valid = d.w_in - w0;                     // pixels of the segment inside the row (>= 1)
        // ---- A rows: g and xhat of this lane's 8 channels.  Pixels beyond the row end: dy = 0 makes g = 0 (their xhat is finite
        // garbage that only meets zeroed X entries and g = 0)
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int px = pl + q * PXP;
            if (px >= valid) g.dy[q] = sb_u32x4{0u, 0u, 0u, 0u};
            const f16x8 dv = __builtin_bit_cast(f16x8, g.dy[q]);
            const f16x8 zv = __builtin_bit_cast(f16x8, g.z[q]);
            f16x8 gv, xv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x2 zf = {(float)zv[2 * e], (float)zv[2 * e + 1]};
                const f32x2 df = {(float)dv[2 * e], (float)dv[2 * e + 1]};
                const f32x2 xh = __builtin_elementwise_fma(zf, isv[e], nmu[e]);
                const f32x2 u = __builtin_elementwise_fma(gis[e], xh, bsh[e]);
                const f32x2 m = {sb_dact<ACT>(u[0], d.act, d.slope), sb_dact<ACT>(u[1], d.act, d.slope)};
                const f32x2 gg = df * m;
                s1[e] += gg;
                s2[e] = __builtin_elementwise_fma(gg, xh, s2[e]);
                gv[2 * e] = (f16)gg[0];
                gv[2 * e + 1] = (f16)gg[1];
                xv[2 * e] = (f16)xh[0];
                xv[2 * e + 1] = (f16)xh[1];
            }
            *reinterpret_cast<f16x8*>(a_wr + q * PXP * PA) = gv;
            *reinterpret_cast<f16x8*>(a_wr + q * PXP * PA + C * 2) = xv;
        }
        // ---- X columns: image value of column (w0 - 1 + lane) goes to pixel j = lane - sft of the column rows (r, sft, ci).
        // Columns beyond the row end were loaded as zeros; the one entry that is NOT zero by that rule is tap sft = 0 of the first
        // pixel beyond the row (it reads the last real column), masked here
        const bool keep0 = lane < valid;
        if (lane < 34) {
#pragma unroll
            for (int rc = 0; rc < 9; ++rc) {
                if (rc < NRC) {
                    const int r = rc / CIN, ci = rc - r * CIN;
                    const f16 v = (f16)g.img[rc];
                    const f16 v0 = keep0 ? v : (f16)0.f;
                    *reinterpret_cast<f16*>(b_wr + ((r * 3 + 0) * CIN + ci) * PB) = v0;
                    *reinterpret_cast<f16*>(b_wr + ((r * 3 + 1) * CIN + ci) * PB - 2) = v;
                    *reinterpret_cast<f16*>(b_wr + ((r * 3 + 2) * CIN + ci) * PB - 4) = v;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- fragments: A transposed (two 8-byte transposing reads per 16-channel block), B straight
        sb_v2i ra[NI][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < NI; ++i) ra[i][h] = sb_read_tr16(a_rd0 + h * 4 * PA + i * 32);
        f16x8 fb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f16x8*>(b_rd + j * 16 * PB);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            asm volatile("" : "+v"(ra[i][0]), "+v"(ra[i][1]));
            const sb_v4i t = {ra[i][0][0], ra[i][0][1], ra[i][1][0], ra[i][1][1]};
            const f16x8 fa = __builtin_bit_cast(f16x8, t);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    };

    Seg cur, nxt;
    Pos pc, pn;
    {
        const int row = gw / segs_per_row;
        pc.sg = gw - row * segs_per_row;
        pc.n = row / d.h;
        pc.h = row - pc.n * d.h;
    }
    // Straight-line pipelined loop: the next segment's loads are ALWAYS issued (past the end: the current segment again, unused),
    // so no branch surrounds a load and the compiler's counted waits survive
    if (gw < nseg) {
        load_seg(pc, cur);
        for (int s = gw; s < nseg; s += nw) {
            pn = pc;
            if (s + nw < nseg) advance(pn);
            load_seg(pn, nxt);
            process(pc, cur);
            cur = nxt;
            pc = pn;
        }
    }

    // ---- per-wave sums -> workgroup row (LDS atomics), one partial row per workgroup
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int m = UPP; m < 64; m <<= 1) {
                s1[e][k] += __shfl_xor(s1[e][k], m);
                s2[e][k] += __shfl_xor(s2[e][k], m);
            }
        }
    __syncthreads();          // part[] is zeroed
    // the four waves add their sums to the workgroup row ONE AFTER THE OTHER (round 6: LDS float atomics made the order - and the last
    // bits of dW, dgamma, dbeta - vary from run to run; within a wave every lane owns its own entries)
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
            if (lane < UPP) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        part[Row::S1 + cu * 8 + 2 * e + k] += s1[e][k];
                        part[Row::S2 + cu * 8 + 2 * e + k] += s2[e][k];
                    }
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int m = 16 * i + 4 * g4 + e, k = 16 * j + q16;
                        if (m < 2 * C) part[m * 32 + k] += acc[i][j][e];
                        else if (m == 2 * C) part[Row::SX + k] += acc[i][j][e];
                    }
        }
        __syncthreads();
    }
    float* const out = d.ws + (long)blockIdx.x * Row::SIZE;
    for (int i = tid; i < Row::SIZE; i += 256) out[i] = part[i];
}

// ---- the same pass with the NEXT conv's data gradient fused in (yh_stem_bwd_desc::dz1): dy is never read - and never written by
// anyone.  Next conv = 3x3 / stride 2 / pad 1, K1 = 64 output channels, C = 32.  For the segment's row h and its 32 columns w:
//     dy[h][w][c] = sum over taps (r, s) with (h + 1 - r), (w + 1 - s) even, ho = (h + 1 - r) / 2, wo = (w + 1 - s) / 2 in range
//                   of sum_k dz1[ho][wo][k] W1[k][c][r][s]
// h even: r = 1;  h odd: r = 0 (ho = (h + 1) / 2) and r = 2 (ho = (h - 1) / 2).  The 32 pixels split into the 16 even columns
// w0 + 2 pc (s = 1, wo = w0 / 2 + pc) and the 16 odd ones (s = 2 with the same wo, s = 0 with wo + 1): per dz1 row two B fragments
// per K step (columns wo and wo + 1, 16 bytes of one dz1 pixel per lane - its channels are contiguous) feed 4 + 8 MFMAs whose A
// fragments are W1's data-gradient image (yh_conv_pack_weights_dgrad: rows = c, flipped taps) resident in LDS.  The results arrive in
// MFMA D layout - 4 channels of one pixel per lane - so z is loaded in that layout too (8 bytes per lane), and g / xhat are written
// into the [pixel][channel] A tile as 8-byte cells; from there on the pass is the unfused kernel's (X tile, transposed reads, MFMAs).
template <int CIN, int ACT>
__global__ __launch_bounds__(256, 2) void stem_bwd_dgrad_kernel(const yh_stem_bwd_desc d, const int segs_per_row, const int nseg) {
    constexpr int C = 32, K1S = 2;              // 32 channels, K1 = 64 = two MFMA K steps
    constexpr int AROWS = 2 * C + 16, PA = AROWS * 2, PB = 96, NI = AROWS / 16;
    constexpr int A_BYTES = 32 * PA, B_BYTES = 32 * PB, WAVE_BYTES = A_BYTES + B_BYTES;
    constexpr int NRC = 3 * CIN;
    constexpr int W1_BYTES = 9 * 2 * K1S * 1024;     // [tap][row group][k step][64 lanes][16 B] = 36 KB
    typedef SbRow<C> Row;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * WAVE_BYTES + W1_BYTES];
    // (the workgroup's partial row reuses the waves' tiles after the loop: 69.6 KB in all, two workgroups per CU with room to spare)
    static_assert(Row::SIZE * 4 <= 4 * WAVE_BYTES, "partial row must fit the tiles");
    float* const part = reinterpret_cast<float*>(smem);
    typedef void __attribute__((address_space(3))) * lptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const At = smem + wave * WAVE_BYTES;
    unsigned char* const Bt = At + A_BYTES;
    unsigned char* const W1l = smem + 4 * WAVE_BYTES;
    const unsigned at_lds = (unsigned)(uintptr_t)(lptr_t)At;
    const int pc = lane & 15, kq = lane >> 4;

    for (int i = lane; i < WAVE_BYTES / 16; i += 64) reinterpret_cast<sb_u32x4*>(At)[i] = sb_u32x4{0u, 0u, 0u, 0u};
    __builtin_amdgcn_wave_barrier();
    if (lane < 32) *reinterpret_cast<f16*>(At + lane * PA + 2 * C * 2) = (f16)1.f;
    {   // W1 fragments: tap (r, s) of the ORIGINAL kernel is tap (2 - r, 2 - s) of the flipped image; lane (pc, kq) of fragment
        // (row group i, k step k) holds image[16 i + pc][tap][32 k + 8 kq .. + 7]
        const f16* const wg = reinterpret_cast<const f16*>(d.w1);
        for (int f = wave; f < 9 * 2 * K1S; f += 4) {
            const int t = f / (2 * K1S), ik = f - t * (2 * K1S), i = ik / K1S, k = ik - i * K1S;
            const int r = t / 3, s = t - 3 * r;
            const int ft = (2 - r) * 3 + (2 - s);
            const sb_u32x4 v = *reinterpret_cast<const sb_u32x4*>(wg + ((long)(i * 16 + pc) * 9 + ft) * d.k1_pad + k * 32 + kq * 8);
            *reinterpret_cast<sb_u32x4*>(W1l + (f * 64 + lane) * 16) = v;
        }
    }
    __syncthreads();

    // per-lane constants in D layout: channels 16 i + 4 kq + e (i = 0, 1; e = 0 .. 3), as pairs
    f32x2 isv[4], nmu[4], gis[4], bsh[4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int ch = 16 * i + 4 * kq + 2 * h + k;
                const float is = d.invstd[ch], mu = d.mean[ch];
                isv[2 * i + h][k] = is;
                nmu[2 * i + h][k] = -mu * is;
                gis[2 * i + h][k] = d.gamma[ch];
                bsh[2 * i + h][k] = d.beta[ch];
            }
    f32x2 s1[4], s2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) s1[e] = s2[e] = f32x2{0.f, 0.f};
    f32x4 acc[NI][2];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};

    const f16* const zg = reinterpret_cast<const f16*>(d.z) + 4 * kq;
    const f16* const dzg = reinterpret_cast<const f16*>(d.dz1) + kq * 8;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const int d_row = nw / segs_per_row, d_sg = nw - d_row * segs_per_row;
    struct Pos { int n, h, sg; };
    auto advance = [&](Pos& p) {
        p.sg += d_sg;
        p.h += d_row;
        if (p.sg >= segs_per_row) { p.sg -= segs_per_row; ++p.h; }
        while (p.h >= d.h) { p.h -= d.h; ++p.n; }
    };

    typedef unsigned int sb_u32x2 __attribute__((ext_vector_type(2)));
    struct Seg { sb_u32x4 bz[2][2][K1S]; sb_u32x2 z[2][2]; float img[9]; };      // [dz1 row slot][column wo / wo + 1][k step]; z[i][even / odd]
    const int HW = d.h * d.w_in;
    auto load_seg = [&](const Pos& ps, Seg& g) {
        const int w0 = ps.sg * 32;
        // dz1 rows: slot 0 = ho (h + 1) >> 1 (r = 1 for even h, r = 0 for odd h), slot 1 = ho (h - 1) >> 1 (r = 2, odd h only);
        // clamped addresses, masked values: no branch around a load
        const int ho0 = (ps.h + 1) >> 1, ho1 = (ps.h - 1) >> 1;
        const int wo = (w0 >> 1) + pc;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int ho = sl ? ho1 : ho0;
            const int hc = min(max(ho, 0), d.h1 - 1);
            const bool rok = sl ? (ps.h & 1) : ho0 < d.h1;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const int wc = min(wo + c2, d.w1_in - 1);
                const unsigned msk = (rok && wo + c2 < d.w1_in) ? 0xffffffffu : 0u;
                const f16* const src = dzg + (((long)ps.n * d.h1 + hc) * d.w1_in + wc) * d.lddz1;
#pragma unroll
                for (int k = 0; k < K1S; ++k) {
                    const sb_u32x4 v = *reinterpret_cast<const sb_u32x4*>(src + k * 32);
                    g.bz[sl][c2][k] = sb_u32x4{v[0] & msk, v[1] & msk, v[2] & msk, v[3] & msk};
                }
            }
        }
        const long p0 = ((long)ps.n * d.h + ps.h) * d.w_in + w0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int px = 2 * pc + b;
                const long p = w0 + px < d.w_in ? p0 + px : p0;
                g.z[i][b] = *reinterpret_cast<const sb_u32x2*>(zg + p * d.ldz + 16 * i);
            }
        const int wi = w0 - 1 + lane;
        const int wcl = min(max(wi, 0), d.w_in - 1);
        const bool wok = lane < 34 && wi >= 0 && wi < d.w_in;
        const float* const xin = d.x + (long)ps.n * CIN * HW + wcl;
#pragma unroll
        for (int rc = 0; rc < 9; ++rc) {
            g.img[rc] = 0.f;
            if (rc < NRC) {
                const int r = rc / CIN, ci = rc - r * CIN;
                const int hi = ps.h + r - 1;
                const int hc = min(max(hi, 0), d.h - 1);
                const float v = xin[ci * HW + hc * d.w_in];
                g.img[rc] = v * ((wok && hi >= 0 && hi < d.h) ? 1.f : 0.f);
            }
        }
    };

    const int q16 = pc, g4 = kq;
    const unsigned a_rd0 = at_lds + (8 * g4 + (q16 >> 2)) * PA + 4 * (q16 & 3) * 2;
    const unsigned char* const b_rd = Bt + q16 * PB + 16 + g4 * 16;
    unsigned char* const b_wr = Bt + 16 + lane * 2;
    const unsigned char* const w1_rd = W1l + lane * 16;
    auto w1frag = [&](int r, int s, int i, int k) {
        return *reinterpret_cast<const f16x8*>(w1_rd + ((((r * 3 + s) * 2 + i) * K1S + k) * 64) * 16);
    };
    auto process = [&](const Pos& ps, Seg& g) {
        const int w0 = ps.sg * 32;
        const int valid = d.w_in - w0;
        // ---- dy of the segment: D layout, even columns (dye) and odd columns (dyo), channels 16 i + 4 kq + e
        f32x4 dye[2], dyo[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) dye[i] = dyo[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool hodd = ps.h & 1;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if (sl == 0 || hodd) {                 // wave-uniform; the masked fragments of an absent row are zeros anyway
                const int r = sl ? 2 : (hodd ? 0 : 1);
#pragma unroll
                for (int k = 0; k < K1S; ++k) {
                    const f16x8 b0 = __builtin_bit_cast(f16x8, g.bz[sl][0][k]), b1 = __builtin_bit_cast(f16x8, g.bz[sl][1][k]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        dye[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1frag(r, 1, i, k), b0, dye[i], 0, 0, 0);
                        dyo[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1frag(r, 2, i, k), b0, dyo[i], 0, 0, 0);
                        dyo[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1frag(r, 0, i, k), b1, dyo[i], 0, 0, 0);
                    }
                }
            }
        }
        // ---- g and xhat in D layout -> the A tile (8-byte cells); pixels beyond the row end: g = 0
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int px = 2 * pc + b;
                const bool ok = px < valid;
                const f16x4 zv = __builtin_bit_cast(f16x4, g.z[i][b]);
                const f32x4 dyv = b ? dyo[i] : dye[i];
                f16x4 gv, xv;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = 2 * i + h;
                    const f32x2 zf = {(float)zv[2 * h], (float)zv[2 * h + 1]};
                    // the data gradient the unfused path would have STORED is rounded to fp16: keep that rounding (same values reach
                    // the sums as in the four-launch form)
                    const f32x2 df = {ok ? (float)(f16)dyv[2 * h] : 0.f, ok ? (float)(f16)dyv[2 * h + 1] : 0.f};
                    const f32x2 xh = __builtin_elementwise_fma(zf, isv[e], nmu[e]);
                    const f32x2 u = __builtin_elementwise_fma(gis[e], xh, bsh[e]);
                    const f32x2 m = {sb_dact<ACT>(u[0], d.act, d.slope), sb_dact<ACT>(u[1], d.act, d.slope)};
                    const f32x2 gg = df * m;
                    s1[e] += gg;
                    s2[e] = __builtin_elementwise_fma(gg, xh, s2[e]);
                    gv[2 * h] = (f16)gg[0];
                    gv[2 * h + 1] = (f16)gg[1];
                    xv[2 * h] = (f16)xh[0];
                    xv[2 * h + 1] = (f16)xh[1];
                }
                unsigned char* const cell = At + px * PA + (16 * i + 4 * kq) * 2;
                *reinterpret_cast<f16x4*>(cell) = gv;
                *reinterpret_cast<f16x4*>(cell + C * 2) = xv;
            }
        // ---- X columns (as the unfused kernel)
        const bool keep0 = lane < valid;
        if (lane < 34) {
#pragma unroll
            for (int rc = 0; rc < 9; ++rc) {
                if (rc < NRC) {
                    const int r = rc / CIN, ci = rc - r * CIN;
                    const f16 v = (f16)g.img[rc];
                    const f16 v0 = keep0 ? v : (f16)0.f;
                    *reinterpret_cast<f16*>(b_wr + ((r * 3 + 0) * CIN + ci) * PB) = v0;
                    *reinterpret_cast<f16*>(b_wr + ((r * 3 + 1) * CIN + ci) * PB - 2) = v;
                    *reinterpret_cast<f16*>(b_wr + ((r * 3 + 2) * CIN + ci) * PB - 4) = v;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        sb_v2i ra[NI][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < NI; ++i) ra[i][h] = sb_read_tr16(a_rd0 + h * 4 * PA + i * 32);
        f16x8 fb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f16x8*>(b_rd + j * 16 * PB);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            asm volatile("" : "+v"(ra[i][0]), "+v"(ra[i][1]));
            const sb_v4i t = {ra[i][0][0], ra[i][0][1], ra[i][1][0], ra[i][1][1]};
            const f16x8 fa = __builtin_bit_cast(f16x8, t);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    };

    Seg cur, nxt;
    Pos pcur, pn;
    {
        const int row = gw / segs_per_row;
        pcur.sg = gw - row * segs_per_row;
        pcur.n = row / d.h;
        pcur.h = row - pcur.n * d.h;
    }
    if (gw < nseg) {
        load_seg(pcur, cur);
        for (int s = gw; s < nseg; s += nw) {
            pn = pcur;
            if (s + nw < nseg) advance(pn);
            load_seg(pn, nxt);
            process(pcur, cur);
            cur = nxt;
            pcur = pn;
        }
    }

    // ---- per-wave sums: lanes that share kq (the 16 pixel lanes) add up; lane pc == 0 of each kq holds channels 16 i + 4 kq + e
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                s1[e][k] += __shfl_xor(s1[e][k], m);
                s2[e][k] += __shfl_xor(s2[e][k], m);
            }
        }
    __syncthreads();          // every wave is done with its tiles
    for (int i = tid; i < Row::SIZE; i += 256) part[i] = 0.f;
    __syncthreads();
    for (int turn = 0; turn < 4; ++turn) {      // one wave at a time, in order: deterministic sums (see stem_bwd_partial_kernel)
        if (wave == turn) {
            if (pc == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const int ch = 16 * i + 4 * kq + 2 * h + k;
                            part[Row::S1 + ch] += s1[2 * i + h][k];
                            part[Row::S2 + ch] += s2[2 * i + h][k];
                        }
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int m = 16 * i + 4 * g4 + e, k = 16 * j + q16;
                        if (m < 2 * C) part[m * 32 + k] += acc[i][j][e];
                        else if (m == 2 * C) part[Row::SX + k] += acc[i][j][e];
                    }
        }
        __syncthreads();
    }
    float* const out = d.ws + (long)blockIdx.x * Row::SIZE;
    for (int i = tid; i < Row::SIZE; i += 256) out[i] = part[i];
}

// one workgroup per output channel: column sums of the partial rows in double, then the closed form
template <int C>
__global__ __launch_bounds__(256) void stem_bwd_final_kernel(const yh_stem_bwd_desc d, const int nparts) {
    typedef SbRow<C> Row;
    __shared__ double red[256];
    __shared__ double colsum[98];
    const int c = blockIdx.x, tid = threadIdx.x;
    // 98 columns this channel needs: Q[c][0..31], R[c][0..31], SX[0..31], S1[c], S2[c]; two row halves per column
    const int col = tid % 98, half = tid / 98;
    double v = 0.0;
    if (tid < 196) {
        int off;
        if (col < 32) off = c * 32 + col;
        else if (col < 64) off = (C + c) * 32 + (col - 32);
        else if (col < 96) off = Row::SX + (col - 64);
        else off = (col == 96 ? Row::S1 : Row::S2) + c;
        for (int r = half; r < nparts; r += 2) v += (double)d.ws[(long)r * Row::SIZE + off];
    }
    red[tid] = v;
    __syncthreads();
    if (tid < 98) colsum[tid] = red[tid] + red[tid + 98];
    __syncthreads();
    const double P = (double)d.n * d.h * d.w_in;
    const double S1 = colsum[96], S2 = colsum[97];
    if (tid == 0) {
        d.dbeta[c] += (float)S1;
        d.dgamma[c] += (float)S2;
    }
    const int ncols = 9 * d.cin;
    if (tid < ncols) {
        const int k = tid;                       // column k = (r * 3 + s) * cin + ci
        const int rs = k / d.cin, ci = k - rs * d.cin;
        const double scale = (double)d.gamma[c] * (double)d.invstd[c];
        const double dw = scale * (colsum[k] - S1 / P * colsum[64 + k] - S2 / P * colsum[32 + k]);
        d.dw[((long)c * d.cin + ci) * 9 + rs] += (float)dw;
    }
}

static int sb_grid(const yh_stem_bwd_desc* d, int* segs_per_row, long* nseg) {
    *segs_per_row = (d->w_in + 31) / 32;
    *nseg = (long)d->n * d->h * *segs_per_row;
    long wgs = (*nseg + 3) / 4;
    if (wgs > 512) wgs = 512;                   // two 4-wave workgroups per CU: 8 waves keep ~36 KB of loads in flight per CU
    return (int)(wgs < 1 ? 1 : wgs);
}

static bool sb_supported(const yh_stem_bwd_desc* d) {
    return d && d->n > 0 && d->h > 0 && d->w_in > 0 && (d->cin == 1 || d->cin == 3) && (d->cout == 16 || d->cout == 32);
}

}  // namespace yh

using namespace yh;

extern "C" int64_t yh_stem_bwd_workspace(const yh_stem_bwd_desc* d) {
    if (!sb_supported(d)) return 0;
    int spr;
    long nseg;
    const int wgs = sb_grid(d, &spr, &nseg);
    return (int64_t)wgs * (d->cout == 32 ? SbRow<32>::SIZE : SbRow<16>::SIZE);
}

extern "C" int yh_stem_bwd(const yh_stem_bwd_desc* d, void* stream) {
    if (!sb_supported(d)) return d ? YH_EUNSUPPORTED : YH_EINVAL;
    if (!d->x || (!d->dy && !d->dz1) || !d->z || !d->gamma || !d->beta || !d->mean || !d->invstd || !d->dgamma || !d->dbeta || !d->dw || !d->ws)
        return YH_EINVAL;
    if (!d->dz1 && (d->lddy % 8 || d->ldz % 8 || !aligned16(d->dy) || !aligned16(d->z))) return YH_EALIGN;
    if ((long)d->n * d->h * d->w_in * (long)(d->lddy > d->ldz ? d->lddy : d->ldz) >= (1L << 40)) return YH_EINVAL;
    int spr;
    long nseg;
    const int wgs = sb_grid(d, &spr, &nseg);
    const int64_t need = (int64_t)wgs * (d->cout == 32 ? SbRow<32>::SIZE : SbRow<16>::SIZE);
    if (d->ws_floats < need) return YH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (nseg + 4L * wgs >= 0x7fffffffL) return YH_EUNSUPPORTED;
    if (d->dz1) {      // the next conv's data gradient fused in: 3x3 / s2 / p1, 64 output channels, onto the 32 channels of this block
        if (!d->w1 || d->cout != 32 || d->k1 != 64 || d->k1_pad != 64 || d->h1 != (d->h + 2 - 3) / 2 + 1 || d->w1_in != (d->w_in + 2 - 3) / 2 + 1)
            return YH_EUNSUPPORTED;
        if (d->lddz1 % 8 || !aligned16(d->dz1) || !aligned16(d->w1) || d->ldz % 4 || (((uintptr_t)d->z) & 7u)) return YH_EALIGN;
#define YH_SBD_GO(CI)                                                                                                    \
    do {                                                                                                                \
        if (d->act == YH_ACT_LEAKY) hipLaunchKernelGGL((stem_bwd_dgrad_kernel<CI, YH_ACT_LEAKY>), dim3(wgs), dim3(256), 0, s, *d, spr, (int)nseg); \
        else if (d->act == YH_ACT_MISH) hipLaunchKernelGGL((stem_bwd_dgrad_kernel<CI, YH_ACT_MISH>), dim3(wgs), dim3(256), 0, s, *d, spr, (int)nseg); \
        else hipLaunchKernelGGL((stem_bwd_dgrad_kernel<CI, -1>), dim3(wgs), dim3(256), 0, s, *d, spr, (int)nseg);        \
        hipLaunchKernelGGL(stem_bwd_final_kernel<32>, dim3(32), dim3(256), 0, s, *d, wgs);                              \
    } while (0)
        if (d->cin == 3) YH_SBD_GO(3);
        else YH_SBD_GO(1);
#undef YH_SBD_GO
        return check_launch();
    }
#define YH_SB_GO(CC, CI)                                                                                                 \
    do {                                                                                                                \
        if (d->act == YH_ACT_LEAKY) hipLaunchKernelGGL((stem_bwd_partial_kernel<CC, CI, YH_ACT_LEAKY>), dim3(wgs), dim3(256), 0, s, *d, spr, (int)nseg); \
        else if (d->act == YH_ACT_MISH) hipLaunchKernelGGL((stem_bwd_partial_kernel<CC, CI, YH_ACT_MISH>), dim3(wgs), dim3(256), 0, s, *d, spr, (int)nseg); \
        else hipLaunchKernelGGL((stem_bwd_partial_kernel<CC, CI, -1>), dim3(wgs), dim3(256), 0, s, *d, spr, (int)nseg);  \
        hipLaunchKernelGGL(stem_bwd_final_kernel<CC>, dim3(CC), dim3(256), 0, s, *d, wgs);                              \
    } while (0)
    if (d->cout == 32 && d->cin == 3) YH_SB_GO(32, 3);
    else if (d->cout == 32 && d->cin == 1) YH_SB_GO(32, 1);
    else if (d->cout == 16 && d->cin == 3) YH_SB_GO(16, 3);
    else if (d->cout == 16 && d->cin == 1) YH_SB_GO(16, 1);
    else return YH_EUNSUPPORTED;
#undef YH_SB_GO
    return check_launch();
}
