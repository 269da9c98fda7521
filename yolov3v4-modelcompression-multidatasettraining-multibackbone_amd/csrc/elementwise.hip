// HBM-bound pieces of the Darknet hot path: weight packing (BN folding), first-layer conv from the
// caller's NCHW fp32 frames, max-pool, channel-slice copy / upsample, shortcut add, YOLO head decode.
// All activations are NHWC with 16-byte vector accesses along the channel axis.
#include "common.h"

#include <type_traits>

namespace yh {

// ---------------------------------------------------------------------------------------- packing
template <typename T>
__global__ void pack_conv_weights_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                         const float* __restrict__ var, float eps, const int32_t* __restrict__ cin_map,
                                         int cout, int cin, int taps, int cin_k, T* __restrict__ packed) {
    const long total = (long)cout * cin * taps;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const long r = i / taps;
        const int ci = (int)(r % cin), co = (int)(r / cin);
        const float scale = gamma ? gamma[co] / sqrtf(var[co] + eps) : 1.f;
        const int pc = cin_map ? cin_map[ci] : ci;
        packed[((long)co * taps + tap) * cin_k + pc] = (T)(w[i] * scale);
    }
}

__global__ void pack_bias_kernel(const float* __restrict__ conv_bias, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, const float* __restrict__ mean,
                                 const float* __restrict__ var, float eps, int cout, float* __restrict__ out) {
    const int co = blockIdx.x * blockDim.x + threadIdx.x;
    if (co >= cout) return;
    float b = conv_bias ? conv_bias[co] : 0.f;
    if (gamma) {
        const float sd = sqrtf(var[co] + eps);
        b = (beta[co] - gamma[co] * mean[co] / sd) + b * (gamma[co] / sd);
    }
    out[co] = b;
}

// stem image: [tap][ci][cout_pad] fp32, tap = r*kw + s
__global__ void pack_stem_weights_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                         const float* __restrict__ var, float eps, int cout, int cin, int taps,
                                         int cout_pad, float* __restrict__ packed) {
    const int total = cout * cin * taps;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int tap = i % taps;
    const int r = i / taps;
    const int ci = r % cin, co = r / cin;
    const float scale = gamma ? gamma[co] / sqrtf(var[co] + eps) : 1.f;
    packed[(tap * cin + ci) * cout_pad + co] = w[i] * scale;
}

// ------------------------------------------------------------------------------------- first layer
// One thread = one output pixel x CO output channels.  Weight and bias addresses are wave-uniform, so they
// come in through scalar loads and the inner loop is 1 vector load + CO FMAs per tap.  (A variant with
// 8 channels per thread and per-lane weight vectors stored better-coalesced 16-byte pieces but measured
// 2.8x slower on MI355X: the per-lane weight loads dominate.)
template <typename T, int CO> struct StemStore;
template <int CO> struct StemStore<f16, CO> {
    static __device__ __forceinline__ void run(f16* dst, const float (&acc)[CO], int co0, int cout) {
#pragma unroll
        for (int g = 0; g < CO / 8; ++g) {
            if (co0 + g * 8 >= cout) break;
            f16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (f16)acc[g * 8 + e];
            *reinterpret_cast<f16x8*>(dst + g * 8) = v;
        }
    }
};
template <int CO> struct StemStore<float, CO> {
    static __device__ __forceinline__ void run(float* dst, const float (&acc)[CO], int co0, int cout) {
#pragma unroll
        for (int g = 0; g < CO / 4; ++g) {
            if (co0 + g * 4 >= cout) break;
            f32x4 v = {acc[g * 4], acc[g * 4 + 1], acc[g * 4 + 2], acc[g * 4 + 3]};
            *reinterpret_cast<f32x4*>(dst + g * 4) = v;
        }
    }
};

template <int CO> struct StemStore<int8_t, CO> {
    static __device__ __forceinline__ void run(int8_t* dst, const float (&acc)[CO], int co0, int cout) {
#pragma unroll
        for (int g = 0; g < CO / 16; ++g) {
            if (co0 + g * 16 >= cout) break;
            unsigned w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned v = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) v |= ((unsigned)(int)acc[g * 16 + q * 4 + e] & 0xffu) << (8 * e);
                w[q] = v;
            }
            *reinterpret_cast<uint4*>(dst + g * 16) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
};

// (Round 2 tried this layer on the matrix cores - 16 pixels x 32 channels per v_mfma_f32_16x16x32_f16 with the im2col fragment
// gathered by 8 scalar loads per lane from the NCHW planes: 1.19 ms against 0.97 ms here.  The gather touches 64 scattered
// addresses per load instruction and is address-path bound; an MFMA stem needs the image staged through LDS first.
// A fully unrolled 3x3x3 form with 432 v_pk_fma_f32 per thread - two output channels per instruction, weights in scalar register
// pairs - compiled as intended and was bit-identical, but ran 4.76 ms: 50 KB of straight-line code per kernel does not live in
// the instruction cache.  The rolled loop below stays.)
template <typename T, int CO, bool K33>
__global__ __launch_bounds__(256) void conv_stem_kernel(const yh_stem_desc d) {
    const long P = (long)d.n * d.ho * d.wo;
    const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int co0 = blockIdx.y * CO;
    if (p >= P) return;
    const int hw = d.ho * d.wo;
    const int n = (int)(p / hw);
    const int rem = (int)(p - (long)n * hw);
    const int ho = rem / d.wo, wo = rem - ho * d.wo;
    const int hi0 = ho * d.stride - d.pad, wi0 = wo * d.stride - d.pad;
    // two output channels per v_pk_fma_f32 (the rolled loop keeps the weights of one tap in scalar registers): the same fused
    // multiply-adds in the same order as a scalar fmaf chain, at twice the VALU rate - this layer is VALU-bound (27 x cout FMAs per
    // pixel against 12 input + 2 cout output bytes)
    f32x2 acc2[CO / 2];
#pragma unroll
    for (int c = 0; c < CO / 2; ++c) acc2[c] = *reinterpret_cast<const f32x2*>(d.bias + co0 + 2 * c);
    const float* xin = d.x + (long)n * d.cin * d.h * d.w_in;
    if constexpr (K33) {
        // 3 x 3 taps of 3 planes (the RGB first layer): the nine samples of one filter row are loaded before their 9 x CO / 2 FMAs, so
        // a wave waits for memory three times per pixel instead of 27; the order of the additions is the rolled loop's.
        // profiles/r03_stem_ab.txt (608 x 608, batch 64): 0.725 ms against 1.00 ms for the tap-by-tap loop (YH_STEM_ROLLED); two
        // pixels per thread with the next tap's weights prefetched into a second scalar register set was bit-identical too but ran
        // 1.00 ms again: 113 VGPRs / 106 SGPRs leave 4 waves per SIMD where this form keeps 7, and occupancy is what hides the loads;
        // 16 instead of 32 output channels per thread (twice the sample loads, more waves): 1.37 ms
#pragma unroll 1
        for (int r = 0; r < 3; ++r) {
            const int hi = hi0 + r;
            const bool rok = (unsigned)hi < (unsigned)d.h;
            const int hc = min(max(hi, 0), d.h - 1);
            f32x4 xa, xb;          // the nine samples in a register vector: the tap loop below stays rolled (one tap's weights in
            float xc;              // scalar registers at a time) and picks its sample by index
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int wi = wi0 + s;
                const bool ok = rok && (unsigned)wi < (unsigned)d.w_in;
                const int wc = min(max(wi, 0), d.w_in - 1);
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float v = xin[((long)ci * d.h + hc) * d.w_in + wc];     // clamped address, value zeroed: no branch per load
                    const float z = ok ? v : 0.f;
                    const int t = s * 3 + ci;
                    if (t < 4) xa[t] = z;
                    else if (t < 8) xb[t - 4] = z;
                    else xc = z;
                }
            }
#pragma unroll 1
            for (int t = 0; t < 9; ++t) {
                const float x1 = t < 4 ? xa[t & 3] : (t < 8 ? xb[t & 3] : xc);
                const f32x2 xx = {x1, x1};
                const f32x2* wrow = reinterpret_cast<const f32x2*>(d.w + (r * 9 + t) * d.cout_pad + co0);
#pragma unroll
                for (int c = 0; c < CO / 2; ++c) acc2[c] = __builtin_elementwise_fma(xx, wrow[c], acc2[c]);
            }
        }
    } else {
        for (int r = 0; r < d.kh; ++r) {
            const int hi = hi0 + r;
            for (int s = 0; s < d.kw; ++s) {
                const int wi = wi0 + s;
                const bool ok = (unsigned)hi < (unsigned)d.h && (unsigned)wi < (unsigned)d.w_in;
                for (int ci = 0; ci < d.cin; ++ci) {
                    const float xv = ok ? xin[((long)ci * d.h + hi) * d.w_in + wi] : 0.f;
                    const f32x2 xx = {xv, xv};
                    const f32x2* wrow = reinterpret_cast<const f32x2*>(d.w + ((r * d.kw + s) * d.cin + ci) * d.cout_pad + co0);
#pragma unroll
                    for (int c = 0; c < CO / 2; ++c) acc2[c] = __builtin_elementwise_fma(xx, wrow[c], acc2[c]);
                }
            }
        }
    }
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = activate(acc2[c / 2][c & 1], d.act, d.slope);
    if constexpr (sizeof(T) == 1) {  // PTQ: quantise onto the activation grid (round half away, clamp)
        const float inv = 1.f / d.out_scale;
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const float t = acc[c] * inv;
            acc[c] = fminf(fmaxf(copysignf(floorf(fabsf(t) + 0.5f), t), -128.f), 127.f);
        }
    }
    T* dst = reinterpret_cast<T*>(d.y) + p * d.ldy + co0;
    StemStore<T, CO>::run(dst, acc, co0, d.cout);
}

// ---------------------------------------------------------------------------------------- max pool
template <typename T> struct Vec16;
template <> struct Vec16<f16> { typedef f16x8 type; static constexpr int N = 8; };
template <> struct Vec16<float> { typedef f32x4 type; static constexpr int N = 4; };

template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const yh_pool_desc d) {
    typedef typename Vec16<T>::type V;
    constexpr int VN = Vec16<T>::N;
    const int cg = d.c / VN;
    const long total = (long)d.n * d.ho * d.wo * cg;
    const T* x = reinterpret_cast<const T*>(d.x);
    T* y = reinterpret_cast<T*>(d.y);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        long r = i / cg;
        const int wo = (int)(r % d.wo);
        r /= d.wo;
        const int ho = (int)(r % d.ho);
        const int n = (int)(r / d.ho);
        float m[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) m[e] = -INFINITY;
        for (int dy = 0; dy < d.k; ++dy) {
            const int hi = ho * d.stride - d.pad_lo + dy;
            for (int dx = 0; dx < d.k; ++dx) {
                const int wi = wo * d.stride - d.pad_lo + dx;
                if ((unsigned)hi < (unsigned)d.h && (unsigned)wi < (unsigned)d.w_in) {
                    const V v = *reinterpret_cast<const V*>(x + (((long)n * d.h + hi) * d.w_in + wi) * d.ldx + g * VN);
#pragma unroll
                    for (int e = 0; e < VN; ++e) m[e] = fmaxf(m[e], (float)v[e]);
                } else if (d.edge_zero && hi >= 0 && wi >= 0) {
#pragma unroll
                    for (int e = 0; e < VN; ++e) m[e] = fmaxf(m[e], 0.f);
                }
            }
        }
        V o;
#pragma unroll
        for (int e = 0; e < VN; ++e) o[e] = (T)m[e];
        *reinterpret_cast<V*>(y + (((long)n * d.ho + ho) * d.wo + wo) * d.ldy + g * VN) = o;
    }
}

// ------------------------------------------------------------------------------ copy / upsample / add
template <typename T>
__global__ __launch_bounds__(256) void copy_channels_kernel(const yh_copy_desc d) {
    typedef typename Vec16<T>::type V;
    constexpr int VN = Vec16<T>::N;
    const int cg = d.c / VN;
    const long total = (long)d.n * d.h * d.w_in * cg;
    const T* x = reinterpret_cast<const T*>(d.x);
    T* y = reinterpret_cast<T*>(d.y);
    const int u = d.ups;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const long pix = i / cg;
        const V v = *reinterpret_cast<const V*>(x + pix * d.ldx + g * VN);
        if (u == 1) {
            *reinterpret_cast<V*>(y + pix * d.ldy + g * VN) = v;
        } else {
            const int wi = (int)(pix % d.w_in);
            const long r = pix / d.w_in;
            const int hi = (int)(r % d.h);
            const long n = r / d.h;
            const long wo_n = (long)d.w_in * u;
            for (int dy = 0; dy < u; ++dy)
                for (int dx = 0; dx < u; ++dx) {
                    const long opix = (n * d.h * u + (long)hi * u + dy) * wo_n + (long)wi * u + dx;
                    *reinterpret_cast<V*>(y + opix * d.ldy + g * VN) = v;
                }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void add_channels_kernel(const yh_add_desc d) {
    typedef typename Vec16<T>::type V;
    constexpr int VN = Vec16<T>::N;
    const int cg = d.c / VN;
    const long total = d.pixels * cg;
    const T* a = reinterpret_cast<const T*>(d.a);
    const T* b = reinterpret_cast<const T*>(d.b);
    T* y = reinterpret_cast<T*>(d.y);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const long pix = i / cg;
        const V va = *reinterpret_cast<const V*>(a + pix * d.lda + g * VN);
        const V vb = *reinterpret_cast<const V*>(b + pix * d.ldb + g * VN);
        V o;
#pragma unroll
        for (int e = 0; e < VN; ++e) o[e] = (T)((float)va[e] + (float)vb[e]);
        *reinterpret_cast<V*>(y + pix * d.ldy + g * VN) = o;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void add_gather_kernel(const yh_add_desc d) {
    const long total = d.pixels * d.c;
    const T* a = reinterpret_cast<const T*>(d.a);
    const T* b = reinterpret_cast<const T*>(d.b);
    T* y = reinterpret_cast<T*>(d.y);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % d.c);
        const long pix = i / d.c;
        const int ka = d.amap[k], kb = d.bmap[k];
        const float va = ka >= 0 ? (float)a[pix * d.lda + ka] : 0.f, vb = kb >= 0 ? (float)b[pix * d.ldb + kb] : 0.f;
        y[pix * d.ldy + k] = (T)(va + vb);
    }
}

// ------------------------------------------------------------------------------------- yolo decode
// grid.x = n * ny: one workgroup per (image, grid row).  Threads sweep the nx cells of the row in memory order - a cell's
// na * no head values are contiguous, so every wave load is one dense run - and write the na output runs of the row (each
// nx * no floats, contiguous in io and raw).  The (cell, anchor, channel) index of a thread advances by 256 elements per trip with
// two carries instead of a division; one exponential per element (of v for w / h, of -v for the sigmoids); 4 independent loads per
// trip, issued one trip ahead; 32-bit offsets.  Byte-bound by design: 4 B read + 4 B written per value, ~25 VALU slots per value.
template <bool RAW>
__global__ __launch_bounds__(256) void yolo_decode_kernel(const yh_decode_desc d) {
    __shared__ float anchor[16];
    if (threadIdx.x < 16) anchor[threadIdx.x] = threadIdx.x < 8 ? d.anchor_w[threadIdx.x & 7] : d.anchor_h[threadIdx.x & 7];
    __syncthreads();
    const int y = blockIdx.x % d.ny, n = blockIdx.x / d.ny;
    const int no = d.no, na = d.na;
    const float* prow = d.p + ((long)n * d.ny + y) * d.nx * d.ldp;
    const int cell_row = d.nx * no;                                 // floats of one anchor's output run of this row
    const int arun = d.ny * cell_row;                               // floats between the runs of consecutive anchors
    float* const io = d.io + ((long)n * d.rows_total + d.row_off) * no + (long)y * cell_row;
    float* const raw = RAW ? d.raw + (long)n * na * arun + (long)y * cell_row : nullptr;
    // element e = (x * na + a) * no + o; this thread starts at e = threadIdx.x and advances by 256
    const int so = 256 % no, st = 256 / no, sa = st % na, sx = st / na;
    int o = threadIdx.x % no, t0 = threadIdx.x / no, a = t0 % na, x = t0 / na;
    const float fy = (float)y;
    auto advance = [&]() {
        o += so;
        const int c = o >= no;
        o -= c ? no : 0;
        a += sa + c;
        const int c2 = a >= na;
        a -= c2 ? na : 0;
        x += sx + c2;
    };
    // branch-free: every lane computes the exponential, the sigmoid and the anchor product and selects
    auto value = [&](float v, int xx, int aa, int oo) {
        const bool wh = oo == 2 || oo == 3;
        const float e = exp_fast(wh ? v : -v);
        const float sg = rcp_fast(1.f + e);
        const float box = (sg + (oo == 0 ? (float)xx : fy)) * d.stride;                  // models.py:410-413: (sigmoid + grid) * stride
        const float size = (e * anchor[(oo == 3 ? 8 : 0) + aa]) * d.stride;             // (exp * anchor) * stride
        return wh ? size : (oo < 2 ? box : sg);
    };
    // Loads and stores share one in-order completion counter: a trip that stores and then loads waits for its stores to be
    // acknowledged before the next values arrive.  Software pipeline: the next trip's U loads are issued BEFORE this trip's stores,
    // and the full trips carry no predicate, so that the compiler can count the wait for them past the stores.
    constexpr int U = 4;
    float v[U];
    int xs[U], as[U], os[U];
    auto load = [&]() {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xs[u] = x; as[u] = a; os[u] = o;
            v[u] = prow[min(x, d.nx - 1) * d.ldp + a * no + o];
            advance();
        }
    };
    const int full_trips = d.nx * na * no / (256 * U);
    load();
    for (int t = 0; t < full_trips; ++t) {
        float cv[U];
        int at[U];
        float out[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cv[u] = v[u];
            at[u] = as[u] * arun + xs[u] * no + os[u];
            out[u] = value(v[u], xs[u], as[u], os[u]);
        }
        load();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (RAW) raw[at[u]] = cv[u];
            io[at[u]] = out[u];
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)          // the ragged last trip
        if (xs[u] < d.nx) {
            const int at = as[u] * arun + xs[u] * no + os[u];
            if constexpr (RAW) raw[at] = v[u];
            io[at] = value(v[u], xs[u], as[u], os[u]);
        }
}

// The same decode for heads whose cell fits one workgroup (na * no <= 256; 255 for the COCO heads): a thread owns ONE (anchor,
// channel) pair for the whole kernel - which of the three formulas applies, the anchor and the output run are per-thread constants -
// and walks over the cells of the grid row, 256 / (na no) cells per pass.  Per value: one add for the source offset, one for the
// destination, the exponential, the reciprocal, two selects.  Loads run one trip (U passes) ahead of the stores, as above.
template <bool RAW>
__global__ __launch_bounds__(256) void yolo_decode_cell_kernel(const yh_decode_desc d) {
    __shared__ float anchor[16];
    if (threadIdx.x < 16) anchor[threadIdx.x] = threadIdx.x < 8 ? d.anchor_w[threadIdx.x & 7] : d.anchor_h[threadIdx.x & 7];
    __syncthreads();
    const int y = blockIdx.x % d.ny, n = blockIdx.x / d.ny;
    const int no = d.no, na = d.na, G = na * no;
    const int cpp = 256 / G;                                        // cells per pass
    const int xs = threadIdx.x / G, r = threadIdx.x - xs * G;
    if (xs >= cpp) return;
    const int a = r / no, o = r - a * no;
    const bool wh = o == 2 || o == 3, box = o < 2;
    const float anc = anchor[(o == 3 ? 8 : 0) + a];
    const float fy = (float)y;
    const float* src = d.p + ((long)n * d.ny + y) * d.nx * d.ldp + r;
    const int cell_row = d.nx * no, arun = d.ny * cell_row;
    float* const io = d.io + ((long)n * d.rows_total + d.row_off) * no + (long)y * cell_row + a * arun + o;
    float* const raw = RAW ? d.raw + (long)n * na * arun + (long)y * cell_row + a * arun + o : nullptr;
    auto value = [&](float v, int x) {
        const float e = exp_fast(wh ? v : -v);
        const float sg = rcp_fast(1.f + e);
        const float b = (sg + (o == 0 ? (float)x : fy)) * d.stride;       // models.py:410-413: (sigmoid + grid) * stride
        const float s = (e * anc) * d.stride;                            // (exp * anchor) * stride
        return wh ? s : (box ? b : sg);
    };
    // What bounds this kernel is bytes in flight: loads alone run at 3.9 TB/s and stores alone at 5 TB/s, but a loop that waits for
    // its 4 loads per trip ran both at 2.7 TB/s combined.  Each thread keeps a ring of D loads in flight: load i + D is issued when
    // value i is taken.  Loads and stores complete through ONE in-order counter and the compiler's counted wait for load i
    // (vmcnt(D - 1): "the D - 1 younger loads may be outstanding") does not know about the stores between them, so in hardware it
    // also waits for the stores older than the youngest D - 1 operations - about D / 2 iterations back, never the recent ones.
    // (Issuing the loads as untracked inline assembly with a hand-written vmcnt(2 D - 1) does not work: the compiler considers the
    // destination register valid - and reusable - the moment the asm statement has executed.)
    constexpr int D = 16;
    float ring[D];
    const int iters = (d.nx + cpp - 1) / cpp;          // uniform over the workgroup; cells beyond nx are loaded clamped and not stored
#pragma unroll
    for (int k = 0; k < D; ++k) ring[k] = src[min(xs + k * cpp, d.nx - 1) * d.ldp];
    for (int base = 0; base < iters; base += D) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const int i = base + k, x = xs + i * cpp;
            const float v = ring[k];
            ring[k] = src[min(x + D * cpp, d.nx - 1) * d.ldp];
            if (x < d.nx) {
                if constexpr (RAW) raw[x * no] = v;
                io[x * no] = value(v, x);
            }
        }
    }
}

static inline unsigned grid_for(long total, int block = 256, long cap = 256L * 16) {
    long g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace yh

using namespace yh;

extern "C" int yh_conv_pack_weights(int dtype, const float* w, const float* conv_bias, const float* bn_gamma,
                                    const float* bn_beta, const float* bn_mean, const float* bn_var, float bn_eps,
                                    const int32_t* cin_map, int cout, int cin, int kh, int kw, int cin_k, int m_pad,
                                    void* packed, float* bias_out, void* stream) {
    if (!w || !packed || !bias_out || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0) return YH_EINVAL;
    if (dtype != YH_F16 && dtype != YH_F32) return YH_EINVAL;
    const bool bn = bn_gamma != nullptr;
    if (bn && (!bn_beta || !bn_mean || !bn_var)) return YH_EINVAL;
    const int bk = dtype == YH_F16 ? 32 : 16;
    if (cin_k % bk || m_pad % 128 || m_pad < cout) return YH_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int taps = kh * kw;
    const size_t esz = dtype == YH_F16 ? 2 : 4;
    hipError_t e = hipMemsetAsync(packed, 0, (size_t)m_pad * taps * cin_k * esz, s);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(bias_out, 0, (size_t)m_pad * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    const long total = (long)cout * cin * taps;
    if (dtype == YH_F16)
        hipLaunchKernelGGL(pack_conv_weights_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, s, w, bn_gamma, bn_var, bn_eps,
                           cin_map, cout, cin, taps, cin_k, (f16*)packed);
    else
        hipLaunchKernelGGL(pack_conv_weights_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, w, bn_gamma, bn_var,
                           bn_eps, cin_map, cout, cin, taps, cin_k, (float*)packed);
    hipLaunchKernelGGL(pack_bias_kernel, dim3((cout + 255) / 256), dim3(256), 0, s, conv_bias, bn_gamma, bn_beta, bn_mean,
                       bn_var, bn_eps, cout, bias_out);
    return check_launch();
}

extern "C" int yh_stem_pack_weights(const float* w, const float* conv_bias, const float* bn_gamma, const float* bn_beta,
                                    const float* bn_mean, const float* bn_var, float bn_eps, int cout, int cin, int kh,
                                    int kw, int cout_pad, float* packed, float* bias_out, void* stream) {
    if (!w || !packed || !bias_out || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0) return YH_EINVAL;
    if (cout_pad % 16 || cout_pad < cout) return YH_EALIGN;
    const bool bn = bn_gamma != nullptr;
    if (bn && (!bn_beta || !bn_mean || !bn_var)) return YH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int taps = kh * kw;
    hipError_t e = hipMemsetAsync(packed, 0, (size_t)taps * cin * cout_pad * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(bias_out, 0, (size_t)cout_pad * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    const int total = cout * cin * taps;
    hipLaunchKernelGGL(pack_stem_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, bn_gamma, bn_var, bn_eps, cout,
                       cin, taps, cout_pad, packed);
    hipLaunchKernelGGL(pack_bias_kernel, dim3((cout + 255) / 256), dim3(256), 0, s, conv_bias, bn_gamma, bn_beta, bn_mean,
                       bn_var, bn_eps, cout, bias_out);
    return check_launch();
}

extern "C" int64_t yh_conv2d_stem_stats_rows(const yh_stem_desc* d) {
    if (!d || getenv("YH_STEM_VALU")) return 0;
    return stem_mfma_stats_rows(*d);
}

extern "C" int yh_conv2d_stem_fwd(const yh_stem_desc* d, void* stream) {
    if (!d || !d->x || !d->w || !d->bias || !d->y) return YH_EINVAL;
    if (d->n <= 0 || d->cin <= 0 || d->cin > 4 || d->h <= 0 || d->w_in <= 0 || d->cout <= 0) return YH_EINVAL;
    if (d->dtype != YH_F16 && d->dtype != YH_F32 && d->dtype != YH_I8) return YH_EINVAL;
    if (d->dtype == YH_I8 && (!(d->out_scale > 0.f) || d->cout % 16 || d->ldy % 16)) return YH_EALIGN;
    if (d->cout_pad % 16 || d->cout_pad < d->cout || d->cout % 8 || d->ldy % 8 || !aligned16(d->y)) return YH_EALIGN;
    if (d->ho != (d->h + 2 * d->pad - d->kh) / d->stride + 1 || d->wo != (d->w_in + 2 * d->pad - d->kw) / d->stride + 1) return YH_EINVAL;
    const long P = (long)d->n * d->ho * d->wo;
    hipStream_t s = (hipStream_t)stream;
    static const bool valu = getenv("YH_STEM_VALU") != nullptr;         // A/B: the packed-FMA kernel below for every shape
    if (!valu) {
        const int rc = launch_stem_mfma(*d, s);                          // conv_stem_mfma.hip: 3 x 3 x 3 taps on the matrix cores
        if (rc != YH_EUNSUPPORTED) return rc;
    }
    if (d->stats_ws) return YH_EUNSUPPORTED;                            // only the MFMA kernel has the statistics epilogue
    const bool wide = d->cout_pad % 32 == 0;
    const dim3 grid((unsigned)((P + 255) / 256), (unsigned)(d->cout_pad / (wide ? 32 : 16)));
    static const bool rolled = getenv("YH_STEM_ROLLED") != nullptr;      // A/B: the tap-by-tap loop for every shape
    const bool k33 = d->kh == 3 && d->kw == 3 && d->cin == 3 && !rolled;
#define YH_STEM_LAUNCH(T, CO)                                                                              \
    do {                                                                                                    \
        if (k33) hipLaunchKernelGGL((conv_stem_kernel<T, CO, true>), grid, dim3(256), 0, s, *d);            \
        else hipLaunchKernelGGL((conv_stem_kernel<T, CO, false>), grid, dim3(256), 0, s, *d);               \
    } while (0)
    if (d->dtype == YH_F16) {
        if (wide) YH_STEM_LAUNCH(f16, 32);
        else YH_STEM_LAUNCH(f16, 16);
    } else if (d->dtype == YH_I8) {
        if (wide) YH_STEM_LAUNCH(int8_t, 32);
        else YH_STEM_LAUNCH(int8_t, 16);
    } else {
        if (wide) YH_STEM_LAUNCH(float, 32);
        else YH_STEM_LAUNCH(float, 16);
    }
#undef YH_STEM_LAUNCH
    return check_launch();
}

static int vec_of(int dtype) { return dtype == YH_F16 ? 8 : 4; }

extern "C" int yh_maxpool2d_fwd(const yh_pool_desc* d, void* stream) {
    if (!d || !d->x || !d->y || d->n <= 0 || d->c <= 0 || d->k <= 0 || d->stride <= 0) return YH_EINVAL;
    if (d->dtype != YH_F16 && d->dtype != YH_F32) return YH_EINVAL;
    const int v = vec_of(d->dtype);
    if (d->c % v || d->ldx % v || d->ldy % v || !aligned16(d->x) || !aligned16(d->y)) return YH_EALIGN;
    const long total = (long)d->n * d->ho * d->wo * (d->c / v);
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == YH_F16) hipLaunchKernelGGL(maxpool_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, s, *d);
    else hipLaunchKernelGGL(maxpool_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, *d);
    return check_launch();
}

extern "C" int yh_copy_channels(const yh_copy_desc* d, void* stream) {
    if (!d || !d->x || !d->y || d->n <= 0 || d->c <= 0 || (d->ups != 1 && d->ups != 2)) return YH_EINVAL;
    if (d->dtype != YH_F16 && d->dtype != YH_F32) return YH_EINVAL;
    const int v = vec_of(d->dtype);
    if (d->c % v || d->ldx % v || d->ldy % v || !aligned16(d->x) || !aligned16(d->y)) return YH_EALIGN;
    const long total = (long)d->n * d->h * d->w_in * (d->c / v);
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == YH_F16) hipLaunchKernelGGL(copy_channels_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, s, *d);
    else hipLaunchKernelGGL(copy_channels_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, *d);
    return check_launch();
}

extern "C" int yh_add_channels(const yh_add_desc* d, void* stream) {
    if (!d || !d->a || !d->b || !d->y || d->pixels <= 0 || d->c <= 0) return YH_EINVAL;
    if (d->dtype != YH_F16 && d->dtype != YH_F32) return YH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (d->amap || d->bmap) {
        if (!d->amap || !d->bmap) return YH_EINVAL;
        const long n = d->pixels * d->c;
        if (d->dtype == YH_F16) hipLaunchKernelGGL(add_gather_kernel<f16>, dim3(grid_for(n)), dim3(256), 0, s, *d);
        else hipLaunchKernelGGL(add_gather_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, *d);
        return check_launch();
    }
    const int v = vec_of(d->dtype);
    if (d->c % v || d->lda % v || d->ldb % v || d->ldy % v || !aligned16(d->a) || !aligned16(d->b) || !aligned16(d->y)) return YH_EALIGN;
    const long total = d->pixels * (d->c / v);
    if (d->dtype == YH_F16) hipLaunchKernelGGL(add_channels_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, s, *d);
    else hipLaunchKernelGGL(add_channels_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, *d);
    return check_launch();
}

extern "C" int yh_yolo_decode(const yh_decode_desc* d, void* stream) {
    if (!d || !d->p || !d->io || d->n <= 0 || d->ny <= 0 || d->nx <= 0 || d->na <= 0 || d->na > 8 || d->no < 5) return YH_EINVAL;
    if (d->ldp < d->na * d->no || d->row_off < 0 || d->row_off + d->na * d->ny * d->nx > d->rows_total) return YH_EINVAL;
    const long rows = (long)d->n * d->ny;
    if (rows > 0x7fffffffL || (long)d->nx * d->ldp > 0x3fffffffL || (long)d->na * d->ny * d->nx * d->no > 0x7fffffffL) return YH_EINVAL;
    static const bool generic = getenv("YH_DECODE_GENERIC") != nullptr;      // A/B: the element-order kernel for every head
    if (d->na * d->no <= 256 && !generic) {
        if (d->raw) hipLaunchKernelGGL(yolo_decode_cell_kernel<true>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, *d);
        else hipLaunchKernelGGL(yolo_decode_cell_kernel<false>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, *d);
        return check_launch();
    }
    if (d->raw) hipLaunchKernelGGL(yolo_decode_kernel<true>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, *d);
    else hipLaunchKernelGGL(yolo_decode_kernel<false>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, *d);
    return check_launch();
}
