// Batched non-maximum suppression for decoded YOLO predictions (reference: utils/utils.py:782-860).
//
// Candidate records are 8 floats (32 bytes): x1, y1, x2, y2, score, cls, key (int bits), 0.
// `key` = row * nc + cls is the position the reference would have emitted the candidate at; ordering is
// score descending, ties by ascending key (= a stable sort of the reference's emission order), so the
// result does not depend on the order the atomic cursor handed out slots.
//
// All IoU arithmetic is plain fp32 in the reference's operation order (no fma contraction: this file is
// compiled with -ffp-contract=off) on boxes offset by cls * 4096, exactly as utils.py:840-841 builds
// them, so suppression decisions agree with a CPU evaluation of the same formula.
#include "common.h"

namespace yh {

constexpr float kMinWH = 2.f, kMaxWH = 4096.f;
constexpr int REC = 8;

__device__ __forceinline__ bool finite6(float a, float b, float c, float d, float e) {
    return isfinite(a) && isfinite(b) && isfinite(c) && isfinite(d) && isfinite(e);
}

// Slot allocation aggregated per wave: the lanes that emit a candidate for the same image in this step share ONE atomicAdd (the
// leader's; a wave spans at most two images) - at test.py's settings (conf 0.001, multi-label) an image emits 10^4 candidates and
// every one of them used to be its own same-address atomic (3.4 ms for 16 images of 21 000).
__device__ __forceinline__ void emit(float* cand, int32_t* count, int img, int cap, bool want, float x1, float y1, float x2,
                                     float y2, float score, int cls, int key) {
    unsigned long long todo = __ballot(want);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int limg = __shfl(img, leader);
        const unsigned long long same = __ballot(want && img == limg);
        int base = 0;
        if (lane == leader) base = atomicAdd(count + limg, __popcll(same));
        base = __shfl(base, leader);
        if (want && img == limg) {
            const int slot = base + __popcll(same & ((1ull << lane) - 1ull));
            if (cand != nullptr && slot < cap) {
                float* r = cand + ((long)img * cap + slot) * REC;
                f32x4 lo = {x1, y1, x2, y2};
                f32x4 hi = {score, (float)cls, __int_as_float(key), 0.f};
                *reinterpret_cast<f32x4*>(r) = lo;
                *reinterpret_cast<f32x4*>(r + 4) = hi;
            }
        }
        todo &= ~same;
    }
}

// multi-label (test.py: every class score above the threshold is a candidate): one thread per (row, class), classes fastest, so a
// wave reads consecutive class scores of ONE row (coalesced) and the row's box / objectness as broadcasts; rows below the
// objectness threshold (99 % of them) cost one load.  Every lane of a wave runs emit() once (the ballots need the whole wave).
__global__ __launch_bounds__(256) void nms_candidates_ml_kernel(const float* __restrict__ pred, int n, int rows, int nc, float conf,
                                                                const uint8_t* __restrict__ class_mask, float* cand, int32_t* count,
                                                                int cap) {
    const long total = (long)n * rows * nc;
    const int no = nc + 5;
    const long span = (long)gridDim.x * blockDim.x;
    const long iters = (total + span - 1) / span;
    for (long it = 0; it < iters; ++it) {
        const long i = it * span + blockIdx.x * (long)blockDim.x + threadIdx.x;
        bool want = i < total;
        const long rw = (want ? i : 0) / nc;
        const int c = (int)((want ? i : 0) - rw * nc);
        const float* x = pred + rw * no;
        const float obj = x[4];
        want = want && obj > conf;
        float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, sc = 0.f;
        if (want) {
            const float w = x[2], h = x[3];
            want = w > kMinWH && w < kMaxWH && h > kMinWH && h < kMaxWH;
            const float cx = x[0], cy = x[1];
            x1 = cx - w / 2; y1 = cy - h / 2; x2 = cx + w / 2; y2 = cy + h / 2;
            sc = x[5 + c] * obj;
            want = want && sc > conf && (!class_mask || class_mask[c]) && finite6(x1, y1, x2, y2, sc);
        }
        const int img = (int)(rw / rows), row = (int)(rw - (long)img * rows);
        emit(cand, count, img, cap, want, x1, y1, x2, y2, sc, c, row * nc + c);
    }
}

// best class only (detect.py): one thread per prediction row
__global__ __launch_bounds__(256) void nms_candidates_kernel(const float* __restrict__ pred, int n, int rows, int nc,
                                                             float conf, const uint8_t* __restrict__ class_mask, float* cand,
                                                             int32_t* count, int cap) {
    const long total = (long)n * rows;
    const int no = nc + 5;
    const long span = (long)gridDim.x * blockDim.x;
    const long iters = (total + span - 1) / span;
    for (long it = 0; it < iters; ++it) {
        const long i = it * span + blockIdx.x * (long)blockDim.x + threadIdx.x;
        bool live = i < total;
        const float* x = pred + (live ? i : 0) * no;
        const float obj = x[4];
        live = live && obj > conf;
        const float w = x[2], h = x[3];
        live = live && (w > kMinWH && w < kMaxWH && h > kMinWH && h < kMaxWH);
        const int img = (int)((live ? i : 0) / rows), row = (int)((live ? i : 0) - (long)img * rows);
        const float cx = x[0], cy = x[1];
        const float x1 = cx - w / 2, y1 = cy - h / 2, x2 = cx + w / 2, y2 = cy + h / 2;
        float best = 0.f;
        int bc = 0;
        if (live) {
            best = x[5] * obj;
            for (int c = 1; c < nc; ++c) {
                const float s = x[5 + c] * obj;
                if (s > best) { best = s; bc = c; }
            }
        }
        const bool want = live && (!class_mask || class_mask[bc]) && finite6(x1, y1, x2, y2, best);
        emit(cand, count, img, cap, want, x1, y1, x2, y2, best, bc, row * nc + bc);
    }
}

// ---- decode + candidates in one pass over the head convolutions' outputs (round 5) -------------------
// YOLOLayer.forward's eval branch (models.py:406-418) followed by the candidate filter above (utils.py:799-827), without the (n, rows,
// 5 + nc) tensor in between: at 608 x 608, batch 64 the decode writes 495 MB of which the filter keeps ~100 rows per image.  One thread
// per prediction row reads the row's objectness logit; only rows above the threshold (0.4 % at detect.py's settings) are decoded at all.
// Every decoded value is the decode kernels' (csrc/elementwise.hip: rcp_fast(1 + exp_fast(-v)), (sigmoid + cell) * stride, (exp_fast(v) *
// anchor) * stride), every candidate test is the kernels' above on those values, the key is the row's position in the concatenated tensor:
// the records equal the two-pass path's as a set, and the sort makes the rest identical.
__device__ __forceinline__ float dec_sigmoid(float v) { return rcp_fast(1.f + exp_fast(-v)); }

template <bool ML>
__global__ __launch_bounds__(256) void yolo_decode_candidates_kernel(const yh_decode_desc d, const float conf,
                                                                     const uint8_t* __restrict__ class_mask, float* cand,
                                                                     int32_t* count, const int cap) {
    __shared__ float anchor[16];
    if (threadIdx.x < 16) anchor[threadIdx.x] = threadIdx.x < 8 ? d.anchor_w[threadIdx.x & 7] : d.anchor_h[threadIdx.x & 7];
    __syncthreads();
    const int nc = d.no - 5, cells = d.ny * d.nx;
    const long per_img = (long)d.na * cells, total = per_img * d.n;
    const long span = (long)gridDim.x * blockDim.x;
    const long iters = (total + span - 1) / span;
    for (long it = 0; it < iters; ++it) {
        const long i = it * span + blockIdx.x * (long)blockDim.x + threadIdx.x;
        bool live = i < total;
        const long ii = live ? i : 0;
        const int img = (int)(ii / per_img);
        const int r = (int)(ii - (long)img * per_img);      // row inside this head: (a ny + y) nx + x (models.py:416 view order)
        const int a = r / cells, yx = r - a * cells, y = yx / d.nx, x = yx - y * d.nx;
        const float* src = d.p + (((long)img * d.ny + y) * d.nx + x) * d.ldp + a * d.no;
        const float obj = dec_sigmoid(src[4]);
        live = live && obj > conf;
        if (__ballot(live) == 0ull) continue;                // wave-uniform: emit() needs every lane of a wave
        float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
        if (live) {
            const float cx = (dec_sigmoid(src[0]) + (float)x) * d.stride, cy = (dec_sigmoid(src[1]) + (float)y) * d.stride;
            const float w = (exp_fast(src[2]) * anchor[a]) * d.stride, h = (exp_fast(src[3]) * anchor[8 + a]) * d.stride;
            live = w > kMinWH && w < kMaxWH && h > kMinWH && h < kMaxWH;
            x1 = cx - w / 2; y1 = cy - h / 2; x2 = cx + w / 2; y2 = cy + h / 2;
        }
        const int row = d.row_off + r;
        if constexpr (ML) {
            for (int c = 0; c < nc; ++c) {
                bool want = live;
                float sc = 0.f;
                if (live) {
                    sc = dec_sigmoid(src[5 + c]) * obj;
                    want = sc > conf && (!class_mask || class_mask[c]) && finite6(x1, y1, x2, y2, sc);
                }
                emit(cand, count, img, cap, want, x1, y1, x2, y2, sc, c, row * nc + c);
            }
        } else {
            float best = 0.f;
            int bc = 0;
            if (live) {
                best = dec_sigmoid(src[5]) * obj;
                for (int c = 1; c < nc; ++c) {
                    const float sv = dec_sigmoid(src[5 + c]) * obj;
                    if (sv > best) { best = sv; bc = c; }
                }
            }
            const bool want = live && (!class_mask || class_mask[bc]) && finite6(x1, y1, x2, y2, best);
            emit(cand, count, img, cap, want, x1, y1, x2, y2, best, bc, row * nc + bc);
        }
    }
}

__device__ __forceinline__ unsigned long long sort_key(float score, int key) {
    // scores are finite and > 0 here: their bit patterns order like the values
    const unsigned sb = 0xFFFFFFFFu - __float_as_uint(score);
    return ((unsigned long long)sb << 32) | (unsigned)key;
}

// rank by counting: position of i = number of records that order before it.  grid (ceil(mmax/256), n)
__global__ __launch_bounds__(256) void nms_sort_kernel(const float* __restrict__ cand, const int32_t* __restrict__ count,
                                                       int cap, float* __restrict__ sorted, uint8_t* __restrict__ cls8) {
    __shared__ unsigned long long keys[256];
    const int img = blockIdx.y;
    int m = count[img];
    if (m > cap) m = cap;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= m) return;
    const float* base = cand + (long)img * cap * REC;
    unsigned long long mine = 0;
    f32x4 lo, hi;
    if (i < m) {
        lo = *reinterpret_cast<const f32x4*>(base + (long)i * REC);
        hi = *reinterpret_cast<const f32x4*>(base + (long)i * REC + 4);
        mine = sort_key(hi[0], __float_as_int(hi[2]));
    }
    int rank = 0;
    for (int j0 = 0; j0 < m; j0 += 256) {
        const int j = j0 + threadIdx.x;
        __syncthreads();
        if (j < m) {
            const f32x4 h = *reinterpret_cast<const f32x4*>(base + (long)j * REC + 4);
            keys[threadIdx.x] = sort_key(h[0], __float_as_int(h[2]));
        }
        __syncthreads();
        const int lim = min(256, m - j0);
        for (int k = 0; k < lim; ++k) rank += keys[k] < mine ? 1 : 0;
    }
    if (i < m) {
        float* dst = sorted + ((long)img * cap + rank) * REC;
        *reinterpret_cast<f32x4*>(dst) = lo;
        *reinterpret_cast<f32x4*>(dst + 4) = hi;
        if (cls8) cls8[(long)img * cap + rank] = (uint8_t)(int)hi[1];      // the class of every sorted position, one byte each (nms_class_kernel)
    }
}

// ---- tile sort + rank by binary search (round 5) --------------------------------------------------------------------------------------
// The counting sort above compares every pair: 4.4 x 10^8 comparisons per image at 21 000 candidates (0.68 ms for 16 images, the largest
// stage left once the suppression runs class by class).  Here tiles of SORT_TILE records are sorted in LDS (bitonic network on the same
// 64-bit keys, the record's slot as payload), and a record's global rank is its rank in its own tile plus, for every other tile, the
// number of keys below it - a binary search, keys being unique.  Same total order, so the sorted list is the counting sort's.
constexpr int SORT_TILE = 2048;

// grid (ceil(mmax / tile), n), 256 threads; tile = min(cap, SORT_TILE), a power of two >= 256
__global__ __launch_bounds__(256) void nms_tile_sort_kernel(const float* __restrict__ cand, const int32_t* __restrict__ count, int cap,
                                                            int tile, unsigned long long* __restrict__ tkey, int32_t* __restrict__ tslot) {
    __shared__ unsigned long long k[SORT_TILE];
    __shared__ int v[SORT_TILE];
    const int img = blockIdx.y, t0 = blockIdx.x * tile;
    int m = count[img];
    if (m > cap) m = cap;
    if (t0 >= m) return;
    const float* base = cand + (long)img * cap * REC;
    for (int e = threadIdx.x; e < tile; e += 256) {
        const int slot = t0 + e;
        unsigned long long key = ~0ull;      // padding sorts behind every record
        if (slot < m) {
            const f32x4 h = *reinterpret_cast<const f32x4*>(base + (long)slot * REC + 4);
            key = sort_key(h[0], __float_as_int(h[2]));
        }
        k[e] = key;
        v[e] = slot;
    }
    __syncthreads();
    for (int kk = 2; kk <= tile; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < (tile >> 1); i += 256) {
                const int l = 2 * i - (i & (j - 1)), p = l + j;
                const unsigned long long a = k[l], b = k[p];
                if ((a > b) == ((l & kk) == 0)) {
                    k[l] = b; k[p] = a;
                    const int va = v[l];
                    v[l] = v[p]; v[p] = va;
                }
            }
            __syncthreads();
        }
    }
    unsigned long long* dk = tkey + (long)img * cap + t0;
    int32_t* dv = tslot + (long)img * cap + t0;
    for (int e = threadIdx.x; e < tile; e += 256) {
        dk[e] = k[e];
        dv[e] = v[e];
    }
}

// grid (ceil(mmax / 256), n), 256 threads: one thread per record of the tile-sorted list
__global__ __launch_bounds__(256) void nms_tile_rank_kernel(const float* __restrict__ cand, const int32_t* __restrict__ count, int cap,
                                                            int tile, const unsigned long long* __restrict__ tkey,
                                                            const int32_t* __restrict__ tslot, float* __restrict__ sorted,
                                                            uint8_t* __restrict__ cls8) {
    const int img = blockIdx.y;
    int m = count[img];
    if (m > cap) m = cap;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= ((m + tile - 1) / tile) * tile) return;
    const int mine_t = e / tile, r = e - mine_t * tile;
    if (r >= min(tile, m - mine_t * tile)) return;      // padding
    const unsigned long long* tk = tkey + (long)img * cap;
    const unsigned long long mine = tk[e];
    const int ntiles = (m + tile - 1) / tile;
    int rank = r;
    for (int g0 = 0; g0 < ntiles; g0 += 8) {
        int lo[8], hi[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = g0 + u;
            lo[u] = 0;
            hi[u] = (t < ntiles && t != mine_t) ? min(tile, m - t * tile) : 0;
        }
        for (int step = 0; step < 12; ++step) {      // 2^11 = SORT_TILE: the interval is empty after 12 halvings; 8 independent loads per round
            unsigned long long q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int mid = (lo[u] + hi[u]) >> 1;
                q[u] = lo[u] < hi[u] ? tk[(long)(g0 + u) * tile + mid] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int mid = (lo[u] + hi[u]) >> 1;
                if (lo[u] < hi[u]) {
                    if (q[u] < mine) lo[u] = mid + 1; else hi[u] = mid;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) rank += lo[u];
    }
    const float* src = cand + ((long)img * cap + tslot[(long)img * cap + e]) * REC;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
    float* dst = sorted + ((long)img * cap + rank) * REC;
    *reinterpret_cast<f32x4*>(dst) = a;
    *reinterpret_cast<f32x4*>(dst + 4) = b;
    if (cls8) cls8[(long)img * cap + rank] = (uint8_t)(int)b[1];
}

__device__ __forceinline__ float iou_off(const f32x4& a, const f32x4& b) {
    const float iw = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.f);
    const float ih = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.f);
    const float inter = iw * ih;
    const float aa = (a[2] - a[0]) * (a[3] - a[1]);
    const float ab = (b[2] - b[0]) * (b[3] - b[1]);
    return inter / (aa + ab - inter);
}

__device__ __forceinline__ f32x4 offset_box(const float* rec, int agnostic) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(rec);
    const float off = agnostic ? 0.f : rec[5] * kMaxWH;
    return f32x4{b[0] + off, b[1] + off, b[2] + off, b[3] + off};
}

// grid (words, words, n), 64 threads: thread t owns row block*64+t, tests it against 64 column boxes
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ sorted, const int32_t* __restrict__ count,
                                                      int cap, int mmax, float thr, int agnostic,
                                                      unsigned long long* __restrict__ mask) {
    __shared__ f32x4 cols[64];
    const int img = blockIdx.z;
    int m = count[img];
    if (m > cap) m = cap;
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (rb * 64 >= m || cb * 64 >= m || cb < rb) return;
    const int words = (mmax + 63) / 64;
    const float* base = sorted + (long)img * cap * REC;
    const int jc = cb * 64 + threadIdx.x;
    if (jc < m) cols[threadIdx.x] = offset_box(base + (long)jc * REC, agnostic);
    __syncthreads();
    const int i = rb * 64 + threadIdx.x;
    if (i >= m) return;
    const f32x4 me = offset_box(base + (long)i * REC, agnostic);
    const int lim = min(64, m - cb * 64);
    unsigned long long bits = 0;
    for (int k = (rb == cb ? threadIdx.x + 1 : 0); k < lim; ++k) {
        // boxes of different classes are offset by 4096 per class: almost every pair has no overlap extent at all.  Rejecting those
        // before the divide changes nothing (inter = 0 gives IoU 0 or NaN, neither > thr) and is most of the pairs at 10^4 candidates
        const f32x4 o = cols[k];
        if (!(fminf(me[2], o[2]) > fmaxf(me[0], o[0]) && fminf(me[3], o[3]) > fmaxf(me[1], o[1]))) continue;
        if (iou_off(me, o) > thr) bits |= 1ull << k;
    }
    mask[((long)img * mmax + i) * words + cb] = bits;
}

// one 256-thread block per image: greedy scan in score order, 64 boxes per step
__global__ __launch_bounds__(256) void nms_reduce_kernel(const unsigned long long* __restrict__ mask,
                                                         const int32_t* __restrict__ count, int cap, int mmax,
                                                         int32_t* __restrict__ keep_idx, int32_t* __restrict__ n_keep) {
    // one dynamic LDS region (no static __shared__ in front of it, so 8-byte cells stay aligned):
    // [words] suppressed bits | [64] diagonal words of the current step | keep bits | kept count
    extern __shared__ __attribute__((aligned(16))) unsigned long long remv[];
    const int img = blockIdx.x;
    int m = count[img];
    if (m > cap) m = cap;
    const int words = (mmax + 63) / 64;
    const int mw = (m + 63) / 64;
    unsigned long long* diag = remv + words;
    unsigned long long& keep_bits = diag[64];
    unsigned long long& kept_total = diag[65];
    const unsigned long long* mk = mask + (long)img * mmax * words;
    for (int w = threadIdx.x; w < mw; w += blockDim.x) remv[w] = 0;
    if (threadIdx.x == 0) kept_total = 0;
    __syncthreads();
    // (A/B, profiles/r04_nms_stages.txt: holding the diagonal words in wave 0's registers and resolving a step through lane reads
    // was SLOWER - 3.2 vs 2.8 ms at 21 000 candidates: the variable-lane read became a ds_bpermute per box.)
    for (int blk = 0; blk < mw; ++blk) {
        const int lim = min(64, m - blk * 64);
        if ((int)threadIdx.x < lim) diag[threadIdx.x] = mk[((long)blk * 64 + threadIdx.x) * words + blk];
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long dead = remv[blk], kb = 0;
            int kt = (int)kept_total;
            for (int b = 0; b < lim; ++b) {
                if (!((dead >> b) & 1ull)) {
                    kb |= 1ull << b;
                    dead |= diag[b];
                    keep_idx[(long)img * cap + kt++] = blk * 64 + b;
                }
            }
            keep_bits = kb;
            kept_total = kt;
        }
        __syncthreads();
        const unsigned long long kb = keep_bits;
        // the suppression rows of this step's kept boxes, OR-ed into the words still to come.  Eight rows per trip, their loads
        // independent of each other (kb is uniform: no divergence) - the one-row-at-a-time form paid a memory latency per kept box
        // (5.6 ms for 16 images of 21 000 candidates, the slowest stage of the evaluation-settings NMS)
        for (int w = blk + 1 + threadIdx.x; w < mw; w += blockDim.x) {
            unsigned long long acc = remv[w];
            const unsigned long long* col = mk + (long)blk * 64 * words + w;
#pragma unroll 1
            for (int b0 = 0; b0 < 64; b0 += 8) {
                const unsigned g8 = (unsigned)(kb >> b0) & 0xffu;
                if (!g8) continue;
                unsigned long long v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = ((g8 >> u) & 1u) ? col[(long)(b0 + u) * words] : 0ull;
                acc |= (v[0] | v[1]) | (v[2] | v[3]) | ((v[4] | v[5]) | (v[6] | v[7]));
            }
            remv[w] = acc;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) n_keep[img] = (int)kept_total;
}

// ---- class-segmented suppression (round 5) -------------------------------------------------------------------------------------------
// The reference offsets every box by cls * 4096 (utils.py:840) so that ONE torchvision.nms call never suppresses across classes.  At
// test.py's settings (conf 0.001, multi-label) an image has 10^4 candidates in 80 classes: the IoU bit mask of the general path is 80 x
// larger than the pairs that can interact, and its greedy scan is ONE serial chain of m / 64 steps with two dependent global-memory
// latencies each (profiles/r04_nms_stages.txt: 1.3 + 2.8 ms of 4.9 for 16 images of 21 000).  Here one workgroup owns one (image, class):
// it collects the class's positions of the score-sorted list (in order), keeps their offset boxes in LDS and runs the same greedy scan
// on them - 64 boxes per step, the step's 64 x 64 IoU bits resolved by wave 0 in scalar registers, the kept boxes of the step tested
// against the later ones straight from LDS.  No mask in memory; 80 x n scans run concurrently.
//
// When is that EXACTLY the reference's result?  Boxes of classes c < c' are offset by >= 4096 against each other in x AND y.  With
// xmin / xmax the extreme raw coordinates of an image's candidates and xmax - xmin <= 4096 (or the same in y), rounding being monotonic,
// fl(a.x2 + 4096 c) <= fl(b.x1 + 4096 c'): the overlap extent is <= 0, the intersection 0, IoU 0 (or NaN) - never > thr, in the scan
// and in the merge weights alike.  nms_compact_kernel checks it per image (extremes collected here through atomics) and reports images
// that fail it, or whose class holds more than SEG_MAX candidates, in state[img][5]: the host then runs the general path on the batch.
// The IoU arithmetic is iou_off() on offset_box() values: the same fp32 operations in the same order as the general path's.
constexpr int SEG_MAX = 2048;

__device__ __forceinline__ unsigned ord_enc(float f) {      // unsigned order == float order (finite values)
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_dec(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

__device__ __forceinline__ bool seg_overlap(const f32x4& a, const f32x4& b) {
    return fminf(a[2], b[2]) > fmaxf(a[0], b[0]) && fminf(a[3], b[3]) > fmaxf(a[1], b[1]);
}

// grid (nc, n), 256 threads.  state[img][8] (host-initialised: [0] = 0, [1], [2] = 0xFFFFFFFF, [3], [4] = 0):
// [0] |= 1 when a class overflows SEG_MAX; [1..4] = ordered encodings of min x1, min y1, max x2, max y2
__global__ __launch_bounds__(256) void nms_class_kernel(const float* __restrict__ sorted, const uint8_t* __restrict__ cls8,
                                                        const int32_t* __restrict__ count, int cap, float thr,
                                                        uint8_t* __restrict__ keep8, unsigned* __restrict__ state) {
    __shared__ f32x4 box[SEG_MAX];
    __shared__ int pos[SEG_MAX];
    __shared__ uint8_t dead[SEG_MAX];
    __shared__ unsigned long long part[4][64];
    __shared__ unsigned long long kb_sh;
    __shared__ int wcnt[4];
    const int img = blockIdx.y, c = blockIdx.x;
    int m = count[img];
    if (m > cap) m = cap;
    if (m <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint8_t* cl = cls8 + (long)img * cap;
    // ---- the class's positions, in score order: wave w scans a quarter of the list (count, then write)
    const int quarter = (((m + 3) >> 2) + 63) & ~63;
    const int lo = wave * quarter, hi = min(m, lo + quarter);
    int cnt = 0;
    for (int p0 = lo; p0 < hi; p0 += 512) {
        unsigned v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + u * 64 + lane;
            v[u] = p < hi ? cl[p] : 0xFFu;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) cnt += __popcll(__ballot(v[u] == (unsigned)c));
    }
    if (lane == 0) wcnt[wave] = cnt;
    __syncthreads();
    const int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    if (total == 0) return;
    if (total > SEG_MAX) {
        if (tid == 0) atomicOr(state + (long)img * 8, 1u);
        return;
    }
    int run = 0;
    for (int w = 0; w < wave; ++w) run += wcnt[w];
    for (int p0 = lo; p0 < hi; p0 += 512) {
        unsigned v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + u * 64 + lane;
            v[u] = p < hi ? cl[p] : 0xFFu;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool hit = v[u] == (unsigned)c;
            const unsigned long long b = __ballot(hit);
            if (hit) pos[run + __popcll(b & below)] = p0 + u * 64 + lane;
            run += __popcll(b);
        }
    }
    __syncthreads();
    // ---- offset boxes into LDS; extremes of the raw coordinates
    const float* base = sorted + (long)img * cap * REC;
    float x1 = INFINITY, y1 = INFINITY, x2 = -INFINITY, y2 = -INFINITY;
    for (int r = tid; r < total; r += 256) {
        const float* rec = base + (long)pos[r] * REC;
        const f32x4 b = *reinterpret_cast<const f32x4*>(rec);
        box[r] = offset_box(rec, 0);
        dead[r] = 0;
        x1 = fminf(x1, b[0]); y1 = fminf(y1, b[1]); x2 = fmaxf(x2, b[2]); y2 = fmaxf(y2, b[3]);
    }
    for (int d = 32; d > 0; d >>= 1) {
        x1 = fminf(x1, __shfl_xor(x1, d)); y1 = fminf(y1, __shfl_xor(y1, d));
        x2 = fmaxf(x2, __shfl_xor(x2, d)); y2 = fmaxf(y2, __shfl_xor(y2, d));
    }
    if (lane == 0 && wave * 64 < total) {      // a wave without a box holds +-inf: nothing to report
        unsigned* st = state + (long)img * 8;
        atomicMin(st + 1, ord_enc(x1)); atomicMin(st + 2, ord_enc(y1));
        atomicMax(st + 3, ord_enc(x2)); atomicMax(st + 4, ord_enc(y2));
    }
    __syncthreads();
    // ---- greedy scan, 64 boxes per step
    uint8_t* kp = keep8 + (long)img * cap;
    const int nblk = (total + 63) >> 6;
    for (int blk = 0; blk < nblk; ++blk) {
        const int b0 = blk * 64, lim = min(64, total - b0);
        {   // the step's own 64 x 64 bits: thread = (row lane, column quarter wave)
            unsigned long long bits = 0;
            if (lane < lim) {
                const f32x4 me = box[b0 + lane];
                const int k1 = min(lim, wave * 16 + 16);
                for (int k = max(wave * 16, lane + 1); k < k1; ++k) {
                    const f32x4 o = box[b0 + k];
                    if (seg_overlap(me, o) && iou_off(me, o) > thr) bits |= 1ull << k;
                }
            }
            part[wave][lane] = bits;
        }
        __syncthreads();
        if (wave == 0) {
            const unsigned long long d = part[0][lane] | part[1][lane] | part[2][lane] | part[3][lane];
            const int dlo = (int)(unsigned)d, dhi = (int)(unsigned)(d >> 32);
            unsigned long long dm = __ballot(lane >= lim || dead[b0 + min(lane, lim - 1)] != 0);
            unsigned long long kb = 0;
#pragma unroll
            for (int b = 0; b < 64; ++b) {      // uniform: scalar registers, lane reads with a constant lane
                if (!((dm >> b) & 1ull)) {
                    kb |= 1ull << b;
                    dm |= ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(dhi, b) << 32) |
                          (unsigned)__builtin_amdgcn_readlane(dlo, b);
                }
            }
            if (lane < lim) kp[pos[b0 + lane]] = (uint8_t)((kb >> lane) & 1ull);
            if (lane == 0) kb_sh = kb;
        }
        __syncthreads();
        const unsigned long long kb = kb_sh;
        if (kb != 0 && blk + 1 < nblk) {
            for (int j = b0 + 64 + tid; j < total; j += 256) {
                if (dead[j]) continue;
                const f32x4 me = box[j];
                unsigned long long todo = kb;
                while (todo) {
                    const int b = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const f32x4 o = box[b0 + b];
                    if (seg_overlap(o, me) && iou_off(o, me) > thr) {
                        dead[j] = 1;
                        break;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// grid (n), 1024 threads: the verdict of the segmented path per image and, where it stands, the kept positions in score order
__global__ __launch_bounds__(1024) void nms_compact_kernel(const uint8_t* __restrict__ keep8, const int32_t* __restrict__ count,
                                                           int cap, unsigned* __restrict__ state, int32_t* __restrict__ keep_idx,
                                                           int32_t* __restrict__ n_keep) {
    __shared__ int wsum[16];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int m = count[img];
    if (m > cap) m = cap;
    unsigned* st = state + (long)img * 8;
    if (m <= 0) {
        if (tid == 0) { n_keep[img] = 0; st[5] = 0; }
        return;
    }
    const double ex = (double)ord_dec(st[3]) - (double)ord_dec(st[1]), ey = (double)ord_dec(st[4]) - (double)ord_dec(st[2]);
    const bool general = st[0] != 0 || !(ex <= (double)kMaxWH || ey <= (double)kMaxWH);
    if (general) {
        if (tid == 0) { n_keep[img] = 0; st[5] = 1; }
        return;
    }
    const uint8_t* kp = keep8 + (long)img * cap;
    const int per = (m + 1023) >> 10;
    const int lo = min(m, tid * per), hi = min(m, lo + per);
    int mine = 0;
    for (int p = lo; p < hi; ++p) mine += kp[p];
    int incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = incl - mine;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    int32_t* dst = keep_idx + (long)img * cap;
    for (int p = lo; p < hi; ++p)
        if (kp[p]) dst[off++] = p;
    if (tid == 1023) { n_keep[img] = off; st[5] = 0; }
}

// one wave per kept box: weighted mean of every box it overlaps (IoU > thr), weights = scores
__global__ __launch_bounds__(256) void nms_merge_kernel(const float* __restrict__ sorted, const int32_t* __restrict__ count,
                                                        const int32_t* __restrict__ keep_idx,
                                                        const int32_t* __restrict__ n_keep, int cap, float thr, int agnostic,
                                                        int merge_lo, int merge_hi, float* __restrict__ out) {
    const int img = blockIdx.y;
    int m = count[img];
    if (m > cap) m = cap;
    const int nk = n_keep[img];
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= nk) return;
    const float* base = sorted + (long)img * cap * REC;
    const int i = keep_idx[(long)img * cap + k];
    const float* me = base + (long)i * REC;
    float o0 = me[0], o1 = me[1], o2 = me[2], o3 = me[3];
    if (m > merge_lo && m < merge_hi) {
        const f32x4 mb = offset_box(me, agnostic);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, sw = 0.f;
        for (int j = lane; j < m; j += 64) {
            const float* r = base + (long)j * REC;
            const f32x4 ob = offset_box(r, agnostic);
            if (iou_off(mb, ob) > thr) {
                const float w = r[4];
                s0 += w * r[0]; s1 += w * r[1]; s2 += w * r[2]; s3 += w * r[3]; sw += w;
            }
        }
        for (int d = 32; d > 0; d >>= 1) {
            s0 += __shfl_xor(s0, d); s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d);
            s3 += __shfl_xor(s3, d); sw += __shfl_xor(sw, d);
        }
        o0 = s0 / sw; o1 = s1 / sw; o2 = s2 / sw; o3 = s3 / sw;
    }
    if (lane == 0) {
        float* dst = out + ((long)img * cap + k) * 6;
        dst[0] = o0; dst[1] = o1; dst[2] = o2; dst[3] = o3; dst[4] = me[4]; dst[5] = me[5];
    }
}

}  // namespace yh

using namespace yh;

extern "C" int yh_nms_candidates(const float* pred, int n, int rows, int nc, float conf_thres, int multi_label,
                                 const uint8_t* class_mask, float* cand, int32_t* count, int cap, void* stream) {
    if (!pred || !count || n <= 0 || rows <= 0 || nc <= 0 || cap < 0) return YH_EINVAL;
    if (cand && !aligned16(cand)) return YH_EALIGN;
    const long total = (long)n * rows * (multi_label ? nc : 1);
    long g = (total + 255) / 256;
    if (g > 65536) g = 65536;
    if (multi_label)
        hipLaunchKernelGGL(nms_candidates_ml_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, pred, n, rows, nc, conf_thres,
                           class_mask, cand, count, cap);
    else
        hipLaunchKernelGGL(nms_candidates_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, pred, n, rows, nc, conf_thres,
                           class_mask, cand, count, cap);
    return check_launch();
}

extern "C" int yh_yolo_decode_candidates(const yh_decode_desc* d, float conf_thres, int multi_label, const uint8_t* class_mask,
                                         float* cand, int32_t* count, int cap, void* stream) {
    if (!d || !d->p || !count || d->n <= 0 || d->ny <= 0 || d->nx <= 0 || d->na <= 0 || d->na > 8 || d->no <= 5 || cap < 0) return YH_EINVAL;
    if (cand && !aligned16(cand)) return YH_EALIGN;
    if ((long)d->rows_total * (d->no - 5) > 0x7fffffffL) return YH_EUNSUPPORTED;      // the key is a 32-bit row * nc + cls
    const long total = (long)d->n * d->na * d->ny * d->nx;
    long g = (total + 255) / 256;
    if (g > 65536) g = 65536;
    if (multi_label)
        hipLaunchKernelGGL(yolo_decode_candidates_kernel<true>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, *d, conf_thres,
                           class_mask, cand, count, cap);
    else
        hipLaunchKernelGGL(yolo_decode_candidates_kernel<false>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, *d, conf_thres,
                           class_mask, cand, count, cap);
    return check_launch();
}

extern "C" int yh_nms_sort(const float* cand, const int32_t* count, int n, int cap, int mmax, float* sorted, void* stream) {
    if (!cand || !count || !sorted || n <= 0 || cap <= 0 || mmax <= 0 || mmax > cap) return YH_EINVAL;
    if (!aligned16(cand) || !aligned16(sorted)) return YH_EALIGN;
    hipLaunchKernelGGL(nms_sort_kernel, dim3((mmax + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, cand, count, cap, sorted,
                       (uint8_t*)nullptr);
    return check_launch();
}

extern "C" int yh_nms_sort_cls(const float* cand, const int32_t* count, int n, int cap, int mmax, float* sorted, uint8_t* cls8,
                               void* stream) {
    if (!cand || !count || !sorted || !cls8 || n <= 0 || cap <= 0 || mmax <= 0 || mmax > cap) return YH_EINVAL;
    if (!aligned16(cand) || !aligned16(sorted)) return YH_EALIGN;
    hipLaunchKernelGGL(nms_sort_kernel, dim3((mmax + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, cand, count, cap, sorted, cls8);
    return check_launch();
}

extern "C" int yh_nms_sort_tiles(const float* cand, const int32_t* count, int n, int cap, int mmax, float* sorted, uint8_t* cls8,
                                 void* ws, size_t ws_bytes, void* stream) {
    if (!cand || !count || !sorted || !ws || n <= 0 || cap <= 0 || mmax <= 0 || mmax > cap) return YH_EINVAL;
    if (!aligned16(cand) || !aligned16(sorted) || !aligned16(ws)) return YH_EALIGN;
    if ((cap & (cap - 1)) != 0 || cap < 256) return YH_EUNSUPPORTED;      // tiles are powers of two that divide the bound
    if (ws_bytes < (size_t)n * cap * 12) return YH_EINVAL;
    const int tile = cap < SORT_TILE ? cap : SORT_TILE;
    unsigned long long* tkey = (unsigned long long*)ws;
    int32_t* tslot = (int32_t*)(tkey + (size_t)n * cap);
    hipLaunchKernelGGL(nms_tile_sort_kernel, dim3((mmax + tile - 1) / tile, n), dim3(256), 0, (hipStream_t)stream, cand, count, cap, tile,
                       tkey, tslot);
    const int padded = ((mmax + tile - 1) / tile) * tile;
    hipLaunchKernelGGL(nms_tile_rank_kernel, dim3((padded + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, cand, count, cap, tile, tkey,
                       tslot, sorted, cls8);
    return check_launch();
}

extern "C" int yh_nms_class_scan(const float* sorted, const uint8_t* cls8, const int32_t* count, int n, int cap, int nc,
                                 float iou_thres, uint8_t* keep8, uint32_t* state, int32_t* keep_idx, int32_t* n_keep, void* stream) {
    if (!sorted || !cls8 || !count || !keep8 || !state || !keep_idx || !n_keep || n <= 0 || cap <= 0 || nc <= 0) return YH_EINVAL;
    if (!aligned16(sorted)) return YH_EALIGN;
    if (nc > 255 || n > 65535) return YH_EUNSUPPORTED;      // one byte per class id, 0xFF = no candidate
    hipLaunchKernelGGL(nms_class_kernel, dim3(nc, n), dim3(256), 0, (hipStream_t)stream, sorted, cls8, count, cap, iou_thres, keep8,
                       (unsigned*)state);
    hipLaunchKernelGGL(nms_compact_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, keep8, count, cap, (unsigned*)state, keep_idx,
                       n_keep);
    return check_launch();
}

extern "C" int yh_nms_mask(const float* sorted, const int32_t* count, int n, int cap, int mmax, float iou_thres, int agnostic,
                           uint64_t* mask, void* stream) {
    if (!sorted || !count || !mask || n <= 0 || cap <= 0 || mmax <= 0 || mmax > cap) return YH_EINVAL;
    if (!aligned16(sorted)) return YH_EALIGN;
    const int words = (mmax + 63) / 64;
    if (words > 65535 || n > 65535) return YH_EUNSUPPORTED;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words, n), dim3(64), 0, (hipStream_t)stream, sorted, count, cap, mmax, iou_thres,
                       agnostic, (unsigned long long*)mask);
    return check_launch();
}

extern "C" int yh_nms_reduce(const uint64_t* mask, const int32_t* count, int n, int cap, int mmax, int32_t* keep_idx,
                             int32_t* n_keep, void* stream) {
    if (!mask || !count || !keep_idx || !n_keep || n <= 0 || cap <= 0 || mmax <= 0 || mmax > cap) return YH_EINVAL;
    const int words = (mmax + 63) / 64;
    const size_t lds = (size_t)(words + 66) * sizeof(unsigned long long);
    if (lds > 60 * 1024) return YH_EUNSUPPORTED;  // > ~480k candidates per image
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(n), dim3(256), lds, (hipStream_t)stream, (const unsigned long long*)mask, count, cap,
                       mmax, keep_idx, n_keep);
    return check_launch();
}

extern "C" int yh_nms_merge(const float* sorted, const int32_t* count, const int32_t* keep_idx, const int32_t* n_keep, int n,
                            int cap, int kmax, float iou_thres, int agnostic, int merge_lo, int merge_hi, float* out,
                            void* stream) {
    if (!sorted || !count || !keep_idx || !n_keep || !out || n <= 0 || cap <= 0 || kmax <= 0) return YH_EINVAL;
    if (!aligned16(sorted)) return YH_EALIGN;
    hipLaunchKernelGGL(nms_merge_kernel, dim3((kmax + 3) / 4, n), dim3(256), 0, (hipStream_t)stream, sorted, count, keep_idx, n_keep,
                       cap, iou_thres, agnostic, merge_lo, merge_hi, out);
    return check_launch();
}
