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

__device__ __forceinline__ unsigned long long sort_key(float score, int key) {
    // scores are finite and > 0 here: their bit patterns order like the values
    const unsigned sb = 0xFFFFFFFFu - __float_as_uint(score);
    return ((unsigned long long)sb << 32) | (unsigned)key;
}

// rank by counting: position of i = number of records that order before it.  grid (ceil(mmax/256), n)
__global__ __launch_bounds__(256) void nms_sort_kernel(const float* __restrict__ cand, const int32_t* __restrict__ count,
                                                       int cap, float* __restrict__ sorted) {
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
    }
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

extern "C" int yh_nms_sort(const float* cand, const int32_t* count, int n, int cap, int mmax, float* sorted, void* stream) {
    if (!cand || !count || !sorted || n <= 0 || cap <= 0 || mmax <= 0 || mmax > cap) return YH_EINVAL;
    if (!aligned16(cand) || !aligned16(sorted)) return YH_EALIGN;
    hipLaunchKernelGGL(nms_sort_kernel, dim3((mmax + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, cand, count, cap, sorted);
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
