// compute_loss of the reference (utils/utils.py:368-432) and its hand-derived gradient, fused: a handful of launches per
// head instead of ~200 small torch kernels and several full-size zero-filled temporaries in autograd.
#include "common.h"

namespace yh {

struct Dual4 {  // value + partial derivatives w.r.t. the four raw box logits
    float v, d[4];
};
__device__ __forceinline__ Dual4 mk(float v) { return Dual4{v, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ Dual4 operator+(const Dual4& a, const Dual4& b) {
    return Dual4{a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2], a.d[3] + b.d[3]}};
}
__device__ __forceinline__ Dual4 operator-(const Dual4& a, const Dual4& b) {
    return Dual4{a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2], a.d[3] - b.d[3]}};
}
__device__ __forceinline__ Dual4 operator*(const Dual4& a, const Dual4& b) {
    Dual4 r;
    r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
__device__ __forceinline__ Dual4 operator/(const Dual4& a, const Dual4& b) {
    Dual4 r;
    const float inv = 1.f / b.v;
    r.v = a.v * inv;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
__device__ __forceinline__ Dual4 dmin(const Dual4& a, const Dual4& b) { return a.v <= b.v ? a : b; }
__device__ __forceinline__ Dual4 dmax(const Dual4& a, const Dual4& b) { return a.v >= b.v ? a : b; }
__device__ __forceinline__ Dual4 clamp0(const Dual4& a) { return a.v > 0.f ? a : mk(0.f); }

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }
// BCEWithLogits with pos_weight: -(pw t log s + (1 - t) log(1 - s)), stable form
__device__ __forceinline__ float bce(float x, float t, float pw) {
    const float sp = fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x)));   // softplus(-x) = -log sigmoid(x)
    return (1.f - t) * x + (1.f + (pw - 1.f) * t) * sp;
}
__device__ __forceinline__ float bce_grad(float x, float t, float pw) {
    const float s = sigmoidf(x);
    return s * (1.f - t + pw * t) - pw * t;
}

// GIoU of the predicted box (from 4 raw logits, xywh) against the target (xywh), bbox_iou(x1y1x2y2=False, GIoU=True)
__device__ __forceinline__ Dual4 giou_of(const float* ps4, const float* an, const float* tb) {
    Dual4 px = mk(sigmoidf(ps4[0])), py = mk(sigmoidf(ps4[1]));
    px.d[0] = px.v * (1.f - px.v);
    py.d[1] = py.v * (1.f - py.v);
    const float ew = expf(ps4[2]), eh = expf(ps4[3]);
    Dual4 pw = mk(fminf(ew, 1e3f) * an[0]), ph = mk(fminf(eh, 1e3f) * an[1]);
    pw.d[2] = ew < 1e3f ? pw.v : 0.f;                    // clamp(max=1e3) cuts the gradient
    ph.d[3] = eh < 1e3f ? ph.v : 0.f;
    const Dual4 half = mk(0.5f);
    const Dual4 ax1 = px - pw * half, ax2 = px + pw * half, ay1 = py - ph * half, ay2 = py + ph * half;
    const Dual4 bx1 = mk(tb[0] - tb[2] / 2), bx2 = mk(tb[0] + tb[2] / 2), by1 = mk(tb[1] - tb[3] / 2), by2 = mk(tb[1] + tb[3] / 2);
    const Dual4 inter = clamp0(dmin(ax2, bx2) - dmax(ax1, bx1)) * clamp0(dmin(ay2, by2) - dmax(ay1, by1));
    const Dual4 w1 = ax2 - ax1, h1 = ay2 - ay1, w2 = bx2 - bx1, h2 = by2 - by1;
    const Dual4 uni = (w1 * h1 + mk(1e-16f)) + w2 * h2 - inter;
    const Dual4 iou = inter / uni;
    const Dual4 cw = dmax(ax2, bx2) - dmin(ax1, bx1), ch = dmax(ay2, by2) - dmin(ay1, by1);
    const Dual4 hull = cw * ch + mk(1e-16f);
    return iou - (hull - uni) / hull;
}

__device__ __forceinline__ void block_sum_atomic(float v, float* dst) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dst, red[0] + red[1] + red[2] + red[3]);
    __syncthreads();
}

// build_targets (utils/utils.py:725-779) for one candidate k = a * nt + t, in the reference's fp32 arithmetic
struct Match {
    int b, a, gy, gx, cls;
    long cell;
    float tb[4], an[2];
};
// 0: matched, -1: wh_iou <= iou_t, > 0: label error mask (class / image / cell out of range)
__device__ __forceinline__ int assign(const yh_loss_desc& d, int k, Match& m) {
    m.a = k / d.nt;
    const float* tg = d.targets + 6L * (k - m.a * d.nt);
    const float gx = tg[2] * (float)d.nx, gy = tg[3] * (float)d.ny, gw = tg[4] * (float)d.nx, gh = tg[5] * (float)d.ny;
    m.an[0] = d.anchors[2 * m.a];
    m.an[1] = d.anchors[2 * m.a + 1];
    const float inter = fminf(m.an[0], gw) * fminf(m.an[1], gh);      // wh_iou (utils.py:325-331)
    if (!(inter / (m.an[0] * m.an[1] + gw * gh - inter) > d.iou_t)) return -1;
    m.b = (int)tg[0];
    m.cls = (int)tg[1];
    m.gx = (int)gx;
    m.gy = (int)gy;
    int err = 0;
    if (m.cls < 0 || m.cls >= d.nc) err |= 1;
    if (m.b < 0 || m.b >= d.bs || m.gx < 0 || m.gx >= d.nx || m.gy < 0 || m.gy >= d.ny) err |= 2;
    if (err) return err;
    m.tb[0] = gx - floorf(gx);
    m.tb[1] = gy - floorf(gy);
    m.tb[2] = gw;
    m.tb[3] = gh;
    m.cell = (((long)m.b * d.na + m.a) * d.ny + m.gy) * d.nx + m.gx;
    return 0;
}

// count the matched candidates; several may share a cell: remember the last one per cell
__global__ __launch_bounds__(256) void loss_assign_kernel(const yh_loss_desc d) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    Match m;
    const int r = k < d.na * d.nt ? assign(d, k, m) : -1;
    if (r == 0) atomicMax(d.winner + m.cell, k);
    if (r > 0) atomicOr(d.count + 1, r);
    const unsigned long long hit = __ballot(r == 0);
    if ((threadIdx.x & 63) == 0 && hit) atomicAdd(d.count, __popcll(hit));
}

// matched candidates, forward: box and class sums, objectness targets.  One thread per (candidate, class) - the class loop of a
// candidate used to run inside one thread (80 dependent global loads + BCEs: 33 - 83 us for 1536 threads, latency bound); the thread
// of class 0 also takes the box term.  nc == 1: one thread per candidate, no class term (as the reference, utils.py:412).
__global__ __launch_bounds__(256) void loss_matched_fwd_kernel(const yh_loss_desc d) {
    const int ncl = d.nc > 1 ? d.nc : 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = idx / ncl, c = idx - k * ncl;
    float lbox = 0.f, lcls = 0.f;
    Match m;
    if (k < d.na * d.nt && assign(d, k, m) == 0) {
        const float* ps = d.p + m.b * d.sb + m.a * d.sa + m.gy * d.sy + m.gx * d.sx;
        if (c == 0) {
            const float box[4] = {ps[0], ps[1], ps[2], ps[3]};
            const Dual4 g = giou_of(box, m.an, m.tb);
            lbox = 1.f - g.v;
            if (d.winner[m.cell] == k) d.tobj[m.cell] = (1.f - d.gr) + d.gr * fmaxf(g.v, 0.f);
        }
        if (d.nc > 1) lcls = bce(ps[5 + c], c == m.cls ? d.cp : d.cn, d.cls_pw);
    }
    block_sum_atomic(lbox, d.sums + 0);
    if (d.nc > 1) block_sum_atomic(lcls, d.sums + 2);
}

// every cell, forward: objectness BCE against tobj (32-bit cell decode: the launcher keeps the cell count below 2^31)
__global__ __launch_bounds__(256) void loss_obj_fwd_kernel(const yh_loss_desc d) {
    const unsigned cells = (unsigned)d.bs * d.na * d.ny * d.nx;
    float acc = 0.f;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += gridDim.x * blockDim.x) {
        const unsigned x = i % (unsigned)d.nx;
        unsigned r = i / (unsigned)d.nx;
        const unsigned y = r % (unsigned)d.ny;
        r /= (unsigned)d.ny;
        const unsigned a = r % (unsigned)d.na, b = r / (unsigned)d.na;
        acc += bce(d.p[(long)b * d.sb + (long)a * d.sa + (long)y * d.sy + (long)x * d.sx + 4], d.tobj[i], d.obj_pw);
    }
    block_sum_atomic(acc, d.sums + 1);
}

// every cell, backward: the whole gradient row of the cell (zeros, objectness term in slot 4).  One workgroup row = the na * no
// values of one grid position (contiguous in the NHWC head tensors: consecutive threads store consecutive floats), the position
// decoded once per row with 32-bit arithmetic.  (The first form decoded every ELEMENT with five 64-bit divisions: 0.41 ms for the
// 76 x 76 head of YOLOv3-608 batch 64 - 1 TB/s - against 0.1 ms of bytes.)
__global__ __launch_bounds__(256) void loss_dense_bwd_kernel(const yh_loss_desc d) {
    const int row = d.na * d.no;                              // any width: a thread takes elements j, j + 256, .. of the row
    const int pixels = d.bs * d.ny * d.nx, hw = d.ny * d.nx;
    const float sc = *d.scale * d.g_obj / (float)((long)d.bs * d.na * d.ny * d.nx);
    for (int j = threadIdx.x; j < row; j += 256) {            // one trip for na * no <= 256 (COCO: 255), more for nc > 80
        const int a = j / d.no, o = j - a * d.no;
        for (int p = blockIdx.x; p < pixels; p += gridDim.x) {
            const int b = p / hw, r = p - b * hw;
            const int y = r / d.nx, x = r - y * d.nx;
            float g = 0.f;
            if (o == 4) {
                const long cell = (((long)b * d.na + a) * d.ny + y) * d.nx + x;
                g = sc * bce_grad(d.p[(long)b * d.sb + (long)a * d.sa + (long)y * d.sy + (long)x * d.sx + 4], d.tobj[cell], d.obj_pw);
            }
            d.grad[(long)b * d.gb + (long)a * d.ga + (long)y * d.gy + (long)x * d.gx + o] = g;
        }
    }
}

// matched candidates, backward: box and class terms added on top of the dense pass (duplicates of a cell accumulate); one thread per
// (candidate, class) as in the forward
__global__ __launch_bounds__(256) void loss_matched_bwd_kernel(const yh_loss_desc d) {
    const int ncl = d.nc > 1 ? d.nc : 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = idx / ncl, c = idx - k * ncl;
    Match m;
    if (k >= d.na * d.nt || assign(d, k, m) != 0) return;
    const float* ps = d.p + m.b * d.sb + m.a * d.sa + m.gy * d.sy + m.gx * d.sx;
    float* gp = d.grad + m.b * d.gb + m.a * d.ga + m.gy * d.gy + m.gx * d.gx;
    const int nb = d.count[0] > 1 ? d.count[0] : 1;
    if (c == 0) {
        const float w_box = *d.scale * d.g_box / (float)nb;
        const float box[4] = {ps[0], ps[1], ps[2], ps[3]};
        const Dual4 g = giou_of(box, m.an, m.tb);
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(gp + i, -w_box * g.d[i]);
    }
    if (d.nc > 1) {
        const float w_cls = *d.scale * d.g_cls / (float)((long)nb * d.nc);
        atomicAdd(gp + 5 + c, w_cls * bce_grad(ps[5 + c], c == m.cls ? d.cp : d.cn, d.cls_pw));
    }
}

static int check_loss(const yh_loss_desc* d, bool bwd) {
    if (!d || !d->p || !d->tobj || d->bs <= 0 || d->na <= 0 || d->ny <= 0 || d->nx <= 0 || d->no < 5 || d->nc != d->no - 5) return YH_EINVAL;
    if (d->nt < 0 || !d->count || (d->nt > 0 && (!d->targets || !d->anchors || (!bwd && !d->winner)))) return YH_EINVAL;
    if (bwd ? (!d->grad || !d->scale) : !d->sums) return YH_EINVAL;
    return YH_OK;
}

static inline unsigned grid_n(long total) {
    long g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace yh

using namespace yh;

extern "C" int yh_yolo_loss_fwd(const yh_loss_desc* d, void* stream) {
    int rc = check_loss(d, false);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (d->nt > 0) {
        const unsigned blocks = (unsigned)(((long)d->na * d->nt + 255) / 256);
        const unsigned cblocks = (unsigned)(((long)d->na * d->nt * (d->nc > 1 ? d->nc : 1) + 255) / 256);
        hipLaunchKernelGGL(loss_assign_kernel, dim3(blocks), dim3(256), 0, s, *d);
        hipLaunchKernelGGL(loss_matched_fwd_kernel, dim3(cblocks), dim3(256), 0, s, *d);
    }
    if ((long)d->bs * d->na * d->ny * d->nx >= 0x7fffffffL) return YH_EUNSUPPORTED;
    hipLaunchKernelGGL(loss_obj_fwd_kernel, dim3(grid_n((long)d->bs * d->na * d->ny * d->nx)), dim3(256), 0, s, *d);
    return check_launch();
}

extern "C" int yh_yolo_loss_bwd(const yh_loss_desc* d, void* stream) {
    int rc = check_loss(d, true);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if ((long)d->bs * d->ny * d->nx >= 0x7fffffffL) return YH_EUNSUPPORTED;
    const long pixels = (long)d->bs * d->ny * d->nx;
    hipLaunchKernelGGL(loss_dense_bwd_kernel, dim3((unsigned)(pixels < 16384 ? pixels : 16384)), dim3(256), 0, s, *d);
    if (d->nt > 0)
        hipLaunchKernelGGL(loss_matched_bwd_kernel, dim3((unsigned)(((long)d->na * d->nt * (d->nc > 1 ? d->nc : 1) + 255) / 256)), dim3(256), 0,
                           s, *d);
    return check_launch();
}
