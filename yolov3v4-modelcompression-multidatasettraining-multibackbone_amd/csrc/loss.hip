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

// several targets may match the same (image, anchor, cell): remember the last one per cell
__global__ __launch_bounds__(256) void loss_winner_kernel(const yh_loss_desc d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.nb) return;
    const int b = d.idx[4 * i], a = d.idx[4 * i + 1], gy = d.idx[4 * i + 2], gx = d.idx[4 * i + 3];
    atomicMax(d.winner + (((long)b * d.na + a) * d.ny + gy) * d.nx + gx, i);
}

// matched targets, forward: box and class sums, objectness targets
__global__ __launch_bounds__(256) void loss_matched_fwd_kernel(const yh_loss_desc d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float lbox = 0.f, lcls = 0.f;
    if (i < d.nb) {
        const int b = d.idx[4 * i], a = d.idx[4 * i + 1], gy = d.idx[4 * i + 2], gx = d.idx[4 * i + 3];
        const float* ps = d.p + b * d.sb + a * d.sa + gy * d.sy + gx * d.sx;
        const float box[4] = {ps[0], ps[1], ps[2], ps[3]};
        const Dual4 g = giou_of(box, d.anchor + 2 * i, d.tbox + 4 * i);
        lbox = 1.f - g.v;
        const long cell = (((long)b * d.na + a) * d.ny + gy) * d.nx + gx;
        if (d.winner[cell] == i) d.tobj[cell] = (1.f - d.gr) + d.gr * fmaxf(g.v, 0.f);
        if (d.nc > 1) {
            const int tc = d.tcls[i];
            for (int c = 0; c < d.nc; ++c) lcls += bce(ps[5 + c], c == tc ? d.cp : d.cn, d.cls_pw);
        }
    }
    block_sum_atomic(lbox, d.sums + 0);
    if (d.nc > 1) block_sum_atomic(lcls, d.sums + 2);
}

// every cell, forward: objectness BCE against tobj
__global__ __launch_bounds__(256) void loss_obj_fwd_kernel(const yh_loss_desc d) {
    const long cells = (long)d.bs * d.na * d.ny * d.nx;
    float acc = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < cells; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % d.nx);
        long r = i / d.nx;
        const int y = (int)(r % d.ny);
        r /= d.ny;
        const int a = (int)(r % d.na);
        const long b = r / d.na;
        acc += bce(d.p[b * d.sb + a * d.sa + y * d.sy + x * d.sx + 4], d.tobj[i], d.obj_pw);
    }
    block_sum_atomic(acc, d.sums + 1);
}

// every cell, backward: the whole gradient row of the cell (zeros, objectness term in slot 4)
__global__ __launch_bounds__(256) void loss_dense_bwd_kernel(const yh_loss_desc d) {
    const long total = (long)d.bs * d.na * d.ny * d.nx * d.no;
    const float sc = *d.scale * d.w_obj;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int o = (int)(i % d.no);
        long cell = i / d.no;
        const int x = (int)(cell % d.nx);
        long r = cell / d.nx;
        const int y = (int)(r % d.ny);
        r /= d.ny;
        const int a = (int)(r % d.na);
        const long b = r / d.na;
        float g = 0.f;
        if (o == 4) g = sc * bce_grad(d.p[b * d.sb + a * d.sa + y * d.sy + x * d.sx + 4], d.tobj[cell], d.obj_pw);
        d.grad[b * d.gb + a * d.ga + y * d.gy + x * d.gx + o] = g;
    }
}

// matched targets, backward: box and class terms added on top of the dense pass (duplicates of a cell accumulate)
__global__ __launch_bounds__(256) void loss_matched_bwd_kernel(const yh_loss_desc d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.nb) return;
    const int b = d.idx[4 * i], a = d.idx[4 * i + 1], gy = d.idx[4 * i + 2], gx = d.idx[4 * i + 3];
    const float* ps = d.p + b * d.sb + a * d.sa + gy * d.sy + gx * d.sx;
    float* gp = d.grad + b * d.gb + a * d.ga + gy * d.gy + gx * d.gx;
    const float sc = *d.scale;
    const float box[4] = {ps[0], ps[1], ps[2], ps[3]};
    const Dual4 g = giou_of(box, d.anchor + 2 * i, d.tbox + 4 * i);
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(gp + k, -sc * d.w_box * g.d[k]);
    if (d.nc > 1) {
        const int tc = d.tcls[i];
        for (int c = 0; c < d.nc; ++c) atomicAdd(gp + 5 + c, sc * d.w_cls * bce_grad(ps[5 + c], c == tc ? d.cp : d.cn, d.cls_pw));
    }
}

static int check_loss(const yh_loss_desc* d, bool bwd) {
    if (!d || !d->p || !d->tobj || d->bs <= 0 || d->na <= 0 || d->ny <= 0 || d->nx <= 0 || d->no < 5 || d->nc != d->no - 5) return YH_EINVAL;
    if (d->nb < 0 || (d->nb > 0 && (!d->idx || !d->tbox || !d->anchor || (d->nc > 1 && !d->tcls) || (!bwd && !d->winner)))) return YH_EINVAL;
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
    if (d->nb > 0) {
        hipLaunchKernelGGL(loss_winner_kernel, dim3((d->nb + 255) / 256), dim3(256), 0, s, *d);
        hipLaunchKernelGGL(loss_matched_fwd_kernel, dim3((d->nb + 255) / 256), dim3(256), 0, s, *d);
    }
    hipLaunchKernelGGL(loss_obj_fwd_kernel, dim3(grid_n((long)d->bs * d->na * d->ny * d->nx)), dim3(256), 0, s, *d);
    return check_launch();
}

extern "C" int yh_yolo_loss_bwd(const yh_loss_desc* d, void* stream) {
    int rc = check_loss(d, true);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_dense_bwd_kernel, dim3(grid_n((long)d->bs * d->na * d->ny * d->nx * d->no)), dim3(256), 0, s, *d);
    if (d->nb > 0) hipLaunchKernelGGL(loss_matched_bwd_kernel, dim3((d->nb + 255) / 256), dim3(256), 0, s, *d);
    return check_launch();
}
