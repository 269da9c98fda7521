// COS-PTQ calibration on the device: the scale search of the reference's quantisers
// (utils/quantized/quantized_ptq_cos.py:64-93 Quantizer.forward in train mode, :838-912 / :1153-1197 the shortcut searches) as ONE
// pass over the tensor for all candidate scales, and the abs-max the quantised concat tracks (:1403-1449).
//
// Candidates j = 0 .. n-1 (n <= 16), scale_j = scale0 * 2^j.  Per element, operation for operation what the modules do in fp32:
//     u = t / scale_j;  r = sign(u) floor(|u| + 0.5);  r = clamp(r, lo, hi);  q = r * scale_j
// (all four are exact for power-of-two scales), then <t, q_j>, <q_j, q_j> and <t, t> accumulated in DOUBLE: every product of two
// fp32 values is exact in double, so the only rounding left is the summation, 2^-29 below the quantities compared.  The host
// search accumulates the same sums in fp32 (torch.cosine_similarity); decisions can only differ where two candidates' cosines
// tie to within fp32 summation noise.  HBM-bound byte work: one read of the tensor, ~35 VALU per candidate per element.
#include "common.h"

namespace yh {

constexpr int CAL_NC = 16;                 // candidates evaluated per pass (the searches use 15 or 8)
constexpr int CAL_SUMS = 2 * CAL_NC + 1;   // dot_j, qq_j, tt
constexpr int CAL_THREADS = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

__global__ __launch_bounds__(CAL_THREADS) void ptq_cos_partial_kernel(const float* __restrict__ t, long count, float scale0, float lo,
                                                                      float hi, int do_clamp, double* __restrict__ partial) {
    double dot[CAL_NC], qq[CAL_NC], tt = 0.0;
#pragma unroll
    for (int j = 0; j < CAL_NC; ++j) dot[j] = qq[j] = 0.0;
    auto element = [&](float v) {
        const double dv = (double)v;
        tt += dv * dv;
        float s = scale0;
#pragma unroll
        for (int j = 0; j < CAL_NC; ++j) {
            const float u = __fdiv_rn(v, s);
            float r = copysignf(floorf(fabsf(u) + 0.5f), u);
            if (do_clamp) r = fminf(fmaxf(r, lo), hi);
            const double q = (double)__fmul_rn(r, s);
            dot[j] += dv * q;
            qq[j] += q * q;
            s = __fmul_rn(s, 2.0f);
        }
    };
    const long stride = (long)gridDim.x * blockDim.x;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if ((((uintptr_t)t) & 15u) == 0) {
        const long n4 = count >> 2;
        const f32x4* t4 = reinterpret_cast<const f32x4*>(t);
        for (long i = i0; i < n4; i += stride) {
            const f32x4 v = t4[i];
            element(v[0]); element(v[1]); element(v[2]); element(v[3]);
        }
        for (long i = (n4 << 2) + i0; i < count; i += stride) element(t[i]);
    } else {
        for (long i = i0; i < count; i += stride) element(t[i]);
    }
    __shared__ double red[CAL_THREADS / 64][CAL_SUMS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < CAL_NC; ++j) {
        const double a = wave_sum(dot[j]), b = wave_sum(qq[j]);
        if (lane == 0) { red[wave][j] = a; red[wave][CAL_NC + j] = b; }
    }
    tt = wave_sum(tt);
    if (lane == 0) red[wave][2 * CAL_NC] = tt;
    __syncthreads();
    if (threadIdx.x < CAL_SUMS) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < CAL_THREADS / 64; ++w) s += red[w][threadIdx.x];   // fixed order: deterministic
        partial[(long)blockIdx.x * CAL_SUMS + threadIdx.x] = s;
    }
}

// one workgroup: sums the per-block partials in block order, then cos_j = dot_j / (|t| |q_j|) (0 when a norm vanishes, as
// torch.cosine_similarity's eps clamp gives) and the FIRST maximum over j < n (the modules' strict `c > best`)
__global__ __launch_bounds__(64) void ptq_cos_final_kernel(const double* __restrict__ partial, int nblocks, int n, double* __restrict__ cos_out,
                                                           int32_t* __restrict__ best) {
    __shared__ double tot[CAL_SUMS];
    if (threadIdx.x < CAL_SUMS) {
        double s = 0.0;
        for (int b = 0; b < nblocks; ++b) s += partial[(long)b * CAL_SUMS + threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double nt = sqrt(tot[2 * CAL_NC]);
        double top = -1.0;
        int arg = 0;
        for (int j = 0; j < n; ++j) {
            const double nq = sqrt(tot[CAL_NC + j]);
            const double c = (nt > 0.0 && nq > 0.0) ? tot[j] / (nt * nq) : 0.0;
            cos_out[j] = c;
            if (c > top) { top = c; arg = j; }
        }
        *best = arg;
    }
}

__global__ __launch_bounds__(256) void absmax_partial_kernel(const float* __restrict__ t, long count, float* __restrict__ partial) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(t[i]));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(64) void absmax_final_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ out) {
    float m = 0.f;
    for (int b = threadIdx.x; b < nblocks; b += 64) m = fmaxf(m, partial[b]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if (threadIdx.x == 0) *out = m;
}

static int cal_blocks(long count) {
    long b = (count + (long)CAL_THREADS * 16 - 1) / ((long)CAL_THREADS * 16);
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return (int)b;
}

}  // namespace yh

using namespace yh;

extern "C" int64_t yh_ptq_search_workspace(int64_t count) {
    if (count <= 0) return 0;
    return (int64_t)cal_blocks(count) * CAL_SUMS * (int64_t)sizeof(double);
}

extern "C" int yh_ptq_cos_search(const float* t, int64_t count, float scale0, int n, float lo, float hi, int do_clamp, void* ws,
                                 int64_t ws_bytes, double* cos_out, int32_t* best, void* stream) {
    if (!t || !ws || !cos_out || !best || count <= 0 || n < 1 || n > CAL_NC || !(scale0 > 0.f) || !(lo < hi)) return YH_EINVAL;
    if ((((uintptr_t)t) & 3u) || (((uintptr_t)ws) & 7u) || (((uintptr_t)cos_out) & 7u)) return YH_EALIGN;
    const int blocks = cal_blocks(count);
    if (ws_bytes < (int64_t)blocks * CAL_SUMS * (int64_t)sizeof(double)) return YH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ptq_cos_partial_kernel, dim3(blocks), dim3(CAL_THREADS), 0, s, t, (long)count, scale0, lo, hi, do_clamp, (double*)ws);
    hipLaunchKernelGGL(ptq_cos_final_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, blocks, n, cos_out, best);
    return check_launch();
}

extern "C" int yh_absmax(const float* t, int64_t count, void* ws, int64_t ws_bytes, float* out, void* stream) {
    if (!t || !ws || !out || count <= 0) return YH_EINVAL;
    const int blocks = cal_blocks(count);
    if (ws_bytes < (int64_t)blocks * (int64_t)sizeof(float)) return YH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(absmax_partial_kernel, dim3(blocks), dim3(256), 0, s, t, (long)count, (float*)ws);
    hipLaunchKernelGGL(absmax_final_kernel, dim3(1), dim3(64), 0, s, (const float*)ws, blocks, out);
    return check_launch();
}
