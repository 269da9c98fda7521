// Micro-benchmark (not part of libyolo_hip.so): LDS read service rate per CU for the fragment reads of the weight-gradient kernels -
// ds_read_b64_tr_b16 (transposing 8-byte read), plain ds_read_b64, ds_read_b128 - as a function of the waves reading, with or
// without MFMAs issued by the same waves.  One workgroup per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LdsArgs { unsigned long long* out; int iters; int mfma; int reads; };

template <int KIND> __device__ __forceinline__ void rd(unsigned addr, v2i& r) {
    if constexpr (KIND == 0) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr));
    else asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(addr));
}

// KIND 0: tr_b16, 1: b64, 2: b128 (half as many instructions for the same bytes)
template <int KIND>
__global__ __launch_bounds__(768) void lds_probe_kernel(const LdsArgs a) {
    extern __shared__ __attribute__((aligned(128))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane & 15, g = lane >> 4;
    // the x-fragment pattern of conv_wgrad_roll: 128-byte rows, row 8g + q/4 (+ a wave-dependent tap offset), swizzled 32-byte column
    unsigned addr[10];
    for (int k = 0; k < 10; ++k) {
        const int row = 8 * g + (q >> 2) + wave * 5 + k * 7;
        const int f = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
        addr[k] = (unsigned)((row & 511) * 128 + (((k & 3) ^ f) << 5) + (q & 3) * 8);
        if (KIND == 2) addr[k] = (unsigned)(((lane * 16) + k * 1024 + wave * 4096) & 0xffff);
    }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    v2i r[20];
    for (int k = 0; k < 20; ++k) r[k] = v2i{lane, k};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < (a.mfma == 2 ? 0 : a.iters); ++it) {
        if constexpr (KIND == 2) {
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                v4i t;
                asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(addr[k]));
                r[2 * k] = v2i{t[0], t[1]};
                r[2 * k + 1] = v2i{t[2], t[3]};
            }
        } else {
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                if (k < a.reads) {
                    rd<KIND>(addr[k], r[2 * k]);
                    rd<KIND>(addr[k] ^ 32, r[2 * k + 1]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]),
                       "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), "+v"(r[16]), "+v"(r[17]), "+v"(r[18]), "+v"(r[19])
                     :: "memory");
        if (a.mfma) {
#pragma unroll
            for (int m = 0; m < 24; ++m) {
                const v4i ta = {r[(m * 2) % 20][0], r[(m * 2) % 20][1], r[(m * 2 + 1) % 20][0], r[(m * 2 + 1) % 20][1]};
                const v4i tb = {r[(m * 2 + 6) % 20][0], r[(m * 2 + 6) % 20][1], r[(m * 2 + 7) % 20][0], r[(m * 2 + 7) % 20][1]};
                acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ta), __builtin_bit_cast(f16x8, tb), acc[m & 3], 0, 0, 0);
            }
        }
        for (int k = 0; k < 10; ++k) addr[k] = (addr[k] + 4096) & 0xffff;
    }
    if (a.mfma == 2) {
        // "rolling refresh": the wave tile of conv_wgrad_roll (4 A x 6 B fragments, 24 MFMAs) with every fragment re-read right after
        // its last MFMA of the step - B fragment j after the four MFMAs of column j, A fragment i after MFMA (i, 5) - so that the 20
        // reads are spread between the MFMAs instead of forming a segment of their own; r[0..7] = A (2 reads each), r[8..19] = B
        f32x4 c[4][6];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) c[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned long long u0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < a.iters; ++it) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]),
                           "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), "+v"(r[16]), "+v"(r[17]), "+v"(r[18]), "+v"(r[19])
                         :: "memory");
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const v4i tb = {r[8 + 2 * j][0], r[8 + 2 * j][1], r[9 + 2 * j][0], r[9 + 2 * j][1]};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const v4i ta = {r[2 * i][0], r[2 * i][1], r[2 * i + 1][0], r[2 * i + 1][1]};
                    c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ta), __builtin_bit_cast(f16x8, tb), c[i][j], 0, 0, 0);
                    if (j == 5) {
                        __builtin_amdgcn_sched_barrier(0);
                        rd<KIND == 2 ? 1 : KIND>(addr[i], r[2 * i]);
                        rd<KIND == 2 ? 1 : KIND>(addr[i] ^ 32, r[2 * i + 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                addr[4 + j] = (addr[4 + j] + 4096) & 0xffff;
                rd<KIND == 2 ? 1 : KIND>(addr[4 + j], r[8 + 2 * j]);
                rd<KIND == 2 ? 1 : KIND>(addr[4 + j] ^ 32, r[9 + 2 * j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            for (int k = 0; k < 4; ++k) addr[k] = (addr[k] + 4096) & 0xffff;
        }
        __syncthreads();
        const unsigned long long u1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) a.out[blockIdx.x] = u1 - u0;
        float s2 = 0;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) s2 += c[i][j][0] + c[i][j][1] + c[i][j][2] + c[i][j][3];
        if (s2 == 1.2345f) a.out[gridDim.x + threadIdx.x] = 2;
        return;
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) a.out[blockIdx.x] = t1 - t0;
    float s = 0;
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    int x = 0;
    for (int k = 0; k < 20; ++k) x ^= r[k][0] ^ r[k][1];
    if (s == 1.2345f && x == 77) a.out[gridDim.x + threadIdx.x] = 1;
}

extern "C" int lds_probe(const LdsArgs* a, int kind, int grid, int waves, void* stream) {
    const size_t ldsb = 128 * 1024;
    auto k0 = lds_probe_kernel<0>;
    auto k1 = lds_probe_kernel<1>;
    auto k2 = lds_probe_kernel<2>;
    static bool raised = false;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised = true;
    }
    if (kind == 0) hipLaunchKernelGGL(k0, dim3(grid), dim3(64 * waves), ldsb, (hipStream_t)stream, *a);
    else if (kind == 1) hipLaunchKernelGGL(k1, dim3(grid), dim3(64 * waves), ldsb, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(k2, dim3(grid), dim3(64 * waves), ldsb, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}
