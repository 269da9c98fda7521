// Micro-benchmark (not part of libyolo_hip.so): what a CU is served when waves stream global memory into LDS by LDS-DMA
// (global_load_lds_dwordx4) or into registers (global_load_dwordx4), as a function of the piece shape (rows x bytes per row, row
// stride), the number of issuing waves, the pieces in flight per wave and the number of active workgroups.  One workgroup per CU
// (LDS sized to force it), no barriers, no compute: the service rate of the memory path alone.  tools/probe/run_probe.py drives it.
#include <hip/hip_runtime.h>
#include <stdint.h>

struct ProbeArgs {
    const unsigned char* src;
    unsigned long long* out;      // per workgroup: cycles
    long wg_stride;               // bytes between the streams of consecutive workgroups
    long wave_stride;             // bytes between the streams of the waves of a workgroup
    long iter_stride;             // bytes a wave's stream advances per piece
    int row_bytes, row_stride;    // a piece = 1024 / row_bytes rows of row_bytes contiguous bytes, row_stride apart
    int iters, depth;             // pieces per wave, pieces a wave keeps in flight (1 .. 16)
    int swz;                      // 1: XOR-permute the 16-byte units inside a row (as the kernels' source-side swizzles do)
    int dz_waves;                 // waves >= dz_waves use the second shape below (two streams per workgroup, as dz + x)
    const unsigned char* src2;
    long wg_stride2, wave_stride2, iter_stride2;
    int row_bytes2, row_stride2;
};

template <int N> __device__ __forceinline__ void waitv() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}
__device__ __forceinline__ void wait_keep(int k) {
    switch (k) {
        case 0: waitv<0>(); break; case 1: waitv<1>(); break; case 2: waitv<2>(); break; case 3: waitv<3>(); break;
        case 4: waitv<4>(); break; case 5: waitv<5>(); break; case 6: waitv<6>(); break; case 7: waitv<7>(); break; case 8: waitv<8>(); break; case 12: waitv<12>(); break;
        default: waitv<16>(); break;
    }
}

// MODE 0: LDS-DMA; MODE 1: loads into registers (consumed by a running XOR so they are not dead)
template <int MODE>
__global__ __launch_bounds__(1024) void dma_probe_kernel(const ProbeArgs a) {
    extern __shared__ __attribute__((aligned(128))) unsigned char lds[];
    typedef const void __attribute__((address_space(1))) * gptr_t;
    typedef void __attribute__((address_space(3))) * lptr_t;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool second = wave >= a.dz_waves;
    const int rb = second ? a.row_bytes2 : a.row_bytes, rs = second ? a.row_stride2 : a.row_stride;
    const int upr = rb >> 4;                        // 16-byte units per row
    const int row = lane / upr;
    int unit = lane - row * upr;
    if (a.swz && upr >= 8) unit ^= ((row & 3) << 1);
    const int w2 = second ? wave - a.dz_waves : wave;
    const unsigned char* p = (second ? a.src2 : a.src) + (long)blockIdx.x * (second ? a.wg_stride2 : a.wg_stride) +
                             (long)w2 * (second ? a.wave_stride2 : a.wave_stride) + (long)row * rs + unit * 16;
    const long step = second ? a.iter_stride2 : a.iter_stride;
    unsigned char* slot = lds + wave * 4096;        // four 1 KB slots per wave (a bandwidth probe: the bytes that land are never read)
    int si = 0;
    uint4 accv = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (MODE == 0) {
        for (int i = 0; i < a.iters; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(slot + si * 1024), 16, 0, 0);
            p += step;
            si = (si + 1) & 3;
            if (i >= a.depth - 1) wait_keep(a.depth - 1);
        }
        waitv<0>();
    } else if constexpr (MODE == 2 || MODE == 3) {
        // the step structure of conv_wgrad_roll_kernel's stream: per step every wave issues ONE piece and waits until its piece of two
        // steps on has landed; MODE 2: three barriers per step, the three wave groups (waves 0-3, 4-7, 8-11) take their turn in
        // different barrier intervals (staggered); MODE 3: one barrier per step, every wave issues in the same interval
        const int grp = wave >> 2;
        for (int k = 0; k < a.depth - 1; ++k) {
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(slot + si * 1024), 16, 0, 0);
            p += step;
            si = (si + 1) & 3;
        }
        __builtin_amdgcn_s_barrier();
        if constexpr (MODE == 2) for (int k = 0; k < grp; ++k) __builtin_amdgcn_s_barrier();
        for (int i = 0; i < a.iters; ++i) {
            if constexpr (MODE == 2) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(slot + si * 1024), 16, 0, 0);
            p += step;
            si = (si + 1) & 3;
            wait_keep(a.depth - 2);
            __builtin_amdgcn_s_barrier();
        }
        if constexpr (MODE == 2) for (int k = grp; k < 2; ++k) __builtin_amdgcn_s_barrier();
        waitv<0>();
    } else {
        // register form: `depth` independent 16-byte loads per trip
        for (int i = 0; i < a.iters; i += 4) {
            uint4 v0 = *reinterpret_cast<const uint4*>(p);
            uint4 v1 = *reinterpret_cast<const uint4*>(p + step);
            uint4 v2 = *reinterpret_cast<const uint4*>(p + 2 * step);
            uint4 v3 = *reinterpret_cast<const uint4*>(p + 3 * step);
            p += 4 * step;
            accv.x ^= v0.x ^ v1.x ^ v2.x ^ v3.x;
            accv.y ^= v0.y ^ v1.y ^ v2.y ^ v3.y;
            accv.z ^= v0.z ^ v1.z ^ v2.z ^ v3.z;
            accv.w ^= v0.w ^ v1.w ^ v2.w ^ v3.w;
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) a.out[blockIdx.x] = t1 - t0;
    if (MODE == 1 && accv.x == 0x12345678u && accv.y == 0x9abcdef0u) a.out[gridDim.x + threadIdx.x] = accv.z + accv.w;
}

extern "C" int dma_probe(const ProbeArgs* a, int mode, int grid, int waves, void* stream) {
    const size_t ldsb = 160 * 1024 - 1024;       // one workgroup per CU
    auto k0 = dma_probe_kernel<0>;
    auto k1 = dma_probe_kernel<1>;
    if (mode == 2) { hipLaunchKernelGGL(dma_probe_kernel<2>, dim3(grid), dim3(64 * waves), ldsb - 1024, (hipStream_t)stream, *a); return (int)hipGetLastError(); }
    if (mode == 3) { hipLaunchKernelGGL(dma_probe_kernel<3>, dim3(grid), dim3(64 * waves), ldsb - 1024, (hipStream_t)stream, *a); return (int)hipGetLastError(); }
    static bool raised = false;
    if (!raised) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised = true;
    }
    if (mode == 0) hipLaunchKernelGGL(k0, dim3(grid), dim3(64 * waves), ldsb, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(k1, dim3(grid), dim3(64 * waves), ldsb, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}
