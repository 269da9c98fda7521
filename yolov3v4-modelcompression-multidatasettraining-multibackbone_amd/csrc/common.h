// Shared device helpers for libyolo_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <set>
#include <utility>

#include "../../include/yolo_hip.h"

namespace yh {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Activation of the conv epilogue (fp32 in, fp32 out).  `act` is wave-uniform.
// mish(v) = v * tanh(softplus(v)) = v * n / (n + 2) with n = e^v (e^v + 2): one exp, one divide.
__device__ __forceinline__ float activate(float v, int act, float slope) {
    switch (act) {
        case YH_ACT_LEAKY: return v > 0.f ? v : v * slope;
        case YH_ACT_RELU: return fmaxf(v, 0.f);
        case YH_ACT_RELU6: return fminf(fmaxf(v, 0.f), 6.f);
        case YH_ACT_HSWISH: return v * (fminf(fmaxf(v + 3.f, 0.f), 6.f) / 6.f);
        case YH_ACT_MISH: {
            float e = expf(fminf(v, 20.f));
            float n = e * (e + 2.f);
            return v > 20.f ? v : v * (n / (n + 2.f));
        }
        default: return v;
    }
}

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }

inline int check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? YH_OK : (int)e;
}

// Dynamic LDS above 64 KB needs the function attribute raised - per DEVICE (a single process may drive several GPUs:
// DataParallel, `--device 0,1`), so the "already raised" state is keyed by (kernel, current device).
inline hipError_t ensure_dynamic_lds(const void* kern, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> raised;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (raised.count({kern, dev})) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) raised.insert({kern, dev});
    return e;
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// conv_stem_mfma.hip: the 3 x 3 x 3-plane first layer on the matrix cores (YH_EUNSUPPORTED: use the kernel in elementwise.hip)
int launch_stem_mfma(const yh_stem_desc& d, hipStream_t stream);
long stem_mfma_stats_rows(const yh_stem_desc& d);

}  // namespace yh
