// Shared host-side helpers for libvidtome_hip.so (gfx950 only; no CUDA/HIP dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/vidtome_hip.h"

typedef __hip_bfloat16 vtm_bf16;

#define VTM_EXPORT extern "C" __attribute__((visibility("default")))

namespace vtm {

char *err_buf();            // thread-local message buffer (api.hip)
int fail(int code, const char *fmt, ...);

inline hipStream_t as_stream(vtm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Call after every kernel launch: picks up launch-configuration errors without synchronising.
inline int launch_status(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VTM_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return VTM_OK;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// model-dtype element -> fp32 (exact for all three dtypes)
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float to_f32<vtm_bf16>(vtm_bf16 v) { return __bfloat162float(v); }

// compute units of the CURRENT device (256 on MI355X; 8 XCDs of 32), cached per device ordinal: a process may
// drive several devices (one stream each), and a planner called for device 1 must not see device 0's answer
constexpr int MAX_DEVICES = 64;
inline int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
    return dev;
}
inline int device_cus() {
    static std::atomic<int> cus[MAX_DEVICES];       // zero-initialised; racing first calls store the same value
    const int dev = current_device();
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// 16-byte global accesses with an optional streaming (non-temporal) policy.  Measured on the HBM-bound kernels
// (profiles/r03_hbm_nt.txt): beyond the 256 MB Infinity Cache non-temporal loads / stores are +5 ... +15 % (no line is
// kept that nobody will read again before it is evicted); for working sets the cache holds they are -10 % (the next
// kernel finds its input in the cache only if the producer left it there).  Callers pick by working-set size.
typedef uint32_t vtm_u32x4 __attribute__((ext_vector_type(4)));
constexpr int64_t STREAM_BYTES = 256ll << 20;       // working sets above this are streamed
template <bool NT>
__device__ __forceinline__ uint4 ld16(const void *p) {
    if constexpr (NT) {
        const vtm_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const vtm_u32x4 *>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    } else {
        return *reinterpret_cast<const uint4 *>(p);
    }
}
template <bool NT>
__device__ __forceinline__ void st16(void *p, uint4 v) {
    if constexpr (NT) {
        const vtm_u32x4 w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<vtm_u32x4 *>(p));
    } else {
        *reinterpret_cast<uint4 *>(p) = v;
    }
}

}  // namespace vtm

#define VTM_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return vtm::fail(VTM_EINVAL, __VA_ARGS__); \
    } while (0)
