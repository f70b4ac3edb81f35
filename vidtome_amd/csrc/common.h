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

// internal cross-file launchers (not part of the C ABI)
int launch_row_norms(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype, int64_t B, int64_t C,
                     const int32_t *rows, int64_t n, float *norms, hipStream_t s, const int32_t *rows2 = nullptr,
                     int64_t n2 = 0, float *norms2 = nullptr, void *zero = nullptr, size_t zero_bytes = 0);
int launch_write_operand(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype, int64_t B,
                         int64_t C, const int32_t *rows, int64_t n, const float *norms, float *out,
                         int64_t n_pad, int64_t C_pad, const int *gate, hipStream_t s,
                         const int32_t *rows2 = nullptr, int64_t n2 = 0, const float *norms2 = nullptr,
                         float *out2 = nullptr, int64_t n_pad2 = 0);
int launch_match(const float *a, const float *b, int64_t B, int64_t Ns, int64_t Nd, int64_t Ns_pad,
                 int64_t Nd_pad, int64_t C_pad, int align, uint64_t *best, const int *gate, bool zero_best,
                 hipStream_t s);

}  // namespace vtm

#define VTM_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return vtm::fail(VTM_EINVAL, __VA_ARGS__); \
    } while (0)
