// vtm_geglu: the gated activation of the block's feed-forward (vidtome/patch.py:187-199 `self.ff(...)`; the Diffusers
// FeedForward of SD blocks is GEGLU: proj(x) -> [value | gate] halves -> value * gelu(gate) -> Linear).  Elementwise,
// HBM-bound: reads (rows, 2 D), writes (rows, D) in one pass instead of torch's chunk + gelu + mul.  gelu is the
// exact (erf) form; the gate is rounded to the tensor dtype before the product, as torch's two ops do.
#include "common.h"

#include <algorithm>

namespace {

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f32<__half>(float v) { return __float2half_rn(v); }
template <> __device__ __forceinline__ vtm_bf16 from_f32<vtm_bf16>(float v) { return __float2bfloat16(v); }

template <typename T>
__global__ __launch_bounds__(256) void geglu_kernel(const T *__restrict__ x, int64_t rows, int64_t D, T *__restrict__ out) {
    constexpr int V = 16 / sizeof(T);   // elements per 16-byte access
    const int64_t chunks = D / V, total = rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / chunks, c = i % chunks;
        const uint4 a = *reinterpret_cast<const uint4 *>(x + r * 2 * D + c * V);
        const uint4 g = *reinterpret_cast<const uint4 *>(x + r * 2 * D + D + c * V);
        const T *pa = reinterpret_cast<const T *>(&a), *pg = reinterpret_cast<const T *>(&g);
        uint4 o;
        T *po = reinterpret_cast<T *>(&o);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float gv = vtm::to_f32(pg[j]);
            const float ge = 0.5f * gv * (1.0f + erff(gv * 0.70710678118654752440f));
            const float gr = vtm::to_f32(from_f32<T>(ge));
            po[j] = from_f32<T>(vtm::to_f32(pa[j]) * gr);
        }
        *reinterpret_cast<uint4 *>(out + r * D + c * V) = o;
    }
}

}  // namespace

VTM_EXPORT int vtm_geglu(const void *x, int dtype, int64_t rows, int64_t D, void *out, vtm_stream_t stream) {
    VTM_REQUIRE(x && out && rows >= 0 && D > 0, "vtm_geglu: bad arguments");
    VTM_REQUIRE(D % 8 == 0, "vtm_geglu: D must be a multiple of 8");
    if (rows == 0) return VTM_OK;
    hipStream_t s = vtm::as_stream(stream);
    const int64_t total = rows * (D / (dtype == VTM_F32 ? 4 : 8));
    const dim3 grid((unsigned)std::min<int64_t>(vtm::cdiv(total, 256), 65536)), block(256);
    switch (dtype) {
        case VTM_F32: hipLaunchKernelGGL(geglu_kernel<float>, grid, block, 0, s, (const float *)x, rows, D, (float *)out); break;
        case VTM_F16: hipLaunchKernelGGL(geglu_kernel<__half>, grid, block, 0, s, (const __half *)x, rows, D, (__half *)out); break;
        case VTM_BF16:
            hipLaunchKernelGGL(geglu_kernel<vtm_bf16>, grid, block, 0, s, (const vtm_bf16 *)x, rows, D, (vtm_bf16 *)out);
            break;
        default: return vtm::fail(VTM_EINVAL, "vtm_geglu: unsupported dtype %d", dtype);
    }
    return vtm::launch_status("vtm_geglu");
}
