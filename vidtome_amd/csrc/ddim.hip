// vtm_cfg_ddim: classifier-free-guidance combine + closed-form DDIM update, fused.
// Reference: generate.py:276-278 (`noise_pred = uncond + guidance_scale * (cond - uncond)`) and
// generate.py:281-311 (`pred_next_x`; sampling branch: pred_x0 = (x - sigma*eps)/mu; x' = mu_prev*pred_x0 +
// sigma_prev*eps; inversion branch with the roles of (mu, sigma) and (mu_prev, sigma_prev) exchanged).
// Elementwise, HBM-bound and tiny (a 16-frame chunk of latents is 262 144 elements).  The reference evaluates
// it as separate torch ops, each rounding to the tensor dtype; the kernel rounds after every operation in the
// same order, so the result is bit-identical to the torch CPU expression in fp32 AND in fp16/bf16.  (How the
// 0-dim fp32 coefficients enter a half-precision op is the host wrapper's business: torch's CPU kernels round
// the multipliers b, c, d to the tensor dtype and divide by the unrounded fp32 a; see _lib.cfg_ddim.)
#include "common.h"

#include <algorithm>

namespace {

template <typename T> __device__ __forceinline__ float rnd(float v);   // round-trip through the model dtype
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }
template <> __device__ __forceinline__ float rnd<__half>(float v) { return __half2float(__float2half_rn(v)); }
template <> __device__ __forceinline__ float rnd<vtm_bf16>(float v) { return __bfloat162float(__float2bfloat16(v)); }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f32<__half>(float v) { return __float2half_rn(v); }
template <> __device__ __forceinline__ vtm_bf16 from_f32<vtm_bf16>(float v) { return __float2bfloat16(v); }

template <typename T>
__global__ __launch_bounds__(256) void cfg_ddim_kernel(const T *__restrict__ x, const T *__restrict__ eps_uncond,
                                                       const T *__restrict__ eps_cond, int64_t n, float guidance,
                                                       float a, float b, float c, float d, T *__restrict__ eps_out,
                                                       T *__restrict__ x_out) {
    // pred_x0 = (x - b*eps)/a ; x' = c*pred_x0 + d*eps   (sampling: a=mu, b=sigma, c=mu_prev, d=sigma_prev)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        // __f*_rn intrinsics: one correctly rounded operation each, never contracted into an fma
        const float u = vtm::to_f32(eps_uncond[i]);
        float e = u;
        if (eps_cond) {
            const float diff = rnd<T>(__fsub_rn(vtm::to_f32(eps_cond[i]), u));
            e = rnd<T>(__fadd_rn(u, rnd<T>(__fmul_rn(guidance, diff))));
        }
        if (eps_out) eps_out[i] = from_f32<T>(e);
        if (x_out) {
            const float xv = vtm::to_f32(x[i]);
            const float x0 = rnd<T>(__fdiv_rn(rnd<T>(__fsub_rn(xv, rnd<T>(__fmul_rn(b, e)))), a));
            x_out[i] = from_f32<T>(rnd<T>(__fadd_rn(rnd<T>(__fmul_rn(c, x0)), rnd<T>(__fmul_rn(d, e)))));
        }
    }
}

}  // namespace

VTM_EXPORT int vtm_cfg_ddim(const void *x, const void *eps_uncond, const void *eps_cond, int dtype, int64_t n,
                            float guidance, float a, float b, float c, float d, void *eps_out, void *x_out,
                            vtm_stream_t stream) {
    VTM_REQUIRE(eps_uncond && n >= 0 && (eps_out || x_out), "vtm_cfg_ddim: null pointer");
    VTM_REQUIRE(!x_out || x, "vtm_cfg_ddim: x is required when x_out is requested");
    if (n == 0) return VTM_OK;
    const dim3 grid((unsigned)std::min<int64_t>(vtm::cdiv(n, 256), 4096)), block(256);
    hipStream_t s = vtm::as_stream(stream);
    switch (dtype) {
        case VTM_F32:
            hipLaunchKernelGGL(cfg_ddim_kernel<float>, grid, block, 0, s, (const float *)x, (const float *)eps_uncond,
                               (const float *)eps_cond, n, guidance, a, b, c, d, (float *)eps_out, (float *)x_out);
            break;
        case VTM_F16:
            hipLaunchKernelGGL(cfg_ddim_kernel<__half>, grid, block, 0, s, (const __half *)x, (const __half *)eps_uncond,
                               (const __half *)eps_cond, n, guidance, a, b, c, d, (__half *)eps_out, (__half *)x_out);
            break;
        case VTM_BF16:
            hipLaunchKernelGGL(cfg_ddim_kernel<vtm_bf16>, grid, block, 0, s, (const vtm_bf16 *)x,
                               (const vtm_bf16 *)eps_uncond, (const vtm_bf16 *)eps_cond, n, guidance, a, b, c, d,
                               (vtm_bf16 *)eps_out, (vtm_bf16 *)x_out);
            break;
        default: return vtm::fail(VTM_EINVAL, "vtm_cfg_ddim: unsupported dtype %d", dtype);
    }
    return vtm::launch_status("vtm_cfg_ddim");
}
