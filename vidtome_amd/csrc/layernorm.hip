// vtm_layernorm: the block's norm1 (vidtome/patch.py:139-146, plain-LayerNorm branch: `self.norm1(hidden_states)`),
// i.e. torch.nn.LayerNorm over the channel axis -- the first operation of the patched segment and the producer of
// the matching metric.  One wave per token row, the row lives in registers: read once, write once (HBM-bound;
// cfg-2 top site: 84 MB in + 84 MB out).  Statistics in fp32, two passes over the registers (mean, then the
// centred sum of squares), y = (x - mean) * rstd * gamma + beta evaluated in fp32 and rounded once.
#include "common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;   // 4 waves
constexpr int MAX_CHUNKS = 4;       // 8-channel chunks per lane: C <= 64 * 8 * 4 = 2048

template <typename T>
__device__ __forceinline__ void load8(const T *p, float (&f)[8]) {
    if constexpr (sizeof(T) == 4) {
        const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else {
        const uint4 v = *reinterpret_cast<const uint4 *>(p);
        const T *e = reinterpret_cast<const T *>(&v);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = vtm::to_f32(e[j]);
    }
}

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f32<__half>(float v) { return __float2half_rn(v); }
template <> __device__ __forceinline__ vtm_bf16 from_f32<vtm_bf16>(float v) { return __float2bfloat16(v); }

template <typename T>
__device__ __forceinline__ void store8(T *p, const float (&f)[8]) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4 *>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
    } else {
        uint4 v;
        T *e = reinterpret_cast<T *>(&v);
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = from_f32<T>(f[j]);
        *reinterpret_cast<uint4 *>(p) = v;
    }
}

// all-lanes sum of a wave: butterfly over DPP row operations (no LDS round trip like ds_bpermute-based shuffles)
__device__ __forceinline__ float wave_sum(float v) {
    // within rows of 16 lanes: quad_perm swaps (xor 1, xor 2), then row_half_mirror / row_mirror reversals
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    // across the four rows: every lane of a row now holds the row's sum
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

template <typename T>
__global__ __launch_bounds__(ROWS_PER_BLOCK * 64) void layernorm_kernel(const T *__restrict__ x, const T *__restrict__ gamma,
                                                                        const T *__restrict__ beta, int64_t rows, int C,
                                                                        float eps, T *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int chunks = C / 8;
    const T *xr = x + row * C;
    float v[MAX_CHUNKS][8];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < MAX_CHUNKS; ++i) {
        const int c = lane + 64 * i;
        if (c < chunks) {
            load8(xr + c * 8, v[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[i][j];
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < MAX_CHUNKS; ++i) {
        if (lane + 64 * i < chunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[i][j] - mean;
                q = __builtin_fmaf(d, d, q);
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    T *orow = out + row * C;
#pragma unroll
    for (int i = 0; i < MAX_CHUNKS; ++i) {
        const int c = lane + 64 * i;
        if (c < chunks) {
            float g[8], b[8], y[8];
            if (gamma) load8(gamma + c * 8, g);
            if (beta) load8(beta + c * 8, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float n = (v[i][j] - mean) * rstd;
                y[j] = gamma ? (beta ? __builtin_fmaf(n, g[j], b[j]) : n * g[j]) : (beta ? n + b[j] : n);
            }
            store8(orow + c * 8, y);
        }
    }
}

}  // namespace

VTM_EXPORT int vtm_layernorm(const void *x, const void *gamma, const void *beta, int dtype, int64_t rows, int64_t C,
                             float eps, void *out, vtm_stream_t stream) {
    VTM_REQUIRE(x && out && rows >= 0, "vtm_layernorm: null pointer");
    VTM_REQUIRE(C > 0 && C % 8 == 0 && C <= 64 * 8 * MAX_CHUNKS, "vtm_layernorm: C must be a multiple of 8, <= %d",
                64 * 8 * MAX_CHUNKS);
    if (rows == 0) return VTM_OK;
    const dim3 grid((unsigned)vtm::cdiv(rows, ROWS_PER_BLOCK)), block(ROWS_PER_BLOCK * 64);
    hipStream_t s = vtm::as_stream(stream);
    switch (dtype) {
        case VTM_F32:
            hipLaunchKernelGGL(layernorm_kernel<float>, grid, block, 0, s, (const float *)x, (const float *)gamma,
                               (const float *)beta, rows, (int)C, eps, (float *)out);
            break;
        case VTM_F16:
            hipLaunchKernelGGL(layernorm_kernel<__half>, grid, block, 0, s, (const __half *)x, (const __half *)gamma,
                               (const __half *)beta, rows, (int)C, eps, (__half *)out);
            break;
        case VTM_BF16:
            hipLaunchKernelGGL(layernorm_kernel<vtm_bf16>, grid, block, 0, s, (const vtm_bf16 *)x, (const vtm_bf16 *)gamma,
                               (const vtm_bf16 *)beta, rows, (int)C, eps, (vtm_bf16 *)out);
            break;
        default: return vtm::fail(VTM_EINVAL, "vtm_layernorm: unsupported dtype %d", dtype);
    }
    return vtm::launch_status("vtm_layernorm");
}
