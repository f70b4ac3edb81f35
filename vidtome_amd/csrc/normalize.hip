// vtm_normalize_gather: cosine normalisation fused with the src/dst split.
// Reference: vidtome/merge.py:76-85 (randframe) and 383-390 (2s):
//     metric = metric / metric.norm(dim=-1, keepdim=True);  a, b = split(metric)
// HBM-bound.  Two kernels:
//   1. row_norms:      one thread per gathered row runs the canonical k-ascending fmaf chain
//                      (the order is part of the bit-exact contract, so it is not tree-reduced);
//   2. write_operand:  one thread per (row, 8-channel group) re-reads the row (L2-resident), divides
//                      and writes 32 contiguous bytes of the k-interleaved MFMA operand layout.
#include "common.h"

namespace {

using vtm::to_f32;

template <typename T> struct Vec16;  // 16-byte vector of T
template <> struct Vec16<float> { static constexpr int N = 4; using type = float4; };
template <> struct Vec16<__half> { static constexpr int N = 8; using type = uint4; };
template <> struct Vec16<vtm_bf16> { static constexpr int N = 8; using type = uint4; };

template <typename T>
__device__ __forceinline__ const T *pool_row(const T *x0, int64_t P0, const T *x1, int64_t P1,
                                             int64_t b, int64_t r, int64_t C) {
    return r < P0 ? x0 + (b * P0 + r) * C : x1 + (b * P1 + (r - P0)) * C;
}

template <typename T>
__global__ __launch_bounds__(256) void row_norms(const T *__restrict__ x0, int64_t P0,
                                                 const T *__restrict__ x1, int64_t P1, int64_t B,
                                                 int64_t C, const int32_t *__restrict__ rows,
                                                 int64_t n, float *__restrict__ norms) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * n) return;
    const int64_t b = idx / n;
    const T *src = pool_row(x0, P0, x1, P1, b, rows[idx], C);
    constexpr int N = Vec16<T>::N;
    using V = typename Vec16<T>::type;
    float acc = 0.0f;
    for (int64_t k = 0; k < C; k += N) {
        V v = *reinterpret_cast<const V *>(src + k);
        const T *e = reinterpret_cast<const T *>(&v);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float f = to_f32(e[j]);
            acc = __builtin_fmaf(f, f, acc);
        }
    }
    norms[idx] = __builtin_sqrtf(acc);
}

template <typename T>
__global__ __launch_bounds__(256) void write_operand(const T *__restrict__ x0, int64_t P0,
                                                     const T *__restrict__ x1, int64_t P1, int64_t B,
                                                     int64_t C, const int32_t *__restrict__ rows,
                                                     int64_t n, const float *__restrict__ norms,
                                                     float *__restrict__ out, int64_t n_pad,
                                                     int64_t C_pad) {
    // one thread per (8-channel group g, row i), rows fastest -> the two 16-byte panel stores of a wave are
    // contiguous 1 KiB segments.  Panel layout: out[b][g][kh][i][e] = xhat[b, i, 8g + 2e + kh].
    const int64_t G = C_pad / 8;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * n_pad * G) return;
    const int64_t i = idx % n_pad;
    const int64_t bg = idx / n_pad;  // b * G + g
    const int64_t g = bg % G, b = bg / G;
    float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
    if (i < n && g * 8 < C) {
        const T *src = pool_row(x0, P0, x1, P1, b, rows[b * n + i], C) + g * 8;
        const float nrm = norms[b * n + i];
        float f[8];
        if constexpr (sizeof(T) == 4) {
            const float4 v0 = *reinterpret_cast<const float4 *>(src);
            const float4 v1 = *reinterpret_cast<const float4 *>(src + 4);
            f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w;
            f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
        } else {
            const uint4 v = *reinterpret_cast<const uint4 *>(src);
            const T *e = reinterpret_cast<const T *>(&v);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = to_f32(e[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = f[j] / nrm;  // IEEE divide (no fast-math in this build)
        lo = make_float4(f[0], f[2], f[4], f[6]);
        hi = make_float4(f[1], f[3], f[5], f[7]);
    }
    float4 *dst = reinterpret_cast<float4 *>(out) + (bg * 2) * n_pad + i;
    dst[0] = lo;
    dst[n_pad] = hi;
}

template <typename T>
int run(const void *x0, int64_t P0, const void *x1, int64_t P1, int64_t B, int64_t C,
        const int32_t *rows, int64_t n, float *norms, float *out, int64_t n_pad, int64_t C_pad,
        hipStream_t s) {
    if (B * n > 0) {
        const int64_t blocks = vtm::cdiv(B * n, 256);
        hipLaunchKernelGGL(row_norms<T>, dim3((unsigned)blocks), dim3(256), 0, s, (const T *)x0, P0,
                           (const T *)x1, P1, B, C, rows, n, norms);
    }
    const int64_t total = B * n_pad * (C_pad / 8);
    if (total > 0) {
        hipLaunchKernelGGL(write_operand<T>, dim3((unsigned)vtm::cdiv(total, 256)), dim3(256), 0, s,
                           (const T *)x0, P0, (const T *)x1, P1, B, C, rows, n, norms, out, n_pad, C_pad);
    }
    return vtm::launch_status("vtm_normalize_gather");
}

}  // namespace

VTM_EXPORT int vtm_normalize_gather(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype,
                                    int64_t B, int64_t C, const int32_t *rows, int64_t n, float *norms,
                                    float *out, int64_t n_pad, int64_t C_pad, vtm_stream_t stream) {
    VTM_REQUIRE(x0 && rows && out && norms, "vtm_normalize_gather: null pointer");
    VTM_REQUIRE(B > 0 && C > 0 && n >= 0 && P0 >= 0 && P1 >= 0, "vtm_normalize_gather: bad sizes");
    VTM_REQUIRE(P1 == 0 || x1, "vtm_normalize_gather: x1 is null but P1 > 0");
    VTM_REQUIRE(C % 8 == 0, "vtm_normalize_gather: C=%lld must be a multiple of 8", (long long)C);
    VTM_REQUIRE(n_pad >= n && n_pad % VTM_MATCH_ROW_PAD == 0, "vtm_normalize_gather: bad n_pad");
    VTM_REQUIRE(C_pad >= C && C_pad % VTM_MATCH_K_PAD == 0, "vtm_normalize_gather: bad C_pad");
    hipStream_t s = vtm::as_stream(stream);
    switch (dtype) {
        case VTM_F32: return run<float>(x0, P0, x1, P1, B, C, rows, n, norms, out, n_pad, C_pad, s);
        case VTM_F16: return run<__half>(x0, P0, x1, P1, B, C, rows, n, norms, out, n_pad, C_pad, s);
        case VTM_BF16: return run<vtm_bf16>(x0, P0, x1, P1, B, C, rows, n, norms, out, n_pad, C_pad, s);
    }
    return vtm::fail(VTM_EINVAL, "vtm_normalize_gather: unsupported dtype %d", dtype);
}
