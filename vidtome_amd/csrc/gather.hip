// Row gathers: the composed merge closure and the composed unmerge closure (+ residual).
// Reference: vidtome/merge.py:119-133 / 423-437 (merge, replace mode = pure row selection),
// merge.py:135-155 / 439-460 (unmerge: every output row written exactly once -> a gather with the
// inverse map, no zero fill, no atomics), vidtome/patch.py:80 (global-token update), patch.py:168-169
// (unmerge + residual add).
// HBM-bound: one 16-byte chunk per thread, consecutive threads walk consecutive chunks of a row, so a
// wave moves 1 KiB of contiguous bytes per row segment.
#include "common.h"

namespace {

__device__ __forceinline__ const char *pool_row_bytes(const char *x0, int64_t P0, const char *x1,
                                                      int64_t P1, int64_t b, int64_t r,
                                                      int64_t row_bytes) {
    return r < P0 ? x0 + (b * P0 + r) * row_bytes : x1 + (b * P1 + (r - P0)) * row_bytes;
}

template <bool NT>
__global__ __launch_bounds__(256) void gather_rows_kernel(const char *__restrict__ x0, int64_t P0,
                                                          const char *__restrict__ x1, int64_t P1,
                                                          int64_t B, int64_t row_bytes,
                                                          const int32_t *__restrict__ map, int64_t M,
                                                          char *__restrict__ out, int64_t out_rows) {
    const int64_t chunks = row_bytes / 16;
    const int64_t total = B * M * chunks;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = idx % chunks;
        const int64_t row = idx / chunks;  // b * M + p
        const int64_t b = row / M;
        const char *src = pool_row_bytes(x0, P0, x1, P1, b, map[row], row_bytes);
        vtm::st16<NT>(out + (b * out_rows + row % M) * row_bytes + c * 16, vtm::ld16<NT>(src + c * 16));
    }
}

template <typename T> struct Add16;
template <> struct Add16<float> {
    __device__ static uint4 apply(uint4 a, uint4 b) {
        float4 x = *reinterpret_cast<float4 *>(&a), y = *reinterpret_cast<float4 *>(&b);
        float4 r = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        return *reinterpret_cast<uint4 *>(&r);
    }
};
template <> struct Add16<__half> {
    // fp16 + fp16 rounded once to fp16 (the sum of two halves is exact in fp32) == torch's half add
    __device__ static uint4 apply(uint4 a, uint4 b) {
        const __half *x = reinterpret_cast<const __half *>(&a), *y = reinterpret_cast<const __half *>(&b);
        uint4 r;
        __half *o = reinterpret_cast<__half *>(&r);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = __float2half_rn(__half2float(x[j]) + __half2float(y[j]));
        return r;
    }
};
template <> struct Add16<vtm_bf16> {
    __device__ static uint4 apply(uint4 a, uint4 b) {
        const vtm_bf16 *x = reinterpret_cast<const vtm_bf16 *>(&a),
                           *y = reinterpret_cast<const vtm_bf16 *>(&b);
        uint4 r;
        vtm_bf16 *o = reinterpret_cast<vtm_bf16 *>(&r);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = __float2bfloat16(__bfloat162float(x[j]) + __bfloat162float(y[j]));
        return r;
    }
};

template <typename T, bool NT>
__global__ __launch_bounds__(256) void unmerge_add_kernel(const char *__restrict__ y, int64_t M,
                                                          const int32_t *__restrict__ inv,
                                                          const char *__restrict__ resid, int64_t B,
                                                          int64_t L, int64_t row_bytes,
                                                          char *__restrict__ out) {
    const int64_t chunks = row_bytes / 16;
    const int64_t total = B * L * chunks;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = idx % chunks;
        const int64_t row = idx / chunks;  // b * L + i
        const int64_t b = row / L;
        uint4 v = vtm::ld16<false>(y + (b * M + inv[row]) * row_bytes + c * 16);   // (merged rows are read by several positions)
        if (resid) {
            const uint4 r = vtm::ld16<NT>(resid + row * row_bytes + c * 16);
            v = Add16<T>::apply(v, r);
        }
        vtm::st16<NT>(out + row * row_bytes + c * 16, v);
    }
}

// ---- the reference's other merge modes (merge.py:127-131: dst.scatter_reduce(-2, dst_idx, src, reduce=mode,
// include_self=True)); never reached from compute_merge, kept for the closure protocol ----
// torch's CPU kernel folds the sources of a destination row into it ONE BY ONE IN INDEX ORDER, in fp32 whatever the tensor
// dtype; a 16-bit tensor is rounded once at the end, "mean" then divides by (1 + number of sources) and rounds again; amax /
// amin propagate NaN (probed bit for bit, tests/golden/make_golden_modes.py).  The pairs arrive sorted by destination
// (stable, so index order survives inside a destination's segment: vtm_sort_desc on the host side): a thread owns 8
// channels of one dst row, finds its segment by bisection and walks it.
enum { RED_SUM = 0, RED_PROD = 1, RED_MEAN = 2, RED_AMAX = 3, RED_AMIN = 4 };

template <typename T>
__device__ __forceinline__ T round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ __half round_to<__half>(float v) { return __float2half_rn(v); }
template <> __device__ __forceinline__ vtm_bf16 round_to<vtm_bf16>(float v) { return __float2bfloat16(v); }

template <typename T>
__global__ __launch_bounds__(256) void merge_reduce_kernel(const T *__restrict__ x, int64_t N, int64_t C, int64_t B,
                                                           const int32_t *__restrict__ src_rows, const int32_t *__restrict__ dst_rows,
                                                           const int32_t *__restrict__ seg_dst, const int32_t *__restrict__ seg_order,
                                                           int64_t r, int64_t Nd, int mode, T *__restrict__ out, int64_t out_ld,
                                                           int64_t out_row0) {
    const int64_t chunks = C / 8;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * Nd * chunks) return;
    const int64_t c = idx % chunks, j = (idx / chunks) % Nd, b = idx / (chunks * Nd);
    const int32_t *sd = seg_dst + b * r;
    int64_t lo = 0, hi = r;                    // first pair whose destination is >= j
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (sd[mid] < j) lo = mid + 1; else hi = mid;
    }
    float acc[8];
    const T *self = x + (b * N + dst_rows[b * Nd + j]) * C + c * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = vtm::to_f32(self[e]);
    float count = 1.0f;
    for (int64_t p = lo; p < r && sd[p] == j; ++p) {
        const T *s = x + (b * N + src_rows[b * r + seg_order[b * r + p]]) * C + c * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = vtm::to_f32(s[e]);
            switch (mode) {
                case RED_PROD: acc[e] = acc[e] * v; break;
                case RED_AMAX: acc[e] = (v != v || acc[e] != acc[e]) ? NAN : fmaxf(acc[e], v); break;
                case RED_AMIN: acc[e] = (v != v || acc[e] != acc[e]) ? NAN : fminf(acc[e], v); break;
                default: acc[e] = acc[e] + v;
            }
        }
        count += 1.0f;
    }
    T *o = out + (b * out_ld + out_row0 + j) * C + c * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        T v = round_to<T>(acc[e]);
        if (mode == RED_MEAN) v = round_to<T>(vtm::to_f32(v) / count);
        o[e] = v;
    }
}

inline int esize(int dtype) { return dtype == VTM_F32 ? 4 : (dtype == VTM_F16 || dtype == VTM_BF16) ? 2 : 0; }

inline unsigned grid_for(int64_t total) {
    int64_t blocks = vtm::cdiv(total, 256);
    const int64_t cap = 256 * 16;  // 256 CUs x 16 blocks, grid-stride beyond that
    return (unsigned)(blocks < cap ? (blocks > 0 ? blocks : 1) : cap);
}

// V^T of an un-merged site: the q | k | v projection GEMM leaves v as columns [c0, c0 + C) of its (BF, N, ldx) output, the
// attention core reads V channel-major (BF, C, ldo >= N).  64 x 64 tiles of 16-bit elements through LDS: 128-byte row
// segments in, 128-byte channel segments out (a strided torch copy of the same 21 MB took 32 us at the C = 1280 sites).
__global__ __launch_bounds__(256) void transpose_cols_kernel(const uint16_t *__restrict__ x, int64_t ldx, int64_t N, int64_t C,
                                                             uint16_t *__restrict__ out, int64_t ldo) {
    __shared__ uint16_t tile[64][64 + 2];      // 33 words per row: column reads step through the banks
    const int64_t b = blockIdx.z, n0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
    const int tid = threadIdx.x;
    const uint16_t *xb = x + b * N * ldx;
    uint16_t *ob = out + b * C * ldo;
#pragma unroll
    for (int t = 0; t < 2; ++t) {              // 64 rows x 8 pieces of 8 elements
        const int q = tid + 256 * t, r = q >> 3, p = q & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (n0 + r < N && c0 + p * 8 < C) v = *reinterpret_cast<const uint4 *>(xb + (n0 + r) * ldx + c0 + p * 8);
        const uint16_t *e = reinterpret_cast<const uint16_t *>(&v);
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[r][p * 8 + k] = e[k];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {              // 64 channels x 8 pieces of 8 tokens
        const int q = tid + 256 * t, c = q >> 3, p = q & 7;
        if (c0 + c < C && n0 + p * 8 < ldo) {  // (ldo is a multiple of 8: the piece stays inside the row; tokens >= N get zeros)
            uint4 v;
            uint16_t *e = reinterpret_cast<uint16_t *>(&v);
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = tile[p * 8 + k][c];
            *reinterpret_cast<uint4 *>(ob + (c0 + c) * ldo + n0 + p * 8) = v;
        }
    }
}

}  // namespace

VTM_EXPORT int vtm_gather_rows(const void *x0, int64_t P0, const void *x1, int64_t P1, int dtype,
                               int64_t B, int64_t C, const int32_t *map, int64_t M, void *out,
                               int64_t out_rows, vtm_stream_t stream) {
    VTM_REQUIRE(x0 && map && out, "vtm_gather_rows: null pointer");
    VTM_REQUIRE(P1 == 0 || x1, "vtm_gather_rows: x1 is null but P1 > 0");
    const int es = esize(dtype);
    VTM_REQUIRE(es, "vtm_gather_rows: unsupported dtype %d", dtype);
    VTM_REQUIRE(out_rows >= M, "vtm_gather_rows: out_rows < M");
    VTM_REQUIRE(B > 0 && C > 0 && M >= 0 && (C * es) % 16 == 0,
                "vtm_gather_rows: row size %lld B must be a multiple of 16", (long long)(C * es));
    if (M == 0) return VTM_OK;
    const int64_t total = B * M * (C * es / 16);
    if (2 * total * 16 > vtm::STREAM_BYTES)     // rows read + rows written do not fit the Infinity Cache: stream them
        hipLaunchKernelGGL(gather_rows_kernel<true>, dim3(grid_for(total)), dim3(256), 0, vtm::as_stream(stream),
                           (const char *)x0, P0, (const char *)x1, P1, B, C * es, map, M, (char *)out, out_rows);
    else
        hipLaunchKernelGGL(gather_rows_kernel<false>, dim3(grid_for(total)), dim3(256), 0, vtm::as_stream(stream),
                           (const char *)x0, P0, (const char *)x1, P1, B, C * es, map, M, (char *)out, out_rows);
    return vtm::launch_status("vtm_gather_rows");
}

VTM_EXPORT int vtm_unmerge_add(const void *y, int64_t M, const int32_t *inv, const void *resid, int dtype,
                               int64_t B, int64_t L, int64_t C, void *out, vtm_stream_t stream) {
    VTM_REQUIRE(y && inv && out, "vtm_unmerge_add: null pointer");
    const int es = esize(dtype);
    VTM_REQUIRE(es, "vtm_unmerge_add: unsupported dtype %d", dtype);
    VTM_REQUIRE(B > 0 && C > 0 && L > 0 && M > 0 && (C * es) % 16 == 0,
                "vtm_unmerge_add: bad sizes (row bytes must be a multiple of 16)");
    const int64_t total = B * L * (C * es / 16);
    hipStream_t s = vtm::as_stream(stream);
    const unsigned g = grid_for(total);
    const bool nt = (B * M * C * es + (resid ? 2 : 1) * total * 16) > vtm::STREAM_BYTES;
#define VTM_UNMERGE(T, NT_)                                                                                            \
    hipLaunchKernelGGL((unmerge_add_kernel<T, NT_>), dim3(g), dim3(256), 0, s, (const char *)y, M, inv, (const char *)resid, B, \
                       L, C * es, (char *)out)
    switch (dtype) {
        case VTM_F32: if (nt) VTM_UNMERGE(float, true); else VTM_UNMERGE(float, false); break;
        case VTM_F16: if (nt) VTM_UNMERGE(__half, true); else VTM_UNMERGE(__half, false); break;
        default: if (nt) VTM_UNMERGE(vtm_bf16, true); else VTM_UNMERGE(vtm_bf16, false);
    }
#undef VTM_UNMERGE
    return vtm::launch_status("vtm_unmerge_add");
}

VTM_EXPORT int vtm_merge_reduce(const void *x, int dtype, int64_t B, int64_t N, int64_t C, const int32_t *src_rows,
                                const int32_t *dst_rows, const int32_t *seg_dst, const int32_t *seg_order, int64_t r,
                                int64_t Nd, int mode, void *out, int64_t out_ld, int64_t out_row0, vtm_stream_t stream) {
    VTM_REQUIRE(x && dst_rows && out && (r == 0 || (src_rows && seg_dst && seg_order)), "vtm_merge_reduce: null pointer");
    VTM_REQUIRE(esize(dtype), "vtm_merge_reduce: unsupported dtype %d", dtype);
    VTM_REQUIRE(B > 0 && N > 0 && C > 0 && C % 8 == 0 && r >= 0 && Nd > 0 && out_ld >= out_row0 + Nd && out_row0 >= 0,
                "vtm_merge_reduce: bad sizes");
    VTM_REQUIRE(mode >= VTM_REDUCE_SUM && mode <= VTM_REDUCE_AMIN, "vtm_merge_reduce: unknown reduce mode %d", mode);
    const int64_t total = B * Nd * (C / 8);
    const dim3 grid((unsigned)vtm::cdiv(total, 256)), block(256);
    hipStream_t s = vtm::as_stream(stream);
    switch (dtype) {
        case VTM_F32:
            hipLaunchKernelGGL(merge_reduce_kernel<float>, grid, block, 0, s, (const float *)x, N, C, B, src_rows, dst_rows,
                               seg_dst, seg_order, r, Nd, mode, (float *)out, out_ld, out_row0);
            break;
        case VTM_F16:
            hipLaunchKernelGGL(merge_reduce_kernel<__half>, grid, block, 0, s, (const __half *)x, N, C, B, src_rows, dst_rows,
                               seg_dst, seg_order, r, Nd, mode, (__half *)out, out_ld, out_row0);
            break;
        default:
            hipLaunchKernelGGL(merge_reduce_kernel<vtm_bf16>, grid, block, 0, s, (const vtm_bf16 *)x, N, C, B, src_rows,
                               dst_rows, seg_dst, seg_order, r, Nd, mode, (vtm_bf16 *)out, out_ld, out_row0);
    }
    return vtm::launch_status("vtm_merge_reduce");
}

VTM_EXPORT int vtm_transpose_cols(const void *x, int64_t ldx, int dtype, int64_t BF, int64_t N, int64_t C, void *out,
                                  int64_t ldo, vtm_stream_t stream) {
    VTM_REQUIRE(x && out, "vtm_transpose_cols: null pointer");
    VTM_REQUIRE(dtype == VTM_F16 || dtype == VTM_BF16, "vtm_transpose_cols: 16-bit tokens only");
    VTM_REQUIRE(BF > 0 && N > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldx >= C && ldo % 8 == 0 && ldo >= N,
                "vtm_transpose_cols: bad sizes (C, ldx, ldo multiples of 8; ldo >= N)");
    VTM_REQUIRE(BF < 65536 && vtm::cdiv(C, 64) < 65536, "vtm_transpose_cols: grid too large");
    VTM_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0, "vtm_transpose_cols: 16-byte aligned operands");
    hipLaunchKernelGGL(transpose_cols_kernel, dim3((unsigned)vtm::cdiv(ldo, 64), (unsigned)vtm::cdiv(C, 64), (unsigned)BF), dim3(256), 0,
                       vtm::as_stream(stream), (const uint16_t *)x, ldx, N, C, (uint16_t *)out, ldo);
    return vtm::launch_status("vtm_transpose_cols");
}
