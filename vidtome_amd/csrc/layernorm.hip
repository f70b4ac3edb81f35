// vtm_layernorm: the block's norm1 (vidtome/patch.py:139-146, plain-LayerNorm branch: `self.norm1(hidden_states)`),
// i.e. torch.nn.LayerNorm over the channel axis -- the first operation of the patched segment and the producer of
// the matching metric.  One wave per token row (2-4 rows per wave in flight), the rows live in registers: read once,
// write once (HBM-bound; cfg-2 top site: 84 MB in + 84 MB out).  Statistics in fp32, two passes over the registers (mean, then the
// centred sum of squares), y = (x - mean) * rstd * gamma + beta evaluated in fp32 and rounded once.
#include "common.h"

namespace {

#ifndef VTM_LN_RG
#define VTM_LN_RG 0         // A/B build switch: rows per wave of the wave-per-row kernel (0 = the shipped 4 / 2 / 2 / 2)
#endif
#ifndef VTM_LN_WAVES
#define VTM_LN_WAVES 4          // (A/B build switch)
#endif
constexpr int WAVES_PER_BLOCK = VTM_LN_WAVES;
constexpr int MAX_CHUNKS = 4;       // 8-channel chunks per lane: C <= 64 * 8 * 4 = 2048

template <typename T, bool NT = false>
__device__ __forceinline__ void load8(const T *p, float (&f)[8]) {
    if constexpr (sizeof(T) == 4) {
        const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else {
        const uint4 v = vtm::ld16<NT>(p);
        const T *e = reinterpret_cast<const T *>(&v);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = vtm::to_f32(e[j]);
    }
}

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f32<__half>(float v) { return __float2half_rn(v); }
template <> __device__ __forceinline__ vtm_bf16 from_f32<vtm_bf16>(float v) { return __float2bfloat16(v); }

template <typename T, bool NT = false>
__device__ __forceinline__ void store8(T *p, const float (&f)[8]) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4 *>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
    } else {
        uint4 v;
        T *e = reinterpret_cast<T *>(&v);
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = from_f32<T>(f[j]);
        vtm::st16<NT>(p, v);
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

// NCH: 8-channel chunks per lane (C <= 512 NCH); R: rows per wave, all loaded before the first reduction so that
// a wave keeps R rows of HBM traffic in flight (one row per wave leaves the kernel latency-bound: 2.4 TB/s)
template <typename T, int NCH, int R, bool NT>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void layernorm_kernel(const T *__restrict__ x, const T *__restrict__ gamma,
                                                                         const T *__restrict__ beta, int64_t rows, int C,
                                                                         float eps, T *__restrict__ out, int64_t panel_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    const int chunks = C / 8;
    float v[R][NCH][8];
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row0 + r < rows ? row0 + r : rows - 1;   // surplus rows recompute the last one, unstored
        const T *xr = x + row * C;
        s[r] = 0.0f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < chunks) load8<T, NT>(xr + c * 8, v[r][i]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            if (lane + 64 * i < chunks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) s[r] += v[r][i][j];
            }
    float mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) mean[r] = wave_sum(s[r]) / (float)C;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (lane + 64 * i < chunks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = v[r][i][j] - mean[r];
                    q = __builtin_fmaf(d, d, q);
                }
            }
        }
        s[r] = q;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) rstd[r] = 1.0f / sqrtf(wave_sum(s[r]) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < chunks) {
            float g[8], b[8];
            if (gamma) load8(gamma + c * 8, g);
            if (beta) load8(beta + c * 8, b);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (row0 + r < rows) {
                    float y[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float n = (v[r][i][j] - mean[r]) * rstd[r];
                        y[j] = gamma ? (beta ? __builtin_fmaf(n, g[j], b[j]) : n * g[j]) : (beta ? n + b[j] : n);
                    }
                    // panel_rows > 0: the k-panel layout [C / 8][panel_rows][8] of the panel GEMMs (ff.hip)
                    store8<T, NT>(panel_rows ? out + ((int64_t)c * panel_rows + (row0 + r)) * 8 : out + (row0 + r) * C + c * 8, y);
                }
            }
        }
    }
}

// sum over aligned groups of LPR lanes (8, 16 or 32): the first log2(LPR) steps of wave_sum's butterfly
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    if constexpr (LPR >= 16)
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    if constexpr (LPR >= 32) v += __shfl_xor(v, 16, 64);
    return v;
}

// The SD channel counts (C = 40 LPR, LPR = 8 / 16 / 32 for 320 / 640 / 1280): LPR lanes per row, every lane 5 pieces of
// 16 bytes at a stride of LPR pieces -- all 64 lanes busy (a wave per 320-channel row keeps 24 of them idle), a load
// instruction covers 64 / LPR rows with 16 LPR contiguous bytes each, and a wave keeps R * 64 / LPR rows in flight.
template <typename T, int LPR, int R, bool NT>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void layernorm_rows_kernel(const T *__restrict__ x, const T *__restrict__ gamma,
                                                                              const T *__restrict__ beta, int64_t rows, float eps,
                                                                              T *__restrict__ out, int64_t panel_rows) {
    constexpr int NCH = 5, C = 8 * NCH * LPR, RPW = 64 / LPR;   // pieces per lane, channels, rows per wave and round
    const int lane = threadIdx.x & 63, g = lane % LPR, sub = lane / LPR;
    const int64_t row0 = ((int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * (RPW * R) + sub;
    float v[R][NCH][8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t row = row0 + (int64_t)r * RPW;
        if (row >= rows) row = rows - 1;                        // surplus rows recompute the last one, unstored
        const T *xr = x + row * C;
#pragma unroll
        for (int i = 0; i < NCH; ++i) load8<T, NT>(xr + (g + LPR * i) * 8, v[r][i]);
    }
    float mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[r][i][j];
        mean[r] = group_sum<LPR>(s) / (float)C;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[r][i][j] - mean[r];
                q = __builtin_fmaf(d, d, q);
            }
        rstd[r] = 1.0f / sqrtf(group_sum<LPR>(q) / (float)C + eps);
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = g + LPR * i;
        float gm[8], bt[8];
        if (gamma) load8(gamma + c * 8, gm);
        if (beta) load8(beta + c * 8, bt);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = row0 + (int64_t)r * RPW;
            if (row < rows) {
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float n = (v[r][i][j] - mean[r]) * rstd[r];
                    y[j] = gamma ? (beta ? __builtin_fmaf(n, gm[j], bt[j]) : n * gm[j]) : (beta ? n + bt[j] : n);
                }
                store8<T, NT>(panel_rows ? out + ((int64_t)c * panel_rows + row) * 8 : out + row * C + c * 8, y);
            }
        }
    }
}

template <typename T>
void launch_layernorm(const void *x, const void *gamma, const void *beta, int64_t rows, int64_t C, float eps, void *out,
                      int64_t panel_rows, hipStream_t s) {
    // input + output beyond the Infinity Cache: streaming loads / stores (common.h)
    const bool nt = 2 * rows * C * (int64_t)sizeof(T) > vtm::STREAM_BYTES;
#ifndef VTM_OLD_LN   // (A/B build switch)
    // measured (profiles/r02_sweeps.txt): +24 % HBM rate at C = 320 beyond the Infinity Cache
    // round 3 (one round of rows per wave, profiles/r03_hbm_nt.txt): the lanes-per-row kernel also wins at C = 640 (6.0 vs 4.2
    // TB/s beyond the Infinity Cache, +6 % at the cfg-2 mid sites) and at C = 1280 for large row counts; the few-thousand-row
    // C = 1280 sites stay with a wave per row (14.7 vs 16.7 us)
    if (C == 320 || C == 640 || (C == 1280 && rows >= 32768)) {
#ifndef VTM_LN_R
#define VTM_LN_R 1          // rows per lane group in flight (A/B in profiles/r03_hbm_nt.txt: 1 beats 2, 3, 4)
#endif
        constexpr int R = VTM_LN_R;
        const int lpr = (int)(C / 40);
        const dim3 g((unsigned)vtm::cdiv(rows, (int64_t)WAVES_PER_BLOCK * (64 / lpr) * R)), b(WAVES_PER_BLOCK * 64);
#define VTM_LNR(LPR, NT_)                                                                                              \
    hipLaunchKernelGGL((layernorm_rows_kernel<T, LPR, R, NT_>), g, b, 0, s, (const T *)x, (const T *)gamma, (const T *)beta, rows, \
                       eps, (T *)out, panel_rows)
        if (lpr == 8) { if (nt) VTM_LNR(8, true); else VTM_LNR(8, false); }
        else if (lpr == 16) { if (nt) VTM_LNR(16, true); else VTM_LNR(16, false); }
        else { if (nt) VTM_LNR(32, true); else VTM_LNR(32, false); }
#undef VTM_LNR
        return;
    }
#endif
    const int nch = (int)vtm::cdiv(C, 512);
    const int R = VTM_LN_RG ? VTM_LN_RG : (nch == 1 ? 4 : 2);
    const dim3 grid((unsigned)vtm::cdiv(rows, (int64_t)WAVES_PER_BLOCK * R)), block(WAVES_PER_BLOCK * 64);
#define VTM_LN(NCH, RR)                                                                                               \
    do {                                                                                                              \
        if (nt)                                                                                                       \
            hipLaunchKernelGGL((layernorm_kernel<T, NCH, RR, true>), grid, block, 0, s, (const T *)x, (const T *)gamma,   \
                               (const T *)beta, rows, (int)C, eps, (T *)out, panel_rows);                              \
        else                                                                                                          \
            hipLaunchKernelGGL((layernorm_kernel<T, NCH, RR, false>), grid, block, 0, s, (const T *)x, (const T *)gamma,  \
                               (const T *)beta, rows, (int)C, eps, (T *)out, panel_rows);                              \
    } while (0)
    switch (nch) {
        case 1: VTM_LN(1, (VTM_LN_RG ? VTM_LN_RG : 4)); break;
        case 2: VTM_LN(2, (VTM_LN_RG ? VTM_LN_RG : 2)); break;
        case 3: VTM_LN(3, (VTM_LN_RG ? VTM_LN_RG : 2)); break;
        default: VTM_LN(4, (VTM_LN_RG ? VTM_LN_RG : 2)); break;
    }
#undef VTM_LN
}

}  // namespace

VTM_EXPORT int vtm_layernorm(const void *x, const void *gamma, const void *beta, int dtype, int64_t rows, int64_t C,
                             float eps, void *out, vtm_stream_t stream) {
    VTM_REQUIRE(x && out && rows >= 0, "vtm_layernorm: null pointer");
    VTM_REQUIRE(C > 0 && C % 8 == 0 && C <= 64 * 8 * MAX_CHUNKS, "vtm_layernorm: C must be a multiple of 8, <= %d",
                64 * 8 * MAX_CHUNKS);
    if (rows == 0) return VTM_OK;
    hipStream_t s = vtm::as_stream(stream);
    switch (dtype) {
        case VTM_F32: launch_layernorm<float>(x, gamma, beta, rows, C, eps, out, 0, s); break;
        case VTM_F16: launch_layernorm<__half>(x, gamma, beta, rows, C, eps, out, 0, s); break;
        case VTM_BF16: launch_layernorm<vtm_bf16>(x, gamma, beta, rows, C, eps, out, 0, s); break;
        default: return vtm::fail(VTM_EINVAL, "vtm_layernorm: unsupported dtype %d", dtype);
    }
    return vtm::launch_status("vtm_layernorm");
}

// The same LayerNorm with its result written in the k-panel layout [C / 8][panel_rows][8] -- the token operand of the panel
// GEMMs of ff.hip (norm3 -> GEGLU projection, norm2 -> to_q of the cross-attention: patch.py:171-199); rows >= `rows` of
// the panels are not written.
VTM_EXPORT int vtm_layernorm_panels(const void *x, const void *gamma, const void *beta, int dtype, int64_t rows, int64_t C,
                                    float eps, void *out, int64_t panel_rows, vtm_stream_t stream) {
    VTM_REQUIRE(x && out && rows > 0, "vtm_layernorm_panels: null pointer");
    VTM_REQUIRE(C > 0 && C % 8 == 0 && C <= 64 * 8 * MAX_CHUNKS, "vtm_layernorm_panels: C must be a multiple of 8, <= %d",
                64 * 8 * MAX_CHUNKS);
    VTM_REQUIRE(panel_rows >= rows, "vtm_layernorm_panels: panel_rows < rows");
    hipStream_t s = vtm::as_stream(stream);
    switch (dtype) {
        case VTM_F16: launch_layernorm<__half>(x, gamma, beta, rows, C, eps, out, panel_rows, s); break;
        case VTM_BF16: launch_layernorm<vtm_bf16>(x, gamma, beta, rows, C, eps, out, panel_rows, s); break;
        default: return vtm::fail(VTM_EINVAL, "vtm_layernorm_panels: dtype must be VTM_F16 or VTM_BF16");
    }
    return vtm::launch_status("vtm_layernorm_panels");
}
