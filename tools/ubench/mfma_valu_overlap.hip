// Micro-benchmark: do MFMA and VALU instructions overlap on one SIMD of gfx950 -- inside one wave (interleaved in
// program order) and across waves?  Reports ns per MFMA for an MFMA stream with N independent VALU fillers per MFMA.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o tools/ubench/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// FILL: number of VALU fillers per MFMA; KIND: 0 = v_fma_f32, 1 = v_exp_f32, 2 = v_cvt_pk, 3 = no MFMA at all (VALU only, FILL per slot)
template <int FILL, int KIND, bool ACC = false>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    f32x16 c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) { c0[i] = seed; c1[i] = seed; c2[i] = seed; c3[i] = seed; }
    h16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + threadIdx.x * 1e-3f + i;
    auto fill = [&]() {
#pragma unroll
        for (int f = 0; f < FILL; ++f) {
            if constexpr (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[f & 7]));
            else if constexpr (KIND == 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[f & 7]) : "v"(v[(f + 1) & 7]));
            else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[f & 7]) : "v"(v[(f + 1) & 7]));
        }
    };
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (KIND != 3) { if constexpr (ACC) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b)); }
            fill();
            if constexpr (KIND != 3) { if constexpr (ACC) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b)); }
            fill();
            if constexpr (KIND != 3) { if constexpr (ACC) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c2) : "v"(a), "v"(b)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b)); }
            fill();
            if constexpr (KIND != 3) { if constexpr (ACC) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c3) : "v"(a), "v"(b)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b)); }
            fill();
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// same stream with v_mfma_f32_16x16x32_f16 (4 passes, 16 cycles): TWO of them per slot = the flops of one 32x32x16
template <int FILL>
__global__ __launch_bounds__(256) void k16(float *out, int iters, float seed) {
    f32x4 c[8];
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) c[j][i] = seed;
    h16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[(2 * u) & 7]) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[(2 * u + 1) & 7]) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[f & 7]) : "v"(v[(f + 1) & 7]));
        }
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) s += c[j][i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FILL>
void run16(int waves_per_simd) {
    const int cus = 256, iters = 1000;
    float *out;
    hipMalloc(&out, sizeof(float) * cus * 8 * 256);
    dim3 grid(cus * waves_per_simd), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k16<FILL>), grid, block, 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k16<FILL>), grid, block, 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("2 x mfma 16x16x32 + fma        fill %2d  waves/SIMD %d: %.3f ms -> %.2f ns per slot per SIMD\n", FILL, waves_per_simd, ms,
           ms * 1e6 / ((double)iters * 16 * waves_per_simd));
    hipFree(out);
}

template <int FILL, int KIND, bool ACC = false>
void run(const char *name, int waves_per_simd) {
    const int cus = 256, iters = 1000;
    float *out;
    hipMalloc(&out, sizeof(float) * cus * 8 * 256);
    dim3 grid(cus * waves_per_simd), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<FILL, KIND, ACC>), grid, block, 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<FILL, KIND, ACC>), grid, block, 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double slots = (double)iters * 16 * waves_per_simd;   // MFMA slots per SIMD
    printf("%-28s fill %2d  waves/SIMD %d: %.3f ms -> %.2f ns per slot per SIMD\n", name, FILL, waves_per_simd, ms,
           ms * 1e6 / slots);
    hipFree(out);
}

int main() {
    for (int w : {2, 4}) {
        run<0, 0>("mfma only", w);
        run<2, 0>("mfma + fma", w);
        run<4, 0>("mfma + fma", w);
        run<6, 0>("mfma + fma", w);
        run<8, 0>("mfma + fma", w);
        run<12, 0>("mfma + fma", w);
        run<4, 3>("fma only", w);
        run<8, 3>("fma only", w);
        run<2, 1>("mfma + exp", w);
        run<4, 1>("mfma + exp", w);
        run<4, 2>("mfma + cvt_pk", w);
        run<0, 0, true>("AGPR mfma only", w);
        run<6, 0, true>("AGPR mfma + fma", w);
        run<8, 0, true>("AGPR mfma + fma", w);
        run<12, 0, true>("AGPR mfma + fma", w);
        run<16, 0, true>("AGPR mfma + fma", w);
        run<4, 1, true>("AGPR mfma + exp", w);
        run16<0>(w); run16<4>(w); run16<6>(w); run16<8>(w); run16<12>(w);
    }
    return 0;
}
