// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops of the attention softmax.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
        if constexpr (OP == 0) {   // v_fma_f32
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %2, %2, %1, %1\n v_fma_f32 %3, %3, %1, %1\n v_fma_f32 %4, %4, %1, %1"
                               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4])::);)
        } else if constexpr (OP == 1) {   // v_exp_f32
            REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3"
                               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])::);)
        } else if constexpr (OP == 2) {   // v_exp_f16
            REP16(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3"
                               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])::);)
        } else if constexpr (OP == 3) {   // v_cvt_pk_f16_f32
            REP16(asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %1, %1, %2\n v_cvt_pk_f16_f32 %2, %2, %3\n v_cvt_pk_f16_f32 %3, %3, %0"
                               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])::);)
        } else if constexpr (OP == 4) {   // v_max3_f32
            REP16(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1"
                               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])::);)
        } else if constexpr (OP == 5) {   // v_pk_fma_f32
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %2, %2\n v_pk_fma_f32 %2, %2, %3, %3\n v_pk_fma_f32 %3, %3, %0, %0"
                               : "+v"(*(double *)&a[0]), "+v"(*(double *)&a[2]), "+v"(*(double *)&a[4]), "+v"(*(double *)&a[6])::);)
        } else if constexpr (OP == 6) {   // v_pk_mul_f16 (packed half)
            REP16(asm volatile("v_pk_mul_f16 %0, %0, %1\n v_pk_mul_f16 %1, %1, %2\n v_pk_mul_f16 %2, %2, %3\n v_pk_mul_f16 %3, %3, %0"
                               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])::);)
        } else if constexpr (OP == 7) {   // 1 exp : 3 fma mix (do they overlap?)
            REP16(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4])::);)
        } else if constexpr (OP == 8) {   // v_ldexp_f32
            REP16(asm volatile("v_ldexp_f32 %0, %0, %1\n v_ldexp_f32 %1, %1, %2\n v_ldexp_f32 %2, %2, %3\n v_ldexp_f32 %3, %3, %0"
                               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])::);)
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, int waves_per_simd) {
    const int cus = 256, iters = 2000;
    float *out;
    hipMalloc(&out, sizeof(float) * cus * 4 * 64 * 8);
    dim3 grid(cus * waves_per_simd), block(256);   // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 64;
    const double ns_per_instr_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
    printf("%-22s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles @2.4 GHz)\n", name,
           waves_per_simd, ms, ns_per_instr_simd, ns_per_instr_simd * 2.4);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w);
        run<1>("v_exp_f32", w);
        run<2>("v_exp_f16", w);
        run<3>("v_cvt_pk_f16_f32", w);
        run<4>("v_max3_f32", w);
        run<5>("v_pk_fma_f32", w);
        run<6>("v_pk_mul_f16", w);
        run<7>("1 exp + 3 fma", w);
        run<8>("v_ldexp_f32", w);
    }
    return 0;
}
