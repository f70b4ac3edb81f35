// Micro-benchmark: v_mfma_f32_32x32x16_f16 issue rate with 1, 2 or 4 independent accumulators (dependent chains).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    f32x16 c[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) c[j][i] = seed;
    h16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[u % NACC]) : "v"(a), "v"(b));
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += c[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd) {
    const int cus = 256, iters = 2000;
    float *out;
    hipMalloc(&out, sizeof(float) * cus * 8 * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC>), dim3(cus * waves_per_simd), dim3(256), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC>), dim3(cus * waves_per_simd), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("accumulators per wave %d, waves/SIMD %d: %.2f ns per MFMA per SIMD\n", NACC, waves_per_simd,
           ms * 1e6 / ((double)iters * 16 * waves_per_simd));
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) { run<1>(w); run<2>(w); run<4>(w); }
    return 0;
}
