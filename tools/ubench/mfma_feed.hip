// Micro-benchmark: MFMA issue rate when the A fragments come from LDS (as in filter_kernel's group loop).
// 8 accumulators (128 VGPRs), per group 8 ds_read_b128 + 16 MFMAs; 2 workgroups of 4 waves per CU.
// hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/ubench/mfma_feed.hip -o tools/ubench/mfma_feed
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// MODE 0: A from LDS each group; 1: A loaded once (registers only); 2: as 0 plus a global load of B per group
// 3: B via asm loads prefetched 2 groups ahead (counted waits); 4: as 3 plus 2 LDS-DMA pieces per group;
// 5: as 3 plus 2 register-staged pieces per group (global_load -> ds_write_b128 one group later)
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float *out, const uint4 *gsrc, int iters) {
    __shared__ uint4 sA[2][8 * 128];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    for (int i = tid; i < 2 * 8 * 128; i += 256) (&sA[0][0])[i] = make_uint4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    f32x16 acc[4][2];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 2; ++b)
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    uint4 rb[2] = {gsrc[tid], gsrc[tid + 256]};
    h16x8 fh[4], fl[4];
    for (int ib = 0; ib < 4; ++ib) {
        fh[ib] = __builtin_bit_cast(h16x8, sA[0][kh * 128 + ib * 32 + l31]);
        fl[ib] = __builtin_bit_cast(h16x8, sA[1][kh * 128 + ib * 32 + l31]);
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 pb[4][2];
    uint4 stage[2];
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (MODE >= 3) {
        for (int g = 0; g < 2; ++g) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pb[g][0]) : "v"(gsrc + tid));
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pb[g][1]) : "v"(gsrc + tid + 256));
        }
        stage[0] = gsrc[tid]; stage[1] = gsrc[tid + 256];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (MODE >= 3) {
                const uint4 *src = gsrc + ((it * 4 + s) * 512 % 65536) + tid;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pb[(s + 2) & 3][0]) : "v"(src));
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pb[(s + 2) & 3][1]) : "v"(src + 256));
                if constexpr (MODE == 4) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const uint32_t lds_off = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void *)&sA[1][(wave * 2 + t) * 64]);
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_off), "v"(src + 512 * t + lane) : "memory");
                    }
                    asm volatile("s_waitcnt vmcnt(6)" : "+v"(pb[s][0]), "+v"(pb[s][1]));
                } else if constexpr (MODE == 5) {
                    // write the pieces fetched one group ago, fetch the next ones
                    sA[1][(wave * 2 + 0) * 64 + lane] = stage[0];
                    sA[1][(wave * 2 + 1) * 64 + lane] = stage[1];
                    stage[0] = src[512];
                    stage[1] = src[1024];
                    asm volatile("s_waitcnt vmcnt(4)" : "+v"(pb[s][0]), "+v"(pb[s][1]));
                } else {
                    asm volatile("s_waitcnt vmcnt(4)" : "+v"(pb[s][0]), "+v"(pb[s][1]));
                }
                rb[0] = __builtin_bit_cast(uint4, pb[s][0]);
                rb[1] = __builtin_bit_cast(uint4, pb[s][1]);
            }
            if constexpr (MODE == 2) {
                rb[0] = gsrc[(it * 4 + s) * 512 % 65536 + tid];
                rb[1] = gsrc[(it * 4 + s) * 512 % 65536 + tid + 256];
            }
            if constexpr (MODE != 1) {
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {
                    fh[ib] = __builtin_bit_cast(h16x8, sA[0][(s * 2 + kh) * 128 + ib * 32 + l31]);
                    fl[ib] = __builtin_bit_cast(h16x8, sA[1][(s * 2 + kh) * 128 + ib * 32 + l31]);
                }
            }
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                const h16x8 bh = __builtin_bit_cast(h16x8, rb[sb]);
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {
                    f32x16 c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ib], bh, acc[ib][sb], 0, 0, 0);
                    acc[ib][sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[ib], bh, c, 0, 0, 0);
                }
            }
        }
    }
    float sum = 0;
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 2; ++b)
            for (int r = 0; r < 16; ++r) sum += acc[a][b][r];
    out[blockIdx.x * 256 + tid] = sum;
}

template <int MODE>
void run(const char *name) {
    const int blocks = 256 * 2, iters = 2000;
    float *out;
    uint4 *src;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&src, sizeof(uint4) * (65536 + 1024));
    hipMemset(src, 0, sizeof(uint4) * (65536 + 1024));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, src, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, src, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 64 * 2;   // 2 waves per SIMD
    printf("%-40s %.3f ms -> %.2f ns per MFMA per SIMD (13.6 = full rate at 2.35 GHz)\n", name, ms, ms * 1e6 / mfma_per_simd);
    hipFree(out); hipFree(src);
}

int main() {
    run<1>("A in registers (no LDS reads)");
    run<0>("A from LDS every group");
    run<2>("A from LDS + B from global every group");
    run<3>("A from LDS + B prefetched (asm, counted)");
    run<4>("... + 2 LDS-DMA pieces per group");
    run<5>("... + 2 register-staged pieces per group");
    return 0;
}
