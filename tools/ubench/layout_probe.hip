// Probes two gfx950 facts the 16-row PV path of attention.hip relies on:
//   1. v_permlane16_swap_b32 (x, y): rows of 16 lanes, x = [x0 x1 x2 x3], y = [y0 y1 y2 y3]
//      -> x' = [x0 y0 x2 y2], y' = [x1 y1 x3 y3]
//   2. v_mfma_f32_16x16x32_f16 operand layout: A lane l = row l%16, k = 8 (l/16) + 0..7; B lane l = col l%16,
//      same k; C lane l = col l%16, rows 4 (l/16) + 0..3
// build: hipcc --offload-arch=gfx950 -O3 -o layout_probe layout_probe.hip ; prints "ok" twice
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void swap_probe(unsigned *o) {
    unsigned x = threadIdx.x, y = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}

__global__ void mfma_probe(float *c, const _Float16 *a, const _Float16 *b) {   // a: 16x32 row-major, b: 32x16 row-major
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    h16x8 fa, fb;
    for (int e = 0; e < 8; ++e) {
        fa[e] = a[i * 32 + 8 * g + e];
        fb[e] = b[(8 * g + e) * 16 + i];
    }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc, 0, 0, 0);
    for (int e = 0; e < 4; ++e) c[(4 * g + e) * 16 + i] = acc[e];
}

int main() {
    unsigned *o;
    hipMalloc(&o, 128 * 4);
    swap_probe<<<1, 64>>>(o);
    unsigned h[128];
    hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
    bool ok = true;
    for (int l = 0; l < 64; ++l) {
        const int row = l / 16, j = l % 16;
        const unsigned ex = (row & 1) ? 100 + 16 * (row - 1) + j : l;          // x' = [x0 y0 x2 y2]
        const unsigned ey = (row & 1) ? 100 + l : 16 * (row + 1) + j;          // y' = [x1 y1 x3 y3]
        if (h[l] != ex || h[64 + l] != ey) ok = false;
    }
    printf("permlane16_swap: %s\n", ok ? "ok" : "MISMATCH");
    if (!ok) {
        for (int l = 0; l < 64; l += 16) printf("  row %d: x' %u.. y' %u..\n", l / 16, h[l], h[64 + l]);
    }
    std::vector<_Float16> a(16 * 32), b(32 * 16);
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < 32; ++k) a[i * 32 + k] = (_Float16)((i * 7 + k * 3) % 11 - 5);
    for (int k = 0; k < 32; ++k)
        for (int j = 0; j < 16; ++j) b[k * 16 + j] = (_Float16)((k * 5 + j * 2) % 13 - 6);
    _Float16 *da, *db;
    float *dc;
    hipMalloc(&da, a.size() * 2);
    hipMalloc(&db, b.size() * 2);
    hipMalloc(&dc, 256 * 4);
    hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
    mfma_probe<<<1, 64>>>(dc, da, db);
    float c[256];
    hipMemcpy(c, dc, sizeof c, hipMemcpyDeviceToHost);
    ok = true;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float s = 0;
            for (int k = 0; k < 32; ++k) s += (float)a[i * 32 + k] * (float)b[k * 16 + j];
            if (c[i * 16 + j] != s) ok = false;
        }
    printf("mfma_f32_16x16x32_f16 layout: %s\n", ok ? "ok" : "MISMATCH");
    return 0;
}
