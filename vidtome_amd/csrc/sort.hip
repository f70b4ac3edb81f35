// vtm_sort_desc: stable descending argsort of the row maxima.
// Reference: `edge_idx = node_max.argsort(dim=-1, descending=True)` (vidtome/merge.py:98,113 / 402,417).
// Canonical order (see include/vidtome_hip.h): descending value, NaN first, ties by ascending index.
// The sort key is the high word of vtm_match's packed result (already an order-preserving integer
// image of node_max), inverted so that an ASCENDING stable LSD radix sort yields the DESCENDING order.
//
// Design (v2): one 1024-thread workgroup per row of keys, 8 passes of 4-bit digits, every global access
// coalesced.  A pass walks the row in tiles of 1024 consecutive keys (one per thread, so thread order ==
// index order inside a tile and tile order == index order across tiles):
//   sweep A  digit histogram from 4 per-bit wave ballots per tile, lane v of every wave accumulates digit v;
//   sweep B  stable scatter: rank inside the wave = popcount(ballot(my digit) & lanes-below), plus the
//            counts of the lower waves of the tile (16 x 16 table in LDS), plus the running per-digit count
//            of the earlier tiles, plus the digit's global base.
// Rows are <= ~110k keys (L2-resident state); the kernel is latency-bound, ~0.1 ms for 49k keys.
#include "common.h"

namespace {

constexpr int T = 1024;      // threads per workgroup
constexpr int WAVES = T / 64;
constexpr int RADIX = 16;    // 4-bit digits
constexpr int PASSES = 8;

// mask of the lanes of this wave whose 4-bit digit equals dv, from the 4 per-bit ballots (4 ballots + a few
// 64-bit xors instead of 16 ballots)
__device__ __forceinline__ unsigned long long same_digit(const unsigned long long (&mb)[4], unsigned long long valid,
                                                         int dv) {
    unsigned long long m = valid;
#pragma unroll
    for (int b = 0; b < 4; ++b) m &= ((dv >> b) & 1) ? mb[b] : ~mb[b];
    return m;
}

__global__ __launch_bounds__(T) void sort_desc_kernel(const uint64_t *__restrict__ best, int64_t n,
                                                      int32_t *__restrict__ perm,
                                                      uint32_t *__restrict__ ws) {
    __shared__ int wcnt[WAVES][RADIX];   // per-wave digit counts (sweep A totals / per-tile counts)
    __shared__ int wpre[WAVES][RADIX];   // exclusive prefix over the waves of the current tile
    __shared__ int base[RADIX];          // global exclusive digit offsets of the pass
    __shared__ int running[RADIX];       // keys of each digit in the tiles already scattered
    __shared__ int ttot[RADIX];          // digit totals of the current tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    const uint64_t *kin = best + row * n;
    uint32_t *k0 = ws + row * 4 * n, *k1 = k0 + n;
    int32_t *p0 = reinterpret_cast<int32_t *>(k1 + n), *p1 = p0 + n;
    const int64_t ntiles = (n + T - 1) / T;
    const unsigned long long below = (1ull << lane) - 1ull;

    for (int pass = 0; pass < PASSES; ++pass) {
        const int sh = pass * 4;
        const uint32_t *ksrc = (pass & 1) ? k1 : k0;
        const int32_t *psrc = (pass & 1) ? p1 : p0;
        uint32_t *kdst = (pass & 1) ? k0 : k1;
        int32_t *pdst = (pass & 1) ? p0 : p1;
        if (pass == PASSES - 1) pdst = perm + row * n;   // last scatter writes the result directly

        // ---- sweep A: histogram (lane v < 16 of every wave accumulates digit v)
        int mycount = 0;
        for (int64_t t = 0; t < ntiles; ++t) {
            const int64_t i = t * T + tid;
            int d = 0;
            if (i < n) d = ((pass == 0 ? ~(uint32_t)(kin[i] >> 32) : ksrc[i]) >> sh) & 15u;
            unsigned long long mb[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) mb[b] = __ballot((d >> b) & 1);
            const unsigned long long valid = __ballot(i < n);
            if (lane < RADIX) mycount += __popcll(same_digit(mb, valid, lane));
        }
        if (lane < RADIX) wcnt[wave][lane] = mycount;
        __syncthreads();
        if (tid < RADIX) {
            int tot = 0;
            for (int w = 0; w < WAVES; ++w) tot += wcnt[w][tid];
            wpre[0][tid] = tot;   // scratch: digit totals
        }
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            for (int v = 0; v < RADIX; ++v) {
                base[v] = acc;
                acc += wpre[0][v];
                running[v] = 0;
            }
        }
        __syncthreads();

        // ---- sweep B: stable scatter, tile by tile
        for (int64_t t = 0; t < ntiles; ++t) {
            const int64_t i = t * T + tid;
            uint32_t key = 0;
            int32_t id = 0;
            int d = 0;
            if (i < n) {
                if (pass == 0) {
                    key = ~(uint32_t)(kin[i] >> 32);
                    id = (int32_t)i;
                } else {
                    key = ksrc[i];
                    id = psrc[i];
                }
                d = (key >> sh) & 15u;
            }
            unsigned long long mb[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) mb[b] = __ballot((d >> b) & 1);
            const unsigned long long valid = __ballot(i < n);
            const int rank = __popcll(same_digit(mb, valid, d) & below);
            if (lane < RADIX) wcnt[wave][lane] = __popcll(same_digit(mb, valid, lane));
            __syncthreads();
            if (tid < RADIX * WAVES) {   // thread (w, v): exclusive prefix of digit v over waves < w
                const int w = tid >> 4, v = tid & 15;
                int acc = 0;
                for (int ww = 0; ww < w; ++ww) acc += wcnt[ww][v];
                wpre[w][v] = acc;
                if (w == WAVES - 1) ttot[v] = acc + wcnt[w][v];
            }
            __syncthreads();
            if (i < n) {
                const int pos = base[d] + running[d] + wpre[wave][d] + rank;
                if (pass != PASSES - 1) kdst[pos] = key;
                pdst[pos] = id;
            }
            __syncthreads();   // everyone has read running[] / wpre[] before they change
            if (tid < RADIX) running[tid] += ttot[tid];
            // (the next tile's first barrier orders this update before anyone reads running[] again)
        }
        __syncthreads();   // workgroup-scope: the next pass reads what this workgroup just wrote
    }
}

}  // namespace

VTM_EXPORT size_t vtm_sort_ws_bytes(int64_t rows, int64_t n) {
    if (rows <= 0 || n <= 0) return 0;
    return (size_t)rows * (size_t)n * 16u;
}

VTM_EXPORT int vtm_sort_desc(const uint64_t *best, int64_t rows, int64_t n, int32_t *perm, void *ws,
                             size_t ws_bytes, vtm_stream_t stream) {
    VTM_REQUIRE(best && perm && ws, "vtm_sort_desc: null pointer");
    VTM_REQUIRE(rows > 0 && n >= 0 && n < (1ll << 31), "vtm_sort_desc: bad sizes");
    if (ws_bytes < vtm_sort_ws_bytes(rows, n))
        return vtm::fail(VTM_EWORKSPACE, "vtm_sort_desc: workspace %zu < %zu bytes", ws_bytes,
                         vtm_sort_ws_bytes(rows, n));
    if (n == 0) return VTM_OK;
    hipLaunchKernelGGL(sort_desc_kernel, dim3((unsigned)rows), dim3(T), 0, vtm::as_stream(stream), best, n,
                       perm, reinterpret_cast<uint32_t *>(ws));
    return vtm::launch_status("vtm_sort_desc");
}
