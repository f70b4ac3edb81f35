// vtm_sort_desc: stable descending argsort of the row maxima.
// Reference: `edge_idx = node_max.argsort(dim=-1, descending=True)` (vidtome/merge.py:98,113 / 402,417).
// Canonical order (see include/vidtome_hip.h): descending value, NaN first, ties by ascending index.
// The sort key is the high word of vtm_match's packed result (already an order-preserving integer
// image of node_max), inverted so that an ASCENDING stable LSD radix sort yields the DESCENDING order.
//
// Design (v4): multi-workgroup LSD radix sort, 4 passes of 8-bit digits, every global access coalesced.
// A row is cut into contiguous segments (one 256-thread workgroup each, a few 256-key tiles per segment),
// so that ~64 CUs work on a row instead of one.  Kernels:
//   hist     (first pass only) per-segment digit counts (wave-aggregated: one LDS add per distinct digit and
//            wave) -> hist[0][row][seg][digit]; also clears the tables of the later passes
//   scatter  every workgroup derives its 256 scatter offsets from the pass's table (exclusive scan in (digit,
//            segment) order), then scatters stably: offset + earlier tiles of the segment + lower waves of the
//            tile + rank inside the wave (popcount of the same-digit lane mask below the lane).  A key's
//            destination tells which segment it belongs to in the NEXT pass, so the scatter also counts the next
//            digit into the next pass's table (one global atomic per key; counts do not depend on order)
// Thread order == index order inside a tile, tiles and segments are in index order, so equal digits keep
// their order (stability).  The kernels are launch-latency sized (rows are <= ~110k keys, L2-resident), which
// is why the digit is 8 bits wide, the scan lives inside the scatter kernel and the later histograms ride on the
// scatters: 5 launches per sort (a 4-bit digit with separate hist / scan / scatter kernels needs 24).
#include "common.h"

namespace {

constexpr int T = 256;       // threads per workgroup = keys per tile
constexpr int WAVES = T / 64;
constexpr int RADIX = 256;   // 8-bit digits (== T: thread d owns digit d in the per-digit steps)
constexpr int PASSES = 4;
static_assert(RADIX == T, "thread d handles digit d");

// mask of the lanes of this wave that hold the same 8-bit digit as this lane (only lanes in `valid`)
__device__ __forceinline__ unsigned long long same_digit_lanes(int d, unsigned long long valid) {
    unsigned long long m = valid;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const unsigned long long bal = __ballot((d >> b) & 1);
        m &= ((d >> b) & 1) ? bal : ~bal;
    }
    return m;
}

__device__ __forceinline__ uint32_t load_key(const uint64_t *kin, const uint32_t *ksrc, int pass, int64_t i) {
    return pass == 0 ? ~(uint32_t)(kin[i] >> 32) : ksrc[i];
}

struct Geo {
    int64_t n;
    int nseg, tiles_per_seg;
};

__global__ __launch_bounds__(T) void sort_hist_kernel(const uint64_t *__restrict__ best, Geo g, int64_t table_ints,
                                                      int *__restrict__ hist) {
    __shared__ int cnt[RADIX];
    const int tid = threadIdx.x, lane = tid & 63;
    const int seg = blockIdx.x, row = blockIdx.y;
    const uint64_t *kin = best + (int64_t)row * g.n;
    const uint32_t *ksrc = nullptr;
    constexpr int pass = 0, sh = 0;
    cnt[tid] = 0;
#pragma unroll
    for (int p = 1; p < PASSES; ++p)   // the tables the scatters of passes 0 .. PASSES-2 count into
        hist[p * table_ints + ((int64_t)row * g.nseg + seg) * RADIX + tid] = 0;
    __syncthreads();
    for (int t = 0; t < g.tiles_per_seg; ++t) {
        const int64_t i = ((int64_t)seg * g.tiles_per_seg + t) * T + tid;
        const bool ok = i < g.n;
        const int d = ok ? (int)((load_key(kin, ksrc, pass, i) >> sh) & 255u) : 0;
        const unsigned long long m = same_digit_lanes(d, __ballot(ok));
        // the lowest lane of every digit group adds the group's size (no two lanes of a wave hit one address)
        if (ok && (m & ((1ull << lane) - 1ull)) == 0) atomicAdd(&cnt[d], __popcll(m));
    }
    __syncthreads();
    hist[((int64_t)row * g.nseg + seg) * RADIX + tid] = cnt[tid];
}

__global__ __launch_bounds__(T) void sort_scatter_kernel(const uint64_t *__restrict__ best,
                                                         const uint32_t *__restrict__ ksrc_all,
                                                         const int32_t *__restrict__ psrc_all,
                                                         uint32_t *__restrict__ kdst_all, int32_t *__restrict__ pdst_all,
                                                         Geo g, int pass, int last, const int *__restrict__ hist,
                                                         int *__restrict__ hist_next) {
    // wcnt[w][d] = (tile tag << 16) | keys of digit d in wave w of the tagged tile; an entry with another tag
    // counts as zero, so the table never has to be cleared
    __shared__ uint32_t wcnt[WAVES][RADIX];
    __shared__ int running[RADIX];   // scatter offset of (digit, this segment) + keys of the digit already scattered
    __shared__ int part[RADIX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg = blockIdx.x, row = blockIdx.y;
    const uint64_t *kin = best + (int64_t)row * g.n;
    const uint32_t *ksrc = ksrc_all + (int64_t)row * g.n;
    const int32_t *psrc = psrc_all + (int64_t)row * g.n;
    uint32_t *kdst = kdst_all + (int64_t)row * g.n;
    int32_t *pdst = pdst_all + (int64_t)row * g.n;
    const int sh = pass * 8;
    const int seg_keys = g.tiles_per_seg * T;
    int *hnext = hist_next + (int64_t)row * g.nseg * RADIX;
    const unsigned long long below = (1ull << lane) - 1ull;
    // scatter offsets of this segment, straight from the histogram table (exclusive scan in (digit, segment)
    // order): thread d owns digit d -- keys of smaller digits in all segments + keys of digit d in earlier
    // segments.  The table is [seg][digit], so the loads are coalesced and independent.
    {
        const int *h = hist + (int64_t)row * g.nseg * RADIX + tid;
        int tot = 0, bef = 0;
#pragma unroll 8
        for (int e = 0; e < g.nseg; ++e) {
            const int c = h[e * RADIX];
            tot += c;
            bef += e < seg ? c : 0;
        }
        part[tid] = tot;
        __syncthreads();
        for (int off = 1; off < T; off <<= 1) {   // Hillis-Steele inclusive scan over the 256 digit totals
            const int v = tid >= off ? part[tid - off] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        running[tid] = part[tid] - tot + bef;
    }
#pragma unroll
    for (int w = 0; w < WAVES; ++w) wcnt[w][tid] = 0xffff0000u;   // tag no tile ever has
    __syncthreads();
    for (int t = 0; t < g.tiles_per_seg; ++t) {
        const int64_t i = ((int64_t)seg * g.tiles_per_seg + t) * T + tid;
        const bool ok = i < g.n;
        uint32_t key = 0;
        int32_t id = 0;
        int d = 0;
        if (ok) {
            key = load_key(kin, ksrc, pass, i);
            id = pass == 0 ? (int32_t)i : psrc[i];
            d = (int)((key >> sh) & 255u);
        }
        const unsigned long long m = same_digit_lanes(d, __ballot(ok));
        const uint32_t tag = (uint32_t)t << 16;
        if (ok && (m & below) == 0) wcnt[wave][d] = tag | (uint32_t)__popcll(m);
        __syncthreads();
        if (ok) {
            int pos = running[d] + __popcll(m & below);
#pragma unroll
            for (int w = 0; w < WAVES - 1; ++w) {
                const uint32_t e = wcnt[w][d];
                if (w < wave && (e & 0xffff0000u) == tag) pos += (int)(e & 0xffffu);
            }
            if (!last) {
                kdst[pos] = key;
                atomicAdd(&hnext[(pos / seg_keys) * RADIX + (int)((key >> (sh + 8)) & 255u)], 1);
            }
            pdst[pos] = id;
        }
        __syncthreads();   // everyone has read running[] / wcnt[] of this tile
        {
            int tot = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const uint32_t e = wcnt[w][tid];
                tot += (e & 0xffff0000u) == tag ? (int)(e & 0xffffu) : 0;
            }
            running[tid] += tot;
        }
        __syncthreads();
    }
}

inline Geo make_geo(int64_t n) {
    Geo g;
    g.n = n;
    const int64_t tiles = vtm::cdiv(n, T);
    int64_t tps = vtm::cdiv(tiles, 64);   // aim for ~64 segments per row
    if (tps < 1) tps = 1;
    g.tiles_per_seg = (int)tps;
    g.nseg = (int)vtm::cdiv(tiles, tps);
    return g;
}

inline size_t aligned(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

VTM_EXPORT size_t vtm_sort_ws_bytes(int64_t rows, int64_t n) {
    if (rows <= 0 || n <= 0) return 0;
    const Geo g = make_geo(n);
    // key / index ping-pong buffers + one histogram table per pass
    return aligned((size_t)rows * n * 16u) + aligned((size_t)PASSES * rows * RADIX * g.nseg * sizeof(int));
}

VTM_EXPORT int vtm_sort_desc(const uint64_t *best, int64_t rows, int64_t n, int32_t *perm, void *ws,
                             size_t ws_bytes, vtm_stream_t stream) {
    VTM_REQUIRE(best && perm && ws, "vtm_sort_desc: null pointer");
    VTM_REQUIRE(rows > 0 && rows <= 65535 && n >= 0 && n < (1ll << 31), "vtm_sort_desc: bad sizes");
    if (ws_bytes < vtm_sort_ws_bytes(rows, n))
        return vtm::fail(VTM_EWORKSPACE, "vtm_sort_desc: workspace %zu < %zu bytes", ws_bytes,
                         vtm_sort_ws_bytes(rows, n));
    if (n == 0) return VTM_OK;
    hipStream_t s = vtm::as_stream(stream);
    const Geo g = make_geo(n);
    VTM_REQUIRE(g.tiles_per_seg < 0xffff, "vtm_sort_desc: row too long for the tile tag");
    char *w = static_cast<char *>(ws);
    uint32_t *k0 = reinterpret_cast<uint32_t *>(w), *k1 = k0 + rows * n;
    int32_t *p0 = reinterpret_cast<int32_t *>(k1 + rows * n), *p1 = p0 + rows * n;
    int *hist = reinterpret_cast<int *>(w + aligned((size_t)rows * n * 16u));
    const dim3 grid((unsigned)g.nseg, (unsigned)rows), block(T);
    const int64_t table_ints = rows * g.nseg * RADIX;
    hipLaunchKernelGGL(sort_hist_kernel, grid, block, 0, s, best, g, table_ints, hist);
    for (int pass = 0; pass < PASSES; ++pass) {
        const uint32_t *ksrc = (pass & 1) ? k1 : k0;
        const int32_t *psrc = (pass & 1) ? p1 : p0;
        uint32_t *kdst = (pass & 1) ? k0 : k1;
        int32_t *pdst = (pass & 1) ? p0 : p1;
        const int last = pass == PASSES - 1;
        if (last) pdst = perm;
        hipLaunchKernelGGL(sort_scatter_kernel, grid, block, 0, s, best, ksrc, psrc, kdst, pdst, g, pass, last,
                           hist + pass * table_ints, hist + (last ? 0 : pass + 1) * table_ints);
    }
    return vtm::launch_status("vtm_sort_desc");
}
