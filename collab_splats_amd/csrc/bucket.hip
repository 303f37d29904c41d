// bucket.hip -- cell-ordered bucketing of the (tile, Gaussian) intersections: the default ordering
// ("cells") of the binning stage.  gfx950 only.  SURVEY.md section 8 row a2.3 (inside gsplat-rade's
// rasterization(), called at /root/reference/collab_splats/models/rade_gs_model.py:439-465).
//
// What has to be produced: for every tile the list of Gaussian rows whose screen rectangle touches it
// (then sorted by depth, misplat_tile_sort).  That is a counting sort of I ~ 6.4 M items into ~8 160 bins.
// gsplat (and round 1 of this build) emits (tile, row) pairs and radix-sorts them: every pair crosses HBM
// 5+ times.  Here every pair is written ONCE (4 bytes: the row) and no tile-id array exists at all:
//
//   1. bucket_count   per row: tile rectangle, count, and the coarse screen CELL (4x4 tiles or larger) of the
//                     rectangle's centre; per-workgroup LDS histogram of the cells, one global atomic per
//                     (workgroup, cell)
//   2. bucket_rows    scan of the cell counts, then the visible rows are scattered into CELL ORDER
//                     (order[]): rows that are neighbours in order[] are neighbours on screen
//   3. bucket_tiles   a workgroup of 1024 consecutive rows of order[] touches a compact window of ~100 tiles:
//                     it counts its intersections per tile in an LDS table and adds ONE global atomic per
//                     (workgroup, tile) -> tile counts -> scan -> offsets; the same walk a second time
//                     reserves a range per (workgroup, tile) with one returning atomic and fills
//                     payload[offsets[tile] + ...] = row, ~50 consecutive entries at a time.
//
// Buckets fill in an arbitrary order, so misplat_tile_sort(unordered = 1) establishes the (depth, row)
// order inside each: the final lists are bit-identical to the (tile, depth, id) order of a 64-bit key sort.
// Every count lives on the DEVICE (counters[0] = number of intersections, counters[1] = visible rows):
// nothing here needs a host read-back, so the whole forward can be enqueued speculatively (capacity
// `cap_isects`, overflow checked by the host afterwards) and captured in a hipGraph.
// HBM-bound integer work: no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "misplat.h"
#include "internal.h"

namespace {

constexpr int kCountThreads = 1024;     // few, large counting workgroups: one global atomic per (workgroup, cell)
constexpr int kRowsPerWg = 1024;        // rows of order[] per workgroup of the tile passes (one per thread)
constexpr int kWinMax = 2560;           // LDS window: (width + 1) * (height + 1) <= this
constexpr int kOwnerChunk = 8 * kRowsPerWg;     // outputs whose owners are resolved at once (8 per thread)
constexpr int kSmallRect = 16;          // rectangles up to 16 x 16 tiles take part in the window

struct CellGrid {
    int shift, cells_x, cells_y, n_cells, n_blocks, rows_per_block;
};

inline CellGrid make_grid(const misplat_params* p) {
    CellGrid g;
    g.shift = 2;                                               // 4 x 4 tiles = 64 x 64 pixels
    for (;;) {
        g.cells_x = (p->tile_w + (1 << g.shift) - 1) >> g.shift;
        g.cells_y = (p->tile_h + (1 << g.shift) - 1) >> g.shift;
        g.n_cells = g.cells_x * g.cells_y * p->n_cams;
        if (g.n_cells <= MISPLAT_BUCKET_MAX_CELLS || g.shift >= 12) break;
        g.shift++;
    }
    const int64_t total = (int64_t)p->n_gauss * p->n_cams;
    int64_t nb = (total + 4095) / 4096;                        // >= 4096 rows per workgroup: few, long histograms
    if (nb > MISPLAT_BUCKET_MAX_BLOCKS) nb = MISPLAT_BUCKET_MAX_BLOCKS;
    if (nb < 1) nb = 1;
    g.n_blocks = (int)nb;
    int64_t rpb = (total + nb - 1) / nb;
    rpb = (rpb + kCountThreads - 1) / kCountThreads * kCountThreads;
    g.rows_per_block = (int)rpb;
    return g;
}

__device__ __forceinline__ void tile_rect(float mx, float my, int rxi, int ryi, int tw, int th,
                                          int& x0, int& x1, int& y0, int& y1) {
    // the fixed fp32 expression sequence the CPU restatement of the tests uses too (bit-exact rectangles)
    const float ts = (float)MISPLAT_TILE;
    float rx = (float)rxi, ry = (float)ryi;
    float fx0 = floorf((mx - rx) / ts), fx1 = ceilf((mx + rx) / ts);
    float fy0 = floorf((my - ry) / ts), fy1 = ceilf((my + ry) / ts);
    float ftw = (float)tw, fth = (float)th;
    fx0 = fx0 > 0.f ? fx0 : 0.f; fx1 = fx1 > 0.f ? fx1 : 0.f;
    fy0 = fy0 > 0.f ? fy0 : 0.f; fy1 = fy1 > 0.f ? fy1 : 0.f;
    x0 = (int)(fx0 < ftw ? fx0 : ftw); x1 = (int)(fx1 < ftw ? fx1 : ftw);
    y0 = (int)(fy0 < fth ? fy0 : fth); y1 = (int)(fy1 < fth ? fy1 : fth);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_take(uint32_t ident, uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)ident, (int)x, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t x) {      // inclusive; lane 63 holds the total
    x += dpp_take<0x111, 0xF>(0u, x);
    x += dpp_take<0x112, 0xF>(0u, x);
    x += dpp_take<0x114, 0xF>(0u, x);
    x += dpp_take<0x118, 0xF>(0u, x);
    x += dpp_take<0x142, 0xA>(0u, x);
    x += dpp_take<0x143, 0xC>(0u, x);
    return x;
}

__device__ __forceinline__ int cell_of(uint32_t xy, uint32_t wh, int cam, int shift, int cells_x, int cells_per_cam) {
    const int cx = (int)(((xy & 0xffffu) + ((wh & 0xffffu) >> 1)) >> shift);
    const int cy = (int)(((xy >> 16) + ((wh >> 16) >> 1)) >> shift);
    return cam * cells_per_cam + cy * cells_x + cx;
}

// ---- 1. per row: rectangle + count; per workgroup: cell histogram (kept per workgroup in cellhist[b][c] and added
// to the global cell counts with one atomic per non-empty cell) and the sum of the counts ------------------------
__global__ __launch_bounds__(kCountThreads) void bucket_count_kernel(
    int64_t total, int n_gauss, int tw, int th, int shift, int cells_x, int cells_per_cam, int n_cells, int rows_per_block,
    const float* __restrict__ means2d, const int32_t* __restrict__ radii, int32_t* __restrict__ tiles_per_gauss,
    uint2* __restrict__ rect2, uint32_t* __restrict__ cellhist, uint32_t* __restrict__ cell_count,
    unsigned long long* __restrict__ n_isects) {
    __shared__ uint32_t hist[MISPLAT_BUCKET_MAX_CELLS];
    __shared__ unsigned long long wsum[kCountThreads / 64];
    for (int c = threadIdx.x; c < n_cells; c += kCountThreads) hist[c] = 0u;
    __syncthreads();
    const int64_t beg = (int64_t)blockIdx.x * rows_per_block;
    const int64_t end = beg + rows_per_block < total ? beg + rows_per_block : total;
    unsigned long long mine = 0ull;
    for (int64_t idx = beg + threadIdx.x; idx < end; idx += kCountThreads) {
        const int rx = radii[2 * idx], ry = radii[2 * idx + 1];
        int n = 0;
        uint2 r2 = make_uint2(0u, 0u);
        if (rx > 0 || ry > 0) {
            int x0, x1, y0, y1;
            tile_rect(means2d[2 * idx], means2d[2 * idx + 1], rx, ry, tw, th, x0, x1, y0, y1);
            n = (x1 - x0) * (y1 - y0);
            if (n > 0) {
                r2 = make_uint2((uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)(x1 - x0) | ((uint32_t)(y1 - y0) << 16));
                atomicAdd(&hist[cell_of(r2.x, r2.y, (int)(idx / n_gauss), shift, cells_x, cells_per_cam)], 1u);
            }
        }
        tiles_per_gauss[idx] = n;
        rect2[idx] = r2;
        mine += (unsigned long long)n;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mine += __shfl_xor(mine, m);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0ull;
#pragma unroll
        for (int w = 0; w < kCountThreads / 64; w++) t += wsum[w];
        if (t) atomicAdd(n_isects, t);
    }
    for (int c = threadIdx.x; c < n_cells; c += kCountThreads) {
        const uint32_t h = hist[c];
        cellhist[(size_t)blockIdx.x * n_cells + c] = h;
        if (h) atomicAdd(&cell_count[c], h);
    }
}

// ---- 2. visible rows -> cell order (the slices of pass 1 again).  Every workgroup scans the (at most 2 048) global cell
// counts itself -- two per thread, a few hundred instructions, instead of a one-workgroup launch in front of this kernel --
// and reserves its range of each cell with one returning atomic on a cursor array.  Workgroup 0 also publishes what the
// host and the later kernels need: cell_offs, counters[1] = visible rows, and the intersection count as a system-scope
// store into the caller's pinned slot (no copy node in the stream).
// Round 4: the rows leave through LDS IN CELL ORDER.  A thread used to store its row at the next free position of its
// cell as it met it: a wave's 64 stores went to ~64 different cache lines, three partial-line streams (order, rect_sorted,
// depth_sorted) of 4 - 8 bytes each -- measured at 5 M Gaussians, every such scattered 4-byte stream costs ~100 us (a
// random gather of the same elements costs the same: the fabric moves whole lines).  Now a workgroup takes its rows in
// chunks of 16 384: ranks them inside (chunk, cell) with an LDS counter, scans the chunk's cell counts, lays the chunk out
// by cell in LDS, and writes it from there -- consecutive threads write consecutive positions of a cell's range.
constexpr int kRowsChunk = 16 * kCountThreads;
constexpr uint32_t kRowNone = 0xffffffffu;
__global__ __launch_bounds__(kCountThreads) void bucket_rows_kernel(
    int64_t total, int n_gauss, int shift, int cells_x, int cells_per_cam, int n_cells, int rows_per_block,
    const int32_t* __restrict__ tiles_per_gauss, const uint2* __restrict__ rect2, const uint32_t* __restrict__ cellhist,
    const uint32_t* __restrict__ cell_count, uint32_t* __restrict__ cell_offs, uint32_t* __restrict__ cell_cursor,
    int32_t* __restrict__ order, uint2* __restrict__ rect_sorted, int64_t* __restrict__ counters,
    long long* __restrict__ n_isects_host, const float* __restrict__ depths, float* __restrict__ depth_sorted) {
    __shared__ uint32_t base[MISPLAT_BUCKET_MAX_CELLS];          // global position of this workgroup's next row of the cell
    __shared__ uint32_t cnt[MISPLAT_BUCKET_MAX_CELLS];           // rows of the cell in the current chunk
    __shared__ uint32_t start[MISPLAT_BUCKET_MAX_CELLS];         // their first slot in sorted[]
    __shared__ uint32_t sorted[kRowsChunk];                      // the chunk in cell order: row in chunk | cell << 14
    __shared__ uint32_t wsum[kCountThreads / 64];
    static_assert(2 * kCountThreads >= MISPLAT_BUCKET_MAX_CELLS, "two cells per thread");
    static_assert(kRowsChunk <= (1 << 14) && MISPLAT_BUCKET_MAX_CELLS <= (1 << 11), "packing of sorted[]");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = 2 * threadIdx.x, c1 = c0 + 1;
    {
        const uint32_t a = c0 < n_cells ? cell_count[c0] : 0u, b2 = c1 < n_cells ? cell_count[c1] : 0u;
        const uint32_t incl = wave_scan_add(a + b2);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t carry = 0u, all = 0u;
#pragma unroll
        for (int w = 0; w < kCountThreads / 64; w++) { carry += (w < wave) ? wsum[w] : 0u; all += wsum[w]; }
        const uint32_t e = carry + incl - (a + b2);
        if (c0 < n_cells) base[c0] = e;
        if (c1 < n_cells) base[c1] = e + a;
        if (blockIdx.x == 0) {
            if (c0 < n_cells) cell_offs[c0] = e;
            if (c1 < n_cells) cell_offs[c1] = e + a;
            if (threadIdx.x == 0) {
                cell_offs[n_cells] = all;
                counters[1] = (int64_t)all;
                if (n_isects_host) __hip_atomic_store(n_isects_host, (long long)counters[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < n_cells; c += kCountThreads) {
        const uint32_t h = cellhist[(size_t)blockIdx.x * n_cells + c];
        base[c] = h ? base[c] + atomicAdd(&cell_cursor[c], h) : 0u;
    }
    const int64_t beg = (int64_t)blockIdx.x * rows_per_block;
    const int64_t end = beg + rows_per_block < total ? beg + rows_per_block : total;
    for (int64_t cb = beg; cb < end; cb += kRowsChunk) {
        for (int c = threadIdx.x; c < n_cells; c += kCountThreads) cnt[c] = 0u;
        __syncthreads();                                         // (also: base[] of the prologue / the previous chunk)
        // rank of every visible row inside its (chunk, cell)
        uint32_t pk[kRowsChunk / kCountThreads];                 // cell | rank << 11, or none
        int32_t tpg[kRowsChunk / kCountThreads];
#pragma unroll
        for (int u = 0; u < kRowsChunk / kCountThreads; u++) {
            const int64_t idx = cb + u * kCountThreads + threadIdx.x;
            tpg[u] = idx < end ? tiles_per_gauss[idx] : 0;
        }
#pragma unroll
        for (int u = 0; u < kRowsChunk / kCountThreads; u++) {
            pk[u] = kRowNone;
            if (tpg[u] > 0) {
                const int64_t idx = cb + u * kCountThreads + threadIdx.x;
                const uint2 r2 = rect2[idx];
                const int c = cell_of(r2.x, r2.y, (int)(idx / n_gauss), shift, cells_x, cells_per_cam);
                pk[u] = (uint32_t)c | (atomicAdd(&cnt[c], 1u) << 11);
            }
        }
        __syncthreads();
        // exclusive scan of the chunk's cell counts (two cells per thread)
        uint32_t n_chunk;
        {
            const uint32_t a = c0 < n_cells ? cnt[c0] : 0u, b2 = c1 < n_cells ? cnt[c1] : 0u;
            const uint32_t incl = wave_scan_add(a + b2);
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            uint32_t carry = 0u;
            n_chunk = 0u;
#pragma unroll
            for (int w = 0; w < kCountThreads / 64; w++) { carry += (w < wave) ? wsum[w] : 0u; n_chunk += wsum[w]; }
            const uint32_t e = carry + incl - (a + b2);
            if (c0 < n_cells) start[c0] = e;
            if (c1 < n_cells) start[c1] = e + a;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kRowsChunk / kCountThreads; u++)
            if (pk[u] != kRowNone) {
                const uint32_t c = pk[u] & 2047u;
                sorted[start[c] + (pk[u] >> 11)] = (uint32_t)(u * kCountThreads + threadIdx.x) | (c << 14);
            }
        __syncthreads();
        // out, in cell order: consecutive threads write consecutive positions of a cell's range
        for (uint32_t k = threadIdx.x; k < n_chunk; k += kCountThreads) {
            const uint32_t e = sorted[k];
            const uint32_t c = e >> 14;
            const int64_t idx = cb + (int64_t)(e & 16383u);
            const uint32_t pos = base[c] + (k - start[c]);
            order[pos] = (int32_t)idx;
            rect_sorted[pos] = rect2[idx];                       // the tile passes read rectangles without a gather
            // (and the per-tile sort its depth keys: bucket entries are positions in order[] then -- bucket_tile_fill_kernel)
            if (depth_sorted) depth_sorted[pos] = depths[idx];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < n_cells; c += kCountThreads) base[c] += cnt[c];
    }
}

// (no visible row at all, or no Gaussian: the publishing part of bucket_rows_kernel alone)
__global__ void bucket_rows_empty_kernel(int n_cells, uint32_t* __restrict__ cell_offs, int64_t* __restrict__ counters,
                                         long long* __restrict__ n_isects_host) {
    for (int c = threadIdx.x; c <= n_cells; c += blockDim.x) cell_offs[c] = 0u;
    if (threadIdx.x == 0) {
        counters[1] = 0;
        if (n_isects_host) __hip_atomic_store(n_isects_host, (long long)counters[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- 3. tile passes over order[] -----------------------------------------------------------------------------
// Shared prologue of the count and the fill pass: load up to 1024 consecutive rows of order[] and choose the
// workgroup's tile WINDOW: the bounding box of the "small" rectangles (<= 16 x 16 tiles) of the first row's camera,
// if (width + 1) * (height + 1) <= kWinMax.  The number of the workgroup's intersections per window tile is obtained
// WITHOUT visiting the intersections: every rectangle adds +1 / -1 at its four corners of a difference array in LDS
// (four LDS atomics per row instead of one per intersection), and a 2-D prefix sum turns that into the counts.
// Rows that take no part in the window (large rectangles, a second camera in the same workgroup, an oversized box)
// touch global memory tile by tile.
struct TileWg {
    uint4 info[kRowsPerWg];         // (fill pass) per row: exclusive scan of the tile counts | x0 | y0 << 16 |
                                    //   rectangle width | depth code << 16 (index mode) | in_window << 31 | bits of 1 / width
    int32_t row[kRowsPerWg];
    int32_t tab[kWinMax];           // difference array -> counts (stride ww + 1) -> (fill pass) cursors
    uint32_t wsum[kRowsPerWg / 64];
    int bb[4];                      // min x, min y, max x, max y of the small rectangles
    uint32_t total;
};

// Loads the rows, publishes rectangles (and, SCAN: the exclusive scan of their tile counts), finds the window and
// leaves the per-tile counts of the window rows in L.tab[(y - wy0) * (ww + 1) + (x - wx0)].  Returns this thread's
// row (or -1), its rectangle and whether it takes part in the window.
// Rows of order[] per workgroup of the tile passes: the grid is sized for ALL rows (the host does not know how many are
// visible), so the visible ones are dealt evenly over it instead of leaving the last workgroups empty -- 977 workgroups
// of 806 rows instead of 770 of 1 024 at 1 M: both tile passes run two balanced rounds.
__device__ __forceinline__ int rows_per_wg(int64_t n_vis) {
    const int64_t per = ((n_vis + gridDim.x - 1) / gridDim.x + 63) & ~(int64_t)63;
    return (int)(per < kRowsPerWg ? per : kRowsPerWg);
}

template <bool SCAN>
__device__ __forceinline__ void tile_wg_prologue(TileWg& L, int64_t n_vis, int n_gauss, const int32_t* __restrict__ order,
                                                 const uint2* __restrict__ rect2, int& cam0, int& wx0, int& wy0, int& ww,
                                                 int& wh, int32_t& my_row, uint2& my_rect, bool& my_in) {
    const int per = rows_per_wg(n_vis);
    const int64_t first = (int64_t)blockIdx.x * per;
    if (threadIdx.x == 0) { L.bb[0] = 0x7fffffff; L.bb[1] = 0x7fffffff; L.bb[2] = -1; L.bb[3] = -1; }
    cam0 = (int)(order[first] / n_gauss);                       // first < n_vis is guaranteed by the caller
    __syncthreads();
    const int e = threadIdx.x;
    const int64_t pos = first + e;
    uint32_t cnt = 0u, xy = 0u, wf = 1u;
    int32_t r = -1;
    uint2 r2 = make_uint2(0u, 0u);
    int mnx = 0x7fffffff, mny = 0x7fffffff, mxx = -1, mxy = -1;
    bool in = false;
    if (e < per && pos < n_vis) {
        r = order[pos];
        r2 = rect2[pos];                                         // rectangles in order[] order (bucket_rows)
        const int w = (int)(r2.y & 0xffffu), h = (int)(r2.y >> 16);
        const int x0 = (int)(r2.x & 0xffffu), y0 = (int)(r2.x >> 16);
        cnt = (uint32_t)(w * h);
        xy = r2.x;
        wf = (uint32_t)w;
        if (w <= kSmallRect && h <= kSmallRect && (int)(r / n_gauss) == cam0) {
            in = true;
            mnx = x0; mny = y0; mxx = x0 + w - 1; mxy = y0 + h - 1;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = 0u;
    if (SCAN) {
        incl = wave_scan_add(cnt);
        if (lane == 63) L.wsum[wave] = incl;
    }
    // wave-level min / max before the LDS atomics
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        mnx = min(mnx, __shfl_xor(mnx, m)); mny = min(mny, __shfl_xor(mny, m));
        mxx = max(mxx, __shfl_xor(mxx, m)); mxy = max(mxy, __shfl_xor(mxy, m));
    }
    if (lane == 0 && mxx >= 0) {
        atomicMin(&L.bb[0], mnx); atomicMin(&L.bb[1], mny); atomicMax(&L.bb[2], mxx); atomicMax(&L.bb[3], mxy);
    }
    __syncthreads();
    wx0 = L.bb[0]; wy0 = L.bb[1];
    ww = L.bb[2] >= 0 ? L.bb[2] - L.bb[0] + 1 : 0;
    wh = L.bb[2] >= 0 ? L.bb[3] - L.bb[1] + 1 : 0;
    if ((int64_t)(ww + 1) * (wh + 1) > kWinMax) { ww = 0; wh = 0; }     // oversized box: everything goes the slow way
    in = in && ww > 0;
    if (SCAN) {
        uint32_t before = incl - cnt, all = 0u;
#pragma unroll
        for (int w = 0; w < kRowsPerWg / 64; w++) { before += (w < wave) ? L.wsum[w] : 0u; all += L.wsum[w]; }
        if (threadIdx.x == 0) L.total = all;
        L.info[e] = make_uint4(before, xy, wf | (in ? 0x80000000u : 0u), __float_as_uint(1.0f / (float)(wf & 0xffffu)));
        L.row[e] = r;
    }
    const int stride = ww + 1;
    const int n_tab = ww > 0 ? stride * (wh + 1) : 0;
    for (int i = threadIdx.x; i < n_tab; i += kRowsPerWg) L.tab[i] = 0;
    __syncthreads();
    if (in) {                                                    // +1 / -1 at the four corners
        const int x0 = (int)(r2.x & 0xffffu) - wx0, y0 = (int)(r2.x >> 16) - wy0;
        const int x1 = x0 + (int)(r2.y & 0xffffu), y1 = y0 + (int)(r2.y >> 16);
        atomicAdd(&L.tab[y0 * stride + x0], 1);
        atomicAdd(&L.tab[y0 * stride + x1], -1);
        atomicAdd(&L.tab[y1 * stride + x0], -1);
        atomicAdd(&L.tab[y1 * stride + x1], 1);
    }
    __syncthreads();
    // 2-D inclusive prefix sum: along x (one thread per window row), then along y (one thread per window column)
    for (int y = threadIdx.x; y < wh; y += kRowsPerWg) {
        int run = 0;
        for (int x = 0; x < ww; x++) { run += L.tab[y * stride + x]; L.tab[y * stride + x] = run; }
    }
    __syncthreads();
    for (int x = threadIdx.x; x < ww; x += kRowsPerWg) {
        int run = 0;
        for (int y = 0; y < wh; y++) { run += L.tab[y * stride + x]; L.tab[y * stride + x] = run; }
    }
    __syncthreads();
    my_row = r; my_rect = r2; my_in = in;
}

__device__ __forceinline__ uint32_t wave_scan_max(uint32_t x) {      // inclusive; lane 63 holds the wave maximum
    x = max(x, dpp_take<0x111, 0xF>(0u, x));
    x = max(x, dpp_take<0x112, 0xF>(0u, x));
    x = max(x, dpp_take<0x114, 0xF>(0u, x));
    x = max(x, dpp_take<0x118, 0xF>(0u, x));
    x = max(x, dpp_take<0x142, 0xA>(0u, x));
    x = max(x, dpp_take<0x143, 0xC>(0u, x));
    return x;
}

__global__ __launch_bounds__(kRowsPerWg) void bucket_tile_count_kernel(
    int n_gauss, int tw, int tiles_per_cam, const int64_t* __restrict__ counters, const int32_t* __restrict__ order,
    const uint2* __restrict__ rect2, int32_t* __restrict__ tile_count) {
    __shared__ TileWg L;
    const int64_t n_vis = counters[1];
    if ((int64_t)blockIdx.x * rows_per_wg(n_vis) >= n_vis) return;
    int cam0, wx0, wy0, ww, wh;
    int32_t r;
    uint2 r2;
    bool in;
    tile_wg_prologue<false>(L, n_vis, n_gauss, order, rect2, cam0, wx0, wy0, ww, wh, r, r2, in);
    if (r >= 0 && !in) {                                         // a row outside the window: tile by tile, global
        const int x0 = (int)(r2.x & 0xffffu), y0 = (int)(r2.x >> 16), w = (int)(r2.y & 0xffffu), h = (int)(r2.y >> 16);
        int32_t* base = tile_count + (r / n_gauss) * tiles_per_cam;
        for (int y = y0; y < y0 + h; y++)
            for (int x = x0; x < x0 + w; x++) atomicAdd(&base[y * tw + x], 1);
    }
    const int stride = ww + 1;
    for (int i = threadIdx.x; i < ww * wh; i += kRowsPerWg) {
        const int y = i / ww, x = i - y * ww;
        const int c = L.tab[y * stride + x];
        if (c) atomicAdd(&tile_count[cam0 * tiles_per_cam + (wy0 + y) * tw + wx0 + x], c);
    }
}

// one workgroup: offsets[0 .. n_tiles] = exclusive scan of the counts (offsets[n_tiles] = total); the counts are
// cleared so that the same buffer serves as the cursors of the fill pass.  Coalesced through LDS chunks.
__global__ __launch_bounds__(1024) void bucket_tile_scan_kernel(int n_tiles, int32_t* __restrict__ tile_count,
                                                                int32_t* __restrict__ offsets) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s, longest_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { carry_s = 0u; longest_s = 0u; }
    __syncthreads();
    uint32_t longest = 0u;
    for (int base = 0; base < n_tiles; base += 4096) {
        // thread t owns tiles base + 4t .. base + 4t + 3 (one 16-byte load)
        const int i0 = base + 4 * threadIdx.x;
        uint32_t c[4];
#pragma unroll
        for (int j = 0; j < 4; j++) c[j] = (i0 + j < n_tiles) ? (uint32_t)tile_count[i0 + j] : 0u;
        const uint32_t s = c[0] + c[1] + c[2] + c[3];
        longest = max(max(longest, max(c[0], c[1])), max(c[2], c[3]));
        const uint32_t incl = wave_scan_add(s);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = carry_s + incl - s, all = 0u;
#pragma unroll
        for (int w = 0; w < 16; w++) { before += (w < wave) ? wsum[w] : 0u; all += wsum[w]; }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (i0 + j < n_tiles) { offsets[i0 + j] = (int32_t)before; tile_count[i0 + j] = 0; }
            before += c[j];
        }
        __syncthreads();
        if (threadIdx.x == 0) carry_s += all;
        __syncthreads();
    }
    longest = wave_scan_max(longest);
    if (lane == 63) atomicMax(&longest_s, longest);
    __syncthreads();
    if (threadIdx.x == 0) {
        offsets[n_tiles] = (int32_t)carry_s;
        offsets[n_tiles + 1] = (int32_t)longest_s;      // the longest bucket: lets the per-tile sort skip unused size classes
    }
}

// fill: payload[offsets[tile] + k] = row (DET: the emission slot cum[row] + q, and isect_gid[slot] = row).
// IDX: payload = the row's POSITION in order[] instead (bucket_rows has written depth_sorted[position] = depths[row]): the
// rows of one tile are neighbours in order[] (cell order), so the per-tile sort's gather of its entries' depths stays inside
// a few hundred KB of depth_sorted -- by row it fetches one cache line per entry from all over depths[] (20 MB at 5 M
// Gaussians: the gather, not the sorting, was what the sort's time went into); row = order[position] where one is needed.
template <bool DET, bool IDX>
__global__ __launch_bounds__(kRowsPerWg) void bucket_tile_fill_kernel(
    int n_gauss, int tw, int tiles_per_cam, const int64_t* __restrict__ counters, const int32_t* __restrict__ order,
    const uint2* __restrict__ rect2, const int32_t* __restrict__ offsets, int32_t* __restrict__ cursors,
    const int64_t* __restrict__ cum, int64_t cap, int32_t* __restrict__ payload, int32_t* __restrict__ isect_gid,
    const float* __restrict__ depth_sorted) {
    __shared__ TileWg L;
    __shared__ uint32_t tbase[kWinMax];
    __shared__ __attribute__((aligned(16))) uint16_t owner[kOwnerChunk];
    const int64_t n_vis = counters[1];
    if ((int64_t)blockIdx.x * rows_per_wg(n_vis) >= n_vis) return;
    int cam0, wx0, wy0, ww, wh;
    int32_t r_;
    uint2 r2_;
    bool in_;
    const int64_t first_pos = (int64_t)blockIdx.x * rows_per_wg(n_vis);
    // (issued BEFORE the prologue's own loads: behind them it was one more exposed memory round trip per workgroup -- a
    // workgroup lives for some 12 us, a handful of dependent round trips under 18 M scattered stores -- and cost the kernel 60 us)
    // (unconditional, from a clamped position: behind a branch the compiler waits for the load where the branch ends)
    float dz_ = 0.f;
    if (IDX) { const int64_t pz = first_pos + threadIdx.x; dz_ = depth_sorted[pz < n_vis ? pz : n_vis - 1]; }
    tile_wg_prologue<true>(L, n_vis, n_gauss, order, rect2, cam0, wx0, wy0, ww, wh, r_, r2_, in_);
    // (index mode: the row's depth code rides in the entry's top bits -- the front kernel of the per-tile sort drops the
    // entries behind its pivot by the code alone; depth_sorted[position] is a coalesced read here)
    // The code travels in spare bits of the row's info word: the expansion below is a chain of LDS round trips per output, and
    // a second array (one more dependent ds_read per output, the row read no longer overlapped) cost the kernel 50 % --
    // 117 -> 180 us at 5 M Gaussians with the same memory traffic (FETCH / WRITE counters equal in both modes).
    if (IDX && r_ >= 0) L.info[threadIdx.x].z |= misplat_internal::depth_code9(dz_) << 16;
    const uint32_t total = L.total;
    const int stride = ww + 1;
    // one returning atomic per (workgroup, tile): a contiguous range of the tile's bucket; tab becomes the cursors
    for (int i = threadIdx.x; i < ww * wh; i += kRowsPerWg) {
        const int y = i / ww, x = i - y * ww;
        const int c = L.tab[y * stride + x];
        const int gt = cam0 * tiles_per_cam + (wy0 + y) * tw + wx0 + x;
        tbase[y * stride + x] = c ? (uint32_t)offsets[gt] + (uint32_t)atomicAdd(&cursors[gt], c) : 0u;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < stride * (wh + 1); i += kRowsPerWg) L.tab[i] = 0;
    __syncthreads();
    // Expansion: output o of the workgroup belongs to the row e with excl[e] <= o < excl[e] + count[e].  Instead of
    // a 10-step binary search per output, the owners of 8192 consecutive outputs are found at once: every row marks
    // its first output with its index, and a max-scan spreads the marks (rows are in ascending output order).
    for (uint32_t base = 0; base < total; base += kOwnerChunk) {
        reinterpret_cast<uint4*>(owner)[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        {
            const int e = threadIdx.x;
            const uint32_t s0 = L.info[e].x;
            const uint32_t s1 = (e + 1 < kRowsPerWg) ? L.info[e + 1].x : total;
            if (s1 > s0) {                                       // a row with at least one output
                if (s0 >= base && s0 < base + kOwnerChunk) owner[s0 - base] = (uint16_t)(e + 1);
                else if (s0 < base && s1 > base) owner[0] = (uint16_t)(e + 1);
            }
        }
        __syncthreads();
        const uint4 pk = reinterpret_cast<const uint4*>(owner)[threadIdx.x];       // 8 entries of 16 bits
        uint32_t own[8] = {pk.x & 0xffffu, pk.x >> 16, pk.y & 0xffffu, pk.y >> 16,
                           pk.z & 0xffffu, pk.z >> 16, pk.w & 0xffffu, pk.w >> 16};
#pragma unroll
        for (int j = 1; j < 8; j++) own[j] = max(own[j], own[j - 1]);
        const uint32_t incl = wave_scan_max(own[7]);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 63) L.wsum[wave] = incl;
        uint32_t before = dpp_take<0x138, 0xF>(0u, incl);        // wave_shr:1 -> the maximum of the lanes before this one
        if (lane == 0) before = 0u;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < kRowsPerWg / 64; w++) before = max(before, (w < wave) ? L.wsum[w] : 0u);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t o = base + 8u * threadIdx.x + j;
            if (o >= total) break;
            const int e = (int)max(own[j], before) - 1;
            const uint4 inf = L.info[e];
            const uint32_t q = o - inf.x;
            const uint32_t w = inf.z & 0xffffu;
            // (q + 0.5) / w is at least 0.5 / w away from an integer: the float quotient truncates exactly
            const uint32_t ry = (uint32_t)(((float)q + 0.5f) * __uint_as_float(inf.w));
            const uint32_t rx = q - ry * w;
            const int tx = (int)((inf.y & 0xffffu) + rx), ty = (int)((inf.y >> 16) + ry);
            int64_t slot;
            if (inf.z >> 31) {
                const int i = (ty - wy0) * stride + (tx - wx0);
                slot = (int64_t)tbase[i] + (int64_t)atomicAdd(&L.tab[i], 1);
            } else {
                const int gt = (L.row[e] / n_gauss) * tiles_per_cam + ty * tw + tx;
                slot = (int64_t)offsets[gt] + (int64_t)atomicAdd(&cursors[gt], 1);
            }
            if (slot < cap) {
                if (DET) {
                    const int32_t r = L.row[e];
                    const int64_t es = cum[r] + (int64_t)q;         // emission slot: the rows of the gradient slab
                    payload[slot] = (int32_t)es;
                    if (es < cap) isect_gid[es] = r;
                } else if (IDX) {                                    // position in order[] | depth code (no row needed)
                    payload[slot] = (int32_t)((uint32_t)(first_pos + e) | (((inf.z >> 16) & 0x1ffu) << misplat_internal::kIdxBits));
                } else {
                    payload[slot] = L.row[e];
                }
            }
        }
        __syncthreads();                                         // owner[] and wsum[] are reused by the next chunk
    }
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH; }

}  // namespace

extern "C" int misplat_bucket_plan(const misplat_params* p, int32_t* n_cells, int32_t* n_blocks) {
    if (!p || p->tile_size != MISPLAT_TILE || p->n_cams < 1 || !n_cells || !n_blocks) return MISPLAT_EINVAL;
    if (p->tile_w > 0xffff || p->tile_h > 0xffff) return MISPLAT_EINVAL;
    const CellGrid g = make_grid(p);
    if (g.n_cells > MISPLAT_BUCKET_MAX_CELLS) return MISPLAT_EINVAL;
    *n_cells = g.n_cells;
    *n_blocks = g.n_blocks;
    return MISPLAT_OK;
}

extern "C" int misplat_bucket_count(const misplat_params* p, const float* means2d, const int32_t* radii,
                                    int32_t* tiles_per_gauss, uint32_t* rect2, uint32_t* cellhist,
                                    uint32_t* cell_count, int64_t* counters, int32_t already_zero,
                                    misplat_stream_t stream) {
    int32_t nc, nb;
    if (misplat_bucket_plan(p, &nc, &nb) != MISPLAT_OK || !cellhist || !cell_count || !counters) return MISPLAT_EINVAL;
    const CellGrid g = make_grid(p);
    const int64_t total = (int64_t)p->n_gauss * p->n_cams;
    if (total > 0 && (!tiles_per_gauss || !rect2)) return MISPLAT_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // cell counts and the two counters start from zero (already_zero: an earlier kernel of the stream cleared them)
    if (!already_zero) {
        if (misplat_internal::fill_bytes(cell_count, sizeof(uint32_t) * (size_t)g.n_cells, 0u, s) != MISPLAT_OK) return MISPLAT_ELAUNCH;
        if (misplat_internal::fill_bytes(counters, 2 * sizeof(int64_t), 0u, s) != MISPLAT_OK) return MISPLAT_ELAUNCH;
    }
    if (total > 0)
        hipLaunchKernelGGL(bucket_count_kernel, dim3(g.n_blocks), dim3(kCountThreads), 0, s, total, p->n_gauss, p->tile_w,
                           p->tile_h, g.shift, g.cells_x, g.cells_x * g.cells_y, g.n_cells, g.rows_per_block, means2d, radii,
                           tiles_per_gauss, (uint2*)rect2, cellhist, cell_count, (unsigned long long*)counters);
    return check_launch();
}

extern "C" int misplat_bucket_rows(const misplat_params* p, const int32_t* tiles_per_gauss, const uint32_t* rect2,
                                   const uint32_t* cellhist, const uint32_t* cell_count, uint32_t* cell_cursor,
                                   uint32_t* cell_offs, int32_t* order, uint32_t* rect_sorted, int64_t* counters,
                                   int32_t* tile_count, int64_t* n_isects_host, int32_t already_zero,
                                   misplat_stream_t stream) {
    return misplat_internal::bucket_rows(p, tiles_per_gauss, rect2, cellhist, cell_count, cell_cursor, cell_offs, order, rect_sorted,
                                         counters, tile_count, n_isects_host, already_zero, nullptr, nullptr, (hipStream_t)stream);
}

int misplat_internal::bucket_rows(const misplat_params* p, const int32_t* tiles_per_gauss, const uint32_t* rect2,
                                  const uint32_t* cellhist, const uint32_t* cell_count, uint32_t* cell_cursor,
                                  uint32_t* cell_offs, int32_t* order, uint32_t* rect_sorted, int64_t* counters,
                                  int32_t* tile_count, int64_t* n_isects_host, int32_t already_zero, const float* depths,
                                  float* depth_sorted, hipStream_t stream) {
    if (depth_sorted && !depths) return MISPLAT_EINVAL;
    int32_t nc, nb;
    if (misplat_bucket_plan(p, &nc, &nb) != MISPLAT_OK || !cell_count || !cell_cursor || !cell_offs || !counters || !tile_count)
        return MISPLAT_EINVAL;
    const CellGrid g = make_grid(p);
    const int64_t total = (int64_t)p->n_gauss * p->n_cams;
    if (total > 0 && (!order || !rect_sorted)) return MISPLAT_EINVAL;
    const int64_t n_tiles = (int64_t)p->tile_w * p->tile_h * p->n_cams;
    if (n_tiles + 1 > 0x7fffffffLL) return MISPLAT_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // the cell cursors and the tile counters of pass 3 start from zero (already_zero: an earlier kernel cleared them)
    if (!(already_zero & 1) && misplat_internal::fill_bytes(cell_cursor, sizeof(uint32_t) * (size_t)g.n_cells, 0u, s) != MISPLAT_OK)
        return MISPLAT_ELAUNCH;
    if (!(already_zero & 2) && misplat_internal::fill_bytes(tile_count, sizeof(int32_t) * (size_t)(n_tiles + 1), 0u, s) != MISPLAT_OK)
        return MISPLAT_ELAUNCH;
    if (total > 0)
        hipLaunchKernelGGL(bucket_rows_kernel, dim3(g.n_blocks), dim3(kCountThreads), 0, s, total, p->n_gauss, g.shift,
                           g.cells_x, g.cells_x * g.cells_y, g.n_cells, g.rows_per_block, tiles_per_gauss,
                           (const uint2*)rect2, cellhist, cell_count, cell_offs, cell_cursor, order, (uint2*)rect_sorted, counters,
                           (long long*)n_isects_host, depths, depth_sorted);
    else
        hipLaunchKernelGGL(bucket_rows_empty_kernel, dim3(1), dim3(256), 0, s, g.n_cells, cell_offs, counters,
                           (long long*)n_isects_host);
    return check_launch();
}

extern "C" int misplat_bucket_tiles(const misplat_params* p, const int32_t* order, const uint32_t* rect_sorted,
                                    const int64_t* counters, int32_t* tile_count, int32_t* offsets, const int64_t* cum,
                                    int64_t cap_isects, int32_t* payload, int32_t* isect_gid, misplat_stream_t stream) {
    return misplat_internal::bucket_tiles(p, order, rect_sorted, counters, tile_count, offsets, cum, cap_isects, payload, isect_gid,
                                          nullptr, (hipStream_t)stream);
}

int misplat_internal::bucket_tiles(const misplat_params* p, const int32_t* order, const uint32_t* rect_sorted,
                                   const int64_t* counters, int32_t* tile_count, int32_t* offsets, const int64_t* cum,
                                   int64_t cap_isects, int32_t* payload, int32_t* isect_gid, const float* depth_sorted,
                                   hipStream_t stream) {
    const bool indexed = depth_sorted != nullptr;
    if (!p || p->tile_size != MISPLAT_TILE || !counters || !tile_count || !offsets || cap_isects < 0 ||
        cap_isects > 0x7fffffffLL || (cum && !isect_gid && cap_isects > 0) ||
        (indexed && (cum || (int64_t)p->n_gauss * p->n_cams > (int64_t)misplat_internal::kIdxMask)))
        return MISPLAT_EINVAL;
    const int64_t total = (int64_t)p->n_gauss * p->n_cams;
    if (total > 0 && (!order || !rect_sorted)) return MISPLAT_EINVAL;
    const int64_t n_tiles = (int64_t)p->tile_w * p->tile_h * p->n_cams;
    const int tiles_per_cam = p->tile_w * p->tile_h;
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = (unsigned)((total + kRowsPerWg - 1) / kRowsPerWg);     // upper bound: workgroups past n_vis leave
    if (grid > 0)
        hipLaunchKernelGGL(bucket_tile_count_kernel, dim3(grid), dim3(kRowsPerWg), 0, s, p->n_gauss, p->tile_w,
                           tiles_per_cam, counters, order, (const uint2*)rect_sorted, tile_count);
    hipLaunchKernelGGL(bucket_tile_scan_kernel, dim3(1), dim3(1024), 0, s, (int)n_tiles, tile_count, offsets);
    if (grid > 0 && cap_isects > 0 && payload) {
        if (cum)
            hipLaunchKernelGGL((bucket_tile_fill_kernel<true, false>), dim3(grid), dim3(kRowsPerWg), 0, s, p->n_gauss, p->tile_w,
                               tiles_per_cam, counters, order, (const uint2*)rect_sorted, offsets, tile_count, cum, cap_isects,
                               payload, isect_gid, depth_sorted);
        else if (indexed)
            hipLaunchKernelGGL((bucket_tile_fill_kernel<false, true>), dim3(grid), dim3(kRowsPerWg), 0, s, p->n_gauss, p->tile_w,
                               tiles_per_cam, counters, order, (const uint2*)rect_sorted, offsets, tile_count, cum, cap_isects,
                               payload, isect_gid, depth_sorted);
        else
            hipLaunchKernelGGL((bucket_tile_fill_kernel<false, false>), dim3(grid), dim3(kRowsPerWg), 0, s, p->n_gauss, p->tile_w,
                               tiles_per_cam, counters, order, (const uint2*)rect_sorted, offsets, tile_count, cum, cap_isects,
                               payload, isect_gid, depth_sorted);
    }
    return check_launch();
}
