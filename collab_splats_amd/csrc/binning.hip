// binning.hip -- the per-tile depth sort (one workgroup per tile: bin + rank, or a register-resident LDS radix),
// gsplat's isect_ids on demand, record packing.
// gfx950 only.  SURVEY.md section 8 row a2.3 (inside gsplat-rade's rasterization(), called at
// /root/reference/collab_splats/models/rade_gs_model.py:439-465).  Integer stage: results are
// bit-exact against the CPU restatement (tests/test_parity_gpu.py).  The default bucketing is in
// csrc/bucket.hip; what both feed is misplat_tile_sort below.
//
// HBM-bound integer/byte work: no MFMA.
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "misplat.h"
#include "internal.h"

namespace {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_take(uint32_t ident, uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)ident, (int)x, CTRL, ROW_MASK, 0xF, false);
}
// inclusive wave scan: lane i ends with op(x_0 .. x_i); lane 63 holds the wave total
template <class Op>
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x, uint32_t ident, Op op) {
    x = op(x, dpp_take<0x111, 0xF>(ident, x));      // row_shr:1
    x = op(x, dpp_take<0x112, 0xF>(ident, x));      // row_shr:2
    x = op(x, dpp_take<0x114, 0xF>(ident, x));      // row_shr:4
    x = op(x, dpp_take<0x118, 0xF>(ident, x));      // row_shr:8
    x = op(x, dpp_take<0x142, 0xA>(ident, x));      // row_bcast:15 -> rows 1, 3
    x = op(x, dpp_take<0x143, 0xC>(ident, x));      // row_bcast:31 -> rows 2, 3
    return x;
}




__global__ __launch_bounds__(256) void pack_kernel(int64_t n_rows, int cd,
                                                   const float* __restrict__ means2d,
                                                   const float* __restrict__ conics,
                                                   const float* __restrict__ opac,
                                                   const float* __restrict__ ray_ts,
                                                   const float* __restrict__ ray_planes,
                                                   const float* __restrict__ normals,
                                                   const float* __restrict__ colors,
                                                   float4* __restrict__ grec) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows;
         r += (int64_t)gridDim.x * blockDim.x) {
        float4 q0 = make_float4(means2d[2 * r], means2d[2 * r + 1], conics[3 * r], conics[3 * r + 1]);
        float4 q1 = make_float4(conics[3 * r + 2], opac[r], ray_ts[r], ray_planes[2 * r]);
        float4 q2 = make_float4(ray_planes[2 * r + 1], normals[3 * r], normals[3 * r + 1], normals[3 * r + 2]);
        float c[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < cd; k++) c[k] = colors[(size_t)r * cd + k];
        grec[4 * r + 0] = q0; grec[4 * r + 1] = q1; grec[4 * r + 2] = q2;
        grec[4 * r + 3] = make_float4(c[0], c[1], c[2], c[3]);
    }
}



__global__ __launch_bounds__(256) void isect_ids_kernel(const uint32_t* __restrict__ tiles,
                                                        const int32_t* __restrict__ flatten_ids,
                                                        const float* __restrict__ depths, int64_t n,
                                                        uint64_t* __restrict__ isect_ids) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        isect_ids[i] = ((uint64_t)tiles[i] << 32) | (uint64_t)__float_as_uint(depths[flatten_ids[i]]);
}


// ---- per-tile ordering: after the intersections have been bucketed by tile with a STABLE sort of
// pairs emitted in row order (so every bucket is in ascending row order), ONE workgroup per tile sorts
// its bucket by the 32 depth bits with a stable LSD radix sort (4 passes x 8 bits) -- which yields
// exactly the (tile, depth, Gaussian id) order of the global key sort.  Tile lists are short (hundreds
// to a few thousand entries): they live in LDS, and 8 000+ independent workgroups replace the ~20
// launch-latency-bound kernels of a global 32-bit depth sort of the rows.
// Per pass: per-wave digit histograms (LDS atomics) -> exclusive scan over (digit, wave) -> every wave
// walks its contiguous chunk 64 entries at a time in order, peers found with one __ballot per digit
// bit, rank = popcount of the lower peers (one ballot per digit bit).
// GLOBAL = true: the ping-pong buffers live in global scratch (lists longer than the largest LDS class).
// Register-resident classes: a bucket of n <= 64*WAVES*R entries; wave w owns the contiguous entries
// [w*64R, (w+1)*64R) and lane l holds entries w*64R + r*64 + l (r < R) in registers for the whole sort.
// Keys are the depth bits minus the bucket's minimum, so only bits = 32 - clz(max - min) of them can
// differ: the sort runs ceil(bits / 9) passes of equal width (<= 9 bits, typically 3 passes instead of
// 4 fixed 8-bit ones, 0 for a bucket of equal depths).  LDS holds the exchange buffer (8 B per entry) and
// two [512 digits][WAVES] histograms used alternately (the idle one is cleared during the scan, saving a
// barrier).  The histogram is digit-major, so the exclusive scan over (digit, wave) is a flat scan in
// which every thread owns 8 consecutive counters; wave-level scans and reductions are DPP (row_shr /
// row_bcast), not LDS permutes.
// Tiles a workgroup of a size class walks.  A grid as large as n_tiles: one tile each.  A smaller grid (the
// rare classes): a contiguous chunk per workgroup, first tested in parallel (one tile per thread) so that
// workgroups without a bucket of the class leave after one round of loads instead of a serial walk.
// Which tiles a sort launch looks at (front-only ordering, below): mode 0 -- all; mode 1 -- those the front kernel left
// undone (state = front_n: < 0); mode 2 -- those the compositing flagged (state = tile_flag: != 0).
struct TileSel {
    const int32_t* state;
    int mode;
    __device__ __forceinline__ bool wants(int t) const {
        return mode == 0 || (mode == 1 ? state[t] < 0 : state[t] != 0);
    }
};

__device__ __forceinline__ bool tile_range(const int32_t* __restrict__ offsets, int n_tiles, int64_t n_isects, int lo,
                                           int hi, int& t_first, int& t_last, int& t_step, int has_longest, TileSel sel) {
    // offsets[n_tiles + 1] (when present) is the longest bucket: a size class above it has nothing to do
    if (has_longest && offsets[n_tiles + 1] <= lo) return false;
    if ((int)gridDim.x >= n_tiles) {
        t_first = blockIdx.x; t_last = n_tiles; t_step = gridDim.x;
        return true;
    }
    const int per = (n_tiles + gridDim.x - 1) / gridDim.x;
    t_first = blockIdx.x * per;
    t_last = min(t_first + per, n_tiles);
    t_step = 1;
    bool mine = false;
    for (int t = t_first + threadIdx.x; t < t_last; t += blockDim.x) {
        const int e1 = min(offsets[t + 1], (int)n_isects);
        const int n = e1 - min(offsets[t], e1);
        mine |= (n > lo && n <= hi) && sel.wants(t);
    }
    return __syncthreads_or(mine) != 0;
}

#ifndef MISPLAT_TS_HISTS
#define MISPLAT_TS_HISTS 1            /* 1: one histogram (16 KB of LDS per 1024-entry block -> 8 blocks per CU), cleared after
                                         the scatter at the price of one more barrier per pass: 75 -> 69 us at 1M / 1080p;
                                         2: the idle histogram is cleared during the scan (24 KB, 6 blocks per CU) */
#endif
template <int WAVES, int R>
struct tile_sort_lds {
    static constexpr int CAP = 64 * WAVES * R, DIG = 512;
    uint32_t xk[CAP], xv[CAP];
    __attribute__((aligned(16))) uint32_t hist2[MISPLAT_TS_HISTS][DIG * WAVES];
    uint32_t wsum[WAVES];
    uint32_t kmin, kmax;
};

// Stable ascending sort of the block's n (key, val) pairs held in registers (entry i = wave*64R + r*64 +
// lane).  Every thread of the block calls it; returns the number of radix passes that ran (0: all keys
// equal).  On return the registers hold the sorted pairs and, if a pass ran, L.xk holds the sorted keys
// minus their minimum (what the tie check of the unordered mode compares).
template <int WAVES, int R>
__device__ __forceinline__ int tile_radix_regs(uint32_t (&key)[R], uint32_t (&val)[R], int n,
                                               tile_sort_lds<WAVES, R>& L) {
    constexpr int DIG = 512;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t lt_lo = lane < 32 ? ((1u << lane) - 1u) : 0xffffffffu;
    const uint32_t lt_hi = lane < 32 ? 0u : ((1u << (lane - 32)) - 1u);
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    (void)DIG;
    uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = wave * 64 * R + r * 64 + lane;
        if (i < n) { mn = min(mn, key[r]); mx = max(mx, key[r]); }
    }
    if (threadIdx.x == 0) { L.kmin = 0xffffffffu; L.kmax = 0u; }
    reinterpret_cast<uint4*>(L.hist2[0])[2 * threadIdx.x] = zero4;
    reinterpret_cast<uint4*>(L.hist2[0])[2 * threadIdx.x + 1] = zero4;
    mn = wave_scan_incl(mn, 0xffffffffu, [](uint32_t a, uint32_t b) { return min(a, b); });
    mx = wave_scan_incl(mx, 0u, [](uint32_t a, uint32_t b) { return max(a, b); });
    __syncthreads();
    if (lane == 63 && wave * 64 * R < n) { atomicMin(&L.kmin, mn); atomicMax(&L.kmax, mx); }
    __syncthreads();
    const uint32_t kmin = L.kmin;
    const uint32_t range = L.kmax - kmin;
    const int bits = range ? 32 - __builtin_clz(range) : 0;
    const int passes = (bits + 8) / 9;                            // 0 when every key is equal
    const int dbits = passes ? (bits + passes - 1) / passes : 0;  // <= 9
#pragma unroll
    for (int r = 0; r < R; r++) key[r] -= kmin;                   // idle lanes: never used
    int hb = 0;
    for (int pass = 0; pass < passes; pass++) {
        const int shift = pass * dbits;
        const uint32_t dmask = (1u << dbits) - 1u;
        uint32_t* hist = L.hist2[hb];
        uint32_t* hnext = L.hist2[MISPLAT_TS_HISTS == 2 ? (hb ^ 1) : 0];
        if (MISPLAT_TS_HISTS == 2) hb ^= 1;
        uint32_t plo[R], phi[R];
        // 1. peers of every entry inside its 64-entry round: one ballot per digit bit
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = wave * 64 * R + r * 64 + lane;
            const unsigned long long v = __ballot(i < n);
            plo[r] = (uint32_t)v; phi[r] = (uint32_t)(v >> 32);
        }
        for (int bit = 0; bit < dbits; bit++) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t m = 0u - ((key[r] >> (shift + bit)) & 1u);      // 0 or ~0
                const unsigned long long bb = __ballot(m != 0u);
                plo[r] &= ~((uint32_t)bb ^ m);
                phi[r] &= ~((uint32_t)(bb >> 32) ^ m);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = wave * 64 * R + r * 64 + lane;
            const uint32_t dg = (key[r] >> shift) & dmask;
            if (i < n && ((plo[r] & lt_lo) | (phi[r] & lt_hi)) == 0u) {
                const uint32_t cnt = (uint32_t)(__popc(plo[r]) + __popc(phi[r]));
                if (R == 1) hist[dg * WAVES + wave] = cnt;
                else atomicAdd(&hist[dg * WAVES + wave], cnt);
            }
        }
        __syncthreads();
        // 2. flat exclusive scan of hist[512*WAVES]: 8 consecutive counters per thread
        const uint4 h0 = reinterpret_cast<uint4*>(hist)[2 * threadIdx.x];
        const uint4 h1 = reinterpret_cast<uint4*>(hist)[2 * threadIdx.x + 1];
        const uint32_t tot = h0.x + h0.y + h0.z + h0.w + h1.x + h1.y + h1.z + h1.w;
        const uint32_t incl = wave_scan_incl(tot, 0u, [](uint32_t a, uint32_t b) { return a + b; });
        if (lane == 63) L.wsum[wave] = incl;
        if (MISPLAT_TS_HISTS == 2) {
            reinterpret_cast<uint4*>(hnext)[2 * threadIdx.x] = zero4;
            reinterpret_cast<uint4*>(hnext)[2 * threadIdx.x + 1] = zero4;
        }
        __syncthreads();
        uint32_t e = incl - tot;
#pragma unroll
        for (int w = 0; w < WAVES; w++) e += (w < wave) ? L.wsum[w] : 0u;
        uint4 e0, e1;
        e0.x = e; e += h0.x; e0.y = e; e += h0.y; e0.z = e; e += h0.z; e0.w = e; e += h0.w;
        e1.x = e; e += h1.x; e1.y = e; e += h1.y; e1.z = e; e += h1.z; e1.w = e;
        reinterpret_cast<uint4*>(hist)[2 * threadIdx.x] = e0;
        reinterpret_cast<uint4*>(hist)[2 * threadIdx.x + 1] = e1;
        __syncthreads();
        // 3. scatter through LDS in order: the rounds of one wave share and advance its digit cursors
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = wave * 64 * R + r * 64 + lane;
            const uint32_t dg = (key[r] >> shift) & dmask;
            const uint32_t rank = (uint32_t)(__popc(plo[r] & lt_lo) + __popc(phi[r] & lt_hi));
            if (i < n) {
                const uint32_t base = hist[dg * WAVES + wave];
                if (R > 1 && rank == 0u) hist[dg * WAVES + wave] = base + (uint32_t)(__popc(plo[r]) + __popc(phi[r]));
                L.xk[base + rank] = key[r];
                L.xv[base + rank] = val[r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = wave * 64 * R + r * 64 + lane;
            if (i < n) { key[r] = L.xk[i]; val[r] = L.xv[i]; }
        }
        if (MISPLAT_TS_HISTS == 1) {          // single histogram: clear it now, one more barrier per pass
            reinterpret_cast<uint4*>(hist)[2 * threadIdx.x] = zero4;
            reinterpret_cast<uint4*>(hist)[2 * threadIdx.x + 1] = zero4;
            __syncthreads();
        }
        // (two histograms) the next pass accumulates into hnext (cleared above, a barrier has passed); xk/xv are
        // next written only after two more barriers
    }
#pragma unroll
    for (int r = 0; r < R; r++) key[r] += kmin;
    return passes;
}

// ---- bin + rank: the fast path of the register classes ---------------------------------------------------------
// A bucket of n entries whose keys (depth bits) are spread over [kmin, kmax] does not need three stable LSD passes:
// ONE counting pass on the top B bits of key - kmin (B = log2 of the histogram size = 512 * WAVES >= n bins, so a
// bin holds about one entry when the depths are spread evenly) puts every entry within a few slots of its final
// place, and the entry then RANKS itself inside its bin by comparing (key, value) with the bin's other entries --
// values are unique and ascend with the Gaussian row (rows themselves, or emission slots, which are assigned in row
// order), so this is exactly the (depth, row) order of the stable / two-sort paths, equal depths included.  Nothing
// here needs stability, so the histogram is not split per wave and the positions inside a bin come from a returning
// LDS atomic: 5 barriers per bucket instead of ~13, ~25 instead of ~180 instructions per entry.
// Crowded bins (depths clustered in a small part of the bucket's range: a wall seen through a few floaters) make the
// ranking quadratic; a bucket whose fullest bin exceeds kBinRankCap entries is left to the LSD sort (returns false,
// before anything has been written).
// MAP (how a bucket entry's VALUE relates to its Gaussian row): 0 -- the value is the row; 1 -- an emission slot of the
// deterministic backward, row = isect_gid[value], slots ascend with the row inside a bucket; 2 -- the row's position in the
// cell-ordered row list (bucket_tile_fill_kernel<., IDX>), row = isect_gid[value] with isect_gid = order[], and `depths` is
// indexed by the VALUE (depth_sorted): the gather stays local.  The result is the (depth, row) order in every case.
constexpr int kBinRankCap = 32;
#ifndef MISPLAT_TS_BINRANK
#define MISPLAT_TS_BINRANK 1
#endif
#ifndef MISPLAT_TS_BINRANK_MASK
#define MISPLAT_TS_BINRANK_MASK (1 | 2 | 16)   /* bit = WAVES of the classes that take the fast path (see sort_bucket_regs) */
#endif
#ifndef MISPLAT_TS_B_BLOCKS
#define MISPLAT_TS_B_BLOCKS 6                  /* launch bound (workgroups per 4 SIMDs x ...) of the eight-wave class */
#endif
template <int WAVES, int R, int MAP>
__device__ __forceinline__ bool tile_bin_rank(const uint32_t (&key)[R], const uint32_t (&val)[R], int n,
                                              tile_sort_lds<WAVES, R>& L, int beg, const int32_t* __restrict__ isect_gid,
                                              int32_t* __restrict__ payload, int32_t* __restrict__ flatten_ids) {
    constexpr int DIG = 512 * WAVES;                               // bins (= words of L.hist2[0])
    constexpr int LOG_DIG = WAVES == 1 ? 9 : (WAVES == 2 ? 10 : (WAVES == 4 ? 11 : (WAVES == 8 ? 12 : 13)));
    static_assert((1 << LOG_DIG) == DIG, "WAVES must be 1, 2, 4, 8 or 16");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    uint32_t* hist = L.hist2[0];
    uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = wave * 64 * R + r * 64 + lane;
        if (i < n) { mn = min(mn, key[r]); mx = max(mx, key[r]); }
    }
    if (threadIdx.x == 0) { L.kmin = 0xffffffffu; L.kmax = 0u; }
    reinterpret_cast<uint4*>(hist)[2 * threadIdx.x] = zero4;
    reinterpret_cast<uint4*>(hist)[2 * threadIdx.x + 1] = zero4;
    mn = wave_scan_incl(mn, 0xffffffffu, [](uint32_t a, uint32_t b) { return min(a, b); });
    mx = wave_scan_incl(mx, 0u, [](uint32_t a, uint32_t b) { return max(a, b); });
    __syncthreads();
    if (lane == 63 && wave * 64 * R < n) { atomicMin(&L.kmin, mn); atomicMax(&L.kmax, mx); }
    __syncthreads();
    const uint32_t kmin = L.kmin;
    const uint32_t range = L.kmax - kmin;
    const int bits = range ? 32 - __builtin_clz(range) : 0;
    const int shift = bits > LOG_DIG ? bits - LOG_DIG : 0;
    // 1. count
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = wave * 64 * R + r * 64 + lane;
        if (i < n) atomicAdd(&hist[(key[r] - kmin) >> shift], 1u);
    }
    __syncthreads();
    // 2. flat exclusive scan of the DIG counters (8 consecutive ones per thread) + the fullest bin
    const uint4 h0 = reinterpret_cast<uint4*>(hist)[2 * threadIdx.x];
    const uint4 h1 = reinterpret_cast<uint4*>(hist)[2 * threadIdx.x + 1];
    const uint32_t fullest = max(max(max(h0.x, h0.y), max(h0.z, h0.w)), max(max(h1.x, h1.y), max(h1.z, h1.w)));
    const uint32_t tot = h0.x + h0.y + h0.z + h0.w + h1.x + h1.y + h1.z + h1.w;
    const uint32_t incl = wave_scan_incl(tot, 0u, [](uint32_t a, uint32_t b) { return a + b; });
    if (lane == 63) L.wsum[wave] = incl;
    if (__syncthreads_or(fullest > (uint32_t)kBinRankCap)) return false;     // (uniform; nothing written yet)
    uint32_t e = incl - tot;
#pragma unroll
    for (int w = 0; w < WAVES; w++) e += (w < wave) ? L.wsum[w] : 0u;
    uint4 e0, e1;
    e0.x = e; e += h0.x; e0.y = e; e += h0.y; e0.z = e; e += h0.z; e0.w = e; e += h0.w;
    e1.x = e; e += h1.x; e1.y = e; e += h1.y; e1.z = e; e += h1.z; e1.w = e;
    reinterpret_cast<uint4*>(hist)[2 * threadIdx.x] = e0;
    reinterpret_cast<uint4*>(hist)[2 * threadIdx.x + 1] = e1;
    __syncthreads();
    // 3. scatter: any order inside a bin (the cursors end up at the bins' ends)
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = wave * 64 * R + r * 64 + lane;
        if (i < n) {
            const uint32_t k = key[r] - kmin;
            const uint32_t pos = atomicAdd(&hist[k >> shift], 1u);
            L.xk[pos] = k;
            L.xv[pos] = val[r];
        }
    }
    __syncthreads();
    // 4. rank inside the bin, write out
#pragma unroll 1
    for (int r = 0; r < R; r++) {
        const int i = wave * 64 * R + r * 64 + lane;
        if (i >= n) continue;
        const uint32_t k = L.xk[i], v = L.xv[i];
        const uint32_t d = k >> shift;
        const int s0 = d ? (int)hist[d - 1] : 0, s1 = (int)hist[d];
        int before = 0;
        for (int j = s0; j < s1; j++) {
            const uint32_t kj = L.xk[j], vj = L.xv[j];
            before += (kj < k || (kj == k && (MAP == 2 ? isect_gid[vj] < isect_gid[v] : vj < v))) ? 1 : 0;
        }
        if (payload) payload[beg + s0 + before] = (int32_t)v;
        flatten_ids[beg + s0 + before] = MAP ? isect_gid[v] : (int32_t)v;
    }
    return true;
}

// UNORDERED: the bucket arrives in arbitrary order (filled with atomic cursors, misplat_tile_scatter).  The
// depth sort alone is then only deterministic when all depths differ, so after it neighbours are compared
// and a bucket with ties (rare in a real scene) is re-sorted by row and then, stably, by depth again.
// One bucket [beg, beg + n) of a register class (n <= 64 * WAVES * R), every thread of the workgroup calls it.
// SHORT_PATH: buckets of at most two entries per thread are rank-sorted (the class that starts at one entry).
template <int WAVES, int R, int MAP, bool UNORDERED, bool SHORT_PATH>
__device__ __forceinline__ void sort_bucket_regs(tile_sort_lds<WAVES, R>& L, int beg, int n, const float* __restrict__ depths,
                                                 const int32_t* __restrict__ isect_gid, int32_t* __restrict__ payload,
                                                 int32_t* __restrict__ flatten_ids) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (SHORT_PATH && n <= 128 * WAVES) {
        // Short bucket (at most two entries per thread): rank = number of entries that sort before mine, counted
        // against all n through LDS broadcasts -- n / 2 iterations of a few compares instead of 3 radix passes.
        // Ties break by row (UNORDERED: the (depth, row) order) or by arrival (stable).
        constexpr int T = 64 * WAVES;
        uint32_t v[2], k[2], tb[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int i = threadIdx.x + q * T;
            v[q] = 0u; k[q] = 0xffffffffu; tb[q] = 0xffffffffu;
            if (i < n) {
                v[q] = (uint32_t)payload[beg + i] & (MAP == 2 ? misplat_internal::kIdxMask : 0xffffffffu);
                const int32_t rw = MAP ? isect_gid[v[q]] : (int32_t)v[q];
                k[q] = __float_as_uint(depths[MAP == 2 ? (int32_t)v[q] : rw]);
                tb[q] = UNORDERED ? (uint32_t)rw : (uint32_t)i;
            }
        }
        uint2* pk = reinterpret_cast<uint2*>(L.xk);                 // (key, tie-break) pairs: 4 T <= CAP words (R >= 4)
        __syncthreads();                                            // (the previous bucket's readers are done)
        pk[threadIdx.x] = make_uint2(k[0], tb[0]);
        pk[threadIdx.x + T] = make_uint2(k[1], tb[1]);
        __syncthreads();
        int rank0 = 0, rank1 = 0;
        const uint4* pk2 = reinterpret_cast<const uint4*>(L.xk);
        const int n2 = (n + 1) >> 1;                                // the pad entry (0xffffffff, 0xffffffff) never sorts before
        const bool two = n > T;                                     // (uniform)
        for (int j = 0; j < n2; j++) {
            const uint4 p = pk2[j];
            rank0 += (p.x < k[0] || (p.x == k[0] && p.y < tb[0])) ? 1 : 0;
            rank0 += (p.z < k[0] || (p.z == k[0] && p.w < tb[0])) ? 1 : 0;
            if (two) {
                rank1 += (p.x < k[1] || (p.x == k[1] && p.y < tb[1])) ? 1 : 0;
                rank1 += (p.z < k[1] || (p.z == k[1] && p.w < tb[1])) ? 1 : 0;
            }
        }
        if ((int)threadIdx.x < n) {
            payload[beg + rank0] = (int32_t)v[0];
            flatten_ids[beg + rank0] = MAP ? isect_gid[v[0]] : (int32_t)v[0];
        }
        if ((int)threadIdx.x + T < n) {
            payload[beg + rank1] = (int32_t)v[1];
            flatten_ids[beg + rank1] = MAP ? isect_gid[v[1]] : (int32_t)v[1];
        }
        return;
    }
    uint32_t key[R], val[R];
    int32_t row[R];
    // all loads of one level are issued before the first use (index clamped instead of predicated)
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = wave * 64 * R + r * 64 + lane;
        val[r] = (uint32_t)payload[beg + (i < n ? i : 0)] & (MAP == 2 ? misplat_internal::kIdxMask : 0xffffffffu);
    }
#pragma unroll
    for (int r = 0; r < R; r++) row[r] = MAP == 1 ? isect_gid[val[r]] : (int32_t)val[r];      // (what indexes `depths`)
#pragma unroll
    for (int r = 0; r < R; r++) key[r] = __float_as_uint(depths[row[r]]);
    // (Classes of up to 1 024 entries only.  Measured at 5 M Gaussians / 1080p, typical bucket 3 900 entries: the
    // eight-wave class 556 us with bin + rank against 480 with three LSD passes -- 16 returning LDS atomics per thread on
    // 4 096 random counters and the longer runs of a 12-bit binning cost more than the two passes they replace, and the
    // extra live registers spill under the class's 80-register bound.)
    if (MISPLAT_TS_BINRANK && (MISPLAT_TS_BINRANK_MASK & WAVES)) {
        // (values ascend with the row in both modes -- rows, or emission slots handed out in row order -- so ranking by
        // (depth, value) is the stable order of the ordered mode and the (depth, row) order of the unordered one)
        if (tile_bin_rank<WAVES, R, MAP>(key, val, n, L, beg, isect_gid, payload, flatten_ids)) {
            __syncthreads();                       // (the LDS image is reused by the next bucket of this workgroup)
            return;
        }
        __syncthreads();
    }
    const int passes = tile_radix_regs<WAVES, R>(key, val, n, L);
    unsigned placed = 0u;                          // bit r: entry r was written by the tie pass below
    if (UNORDERED) {
        // Equal depths: with 4 000 entries per bucket (5 M Gaussians at 1080p) a third of the buckets holds a pair of
        // equal floats, and re-sorting the whole bucket twice for it tripled their cost.  A run of equal keys is
        // short: every entry of one looks at its run (the sorted keys and values are still in LDS), takes the slot
        // its row earns inside it and is written there; only a run longer than 8 falls back to the two extra sorts
        // (which then rewrite everything).  Rolled loop over LDS, not over the register arrays: unrolled it cost
        // 45 VGPRs and half the occupancy of the common path.
        bool redo = false;
        if (passes == 0) redo = n > 1;
        else {
#pragma unroll 1
            for (int r = 0; r < R; r++) {
                const int i = wave * 64 * R + r * 64 + lane;
                if (i >= n) continue;
                const uint32_t k = L.xk[i];
                const bool tp = i > 0 && L.xk[i - 1] == k, tn = i + 1 < n && L.xk[i + 1] == k;
                if (!(tp || tn)) continue;
                int s = i, e = i;
                while (s > 0 && i - s < 8 && L.xk[s - 1] == k) s--;
                while (e + 1 < n && e - i < 8 && L.xk[e + 1] == k) e++;
                if (i - s == 8 || e - i == 8) { redo = true; continue; }
                const uint32_t v = L.xv[i];
                const uint32_t mine = MAP ? (uint32_t)isect_gid[v] : v;
                int before = 0;
                for (int j = s; j <= e; j++) {
                    if (j == i) continue;
                    const uint32_t vj = L.xv[j];
                    before += ((MAP ? (uint32_t)isect_gid[vj] : vj) < mine) ? 1 : 0;
                }
                payload[beg + s + before] = (int32_t)v;
                flatten_ids[beg + s + before] = (int32_t)mine;
                placed |= 1u << r;
            }
        }
        if (__syncthreads_or(redo)) {
#pragma unroll
            for (int r = 0; r < R; r++) key[r] = MAP ? (uint32_t)isect_gid[val[r]] : val[r];
            tile_radix_regs<WAVES, R>(key, val, n, L);            // by row
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; r++) key[r] = __float_as_uint(depths[MAP == 2 ? val[r] : key[r]]);
            tile_radix_regs<WAVES, R>(key, val, n, L);            // stably by depth
            placed = 0u;
        }
    }
    if (MAP) {
#pragma unroll
        for (int r = 0; r < R; r++) row[r] = isect_gid[val[r]];
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = wave * 64 * R + r * 64 + lane;
        if (i < n && !((placed >> r) & 1u)) {
            payload[beg + i] = (int32_t)val[r];
            flatten_ids[beg + i] = MAP ? row[r] : (int32_t)val[r];
        }
    }
    __syncthreads();
}

template <int WAVES, int R, int MAP, bool UNORDERED>
__global__ __launch_bounds__(64 * WAVES, (WAVES == 8 && R == 8) ? MISPLAT_TS_B_BLOCKS : (WAVES == 1 ? 4 : 1)) void tile_sort_reg_kernel(const int32_t* __restrict__ offsets, int n_tiles,
                                                                   int64_t n_isects, int lo, int hi,
                                                                   const float* __restrict__ depths,
                                                                   const int32_t* __restrict__ isect_gid,
                                                                   int32_t* __restrict__ payload,
                                                                   int32_t* __restrict__ flatten_ids, int has_longest,
                                                                   TileSel sel) {
    __shared__ tile_sort_lds<WAVES, R> L;
    int t_first, t_last, t_step;
    if (!tile_range(offsets, n_tiles, n_isects, lo, hi, t_first, t_last, t_step, has_longest, sel)) return;
    for (int t = t_first; t < t_last; t += t_step) {
        const int end = min(offsets[t + 1], (int)n_isects);     // offsets: n_tiles + 1 entries; n_isects: buffer capacity
        const int beg = min(offsets[t], end);
        const int n = end - beg;
        if (n <= lo || n > hi || !sel.wants(t)) continue;      // uniform over the block (n >= 1 from here)
        if (lo == 0) sort_bucket_regs<WAVES, R, MAP, UNORDERED, true>(L, beg, n, depths, isect_gid, payload, flatten_ids);
        else sort_bucket_regs<WAVES, R, MAP, UNORDERED, false>(L, beg, n, depths, isect_gid, payload, flatten_ids);
    }
}

// Fallback for buckets longer than the largest register class: same algorithm with fixed 8-bit digits,
// entries walked 64 at a time per wave, ping-pong buffers in global scratch (GLOBAL) or dynamic LDS.
// UNORDERED: four passes over the row bits come first, so the result is the (depth, row) order whatever
// the order of arrival.
// One bucket [beg, beg + n) of any length, ping-pong buffers in global scratch (GLOBAL) or in lds32 behind the histograms.
template <int CAP, int WAVES, int MAP, bool GLOBAL, bool UNORDERED>
__device__ __forceinline__ void sort_bucket_passes(uint32_t* lds32, int beg, int n, const float* __restrict__ depths,
                                                   const int32_t* __restrict__ isect_gid, int32_t* payload,
                                                   int32_t* flatten_ids, uint32_t* scratch) {
    constexpr int THREADS = 64 * WAVES;
    // buffers: keys / values, ping and pong; hist[WAVES][256] always in LDS
    uint32_t* hist = lds32;
    // GLOBAL: the bucket's own ranges of flatten_ids (keys) and payload (values) are one of the two buffers -- the even
    // number of passes ends there -- and scratch[2 * n_isects] holds the other
    uint32_t* buf = GLOBAL ? scratch + (size_t)beg * 2 : lds32 + WAVES * 256;
    const int cap = GLOBAL ? n : CAP;
    uint32_t* k0 = GLOBAL ? reinterpret_cast<uint32_t*>(flatten_ids) + beg : buf;
    uint32_t* k1 = GLOBAL ? buf : buf + cap;
    uint32_t* v0 = GLOBAL ? reinterpret_cast<uint32_t*>(payload) + beg : buf + 2 * cap;
    uint32_t* v1 = GLOBAL ? buf + cap : buf + 3 * cap;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < n; i += THREADS) {
        const int32_t v = (int32_t)((uint32_t)payload[beg + i] & (MAP == 2 ? misplat_internal::kIdxMask : 0xffffffffu));
        const int32_t row = MAP == 1 ? isect_gid[v] : v;             // (what indexes `depths`)
        k0[i] = __float_as_uint(depths[row]);
        v0[i] = (uint32_t)v;
    }
    const int per = ((n + WAVES - 1) / WAVES + 63) & ~63;     // entries per wave, whole rounds
    const int cbeg = min(wave * per, n), cend = min(cbeg + per, n);
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t* kin = k0; uint32_t* vin = v0; uint32_t* kout = k1; uint32_t* vout = v1;
    for (int pass = 0; pass < (UNORDERED ? 8 : 4); pass++) {
        const bool by_row = UNORDERED && pass < 4;
        const int shift = 8 * (pass & 3);
        auto digit_of = [&](uint32_t k, uint32_t v) -> uint32_t {
            const uint32_t src = by_row ? (MAP ? (uint32_t)isect_gid[v] : v) : k;
            return (src >> shift) & 255u;
        };
        for (int b = threadIdx.x; b < WAVES * 256; b += THREADS) hist[b] = 0u;
        if (GLOBAL) __threadfence_block();
        __syncthreads();
        for (int i = cbeg + lane; i < cend; i += 64) atomicAdd(&hist[wave * 256 + digit_of(kin[i], vin[i])], 1u);
        __syncthreads();
        // exclusive scan over (digit major, wave minor): thread d < 256 owns digit d
        uint32_t tot = 0;
        if (threadIdx.x < 256) {
            for (int w = 0; w < WAVES; w++) { const uint32_t h = hist[w * 256 + threadIdx.x]; hist[w * 256 + threadIdx.x] = tot; tot += h; }
        }
        // block-exclusive scan of the 256 digit totals (4 waves of 64: wave scan + carry through LDS slot reuse)
        __shared__ uint32_t wsum[4];
        uint32_t incl = tot;
        if (threadIdx.x < 256) {
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(incl, off);
                if (lane >= off) incl += o;
            }
            if (lane == 63) wsum[wave] = incl;
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            uint32_t carry = 0;
            for (int w = 0; w < wave; w++) carry += wsum[w];
            const uint32_t base = carry + incl - tot;
            for (int w = 0; w < WAVES; w++) hist[w * 256 + threadIdx.x] += base;
        }
        __syncthreads();
        for (int r = cbeg; r < cend; r += 64) {
            const int i = r + lane;
            const bool valid = i < cend;
            const uint32_t key = valid ? kin[i] : 0u;
            const uint32_t val = valid ? vin[i] : 0u;
            const uint32_t dg = valid ? digit_of(key, val) : 0u;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 8; bit++) {
                const bool on = (dg >> bit) & 1u;
                const unsigned long long bb = __ballot(on);
                peers &= on ? bb : ~bb;
            }
            const int rank = __popcll(peers & lt);
            uint32_t pos = 0;
            if (valid) pos = hist[wave * 256 + dg] + (uint32_t)rank;
            __builtin_amdgcn_wave_barrier();
            if (valid && rank == 0) hist[wave * 256 + dg] += (uint32_t)__popcll(peers);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (valid) { kout[pos] = key; vout[pos] = val; }
        }
        if (GLOBAL) __threadfence_block();
        __syncthreads();
        uint32_t* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
    for (int i = threadIdx.x; i < n; i += THREADS) {
        const int32_t v = (int32_t)vin[i];
        payload[beg + i] = v;
        flatten_ids[beg + i] = MAP ? isect_gid[v] : v;
    }
    __syncthreads();
}

// Every size class that has no grid of its own in this launch plan, in ONE launch of 1 024-thread workgroups: a
// workgroup tests its contiguous chunk of tiles in parallel, leaves if none of them is its business, and otherwise sorts
// the buckets of the uncovered classes one after the other -- up to 8 192 entries in registers (16 waves x 8 per lane,
// whatever the bucket's own class), longer ones through global scratch.  `covered`: bit c set = class c (<= 1 024,
// <= 4 096, <= 8 192 entries) is sorted by a launch of its own.  Replaces three near-empty launches of ~5 us each.
__device__ __forceinline__ int size_class(int n) { return n <= 1024 ? 0 : (n <= 4096 ? 1 : (n <= 8192 ? 2 : 3)); }
template <int MAP, bool UNORDERED>
__global__ __launch_bounds__(1024) void tile_sort_rest_kernel(const int32_t* __restrict__ offsets, int n_tiles, int64_t n_isects,
                                                              int covered, const float* __restrict__ depths,
                                                              const int32_t* __restrict__ isect_gid, int32_t* payload,
                                                              int32_t* flatten_ids, uint32_t* scratch, TileSel sel) {
    __shared__ tile_sort_lds<16, 8> L;
    __shared__ uint32_t hist_passes[16 * 256];
    // Tiles are dealt round robin (workgroup b: tiles b, b + grid, ...): the long buckets of a view sit together on the screen --
    // in contiguous chunks a few workgroups sorted all of them one after the other (the rotated views of the bench: 28 us on
    // average for this launch, 5 us on the identity view).
    bool mine = false;
    for (int t = blockIdx.x + (int)threadIdx.x * (int)gridDim.x; t < n_tiles; t += (int)(blockDim.x * gridDim.x)) {
        const int e1 = min(offsets[t + 1], (int)n_isects);
        const int n = e1 - min(offsets[t], e1);
        mine |= n > 0 && !((covered >> size_class(n)) & 1) && sel.wants(t);
    }
    if (!__syncthreads_or(mine)) return;
    for (int t = blockIdx.x; t < n_tiles; t += (int)gridDim.x) {
        const int end = min(offsets[t + 1], (int)n_isects);
        const int beg = min(offsets[t], end);
        const int n = end - beg;
        if (n <= 0) continue;
        const int cls = size_class(n);
        if (((covered >> cls) & 1) || !sel.wants(t)) continue;      // (uniform over the block)
        if (cls < 3) sort_bucket_regs<16, 8, MAP, UNORDERED, false>(L, beg, n, depths, isect_gid, payload, flatten_ids);
        else sort_bucket_passes<0, 16, MAP, true, UNORDERED>(hist_passes, beg, n, depths, isect_gid, payload, flatten_ids, scratch);
    }
}

inline int grid_for(int64_t n, int block) {
    int64_t b = (n + block - 1) / block;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}
inline int check_launch() { return hipGetLastError() == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH; }

}  // namespace






extern "C" int misplat_pack(int64_t n_rows, int32_t color_dim, const float* means2d, const float* conics,
                            const float* opacities_eff, const float* ray_ts, const float* ray_planes,
                            const float* normals, const float* colors, float* grec,
                            misplat_stream_t stream) {
    if (n_rows < 0 || color_dim < 1 || color_dim > 4) return MISPLAT_EINVAL;
    if (n_rows == 0) return MISPLAT_OK;
    hipLaunchKernelGGL(pack_kernel, dim3(grid_for(n_rows, 256)), dim3(256), 0, (hipStream_t)stream, n_rows,
                       color_dim, means2d, conics, opacities_eff, ray_ts, ray_planes, normals, colors,
                       (float4*)grec);
    return check_launch();
}



extern "C" int misplat_isect_ids(const uint32_t* tiles_sorted, const int32_t* flatten_ids, const float* depths,
                                 int64_t n_isects, uint64_t* isect_ids, misplat_stream_t stream) {
    if (n_isects < 0) return MISPLAT_EINVAL;
    if (n_isects == 0) return MISPLAT_OK;
    hipLaunchKernelGGL(isect_ids_kernel, dim3(grid_for(n_isects, 256)), dim3(256), 0, (hipStream_t)stream,
                       tiles_sorted, flatten_ids, depths, n_isects, isect_ids);
    return check_launch();
}


// ---- front-only ordering (dense scenes) --------------------------------------------------------------------------------
// A dense scene is mostly a hidden scene: at 5 M Gaussians / 1080p the compositing traverses 9 % of the 32 M bucket
// entries the per-tile sort orders (the rest lies behind the depth at which every pixel of the tile has saturated).  A
// training loop revisits its cameras, so the view-keyed record that holds a view's launch order (misplat_params.unit_sel)
// also holds, per tile, the DEPTH its last visit reached (the deepest list entry a band looked at; +inf when a band ran
// through its whole list).  This kernel reads a tile's bucket once, keeps the entries at or in front of that pivot (x a
// margin) -- ONE compare per entry: no histogram, no LDS atomic, no barrier for the nine entries in ten behind it --,
// sorts the survivors (bin + rank: exactly the (depth, row) order, they are the first nf entries of the fully sorted list)
// and writes them (their rows) to the head of the tile's range of flatten_ids; front_n[tile] = nf.  Entries are positions in
// the cell-ordered row list (MAP 2: depths indexed by the entry, row = row_map[entry]).  payload is NOT touched: it stays a
// permutation of the bucket, from which a later launch can sort the whole tile -- the compositing forward flags a tile
// whose pixels are still alive at the end of a truncated list (tile_flag), the flagged tiles are sorted in full and
// composited again, and meta["flatten_ids"] is completed the same way when someone asks for it.  Exact results always;
// the pivot only decides how much is sorted.
// A tile the kernel does not take (no record for these cameras yet, a short bucket, an infinite pivot, more survivors
// than the 2 048 it holds, depths too clustered for bin + rank) gets front_n = -1: the regular size-class kernels, which
// skip the tiles that are done, sort it in full.
constexpr int kFrontWaves = 4, kFrontR = 8, kFrontCap = 64 * kFrontWaves * kFrontR;
constexpr int kFrontU = 8;             // bucket entries per thread and round of the scan: all their loads in flight together
__global__ __launch_bounds__(64 * kFrontWaves) void tile_sort_front_kernel(
    const int32_t* __restrict__ offsets, int n_tiles, int64_t n_isects, const float* __restrict__ depths,
    const int32_t* __restrict__ row_map, const int32_t* __restrict__ payload, int32_t* __restrict__ flatten_ids,
    misplat_internal::FrontSort F) {
    __shared__ tile_sort_lds<kFrontWaves, kFrontR> L;
    __shared__ int s_nf;
    constexpr int T = 64 * kFrontWaves;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t* rec = nullptr;
    if (F.order_sel[1] != 0 && (unsigned)F.order_sel[0] < (unsigned)F.order_slots) {
        rec = F.order_table + (size_t)F.order_sel[0] * F.order_stride;
        if (rec[3] != 1) rec = nullptr;                          // (a record written without pivots)
    }
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int end = min(offsets[t + 1], (int)n_isects);
        const int beg = min(offsets[t], end);
        const int n = end - beg;
        float pivot = 0.f;
        bool take = rec != nullptr && n >= F.min_bucket;
        if (take) {
            pivot = __int_as_float(rec[F.pivot_off + t]) * F.margin;
            take = pivot < __builtin_inff() && pivot == pivot;   // (+inf: the last visit needed the whole list)
        }
        const uint32_t qp = misplat_internal::depth_code9(pivot);   // (monotone: d <= pivot implies code(d) <= code(pivot))
        if (!take) {                                             // (uniform over the block)
            if (threadIdx.x == 0) { F.front_n[t] = -1; F.tile_flag[t] = 0; }
            continue;
        }
        if (threadIdx.x == 0) s_nf = 0;
        __syncthreads();
        for (int base = 0; base < n; base += kFrontU * T) {
            int32_t v[kFrontU];
            float d[kFrontU];
#pragma unroll
            for (int u = 0; u < kFrontU; u++) { const int i = base + u * T + (int)threadIdx.x; v[u] = payload[beg + (i < n ? i : 0)]; }
            // an entry whose depth code is above the pivot's lies behind it: dropped without fetching its depth (nine in ten)
#pragma unroll
            for (int u = 0; u < kFrontU; u++) {
                const int i = base + u * T + (int)threadIdx.x;
                const bool cand = i < n && ((uint32_t)v[u] >> misplat_internal::kIdxBits) <= qp;
                v[u] = (int32_t)((uint32_t)v[u] & misplat_internal::kIdxMask);
                d[u] = cand ? depths[v[u]] : __builtin_inff();
            }
#pragma unroll
            for (int u = 0; u < kFrontU; u++) {
                const bool keep = d[u] <= pivot;
                const unsigned long long m = __ballot(keep);
                if (m == 0ull) continue;                         // (uniform over the wave)
                int pos0 = 0;
                if (lane == 0) pos0 = atomicAdd(&s_nf, __popcll(m));
                pos0 = __shfl(pos0, 0);
                const int pos = pos0 + __popcll(m & ((1ull << lane) - 1ull));
                if (keep && pos < kFrontCap) { L.xk[pos] = __float_as_uint(d[u]); L.xv[pos] = (uint32_t)v[u]; }
            }
        }
        __syncthreads();
        const int nf = s_nf;
        bool done = nf <= kFrontCap;
        if (done && nf > 0) {
            uint32_t key[kFrontR], val[kFrontR];
#pragma unroll
            for (int r = 0; r < kFrontR; r++) {
                const int i = wave * 64 * kFrontR + r * 64 + lane;
                key[r] = i < nf ? L.xk[i] : 0u; val[r] = i < nf ? L.xv[i] : 0u;
            }
            __syncthreads();                                     // (bin + rank reuses xk / xv)
            done = tile_bin_rank<kFrontWaves, kFrontR, 2>(key, val, nf, L, beg, row_map, nullptr, flatten_ids);
        }
        __syncthreads();                                         // (the LDS image is reused by the next tile)
        if (threadIdx.x == 0) { F.front_n[t] = done ? nf : -1; F.tile_flag[t] = 0; }
    }
}

// Sort every tile's bucket (arbitrary order on entry) into (depth, row) order.  payload: in/out
// (rows, or emission slots when isect_gid != NULL); flatten_ids: out (rows in final order);
// scratch: 2 * n_isects uint32, only touched by tiles longer than the largest LDS class (8192 entries).
// sel: which tiles (TileSel); rest_only: everything through the one launch of tile_sort_rest_kernel (few tiles expected).
template <int MAP, bool UNORDERED>
static int launch_tile_sort(const int32_t* offsets, int32_t n_tiles, int64_t n_isects, const float* depths,
                            const int32_t* isect_gid, int32_t* payload, int32_t* flatten_ids, uint32_t* scratch,
                            int has_longest, hipStream_t s, TileSel sel = TileSel{nullptr, 0}, bool rest_only = false,
                            int64_t est_isects = -1) {
    // size classes (entries per bucket): <=1024 | <=4096 | <=8192 | longer (global scratch).  Every class walks
    // all tiles and skips buckets of the other classes.  A class that the typical bucket (n_isects / n_tiles)
    // can reach gets one workgroup per tile; the others get a small grid whose workgroups test their chunk of
    // tiles in parallel first (tile_range), so an unused class costs a few microseconds.
    // (n_isects may be the CAPACITY of a speculative launch, about 1.25 x the real count: 4 / 5 of it is the estimate)
    // est_isects >= 0: the caller's own estimate of the count (a capacity may be far above it: ops.py keeps capacities put)
    const int64_t avg = est_isects >= 0 ? est_isects / n_tiles : n_isects / n_tiles * 4 / 5;
    // (behind the front kernel most tiles are done: a workgroup per tile would be ~8 000 workgroups of up to 1 024 threads
    // and 64 KB of LDS that start, read two words and leave -- 16 dispatch rounds, 31 us at 5 M Gaussians; 1 024 workgroups
    // test their 8 tiles in parallel and leave at once, or sort the few that are left; a view's first visit, when ALL are left, runs ~1.5 x slower here)
    const int full = sel.mode == 1 ? (n_tiles < 1024 ? n_tiles : 1024) : (n_tiles < 65536 ? n_tiles : 65536);
    const int few = n_tiles < 512 ? n_tiles : 512;
    // A class the typical bucket (n_isects / n_tiles) can reach gets a grid of its own, one workgroup per tile; every other
    // class goes through ONE more launch (tile_sort_rest_kernel): a 1 M scene is two launches (was four), a 5 M one three.
    int covered = 0;
    // <= 1024 entries.  Sparse scenes (typical bucket under 256 entries: 100 k Gaussians at 1080p): ONE wavefront per
    // bucket, 16 entries per lane, no barrier ever waits for another wave (0.496 -> 0.471 ms per step there); dense ones:
    // two waves, 8 entries per lane (measured at 1 M, typical bucket 800: one wave x 16 +12 us, four waves x 4 +5 us).
    if (avg < 2048 && !rest_only) {
        covered |= 1;
        if (avg < 256)
            hipLaunchKernelGGL((tile_sort_reg_kernel<1, 16, MAP, UNORDERED>), dim3(full), dim3(64), 0, s, offsets, n_tiles,
                               n_isects, 0, 1024, depths, isect_gid, payload, flatten_ids, has_longest, sel);
        else
            hipLaunchKernelGGL((tile_sort_reg_kernel<2, 8, MAP, UNORDERED>), dim3(full), dim3(128), 0, s, offsets, n_tiles,
                               n_isects, 0, 1024, depths, isect_gid, payload, flatten_ids, has_longest, sel);
    }
    if (avg >= 1024 && !rest_only) {
        covered |= 2;
        hipLaunchKernelGGL((tile_sort_reg_kernel<8, 8, MAP, UNORDERED>), dim3(full), dim3(512), 0, s, offsets, n_tiles,
                           n_isects, 1024, 4096, depths, isect_gid, payload, flatten_ids, has_longest, sel);
    }
    if (avg >= 2048 && !rest_only) {
        covered |= 4;
        hipLaunchKernelGGL((tile_sort_reg_kernel<16, 8, MAP, UNORDERED>), dim3(full), dim3(1024), 0, s, offsets, n_tiles,
                           n_isects, 4096, 8192, depths, isect_gid, payload, flatten_ids, has_longest, sel);
    }
    hipLaunchKernelGGL((tile_sort_rest_kernel<MAP, UNORDERED>), dim3(few), dim3(1024), 0, s, offsets, n_tiles, n_isects,
                       covered, depths, isect_gid, payload, flatten_ids, scratch, sel);
    return check_launch();
}

// Front-only ordering, first part: the front kernel, then the regular size classes for the tiles it left undone.
int misplat_internal::tile_sort_front(const int32_t* offsets, int32_t n_tiles, int64_t n_isects, const float* depths,
                                      const int32_t* row_map, int32_t* payload, int32_t* flatten_ids, uint32_t* scratch,
                                      const FrontSort& F, int64_t est_isects, hipStream_t s) {
    if (n_isects < 0 || n_tiles < 1 || n_isects > 0x7fffffffLL || !scratch || !row_map || !F.order_table || !F.order_sel || !F.front_n ||
        !F.tile_flag || F.order_slots < 1 || F.pivot_off < MISPLAT_ORDER_HEADER || F.order_stride < F.pivot_off + n_tiles)
        return MISPLAT_EINVAL;
    const int grid = n_tiles < 65536 ? n_tiles : 65536;
    hipLaunchKernelGGL(tile_sort_front_kernel, dim3(grid), dim3(64 * kFrontWaves), 0, s, offsets, n_tiles, n_isects, depths,
                       row_map, (const int32_t*)payload, flatten_ids, F);
    if (n_isects == 0) return check_launch();
    return launch_tile_sort<2, true>(offsets, n_tiles, n_isects, depths, row_map, payload, flatten_ids, scratch, 1, s,
                                     TileSel{F.front_n, 1}, false, est_isects);
}

// Second part: the tiles whose flag the compositing forward set, in full (one launch; few tiles are expected).
int misplat_internal::tile_sort_flagged(const int32_t* offsets, int32_t n_tiles, int64_t n_isects, const float* depths,
                                        const int32_t* row_map, int32_t* payload, int32_t* flatten_ids, uint32_t* scratch,
                                        const int32_t* tile_flag, hipStream_t s) {
    if (n_isects < 0 || n_tiles < 1 || n_isects > 0x7fffffffLL || !scratch || !tile_flag || !row_map) return MISPLAT_EINVAL;
    if (n_isects == 0) return MISPLAT_OK;
    return launch_tile_sort<2, true>(offsets, n_tiles, n_isects, depths, row_map, payload, flatten_ids, scratch, 1, s,
                                     TileSel{tile_flag, 2}, true);
}

extern "C" int misplat_tile_sort(const int32_t* offsets, int32_t n_tiles_total, int64_t n_isects,
                                 const float* depths, const int32_t* isect_gid, int32_t* payload,
                                 int32_t* flatten_ids, uint32_t* scratch, int32_t flags,
                                 misplat_stream_t stream) {
    return misplat_internal::tile_sort(offsets, n_tiles_total, n_isects, -1, depths, isect_gid, payload, flatten_ids, scratch, flags,
                                       (hipStream_t)stream);
}

int misplat_internal::tile_sort(const int32_t* offsets, int32_t n_tiles_total, int64_t n_isects, int64_t est_isects,
                                const float* depths, const int32_t* isect_gid, int32_t* payload, int32_t* flatten_ids,
                                uint32_t* scratch, int32_t flags, hipStream_t stream) {
    if (n_isects < 0 || n_tiles_total < 1 || n_isects > 0x7fffffffLL || !scratch || (flags & ~7) || ((flags & 4) && !isect_gid))
        return MISPLAT_EINVAL;
    if (n_isects == 0) return MISPLAT_OK;
    hipStream_t s = (hipStream_t)stream;
    const int has_longest = (flags >> 1) & 1;      // (bit 0, "unordered", is what every bucket is now: accepted, ignored)
    const TileSel all{nullptr, 0};
    if (flags & 4)                                 // entries are positions in the cell-ordered row list (see MAP)
        return launch_tile_sort<2, true>(offsets, n_tiles_total, n_isects, depths, isect_gid, payload, flatten_ids, scratch,
                                         has_longest, s, all, false, est_isects);
    if (isect_gid)
        return launch_tile_sort<1, true>(offsets, n_tiles_total, n_isects, depths, isect_gid, payload, flatten_ids, scratch,
                                         has_longest, s, all, false, est_isects);
    return launch_tile_sort<0, true>(offsets, n_tiles_total, n_isects, depths, isect_gid, payload, flatten_ids, scratch,
                                     has_longest, s, all, false, est_isects);
}
