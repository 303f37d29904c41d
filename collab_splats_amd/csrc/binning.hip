// binning.hip -- tile intersection, key emission, radix sort, per-tile offsets, record packing.
// gfx950 only.  SURVEY.md section 8 row a2.3 (inside gsplat-rade's rasterization(), called at
// /root/reference/collab_splats/models/rade_gs_model.py:439-465).  Integer stage: results are
// bit-exact against the CPU restatement (tests/test_parity_gpu.py).
//
// HBM-bound integer/byte work: no MFMA.  key = ((cam*tiles + tile) << 32) | depth bits, value =
// emission slot; the sort only touches the significant bits (32 + ceil(log2(C*tiles))).
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include "misplat.h"

namespace {

__device__ __forceinline__ void tile_rect(float mx, float my, int rxi, int ryi, int tw, int th,
                                          int& x0, int& x1, int& y0, int& y1) {
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

__global__ __launch_bounds__(256) void tile_count_kernel(int64_t total, int tw, int th,
                                                         const float* __restrict__ means2d,
                                                         const int32_t* __restrict__ radii,
                                                         int32_t* __restrict__ tiles_per_gauss) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        int rx = radii[2 * idx], ry = radii[2 * idx + 1];
        int n = 0;
        if (rx > 0 || ry > 0) {
            int x0, x1, y0, y1;
            tile_rect(means2d[2 * idx], means2d[2 * idx + 1], rx, ry, tw, th, x0, x1, y0, y1);
            n = (x1 - x0) * (y1 - y0);
        }
        tiles_per_gauss[idx] = n;
    }
}

__global__ __launch_bounds__(256) void tile_emit_kernel(int64_t total, int n_gauss, int tw, int th,
                                                        const float* __restrict__ means2d,
                                                        const int32_t* __restrict__ radii,
                                                        const float* __restrict__ depths,
                                                        const int64_t* __restrict__ cum,
                                                        uint64_t* __restrict__ keys,
                                                        int32_t* __restrict__ slot_ids,
                                                        int32_t* __restrict__ isect_gid) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        int rx = radii[2 * idx], ry = radii[2 * idx + 1];
        if (!(rx > 0 || ry > 0)) continue;
        int x0, x1, y0, y1;
        tile_rect(means2d[2 * idx], means2d[2 * idx + 1], rx, ry, tw, th, x0, x1, y0, y1);
        const int cam = (int)(idx / n_gauss);
        const uint64_t base = (uint64_t)cam * (uint64_t)(tw * th);
        const uint64_t dbits = (uint64_t)__float_as_uint(depths[idx]);
        int64_t j = cum[idx];
        for (int ty = y0; ty < y1; ty++)
            for (int tx = x0; tx < x1; tx++) {
                keys[j] = ((base + (uint64_t)(ty * tw + tx)) << 32) | dbits;
                slot_ids[j] = (int32_t)j;
                isect_gid[j] = (int32_t)idx;
                j++;
            }
    }
}

// threads 0..n (inclusive): thread i closes every tile in (tile(i-1), tile(i)]
__global__ __launch_bounds__(256) void tile_offsets_kernel(const uint64_t* __restrict__ keys,
                                                           const int32_t* __restrict__ slots,
                                                           const int32_t* __restrict__ isect_gid,
                                                           int64_t n, int n_tiles,
                                                           int32_t* __restrict__ offsets,
                                                           int32_t* __restrict__ flatten_ids) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t cur = (i < n) ? (int64_t)(keys[i] >> 32) : (int64_t)n_tiles - 1;
        int64_t prev = (i > 0) ? (int64_t)(keys[i - 1] >> 32) : -1;
        if (i == n) {
            for (int64_t t = prev + 1; t < n_tiles; t++) offsets[t] = (int32_t)n;
        } else {
            for (int64_t t = prev + 1; t <= cur; t++) offsets[t] = (int32_t)i;
            flatten_ids[i] = isect_gid[slots[i]];
        }
    }
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

// ---- two-stage ordering (what bin_tiles() uses): Gaussians are sorted by (camera, depth) once,
// intersections are emitted in that order, and a STABLE sort on the tile id alone then yields exactly
// the (tile, depth, Gaussian id) order of the one-shot 64-bit key sort, touching 4-byte keys in 2
// radix passes instead of 12-byte pairs in 5-6.
__global__ __launch_bounds__(256) void depth_keys_kernel(int64_t total, int n_gauss, int n_cams,
                                                         const int32_t* __restrict__ radii,
                                                         const float* __restrict__ depths,
                                                         uint64_t* __restrict__ keys, int32_t* __restrict__ ids) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const bool vis = radii[2 * idx] > 0 || radii[2 * idx + 1] > 0;
        const uint64_t cam = vis ? (uint64_t)(idx / n_gauss) : (uint64_t)n_cams;   // culled rows sort last
        keys[idx] = (cam << 32) | (uint64_t)(vis ? __float_as_uint(depths[idx]) : 0u);
        ids[idx] = (int32_t)idx;
    }
}

// rank r of the depth order -> Gaussian row order[r]; cum_ordered = exclusive scan of its tile counts
__global__ __launch_bounds__(256) void tile_emit_ordered_kernel(int64_t total, int n_gauss, int tw, int th,
                                                                const int32_t* __restrict__ order,
                                                                const float* __restrict__ means2d,
                                                                const int32_t* __restrict__ radii,
                                                                const int64_t* __restrict__ cum_ordered,
                                                                uint32_t* __restrict__ tile_ids,
                                                                int32_t* __restrict__ slot_ids,
                                                                int32_t* __restrict__ isect_gid) {
    // slot_ids == NULL: only the Gaussian row is emitted (it is then the sort payload)
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < total;
         r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t idx = order[r];
        int rx = radii[2 * idx], ry = radii[2 * idx + 1];
        if (!(rx > 0 || ry > 0)) continue;
        int x0, x1, y0, y1;
        tile_rect(means2d[2 * idx], means2d[2 * idx + 1], rx, ry, tw, th, x0, x1, y0, y1);
        const uint32_t base = (uint32_t)(idx / n_gauss) * (uint32_t)(tw * th);
        int64_t j = cum_ordered[r];
        for (int ty = y0; ty < y1; ty++)
            for (int tx = x0; tx < x1; tx++) {
                tile_ids[j] = base + (uint32_t)(ty * tw + tx);
                if (slot_ids) slot_ids[j] = (int32_t)j;
                isect_gid[j] = (int32_t)idx;
                j++;
            }
    }
}

// offsets[t] = first sorted position whose tile id is >= t.  Four positions per thread, all loads
// issued before any use.
__global__ __launch_bounds__(256) void tile_offsets32_kernel(const uint32_t* __restrict__ tiles, int64_t n,
                                                             int n_tiles, int32_t* __restrict__ offsets) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 <= n; i0 += 4 * stride) {
        int64_t cur[4], prev[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t i = i0 + u * stride;
            cur[u] = (i < n) ? (int64_t)tiles[i] : (int64_t)n_tiles - 1;
            prev[u] = (i > 0 && i <= n) ? (int64_t)tiles[i - 1] : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t i = i0 + u * stride;
            if (i > n) continue;
            const int64_t hi = (i == n) ? (int64_t)n_tiles - 1 : cur[u];
            for (int64_t t = prev[u] + 1; t <= hi; t++) offsets[t] = (int32_t)i;
        }
    }
}

__global__ __launch_bounds__(256) void isect_ids_kernel(const uint32_t* __restrict__ tiles,
                                                        const int32_t* __restrict__ flatten_ids,
                                                        const float* __restrict__ depths, int64_t n,
                                                        uint64_t* __restrict__ isect_ids) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        isect_ids[i] = ((uint64_t)tiles[i] << 32) | (uint64_t)__float_as_uint(depths[flatten_ids[i]]);
}

// 32-bit depth keys for the single-camera case (culled rows: 0xffffffff, sorted last)
__global__ __launch_bounds__(256) void depth_keys32_kernel(int64_t total, const int32_t* __restrict__ radii,
                                                           const float* __restrict__ depths,
                                                           uint32_t* __restrict__ keys, int32_t* __restrict__ ids) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const bool vis = radii[2 * idx] > 0 || radii[2 * idx + 1] > 0;
        keys[idx] = vis ? __float_as_uint(depths[idx]) : 0xffffffffu;
        ids[idx] = (int32_t)idx;
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

extern "C" int misplat_tile_count(const misplat_params* p, const float* means2d, const int32_t* radii,
                                  int32_t* tiles_per_gauss, misplat_stream_t stream) {
    if (!p || p->tile_size != MISPLAT_TILE) return MISPLAT_EINVAL;
    int64_t total = (int64_t)p->n_gauss * p->n_cams;
    if (total == 0) return MISPLAT_OK;
    hipLaunchKernelGGL(tile_count_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, total,
                       p->tile_w, p->tile_h, means2d, radii, tiles_per_gauss);
    return check_launch();
}

extern "C" int misplat_tile_emit(const misplat_params* p, const float* means2d, const int32_t* radii,
                                 const float* depths, const int64_t* cum, uint64_t* keys,
                                 int32_t* slot_ids, int32_t* isect_gid, misplat_stream_t stream) {
    if (!p || p->tile_size != MISPLAT_TILE) return MISPLAT_EINVAL;
    int64_t total = (int64_t)p->n_gauss * p->n_cams;
    if (total == 0) return MISPLAT_OK;
    hipLaunchKernelGGL(tile_emit_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, total,
                       p->n_gauss, p->tile_w, p->tile_h, means2d, radii, depths, cum, keys, slot_ids, isect_gid);
    return check_launch();
}

extern "C" size_t misplat_sort_workspace_bytes(int64_t n_isects, int32_t end_bit) {
    size_t bytes = 0;
    if (n_isects <= 0) return 16;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                             (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)n_isects, 0u,
                                             (unsigned)end_bit, (hipStream_t) nullptr);
    if (e != hipSuccess) return 0;
    return bytes < 16 ? 16 : bytes;
}

extern "C" int misplat_sort_pairs(void* workspace, size_t workspace_bytes, const uint64_t* keys_in,
                                  uint64_t* keys_out, const int32_t* vals_in, int32_t* vals_out,
                                  int64_t n_isects, int32_t end_bit, misplat_stream_t stream) {
    if (n_isects < 0 || end_bit < 1 || end_bit > 64) return MISPLAT_EINVAL;
    if (n_isects == 0) return MISPLAT_OK;
    size_t need = 0;
    if (rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, (size_t)n_isects, 0u,
                                  (unsigned)end_bit, (hipStream_t)stream) != hipSuccess)
        return MISPLAT_ELAUNCH;
    if (need > workspace_bytes) return MISPLAT_EWORKSPACE;
    hipError_t e = rocprim::radix_sort_pairs(workspace, workspace_bytes, keys_in, keys_out, vals_in, vals_out,
                                             (size_t)n_isects, 0u, (unsigned)end_bit, (hipStream_t)stream);
    return e == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH;
}

extern "C" int misplat_tile_offsets(const uint64_t* keys_sorted, const int32_t* slots_sorted,
                                    const int32_t* isect_gid, int64_t n_isects, int32_t n_tiles_total,
                                    int32_t* offsets, int32_t* flatten_ids, misplat_stream_t stream) {
    if (n_isects < 0 || n_tiles_total < 1) return MISPLAT_EINVAL;
    hipLaunchKernelGGL(tile_offsets_kernel, dim3(grid_for(n_isects + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                       keys_sorted, slots_sorted, isect_gid, n_isects, n_tiles_total, offsets, flatten_ids);
    return check_launch();
}

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

extern "C" int misplat_depth_keys(const misplat_params* p, const int32_t* radii, const float* depths,
                                  uint64_t* keys, int32_t* ids, misplat_stream_t stream) {
    if (!p || p->n_cams < 1) return MISPLAT_EINVAL;
    int64_t total = (int64_t)p->n_gauss * p->n_cams;
    if (total == 0) return MISPLAT_OK;
    hipLaunchKernelGGL(depth_keys_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, total,
                       p->n_gauss, p->n_cams, radii, depths, keys, ids);
    return check_launch();
}

extern "C" int misplat_tile_emit_ordered(const misplat_params* p, const int32_t* order, const float* means2d,
                                         const int32_t* radii, const int64_t* cum_ordered, uint32_t* tile_ids,
                                         int32_t* slot_ids, int32_t* isect_gid, misplat_stream_t stream) {
    if (!p || p->tile_size != MISPLAT_TILE) return MISPLAT_EINVAL;
    int64_t total = (int64_t)p->n_gauss * p->n_cams;
    if (total == 0) return MISPLAT_OK;
    hipLaunchKernelGGL(tile_emit_ordered_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, total,
                       p->n_gauss, p->tile_w, p->tile_h, order, means2d, radii, cum_ordered, tile_ids, slot_ids,
                       isect_gid);
    return check_launch();
}

extern "C" size_t misplat_sort32_workspace_bytes(int64_t n, int32_t end_bit) {
    size_t bytes = 0;
    if (n <= 0) return 16;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                             (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)n, 0u,
                                             (unsigned)end_bit, (hipStream_t) nullptr);
    if (e != hipSuccess) return 0;
    return bytes < 16 ? 16 : bytes;
}

extern "C" int misplat_sort32_pairs(void* workspace, size_t workspace_bytes, const uint32_t* keys_in,
                                    uint32_t* keys_out, const int32_t* vals_in, int32_t* vals_out, int64_t n,
                                    int32_t end_bit, misplat_stream_t stream) {
    if (n < 0 || end_bit < 1 || end_bit > 32) return MISPLAT_EINVAL;
    if (n == 0) return MISPLAT_OK;
    size_t need = 0;
    if (rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                  (unsigned)end_bit, (hipStream_t)stream) != hipSuccess)
        return MISPLAT_ELAUNCH;
    if (need > workspace_bytes) return MISPLAT_EWORKSPACE;
    hipError_t e = rocprim::radix_sort_pairs(workspace, workspace_bytes, keys_in, keys_out, vals_in, vals_out,
                                             (size_t)n, 0u, (unsigned)end_bit, (hipStream_t)stream);
    return e == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH;
}

extern "C" int misplat_tile_offsets32(const uint32_t* tiles_sorted, int64_t n_isects, int32_t n_tiles_total,
                                      int32_t* offsets, misplat_stream_t stream) {
    if (n_isects < 0 || n_tiles_total < 1) return MISPLAT_EINVAL;
    hipLaunchKernelGGL(tile_offsets32_kernel, dim3(grid_for((n_isects + 4) / 4, 256)), dim3(256), 0,
                       (hipStream_t)stream, tiles_sorted, n_isects, n_tiles_total, offsets);
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

extern "C" int misplat_depth_keys32(const misplat_params* p, const int32_t* radii, const float* depths,
                                    uint32_t* keys, int32_t* ids, misplat_stream_t stream) {
    if (!p || p->n_cams != 1) return MISPLAT_EINVAL;
    int64_t total = (int64_t)p->n_gauss;
    if (total == 0) return MISPLAT_OK;
    hipLaunchKernelGGL(depth_keys32_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, total,
                       radii, depths, keys, ids);
    return check_launch();
}
