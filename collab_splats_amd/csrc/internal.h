// internal.h -- entry points shared between the translation units of libmisplat.so that are NOT part of the C ABI
// (C++ linkage: they do not appear among the exported misplat_* symbols).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "misplat.h"

namespace misplat_internal {

// dst[0 .. n_floats) = 0 with at most max_blocks workgroups of 256 threads (0: as many as it takes).  A small grid makes
// a background fill: beside a compute-bound kernel on another graph branch it trickles along without taking the
// machine away from it.
int zero_fill(float* dst, int64_t n_floats, int max_blocks, hipStream_t s);

// dst[0 .. bytes) = the byte pattern of `word` repeated (word-aligned fills: every 4 bytes hold `word`; a byte fill: pass
// a word of four equal bytes), by a KERNEL.  The library enqueues no memset at all: a memset node inside a captured
// hipGraph was observed not to be applied on replay (ROCm 7.2), and both the library's own graph
// cache and a caller's whole-step capture (graphs.GraphedStep) may record any of these sequences.
int fill_bytes(void* dst, size_t bytes, uint32_t word, hipStream_t s);

// Tensors to clear "in the background" of a compute-bound kernel: the last (at_head: the first) `blocks` workgroups (64
// threads each) of that kernel's grid write the zeros, the others do the kernel's own work.  n[k] floats at p[k]
// (16-byte aligned).  blocks / at_head are chosen by the kernel's launcher.
struct FillList {
    float* p[8];
    int64_t n[8];
    int count, blocks, at_head;
};

// misplat_blend_bwd_atomic that also clears the tensors of `fills` (or NULL) -- inside the compositing kernel's own grid
// where its layout allows (a parallel graph branch costs ~10 us at either end), with plain fills ahead of it otherwise.
int blend_bwd_atomic(const misplat_params* p, int32_t color_dim, const float* Ks, const float* grec, const int32_t* flatten_ids,
                     const int32_t* offsets, int64_t n_isects, const float* alpha, const int32_t* last_ids,
                     const int32_t* median_ids, const float* render, const float* v_render, const float* v_alpha,
                     const float* v_exp_depth, const float* v_med_depth, const float* v_normal, float* v_grec, float* v_abs,
                     int32_t v_grec_is_zero, const FillList* fills, hipStream_t s,
                     bool mean_sums = false /* row slots 0 - 1 = sums the mean2d gradient is linear in (blend_bwd_kernel, MSUM): for
                                               gauss_bwd_sparse with the same flag; v_abs must be NULL */);

// misplat_project_pack_fwd whose on-demand-colour mode (lazy_rows != NULL: colour slots start UNSET) clears the gradient
// rows only when clear_lazy_rows is set -- otherwise blend_fwd_lazy clears the rows it reaches (below).  order_table /
// order_sel (or NULL): the kernel also picks the launch-order record of the call's cameras (misplat_params.unit_sel).
int project_pack_fwd(const misplat_params* p, const float* means, const float* quats, const float* scales,
                     const float* opacities, const float* viewmats, const float* Ks, int32_t* radii, float* means2d,
                     float* depths, float* compensations, float* grec, uint32_t* zero_words, int32_t n_zero,
                     float* lazy_rows, float* abs_rows, int32_t clear_lazy_rows, const int32_t* order_table,
                     int32_t* order_sel, int32_t order_slots, int32_t order_stride, hipStream_t s);

// misplat_unit_order into the record of a view-keyed table that `sel` names (misplat_params.unit_sel).
// unit_reach (or NULL): [units] depths the forward reached -- the record also gets its per-tile pivots (front-only ordering)
// at word MISPLAT_ORDER_HEADER + 8 * ceil(units / 8); stride must cover them.
int unit_order_table(const misplat_params* p, const int32_t* unit_work, int32_t* table, int32_t* sel,
                     int32_t stride, int32_t slots, const float* unit_reach, hipStream_t s);

// misplat_blend_fwd_lazy that also clears row g of rows_on_touch[C*N,16] (or NULL) when it sets the colour of record g.
// N-D records on demand (blend.hip): see the definition.
int blend_fwd_x_lazy(const misplat_params* p, int32_t n_channels, int32_t nxq, const float* Ks, float* grec,
                     const int32_t* flatten_ids, const int32_t* offsets, int64_t n_isects, float* render, float* alpha,
                     float* exp_depth, float* med_depth, float* normal, int32_t* last_ids, int32_t* median_ids, const float* means,
                     const float* viewmats, const float* coeffs, const float* coeffs_rest, int32_t sh_degree, int32_t depth_channel,
                     const float* depths, const float* features, int32_t n_feat, float* rows_on_touch, float* rows_on_touch_x,
                     hipStream_t s);
int blend_fwd_lazy(const misplat_params* p, int32_t color_dim, const float* Ks, float* grec, const int32_t* flatten_ids,
                   const int32_t* offsets, int64_t n_isects, float* render, float* alpha, float* exp_depth, float* med_depth,
                   float* normal, int32_t* last_ids, int32_t* median_ids, const float* means, const float* viewmats,
                   const float* coeffs, const float* coeffs_rest, int32_t sh_degree, int32_t depth_channel,
                   const float* depths, float* sh_aux, float* rows_on_touch, hipStream_t s);

// color_bwd + project_pack_bwd of the rows flagged in misplat_params.touched as ONE launch: one camera, SH colours
// (16 coefficients, no Jacobian cache), no separate mean2d gradient, every output cleared beforehand (FillList above).
// The SH direction gradient stays in registers: no v_means_dir.
int gauss_bwd_sparse(const misplat_params* p, int32_t sh_degree, int32_t depth_slot, const float* means, const float* quats,
                     const float* scales, const float* opacities, const float* viewmats, const float* Ks, const float* coeffs,
                     const float* coeffs_rest, const float* compensations, const float* v_grec, float* v_coeffs,
                     float* v_coeffs_rest, float* v_means, float* v_quats, float* v_scales, float* v_opacities,
                     float* v_means2d_out /* or NULL: [N,2] (cleared like the others), columns 0:2 of the flagged rows */,
                     hipStream_t s, const float* v_featx = nullptr, int32_t nxq = 0,
                     float* v_features = nullptr /* N-D records: [N, n_feat] (cleared like the others), the flagged rows */,
                     int32_t n_feat = 0, int32_t depth_in_featx = 0,
                     bool mean_sums = false /* the rows' slots 0 - 1 are blend_bwd_atomic(mean_sums)'s sums */);

// Bucket entries in index mode (bucket_tiles(indexed)): position in the cell-ordered row list in the low 23 bits, a 9-bit
// MONOTONE code of the row's depth above them (6 bits of the float's exponent from 2^-7 up, 3 bits of mantissa: buckets 9 %
// wide).  The front kernel drops an entry whose code is above the pivot's without touching its depth; everything else masks
// the code off.  Index mode therefore serves up to 2^23 (8.4 M) rows per call.
constexpr uint32_t kIdxBits = 23, kIdxMask = (1u << kIdxBits) - 1u;
__host__ __device__ inline uint32_t depth_code9(float d) {
    uint32_t b;
    __builtin_memcpy(&b, &d, 4);
    if (d != d || (b >> 31)) return 0u;                           // (NaN / negative: never produced for a visible row)
    const int e = (int)(b >> 23) - 120;
    if (e < 0) return 0u;
    if (e > 63) return 511u;
    return ((uint32_t)e << 3) | ((b >> 20) & 7u);
}

// Front-only ordering (csrc/binning.hip, tile_sort_front_kernel): the view-keyed table of misplat_params.unit_sel, whose
// records hold per-tile depth pivots at word pivot_off; front_n[n_tiles] / tile_flag[n_tiles] are written for every tile.
struct FrontSort {
    const int32_t* order_table;
    const int32_t* order_sel;
    int32_t order_slots, order_stride, pivot_off;
    float margin;
    int32_t min_bucket;
    int32_t* front_n;
    int32_t* tile_flag;
};
// (bucket entries are positions in the cell-ordered row list: depths = depth_sorted, row_map = order -- bucket_tiles below)
int tile_sort_front(const int32_t* offsets, int32_t n_tiles, int64_t n_isects, const float* depths, const int32_t* row_map,
                    int32_t* payload, int32_t* flatten_ids, uint32_t* scratch, const FrontSort& F, int64_t est_isects,
                    hipStream_t s);
// misplat_tile_sort with the caller's estimate of the intersection count (or -1): it picks the size classes that get a
// grid of their own; n_isects itself may be a capacity far above the count.
int tile_sort(const int32_t* offsets, int32_t n_tiles_total, int64_t n_isects, int64_t est_isects, const float* depths,
              const int32_t* isect_gid, int32_t* payload, int32_t* flatten_ids, uint32_t* scratch, int32_t flags, hipStream_t s);
int tile_sort_flagged(const int32_t* offsets, int32_t n_tiles, int64_t n_isects, const float* depths, const int32_t* row_map,
                      int32_t* payload, int32_t* flatten_ids, uint32_t* scratch, const int32_t* tile_flag, hipStream_t s);

// misplat_bucket_rows that also writes depth_sorted[position] = depths[row] (both or neither), and misplat_bucket_tiles whose
// payload (indexed; atomic gradient mode only) holds every row's POSITION in order[] instead of the row:
// misplat_tile_sort(flags | 4, isect_gid = order, depths = depth_sorted) then gathers its keys from a few hundred KB around
// the tile instead of from all over depths[].
int bucket_rows(const misplat_params* p, const int32_t* tiles_per_gauss, const uint32_t* rect2, const uint32_t* cellhist,
                const uint32_t* cell_count, uint32_t* cell_cursor, uint32_t* cell_offs, int32_t* order, uint32_t* rect_sorted,
                int64_t* counters, int32_t* tile_count, int64_t* n_isects_host, int32_t already_zero, const float* depths,
                float* depth_sorted, hipStream_t stream);
int bucket_tiles(const misplat_params* p, const int32_t* order, const uint32_t* rect_sorted, const int64_t* counters,
                 int32_t* tile_count, int32_t* offsets, const int64_t* cum, int64_t cap_isects, int32_t* payload,
                 int32_t* isect_gid, const float* depth_sorted /* indexed mode, or NULL */, hipStream_t stream);

// The colour stage with N-D channels (rade_features_model.py:427-476: SH colours + F distilled features, 16 fused channels,
// 17 with RGB+ED).  color_fwd = misplat_color_fwd whose record slot 3 takes slot3[g * slot3_stride] (SH colours only: the
// first feature channel rides in the record); color_fwd_x / color_bwd_x = misplat_color_fwd_x / _bwd_x for a source
// [(C,)N,D] that supplies the fused channels FROM n_pre ON (n_pre = 3 behind SH colours: group 0 is then the colour
// kernel's), clearing the gradient rows the compositing backward will add into (zero_grec [C*N,16], zero_featx [C*N,4 nxq]).
int color_fwd(const misplat_params* p, int32_t sh_degree, int32_t K_or_D, int32_t n_color, int32_t per_cam, int32_t depth_channel,
              const float* means, const float* viewmats, const float* coeffs_or_colors, const float* coeffs_rest,
              const int32_t* radii, const float* depths, float* grec, float* sh_aux, float* zero_rows, const float* slot3,
              int32_t slot3_stride, hipStream_t stream);
int color_fwd_x(const misplat_params* p, int32_t D, int32_t n_pre, int32_t per_cam, int32_t depth_channel, int32_t nxq,
                const float* colors, const int32_t* radii, const float* depths, float* grec, float* featx, float* zero_grec,
                float* zero_featx, hipStream_t stream);
int color_bwd_x(const misplat_params* p, int32_t D, int32_t n_pre, int32_t per_cam, int32_t nxq, const int32_t* radii,
                const float* v_grec, const float* v_featx, float* v_colors, hipStream_t stream);
// misplat_blend_bwd_x_atomic whose fills are skipped for what the forward cleared: zero_flags bit 0 v_grec, 1 v_abs, 2 v_featx.
int blend_bwd_x_atomic(const misplat_params* p, int32_t n_channels, int32_t nxq, const float* Ks, const float* grec,
                       const float* featx, const int32_t* flatten_ids, const int32_t* offsets, int64_t n_isects,
                       const float* alpha, const int32_t* last_ids, const int32_t* median_ids, const float* render,
                       const float* v_render, const float* v_alpha, const float* v_exp_depth, const float* v_med_depth,
                       const float* v_normal, float* v_grec, float* v_featx, float* v_abs, int32_t zero_flags,
                       const FillList* fills /* or NULL: as blend_bwd_atomic */, hipStream_t s,
                       const float* features = nullptr /* featx == NULL: channels 4.. from features [N, n_feat] (+ depths) */,
                       int32_t n_feat = 0, int32_t depth_channel = 0, const float* depths = nullptr,
                       bool mean_sums = false /* as blend_bwd_atomic */);

}  // namespace misplat_internal
